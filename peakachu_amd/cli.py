"""Command line of the MI355X scoring path.

Sub-commands, flags and defaults are those of the reference's launcher
(scripts/peakachu:5-93) for `score_chromosome`, `score_genome` and `pool`, so existing
command lines keep working; help texts are this build's own.  `pool` is host-only (the
reference runs it on the CPU as well).  `train` and `depth` are outside this build.
"""
import argparse
import importlib
import sys


def _lazy(module):
    """`main` of a sub-command's module, imported when the command runs: `pool` pulls in
    scipy.signal (0.7 s), which the scoring commands never use."""
    def main(args):
        return importlib.import_module("." + module, __package__).main(args)
    return main

# (flags, keyword arguments) per option group; a group is attached to the listed commands
_COMMON_SCORING = [
    (("-p", "--path"), dict(help="contact map: .cool, .mcool::/resolutions/<binsize> or a .pkmap.npz container")),
    (("--clr-weight-name",), dict(default="weight",
                                  help="bin-weight column used for balancing; 'raw' scores the raw counts")),
    (("-m", "--model"), dict(type=str, help="model: pickled sklearn forest or flat-forest .npz")),
    (("-l", "--lower"), dict(type=int, default=6, help="smallest bin distance scored")),
    (("-u", "--upper"), dict(type=int, default=300, help="largest bin distance scored")),
    (("--minimum-prob",), dict(type=float, default=0.5,
                               help="report pixels whose probability exceeds this value")),
    (("-O", "--output"), dict(help="bedpe file to write")),
]
_RESOLUTION = (("-r", "--resolution"), dict(type=int, default=10000, help="bin size in bp"))


def _build_parser():
    parser = argparse.ArgumentParser(
        prog="peakachu-amd", description="Peakachu scoring on MI355X (HIP); loop calling on the host.",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    sub = parser.add_subparsers(dest="subcommands")
    commands = {
        "score_chromosome": (_lazy("score_chromosome"), "score the candidate pixels of one chromosome"),
        "score_genome": (_lazy("score_genome"), "score every selected chromosome of a contact map"),
        "pool": (_lazy("call_loops"), "cluster scored pixels into loop calls"),
    }
    parsers = {}
    for name, (func, text) in commands.items():
        parsers[name] = sub.add_parser(name, help=text)
        parsers[name].set_defaults(func=func)
        parsers[name].add_argument(*_RESOLUTION[0], **_RESOLUTION[1])
    for name in ("score_chromosome", "score_genome"):
        for flags, kw in _COMMON_SCORING:
            parsers[name].add_argument(*flags, **kw)
    parsers["score_chromosome"].add_argument("-C", "--chrom", help="label of the chromosome to score")
    parsers["score_genome"].add_argument(
        "-C", "--chroms", nargs="*", default=["#", "X"],
        help="chromosomes to score; '#' selects all numbered ones, no value selects everything")
    parsers["pool"].add_argument("-i", "--infile", help="bedpe written by score_chromosome / score_genome")
    parsers["pool"].add_argument("-o", "--outfile", help="bedpe of loop calls to write")
    parsers["pool"].add_argument("-t", "--threshold", type=float, default=0.9,
                                 help="pixels below this probability are ignored")
    return parser, tuple(commands)


def getargs(argv=None):
    parser, names = _build_parser()
    commands = sys.argv[1:] if argv is None else list(argv)
    # like the reference: nothing, or a bare sub-command, prints help
    if not commands or (len(commands) == 1 and commands[0] in names):
        commands.append("-h")
    return parser.parse_args(commands), commands


def run(argv=None):
    args, commands = getargs(argv)
    if commands[0] in ("-h", "--help"):
        return
    if args.subcommands == "score_genome" and argv is None:
        # the command line as the reference documents it, on a box with several GPUs: one copy of
        # this command per GPU (children; decided before anything has touched HIP), chromosomes
        # dealt to them.  PK_DEVICES=<n> limits the fan-out, PK_NO_SPAWN=1 switches it off; under a
        # launcher that has set WORLD_SIZE this process IS a rank.
        from . import launch
        n = launch.wanted_ranks()
        if n > 1:
            raise SystemExit(launch.spawn(n))
    args.func(args)
    _stage_report()


def _stage_report():
    """PK_STAGE_TIMES=1 PK_STAGE_REPORT=<file>: the wall time per stage of this run, as JSON."""
    import os
    from . import stagetime
    path = os.environ.get("PK_STAGE_REPORT")
    if stagetime.ENABLED and path:
        import json
        with open(path, "w") as fh:
            json.dump(stagetime.report(), fh)


if __name__ == "__main__":
    run()
