"""Command line of the MI355X scoring path: the `score_chromosome`,
`score_genome` and `pool` sub-commands of `scripts/peakachu`
(scripts/peakachu:5-93) with the same flags and defaults.  `pool` is host-only
(the reference runs it on the CPU as well).  The reference's other sub-commands
(train, depth) are outside this build's scope."""
import argparse
import sys

from . import call_loops, score_chromosome, score_genome


def getargs(argv=None):
    parser = argparse.ArgumentParser(description='''Unveil Hi-C Anchors and Peaks (MI355X scoring path).''',
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    subparsers = parser.add_subparsers(dest='subcommands')
    subchrom = subparsers.add_parser('score_chromosome',
                                     help='''Calculate interaction probability per pixel for a chromosome''')
    subchrom.set_defaults(func=score_chromosome.main)
    subgen = subparsers.add_parser('score_genome',
                                   help='''Calculate interaction probability per pixel for the whole genome''')
    subgen.set_defaults(func=score_genome.main)
    subpool = subparsers.add_parser('pool',
                                    help='Print centroid loci from score_genome/score_chromosome output')
    subpool.set_defaults(func=call_loops.main)
    for i in (subchrom, subgen, subpool):
        i.add_argument('-r', '--resolution', help='Resolution in bp (default 10000)',
                       type=int, default=10000)
    for i in (subchrom, subgen):
        i.add_argument('-p', '--path', help='Path to a .cool URI string (or a .pkmap.npz container)')
        i.add_argument('--clr-weight-name', default='weight',
                       help='''The name of the weight column in your Cooler URI for normalizing
                       the contact signals. Specify it to "raw" if you want to use the raw signals.''')
    subchrom.add_argument('-C', '--chrom', help='''Chromosome label. Only contact data within the
                          specified chromosome will be considered.''')
    subgen.add_argument('-C', '--chroms', nargs='*', default=['#', 'X'],
                        help='List of chromosome labels. Only contact data within the specified '
                        'chromosomes will be included. Specially, "#" stands for chromosomes '
                        'with numerical labels. "--chroms" with zero argument will include '
                        'all chromosome data. (default "#" X)')
    for i in (subchrom, subgen):
        i.add_argument('-m', '--model', type=str,
                       help='''Path to pickled model file (or a flat-forest .npz).''')
        i.add_argument('-l', '--lower', type=int, default=6,
                       help='''Lower bound of distance between loci in bins (default 6).''')
        i.add_argument('-u', '--upper', type=int, default=300,
                       help='''Upper bound of distance between loci in bins (default 300).''')
        i.add_argument('--minimum-prob', type=float, default=0.5,
                       help='''Only output pixels with probability score greater than this value (default 0.5)''')
        i.add_argument('-O', '--output', help='Output file name.')
    subpool.add_argument('-i', '--infile',
                         help='Path to the bedpe file outputted from score_chromosome or score_genome')
    subpool.add_argument('-o', '--outfile', help='Output file name.')
    subpool.add_argument('-t', '--threshold', type=float, default=0.9,
                         help='Probability threshold applied before peak calling (default 0.9)')
    commands = sys.argv[1:] if argv is None else list(argv)
    if ((not commands) or ((commands[0] in ['score_chromosome', 'score_genome', 'pool'])
                           and len(commands) == 1)):
        commands.append('-h')
    args = parser.parse_args(commands)
    return args, commands


def run(argv=None):
    args, commands = getargs(argv)
    if commands[0] not in ['-h', '--help']:
        args.func(args)


if __name__ == '__main__':
    run()
