"""`peakachu score_genome` for the MI355X path (peakachu/score_genome.py:3-84).

Same flags, same chromosome selection, same output.  Started as the reference documents it
(README.md:127) on a box with several GPUs, the command runs one copy of itself per GPU
(cli.run -> peakachu_amd.launch; a launcher that sets RANK / WORLD_SIZE works too): chromosomes are dealt
to the ranks and rank 0 writes the bedpe in the reference's order after one RCCL gather.
"""
import contextlib
import os

import numpy as np

from . import dist, io, scoreUtils, utils
from .forest import load_model
from .stagetime import stage


def select_chromosomes(chromnames, chroms):
    """peakachu/score_genome.py:39-44."""
    queue = []
    for key in chromnames:
        chromlabel = key.lstrip('chr')
        if (not chroms) or (chromlabel.isdigit() and '#' in chroms) or (chromlabel in chroms):
            queue.append(key)
    return queue


def fetch_inputs(Lib, key, correct):
    """The reads of peakachu/score_genome.py:55-57 (balanced) / :63 (raw) for one chromosome.
    A reader that can hand out the chromosome as the file stores it (cool.CoolFile.upper: the
    upper triangle, no mirroring, no balancing) does so, and the device mirrors and balances
    (scoreUtils.Chromosome.from_upper); PK_UPPER=0 keeps the host matrices of the reference."""
    if hasattr(Lib, "upper") and os.environ.get("PK_UPPER", "1") != "0":
        try:
            pixels = Lib.upper(key)
        except OverflowError:
            pixels = None   # (more pixels than the device path's 32-bit row pointers: the host matrices below)
        if pixels is not None:
            bias, column = Lib.bias(correct, key) if correct else (None, None)
            return UpperInputs(pixels, bias, column)
    if correct:
        # the built-in reader keeps the chromosome's pixels for the second fetch (cooler has
        # no such notion: nullcontext)
        hold = getattr(Lib, "hold", None)
        with (hold(key) if hold else contextlib.nullcontext()):
            return (Lib.matrix(balance=correct, sparse=True).fetch(key),
                    Lib.matrix(balance=False, sparse=True).fetch(key),
                    Lib.bins().fetch(key)[correct].values)
    return Lib.matrix(balance=False, sparse=True).fetch(key), None, None


class UpperInputs:
    def __init__(self, pixels, bias, weights):
        self.pixels, self.bias, self.weights = pixels, bias, weights


def prefetched(Lib, keys, correct):
    """(key, inputs) for every key, in order; the NEXT chromosomes (three by default,
    PK_PREFETCH=n: measured on a 3.0e8-pixel map, 1 / 2 / 3 / 4 readers: 2.44 / 1.81 / 1.68 / 1.67 s)
    are read on background threads while the caller prepares and scores the
    current one.  Reading and inflating a chromosome costs more host time than scoring it on
    the GPU; zlib and numpy release the GIL, and the built-in readers use positional reads
    (h5lite.at) and a locked pixel cache (cool.CoolFile._mirrored), so two chromosomes can be
    in the making at once.  Each chromosome's own reads stay in the reference's order."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    keys = list(keys)
    if not keys:
        return
    depth = max(1, int(os.environ.get("PK_PREFETCH", "3")))
    with ThreadPoolExecutor(max_workers=depth) as pool:
        ahead = deque(pool.submit(fetch_inputs, Lib, k, correct) for k in keys[:depth])
        for i, key in enumerate(keys):
            with stage("wait for the reader threads"):
                cur = ahead.popleft().result()
            if i + depth < len(keys):
                ahead.append(pool.submit(fetch_inputs, Lib, keys[i + depth], correct))
            yield key, cur


def build_chromosome(Lib, key, cname, model, correct, args, width, device, inputs=None):
    """peakachu/score_genome.py:53-67 (the .cool branch)."""
    if inputs is None:
        inputs = fetch_inputs(Lib, key, correct)
    if isinstance(inputs, UpperInputs):
        return scoreUtils.Chromosome.from_upper(inputs.pixels, model=model, bias=inputs.bias, weights=inputs.weights,
                                                cname=cname, lower=args.lower, upper=args.upper,
                                                res=args.resolution, width=width, device=device)
    M, raw_M, weights = inputs
    if correct:
        with stage("tocsr"):
            M, raw_M = utils.tocsr(M), utils.tocsr(raw_M)
        return scoreUtils.Chromosome(M, model=model, raw_M=raw_M, weights=weights, cname=cname,
                                     lower=args.lower, upper=args.upper, res=args.resolution,
                                     width=width, device=device)
    with stage("tocsr"):
        M = utils.tocsr(M)
    return scoreUtils.Chromosome(M, model=model, raw_M=M, weights=None, cname=cname,
                                 lower=args.lower, upper=args.upper, res=args.resolution,
                                 width=width, device=device)


def warm_imports(device=0):
    """scipy.stats (the Poisson tables), scikit-learn's isotonic module (the expected curve's
    fit, peakachu/utils.py:173) and the HIP runtime's first contact with the device take a few
    tenths of a second; started on a thread here, that happens while the main thread opens the
    model and the first chromosome is read."""
    import threading

    def work():
        try:  # (the device context of this rank: ctypes releases the GIL while HIP initialises)
            from . import _lib
            _lib.load().pk_device_synchronize(int(device))
        except Exception:
            pass  # whatever is wrong with the library or the device, the first real call says it
        import scipy.stats  # noqa: F401  (the Poisson tables of get_candidate)
        try:
            import sklearn.isotonic  # noqa: F401
        except ImportError:
            pass
    t = threading.Thread(target=work, name="pk-warm-imports", daemon=True)
    t.start()
    return t


def join_warm(thread, timeout=60.0):
    """Before a driver returns -- normally or with an error (bad model path, unknown chromosome,
    missing weight column: all of them can fail within milliseconds) -- the warm-up thread must be out
    of the HIP runtime's initialisation: an interpreter that tears down libamdhip64's static objects
    while a thread is still inside hipInit can hang or abort instead of showing the error."""
    if thread is not None and thread.is_alive():
        thread.join(timeout)


def main(args):
    warm = warm_imports(dist.rank_info()[1])
    try:
        return _main(args)
    finally:
        join_warm(warm)


def _main(args):
    np.seterr(divide='ignore', invalid='ignore')
    rank, local_rank, world = dist.rank_info()
    if rank == 0 and os.path.exists(args.output):
        os.remove(args.output)

    with stage("load the model"):
        model = load_model(args.model)
    correct = False if args.clr_weight_name.lower() == 'raw' else args.clr_weight_name
    width = int((np.sqrt(model.feature_importances_.size) - 1) / 2)
    with stage("open the contact map"):
        Lib = io.open_map(args.path)
    queue = select_chromosomes(Lib.chromnames[:], args.chroms)

    # PK_FORCE_DIST=1 sends a single rank down the multi-rank branch (tests)
    if world == 1 and os.environ.get("PK_FORCE_DIST") != "1":
        for key, inputs in prefetched(Lib, queue, correct):
            cname = key if key.startswith('chr') else 'chr' + key
            X = build_chromosome(Lib, key, cname, model, correct, args, width, local_rank, inputs)
            result, R = X.score(thre=args.minimum_prob)
            with stage("writeBed"):
                X.writeBed(args.output, result, R)
        return

    # N ranks: chromosomes are independent units; weigh them by bin count^1
    # (candidates grow linearly with the chromosome length at a fixed band).  The sizes
    # come from the container's metadata: no rank reads a matrix it does not score.
    sizes = [io.chrom_bins(Lib, k) for k in queue]
    mine = dist.lpt_assign(sizes, world)[rank]
    # the transport exists BEFORE any scoring, so that a rank that fails can tell the
    # others: all ranks leave together, with an error, instead of waiting in the gather
    transport = make_transport(local_rank)
    try:
        recs, failure = [], None
        try:
            for qi, (key, inputs) in zip(mine, prefetched(Lib, [queue[q] for q in mine], correct)):
                cname = key if key.startswith('chr') else 'chr' + key
                X = build_chromosome(Lib, key, cname, model, correct, args, width, local_rank, inputs)
                result, R = X.score(thre=args.minimum_prob)
                r, c = result.nonzero()
                p = np.asarray(result[r, c]).ravel() if r.size else np.zeros(0)
                s = np.asarray(R[r, c]).ravel() if r.size else np.zeros(0)
                recs.append(dist.pack_records(qi, r, c, p, s))
        except Exception as e:  # reported below, on every rank
            failure = "rank %d: %s: %s" % (rank, type(e).__name__, e)
        failures = transport.all_failures(failure)
        if failures:
            raise RuntimeError("score_genome failed on %d of %d ranks: %s"
                               % (len(failures), world, "; ".join(failures)))
        local = np.concatenate(recs) if recs else np.empty(0, dist.RECORD)
        try:
            allrec = dist.gather_records(local, transport)
        except Exception as e:
            # e.g. PK_E_COMM: a peer went away and the bounded wait inside the gather ran out
            # (PK_COMM_TIMEOUT).  Say so and leave with a code of its own; nothing is retried here
            # (this process has touched the GPU: a retry is a fresh process)
            import sys
            sys.stderr.write("peakachu-amd score_genome: rank %d of %d: the gather of the scored pixels failed: %s\n"
                             % (rank, world, e))
            raise SystemExit(5)
        if rank == 0:
            write_gathered(args.output, allrec, queue, args.resolution, Lib)
        transport.barrier()
    finally:
        transport.close()


def make_transport(local_rank):
    """RCCL on GPUs; PK_TRANSPORT=tcp sends the records over the host rendezvous (tests)."""
    if os.environ.get("PK_TRANSPORT") == "tcp":
        return dist.TcpTransport()
    return dist.RcclTransport(local_rank)


def write_gathered(output, allrec, queue, res, Lib=None):
    """Rank 0: the reference's file order = chromosomes in queue order, pixels
    (row, col)-sorted within a chromosome (peakachu/scoreUtils.py:127-135)."""
    from scipy import sparse
    for qi, key in enumerate(queue):
        cname = key if key.startswith('chr') else 'chr' + key
        sel = allrec[allrec["chrom"] == qi]
        if sel.size == 0:
            continue
        n = int(max(sel["x"].max(), sel["y"].max())) + 1
        prob = sparse.csr_matrix((sel["prob"], (sel["x"], sel["y"])), shape=(n, n))
        sig = sparse.csr_matrix((sel["signal"], (sel["x"], sel["y"])), shape=(n, n))
        scoreUtils.write_bedpe(output, cname, res, prob, sig)
