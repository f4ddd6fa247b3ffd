"""Worker of tests/test_dist.py: one rank of a world_size-2 (or 3) job on CPUs, launched
the way the driver launches bench.py or by peakachu_amd.launch.  Exercises the N>1 host
path (chromosome dealing, candidate-block sharding, packing, gather to rank 0, ordered
merge, failure vote) over the product's own rendezvous (peakachu_amd.rendezvous, TCP) with
the CPU oracle as the per-rank scorer -- the GPU ranks run the same code with the HIP
library and the RCCL transport instead."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_io as gio  # noqa: E402
from oracle import oracle_np as onp  # noqa: E402
from peakachu_amd import dist, io, score_genome, utils  # noqa: E402


def oracle_chromosome(lib, key, fo, w, lower, upper, thre):
    M = utils.tocsr(lib.matrix(balance=False, sparse=True).fetch(key))
    lower = max(lower, w + 1)
    upper = min(upper, M.shape[0] - 2 * w)
    e = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = utils.candidates(M, e, None, lower, upper)
    return onp.score(Mf, e, w, fo, thre, x, y)


def main():
    out_dir = sys.argv[1]
    tr = dist.TcpTransport()
    rank, world = tr.rank, tr.world
    np.seterr(divide="ignore", invalid="ignore")
    z = gio.load("g6_driver.npz")
    fo = gio.forest(str(z["forest"]))
    lib = io.open_map(os.path.join(gio.GOLD, str(z["container"])))
    # ---- 1. score_genome: chromosomes dealt to ranks, one gather, ordered merge
    queue = score_genome.select_chromosomes(lib.chromnames[:], ["#", "X"])
    sizes = [io.chrom_bins(lib, k) for k in queue]   # container metadata, no matrix is read
    assert sizes == [lib.matrix(balance=False, sparse=True).fetch(k).shape[0] for k in queue]
    mine = dist.lpt_assign(sizes, world)[rank]
    recs = []
    for qi in mine:
        ox, oy, op, osig = oracle_chromosome(lib, queue[qi], fo, 5, int(z["lower"]),
                                             int(z["upper"]), 0.5)
        recs.append(dist.pack_records(qi, ox, oy, op, osig))
    local = np.concatenate(recs) if recs else np.empty(0, dist.RECORD)
    allrec = dist.gather_records(local, tr)
    ok = True
    if rank == 0:
        path = os.path.join(out_dir, "genome.bedpe")
        score_genome.write_gathered(path, allrec, queue, 10000)
        text = open(path).read() if os.path.exists(path) else ""
        ok = ok and (text == str(z["genome_raw"]))
    # ---- 1b. a failing rank is reported to every rank (score_genome leaves together)
    fails = tr.all_failures("rank %d: boom" % rank if rank == world - 1 else None)
    ok = ok and fails == ["rank %d: boom" % (world - 1)]
    ok = ok and tr.all_failures(None) == []
    # ---- 2. one chromosome, candidate blocks cut at batch boundaries
    q = gio.load("g4_batch_quirk.npz")
    w, upper = int(q["w"]), int(q["upper"])
    Mf = onp.band_filter(gio.sym_matrix(q, "R"), w, upper)
    fo4 = gio.forest(str(q["forest"]))
    lo, hi = dist.block_ranges(q["bx"].size, world, 100000)[rank]
    ox, oy, op, osig = onp.score(Mf, q["exp_arr"], w, fo4, 0.5, q["bx"][lo:hi], q["by"][lo:hi])
    allrec = dist.gather_records(dist.pack_records(0, ox, oy, op, osig), tr)
    oks = tr.all_failures(None if ok else "rank %d: mismatch" % rank)
    ok = ok and not oks
    if rank == 0:
        order = np.lexsort((allrec["y"], allrec["x"]))
        a = allrec[order]
        ok = ok and np.array_equal(a["x"], q["b_ri"]) and np.array_equal(a["y"], q["b_ci"])
        ok = ok and np.array_equal(gio.bits(a["prob"]), gio.bits(q["b_prob"]))
        ok = ok and np.array_equal(gio.bits(a["signal"]), gio.bits(q["b_signal"]))
        open(os.path.join(out_dir, "result.txt"), "w").write("OK" if ok else "MISMATCH")
    tr.barrier()
    tr.close()


if __name__ == "__main__":
    main()
