#!/usr/bin/env python
"""Build-container helper (needs /root/reference; no GPU): `peakachu pool` -- the REFERENCE's own
call_loops.main against peakachu_amd.call_loops.main on many synthetic scored-pixel files
(blobs, stripes, isolated pixels, exact ties; tools/make_golden.py's generator), thresholds and
resolutions.  The two output files must be byte-identical.
usage: tests/fuzz/fuzz_pool.py [n_seeds] [first_seed]"""
import argparse, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden  # sets up the reference import (identity numba.njit) and sys.path
from peakachu import call_loops as ref_pool
from peakachu_amd import call_loops as my_pool


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    tmp = tempfile.mkdtemp()
    t_ref = t_my = 0.0
    lines = 0
    for seed in range(first, first + n):
        res = (10000, 5000, 25000)[seed % 3]
        text = make_golden.pool_input(seed, n_chrom=1 + seed % 3, res=res)
        fin = os.path.join(tmp, "in.bedpe")
        open(fin, "w").write(text)
        for thre in (0.9, 0.5, 0.97, 0.0, 0.999)[: 3 + seed % 3]:
            fa, fb = os.path.join(tmp, "a.bedpe"), os.path.join(tmp, "b.bedpe")
            for f in (fa, fb):
                if os.path.exists(f):
                    os.remove(f)
            t0 = time.time()
            ref_pool.main(argparse.Namespace(resolution=res, infile=fin, outfile=fa, threshold=thre))
            t1 = time.time()
            my_pool.main(argparse.Namespace(resolution=res, infile=fin, outfile=fb, threshold=thre))
            t2 = time.time()
            t_ref += t1 - t0
            t_my += t2 - t1
            a = open(fa).read() if os.path.exists(fa) else None
            b = open(fb).read() if os.path.exists(fb) else None
            ok = a == b
            lines += a.count("\n") if a else 0
            print("seed %4d res %5d thre %.3f: %5d pixels -> %4s loops %s" % (
                seed, res, thre, text.count("\n"), a.count("\n") if a is not None else "none", "ok" if ok else "MISMATCH"))
            if not ok:
                sys.exit(1)
    print("all identical; %d loop lines compared; reference %.1f s, this build %.1f s" % (lines, t_ref, t_my))


if __name__ == "__main__":
    main()
