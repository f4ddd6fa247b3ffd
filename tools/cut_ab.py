#!/usr/bin/env python
"""GPU-box helper (round 5): the forest cut in two (pk_forest_q.hip, q_pick_cut) against the one-launch
kernel on ONE box.  For every workload / threshold: warm microseconds per pk_score_run, the forest's
and the tail's HIP-event times, the cut, the candidates parked, and the check that the scored pixels
are the uncut run's bit for bit and that only decided candidates read probability 0.

usage: tools/cut_ab.py [w=5] [--cuts 5,6,7,8,9,10] [--thre 0.5,0.6,0.7,0.9] [--n 0]"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from peakachu_amd import _lib  # noqa: E402


def digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def timed(L, cd, hm, hf, w, thre, reps):
    cd.run(hm, hf, w, thre)
    cd.run(hm, hf, w, thre)
    L.pk_prof_enable(1)
    L.pk_prof_reset()
    _lib.check(L.pk_device_synchronize(0), "sync")
    t0 = time.perf_counter()
    for _ in range(reps):
        n = cd.run(hm, hf, w, thre)
    _lib.check(L.pk_device_synchronize(0), "sync")
    el = (time.perf_counter() - t0) / reps * 1e6
    L.pk_prof_enable(0)
    k = {c: _lib.prof_get(c)[0] / reps * 1e3 for c in ("extract", "quant", "forest", "forest_tail", "compact")}
    return el, k, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("w", nargs="?", type=int, default=5)
    ap.add_argument("--cuts", default="")
    ap.add_argument("--thre", default="0.5,0.6,0.7,0.9")
    ap.add_argument("--n", type=int, default=0, help="first n candidates only (0: all)")
    ap.add_argument("--bins", type=int, default=30000)
    ap.add_argument("--band", type=int, default=200)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--forest", default=None)
    ap.add_argument("--min", type=int, default=0, help="forest_split_min (0: the library's default)")
    a = ap.parse_args()
    w = a.w
    L = _lib.require_device()
    Mf, e, x, y, upper = bench.build_workload(0, a.bins, a.band, w, 6, a.band)
    if a.n:
        x, y = x[:a.n], y[:a.n]
    fo = bench.load_forest(a.forest, w, (2 * w + 1) ** 2)
    hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, Mf.shape[0], e, -2 * w + 1, upper + 2 * w - 1)
    hf = _lib.HipForest(fo, options={"forest_split_min": a.min} if a.min else None)
    print("w=%d  %d candidates, %d trees" % (w, x.size, fo.tree_off.size - 1), flush=True)
    for thre in [float(t) for t in a.thre.split(",")]:
        # the reference: no permission, every candidate's probability
        cd0 = _lib.HipCands(x, y)
        us0, k0, n0 = timed(L, cd0, hm, hf, w, thre, a.reps)
        base = digest(*cd0.fetch())
        st0, pr0 = cd0.fetch_all()
        cd0.close()
        print("thre %.2f  one launch                : %8.1f us/call  forest %8.1f us  pixels %d" %
              (thre, us0, k0["forest"], n0), flush=True)
        legs = [("cut (library's choice)", {})]
        legs += [("cut in front of group %d" % c, {"forest_split_at": c}) for c in
                 ([int(c) for c in a.cuts.split(",")] if a.cuts else [])]
        legs.append(("in-kernel exit (no cut)", {"forest_split": 0, "early_exit": 1}))
        for name, fopt in legs:
            for k_, v in fopt.items():
                if k_.startswith("forest_"):
                    hf.set_option(k_, v)
            cd = _lib.HipCands(x, y, options={k_: v for k_, v in fopt.items() if not k_.startswith("forest_")})
            cd.set_prune(True)
            us, k, n = timed(L, cd, hm, hf, w, thre, a.reps)
            ok = n == n0 and digest(*cd.fetch()) == base
            st, pr = cd.fetch_all()
            same = pr.view(np.uint64) == pr0.view(np.uint64)
            ok = ok and np.array_equal(st, st0) and bool(np.all(same | (pr == 0.0))) and bool(np.all(pr0[~same] <= thre))
            g = hf.get_option("stat_split_group")
            print("thre %.2f  %-26s: %8.1f us/call  forest %8.1f us (tail %6.1f)  cut %2d (%3d trees)  parked %8d  "
                  "zeroed %.3f  %s" % (thre, name, us, k["forest"], k["forest_tail"], g,
                                      hf.get_option("stat_split_trees"), hf.get_option("stat_split_parked"),
                                      float((~same).mean()), "same pixels" if ok else "MISMATCH"), flush=True)
            cd.close()
            hf.set_option("forest_split", 1)
            hf.set_option("forest_split_at", 0)
            if not ok and not os.environ.get("PK_CUT_AB_NOCHECK"):  # (timing ablations change the results)
                sys.exit(1)


if __name__ == "__main__":
    main()
