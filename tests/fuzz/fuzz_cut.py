#!/usr/bin/env python
"""GPU-box helper (round 5): the forest CUT IN TWO (pk_forest_q.hip: head over every candidate, the open
ones parked with their partial sums and rank codes, tail over those) on seeded random inputs against the
CPU oracle, bit for bit -- lists long enough for the parked candidates to fit the chunk's float tiles,
forests of 20 .. 400 trees (many tree groups), window sizes that reach every rank kernel (w = 5, 6:
forest_qr_kernel; 7: the 128-candidate generic kernel; 11: the two-tile wide kernel; 13: the 64-candidate
generic kernel), any cut (forced in front of a random group, or the library's own choice with its
learning over repeated calls), thresholds 0 .. 0.9, chunked lists, NaN features (flat matrices), batch
sizes that skip batches.  Every case also checks that a candidate's probability is either the uncut
run's or 0, and that only losers were decided.
usage: tests/fuzz/fuzz_cut.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scipy import sparse  # noqa: E402
from oracle import oracle_np as onp  # noqa: E402
from peakachu_amd import _lib, synth, utils  # noqa: E402
from test_gpu_parity import random_forest_arrays, flat, hip_matrix  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
    _lib.require_device()
    t0 = time.time()
    cuts = {}
    for case in range(n_cases):
        seed = seed0 + case
        rng = np.random.default_rng(seed)
        w = int(rng.choice([5, 5, 6, 6, 7, 11, 11, 13]))
        n = int(rng.integers(900, 2200))
        band = int(rng.integers(60, 140))
        upper = int(rng.integers(band // 2, band))
        M, _ = synth.synth_band(n, band, seed=seed, loops=n // 40)
        kind = int(rng.integers(0, 4))
        raw = True
        if kind == 1:   # balanced values, NaN weights
            M = synth.balance(M, synth.synth_weights(n, seed, n_nan=3))
            raw = False
        M = sparse.csr_matrix(M, dtype=np.float64)
        if kind == 2:   # a flat stretch: constant windows -> NaN features (status 2: the walk with missing_go_to_left)
            M.data[: M.data.size // 3] = 2.0
            raw = False
        e = utils.calculate_expected(M, upper + 2 * w, raw=raw)
        if kind == 2:
            e = e[:4].copy()   # (windows stay unnormalised: the flat stretch is truly flat)
        Mf = utils.band_filter(M, w, upper)
        x, y = synth.all_band_pixels(Mf, w + 1, upper)
        F = (2 * w + 1) ** 2
        T = int(rng.choice([20, 40, 100, 100, 200, 400]))
        fo = random_forest_arrays(F, T, seed, depth=int(rng.integers(4, 11)))
        # leaf values like a fitted forest's (mostly small, some 1.0): candidates get decided at different groups
        lv = rng.random(fo["p1"].size)
        fo["p1"] = np.where(lv < 0.7, rng.integers(0, 3, lv.size) / 16.0, np.where(lv < 0.8, 1.0, fo["p1"])).astype(np.float64)
        if rng.random() < 0.5:
            fo["miss_left"] = rng.integers(0, 2, fo["miss_left"].size).astype(np.uint8)
        batch = int(rng.choice([64, 97, 4096, 100000]))
        opts = {"forest_split_min": 1}
        if rng.random() < 0.4:
            opts["chunk"] = int(rng.choice([65536, 100000, 150000]))
        hm = hip_matrix(Mf, e, w, upper, options=opts)
        hf = _lib.HipForest(flat(fo), options=opts)
        # the one-launch run of the device (every candidate's probability); the threshold is taken from its
        # distribution, so that a cut has something to decide and something to leave open
        cd0 = _lib.HipCands(x, y, options=opts)
        cd0.run(hm, hf, w, 0.5, batch)
        st0, pr0 = cd0.fetch_all()
        cd0.close()
        live = pr0[st0 != 0]
        q = float(rng.choice([0.0, 0.5, 0.9, 0.97, 0.999]))
        thre = float(np.quantile(live, q)) if (q > 0 and live.size) else 0.0
        rx, ry, rp, rs = onp.score(Mf, e, w, fo, thre, x, y, batch=batch, threads=0)
        n_grp = hf.get_option("stat_q_groups")
        at = 0 if (n_grp <= 1 or rng.random() < 0.4) else int(rng.integers(1, n_grp))
        hf.set_option("forest_split_at", at)
        ok = True
        seen = []
        for rep in range(1 if at else 3):   # (the library's own cut learns from call to call)
            cd = _lib.HipCands(x, y, options=opts)
            cd.set_prune(True)
            cd.run(hm, hf, w, thre, batch)
            ox, oy, op, osig = cd.fetch()
            st, pr = cd.fetch_all()
            cd.close()
            same = pr.view(np.uint64) == pr0.view(np.uint64)
            ok = ok and (np.array_equal(ox, rx) and np.array_equal(oy, ry)
                         and np.array_equal(op.view(np.uint64), rp.view(np.uint64))
                         and np.array_equal(np.asarray(osig).view(np.uint64), np.asarray(rs).view(np.uint64))
                         and np.array_equal(st, st0) and bool(np.all(same | (pr == 0.0)))
                         and bool(np.all(pr0[~same] <= thre)))
            seen.append(int(hf.get_option("stat_split_group")))
        fam = int(hf.get_option("stat_family"))
        if any(seen):
            cuts[fam] = cuts.get(fam, 0) + 1
        print("case %3d seed=%d w=%2d n=%4d cands=%7d kind=%d T=%3d groups=%2d thre=%.4f batch=%6d %s: family %d, cut %s of %d, "
              "%d pixels, %d NaN-feature candidates %s" % (case, seed, w, n, x.size, kind, T, n_grp, thre, batch,
                                                          {k: v for k, v in opts.items() if k != "forest_split_min"}, fam,
                                                          seen if not at else "forced %d -> %s" % (at, seen), n_grp, rx.size,
                                                          int((st0 == 2).sum()), "ok" if ok else "MISMATCH"))
        sys.stdout.flush()
        hf.close()
        hm.close()
        if not ok:
            sys.exit(1)
    print("all %d cases bit-exact in %.0f s; cases with a cut launch by kernel family (1 qr, 2 generic, 3 two-tile): %s"
          % (n_cases, time.time() - t0, dict(sorted(cuts.items()))))


if __name__ == "__main__":
    main()
