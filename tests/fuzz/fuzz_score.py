#!/usr/bin/env python
"""GPU-box helper: the whole scoring path (pk_score: extract + forest + threshold + batch
rule) on seeded random inputs against the CPU oracle, bit for bit -- the shapes of
tests/test_gpu_parity.py::test_randomised_score_vs_oracle, many more of them, plus matrices
the fast extractor must hand to the general one (NaN / negative / tiny / huge cells, a zero
or NaN in the expected curve) and random library options.
usage: tests/fuzz/fuzz_score.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scipy import sparse
from oracle import oracle_np as onp
from peakachu_amd import _lib, synth, utils
from test_gpu_parity import random_forest_arrays, flat, hip_matrix


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    L = _lib.require_device()
    t0 = time.time()
    stats = {}
    for case in range(n_cases):
        seed = seed0 + case
        rng = np.random.default_rng(seed)
        w = int(rng.choice([1, 2, 3, 4, 5, 5, 5, 6, 6, 6, 7, 8, 11, 11, 11, 13, 15]))
        n = int(rng.integers(8 * w + 60, 900))
        band = int(rng.integers(4 * w + 10, min(200, n // 2)))
        upper = int(rng.integers(2 * w + 4, band))
        M, _ = synth.synth_band(n, band, seed=seed, loops=max(2, n // 30))
        kind = int(rng.integers(0, 9))
        raw = True
        if kind == 1:   # balanced, non-integer values with NaN weights
            M = synth.balance(M, synth.synth_weights(n, seed, n_nan=3))
            raw = False
        M = sparse.csr_matrix(M, dtype=np.float64)
        if kind == 2:   # dirty cells: the fast extractor's preconditions fail
            idx = rng.choice(M.data.size, 12, replace=False)
            M.data[idx[:3]] = np.nan
            M.data[idx[3:6]] = -M.data[idx[3:6]]
            M.data[idx[6:9]] = 1e-120
            M.data[idx[9:]] = 1e160
            raw = False
        if kind == 6:   # flat windows: equal counts everywhere (min-max scaling -> 0 / 0 -> NaN features)
            M.data[:] = float(rng.choice([1.0, 3.0, 0.5]))
            raw = False
        if kind == 7:   # cells exactly on the limits the fast extractor's precondition tests
            idx = rng.choice(M.data.size, 8, replace=False)
            M.data[idx] = rng.choice([1e-100, 9.99e-101, 1e150, 9.9e149, 5e-324, 1.7e308], 8)
            raw = False
        if kind == 8:   # negative zeros and a few exact zeros stored explicitly
            idx = rng.choice(M.data.size, 10, replace=False)
            M.data[idx[:5]] = -0.0
            M.data[idx[5:]] = 0.0
            raw = False
        try:
            e = utils.calculate_expected(M, upper + 2 * w, raw=raw)
        except ValueError as err:  # a diagonal mean overflowed: the reference's isotonic fit refuses it too
            print("case %3d seed=%d kind=%d: expected curve refused (%s)" % (case, seed, kind, str(err)[:50]))
            continue
        if kind == 6:
            e = e[: max(3, int(rng.integers(3, e.size)))].copy()   # most windows stay unnormalised: truly flat
        if kind == 3:
            e = e.copy(); e[int(rng.integers(0, e.size))] = 0.0
        if kind == 4:
            e = e[: max(3, e.size // 2)].copy()   # too short: windows far from the diagonal stay unnormalised
        Mf = utils.band_filter(M, w, upper)
        x, y = synth.all_band_pixels(Mf, w + 1, upper)
        keep = rng.random(x.size) < float(rng.choice([0.6, 0.6, 0.93, 1.0]))   # thinned lists, and lists of runs
        x, y = x[keep], y[keep]
        x = np.r_[x, [0, 1, n - w - 2]].astype(np.int32)
        y = np.r_[y, [w + 2, w + 3, n - 1]].astype(np.int32)
        F = (2 * w + 1) ** 2
        fo = random_forest_arrays(F, int(rng.integers(3, 60)), seed, depth=int(rng.integers(3, 12)))
        if rng.random() < 0.5:
            fo["miss_left"] = rng.integers(0, 2, fo["miss_left"].size).astype(np.uint8)
        thre = float(rng.choice([0.0, 0.3, 0.5, 0.8]))
        batch = int(rng.choice([1, 2, 97, 4096, 100000]))
        opts = {}
        if rng.random() < 0.3:
            opts["chunk"] = int(rng.choice([1024, 5000, 65536]))
        if rng.random() < 0.2:
            opts["forest_q_persist"] = int(rng.choice([0, -1, -4]))
        if rng.random() < 0.1:
            opts["forest_q"] = 0
        if rng.random() < 0.1:
            opts["extract_pair"] = 0
        if rng.random() < 0.1:
            opts["extract_clean"] = 0
        if rng.random() < 0.15:
            opts["extract_row16"] = 0
        if rng.random() < 0.4:     # the LDS-staged extractor: never / on every list (default: lists of runs only)
            opts["extract_strip"] = int(rng.choice([0, 2, 2]))
        if rng.random() < 0.15:
            opts["early_exit"] = 1
        if rng.random() < 0.4:     # the forest cut in two (pk_score always allows it): any length, any cut
            opts["forest_split_min"] = 1
            opts["forest_split_at"] = int(rng.choice([0, 0, 1, 2, 3, 5]))
        hm = hip_matrix(Mf, e, w, upper, options=opts)   # (options are the handles' own)
        hf = _lib.HipForest(flat(fo), options=opts)
        ox, oy, op, osig = hm.score(hf, w, thre, x, y, batch=batch)
        rx, ry, rp, rs = onp.score(Mf, e, w, fo, thre, x, y, batch=batch, threads=8)
        ok = (np.array_equal(ox, rx) and np.array_equal(oy, ry)
              and np.array_equal(op.view(np.uint64), rp.view(np.uint64))
              and np.array_equal(np.asarray(osig).view(np.uint64), np.asarray(rs).view(np.uint64)))
        stats[kind] = stats.get(kind, 0) + 1
        cut = hf.get_option("stat_split_group")
        stats["cut"] = stats.get("cut", 0) + (1 if cut > 0 else 0)
        print("case %3d seed=%d w=%2d n=%3d band=%3d upper=%3d kind=%d cands=%6d thre=%.1f batch=%6d %s: %d pixels%s %s" % (
            case, seed, w, n, band, upper, kind, x.size, thre, batch, opts, ox.size,
            " (forest cut in front of group %d, family %d)" % (cut, hf.get_option("stat_family")) if cut > 0 else "",
            "ok" if ok else "MISMATCH"))
        sys.stdout.flush()
        if not ok:
            sys.exit(1)
    cuts = stats.pop("cut", 0)
    print("all %d cases bit-exact in %.0f s; matrix kinds: %s; %d with the forest cut in two" % (
        n_cases, time.time() - t0, dict(sorted(stats.items())), cuts))


if __name__ == "__main__":
    main()
