#!/usr/bin/env python
"""GPU-box helper: pk_extract / getwindow on ARBITRARY coordinates (round 3) against the CPU
oracle, bit for bit: pixels above, on and below the diagonal, within and beyond 2w of it,
off-matrix entries, windows whose columns start left of the matrix, and calls the reference
answers with an IndexError (both sides must refuse those).  Random w (1 .. 15), raw / balanced
/ dirty matrices, expected curves cut short.
usage: tests/fuzz/fuzz_getwindow.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scipy import sparse
from oracle import oracle_np as onp
from peakachu_amd import _lib, synth, utils
from test_gpu_parity import hip_matrix


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 70000
    _lib.require_device()
    t0 = time.time()
    n_win = n_low = n_wrap = n_raise = 0
    for case in range(n_cases):
        seed = seed0 + case
        rng = np.random.default_rng(seed)
        w = int(rng.choice([1, 2, 3, 5, 5, 6, 6, 7, 9, 11, 11, 13, 15]))
        n = int(rng.integers(6 * w + 20, 500))
        band = int(rng.integers(4 * w + 6, max(4 * w + 7, min(160, n // 2))))
        upper = int(rng.integers(2 * w + 3, band))
        M, _ = synth.synth_band(n, band, seed=seed, loops=max(2, n // 30))
        kind = int(rng.integers(0, 4))
        raw = True
        if kind == 1:
            M = synth.balance(M, synth.synth_weights(n, seed, n_nan=2))
            raw = False
        M = sparse.csr_matrix(M, dtype=np.float64)
        if kind == 2:
            idx = rng.choice(M.data.size, 6, replace=False)
            M.data[idx[:3]] = np.nan
            M.data[idx[3:]] = -M.data[idx[3:]]
            raw = False
        e = utils.calculate_expected(M, upper + 2 * w, raw=raw)
        if kind == 3:
            e = e[: max(3, e.size // 2)].copy()
        Mf = utils.band_filter(M, w, upper)
        k = int(rng.integers(1, 400))
        x = rng.integers(-3, n + 3, k)
        style = rng.integers(0, 4, k)
        y = np.where(style == 0, x + rng.integers(0, upper + 2 * w + 3, k),       # above the diagonal
             np.where(style == 1, x - rng.integers(0, 3 * w + 2, k),              # below, near it
             np.where(style == 2, rng.integers(-3, n + 3, k),                      # anywhere
                      rng.integers(0, 2 * w + 1, k))))                             # near column 0 (wraps when x > y)
        passes = (x - w >= 0) & (y + w + 1 <= n)
        raises = passes & ((x + w >= n) | (y - w < -n))
        hm = hip_matrix(Mf, e, w, upper)
        if raises.any():
            n_raise += 1
            for fn in (lambda: hm.extract(w, x, y), lambda: onp.extract(Mf, e, w, x, y)):
                try:
                    fn()
                    print("case %d seed=%d: a call the reference refuses was answered" % (case, seed))
                    sys.exit(1)
                except (IndexError, _lib.PeakachuHipError):
                    pass
            x, y = x[~raises], y[~raises]
        f64, f32, keep = hm.extract(w, x, y, want64=True, want32=True)
        ref, rkeep = onp.extract(Mf, e, w, x, y)
        ok = (np.array_equal(keep, rkeep) and np.array_equal(f64.view(np.uint64), ref.view(np.uint64))
              and np.array_equal(f32.view(np.uint32), ref.astype(np.float32).view(np.uint32)))
        n_win += keep.size
        n_low += int(np.sum(x[keep] > y[keep]))
        n_wrap += int(np.sum((x[keep] > y[keep]) & (y[keep] < w)))
        if case % 50 == 0 or not ok:
            print("case %4d seed=%d w=%2d n=%3d upper=%3d kind=%d coords=%3d kept=%3d %s" % (
                case, seed, w, n, upper, kind, x.size, keep.size, "ok" if ok else "MISMATCH"))
            sys.stdout.flush()
        if not ok:
            sys.exit(1)
    print("all %d cases bit-exact in %.0f s: %d windows, %d of them below the diagonal, %d with wrapped columns; "
          "%d calls with coordinates the reference raises on (refused by both)" % (
              n_cases, time.time() - t0, n_win, n_low, n_wrap, n_raise))


if __name__ == "__main__":
    main()
