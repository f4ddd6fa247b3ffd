#!/usr/bin/env python
"""GPU-box helper: `Chromosome` (device-side preparation: band filter, validity flags, diagonal
means, Poisson candidates, then scoring) on seeded random inputs against the host restatement
of the reference's constructor (peakachu/scoreUtils.py:10-68: utils.calculate_expected on the
host, utils.band_filter, utils.candidates with scipy per pixel) and the oracle's score.
Raw / balanced (NaN weights) / hic-style (raw_M separate, non-integer M) modes, thin and dense
maps, empty far diagonals, NaN / inf cells, tiny chromosomes, upper beyond the matrix.
usage: tests/fuzz/fuzz_chromosome.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scipy import sparse
from oracle import oracle_np as onp
from peakachu_amd import _lib, scoreUtils, synth, utils
from peakachu_amd.forest import FlatForest
from test_gpu_parity import random_forest_arrays, flat


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    _lib.require_device()
    t0 = time.time()
    npix = 0
    n_upper = 0
    for case in range(n_cases):
        seed = seed0 + case
        rng = np.random.default_rng(seed)
        w = int(rng.choice([5, 5, 6, 3, 11]))
        n = int(rng.integers(10 * w + 40, 1500))
        band = int(rng.integers(4 * w + 12, min(260, n // 2)))
        upper = int(rng.choice([band - 2 * w - 1, band, band + 40, n, n + 50]))
        lower = int(rng.choice([1, 6, w + 3, 20]))
        raw, _ = synth.synth_band(n, band, seed=seed, loops=max(2, n // 40))
        raw = sparse.csr_matrix(raw, dtype=np.float64)
        thin = rng.random() < 0.3
        if thin:  # low coverage: far diagonals nearly (or entirely) empty
            keep = rng.random(raw.data.size) < 0.15
            raw.data[~keep] = 0
            raw.eliminate_zeros()
            coo = raw.tocoo()
            far = (coo.col - coo.row) > band * 0.7
            coo.data[far & (rng.random(coo.data.size) < 0.98)] = 0
            raw = sparse.csr_matrix(coo); raw.eliminate_zeros()
        mode = str(rng.choice(["raw", "weights", "hic"]))
        weights = None
        if mode == "raw":
            M = raw
        elif mode == "weights":
            weights = synth.synth_weights(n, seed, n_nan=int(rng.integers(0, 9)))
            M = synth.balance(raw, weights)
        else:
            M = sparse.csr_matrix(raw * 0.37)
        injected = False
        if rng.random() < 0.15 and mode != "raw":  # a few non-finite cells
            injected = True
            M = M.copy()
            idx = rng.choice(M.data.size, min(5, M.data.size), replace=False)
            M.data[idx[:3]] = np.nan
            M.data[idx[3:]] = np.inf
        F = (2 * w + 1) ** 2
        fo = random_forest_arrays(F, int(rng.integers(3, 30)), seed, depth=int(rng.integers(3, 10)))
        model = flat(fo)
        thre = float(rng.choice([0.0, 0.3, 0.5]))
        raw_arg = raw if mode != "raw" else M
        try:
            ch = scoreUtils.Chromosome(M, model, raw_M=raw_arg, weights=weights, lower=lower, upper=upper, width=w)
        except Exception as e:
            # e.g. an infinite cell: the isotonic fit of the expected curve refuses it in the
            # reference as well -- the host restatement must fail the same way
            try:
                utils.calculate_expected(M, min(upper, n - 2 * w) + 2 * w, raw=weights is None)
            except Exception as e2:
                assert type(e2) is type(e), (e, e2)
                print("case %3d seed=%d %s: both raise %s (%s)" % (case, seed, mode, type(e).__name__, str(e)[:60]))
                continue
            print("case %3d seed=%d: only the device path raised %s: %s" % (case, seed, type(e).__name__, e))
            raise
        # the same chromosome from its pixel table as a .cool stores it (Chromosome.from_upper: mirrored and
        # balanced on the device) must be the same object: expected curve, background, candidates
        if mode in ("raw", "weights") and not injected:
            tri = sparse.triu(raw, 0, format="csr")
            tri.sort_indices()
            px = utils.UpperPixels(n, tri.indptr.astype(np.int32), tri.indices.astype(np.int32),
                                   tri.data.astype(np.int32) if rng.random() < 0.7 else tri.data.astype(np.float64))
            chu = scoreUtils.Chromosome.from_upper(px, model, bias=weights, weights=weights, lower=lower, upper=upper, width=w)
            # against the matrix constructor on the MIRROR IMAGE of that table (a thinned map is not
            # symmetric any more; balanced values as cooler makes them, (w[row] * w[col]) * count:
            # synth.balance multiplies in another order)
            sym = px.symmetric()
            chm = scoreUtils.Chromosome(sym if mode == "raw" else px.symmetric(weights), model, raw_M=sym,
                                        weights=weights, lower=lower, upper=upper, width=w)
            same = (np.array_equal(bits(chu.exp_arr), bits(chm.exp_arr)) and np.array_equal(bits(chu.background), bits(chm.background))
                    and np.array_equal(chu.ridx, chm.ridx) and np.array_equal(chu.cidx, chm.cidx))
            ru, rm = sparse.csr_matrix(chu.score(thre)[0]), sparse.csr_matrix(chm.score(thre)[0])
            n_upper += 1
        else:
            chu, same, ru, rm = None, True, None, None
        # host restatement of the constructor
        lo = max(lower, w + 1)
        up = min(upper, n - 2 * w)
        if weights is None:
            e_ref = utils.calculate_expected(M, up + 2 * w, raw=True)
            bg_ref = e_ref if mode == "raw" else utils.calculate_expected(raw, up + 2 * w, raw=True)
        else:
            e_ref = utils.calculate_expected(M, up + 2 * w, raw=False)
            bg_ref = e_ref
        ok = np.array_equal(bits(ch.exp_arr), bits(e_ref)) and np.array_equal(bits(ch.background), bits(bg_ref))
        Mf = utils.band_filter(M, w, up)
        d = ch.M - Mf
        ok_band = (abs(d) > 0).nnz == 0 and ch.M.nnz == Mf.nnz
        rx, ry = utils.candidates(raw_arg, bg_ref, weights, lo, up)
        ok_c = np.array_equal(ch.ridx, rx) and np.array_equal(ch.cidx, ry)
        res, R = ch.score(thre)
        if ru is not None:
            ru.sort_indices(), rm.sort_indices()
            same = same and np.array_equal(ru.indptr, rm.indptr) and np.array_equal(ru.indices, rm.indices) and \
                np.array_equal(bits(ru.data), bits(rm.data))
        ox, oy = res.nonzero()
        px, py, pp, ps = onp.score(Mf, e_ref, w, fo, thre, np.asarray(rx, np.int32), np.asarray(ry, np.int32), threads=8)
        got = sparse.csr_matrix(res)
        ref = sparse.csr_matrix((pp, (px, py)), shape=(n, n)) if px.size else sparse.csr_matrix((n, n))
        ok_s = (got != ref).nnz == 0 and np.array_equal(bits(np.asarray(got[px, py]).ravel()), bits(pp)) if px.size else got.nnz == 0
        npix += int(px.size)
        print("case %3d seed=%d w=%2d n=%4d band=%3d lower=%2d upper=%4d %-7s thin=%d cands=%6d scored=%5d device_cands=%s: exp %s band %s cands %s score %s" % (
            case, seed, w, n, band, lower, upper, mode, thin, len(rx), px.size, ch._cands is not None,
            "ok" if ok else "MISMATCH", "ok" if ok_band else "MISMATCH", "ok" if ok_c else "MISMATCH", "ok" if ok_s else "MISMATCH")
            + ("" if chu is None else " from_upper %s" % ("ok" if same else "MISMATCH")))
        sys.stdout.flush()
        if not (ok and ok_band and ok_c and ok_s and same):
            sys.exit(1)
    print("all %d cases identical in %.0f s (%d scored pixels compared; %d chromosomes also built from their pixel table)"
          % (n_cases, time.time() - t0, npix, n_upper))


if __name__ == "__main__":
    main()
