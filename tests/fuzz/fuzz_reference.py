#!/usr/bin/env python
"""Build-container helper (needs /root/reference; no GPU): the REFERENCE's own `Chromosome`
(constructor, get_candidate, score) run on many seeded random chromosomes, against the
oracle chain the GPU path is tested with -- utils.calculate_expected / band_filter /
candidates on the host and oracle.score (oracle/pk_oracle.c).  Widens the pin of the oracle
beyond the fixed fixtures of tests/golden/: raw / balanced (NaN weights) / separate-raw
modes, w = 3 .. 6, random forests fitted by the installed scikit-learn, random thresholds.

One known caveat (tools/make_golden.py): with numba absent the reference's window[:w,:w].mean()
is numpy's pairwise sum, production numba's is sequential (the oracle's order); a pixel whose
centre / mean ratio sits within an ulp of 0.1 could differ.  Such a case is reported, not
hidden; none has occurred.
usage: tests/fuzz/fuzz_reference.py [n_cases] [first_seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as mg  # sets up the reference import (identity numba.njit)
from scipy import sparse
from sklearn.ensemble import RandomForestClassifier
from oracle import oracle_np as onp
from peakachu_amd import synth, utils
from peakachu_amd.forest import FlatForest


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    t0 = time.time()
    npix = ncand = 0
    for seed in range(first, first + n_cases):
        rng = np.random.default_rng(seed)
        w = int(rng.choice([5, 5, 6, 3, 4, 7, 11]))
        n = int(rng.integers(12 * w + 60, 12 * w + 360))
        band = int(rng.integers(4 * w + 12, min(90, n // 2)))
        upper = int(rng.choice([band - 2 * w - 1, band, band + 30, n]))
        lower = int(rng.choice([1, 6, w + 3]))
        raw, _ = synth.synth_band(n, band, seed=seed, loops=max(2, n // 40))
        raw = sparse.csr_matrix(raw, dtype=np.float64)
        mode = str(rng.choice(["raw", "weights", "hic"]))
        weights = None
        if mode == "raw":
            M = raw
        elif mode == "weights":
            weights = synth.synth_weights(n, seed, n_nan=int(rng.integers(0, 6)))
            M = synth.balance(raw, weights)
        else:
            M = sparse.csr_matrix(raw * 0.37)
        F = (2 * w + 1) ** 2
        # a forest fitted on noise with a planted centre signal (any fitted forest will do)
        Xtr = rng.random((400, F))
        ytr = (Xtr[:, F // 2] > 0.6).astype(int)
        rf = RandomForestClassifier(n_estimators=int(rng.integers(3, 15)), max_depth=int(rng.integers(3, 9)),
                                    class_weight=rng.choice([None, "balanced"]), n_jobs=1, random_state=seed).fit(Xtr, ytr)
        thre = float(rng.choice([0.0, 0.3, 0.5]))
        # ---- the reference
        ch = mg.make_chrom(M, rf, w, lower=lower, upper=upper, weights=weights,
                           raw_M=(raw if mode != "raw" else None), cname="chr1")
        ref_M = ch.M.copy()
        ref = mg.run_score(ch, thre)
        # ---- the oracle chain
        lo, up = max(lower, w + 1), min(upper, n - 2 * w)
        if weights is None:
            e = utils.calculate_expected(M, up + 2 * w, raw=True)
            bg = e if mode == "raw" else utils.calculate_expected(raw, up + 2 * w, raw=True)
        else:
            e = utils.calculate_expected(M, up + 2 * w, raw=False)
            bg = e
        ok_e = np.array_equal(bits(e), bits(ch.exp_arr)) and np.array_equal(bits(bg), bits(ch.background))
        Mf = utils.band_filter(M, w, up)
        ok_b = (abs(Mf - ref_M) > 0).nnz == 0 and Mf.nnz == ref_M.nnz
        if not ok_b:
            print('   band: nnz', Mf.nnz, ref_M.nnz, 'differing', (abs(Mf - ref_M) > 0).nnz, 'NaN in ours', int(np.isnan(Mf.data).sum()), 'theirs', int(np.isnan(ref_M.data).sum()))
        cx, cy = utils.candidates(raw if mode != "raw" else M, bg, weights, lo, up)
        ok_c = np.array_equal(cx, ch.ridx) and np.array_equal(cy, ch.cidx)
        ff = FlatForest.from_sklearn(rf)
        fo = {k: getattr(ff, k) for k in FlatForest.FIELDS}
        ox, oy, op, osig = onp.score(Mf, e, w, fo, thre, cx.astype(np.int32), cy.astype(np.int32))
        # the reference returns a matrix: compare as (row, col)-sorted triples
        o = np.lexsort((oy, ox))
        ok_s = (np.array_equal(ox[o], ref["ri"]) and np.array_equal(oy[o], ref["ci"])
                and np.array_equal(bits(op[o]), bits(ref["prob"])) and np.array_equal(bits(osig[o]), bits(ref["signal"])))
        # ---- getwindow itself on arbitrary upper-triangle coordinates (not only candidates):
        # float64 features and the list of coordinates that survive the filters
        k = 300
        gx = rng.integers(0, n - 1, k)
        gy = np.minimum(gx + rng.integers(0, up + 3, k), n - 1)
        ch2 = mg.make_chrom(M, rf, w, lower=lower, upper=upper, weights=weights,
                            raw_M=(raw if mode != "raw" else None), cname="chr1")
        fea_ref, clist = ch2.getwindow(list(zip(gx.tolist(), gy.tolist())))
        fea_ref = np.asarray(fea_ref, np.float64).reshape(-1, F)
        fea_o, keep = onp.extract(Mf, e, w, gx, gy)
        clist = np.asarray(clist, np.int64).reshape(-1, 2)
        ok_g = (np.array_equal(clist[:, 0], gx[keep]) and np.array_equal(clist[:, 1], gy[keep])
                and fea_ref.shape == fea_o.shape
                and bool(((bits(fea_ref) == bits(fea_o)) | (np.isnan(fea_ref) & np.isnan(fea_o))).all()))
        if not ok_g:
            print("   getwindow MISMATCH: kept", clist.shape[0], keep.size)
            sys.exit(1)
        # ---- round 3: coordinates of ANY kind -- below the diagonal (near it and far), off the
        # matrix, near column 0 (scipy counts a negative column from the far end) -- and the
        # IndexError the reference's fancy index raises for a window row beyond the matrix
        k2 = 200
        ax = rng.integers(-2, n + 2, k2)
        style = rng.integers(0, 3, k2)
        ay = np.where(style == 0, ax - rng.integers(0, 3 * w + 2, k2),
             np.where(style == 1, rng.integers(-2, n + 2, k2), rng.integers(0, 2 * w + 1, k2)))
        passes = (ax - w >= 0) & (ay + w + 1 <= n)
        raising = passes & ((ax + w >= n) | (ay - w < -n))
        if raising.any():
            for fn in (lambda: ch2.getwindow(list(zip(ax.tolist(), ay.tolist()))), lambda: onp.extract(Mf, e, w, ax, ay)):
                try:
                    fn()
                    print("   a call with a window row beyond the matrix did not raise")
                    sys.exit(1)
                except IndexError:
                    pass
            ax, ay = ax[~raising], ay[~raising]
        fea_ref, clist = ch2.getwindow(list(zip(ax.tolist(), ay.tolist())))
        fea_ref = np.asarray(fea_ref, np.float64).reshape(-1, F)
        clist = np.asarray(clist, np.int64).reshape(-1, 2)
        fea_o, keep = onp.extract(Mf, e, w, ax, ay)
        ok_a = (np.array_equal(clist[:, 0], ax[keep]) and np.array_equal(clist[:, 1], ay[keep])
                and fea_ref.shape == fea_o.shape
                and bool(((bits(fea_ref) == bits(fea_o)) | (np.isnan(fea_ref) & np.isnan(fea_o))).all()))
        if not ok_a:
            print("   getwindow (any coordinates) MISMATCH: kept", clist.shape[0], keep.size)
            sys.exit(1)
        n_any = globals().get("_n_any", 0) + clist.shape[0]
        n_low = globals().get("_n_low", 0) + int(np.sum(clist[:, 0] > clist[:, 1]))
        globals()["_n_any"], globals()["_n_low"] = n_any, n_low
        nwin = clist.shape[0]
        npix += int(ref["ri"].size)
        ncand += int(len(cx))
        print("seed %4d w=%d n=%3d band=%2d lower=%2d upper=%3d %-7s trees=%2d thre=%.1f cands=%5d scored=%5d: exp %s band %s cands %s score %s" % (
            seed, w, n, band, lower, upper, mode, len(rf.estimators_), thre, len(cx), ref["ri"].size,
            "ok" if ok_e else "MISMATCH", "ok" if ok_b else "MISMATCH", "ok" if ok_c else "MISMATCH", "ok" if ok_s else "MISMATCH"),
              "getwindow %d/%d ok" % (nwin, k))
        sys.stdout.flush()
        if not (ok_e and ok_b and ok_c and ok_s):
            sys.exit(1)
    print("all %d chromosomes identical to the reference in %.0f s (%d candidates, %d scored pixels; "
          "getwindow on any coordinates: %d windows kept, %d of them below the diagonal)" % (
              n_cases, time.time() - t0, ncand, npix, globals().get("_n_any", 0), globals().get("_n_low", 0)))


if __name__ == "__main__":
    main()
