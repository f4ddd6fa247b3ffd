#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own Python.

Runs only in the build container (needs /root/reference, read-only).  The
reference imports `numba`, which is not installed here; its three @njit
helpers (peakachu/utils.py:180-237) are plain numpy Python, so `numba.njit`
is provided as the identity decorator (an in-memory module, nothing written
to disk).  One numerical caveat follows from that: numba's array.mean()
accumulates sequentially while numpy's is pairwise, which can differ by an
ulp on window[:w,:w].mean() (utils.py:228).  Only two `>` tests consume that
mean, so this script asserts that no fixture window lies near either
decision boundary; the build follows numba's (production) order.

Fixtures are data (inputs + the reference's outputs); no reference source is
copied.  Every file records the seed and library versions.
"""
import io
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

_nb = types.ModuleType("numba")
_nb.njit = lambda f=None, *a, **k: f if callable(f) else (lambda g: g)
sys.modules["numba"] = _nb
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import scipy  # noqa: E402
import sklearn  # noqa: E402
from scipy import sparse  # noqa: E402
from sklearn.ensemble import RandomForestClassifier  # noqa: E402

from peakachu import scoreUtils, trainUtils, utils as ref_utils  # noqa: E402
from peakachu_amd import synth  # noqa: E402

np.seterr(divide="ignore", invalid="ignore")  # as peakachu/score_genome.py:9

VERS = dict(numpy=np.__version__, scipy=scipy.__version__, sklearn=sklearn.__version__,
            reference="tariks/peakachu v2.3", numba="absent (njit = identity)")


def csr_parts(M, prefix):
    M = sparse.csr_matrix(M, dtype=np.float64)
    M.sum_duplicates()
    M.sort_indices()
    return {prefix + "_indptr": M.indptr.astype(np.int32),
            prefix + "_indices": M.indices.astype(np.int32),
            prefix + "_data": M.data.astype(np.float64),
            prefix + "_n": np.int64(M.shape[0])}


def sym_parts(M, prefix):
    """Symmetric integer-count matrix -> upper-triangle COO (small ints).
    tests/golden_io.py rebuilds the symmetric float64 CSR from it."""
    U = sparse.triu(sparse.csr_matrix(M), k=0).tocoo()
    finite = np.isfinite(U.data)
    assert np.all(U.data[finite] == np.round(U.data[finite])) and np.abs(U.data[finite]).max() < 2 ** 15
    val = np.where(finite, U.data, -1).astype(np.int16)  # -1 marks a NaN cell
    assert not np.any(U.data[finite] < 0)
    chk = sparse.csr_matrix(M) - sparse.csr_matrix(M).T
    assert chk.nnz == 0 or np.all(~np.isfinite(chk.data) | (chk.data == 0))
    return {prefix + "_urow": U.row.astype(np.int16 if M.shape[0] < 2 ** 15 else np.int32),
            prefix + "_ucol": U.col.astype(np.int16 if M.shape[0] < 2 ** 15 else np.int32),
            prefix + "_uval": val, prefix + "_n": np.int64(M.shape[0])}


def digest(M):
    """sha256 over a canonical CSR's three arrays (pins large derived
    matrices without storing them)."""
    import hashlib
    M = sparse.csr_matrix(M, dtype=np.float64)
    M.sum_duplicates(); M.sort_indices()
    h = hashlib.sha256()
    h.update(M.indptr.astype(np.int32).tobytes())
    h.update(M.indices.astype(np.int32).tobytes())
    h.update(M.data.astype(np.float64).tobytes())
    return np.array(h.hexdigest())


def save(name, **arrs):
    arrs["versions"] = np.array(repr(VERS))
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def holey_band(n, band, seed, decay=1.25, lam0=80.0, floor=0.04):
    """Test matrix with sparse far diagonals, dead bins and planted loops so
    every filter branch of distance_normalize is reachable."""
    rng = np.random.default_rng(seed)
    d = np.arange(band + 1)
    lam = lam0 / (1.0 + d) ** decay + floor
    cnt = rng.poisson(lam[:, None], size=(band + 1, n)).astype(np.float64)
    ii = np.broadcast_to(np.arange(n), cnt.shape)
    dd = np.broadcast_to(d[:, None], cnt.shape)
    ok = (ii + dd < n) & (cnt != 0)
    r = ii[ok]
    c = r + dd[ok]
    v = cnt[ok]
    dead = rng.choice(n, size=max(3, n // 60), replace=False)
    alive = ~(np.isin(r, dead) | np.isin(c, dead))
    r, c, v = r[alive], c[alive], v[alive]
    # loops
    L = max(4, n // 25)
    la = rng.integers(12, n - band - 12, size=L)
    ld = rng.integers(8, band - 8, size=L)
    extra = {}
    for a, q in zip(la, ld):
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                extra[(a + di, a + q + dj)] = 25.0 if (di or dj) else 40.0
    er = np.array([k[0] for k in extra]); ec = np.array([k[1] for k in extra])
    ev = np.array(list(extra.values()))
    r = np.r_[r, er]; c = np.r_[c, ec]; v = np.r_[v, ev]
    off = r != c
    R = np.r_[r, c[off]]; Cc = np.r_[c, r[off]]; V = np.r_[v, v[off]]
    M = sparse.csr_matrix((V, (R, Cc)), shape=(n, n), dtype=np.float64)
    M.sum_duplicates(); M.sort_indices()
    return M, np.stack([la, la + ld], 1), dead


def branch_stats(M, exp_arr, w, xi, yi):
    """Classify each coord by the branch it takes (test tooling; numpy)."""
    n = M.shape[0]
    S = 2 * w + 1
    stats = dict(off=0, sparse=0, ll0=0, p2ll=0, keep=0)
    margin = np.inf
    for x, y in zip(xi, yi):
        if not (x - w >= 0 and y + w + 1 <= n):
            stats["off"] += 1
            continue
        win = np.asarray(M[x - w:x + w + 1, y - w:y + w + 1].todense(), dtype=float)
        win[np.isnan(win)] = 0
        if np.count_nonzero(win) < win.size * 0.1:
            stats["sparse"] += 1
            continue
        acc = 0.0
        for v in win[:w, :w].ravel():
            acc += v
        ll = acc / (w * w)
        ll_np = win[:w, :w].mean()
        if not ll > 0:
            assert not ll_np > 0
            stats["ll0"] += 1
            continue
        p = win[w, w] / ll
        margin = min(margin, abs(p - 0.1) / 0.1)
        assert (p > 0.1) == (win[w, w] / ll_np > 0.1)
        if not p > 0.1:
            stats["p2ll"] += 1
            continue
        stats["keep"] += 1
    assert margin > 1e-9, "fixture window too close to the p2LL boundary"
    return stats


class _Dummy:
    """Stand-in model for fixtures that never call predict_proba."""
    feature_importances_ = np.zeros(121)


def make_chrom(M, model, w, lower=6, upper=100, weights=None, raw_M=None, cname="chrT"):
    return scoreUtils.Chromosome(M, model, raw_M=(M if raw_M is None else raw_M),
                                 weights=weights, lower=lower, upper=upper,
                                 cname=cname, res=10000, width=w)


# --------------------------------------------------------------------- G1
def g1_extract():
    """Chromosome.getwindow (scoreUtils.py:70-93) on hand-picked + random
    coordinates; raw integer matrices for w = 5, 6, 11."""
    for w, n, band, upper, ncoord, seed in ((5, 600, 120, 100, 330, 11),
                                            (6, 600, 120, 100, 230, 12),
                                            (11, 500, 140, 100, 50, 13)):
        M, loops, dead = holey_band(n, band, seed, decay=1.6)
        # carve an empty top-left w x w block next to a dense pixel so the
        # `ll_mean > 0` branch (utils.py:229) is hit for every w
        hx, hy = 300, 318
        M = M.tolil()
        M[hx - w:hx, hy - w:hy] = 0
        M[hy - w:hy, hx - w:hx] = 0
        M = M.tocsr()
        M.eliminate_zeros()
        ch = make_chrom(M, _Dummy(), w, upper=upper)
        rng = np.random.default_rng(seed + 100)
        x = rng.integers(0, n, size=ncoord)
        dd = rng.integers(0, upper + 1, size=ncoord)
        y = x + dd
        # special coordinates: matrix edges, the strict band limit (a window
        # corner at distance upper+2w is dropped by scoreUtils.py:31), loops,
        # dead bins, the main diagonal, the carved block
        sx = [0, w - 1, w, n - w - 1 - 20, 50, 60, 70, int(dead[0]), int(dead[1]) - 2, 200, hx]
        sy = [20, w + 10, w + 30, n - w - 1, 50 + upper, 60 + upper - 1, 70, int(dead[0]) + 9,
              int(dead[1]) + 9, 200, hy]
        x = np.r_[x, sx, loops[:12, 0], loops[:6, 0] + 1]
        y = np.r_[y, sy, loops[:12, 1], loops[:6, 1] - 1]
        okc = (y < n)  # scipy fancy-indexing raises on out-of-range columns
        x, y = x[okc], y[okc]
        coords = [(int(a), int(b)) for a, b in zip(x, y)]
        fea, clist = ch.getwindow(coords)
        st = branch_stats(ch.M, ch.exp_arr, w, x, y)
        print("G1 w=%d branches %s" % (w, st))
        assert all(v > 0 for v in st.values()), st
        assert fea.shape == (st["keep"], (2 * w + 1) ** 2)
        save("g1_extract_w%d.npz" % w, w=np.int32(w), seed=np.int64(seed), upper=np.int32(upper),
             exp_arr=ch.exp_arr, x=x.astype(np.int64), y=y.astype(np.int64),
             fea=np.asarray(fea, np.float64), clist=np.asarray(clist, np.int64),
             Mf_sha=digest(ch.M), **sym_parts(M, "M"))
    # balanced (non-integer) values, w=5: M = raw * w_i * w_j
    w, n, band, upper, seed = 5, 600, 120, 100, 21
    raw, loops, dead = holey_band(n, band, seed, decay=1.0, lam0=150.0, floor=0.3)
    wts = synth.synth_weights(n, seed)
    B = synth.balance(raw, wts)
    ch = make_chrom(B, _Dummy(), w, upper=upper, weights=wts, raw_M=raw)
    rng = np.random.default_rng(seed + 100)
    x = rng.integers(0, n - 1, size=150)
    y = np.minimum(x + rng.integers(0, upper + 1, size=150), n - 1)
    fea, clist = ch.getwindow([(int(a), int(b)) for a, b in zip(x, y)])
    st = branch_stats(ch.M, ch.exp_arr, w, x, y)
    print("G1 balanced branches", st)
    save("g1_extract_w5_balanced.npz", w=np.int32(w), seed=np.int64(seed), upper=np.int32(upper),
         exp_arr=ch.exp_arr, x=x.astype(np.int64), y=y.astype(np.int64), weights=wts,
         fea=np.asarray(fea, np.float64), clist=np.asarray(clist, np.int64),
         Mf_sha=digest(ch.M), **sym_parts(raw, "R"))


# --------------------------------------------------------------------- G8
def g8_any_coords():
    """Chromosome.getwindow (scoreUtils.py:70-93) on coordinates get_candidate never makes:
    lower-triangle pixels (x > y) near the diagonal -- answered from the stored diagonals
    -2w < col-row, cells below read 0 --, further below it (all-zero windows, dropped),
    windows whose columns start left of the matrix (x > y, y < w: scipy counts a negative
    column from the far end), coordinates the mask drops, and the IndexError scipy raises
    for a row x+w >= n."""
    out = {}
    for w, n, band, upper, seed in ((5, 400, 120, 100, 31), (6, 400, 120, 100, 32), (11, 300, 140, 100, 33)):
        M, loops, dead = holey_band(n, band, seed, decay=1.2)
        ch = make_chrom(M, _Dummy(), w, upper=upper)
        rng = np.random.default_rng(seed + 100)
        k = 260
        x = rng.integers(w, n - w, size=k)
        below = rng.integers(1, 3 * w + 3, size=k)          # x - y in [1, 3w+2]
        y = x - below
        # some upper-triangle and on-diagonal ones in between, masked ones, wrapped columns
        xs = np.r_[x, rng.integers(w, n - w, size=40), [w, w + 1, 2 * w, 30, 31, 3, -4, n + 5, 50]]
        ys = np.r_[y, xs[k:k + 40] + rng.integers(0, 2 * w, size=40), [0, 1, w - 1, 2, w - 2, 40, 9, n + 9, n - 2]]
        okm = (xs - w >= 0) & (ys + w + 1 <= n)
        assert not np.any(okm & ((xs + w >= n) | (ys - w < -n)))   # nothing here may raise
        coords = [(int(a), int(b)) for a, b in zip(xs, ys)]
        fea, clist = ch.getwindow(coords)
        fea = np.asarray(fea, np.float64).reshape(-1, (2 * w + 1) ** 2)
        clist = np.asarray(clist, np.int64).reshape(-1, 2)
        lower_kept = int(np.sum(clist[:, 0] > clist[:, 1]))
        wrapped_kept = int(np.sum((clist[:, 0] > clist[:, 1]) & (clist[:, 1] < w)))
        print("G8 w=%d: %d coords, %d kept, %d of them below the diagonal, %d with wrapped columns"
              % (w, len(coords), len(clist), lower_kept, wrapped_kept))
        assert lower_kept > 20
        # the reference raises where a window row leaves the matrix
        raised = False
        try:
            ch.getwindow([(n - w, n - 3 * w)])
        except IndexError:
            raised = True
        assert raised
        out.update({"w%d_x" % w: xs.astype(np.int64), "w%d_y" % w: ys.astype(np.int64),
                    "w%d_fea" % w: fea, "w%d_clist" % w: clist, "w%d_exp_arr" % w: ch.exp_arr,
                    "w%d_upper" % w: np.int32(upper), "w%d_seed" % w: np.int64(seed),
                    "w%d_raises" % w: np.array([n - w, n - 3 * w], np.int64),
                    "w%d_Mf_sha" % w: digest(ch.M)})
        out.update(sym_parts(M, "w%d_M" % w))
    save("g8_any_coords.npz", **out)


# ---------------------------------------------------------------- G2 + G5
def train_forest(M, loops, w, T, seed, class_weight=None, n_neg=700):
    n = M.shape[0]
    rng = np.random.default_rng(seed)
    pos = [(int(a), int(b)) for a, b in loops]
    nx = rng.integers(w, n - 140, size=n_neg)
    neg = [(int(a), int(a + q)) for a, q in zip(nx, rng.integers(w + 1, 100, size=n_neg))]
    fp = trainUtils.buildmatrix(M, pos, w=w)
    fn = trainUtils.buildmatrix(M, neg, w=w)
    X = np.r_[fp, fn]
    yl = np.r_[np.ones(len(fp)), np.zeros(len(fn))]
    # label noise so that trees grow beyond a handful of nodes
    flip = rng.random(yl.size) < 0.08
    yl = np.where(flip, 1 - yl, yl)
    rf = RandomForestClassifier(n_estimators=T, max_depth=20, max_features="sqrt",
                                n_jobs=1, random_state=seed, class_weight=class_weight)
    rf.fit(X, yl)
    return rf, X


def oracle_forest_arrays(rf):
    from oracle import oracle_np
    return oracle_np.forest_arrays(rf)


def g2_forest():
    """RandomForestClassifier.predict_proba(X)[:,1] (scoreUtils.py:109) for
    three class_weight settings; X is float32-representable, with NaNs."""
    w, n, band, seed = 5, 1500, 120, 31
    M, loops, dead = holey_band(n, band, seed, decay=1.0, lam0=150.0, floor=0.3)
    out = {}
    Xq = None
    for tag, T, cw in (("plain", 100, None), ("balanced", 12, "balanced"),
                       ("subsample", 6, "balanced_subsample")):
        rf, X = train_forest(M, loops, w, T, seed, cw)
        nodes = [e.tree_.node_count for e in rf.estimators_]
        depth = [e.tree_.max_depth for e in rf.estimators_]
        print("G2 %-9s T=%d nodes/tree mean %.0f max %d depth max %d" %
              (tag, T, np.mean(nodes), max(nodes), max(depth)))
        if Xq is None:
            rng = np.random.default_rng(seed + 5)
            Xq = np.r_[X[:200], rng.random((100, X.shape[1]))].astype(np.float32)
            # NaN features (sklearn 1.7.2 routes them by missing_go_to_left)
            Xq[5, :] = np.nan
            Xq[17, rng.integers(0, X.shape[1], 30)] = np.nan
            Xq[33, 60] = np.nan
        p = rf.predict_proba(Xq.astype(np.float64))[:, 1]
        fo = oracle_forest_arrays(rf)
        extra = dict(X=Xq) if tag == "plain" else {}
        save("g2_forest_%s.npz" % tag, seed=np.int64(seed), p=p, **extra,
             **{"fo_" + k: v for k, v in fo.items()})
        out[tag] = rf
    return out


def g5_buildmatrix():
    """trainUtils.buildmatrix (trainUtils.py:12-44) incl. NaN cells, the
    extra yi-xi>w mask and the `< 10 coords -> None` case."""
    w, n, band, seed = 5, 500, 120, 41
    M, loops, dead = holey_band(n, band, seed, decay=1.0, lam0=150.0, floor=0.3)
    M = M.tolil()
    M[100, 130] = np.nan; M[130, 100] = np.nan
    M[101, 131] = np.nan; M[131, 101] = np.nan
    M = M.tocsr()
    rng = np.random.default_rng(seed)
    x = rng.integers(0, n, size=120)
    y = np.minimum(x + rng.integers(0, 110, size=120), n - 1)
    x = np.r_[x, 100, 101, 99, 102]
    y = np.r_[y, 131, 130, 129, 133]
    coords = [(int(a), int(b)) for a, b in zip(x, y)]
    fea = trainUtils.buildmatrix(M, coords, w=w)
    mask = (x - w >= 0) & (y + w + 1 <= n) & (y - x > w)
    maxdis = int(np.abs(x[mask] - y[mask]).max()) + 2 * w
    exp_arr = ref_utils.calculate_expected(M, maxdis)
    few = trainUtils.buildmatrix(M, coords[:5], w=w)
    assert few is None
    save("g5_buildmatrix.npz", w=np.int32(w), seed=np.int64(seed), x=x.astype(np.int64),
         y=y.astype(np.int64), fea=np.asarray(fea, np.float64), exp_arr=exp_arr,
         maxdis=np.int64(maxdis), few_is_none=np.bool_(few is None), **sym_parts(M, "M"))


# ---------------------------------------------------------------- G3 + G4
def run_score(ch, thre):
    buf = io.StringIO()
    old = sys.stdout
    sys.stdout = buf
    try:
        result, R = ch.score(thre=thre)
    finally:
        sys.stdout = old
    with tempfile.NamedTemporaryFile("r", suffix=".bedpe", delete=False) as tf:
        path = tf.name
    os.remove(path)
    ch.writeBed(path, result, R)
    if os.path.exists(path):
        text = open(path).read()
        os.remove(path)
    else:
        text = ""
    r, c = result.nonzero()
    prob = np.asarray(result[r, c]).ravel() if r.size else np.zeros(0)
    sig = np.asarray(R[r, c]).ravel() if r.size else np.zeros(0)
    return dict(ri=r.astype(np.int64), ci=c.astype(np.int64), prob=prob, signal=sig,
                bedpe=np.array(text))


def chrom_state(ch):
    return dict(exp_arr=ch.exp_arr.copy(), background=ch.background.copy(),
                ridx=ch.ridx.astype(np.int64), cidx=ch.cidx.astype(np.int64),
                Mf_sha=digest(ch.M), Mf_nnz=np.int64(ch.M.nnz))


def g3_end_to_end(forests):
    """Chromosome.__init__/score/writeBed (scoreUtils.py:10-135) end to end
    with the G2 'plain' forest: raw mode, --minimum-prob 0, cooler-balanced
    mode with NaN weights, and the .hic-style M-is-not-raw_M mode."""
    rf = forests["plain"]
    w, n, band, upper, seed = 5, 900, 130, 100, 51
    raw, loops, dead = holey_band(n, band, seed, decay=1.0, lam0=150.0, floor=0.3)
    common = dict(w=np.int32(w), seed=np.int64(seed), lower=np.int32(6), upper=np.int32(upper),
                  res=np.int64(10000), forest=np.array("g2_forest_plain.npz"))
    for thre, tag in ((0.5, "raw"), (0.0, "raw_minprob0")):
        ch = make_chrom(raw, rf, w, upper=upper, cname="chr7")
        pre = chrom_state(ch)
        res = run_score(ch, thre)
        print("G3 %-13s candidates %d scored>thre %d" % (tag, pre["ridx"].size, res["ri"].size))
        save("g3_score_%s.npz" % tag, thre=np.float64(thre), cname=np.array("chr7"),
             mode=np.array("raw"), **common, **sym_parts(raw, "R"), **pre, **res)
    wts = synth.synth_weights(n, seed)
    B = synth.balance(raw, wts)
    ch = make_chrom(B, rf, w, upper=upper, weights=wts, raw_M=raw, cname="chrX")
    pre = chrom_state(ch)
    res = run_score(ch, 0.5)
    print("G3 weights       candidates %d scored>thre %d" % (pre["ridx"].size, res["ri"].size))
    save("g3_score_weights.npz", thre=np.float64(0.5), cname=np.array("chrX"),
         mode=np.array("weights"), weights=wts, **common, **sym_parts(raw, "R"), **pre, **res)
    # .hic-style: normalised matrix without weights, separate raw matrix
    # (peakachu/scoreUtils.py:18-21, M is not raw_M).  M = finite(B) * 200.
    Bf = sparse.csr_matrix(B)
    Bf.data = np.where(np.isfinite(Bf.data), Bf.data, 0.0)
    Bf.eliminate_zeros()
    Bf = Bf * 200.0
    ch = make_chrom(Bf, rf, w, upper=upper, weights=None, raw_M=raw, cname="chr2")
    pre = chrom_state(ch)
    res = run_score(ch, 0.5)
    print("G3 hic-style     candidates %d scored>thre %d" % (pre["ridx"].size, res["ri"].size))
    save("g3_score_hicstyle.npz", thre=np.float64(0.5), cname=np.array("chr2"),
         mode=np.array("hicstyle"), weights=wts, **common, **sym_parts(raw, "R"), **pre, **res)
    return raw, rf


def g4_batch_quirk(raw, rf):
    """scoreUtils.py:108 `if fea.shape[0] > 1`: a batch with exactly one
    surviving window is dropped.  (a) a one-candidate list; (b) 100 001
    candidates whose second batch (size 1) survives the filters."""
    w, upper = 5, 100
    ch = make_chrom(raw, rf, w, upper=upper, cname="chr7")
    good_x, good_y = ch.ridx.copy(), ch.cidx.copy()
    fea, clist = ch.getwindow(list(zip(good_x, good_y)))
    clist = np.asarray(clist)
    p = rf.predict_proba(fea)[:, 1]
    hot = clist[np.argmax(p)]  # survives and scores high
    assert p.max() > 0.5
    ch.ridx, ch.cidx = np.array([hot[0]]), np.array([hot[1]])
    res_a = run_score(ch, 0.5)
    assert res_a["ri"].size == 0
    # (b): first batch = every real candidate except the hot pixel, padded
    # with coords in the empty far-diagonal area (filtered as too sparse);
    # last batch = the hot pixel alone
    ch = make_chrom(raw, rf, w, upper=upper, cname="chr7")
    n = raw.shape[0]
    nothot = ~((good_x == hot[0]) & (good_y == hot[1]))
    good_x, good_y = good_x[nothot], good_y[nothot]
    npad = 100000 - good_x.size
    px = w + (np.arange(npad) % (n - 140 - 2 * w))
    py = px + 134  # beyond the populated band -> empty windows
    bx = np.r_[good_x, px, hot[0]]
    by = np.r_[good_y, py, hot[1]]
    assert bx.size == 100001
    ch.ridx, ch.cidx = bx, by
    res_b = run_score(ch, 0.5)
    assert not np.any((res_b["ri"] == hot[0]) & (res_b["ci"] == hot[1]))
    save("g4_batch_quirk.npz", w=np.int32(w), upper=np.int32(upper), thre=np.float64(0.5),
         hot=hot.astype(np.int64), exp_arr=ch.exp_arr, bx=bx.astype(np.int32),
         by=by.astype(np.int32), a_n=np.int64(res_a["ri"].size),
         b_ri=res_b["ri"], b_ci=res_b["ci"], b_prob=res_b["prob"], b_signal=res_b["signal"],
         forest=np.array("g2_forest_plain.npz"), **sym_parts(raw, "R"))
    print("G4 (a) kept %d, (b) kept %d of 100001" % (res_a["ri"].size, res_b["ri"].size))


def g6_driver(forests):
    """The reference's own score_genome.main / score_chromosome.main
    (peakachu/score_genome.py:3-84, score_chromosome.py:3-71) run against a
    multi-chromosome container through a file-backed stand-in for `cooler`
    (the package is not installed): chromosome filter "# X", chrM excluded,
    an un-prefixed label, chromosome order, balanced and raw modes."""
    import joblib
    from peakachu import score_genome, score_chromosome
    from peakachu_amd import io as pkio
    rf = forests["plain"]
    chroms = {}
    for i, (name, n) in enumerate((("chr1", 420), ("chr2", 380), ("3", 300), ("chrX", 340),
                                   ("chrM", 150))):
        raw, loops, dead = holey_band(n, 90, 61 + i, decay=1.0, lam0=150.0, floor=0.3)
        chroms[name] = (raw, synth.synth_weights(n, 61 + i, n_nan=3))
    cont = os.path.join(OUT, "g6_genome.pkmap.npz")
    pkio.write_pkmap(cont, chroms, resolution=10000)
    fake = types.ModuleType("cooler")
    fake.Cooler = pkio.PkMap
    sys.modules["cooler"] = fake
    tmpd = tempfile.mkdtemp()
    mpath = os.path.join(tmpd, "model.pkl")
    joblib.dump(rf, mpath)
    out = {}

    class A:
        pass
    for tag, wname, chrsel in (("weight", "weight", ["#", "X"]), ("raw", "raw", ["#", "X"]),
                               ("all", "raw", [])):
        a = A()
        a.output = os.path.join(tmpd, tag + ".bedpe")
        a.model, a.path, a.clr_weight_name = mpath, cont, wname
        a.chroms, a.lower, a.upper, a.resolution, a.minimum_prob = chrsel, 6, 80, 10000, 0.5
        buf = io.StringIO(); old = sys.stdout; sys.stdout = buf
        try:
            score_genome.main(a)
        finally:
            sys.stdout = old
        out["genome_" + tag] = np.array(open(a.output).read() if os.path.exists(a.output) else "")
        print("G6 score_genome %-6s %d lines" % (tag, str(out["genome_" + tag]).count("\n")))
    a = A()
    a.output = os.path.join(tmpd, "c.bedpe")
    a.model, a.path, a.clr_weight_name, a.chrom = mpath, cont, "weight", "3"
    a.lower, a.upper, a.resolution, a.minimum_prob = 6, 80, 10000, 0.5
    buf = io.StringIO(); old = sys.stdout; sys.stdout = buf
    try:
        score_chromosome.main(a)
    finally:
        sys.stdout = old
    out["chrom_3_weight"] = np.array(open(a.output).read() if os.path.exists(a.output) else "")
    save("g6_driver.npz", forest=np.array("g2_forest_plain.npz"),
         container=np.array("g6_genome.pkmap.npz"), lower=np.int32(6), upper=np.int32(80),
         **out)
    del sys.modules["cooler"]


def pool_input(seed, n_chrom=3, res=10000):
    """A scored-pixel bedpe as score_genome writes it: blobs of pixels around planted
    loops (probabilities 0.5..1, decaying away from the summit), stripes (many pixels on
    one row / column, which is what find_anchors looks for), isolated pixels, and ties."""
    rng = np.random.default_rng(seed)
    lines = []
    for c in range(n_chrom):
        chrom = "chr%d" % (c + 1) if c < n_chrom - 1 else "chrX"
        pix = {}
        n = 2500
        for _ in range(40):                       # blobs
            a = int(rng.integers(20, n - 300)); d = int(rng.integers(8, 250))
            rad = int(rng.integers(1, 5)); peak = rng.uniform(0.75, 1.0)
            sig = rng.uniform(5, 60)
            for di in range(-rad, rad + 1):
                for dj in range(-rad, rad + 1):
                    if rng.random() < 0.75 - 0.08 * (abs(di) + abs(dj)):
                        pr = peak - 0.05 * (abs(di) + abs(dj)) - rng.uniform(0, 0.05)
                        if pr > 0.5:
                            pix[(a + di, a + d + dj)] = (pr, sig * (1 - 0.1 * (abs(di) + abs(dj))))
        for _ in range(6):                        # stripes along a row or a column
            a = int(rng.integers(20, n - 300)); horiz = rng.random() < 0.5
            for k in range(int(rng.integers(6, 25))):
                o = a + 8 + 2 * k + int(rng.integers(0, 2))
                key = (a, o) if horiz else (o - 200 if o - 200 > 0 else o, a + 300)
                if key[0] < key[1]:
                    pix[key] = (rng.uniform(0.55, 1.0), rng.uniform(3, 40))
        for _ in range(120):                      # isolated pixels
            a = int(rng.integers(20, n - 300)); d = int(rng.integers(8, 250))
            pix[(a, a + d)] = (rng.uniform(0.5, 1.0), float(rng.integers(2, 30)))
        keys = sorted(pix)
        for k in keys[::17]:                      # exact ties in value
            pix[k] = (pix[k][0], 12.0)
        for (i, j) in keys:
            pr, sg = pix[(i, j)]
            lines.append("\t".join([chrom, str(i * res), str(i * res + res), chrom, str(j * res),
                                    str(j * res + res), str(np.float64(pr)), str(np.float64(sg))]))
    return "\n".join(lines) + "\n"


def g7_pool():
    """`peakachu pool` (peakachu/call_loops.py:3-26, peakachu/peakacluster.py:7-172):
    the reference's own main() on synthetic scored-pixel files."""
    import argparse
    from peakachu import call_loops, peakacluster
    tmp = tempfile.mkdtemp()
    out = {}
    for seed, res in ((1, 10000), (2, 5000)):
        text = pool_input(seed, res=res)
        fin = os.path.join(tmp, "in%d.bedpe" % seed)
        open(fin, "w").write(text)
        out["in%d" % seed] = np.frombuffer(text.encode(), np.uint8)
        out["res%d" % seed] = np.int64(res)
        for thre in (0.9, 0.5, 0.97):
            fo = os.path.join(tmp, "out.bedpe")
            call_loops.main(argparse.Namespace(resolution=res, infile=fin, outfile=fo, threshold=thre))
            got = open(fo).read()
            out["out%d_t%g" % (seed, thre)] = np.frombuffer(got.encode(), np.uint8)
            print("g7 seed %d thre %g: %d -> %d lines" % (seed, thre, text.count("\n"), got.count("\n")))
        # function level: the representatives local_clustering returns for one chromosome
        D = {}
        for line in text.splitlines():
            q = line.split()
            if q[0] == "chr1" and float(q[6]) >= 0.5:
                D[(int(q[1]) // res, int(q[4]) // res)] = float(q[7])
        reps = sorted(set(t[0] for t in peakacluster.local_clustering(D, min_count=3, r=2)))
        out["reps%d" % seed] = np.array(reps, np.int64)
        xa = sorted(peakacluster.find_anchors(np.r_[[k[0] for k in D]], min_count=3, min_dis=2))
        out["xanchors%d" % seed] = np.array(xa, np.int64)
    save("g7_pool.npz", seeds=np.array([1, 2]), **out)


def main():
    """`make_golden.py` regenerates everything; `make_golden.py g6` only the
    named groups (g2 is always run: the later groups need its forest)."""
    os.makedirs(OUT, exist_ok=True)
    want = set(sys.argv[1:]) or {"g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8"}
    if "g8" in want:
        g8_any_coords()
        if want == {"g8"}:
            return
    if "g7" in want:
        g7_pool()
        if want == {"g7"}:
            return
    if "g1" in want:
        g1_extract()
    global save
    real_save = save
    if "g2" not in want:
        save = lambda *a, **k: None  # train the forest without rewriting its fixture
    forests = g2_forest()
    save = real_save
    if "g5" in want:
        g5_buildmatrix()
    if "g3" in want or "g4" in want:
        if "g3" not in want:
            save = lambda *a, **k: None
        raw, rf = g3_end_to_end(forests)
        save = real_save
        if "g4" in want:
            g4_batch_quirk(raw, rf)
    if "g6" in want:
        g6_driver(forests)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("total %.1f KB" % (tot / 1024))


if __name__ == "__main__":
    main()
