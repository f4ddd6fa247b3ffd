"""Parity of the HIP path against the reference's golden vectors and the CPU
oracle, through the C ABI.  Bit-exact: float64 features, pixel indices,
probabilities, signal.  Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest

import golden_io as gio
from oracle import oracle_np as onp
from peakachu_amd import _lib, synth, utils
from peakachu_amd.forest import FlatForest

pytestmark = pytest.mark.gpu


def flat(fo):
    return FlatForest(int(fo["F"]), fo["tree_off"], fo["left"], fo["right"], fo["feat"],
                      fo["thr"], fo["miss_left"], fo["p1"])


def hip_matrix(Mf, exp_arr, w, upper, options=None):
    """(options: this handle's own -- no test sets a process-wide option any more, round 4)"""
    Mf = utils.canonical_csr(Mf)
    return _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, Mf.shape[0], exp_arr,
                          -2 * w + 1, upper + 2 * w - 1, options=options)


@pytest.mark.parametrize("pair", [1, 0, 2])
@pytest.mark.parametrize("name", ["g1_extract_w5.npz", "g1_extract_w6.npz", "g1_extract_w11.npz",
                                  "g1_extract_w5_balanced.npz"])
def test_extract_golden(hip_lib, name, pair):
    """The extract kernels for w=5/6 (two lanes per candidate with the strip staged in LDS -- the default --,
    two lanes per candidate gathering into registers [pair = 2], one lane per candidate [0]) and the LDS
    kernel for w=11."""
    _extract_golden(name, pair)


def _extract_golden(name, pair=1):
    z = gio.load(name)
    L = _lib.load()
    clean0 = L.pk_get_option(b"stat_extract_clean")
    w, upper = int(z["w"]), int(z["upper"])
    if "weights" in z.files:
        M = gio.balance(gio.sym_matrix(z, "R"), z["weights"])
    else:
        M = gio.sym_matrix(z, "M")
    Mf = utils.band_filter(M, w, upper)
    assert gio.digest(Mf) == str(z["Mf_sha"])
    ok = z["x"] <= z["y"]
    x, y = z["x"][ok], z["y"][ok]
    strip0 = L.pk_get_option(b"stat_extract_strip")
    hm = hip_matrix(Mf, z["exp_arr"], w, upper, options={"extract_pair": 1 if pair else 0, "extract_strip": 0 if pair == 2 else 2})
    f64, f32, keep = hm.extract(w, x, y, want64=True, want32=True)
    if w in (5, 6) and "balanced" in name:   # (integer counts of the other fixtures qualify as well; this one must)
        assert (L.pk_get_option(b"stat_extract_strip") > strip0) == (pair == 1)
    assert np.array_equal(np.stack([x[keep], y[keep]], 1), z["clist"])
    assert np.array_equal(gio.bits(f64), gio.bits(z["fea"]))
    assert np.array_equal(f32, z["fea"].astype(np.float32))
    if int(z["w"]) in (5, 6) and hm.get_option("extract_pair") and "balanced" in name:
        # balanced (float, ~1e-3) values still qualify for the pre-divided band
        assert L.pk_get_option(b"stat_extract_clean") > clean0


@pytest.mark.parametrize("w", [5, 6, 11])
def test_extract_any_coordinates_golden(hip_lib, w):
    """G8: the reference's getwindow on lower-triangle coordinates (x > y: served from the
    stored diagonals), coordinates its mask drops, windows whose columns wrap around column
    0, through pk_extract and through Chromosome.getwindow; IndexError where scipy raises."""
    from peakachu_amd import scoreUtils as su
    z = gio.load("g8_any_coords.npz")
    p = "w%d_" % w
    upper = int(z[p + "upper"])
    M = gio.sym_matrix(z, p + "M")
    Mf = utils.band_filter(M, w, upper)
    assert gio.digest(Mf) == str(z[p + "Mf_sha"])
    x, y = z[p + "x"], z[p + "y"]
    hm = hip_matrix(Mf, z[p + "exp_arr"], w, upper)
    f64, f32, keep = hm.extract(w, x, y, want64=True, want32=True)
    assert np.array_equal(np.stack([x[keep], y[keep]], 1), z[p + "clist"])
    assert np.array_equal(gio.bits(f64), gio.bits(z[p + "fea"]))
    assert np.array_equal(f32, z[p + "fea"].astype(np.float32))
    with pytest.raises(_lib.PeakachuHipError, match="IndexError"):
        hm.extract(w, z[p + "raises"][:1], z[p + "raises"][1:])

    class _M:  # getwindow never calls the model
        feature_importances_ = np.zeros((2 * w + 1) ** 2)
    ch = su.Chromosome(M, _M(), raw_M=M, lower=6, upper=upper, width=w)
    assert np.array_equal(gio.bits(ch.exp_arr), gio.bits(z[p + "exp_arr"]))
    fea, clist = ch.getwindow([(int(a), int(b)) for a, b in zip(x, y)])
    assert np.array_equal(clist, z[p + "clist"])
    assert np.array_equal(gio.bits(fea), gio.bits(z[p + "fea"]))
    with pytest.raises(IndexError):
        ch.getwindow([tuple(int(v) for v in z[p + "raises"])])
    # a standard call afterwards still takes the fast kernels
    ok = (x <= y) & (x >= 0) & (y < M.shape[0])
    f64b, _, keepb = hm.extract(w, x[ok], y[ok])
    sel = np.flatnonzero(ok)[keepb]
    assert np.array_equal(gio.bits(f64b), gio.bits(f64[np.searchsorted(keep, sel)]))


@pytest.mark.parametrize("tag", ["plain", "balanced", "subsample"])
@pytest.mark.parametrize("ilp,lds,slots,img,q", [
    (4, 0, 8, 0, 0), (1, 0, 8, 0, 0), (8, 0, 8, 0, 0),
    (4, 160, 8, 0, 0), (4, 160, 4, 0, 0), (4, 160, 2, 0, 0),
    (4, 160, 6, 0, 0), (4, 2, 8, 0, 0), (4, 1, 4, 0, 0),
    (4, 160, 0, 1, 0), (4, 160, 2, 1, 0), (4, 160, 4, 1, 0),
    (4, 160, 5, 1, 0), (4, 160, 6, 1, 0), (4, 160, 7, 1, 0),
    (4, 160, 8, 1, 0),
    # q = walks per lane of the rank kernel + 1 (1 = automatic)
    (4, 160, 0, 1, 1), (4, 160, 4, 1, 5), (4, 160, 9, 1, 5), (4, 160, 16, 1, 5),
    (4, 160, 0, 1, 3), (4, 160, 2, 1, 3), (4, 160, 3, 1, 3), (4, 160, 13, 1, 3),
    (4, 160, 16, 1, 3)])
def test_forest_golden(hip_lib, tag, ilp, lds, slots, img, q):
    """Every forest kernel variant: no LDS at all (lds=0: the gmem kernel) with 1/4/8 chains per
    lane; trees streamed through LDS in barrier-separated groups with 2..8
    tree slots; a tree buffer so small (1-2 KiB) that some trees are walked
    from global memory; the LDS-image kernel (img=1: fixed-depth walks
    over absolute LDS addresses) with automatic and forced slot counts; the rank kernel (q>0: 16-bit rank
    codes, 4-byte nodes, 2 or 4 walks per lane) with automatic and forced shapes."""
    z = gio.load("g2_forest_%s.npz" % tag)
    X = gio.load("g2_forest_plain.npz")["X"]
    hf = _lib.HipForest(flat(gio.forest(z)), options={
        "forest_img": img, "forest_q": 1 if q else 0, "forest_q_ch": q - 1 if q > 1 else 0, "forest_ilp": ilp,
        "forest_lds": lds, "forest_slots": slots})
    p = hf.predict(X)
    assert np.array_equal(gio.bits(p), gio.bits(z["p"]))


@pytest.mark.parametrize("w", [5, 6, 11, 3])
def test_extract_with_the_taps_of_another_numpy(hip_lib, w):
    """pk_set_gauss_taps: with the Gaussian taps numpy 1.26.4 computes (two of five differ by
    one ulp from numpy 2.2's; tests/golden/gauss_scipy171.npz) every extractor kernel gives
    the float64 features the CPU restatement gives with the same taps -- i.e. the reference
    of THAT environment -- and they differ from the default ones."""
    L = _lib.load()
    k171 = gio.load("gauss_scipy171.npz")["taps"][4:].copy()
    k0 = _lib.gauss_taps()
    M, _ = synth.synth_band(900, 120, seed=w)
    upper = 100
    e = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, w + 1, upper)
    x, y = x[::7], y[::7]
    hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, Mf.shape[0], e, -2 * w + 1, upper + 2 * w - 1)
    fea0, _, keep0 = hm.extract(w, x, y)
    try:
        _lib.check(L.pk_set_gauss_taps(k171), "taps")
        onp.set_gauss_taps(k171)
        fea1, _, keep1 = hm.extract(w, x, y)
        ref, rkeep = onp.extract(Mf, e, w, x, y)
    finally:
        _lib.check(L.pk_set_gauss_taps(k0), "taps")
        onp.set_gauss_taps(k0)
    assert np.array_equal(keep1, rkeep) and np.array_equal(gio.bits(fea1), gio.bits(ref))
    assert np.array_equal(keep0, keep1) and not np.array_equal(gio.bits(fea0), gio.bits(fea1))
    assert np.abs(fea0 - fea1).max() < 1e-14
    fea2 = hm.extract(w, x, y)[0]  # the defaults are back
    assert np.array_equal(gio.bits(fea2), gio.bits(fea0))
    bad = np.array([0.4, 0.5, 0.1, 0.01, 0.001])
    assert L.pk_set_gauss_taps(bad) != 0  # not decreasing


@pytest.mark.parametrize("q", [1, 0])
@pytest.mark.parametrize("tag,name", [("balanced", "old_sklearn_rf_balanced.xz.joblib"),
                                      ("plain", "old_sklearn_rf_plain.xz.joblib")])
def test_old_sklearn_pickle_on_device(hip_lib, tag, name, q):
    """A model file written by scikit-learn 0.24.2 / joblib 1.1.0 (counts in tree_.value, no
    missing_go_to_left; tools/make_old_sklearn_fixture.py) -> load_model -> HIP forest: that
    scikit-learn's own predict_proba[:, 1], bit for bit (rank kernel and float kernels)."""
    import os
    from peakachu_amd.forest import load_model
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "old_sklearn_rf.npz"))
    ff = load_model(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))
    p = _lib.HipForest(ff, options={"forest_q": q}).predict(z["X"])
    assert np.array_equal(gio.bits(p), gio.bits(z["p_" + tag]))


@pytest.mark.parametrize("name,opts", [("forest_w5_t100.npz", {}), ("forest_w5_t100.npz", {"forest_q_wpt": 1}),
                                       ("forest_w5_t100.npz", {"forest_q_ch": 2}),
                                       ("forest_w5_t100.npz", {"forest_q": 0}),
                                       ("forest_w6_t100.npz", {}), ("forest_w6_t100.npz", {"forest_slots": 5})])
def test_forest_threshold_edges(hip_lib, name, opts):
    """Features sitting exactly on split thresholds, one ulp below and above, +-inf, -0.0
    and NaN, on the benchmark forests: the rank quantizer (lookup cells + exact compare)
    and the 4-byte-node walk must decide every split like sklearn's float compare."""
    import os
    from peakachu_amd.forest import FlatForest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ff = FlatForest.load(os.path.join(root, "peakachu_amd", "data", name))
    fo = {k: getattr(ff, k) for k in FlatForest.FIELDS}
    rng = np.random.default_rng(11)
    N = 3000
    X = rng.random((N, ff.F)).astype(np.float32)
    inner = np.flatnonzero(ff.left != -1)
    t32 = ff.thr[inner].astype(np.float32)
    t32 = np.where(t32.astype(np.float64) > ff.thr[inner], np.nextafter(t32, np.float32(-np.inf)), t32)
    for k in range(N):
        for j in rng.choice(inner.size, 12, replace=False):
            t = t32[j]
            X[k, ff.feat[inner[j]]] = (t, np.nextafter(t, np.float32(-np.inf)),
                                       np.nextafter(t, np.float32(np.inf)))[(k + j) % 3]
    X[5, 3] = np.nan
    X[6, :] = np.nan
    X[7, 0] = np.inf
    X[8, 1] = -np.inf
    X[9, 2] = -0.0
    X[10, :] = 0.0
    X[11, :] = 1.0
    ref = onp.predict(fo, X)
    p = _lib.HipForest(ff, options=opts).predict(X)
    assert np.array_equal(gio.bits(p), gio.bits(ref))


@pytest.mark.parametrize("opts", [{}, {"forest_q_early": 1}, {"forest_q_wpt": 1}, {"forest_q_ch": 2},
                                  {"forest_q_prio": 0}, {"forest_q_persist": 0}, {"forest_q_persist": -1},
                                  {"forest_q_persist": -5},
                                  {"forest_q_persist": -3, "forest_q_ch": 2},
                                  {"forest_slots": 5}, {"forest_slots": 5, "forest_q_early": 1},
                                  {"forest_slots": 3, "forest_q_early": 1}])
@pytest.mark.parametrize("name", ["forest_w5_t100.npz", "forest_w6_t100.npz"])
def test_forest_q_modes(hip_lib, name, opts):
    """The rank kernel's shapes (early staging through fixed tree slots or packed groups,
    one or two waves per tree, 2 or 4 walks per lane, forced group sizes, pruning) on the
    benchmark forests: random features incl. exact 0 / 1 and NaN rows, 2 999 rows so that
    the last workgroup is partial, bit-exact against the oracle."""
    import os
    from peakachu_amd.forest import FlatForest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ff = FlatForest.load(os.path.join(root, "peakachu_amd", "data", name))
    fo = {k: getattr(ff, k) for k in FlatForest.FIELDS}
    rng = np.random.default_rng(21)
    X = rng.random((2999, ff.F)).astype(np.float32)
    X[::5] = (X[::5] > 0.5).astype(np.float32)
    X[7, 3] = np.nan
    X[300, :] = np.nan
    X[2998, 0] = np.nan
    ref = onp.predict(fo, X)
    p = _lib.HipForest(ff, options={k: v for k, v in opts.items() if k != "early_exit"}).predict(X)
    assert np.array_equal(gio.bits(p), gio.bits(ref))


@pytest.mark.parametrize("opts", [{}, {"forest_slots": 5}, {"forest_q_persist": -2}, {"forest_q_prio": 0},
                                  {"forest_q": 0}, {"forest_q_two": 0}, {"forest_q_two": 0, "forest_q_persist": -2},
                                  {"forest_q_persist": 0}, {"forest_slots": 16}, {"forest_q_help": 0},
                                  {"forest_slots": 5, "forest_q_help": 0}, {"forest_slots": 9},
                                  {"forest_slots": 12}, {"forest_slots": 3, "forest_q_persist": -2}])
@pytest.mark.parametrize("F,with_miss", [(529, False), (300, False), (529, True)])
def test_forest_wide_format(hip_lib, F, with_miss, opts):
    """More than 255 features (w = 11: 529): the rank kernel on 64-candidate tiles with the wide
    node word (10-bit feature, 11-bit pair index), two tiles per workgroup trip (forest_q2_kernel:
    the second one waits in registers; the groups staged by the waves without a tree when there are
    enough of them -- forest_slots 3 / 5 / 9 -- else by every thread) or one -- with and without
    missing_go_to_left nodes (the wide word keeps that flag as the parity of the pair index).
    Rows with exact 0 / 1,
    values on thresholds, NaN cells and all-NaN rows; bit-exact against the oracle."""
    from test_forest_qimage import _random_forest
    fo = _random_forest(F, 40, 1500, 16, seed=F + (7 if with_miss else 0))
    if with_miss:
        fo["miss_left"][::3] = 1
    ff = FlatForest(F, *(fo[k] for k in FlatForest.FIELDS))
    rng = np.random.default_rng(F)
    X = rng.random((1301, F)).astype(np.float32)
    X[::5] = (X[::5] > 0.5).astype(np.float32)
    inner = np.flatnonzero(fo["left"] != -1)
    for k in range(400):
        j = inner[k * 13 % inner.size]
        t = np.float32(fo["thr"][j])
        X[100 + k, fo["feat"][j]] = (t, np.nextafter(t, np.float32(-np.inf)), np.nextafter(t, np.float32(np.inf)))[k % 3]
    X[7, 3] = np.nan
    X[300, :] = np.nan
    X[1300, F - 1] = np.nan
    ref = onp.predict(fo, X)
    L = _lib.load()
    L.pk_prof_enable(1)
    L.pk_prof_reset()
    try:
        p = _lib.HipForest(ff, options=opts).predict(X)
        quant_launches = _lib.prof_get("quant")[1]
    finally:
        L.pk_prof_enable(0)
    assert np.array_equal(gio.bits(p), gio.bits(ref))
    # the rank path (its quantizer) ran whenever it was allowed to: missing_go_to_left nodes fit the
    # wide word since round 4 (the flag is the parity of the node's pair index)
    assert (quant_launches > 0) == (opts.get("forest_q", 1) != 0)


def _g3_matrix(z):
    raw = gio.sym_matrix(z, "R")
    mode = str(z["mode"])
    if mode == "raw":
        return raw
    if mode == "weights":
        return gio.balance(raw, z["weights"])
    return gio.hicstyle(raw, z["weights"])


@pytest.mark.parametrize("name", ["g3_score_raw.npz", "g3_score_raw_minprob0.npz",
                                  "g3_score_weights.npz", "g3_score_hicstyle.npz"])
def test_score_golden(hip_lib, name):
    z = gio.load(name)
    w, upper = int(z["w"]), int(z["upper"])
    Mf = utils.band_filter(_g3_matrix(z), w, upper)
    hm = hip_matrix(Mf, z["exp_arr"], w, upper)
    hf = _lib.HipForest(flat(gio.forest(str(z["forest"]))))
    ox, oy, op, osig = hm.score(hf, w, float(z["thre"]), z["ridx"], z["cidx"])
    order = np.lexsort((oy, ox))
    assert np.array_equal(ox[order], z["ri"]) and np.array_equal(oy[order], z["ci"])
    assert np.array_equal(gio.bits(op[order]), gio.bits(z["prob"]))
    assert np.array_equal(gio.bits(osig[order]), gio.bits(z["signal"]))


def test_batch_quirk_golden(hip_lib):
    z = gio.load("g4_batch_quirk.npz")
    w, upper = int(z["w"]), int(z["upper"])
    Mf = utils.band_filter(gio.sym_matrix(z, "R"), w, upper)
    hm = hip_matrix(Mf, z["exp_arr"], w, upper)
    hf = _lib.HipForest(flat(gio.forest(str(z["forest"]))))
    hot = z["hot"]
    ox, oy, op, osig = hm.score(hf, w, 0.5, hot[:1], hot[1:])
    assert ox.size == 0
    ox, oy, op, osig = hm.score(hf, w, 0.5, z["bx"], z["by"])
    order = np.lexsort((oy, ox))
    assert np.array_equal(ox[order], z["b_ri"]) and np.array_equal(oy[order], z["b_ci"])
    assert np.array_equal(gio.bits(op[order]), gio.bits(z["b_prob"]))
    assert np.array_equal(gio.bits(osig[order]), gio.bits(z["b_signal"]))


@pytest.mark.parametrize("w,seed,batch", [(5, 1, 100000), (5, 2, 777), (6, 3, 5000), (11, 4, 100000)])
def test_score_vs_oracle_synthetic(hip_lib, w, seed, batch):
    """Seeded synthetic band matrix, every non-zero band pixel a candidate,
    a forest trained here on the fly is not needed: reuse the golden forest
    for w=5 and a random forest of matching width otherwise."""
    upper = 70
    M, loops = synth.synth_band(900 if w < 11 else 400, 90, seed=seed)
    exp_arr = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, w + 1, upper)
    if w == 11:
        x, y = x[::7], y[::7]
    F = (2 * w + 1) ** 2
    fo = gio.forest("g2_forest_plain.npz") if w == 5 else random_forest_arrays(F, 40, seed)
    hm = hip_matrix(Mf, exp_arr, w, upper)
    hf = _lib.HipForest(flat(fo))
    ox, oy, op, osig = hm.score(hf, w, 0.3, x, y, batch=batch)
    rx, ry, rp, rs = onp.score(Mf, exp_arr, w, fo, 0.3, x, y, batch=batch, threads=8)
    assert rx.size > 10
    assert np.array_equal(ox, rx) and np.array_equal(oy, ry)
    assert np.array_equal(gio.bits(op), gio.bits(rp))
    assert np.array_equal(gio.bits(osig), gio.bits(rs))


@pytest.mark.parametrize("opts", [dict(overlap=1), dict(overlap=1, chunk=4096), dict(chunk=4096), dict()])
def test_score_twice_with_other_coordinates(hip_lib, opts):
    """pk_score keeps its device-side candidate list between calls and lends it the caller's
    host coordinates: a second call with OTHER coordinates (fewer, so the cached list is
    reused) must score those, under every way the coordinates can travel (up front with
    the two-stream pipeline, chunk by chunk behind the kernels)."""
    w, upper = 5, 70
    M, loops = synth.synth_band(900, 90, seed=11)
    exp_arr = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, w + 1, upper)
    fo = gio.forest("g2_forest_plain.npz")
    hm = hip_matrix(Mf, exp_arr, w, upper, options=opts)  # (pk_score has no candidate handle: the matrix's options)
    hf = _lib.HipForest(flat(fo), options=opts)
    lists = [(x, y), (x[1::3].copy(), y[1::3].copy()), (x[::-1][: x.size // 2].copy(), y[::-1][: x.size // 2].copy()),
             (x[5:6].copy(), y[5:6].copy()), (x, y)]
    for xs, ys in lists:
        got = hm.score(hf, w, 0.3, xs, ys, batch=1000)
        ref = onp.score(Mf, exp_arr, w, fo, 0.3, xs, ys, batch=1000, threads=8)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        assert np.array_equal(gio.bits(got[2]), gio.bits(ref[2]))
        assert np.array_equal(gio.bits(got[3]), gio.bits(ref[3]))


def random_forest_arrays(F, T, seed, depth=9):
    """Random (untrained) trees in sklearn's array layout: enough to check
    that walk + accumulation agree with the oracle on any forest."""
    rng = np.random.default_rng(seed)
    offs, left, right, feat, thr, miss, p1 = [0], [], [], [], [], [], []
    for _ in range(T):
        l, r, f, t, m, p = [], [], [], [], [], []

        def grow(d):
            i = len(l)
            l.append(-1); r.append(-1); f.append(-2); t.append(-2.0); m.append(0)
            p.append(float(rng.integers(0, 5)) / 4.0)
            if d < depth and (d < 2 or rng.random() < 0.75):
                f[i] = int(rng.integers(0, F)); t[i] = float(rng.random()); m[i] = int(rng.integers(0, 2))
                l[i] = grow(d + 1)
                r[i] = grow(d + 1)
            return i
        grow(0)
        left += l; right += r; feat += f; thr += t; miss += m; p1 += p
        offs.append(len(left))
    return dict(tree_off=np.array(offs, np.int32), left=np.array(left, np.int32),
                right=np.array(right, np.int32), feat=np.array(feat, np.int32),
                thr=np.array(thr, np.float64), miss_left=np.array(miss, np.uint8),
                p1=np.array(p1, np.float64), F=np.int32(F))


def test_empty_and_edge_inputs(hip_lib):
    z = gio.load("g3_score_raw.npz")
    w, upper = int(z["w"]), int(z["upper"])
    Mf = utils.band_filter(gio.sym_matrix(z, "R"), w, upper)
    hm = hip_matrix(Mf, z["exp_arr"], w, upper)
    hf = _lib.HipForest(flat(gio.forest(str(z["forest"]))))
    e = np.zeros(0, np.int32)
    assert hm.score(hf, w, 0.5, e, e)[0].size == 0
    n = Mf.shape[0]
    # windows that leave the matrix are skipped, not errors (scoreUtils.py:75)
    x = np.array([0, 2, n - 30, n - 3], np.int32)
    y = np.array([10, 12, n - 2, n - 1], np.int32)
    f64, _, keep = hm.extract(w, x, y)
    assert keep.size == 0
    # getwindow takes any coordinates (round 3; G8 pins the values): below the diagonal is
    # answered, outside the matrix is dropped, as in the reference
    f64, _, keep = hm.extract(w, np.array([50, 50, -3], np.int32), np.array([40, n, 20], np.int32))
    ref, rkeep = onp.extract(Mf, z["exp_arr"], w, np.array([50, 50, -3]), np.array([40, n, 20]))
    assert np.array_equal(keep, rkeep) and np.array_equal(gio.bits(f64), gio.bits(ref))
    with pytest.raises(_lib.PeakachuHipError, match="IndexError"):
        hm.extract(w, np.array([n - 2], np.int32), np.array([n - 40], np.int32))  # a row beyond the matrix
    # pk_score keeps its contract: candidates are upper-triangle pixels of the matrix
    with pytest.raises(_lib.PeakachuHipError):
        hm.score(hf, w, 0.5, np.array([50], np.int32), np.array([40], np.int32))


def test_score_checks_streamed_coordinates_on_the_device(hip_lib):
    """pk_score with host buffers: from the second call on the coordinates travel chunk by chunk
    behind the kernels and are checked where they arrive (round 4: the host-side pass cost a fifth
    of the call).  A coordinate outside the contract is named, nothing reads out of bounds, and the
    next call is served as if nothing had happened."""
    z = gio.load("g3_score_raw.npz")
    w, upper = int(z["w"]), int(z["upper"])
    Mf = utils.band_filter(gio.sym_matrix(z, "R"), w, upper)
    hm = hip_matrix(Mf, z["exp_arr"], w, upper)
    hf = _lib.HipForest(flat(gio.forest(str(z["forest"]))))
    reps = 700000 // z["ridx"].size + 1   # several upload chunks (the first is 262 144 candidates)
    x = np.tile(z["ridx"].astype(np.int32), reps)
    y = np.tile(z["cidx"].astype(np.int32), reps)
    first = hm.score(hf, w, 0.5, x, y)          # sizes the reusable device-side list (host check)
    again = hm.score(hf, w, 0.5, x, y)          # streamed, device check
    assert all(np.array_equal(gio.bits(a) if a.dtype == np.float64 else a,
                              gio.bits(b) if b.dtype == np.float64 else b) for a, b in zip(first, again))
    for bad_at, (bx, by) in ((400000, (50, 40)), (5, (-1, 3)), (x.size - 1, (10, Mf.shape[0]))):
        xb, yb = x.copy(), y.copy()
        xb[bad_at], yb[bad_at] = bx, by
        with pytest.raises(_lib.PeakachuHipError, match=r"coordinate %d = \(%d, %d\)" % (bad_at, bx, by)):
            hm.score(hf, w, 0.5, xb, yb)
        ok = hm.score(hf, w, 0.5, x, y)
        assert all(np.array_equal(gio.bits(a) if a.dtype == np.float64 else a,
                                  gio.bits(b) if b.dtype == np.float64 else b) for a, b in zip(first, ok))


def test_score_returns_pixels_with_the_count_or_by_fetch(hip_lib):
    """pk_score brings the scored pixels back in the same pinned copy as their count while there are
    at most 8 192 of them (PK_RET_INLINE) and by pk_score_fetch's four copies beyond: both routes
    against pk_score_run + pk_score_fetch on a device-resident list, on either side of the limit and
    for an empty result."""
    z = gio.load("g3_score_raw.npz")
    w, upper = int(z["w"]), int(z["upper"])
    Mf = utils.band_filter(gio.sym_matrix(z, "R"), w, upper)
    hm = hip_matrix(Mf, z["exp_arr"], w, upper)
    hf = _lib.HipForest(flat(gio.forest(str(z["forest"]))))
    reps = 300000 // z["ridx"].size + 1
    x = np.tile(z["ridx"].astype(np.int32), reps)
    y = np.tile(z["cidx"].astype(np.int32), reps)
    cd = _lib.HipCands(x, y)
    cd.run(hm, hf, w, -1.0)            # every window that passes the filters is "scored"
    passing = np.cumsum(cd.fetch_all()[0] != 0)
    assert passing[-1] > 20000
    sizes = []
    for n_cand, thre in ((int(np.searchsorted(passing, 8192)) + 1, -1.0), (int(np.searchsorted(passing, 8193)) + 1, -1.0),
                         (x.size, -1.0), (x.size, 0.5), (x.size, 2.0)):
        xs, ys = x[:n_cand].copy(), y[:n_cand].copy()
        for _ in range(2):             # first call: whole upload; second: streamed
            got = hm.score(hf, w, thre, xs, ys)
        ref_cd = _lib.HipCands(xs, ys)
        ref_cd.run(hm, hf, w, thre)
        want = ref_cd.fetch()
        sizes.append(got[0].size)
        assert got[0].size == want[0].size
        for a, b in zip(got, want):
            assert np.array_equal(gio.bits(a) if a.dtype == np.float64 else a, gio.bits(b) if b.dtype == np.float64 else b)
    assert sizes[0] == 8192 and sizes[1] == 8193 and sizes[2] > 20000 and sizes[4] == 0, sizes


def test_options_belong_to_their_handle(hip_lib):
    """Round 4: every handle carries its own options (rounds 1-3 had one process-wide set, and a
    knob left set by one caller changed every other caller).  Two matrices and two forests with
    different options, used alternately in one process, each take their own route -- and give the
    same bits; a changed process default reaches only handles created afterwards."""
    L = hip_lib
    z = gio.load("g3_score_raw.npz")
    w, upper = int(z["w"]), int(z["upper"])
    Mf = utils.band_filter(gio.sym_matrix(z, "R"), w, upper)
    x, y = z["ridx"].astype(np.int32), z["cidx"].astype(np.int32)
    m_clean = hip_matrix(Mf, z["exp_arr"], w, upper)
    m_general = hip_matrix(Mf, z["exp_arr"], w, upper, options={"extract_clean": 0})
    assert (m_clean.get_option("extract_clean"), m_general.get_option("extract_clean")) == (1, 0)
    feats = []
    for m, want_clean in ((m_clean, True), (m_general, False), (m_clean, True), (m_general, False)):
        before = (L.pk_get_option(b"stat_extract_clean"), L.pk_get_option(b"stat_extract_general"))
        feats.append(m.extract(w, x, y)[0])
        ran_clean = L.pk_get_option(b"stat_extract_clean") > before[0]
        ran_general = L.pk_get_option(b"stat_extract_general") > before[1]
        assert (ran_clean, ran_general) == (want_clean, not want_clean)
    assert all(np.array_equal(gio.bits(feats[0]), gio.bits(f)) for f in feats[1:])
    fo = flat(gio.forest(str(z["forest"])))
    f_rank, f_float = _lib.HipForest(fo), _lib.HipForest(fo, options={"forest_q": 0})
    X = feats[0].astype(np.float32)
    probs = []
    for f, want_rank in ((f_rank, True), (f_float, False), (f_rank, True)):
        L.pk_prof_enable(1)
        L.pk_prof_reset()
        probs.append(f.predict(X))
        assert (_lib.prof_get("quant")[1] > 0) == want_rank   # the rank quantizer ran only for its handle
        L.pk_prof_enable(0)
    assert all(np.array_equal(gio.bits(probs[0]), gio.bits(p)) for p in probs[1:])
    # the process default: for handles made AFTER the change, and only those
    old = L.pk_get_option(b"forest_q")
    try:
        _lib.set_option("forest_q", 0)
        assert f_rank.get_option("forest_q") == 1
        assert _lib.HipForest(fo).get_option("forest_q") == 0
    finally:
        _lib.set_option("forest_q", old)
    assert L.pk_forest_set_option(f_rank.h, b"no_such_option", 1) == _lib.PK_E_INVALID
    assert L.pk_cands_set_option(None, b"chunk", 4096) == _lib.PK_E_INVALID


def test_threads_on_one_device_are_serialised(hip_lib):
    """include/peakachu_hip.h 'threading': the lock is per device -- four threads that extract, predict
    and score on the SAME device at once (own handles, and one shared forest) are serialised by the
    library and all get the bits a single thread gets; the kernel timers keep counting under it."""
    import threading
    L = hip_lib
    z = gio.load("g3_score_raw.npz")
    w, upper = int(z["w"]), int(z["upper"])
    Mf = utils.band_filter(gio.sym_matrix(z, "R"), w, upper)
    x, y = z["ridx"].astype(np.int32), z["cidx"].astype(np.int32)
    fo = flat(gio.forest(str(z["forest"])))
    shared = _lib.HipForest(fo)
    m0 = hip_matrix(Mf, z["exp_arr"], w, upper)
    fea0 = m0.extract(w, x, y)[0]
    p0 = shared.predict(fea0.astype(np.float32))
    s0 = m0.score(shared, w, 0.5, x, y)
    out, errs = {}, []

    def work(k):
        try:
            m = hip_matrix(Mf, z["exp_arr"], w, upper, options={"extract_clean": k & 1})
            f = shared if k & 2 else _lib.HipForest(fo, options={"forest_q": (k >> 1) & 1 ^ 1})
            for _ in range(3):
                fea = m.extract(w, x, y)[0]
                out[k] = (fea, f.predict(fea.astype(np.float32)), m.score(f, w, 0.5, x, y))
        except Exception as e:  # pragma: no cover
            errs.append(e)
    L.pk_prof_enable(1)
    L.pk_prof_reset()
    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    launches = _lib.prof_get("forest")[1]
    L.pk_prof_enable(0)
    assert not errs, errs
    assert launches >= 4 * 3 * 2
    for k in range(4):
        fea, p, sc = out[k]
        assert np.array_equal(gio.bits(fea), gio.bits(fea0))
        assert np.array_equal(gio.bits(p), gio.bits(p0))
        assert all(np.array_equal(gio.bits(np.asarray(a, np.float64)), gio.bits(np.asarray(b, np.float64))) for a, b in zip(sc, s0))


def test_chromosome_drop_in(hip_lib, tmp_path):
    """The mirror class reproduces the reference's bedpe text byte for byte."""
    from peakachu_amd import scoreUtils
    for name in ["g3_score_raw.npz", "g3_score_weights.npz", "g3_score_hicstyle.npz",
                 "g3_score_raw_minprob0.npz"]:
        z = gio.load(name)
        raw = gio.sym_matrix(z, "R")
        mode = str(z["mode"])
        model = flat(gio.forest(str(z["forest"])))
        kw = dict(lower=int(z["lower"]), upper=int(z["upper"]), cname=str(z["cname"]),
                  res=int(z["res"]), width=int(z["w"]))
        if mode == "raw":
            ch = scoreUtils.Chromosome(raw, model, raw_M=raw, weights=None, **kw)
        elif mode == "weights":
            ch = scoreUtils.Chromosome(gio.balance(raw, z["weights"]), model, raw_M=raw,
                                       weights=z["weights"], **kw)
        else:
            ch = scoreUtils.Chromosome(gio.hicstyle(raw, z["weights"]), model, raw_M=raw,
                                       weights=None, **kw)
        assert np.array_equal(ch.ridx, z["ridx"]) and np.array_equal(ch.cidx, z["cidx"])
        result, R = ch.score(thre=float(z["thre"]))
        out = tmp_path / (name + ".bedpe")
        ch.writeBed(str(out), result, R)
        text = out.read_text() if out.exists() else ""
        assert text == str(z["bedpe"])


@pytest.mark.parametrize("tag,wname,chroms", [("weight", "weight", ["#", "X"]),
                                              ("raw", "raw", ["#", "X"]), ("all", "raw", [])])
def test_score_genome_cli_matches_reference(hip_lib, tmp_path, tag, wname, chroms):
    """`peakachu score_genome` drop-in: byte-identical bedpe to the reference's
    own score_genome.main on the same container (golden G6)."""
    import os
    from peakachu_amd import cli
    z = gio.load("g6_driver.npz")
    model = tmp_path / "forest.npz"
    flat(gio.forest(str(z["forest"]))).save(str(model))
    out = tmp_path / (tag + ".bedpe")
    out.write_text("stale\n")  # must be removed first (score_genome.py:11-12)
    argv = ["score_genome", "-p", os.path.join(gio.GOLD, str(z["container"])), "-m", str(model),
            "-O", str(out), "--clr-weight-name", wname, "-u", str(int(z["upper"])), "-C"] + chroms
    cli.run(argv)
    assert out.read_text() == str(z["genome_" + tag])


@pytest.mark.parametrize("uri", ["cool_small.cool", "cool_small.mcool::/resolutions/10000"])
@pytest.mark.parametrize("wname", ["raw", "weight", "KR"])
def test_score_genome_on_a_cool_file(hip_lib, tmp_path, uri, wname):
    """`score_genome -p map.cool` with no cooler / h5py installed (the built-in reader): the
    bedpe equals what the same chromosomes give when the matrices cooler would return
    (tests/golden/cool_small_expected.npz, see tests/test_cool.py) are handed to `Chromosome`
    directly, i.e. peakachu/score_genome.py:53-67 with the file access taken out."""
    import os
    from scipy import sparse
    from peakachu_amd import cli, scoreUtils
    from peakachu_amd.forest import load_model
    z = np.load(os.path.join(gio.GOLD, "cool_small_expected.npz"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = os.path.join(root, "peakachu_amd", "data", "forest_w5_t100.npz")
    out = tmp_path / "cool.bedpe"
    cli.run(["score_genome", "-p", os.path.join(gio.GOLD, uri), "-m", model, "-O", str(out),
             "--clr-weight-name", wname, "-r", "10000", "-l", "6", "-u", "50", "--minimum-prob", "0.05"])
    got = out.read_text()
    ref = tmp_path / "ref.bedpe"
    mdl = load_model(model)

    def csr(name, tag):
        n = int(z[name + "/n"])
        return sparse.csr_matrix((z["%s/%s/data" % (name, tag)], z["%s/%s/indices" % (name, tag)],
                                  z["%s/%s/indptr" % (name, tag)]), shape=(n, n))
    for name in [str(c) for c in z["chromnames"]]:
        raw = csr(name, "raw")
        if wname == "raw":
            X = scoreUtils.Chromosome(raw, model=mdl, raw_M=raw, weights=None, cname=name, lower=6, upper=50,
                                      res=10000, width=5)
        else:
            X = scoreUtils.Chromosome(csr(name, wname), model=mdl, raw_M=raw, weights=z[name + "/" + wname],
                                      cname=name, lower=6, upper=50, res=10000, width=5)
        result, R = X.score(thre=0.05)
        X.writeBed(str(ref), result, R)
    want = ref.read_text() if ref.exists() else ""
    assert got == want
    assert len(got.splitlines()) > 20  # the comparison is not vacuous


def test_score_chromosome_cli_matches_reference(hip_lib, tmp_path):
    import os
    from peakachu_amd import cli
    z = gio.load("g6_driver.npz")
    model = tmp_path / "forest.npz"
    flat(gio.forest(str(z["forest"]))).save(str(model))
    out = tmp_path / "c.bedpe"
    cli.run(["score_chromosome", "-p", os.path.join(gio.GOLD, str(z["container"])), "-m",
             str(model), "-O", str(out), "-C", "3", "-u", str(int(z["upper"]))])
    assert out.read_text() == str(z["chrom_3_weight"])


def test_buildmatrix_drop_in(hip_lib):
    from peakachu_amd import trainUtils
    z = gio.load("g5_buildmatrix.npz")
    M = gio.sym_matrix(z, "M")
    coords = list(zip(z["x"].tolist(), z["y"].tolist()))
    fea = trainUtils.buildmatrix(M, coords, w=int(z["w"]))
    assert np.array_equal(gio.bits(np.asarray(fea)), gio.bits(z["fea"]))
    assert trainUtils.buildmatrix(M, coords[:5], w=int(z["w"])) is None


def test_rccl_gather_single_rank(hip_lib):
    """The RCCL gather code path with a 1-rank communicator (the only shape a
    one-GPU box can run): results come back unchanged and in order."""
    import ctypes as C
    L = hip_lib
    z = gio.load("g3_score_raw_minprob0.npz")
    w, upper = int(z["w"]), int(z["upper"])
    Mf = utils.band_filter(gio.sym_matrix(z, "R"), w, upper)
    hm = hip_matrix(Mf, z["exp_arr"], w, upper)
    hf = _lib.HipForest(flat(gio.forest(str(z["forest"]))))
    cd = _lib.HipCands(z["ridx"], z["cidx"])
    n = cd.run(hm, hf, w, 0.0)
    ox, oy, op, osig = cd.fetch()
    uid = np.zeros(128, np.uint8)
    _lib.check(L.pk_comm_unique_id(uid), "uid")
    comm = L.pk_comm_create(0, 1, 0, uid)
    assert comm, _lib.last_error()
    try:
        counts = np.zeros(1, np.int64)
        gx = np.empty(n, np.int32); gy = np.empty(n, np.int32)
        gp = np.empty(n, np.float64); gs = np.empty(n, np.float64)
        _lib.check(L.pk_comm_gather_scored(comm, cd.h, counts, n, gx.ctypes.data, gy.ctypes.data,
                                           gp.ctypes.data, gs.ctypes.data), "gather")
        assert counts[0] == n and n > 100
        assert np.array_equal(gx, ox) and np.array_equal(gy, oy)
        assert np.array_equal(gio.bits(gp), gio.bits(op)) and np.array_equal(gio.bits(gs), gio.bits(osig))
        payload = np.arange(1000, dtype=np.uint8)
        recv = np.zeros(1000, np.uint8)
        _lib.check(L.pk_comm_gatherv_bytes(comm, payload.ctypes.data, 1000, counts,
                                           recv.ctypes.data, 1000), "gatherv")
        assert counts[0] == 1000 and np.array_equal(recv, payload)
        # the refusal branches (the protocol itself runs with 1..8 ranks and injected failures in
        # tests/test_comm_protocol.py): buffers one record / one byte too small are refused with
        # PK_E_INVALID, nothing is written, and the communicator stays in step -- the next call with
        # the right capacity succeeds
        gx[:] = -7
        assert L.pk_comm_gather_scored(comm, cd.h, counts, n - 1, gx.ctypes.data, gy.ctypes.data,
                                       gp.ctypes.data, gs.ctypes.data) == _lib.PK_E_INVALID
        assert "exceed the root's capacity" in _lib.last_error() and (gx == -7).all()
        assert L.pk_comm_gather_scored(comm, cd.h, counts, n, None, gy.ctypes.data,
                                       gp.ctypes.data, gs.ctypes.data) == _lib.PK_E_INVALID
        _lib.check(L.pk_comm_gather_scored(comm, cd.h, counts, n, gx.ctypes.data, gy.ctypes.data,
                                           gp.ctypes.data, gs.ctypes.data), "gather after a refusal")
        assert np.array_equal(gx, ox) and np.array_equal(gio.bits(gs), gio.bits(osig))
        recv[:] = 0
        assert L.pk_comm_gatherv_bytes(comm, payload.ctypes.data, 1000, counts, recv.ctypes.data, 999) == _lib.PK_E_INVALID
        assert not recv.any()
        _lib.check(L.pk_comm_gatherv_bytes(comm, payload.ctypes.data, 1000, counts,
                                           recv.ctypes.data, 1000), "gatherv after a refusal")
        assert np.array_equal(recv, payload)
    finally:
        L.pk_comm_destroy(comm)


def big_tree_forest(F, seed, n_small=5, big_nodes=60001):
    """A forest whose middle tree has ~60 000 nodes (right offsets beyond the
    13-bit field -> side table; too large for any LDS buffer -> walked from
    global memory) between ordinary small trees, with non-pure leaf values."""
    rng = np.random.default_rng(seed)

    def grow(n_nodes):
        left = [-1]; right = [-1]; feat = [-2]; thr = [-2.0]; p1 = [float(rng.random())]
        frontier = [0]
        while frontier and len(left) + 2 <= n_nodes:
            i = frontier.pop(int(rng.integers(0, len(frontier))))
            feat[i] = int(rng.integers(0, F)); thr[i] = float(rng.random())
            for side in (left, right):
                side[i] = len(left)
                left.append(-1); right.append(-1); feat.append(-2); thr.append(-2.0)
                p1.append(float(rng.random()) if rng.random() < 0.7 else float(rng.integers(0, 2)))
                frontier.append(len(left) - 1)
        return left, right, feat, thr, p1

    offs, cols = [0], {k: [] for k in ("left", "right", "feat", "thr", "miss_left", "p1")}
    sizes = [301] * n_small + [big_nodes] + [301] * n_small
    for n_nodes in sizes:
        l, r, f, t, p = grow(n_nodes)
        cols["left"] += l; cols["right"] += r; cols["feat"] += f; cols["thr"] += t; cols["p1"] += p
        cols["miss_left"] += [int(v) for v in rng.integers(0, 2, len(l))]
        offs.append(len(cols["left"]))
    return dict(tree_off=np.array(offs, np.int32), left=np.array(cols["left"], np.int32),
                right=np.array(cols["right"], np.int32), feat=np.array(cols["feat"], np.int32),
                thr=np.array(cols["thr"], np.float64),
                miss_left=np.array(cols["miss_left"], np.uint8),
                p1=np.array(cols["p1"], np.float64), F=np.int32(F))


@pytest.mark.parametrize("lds,img,q", [(160, 0, 0), (0, 0, 0), (160, 1, 0), (160, 1, 1)])
def test_giant_tree_side_table(hip_lib, lds, img, q):
    F = 121
    fo = big_tree_forest(F, seed=9)
    rng = np.random.default_rng(3)
    X = rng.random((700, F)).astype(np.float32)
    X[5, :] = np.nan
    X[11, rng.integers(0, F, 40)] = np.nan
    ref = onp.predict(fo, X)
    # q=1: the tree exceeds the rank format -> falls back; img=1: the tree does not fit the LDS -> falls back
    hf = _lib.HipForest(flat(fo), options={"forest_q": q, "forest_lds": lds, "forest_img": img})
    info = hf.info()
    p = hf.predict(X)
    assert info["n_nodes"] > 30000
    assert np.array_equal(gio.bits(p), gio.bits(ref))


def test_forest_create_rejects_malformed(hip_lib):
    fo = random_forest_arrays(121, 3, seed=1, depth=4)
    bad = dict(fo); bad["feat"] = fo["feat"].copy()
    bad["feat"][np.flatnonzero(fo["left"] != -1)[0]] = 500   # feature out of range
    with pytest.raises(_lib.PeakachuHipError):
        _lib.HipForest(flat(bad))
    bad = dict(fo); bad["left"] = fo["left"].copy()
    bad["left"][0] = 0                                        # a cycle
    with pytest.raises(_lib.PeakachuHipError):
        _lib.HipForest(flat(bad))
    bad = dict(fo); bad["right"] = fo["right"].copy()
    bad["right"][0] = 10 ** 6                                 # child outside the tree
    with pytest.raises(_lib.PeakachuHipError):
        _lib.HipForest(flat(bad))


@pytest.mark.parametrize("seed", list(range(24)))
def test_randomised_score_vs_oracle(hip_lib, seed):
    """Seeded random configurations: window size (all extractor kernels),
    matrix size / band, balanced values, batch size, threshold, forest shape
    (NaN routing included), a few candidates at the matrix edges."""
    rng = np.random.default_rng(1000 + seed)
    w = int(rng.choice([1, 2, 3, 4, 5, 5, 6, 6, 7, 8, 11, 15]))
    n = int(rng.integers(8 * w + 60, 700))
    band = int(rng.integers(4 * w + 10, min(160, n // 2)))
    upper = int(rng.integers(2 * w + 4, band))
    M, _ = synth.synth_band(n, band, seed=seed, loops=max(2, n // 30))
    if seed % 3 == 1:   # balanced, non-integer values with NaN weights
        wts = synth.synth_weights(n, seed, n_nan=3)
        M = synth.balance(M, wts)
        e = utils.calculate_expected(M, upper + 2 * w, raw=False)
    else:
        e = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, w + 1, upper)
    keep = rng.random(x.size) < 0.5
    x, y = x[keep], y[keep]
    # a few candidates whose window leaves the matrix (skipped, scoreUtils.py:75)
    x = np.r_[x, [0, 1, n - w - 2]].astype(np.int32)
    y = np.r_[y, [w + 2, w + 3, n - 1]].astype(np.int32)
    F = (2 * w + 1) ** 2
    fo = random_forest_arrays(F, int(rng.integers(3, 40)), seed, depth=int(rng.integers(3, 11)))
    fo["miss_left"] = rng.integers(0, 2, fo["miss_left"].size).astype(np.uint8)
    thre = float(rng.choice([0.0, 0.3, 0.5, 0.8]))
    batch = int(rng.choice([1, 2, 97, 4096, 100000]))
    hm = hip_matrix(Mf, e, w, upper)
    hf = _lib.HipForest(flat(fo))
    ox, oy, op, osig = hm.score(hf, w, thre, x, y, batch=batch)
    rx, ry, rp, rs = onp.score(Mf, e, w, fo, thre, x, y, batch=batch, threads=4)
    assert np.array_equal(ox, rx) and np.array_equal(oy, ry), (w, n, band, upper, batch, thre)
    assert np.array_equal(gio.bits(op), gio.bits(rp))
    assert np.array_equal(gio.bits(osig), gio.bits(rs))


@pytest.mark.parametrize("w", [5, 6])
@pytest.mark.parametrize("bad", [0.0, float("nan"), -1.0])
def test_expected_value_beyond_the_band(hip_lib, w, bad):
    """Found by tests/fuzz/fuzz_score.py: the corner cell of a candidate at the largest distance lies
    at |col - row| = upper + 2w, one diagonal beyond the band (scoreUtils.py:30 filters with a
    strict <).  The reference divides that (zero) cell by the expected value all the same:
    0 / 0 = NaN, 0 / -1 = -0.  The pre-divided-band extractor takes a plain +0 there, so a
    matrix whose LAST expected value is not positive and finite must go to the general
    extractor, although no stored diagonal uses that value."""
    n, band, upper = 400, 60, 40
    M, _ = synth.synth_band(n, band, seed=3, loops=6)
    e = utils.calculate_expected(M, upper + 2 * w, raw=True).copy()
    e[-1] = bad
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, upper - 1, upper)   # the two largest distances
    fo = random_forest_arrays((2 * w + 1) ** 2, 9, 5, depth=6)
    hm = hip_matrix(Mf, e, w, upper)
    hf = _lib.HipForest(flat(fo))
    ox, oy, op, osig = hm.score(hf, w, 0.0, x, y)
    rx, ry, rp, rs = onp.score(Mf, e, w, fo, 0.0, x, y)
    assert np.array_equal(ox, rx) and np.array_equal(oy, ry) and rx.size > 100
    assert np.array_equal(gio.bits(op), gio.bits(rp))
    f64, _, keep = hm.extract(w, x, y)
    ref, rkeep = onp.extract(Mf, e, w, x, y)
    assert np.array_equal(keep, rkeep)
    same = (gio.bits(f64) == gio.bits(ref)) | (np.isnan(f64) & np.isnan(ref))
    assert same.all()
    if bad != bad or bad == 0.0:
        assert np.isnan(ref[(y - x)[rkeep] == upper]).all()   # every d = upper window turns NaN


@pytest.mark.parametrize("mode", ["raw", "weights", "separate_raw"])
def test_candidates_on_device_match_scipy(hip_lib, mode):
    """get_candidate on the device (tables made with scipy) against the host
    path that calls scipy.stats.poisson.sf per pixel, on a 6 000-bin map."""
    from peakachu_amd import scoreUtils
    n, band, w, upper = 6000, 200, 5, 200
    raw, _ = synth.synth_band(n, band, seed=11)
    model = flat(gio.forest("g2_forest_plain.npz"))
    if mode == "raw":
        ch = scoreUtils.Chromosome(raw, model, raw_M=raw, weights=None, upper=upper, width=w)
        ref = utils.candidates(raw, ch.background, None, ch.lower, ch.upper)
    elif mode == "weights":
        wts = synth.synth_weights(n, 11, n_nan=7)
        ch = scoreUtils.Chromosome(synth.balance(raw, wts), model, raw_M=raw, weights=wts,
                                   upper=upper, width=w)
        ref = utils.candidates(raw, ch.background, wts, ch.lower, ch.upper)
    else:
        ch = scoreUtils.Chromosome(raw * 0.37, model, raw_M=raw, weights=None, upper=upper, width=w)
        ref = utils.candidates(raw, ch.background, None, ch.lower, ch.upper)
    assert ch._cands is not None, "device path was not taken"
    assert ref[0].size > 1000
    assert np.array_equal(ch.ridx, ref[0]) and np.array_equal(ch.cidx, ref[1])
    # scoring from the device-resident list equals scoring from host coordinates
    res1, _ = ch.score(0.5)
    ch2 = scoreUtils.Chromosome(ch.raw_M if mode != "separate_raw" else raw * 0.37, model,
                                raw_M=raw, weights=ch.weights, upper=upper, width=w) \
        if mode != "weights" else None
    if ch2 is not None:
        ch2.ridx, ch2.cidx = ref[0].copy(), ref[1].copy()   # forces the host-coordinate path
        res2, _ = ch2.score(0.5)
        assert (res1 != res2).nnz == 0


def test_candidates_fall_back_for_non_integer_counts(hip_lib):
    from peakachu_amd import scoreUtils
    raw, _ = synth.synth_band(800, 90, seed=5)
    raw = raw * 1.5
    model = flat(gio.forest("g2_forest_plain.npz"))
    ch = scoreUtils.Chromosome(raw, model, raw_M=raw, weights=None, upper=70, width=5)
    assert ch._cands is None
    ref = utils.candidates(raw, ch.background, None, ch.lower, ch.upper)
    assert np.array_equal(ch.ridx, ref[0]) and np.array_equal(ch.cidx, ref[1])


def test_score_genome_distributed_branch_single_rank(hip_lib, tmp_path):
    """The multi-rank branch of score_genome (chromosome dealing, RCCL transport,
    packed-record gather, ordered merge on rank 0) with a 1-rank group launched
    like the driver launches ranks; output must equal the reference's bedpe."""
    import os, socket, subprocess, sys
    z = gio.load("g6_driver.npz")
    model = tmp_path / "forest.npz"
    flat(gio.forest(str(z["forest"]))).save(str(model))
    out = tmp_path / "dist.bedpe"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "scripts", "peakachu-amd"), "score_genome", "-p",
           os.path.join(gio.GOLD, str(z["container"])), "-m", str(model), "-O", str(out),
           "--clr-weight-name", "weight", "-u", str(int(z["upper"]))]
    env = dict(os.environ, PK_FORCE_DIST="1")
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert out.read_text() == str(z["genome_weight"])


@pytest.mark.parametrize("ranks", [2, 3])
def test_score_genome_fans_out_from_one_command(hip_lib, tmp_path, ranks):
    """The product's own launcher on the GPU (round 4): the score_genome command line, run through
    peakachu_amd.launch.spawn, becomes `ranks` child processes that find each other over the
    standard-library rendezvous, score the chromosomes dealt to them ON THE DEVICE and merge on
    rank 0; the bedpe must equal the reference's.  One GPU here, so all ranks share device 0
    (LOCAL_RANK forced to 0) and the records travel over the rendezvous (PK_TRANSPORT=tcp: RCCL
    refuses two ranks on one GPU); with one GPU per rank the same command uses RCCL."""
    import os, subprocess, sys
    z = gio.load("g6_driver.npz")
    model = tmp_path / "forest.npz"
    flat(gio.forest(str(z["forest"]))).save(str(model))
    out = tmp_path / "fan.bedpe"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv = [os.path.join(root, "scripts", "peakachu-amd"), "score_genome", "-p",
            os.path.join(gio.GOLD, str(z["container"])), "-m", str(model), "-O", str(out),
            "--clr-weight-name", "weight", "-u", str(int(z["upper"]))]
    code = ("import sys; sys.path.insert(0, %r); from peakachu_amd import launch; "
            "sys.exit(launch.spawn(%d, argv=%r, env_extra={'LOCAL_RANK': '0', 'PK_TRANSPORT': 'tcp'}))"
            % (root, ranks, argv))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PK_RDZV_FILE")}
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert out.read_text() == str(z["genome_weight"])


@pytest.mark.parametrize("name", ["g3_score_raw.npz", "g3_score_weights.npz",
                                  "g3_score_hicstyle.npz", "g5_buildmatrix.npz"])
def test_expected_on_device_golden(hip_lib, name):
    """calculate_expected with the diagonal means from the device equals the
    reference's exp_arr bit for bit."""
    z = gio.load(name)
    if name.startswith("g5"):
        M = gio.sym_matrix(z, "M")
        e = utils.calculate_expected(M, int(z["maxdis"]), raw=False, device=0)
    else:
        raw = gio.sym_matrix(z, "R")
        mode = str(z["mode"])
        w = int(z["w"])
        M = raw if mode == "raw" else (gio.balance(raw, z["weights"]) if mode == "weights"
                                       else gio.hicstyle(raw, z["weights"]))
        upper = min(int(z["upper"]), M.shape[0] - 2 * w)
        e = utils.calculate_expected(M, upper + 2 * w, raw=(mode != "weights"), device=0)
    assert np.array_equal(gio.bits(e), gio.bits(z["exp_arr"]))


@pytest.mark.parametrize("n,raw", [(30000, True), (9001, False), (20011, True)])
def test_expected_on_device_matches_numpy_large(hip_lib, n, raw):
    """Diagonals longer than numpy's 8192-element reduction buffer."""
    M, _ = synth.synth_band(n, 150, seed=n)
    if not raw:
        M = synth.balance(M, synth.synth_weights(n, 3, n_nan=9))
    host = utils.calculate_expected(M, 170, raw=raw)
    dev = utils.calculate_expected(M, 170, raw=raw, device=0)
    assert np.array_equal(gio.bits(host), gio.bits(dev))


@pytest.mark.parametrize("w", [5, 6, 7])
def test_extract_unnormalised_when_exp_arr_too_short(hip_lib, w):
    """distance_normaize_core returns the window unnormalised when its largest
    |col-row| is outside exp_arr (peakachu/utils.py:191-192); both extractor
    kernels and the generic one must do the same."""
    n, band, upper = 400, 90, 70
    M, _ = synth.synth_band(n, band, seed=w)
    e_full = utils.calculate_expected(M, upper + 2 * w, raw=True)
    e_short = e_full[:40 + 2 * w].copy()          # candidates with d + 2w >= len are not normalised
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, w + 1, upper)
    sel = (y - x > 30 - 2 * w) & (y - x < 52)     # straddle the boundary
    x, y = x[sel][::3], y[sel][::3]
    Mc = utils.canonical_csr(Mf)
    for pair in (1, 0):
        hm = _lib.HipMatrix(Mc.indptr, Mc.indices, Mc.data, n, e_short, -2 * w + 1,
                            upper + 2 * w - 1, options={"extract_pair": pair})
        f64, _, keep = hm.extract(w, x, y)
        fea, keep_ref = onp.extract(Mf, e_short, w, x, y)
        assert np.array_equal(keep, keep_ref) and keep.size > 50
        assert np.array_equal(gio.bits(f64), gio.bits(fea))
    d = (y - x)[keep]
    assert (d + 2 * w >= e_short.size).any() and (d + 2 * w < e_short.size).any()


@pytest.mark.parametrize("w", [5, 6])
@pytest.mark.parametrize("poison", ["none", "nan", "negative", "negzero", "inf", "huge",
                                    "exp_zero", "exp_nan", "exp_inf", "exp_tiny"])
def test_clean_extractor_and_its_fallback(hip_lib, w, poison):
    """The two-lane extractor reads a pre-divided band and drops NaN handling when the
    matrix qualifies (non-negative finite counts, positive finite expected values).  A
    single offending cell or expected value must send the whole matrix down the general
    kernel; either way the float64 features equal the oracle's bit for bit, and the
    clean kernel equals the general one on clean input."""
    n, band, upper = 500, 80, 60
    M, _ = synth.synth_band(n, band, seed=11 + w)
    e = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, w + 1, upper)
    x, y = x[::2], y[::2]
    Mc = utils.canonical_csr(Mf).copy()
    Mc.data = Mc.data.astype(np.float64)
    e = e.copy()
    k = Mc.data.size // 2
    if poison == "nan":
        Mc.data[k] = np.nan
    elif poison == "negative":
        Mc.data[k] = -3.0
    elif poison == "negzero":
        Mc.data[k] = -0.0
    elif poison == "inf":
        Mc.data[k] = np.inf
    elif poison == "huge":
        Mc.data[k] = 1e200
    elif poison == "exp_zero":
        e[7] = 0.0
    elif poison == "exp_nan":
        e[7] = np.nan
    elif poison == "exp_inf":
        e[7] = np.inf
    elif poison == "exp_tiny":
        e[7] = 1e-200
    got = {}
    L = _lib.load()
    for clean in (1, 0):
        before = (L.pk_get_option(b"stat_extract_clean"), L.pk_get_option(b"stat_extract_general"))
        hm = _lib.HipMatrix(Mc.indptr, Mc.indices, Mc.data, n, e, -2 * w + 1,
                            upper + 2 * w - 1, options={"extract_clean": clean})
        got[clean] = hm.extract(w, x, y)
        ran_clean = L.pk_get_option(b"stat_extract_clean") > before[0]
        ran_general = L.pk_get_option(b"stat_extract_general") > before[1]
        # the clean kernel runs exactly when it is allowed and the matrix is unpoisoned
        # ("exp_tiny" = 1e-200 is a legal expected value but makes the quotients huge)
        assert ran_clean == (clean == 1 and poison == "none"), (poison, clean)
        assert ran_general == (not ran_clean)
    with np.errstate(all="ignore"):
        fea, keep_ref = onp.extract(Mc, e, w, x, y)
    # NaN features (a NaN expected value poisons every window that touches its diagonal):
    # the positions must agree; the sign bit of a NaN is not defined by IEEE 754 and
    # differs between x86 (subsd keeps it) and gfx950 (v_add_f64 with a negated operand
    # flips it), so NaNs are compared as NaNs, everything else bit for bit
    nan_ref = np.isnan(fea)
    for clean in (1, 0):
        f64, _, keep = got[clean]
        assert np.array_equal(keep, keep_ref) and keep.size > 100, (poison, clean)
        assert np.array_equal(np.isnan(f64), nan_ref), (poison, clean)
        assert np.array_equal(gio.bits(f64)[~nan_ref], gio.bits(fea)[~nan_ref]), (poison, clean)
    assert np.array_equal(gio.bits(got[0][0]), gio.bits(got[1][0]))   # general == clean path
    if poison in ("none", "exp_tiny"):
        assert not nan_ref.any()


def test_large_models_stay_on_the_rank_kernels(hip_lib):
    """Round 4: (i) features with 2 048 .. 4 095 distinct thresholds keep ONE rank-tile row (the 12-bit rank
    word), (ii) a tree with more child pairs than the pair field counts is cut into pieces -- both used to
    send a model to the float kernels.  predict_proba through the C ABI against the oracle, NaN features and
    missing_go_to_left nodes included; the plan's read-only options tell which route ran."""
    from test_forest_qimage import _random_forest
    L = hip_lib
    rng = np.random.default_rng(11)
    # (i) 40 features, one of them with ~2 900 thresholds; option 2 = take the 12-bit word whenever it saves rows
    fo = _random_forest(40, 12, 1601, 30, 21)
    fo["F"] = 40
    inner = np.flatnonzero(fo["left"] != -1)
    fo["feat"][inner[rng.random(inner.size) < 0.3]] = 7
    fo["miss_left"][inner] = rng.random(inner.size) < 0.4
    X = rng.random((5000, 40)).astype(np.float32)
    X[:300, 7] = np.float32(fo["thr"][inner][:300])
    X[300:330, 7] = np.nan
    X[330:340] = np.nan
    ref = onp.predict(fo, X)
    for rank12, want_mode, want_rows in ((0, 0, 41), (2, 2, 40)):
        hf = _lib.HipForest(flat(fo), options={"forest_q_rank12": rank12})
        L.pk_prof_enable(1); L.pk_prof_reset()
        p = hf.predict(X)
        assert _lib.prof_get("quant")[1] > 0
        L.pk_prof_enable(0)
        assert (hf.get_option("stat_q_mode"), hf.get_option("stat_q_rows")) == (want_mode, want_rows)
        assert np.array_equal(gio.bits(p), gio.bits(ref))
    # (ii) five trees of ~5 500 child pairs each (the pair field counts 4 096)
    fo = _random_forest(30, 5, 11001, 40, 5)
    fo["F"] = 30
    inner = np.flatnonzero(fo["left"] != -1)
    fo["miss_left"][inner] = rng.random(inner.size) < 0.3
    X = rng.random((3000, 30)).astype(np.float32)
    X[:40, 3] = np.nan
    X[50:250, :] = np.float32(fo["thr"][inner][rng.integers(0, inner.size, (200, 30))])
    hf = _lib.HipForest(flat(fo))
    L.pk_prof_enable(1); L.pk_prof_reset()
    p = hf.predict(X)
    assert _lib.prof_get("quant")[1] > 0                     # the rank kernels ran
    L.pk_prof_enable(0)
    assert hf.get_option("stat_q_trees") > 5
    assert np.array_equal(gio.bits(p), gio.bits(onp.predict(fo, X)))
