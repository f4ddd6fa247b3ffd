"""The LDS-staged extractor (csrc/pk_extract.hip: extract_pair_strip_kernel; option extract_strip, default on for
lists of neighbours on clean matrices at w = 5, 6) against the register-gather kernel and the CPU oracle:
Chromosome.getwindow's arithmetic (peakachu/scoreUtils.py:70-93, utils.py:180-237) on lists that are dense,
gappy, shuffled, that change diagonal inside a wave, hug the matrix edges and the band's last diagonal (whose
far window corner lies outside the stored band: scoreUtils.py:30-33), normalised and not.  Bit for bit."""
import os

import numpy as np
import pytest

import golden_io as gio
from oracle import oracle_np as onp
from peakachu_amd import _lib, synth, utils
from peakachu_amd.forest import FlatForest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lists(x, y, rng):
    out = {"all": (x, y), "every3rd": (x[::3].copy(), y[::3].copy()), "every47th": (x[::47].copy(), y[::47].copy())}
    p = rng.permutation(x.size)
    out["shuffled"] = (x[p].copy(), y[p].copy())
    nblk = x.size // 100   # runs of 100 neighbours in random order: the diagonal changes inside a wave
    idx = (rng.permutation(nblk)[:, None] * 100 + np.arange(100)[None, :]).ravel()
    out["runs"] = (x[idx].copy(), y[idx].copy())
    out["tail"] = (x[-77:].copy(), y[-77:].copy())
    out["one"] = (x[1234:1235].copy(), y[1234:1235].copy())
    return out


def _matrix(Mf, e, w, upper, strip):
    Mc = utils.canonical_csr(Mf)
    return _lib.HipMatrix(Mc.indptr, Mc.indices, Mc.data, Mc.shape[0], e, -2 * w + 1, upper + 2 * w - 1,
                          options={"extract_strip": strip})   # 2: every list, 0: never, 1: lists of runs only


@pytest.mark.parametrize("w", [5, 6])
@pytest.mark.parametrize("shape", [(700, 80, 60), (2050, 40, 40), (130, 40, 30)])
def test_staged_features_equal_the_oracles(w, shape):
    """float64 features and survivor lists through pk_extract on every list shape, against the oracle and against
    the register-gather kernel; the counter shows which kernel ran."""
    n, band, upper = shape
    L = _lib.load()
    M, _ = synth.synth_band(n, band, seed=n + w)
    upper = min(upper, n - 2 * w)
    e = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, 0, upper)   # from the main diagonal on: d < 2w, windows off the matrix
    rng = np.random.default_rng(n * 10 + w)
    hs, hr = _matrix(Mf, e, w, upper, 2), _matrix(Mf, e, w, upper, 0)
    try:
        for name, (lx, ly) in _lists(x, y, rng).items():
            before = L.pk_get_option(b"stat_extract_strip")
            f64, _, keep = hs.extract(w, lx, ly)
            assert L.pk_get_option(b"stat_extract_strip") > before, name
            before = L.pk_get_option(b"stat_extract_strip")
            g64, _, keep_r = hr.extract(w, lx, ly)
            assert L.pk_get_option(b"stat_extract_strip") == before
            fea, keep_o = onp.extract(Mf, e, w, lx, ly)
            assert np.array_equal(keep, keep_o) and np.array_equal(keep_r, keep_o), name
            assert np.array_equal(gio.bits(f64), gio.bits(fea)), name
            assert np.array_equal(gio.bits(g64), gio.bits(fea)), name
        assert (y - x == upper).any() and (x < w).any()   # the band's last diagonal and edge windows were there
    finally:
        hs.close(); hr.close()


@pytest.mark.parametrize("w", [5, 6])
def test_staged_windows_that_are_not_normalised(w):
    """distance_normaize_core leaves a window unnormalised when its largest |col-row| is outside exp_arr
    (peakachu/utils.py:191-192): the staged kernel then stages raw counts; candidates on both sides of the
    boundary, neighbours on their diagonals."""
    n, band, upper = 400, 90, 70
    M, _ = synth.synth_band(n, band, seed=w)
    e_short = utils.calculate_expected(M, upper + 2 * w, raw=True)[:40 + 2 * w].copy()
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, w + 1, upper)
    sel = (y - x > 30 - 2 * w) & (y - x < 52)
    x, y = x[sel], y[sel]
    hs = _matrix(Mf, e_short, w, upper, 2)
    try:
        f64, _, keep = hs.extract(w, x, y)
    finally:
        hs.close()
    fea, keep_o = onp.extract(Mf, e_short, w, x, y)
    assert np.array_equal(keep, keep_o) and keep.size > 1000
    assert np.array_equal(gio.bits(f64), gio.bits(fea))
    d = (y - x)[keep]
    assert (d + 2 * w >= e_short.size).any() and (d + 2 * w < e_short.size).any()


@pytest.mark.parametrize("w", [5, 6])
def test_scoring_through_the_staged_extractor(w):
    """pk_score_run with either extractor: status and probability of EVERY candidate and the scored pixels,
    bit for bit the same and the oracle's (several chunks, a batch rule in between)."""
    n, band, upper = 3000, 120, 100
    M, _ = synth.synth_band(n, band, seed=5 + w)
    e = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, 0, upper)
    fo = FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", "forest_w%d_t100.npz" % w))
    hf = _lib.HipForest(fo)
    got = {}
    try:
        for strip in (2, 0):
            hm = _matrix(Mf, e, w, upper, strip)
            cd = _lib.HipCands(x, y, options={"chunk": 65536})
            cd.run(hm, hf, w, 0.4, 10000)
            st, pr = cd.fetch_all()
            got[strip] = (st.copy(), pr.copy(), cd.fetch())
            cd.close(); hm.close()
    finally:
        hf.close()
    assert np.array_equal(got[2][0], got[0][0]) and np.array_equal(gio.bits(got[2][1]), gio.bits(got[0][1]))
    fod = {k: getattr(fo, k) for k in FlatForest.FIELDS}
    rx, ry, rp, rs = onp.score(Mf, e, w, fod, 0.4, x, y, batch=10000, threads=0)
    ox, oy, op, osig = got[2][2]
    assert rx.size > 100 and np.array_equal(ox, rx) and np.array_equal(oy, ry)
    assert np.array_equal(gio.bits(op), gio.bits(rp)) and np.array_equal(gio.bits(osig), gio.bits(rs))


def test_the_library_stages_lists_of_runs_only():
    """Default option (extract_strip = 1): a list whose batches of 32 are runs on one diagonal goes through the
    staged kernel, a strided, a shuffled or a Poisson-thinned one through the register-gather kernel (a wave
    would need a pass per run).  The route only: results are equal either way (tests above)."""
    w, n, band, upper = 5, 3000, 120, 100
    L = _lib.load()
    M, _ = synth.synth_band(n, band, seed=3)
    e = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, w + 1, upper)
    fo = FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", "forest_w5_t100.npz"))
    hf, hm = _lib.HipForest(fo), _matrix(Mf, e, w, upper, 1)
    rng = np.random.default_rng(0)
    keep = rng.random(x.size) < 0.93           # a band with 7 % empty pixels: still runs
    p = rng.permutation(x.size)
    try:
        for name, lx, ly, want in (("all", x, y, True), ("thinned 7 %", x[keep], y[keep], True),
                                   ("every 4th", x[::4], y[::4], False), ("shuffled", x[p], y[p], False),
                                   ("one in 50", x[::50], y[::50], False), ("short", x[:20], y[:20], True)):
            cd = _lib.HipCands(lx.copy(), ly.copy())
            before = L.pk_get_option(b"stat_extract_strip")
            cd.run(hm, hf, w, 0.5)
            assert (L.pk_get_option(b"stat_extract_strip") > before) == want, name
            cd.close()
    finally:
        hm.close(); hf.close()


def test_w11_workgroup_stores_equal_the_one_wave_kernel():
    """w = 11 (23 x 23 windows; four windows per wave): since round 6 four waves form a workgroup and write the
    features of their sixteen candidates together, 64 contiguous bytes per feature (option extract_row16 = 1;
    2 = the one-wave kernel of rounds 3-5).  Status and probability of every candidate, bit for bit, on lists
    whose last workgroup is partly filled and whose waves have windows off the matrix; scored pixels = the oracle's."""
    w, n, band, upper = 11, 900, 90, 60
    M, _ = synth.synth_band(n, band, seed=17)
    e = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, 0, upper)
    fo = FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", "forest_w11_t500.npz"))
    hf = _lib.HipForest(fo)
    Mc = utils.canonical_csr(Mf)
    rng = np.random.default_rng(2)
    p = rng.permutation(x.size)[:5000]
    try:
        for name, lx, ly in (("all", x, y), ("odd tail", x[:1003], y[:1003]), ("seven", x[40:47], y[40:47]),
                             ("shuffled", x[p], y[p])):
            got = {}
            for form in (1, 2):
                hm = _lib.HipMatrix(Mc.indptr, Mc.indices, Mc.data, n, e, -2 * w + 1, upper + 2 * w - 1,
                                    options={"extract_row16": form})
                cd = _lib.HipCands(lx.copy(), ly.copy())
                cd.run(hm, hf, w, 0.5)
                st, pr = cd.fetch_all()
                got[form] = (st.copy(), pr.copy(), cd.fetch())
                cd.close(); hm.close()
            assert np.array_equal(got[1][0], got[2][0]) and np.array_equal(gio.bits(got[1][1]), gio.bits(got[2][1])), name
            if name == "all":
                fod = {k: getattr(fo, k) for k in FlatForest.FIELDS}
                rx, ry, rp, rs = onp.score(Mf, e, w, fod, 0.5, lx, ly, threads=0)
                ox, oy, op, osig = got[1][2]
                assert rx.size > 10 and np.array_equal(ox, rx) and np.array_equal(oy, ry)
                assert np.array_equal(gio.bits(op), gio.bits(rp)) and np.array_equal(gio.bits(osig), gio.bits(rs))
    finally:
        hf.close()


@pytest.mark.parametrize("w", [5, 6])
def test_staged_windows_that_reach_outside_a_narrow_band(w):
    """A matrix handle whose stored band [dlo, dhi] is NARROWER than the windows of its candidates reach on both
    sides (pk_matrix_create takes any band; cells outside it are absent: they read 0, peakachu/scoreUtils.py:81):
    the staged kernel fetches such rows from a neighbour inside the band and zeroes them in LDS -- here for most
    rows of most windows, not only the far corner of the last diagonal."""
    from scipy import sparse
    n, band, upper = 900, 70, 50
    M, _ = synth.synth_band(n, band, seed=40 + w)
    e = utils.calculate_expected(M, upper + 2 * w, raw=True)
    dlo, dhi = 3, upper - 4                      # the main diagonal's neighbourhood and the far band are not stored
    C = sparse.coo_matrix(M)
    keep = (C.col - C.row >= dlo) & (C.col - C.row <= dhi)
    Mn = sparse.csr_matrix((C.data[keep], (C.row[keep], C.col[keep])), shape=M.shape)
    Mc = utils.canonical_csr(Mn)
    x, y = synth.all_band_pixels(Mn, dlo, dhi)
    x, y = np.ascontiguousarray(x, np.int32), np.ascontiguousarray(y, np.int32)
    got = {}
    for strip in (2, 0):
        hm = _lib.HipMatrix(Mc.indptr, Mc.indices, Mc.data, n, e, dlo, dhi, options={"extract_strip": strip})
        try:
            got[strip] = hm.extract(w, x, y)
        finally:
            hm.close()
    fea, keep_o = onp.extract(Mn, e, w, x, y)
    for strip in (2, 0):
        f64, _, keep_g = got[strip]
        assert np.array_equal(keep_g, keep_o) and keep_o.size > 1000, strip
        assert np.array_equal(gio.bits(f64), gio.bits(fea)), strip
