"""Host-side logic against the reference's golden outputs (CPU only):
expected-by-distance, band filter, candidate selection, chromosome queue,
sharding helpers, the map container and bedpe formatting."""
import numpy as np
import pytest

import golden_io as gio
from peakachu_amd import cli, dist, io, score_genome, scoreUtils, utils


def _mode_inputs(z):
    raw = gio.sym_matrix(z, "R")
    mode = str(z["mode"])
    if mode == "raw":
        return raw, raw, None, True
    if mode == "weights":
        return gio.balance(raw, z["weights"]), raw, z["weights"], False
    return gio.hicstyle(raw, z["weights"]), raw, None, True


@pytest.mark.parametrize("name", ["g3_score_raw.npz", "g3_score_weights.npz",
                                  "g3_score_hicstyle.npz"])
def test_expected_bandfilter_candidates(name):
    z = gio.load(name)
    w = int(z["w"])
    M, raw, wts, rawmode = _mode_inputs(z)
    lower = max(int(z["lower"]), w + 1)
    upper = min(int(z["upper"]), M.shape[0] - 2 * w)
    e = utils.calculate_expected(M, upper + 2 * w, raw=rawmode)
    assert np.array_equal(gio.bits(e), gio.bits(z["exp_arr"]))
    if str(z["mode"]) == "hicstyle":
        bg = utils.calculate_expected(raw, upper + 2 * w, raw=True)
    else:
        bg = e
    assert np.array_equal(gio.bits(bg), gio.bits(z["background"]))
    Mf = utils.band_filter(M, w, upper)
    assert gio.digest(Mf) == str(z["Mf_sha"])
    x, y = utils.candidates(raw, bg, wts, lower, upper)
    assert np.array_equal(x, z["ridx"]) and np.array_equal(y, z["cidx"])


def test_buildmatrix_expected():
    z = gio.load("g5_buildmatrix.npz")
    M = gio.sym_matrix(z, "M")
    e = utils.calculate_expected(M, int(z["maxdis"]))
    assert np.array_equal(gio.bits(e), gio.bits(z["exp_arr"]))


def test_select_chromosomes():
    names = ["chr1", "chr2", "3", "chrX", "chrM", "chrY", "chr10_random"]
    assert score_genome.select_chromosomes(names, ["#", "X"]) == ["chr1", "chr2", "3", "chrX"]
    assert score_genome.select_chromosomes(names, []) == names
    assert score_genome.select_chromosomes(names, ["M", "Y"]) == ["chrM", "chrY"]


def test_lpt_and_blocks():
    w = [50, 10, 40, 30, 20, 60, 5]
    own = dist.lpt_assign(w, 3)
    assert sorted(sum(own, [])) == list(range(len(w)))
    loads = [sum(w[i] for i in o) for o in own]
    assert max(loads) - min(loads) <= max(w)
    assert dist.lpt_assign(w, 3) == own  # deterministic
    assert dist.lpt_assign(w, 1) == [list(range(len(w)))]
    for N, R, B in ((100001, 2, 100000), (5557769, 8, 100000), (10, 4, 100000), (0, 3, 7)):
        rr = dist.block_ranges(N, R, B)
        assert rr[0][0] == 0 and rr[-1][1] == N
        for (a, b), (c, d) in zip(rr, rr[1:]):
            assert b == c and a <= b
        assert all(a % B == 0 for a, _ in rr)


def test_pkmap_roundtrip(tmp_path):
    from peakachu_amd import synth
    M, _ = synth.synth_band(200, 40, seed=5)
    wts = synth.synth_weights(200, 5, n_nan=2)
    path = str(tmp_path / "m.pkmap.npz")
    io.write_pkmap(path, {"chr1": (M, wts), "chr2": (M, None)}, resolution=5000)
    lib = io.open_map(path)
    assert lib.chromnames == ["chr1", "chr2"] and lib.binsize == 5000
    raw = utils.tocsr(lib.matrix(balance=False, sparse=True).fetch("chr1"))
    assert (raw != M).nnz == 0
    bal = utils.tocsr(lib.matrix(balance="weight", sparse=True).fetch("chr1"))
    ref = gio.balance(M, wts)
    assert np.array_equal(gio.bits(bal.data), gio.bits(ref.data))
    assert np.array_equal(gio.bits(lib.bins().fetch("chr1")["weight"].values), gio.bits(wts))
    with pytest.raises(KeyError):
        lib.bins().fetch("chr2")


def test_write_bedpe_format(tmp_path):
    from scipy import sparse
    r = np.array([3, 3, 1]); c = np.array([9, 4, 7])
    p = np.array([0.5781894544004801, 1.0, 1e-05]); s = np.array([6.0, 0.25, 3.0])
    P = sparse.csr_matrix((p, (r, c)), shape=(12, 12))
    S = sparse.csr_matrix((s, (r, c)), shape=(12, 12))
    out = tmp_path / "o.bedpe"
    scoreUtils.write_bedpe(str(out), "chr9", 10000, P, S)
    scoreUtils.write_bedpe(str(out), "chr9", 10000, P, S)  # append mode
    lines = out.read_text().splitlines()
    assert lines[0] == "chr9\t10000\t20000\tchr9\t70000\t80000\t1e-05\t3.0"
    assert lines[1] == "chr9\t30000\t40000\tchr9\t40000\t50000\t1.0\t0.25"
    assert lines[2] == "chr9\t30000\t40000\tchr9\t90000\t100000\t0.5781894544004801\t6.0"
    assert len(lines) == 6


def test_cli_defaults_match_reference():
    args, _ = cli.getargs(["score_genome", "-p", "x.npz", "-m", "m.npz", "-O", "o"])
    assert (args.resolution, args.clr_weight_name, args.chroms, args.lower, args.upper,
            args.minimum_prob) == (10000, "weight", ["#", "X"], 6, 300, 0.5)
    args, _ = cli.getargs(["score_chromosome", "-C", "chr21", "--clr-weight-name", "raw",
                           "-u", "400", "--minimum-prob", "0.1"])
    assert (args.chrom, args.clr_weight_name, args.upper, args.minimum_prob) == \
        ("chr21", "raw", 400, 0.1)
    args, _ = cli.getargs(["score_genome", "-C"])
    assert args.chroms == []


def test_numpy_sum_model():
    """The summation order pk_expected_means reproduces on the device: numpy adds
    8192-element buffers one after the other, each summed pairwise (128-element
    blocks, 8 accumulators, split at n/2 rounded down to a multiple of 8)."""
    def pairwise(a, lo, n):
        if n < 8:
            r = 0.0
            for i in range(n):
                r = r + a[lo + i]
            return r
        if n <= 128:
            r = [a[lo + j] for j in range(8)]
            i = 8
            while i < n - (n % 8):
                for j in range(8):
                    r[j] = r[j] + a[lo + i + j]
                i += 8
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
            while i < n:
                res = res + a[lo + i]
                i += 1
            return res
        h = n // 2
        h -= h % 8
        return pairwise(a, lo, h) + pairwise(a, lo + h, n - h)

    def model(a):
        acc, lo = None, 0
        while lo < len(a):
            m = min(8192, len(a) - lo)
            p = pairwise(a, lo, m)
            acc = p if acc is None else acc + p
            lo += m
        return acc

    assert np.getbufsize() == 8192
    rng = np.random.default_rng(3)
    for n in [1, 7, 8, 9, 127, 128, 129, 1000, 8191, 8192, 8193, 16385, 29990]:
        a = rng.random(n) * rng.choice([1.0, 1e4])
        a[rng.random(n) < 0.2] = 0
        for arr in (a, a[rng.random(n) < 0.9]):
            if len(arr):
                assert float(arr.mean()) == model(arr) / len(arr)


def test_isotonic_fit_without_sklearn():
    """utils.isotonic_expected_restated (used when scikit-learn is not installed): the expected
    curve's non-increasing fit as scikit-learn <= 1.3 computes it -- the reference's pin.
    (a) Its pooling equals scikit-learn's own Cython routine, which this installation still
    ships although its IsotonicRegression now delegates to scipy >= 1.12 (another summation
    order: the two differ by ulps, which is why utils.isotonic_expected calls whatever
    scikit-learn is installed, like the reference); (b) the whole function equals what
    scikit-learn 0.24.2's IsotonicRegression returned for 120 curves
    (tools/make_isotonic_fixture.py, the image's Anaconda interpreter)."""
    from peakachu_amd import utils
    try:
        from sklearn._isotonic import _inplace_contiguous_isotonic_regression as cy_pava
    except ImportError:
        cy_pava = None
    rng = np.random.default_rng(8)
    if cy_pava is not None:
        for k in range(300):
            n = int(rng.integers(1, 300))
            y = np.round(rng.random(n) * (20 if k % 2 else 1e6)) / (20 if k % 2 else 1e6) if k % 3 else rng.random(n)
            a = utils._pava_increasing(y.copy())
            b = y.copy()
            cy_pava(b, np.ones(n))
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    z = gio.load("isotonic_sk0242.npz")
    assert str(z["version"]) == "0.24.2"
    off = np.concatenate([[0], np.cumsum(z["n"])])
    for k in range(z["n"].size):
        e, want = z["x"][off[k]:off[k + 1]], z["y"][off[k]:off[k + 1]]
        got = utils.isotonic_expected_restated(e.copy())
        assert got.dtype == np.float64 and np.array_equal(got.view(np.uint64), want.view(np.uint64)), k
    with pytest.raises(ValueError):
        utils.isotonic_expected_restated(np.zeros(5))


def test_host_side_runs_without_sklearn_joblib_cooler_h5py(tmp_path):
    """The host side of the scoring path in an interpreter where scikit-learn, joblib, cooler
    and h5py cannot be imported: the expected curve, a pickled model (written by scikit-learn
    0.24.2) and a .cool contact map are all handled with numpy / scipy alone."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys
        class Block:
            def find_spec(self, name, path=None, target=None):
                if name.split(".")[0] in ("sklearn", "joblib", "cooler", "h5py"):
                    raise ImportError("blocked for this test: " + name)
        sys.meta_path.insert(0, Block())
        import numpy as np
        sys.path.insert(0, %r)
        from peakachu_amd import io, utils, synth
        from peakachu_amd.forest import load_model
        M, _ = synth.synth_band(300, 60, seed=1)
        e = utils.calculate_expected(M, 50, raw=True)
        assert e.shape == (51,) and np.all(np.diff(e) <= 0) and np.all(e > 0)
        ff = load_model(%r)
        assert ff.T == 12 and ff.F == 121
        lib = io.open_map(%r)
        assert type(lib).__name__ == "CoolFile" and lib.chromnames == ["chr1", "chr2", "chrX"]
        assert lib.matrix(balance="weight", sparse=True).fetch("chr2").shape == (200, 200)
        assert not any(m.split(".")[0] in ("sklearn", "joblib", "cooler", "h5py") for m in sys.modules)
        print("ok")
    """) % (root, os.path.join(gio.GOLD, "old_sklearn_rf_plain.xz.joblib"), os.path.join(gio.GOLD, "cool_small.cool"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_count_thresholds_in_one_call_equal_one_call_per_diagonal():
    """utils._poisson_count_thresholds evaluates scipy's sf for every diagonal's counts in one
    call; the answers are those of a call per diagonal (peakachu/scoreUtils.py:59-60,67: the
    decision is scipy's)."""
    from scipy import stats
    from peakachu_amd import utils
    rng = np.random.default_rng(11)
    mu = np.concatenate([200 / (1 + np.arange(320)) ** 0.9 + 0.3, rng.uniform(0, 5, 150), 10 ** rng.uniform(-12, 4, 150),
                         [0, -1, np.nan, np.inf, 1e-300, 3e5]])
    want = np.full(mu.size, np.iinfo(np.int64).max, np.int64)
    for i in np.flatnonzero(np.isfinite(mu) & (mu > 0)):
        hi = int(mu[i] + 10.0 * np.sqrt(mu[i]) + 30)
        while hi <= 1 << 24:
            ks = np.arange(1, hi + 1, dtype=np.float64)
            with np.errstate(all="ignore"):
                hit = np.flatnonzero(stats.poisson.sf(ks, mu[i]) < 0.01)
            if hit.size:
                want[i] = int(ks[hit[0]])
                break
            hi *= 2
    assert np.array_equal(utils._poisson_count_thresholds(mu), want)
