"""BASELINE.json-size runs on the GPU:
  * FULL parity: every one of the 5.56 M candidates of configs[1] (and every 4th of
    the 34.9 M of configs[3]) against the CPU oracle run on all host cores -- scored
    pixels (row, col, probability, signal) and the per-candidate status and
    probability (pk_score_fetch_all), bit for bit;
  * invariance: the result does not depend on chunking, the forest kernel
    variant or how candidates are sharded into blocks (cut at batch multiples);
  * order: outputs are in candidate order; signal equals M[row, col];
  * idempotence: a second run of the same handles gives the same bytes.
Needs an MI355X: -m gpu."""
import hashlib
import os

import numpy as np
import pytest

import golden_io as gio
from oracle import oracle_np as onp
from peakachu_amd import _lib, dist, synth, utils
from peakachu_amd.forest import FlatForest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


class handle_options:
    """`with handle_options(opts, hm, hf):` -- the options that concern each of these (shared)
    handles, for the block; afterwards the handles are back at the process defaults.  No test
    touches a process-wide option (round 4: options are per handle)."""

    def __init__(self, opts, *handles):
        self.opts, self.handles = dict(opts), handles

    def __enter__(self):
        for h in self.handles:
            h.set_options(self.opts)
        return self

    def __exit__(self, *exc):
        L = _lib.load()
        for h in self.handles:
            h.set_options({k: L.pk_get_option(k.encode()) for k in self.opts})


def random_forest(F, T, seed, depth=14, p_split=0.8):
    from test_gpu_parity import random_forest_arrays
    return random_forest_arrays(F, T, seed, depth=depth)


@pytest.fixture(scope="module")
def config2(hip_lib):
    """configs[1]: 30k x 30k, 200-bin band, w=5, the committed 100-tree forest."""
    w, n, band = 5, 30000, 200
    M, _ = synth.synth_band(n, band, seed=0)
    upper = band
    e = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, w + 1, upper)
    fo = FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", "forest_w5_t100.npz"))
    hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, n, e, -2 * w + 1, upper + 2 * w - 1)
    hf = _lib.HipForest(fo)
    return dict(w=w, Mf=Mf, e=e, x=x, y=y, fo=fo, hm=hm, hf=hf)


def test_config2_properties(config2):
    c = config2
    w, x, y = c["w"], c["x"], c["y"]
    assert x.size > 5_000_000
    cd = _lib.HipCands(x, y)
    n1 = cd.run(c["hm"], c["hf"], w, 0.5)
    ox, oy, op, osig = cd.fetch()
    st, pr = cd.fetch_all()
    base = digest(ox, oy, op, osig)
    # idempotence
    assert cd.run(c["hm"], c["hf"], w, 0.5) == n1
    assert digest(*cd.fetch()) == base
    # candidate order + signal
    pos = {}
    key = x.astype(np.int64) * 30000 + y
    idx = np.searchsorted(np.sort(key), ox.astype(np.int64) * 30000 + oy)
    order_in_input = np.argsort(key, kind="stable")[idx]
    assert np.all(np.diff(order_in_input) > 0)
    assert np.array_equal(osig, np.asarray(c["Mf"][ox, oy]).ravel())
    assert np.all(op > 0.5) and np.all(st[order_in_input] == 1)
    assert np.array_equal(gio.bits(pr[order_in_input]), gio.bits(op))
    # invariance to chunking and kernel variant
    if True:
        for opts in (dict(chunk=65536), dict(chunk=1000003), dict(forest_slots=12), dict(forest_q_ch=2),
                     dict(forest_q_ch=2, forest_slots=7), dict(forest_q_persist=0), dict(forest_q_persist=2),
                     dict(forest_q_persist=-7), dict(forest_q_persist=1, chunk=65536),
                     dict(forest_q=0), dict(forest_q=0, forest_slots=5),
                     dict(forest_q=0, forest_img=0),
                     dict(forest_q=0, forest_img=0, forest_slots=4),
                     dict(forest_lds=0), dict(extract_pair=0), dict(extract_strip=0), dict(extract_strip=2, chunk=1000003),
                     dict(overlap=1), dict(overlap=1, chunk=65536),
                     dict(sub_chunk=262144), dict(sub_chunk=100000, chunk=1000000)):
            with handle_options(opts, c["hm"], c["hf"]):
                cd2 = _lib.HipCands(x, y, options=opts)
                assert cd2.run(c["hm"], c["hf"], w, 0.5) == n1
                assert digest(*cd2.fetch()) == base, opts
                st2, pr2 = cd2.fetch_all()
                assert np.array_equal(st2, st) and np.array_equal(gio.bits(pr2), gio.bits(pr))
                cd2.close()
    # invariance to sharding into batch-aligned blocks (the 8-GPU split)
    parts = []
    for lo, hi in dist.block_ranges(x.size, 8, 100000):
        cdr = _lib.HipCands(x[lo:hi], y[lo:hi])
        cdr.run(c["hm"], c["hf"], w, 0.5)
        parts.append(cdr.fetch())
        cdr.close()
    cat = [np.concatenate([p[i] for p in parts]) for i in range(4)]
    assert digest(*cat) == base
    # full parity with the oracle: every candidate's status and probability, every scored pixel
    fod = {k: getattr(c["fo"], k) for k in FlatForest.FIELDS}
    (rx, ry, rp, rs), st_ref, pr_ref = onp.score_all(c["Mf"], c["e"], w, fod, 0.5, x, y)
    assert np.array_equal(ox, rx) and np.array_equal(oy, ry)
    assert np.array_equal(gio.bits(op), gio.bits(rp)) and np.array_equal(gio.bits(osig), gio.bits(rs))
    assert np.array_equal(st != 0, st_ref != 0)
    assert np.array_equal(gio.bits(pr), gio.bits(pr_ref))
    # the headline AS bench.py RUNS IT: the whole list, default options, with the permission Chromosome.score gives
    # (decided candidates may end at probability 0), which cuts the forest in two -- against the oracle's pixels
    cdp = _lib.HipCands(x, y)
    cdp.set_prune(True)
    for _ in range(3):     # (the cut places itself by what the first calls show)
        assert cdp.run(c["hm"], c["hf"], w, 0.5) == n1
    assert c["hf"].get_option("stat_split_group") > 0, "the forest was not cut"
    px, py, pp, ps = cdp.fetch()
    cdp.close()
    assert np.array_equal(px, rx) and np.array_equal(py, ry)
    assert np.array_equal(gio.bits(pp), gio.bits(rp)) and np.array_equal(gio.bits(ps), gio.bits(rs))


def test_config4_full_parity_trained_forest(hip_lib):
    """configs[3]: 5 kb map (60 000 bins, 800-bin band, upper = 800), the TRAINED w=5
    forest, EVERY band pixel (34.9 M candidates; rounds 3-5: every 4th): scored pixels and per-candidate
    status / probability bit-exact against the oracle on all host cores (about half a minute of them)."""
    w, n, band, upper, stride = 5, 60000, 800, 800, 1
    M, _ = synth.synth_band(n, band, seed=4)
    e = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, w + 1, upper)
    x, y = x[::stride].copy(), y[::stride].copy()
    fo = FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", "forest_w5_t100.npz"))
    hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, n, e, -2 * w + 1, upper + 2 * w - 1)
    hf = _lib.HipForest(fo)
    cd = _lib.HipCands(x, y)
    n1 = cd.run(hm, hf, w, 0.5)
    ox, oy, op, osig = cd.fetch()
    st, pr = cd.fetch_all()
    assert x.size > 8_000_000 and n1 > 0
    fod = {k: getattr(fo, k) for k in FlatForest.FIELDS}
    (rx, ry, rp, rs), st_ref, pr_ref = onp.score_all(Mf, e, w, fod, 0.5, x, y)
    assert np.array_equal(ox, rx) and np.array_equal(oy, ry)
    assert np.array_equal(gio.bits(op), gio.bits(rp)) and np.array_equal(gio.bits(osig), gio.bits(rs))
    assert np.array_equal(st != 0, st_ref != 0)
    assert np.array_equal(gio.bits(pr), gio.bits(pr_ref))


def _full_parity(w, n, band, upper, fo, thre=0.5, seed=0, min_cands=0, options=None):
    """The whole candidate list of a bench.py workload on the GPU and through the oracle (all
    host cores): scored pixels, per-candidate status and probability, bit for bit."""
    import bench
    Mf, e, x, y, upper = bench.build_workload(seed, n, band, w, 6, upper)
    hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, n, e, -2 * w + 1, upper + 2 * w - 1, options=options)
    hf = _lib.HipForest(fo, options=options)
    cd = _lib.HipCands(x, y, options=options)
    n1 = cd.run(hm, hf, w, thre)
    ox, oy, op, osig = cd.fetch()
    st, pr = cd.fetch_all()
    assert x.size >= min_cands and n1 > 0
    fod = {k: getattr(fo, k) for k in FlatForest.FIELDS}
    (rx, ry, rp, rs), st_ref, pr_ref = onp.score_all(Mf, e, w, fod, thre, x, y)
    assert rx.size == n1
    assert np.array_equal(ox, rx) and np.array_equal(oy, ry)
    assert np.array_equal(gio.bits(op), gio.bits(rp)) and np.array_equal(gio.bits(osig), gio.bits(rs))
    assert np.array_equal(st != 0, st_ref != 0)
    assert np.array_equal(gio.bits(pr), gio.bits(pr_ref))
    return x.size, n1


def test_w6_full_parity_trained_forest(hip_lib):
    """The width of every released 5 kb / 10 kb model (README.md:142): w = 6, 30 000 bins,
    300-bin band, the TRAINED forest_w6_t100 -- all 7.95 M candidates against the oracle."""
    fo = FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", "forest_w6_t100.npz"))
    n_cand, n_pix = _full_parity(6, 30000, 300, 300, fo, min_cands=7_900_000)
    assert n_pix > 1000


def test_config5_full_parity_fitted_forest(hip_lib):
    """configs[4] on the forest SURVEY 8d names: RandomForestClassifier(500 trees, max_depth 20) FITTED
    on buildmatrix features (peakachu/trainUtils.py:50-57; tools/make_forest.py -w 11 -T 500 ->
    peakachu_amd/data/forest_w11_t500.npz), exactly as bench.py's configs[4] leg runs it: 23 x 23
    windows, 529 features, all 1.42 M candidates against the oracle."""
    fo = FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", "forest_w11_t500.npz"))
    assert fo.T == 500 and fo.F == 529
    n_cand, n_pix = _full_parity(11, 8000, 200, 200, fo, min_cands=1_400_000)
    assert n_pix > 100


def test_config5_full_parity_at_bench_size(hip_lib):
    """The second stress leg of configs[4] (-w 11 --forest random:500:20 --bins 8000): 500 UNTRAINED
    random trees of depth <= 20 (rounds 1-3's stand-in; 70 % of the candidates score), all 1.42 M
    candidates."""
    import bench
    fo = bench.load_forest("random:500:20", 11, 529)
    n_cand, n_pix = _full_parity(11, 8000, 200, 200, fo, min_cands=1_400_000)
    assert n_pix > 100_000
    # the two-tile kernel with every thread staging its share of a group (no helper waves), and the
    # one-tile kernel (what Chromosome.score's pruned runs use), on the same workload
    for name in ("forest_q_help", "forest_q_two"):
        assert _full_parity(11, 8000, 200, 200, fo, min_cands=1_400_000, options={name: 0}) == (n_cand, n_pix)


@pytest.mark.parametrize("w,T,n,band,upper,stride", [(6, 100, 20000, 300, 300, 7),
                                                     (11, 500, 4000, 120, 100, 3),
                                                     (5, 100, 60000, 800, 800, 40)])
def test_other_configs_sampled_parity(hip_lib, w, T, n, band, upper, stride):
    """w=6 (the released models' window), w=11 x 500 trees (configs[4]) and the
    5 kb band (configs[3]: 2x bins, upper = 800) on random forests of matching
    width: every `stride`-th band pixel is scored; a sample is checked
    bit-exactly against the oracle, the rest through kernel-variant invariance."""
    M, _ = synth.synth_band(n, band, seed=w)
    e = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, w + 1, upper)
    x, y = x[::stride], y[::stride]
    F = (2 * w + 1) ** 2
    fo = random_forest(F, T, seed=w)
    ff = FlatForest(F, fo["tree_off"], fo["left"], fo["right"], fo["feat"], fo["thr"],
                    fo["miss_left"], fo["p1"])
    hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, n, e, -2 * w + 1, upper + 2 * w - 1)
    hf = _lib.HipForest(ff)
    cd = _lib.HipCands(x, y)
    thre = 0.45
    n1 = cd.run(hm, hf, w, thre)
    out = cd.fetch()
    st, pr = cd.fetch_all()
    assert n1 > 0 and st.sum() > x.size // 2
    with handle_options({"forest_lds": 0}, hm, hf):
        cd2 = _lib.HipCands(x, y)
        assert cd2.run(hm, hf, w, thre) == n1
        assert digest(*cd2.fetch()) == digest(*out)
    rng = np.random.default_rng(w)
    sel = np.sort(rng.choice(x.size, min(x.size, 3000), replace=False))
    fea, keep = onp.extract(Mf, e, w, x[sel], y[sel])
    p_ref = onp.predict(fo, fea.astype(np.float32))
    assert np.array_equal(np.flatnonzero(st[sel]), keep)
    assert np.array_equal(gio.bits(pr[sel][keep]), gio.bits(p_ref))


def test_host_buffer_call_streams_its_coordinates(config2):
    """pk_score (host coordinate and result buffers) on the whole workload: the coordinates
    travel in growing chunks behind the kernels of the previous chunk (256 Ki, 512 Ki, ... up to
    the chunk size); the scored pixels equal those of the device-resident candidate list, also
    when the call is repeated (reused device buffers) and for a list shorter than one chunk."""
    c = config2
    w = c["w"]
    for sl in (slice(None), slice(0, 3_000_001), slice(5, 200_000), slice(0, 0), slice(7, 8)):
        x, y = c["x"][sl], c["y"][sl]
        cd = _lib.HipCands(x, y)
        cd.run(c["hm"], c["hf"], w, 0.5)
        base = digest(*cd.fetch())
        cd.close()
        for _ in range(2):
            assert digest(*c["hm"].score(c["hf"], w, 0.5, x, y)) == base


@pytest.mark.parametrize("thre", [0.5, 0.2, 0.9, 0.0])
def test_early_exit_same_pixels(config2, thre):
    """Option early_exit: candidates whose sum provably cannot exceed thre*T stop
    walking.  The scored pixels (indices, probabilities, signal) are identical;
    only the diagnostic probability of a pruned candidate reads 0."""
    c = config2
    w = c["w"]
    sl = slice(0, 1_500_000)
    x, y = c["x"][sl], c["y"][sl]
    cd = _lib.HipCands(x, y)
    n1 = cd.run(c["hm"], c["hf"], w, thre)
    base = digest(*cd.fetch())
    st, pr = cd.fetch_all()
    cd2 = _lib.HipCands(x, y, options={"early_exit": 1})
    assert cd2.run(c["hm"], c["hf"], w, thre) == n1
    assert digest(*cd2.fetch()) == base
    st2, pr2 = cd2.fetch_all()
    assert np.array_equal(st2, st)
    same = gio.bits(pr2) == gio.bits(pr)
    assert np.all(same | (pr2 == 0.0))
    assert np.all(pr[~same] <= thre)          # only losers were pruned
    if thre >= 0.2:
        assert (~same).mean() > 0.3           # and pruning actually happened
    else:
        assert np.all(same) or thre > 0.0


@pytest.mark.parametrize("thre,n,prunes", [(0.5, 1_500_000, True), (0.9, 1_500_000, True), (0.9, 200_000, False),
                                           (0.6, 1_500_000, True), (0.2, 1_500_000, False)])
def test_permission_to_stop_early_is_used_where_it_pays(config2, thre, n, prunes):
    """pk_cands_set_prune ALLOWS the early exit (what Chromosome.score sets); the library applies it
    where it pays: launches of at least 2^18 candidates are cut in two at a tree-group boundary (from
    the default threshold 0.5 on: tools/cut_ab.py), shorter ones run whole (profiles/r05_prune_ab.log).
    The scored pixels never depend on it."""
    c = config2
    w = c["w"]
    x, y = c["x"][:n], c["y"][:n]
    cd = _lib.HipCands(x, y)
    n1 = cd.run(c["hm"], c["hf"], w, thre)
    base = digest(*cd.fetch())
    st, pr = cd.fetch_all()
    cd2 = _lib.HipCands(x, y)
    cd2.set_prune(True)
    assert cd2.run(c["hm"], c["hf"], w, thre) == n1
    assert digest(*cd2.fetch()) == base
    st2, pr2 = cd2.fetch_all()
    assert np.array_equal(st2, st)
    same = gio.bits(pr2) == gio.bits(pr)
    assert np.all(same | (pr2 == 0.0)) and np.all(pr[~same] <= thre)
    assert bool((~same).any()) == prunes


@pytest.mark.parametrize("w,nan_diag", [(5, None), (5, 150), (6, None)])
def test_cut_forest_scores_the_same_pixels(config2, w, nan_diag):
    """The forest cut in two (pk_forest_q.hip: head over every candidate, the still-open ones parked with
    their partial sums and rank codes, tail over the parked ones): whatever group it is cut in front of,
    however many launches a call takes and whether candidates carry NaN features (status 2: the walk
    with missing_go_to_left), the scored pixels are the one-launch kernel's bit for bit, a candidate's
    probability is either that kernel's or -- decided at the cut -- 0, and only losers are decided."""
    c = config2
    M = c["Mf"]
    e = c["e"] if w == 5 else utils.calculate_expected(M, 200 + 2 * w, raw=True)
    if nan_diag is not None:   # a NaN expected value poisons every window that touches its diagonal
        e = e.copy()
        e[nan_diag] = np.nan
    fo = c["fo"] if w == 5 else FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", "forest_w%d_t100.npz" % w))
    hm = _lib.HipMatrix(M.indptr, M.indices, M.data, M.shape[0], e, -2 * w + 1, 200 + 2 * w - 1)
    hf = _lib.HipForest(fo)
    n = 1_300_003
    x, y = c["x"][-n:], c["y"][-n:]          # (the far diagonals: the poisoned one is among them)
    thre = 0.5
    cd = _lib.HipCands(x, y, options={"chunk": 500_000})
    n1 = cd.run(hm, hf, w, thre)
    base = digest(*cd.fetch())
    st, pr = cd.fetch_all()
    assert hf.get_option("stat_family") == 1 and hf.get_option("stat_split_group") == 0
    if nan_diag is not None:
        assert (st == 2).sum() > 10_000
    n_grp = hf.get_option("stat_q_groups")
    assert n_grp >= 8
    hf.set_option("forest_split_min", 1)
    parked = {}
    late = n_grp * 3 // 4          # (three quarters of the trees walked: most candidates are decided at 0.5)
    for at in (0, 1, late, n_grp - 1):
        hf.set_option("forest_split_at", at)
        cd2 = _lib.HipCands(x, y, options={"chunk": 500_000})
        cd2.set_prune(True)
        assert cd2.run(hm, hf, w, thre) == n1
        assert digest(*cd2.fetch()) == base
        st2, pr2 = cd2.fetch_all()
        assert np.array_equal(st2, st)
        same = gio.bits(pr2) == gio.bits(pr)
        assert np.all(same | (pr2 == 0.0)) and np.all(pr[~same] <= thre)
        g = hf.get_option("stat_split_group")
        assert g == at if at else 0 < g < n_grp
        parked[at] = hf.get_option("stat_split_parked")
        # every candidate that kept its probability and could still win was parked (+ the unfinished blocks)
        assert parked[at] >= int((pr > thre).sum())
        assert hf.get_option("stat_split_trees") in range(1, 100)
        cd2.close()
    assert parked[1] > parked[late] >= parked[n_grp - 1]   # the later the cut, the fewer are open
    assert parked[1] >= int((st != 0).sum())                       # (after 8 of 100 trees nobody is decided at 0.5)
    # the float tiles made piece by piece (sub_chunk), or two tile buffers with the extractor of the next chunk
    # beside the forest (overlap: no cut then -- the parked candidates live in the chunk's dead float tiles)
    hf.set_option("forest_split_at", 0)
    for extra, cut_expected in (({"sub_chunk": 131072}, True), ({"overlap": 1}, False)):
        cd4 = _lib.HipCands(x, y, options=dict({"chunk": 500_000}, **extra))
        cd4.set_prune(True)
        assert cd4.run(hm, hf, w, thre) == n1 and digest(*cd4.fetch()) == base
        assert (hf.get_option("stat_split_group") > 0) == cut_expected, extra
        cd4.close()
    # the permission withdrawn: one launch, every probability
    hf.set_option("forest_split_at", 0)
    cd3 = _lib.HipCands(x, y, options={"chunk": 500_000})
    assert cd3.run(hm, hf, w, thre) == n1 and hf.get_option("stat_split_group") == 0
    assert np.array_equal(gio.bits(cd3.fetch_all()[1]), gio.bits(pr))


def test_cut_forest_gives_up_where_nobody_is_decided(config2):
    """A forest of untrained random trees (p ~ 0.5 everywhere) parks every candidate wherever it is cut:
    the library learns that from the calls themselves (pk_forest_cut_feedback) -- the cut moves later,
    then is given up for this threshold -- and the scored pixels never change."""
    c = config2
    w = c["w"]
    import bench
    fo = bench.load_forest("random:100:14", w, 121)
    hf = _lib.HipForest(fo, options={"forest_split_min": 1})
    x, y = c["x"][:700_000], c["y"][:700_000]
    cd0 = _lib.HipCands(x, y)
    n0 = cd0.run(c["hm"], hf, w, 0.5)
    base = digest(*cd0.fetch())
    if hf.get_option("stat_family") != 1:
        pytest.skip("the random forest did not take the default rank kernel")
    cd = _lib.HipCands(x, y)
    cd.set_prune(True)
    cuts = []
    for _ in range(8):
        assert cd.run(c["hm"], hf, w, 0.5) == n0 and digest(*cd.fetch()) == base
        cuts.append(hf.get_option("stat_split_group"))
    assert cuts[0] > 0 and cuts[-1] == 0 and hf.get_option("stat_split_shift") == -1
    assert all(b >= a or b == 0 for a, b in zip(cuts, cuts[1:]))   # later and later, then not at all
    # another threshold starts afresh
    assert cd.run(c["hm"], hf, w, 0.9) >= 0 and hf.get_option("stat_split_group") > 0


def test_short_lists_take_one_launch_behind_the_forest(config2):
    """Lists of up to 2^14 candidates: batch rule, p > thre, ordered compaction and the reply of the call
    in ONE single-workgroup launch (compact_small_kernel) -- the same pixels as the four kernels of long
    lists, for lengths around a 64-candidate group and the 2^14 limit, batches that cut groups in two and
    thresholds that keep everything or almost nothing; through pk_score_run and through pk_score (whose
    pixels come back inside the reply)."""
    c = config2
    w = c["w"]
    rng = np.random.default_rng(9)
    for n in (1, 2, 63, 64, 65, 1000, 4097, 16383, 16384, 16385):
        sel = np.sort(rng.choice(c["x"].size, n, replace=False))
        x, y = c["x"][sel], c["y"][sel]
        for batch in (64, 97, 5000, 100000):
            for thre in (0.0, 0.02, 0.5):
                got = {}
                for small in (1, 0):
                    cd = _lib.HipCands(x, y, options={"compact_small": small})
                    k = cd.run(c["hm"], c["hf"], w, thre, batch)
                    got[small] = (k, digest(*cd.fetch()))
                    cd.close()
                assert got[1] == got[0], (n, batch, thre)
                with handle_options({"compact_small": 1}, c["hm"]):
                    a = digest(*c["hm"].score(c["hf"], w, thre, x, y, batch=batch))
                with handle_options({"compact_small": 0}, c["hm"]):
                    b = digest(*c["hm"].score(c["hf"], w, thre, x, y, batch=batch))
                assert a == b == got[0][1], (n, batch, thre)


@pytest.mark.parametrize("w", [5, 6])
def test_scattered_lists_load_their_windows_diagonal_by_diagonal(config2, w):
    """A list whose consecutive candidates are not neighbours (get_candidate's lists: one band pixel in tens)
    is extracted with every lane's loads in the order of its window's diagonals (option extract_diag;
    neighbours on a diagonal share a line of the band) -- the order of LOADS only: status, probability of
    every candidate and the scored pixels are those of the column order, bit for bit, and the oracle's."""
    c = config2
    M = c["Mf"]
    e = c["e"] if w == 5 else utils.calculate_expected(M, 200 + 2 * w, raw=True)
    fo = c["fo"] if w == 5 else FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", "forest_w%d_t100.npz" % w))
    hf = _lib.HipForest(fo)
    x, y = c["x"][3::41].copy(), c["y"][3::41].copy()          # ~135 000 scattered candidates
    got = {}
    for diag in (0, 1, 2):
        hm = _lib.HipMatrix(M.indptr, M.indices, M.data, M.shape[0], e, -2 * w + 1, 200 + 2 * w - 1,
                            options={"extract_diag": diag})
        cd = _lib.HipCands(x, y)
        cd.run(hm, hf, w, 0.3)
        got[diag] = digest(*cd.fetch(), *cd.fetch_all())
        if diag == 1:
            ox, oy, op, osig = cd.fetch()
        cd.close()
        hm.close()
    assert got[0] == got[1] == got[2]
    fod = {k: getattr(fo, k) for k in FlatForest.FIELDS}
    rx, ry, rp, rs = onp.score(M, e, w, fod, 0.3, x, y, threads=0)
    assert np.array_equal(ox, rx) and np.array_equal(oy, ry)
    assert np.array_equal(gio.bits(op), gio.bits(rp)) and np.array_equal(gio.bits(osig), gio.bits(rs))


def test_extract_and_predict_across_chunks(config2):
    """pk_extract (65 536-candidate staging chunks) and pk_predict (512 k chunks)
    on inputs larger than one chunk: survivor order and sampled values vs the oracle."""
    c = config2
    w = c["w"]
    x, y = c["x"][:200_000], c["y"][:200_000]
    f64, f32, keep = c["hm"].extract(w, x, y, want64=True, want32=True)
    assert np.all(np.diff(keep) > 0) and keep.size > 190_000
    rng = np.random.default_rng(5)
    sel = np.sort(rng.choice(keep.size, 2000, replace=False))
    fea_ref, keep_ref = onp.extract(c["Mf"], c["e"], w, x[keep[sel]], y[keep[sel]])
    assert keep_ref.size == sel.size
    assert np.array_equal(gio.bits(f64[sel]), gio.bits(fea_ref))
    assert np.array_equal(f32[sel], fea_ref.astype(np.float32))
    # predict on > 524 288 rows: tile the extracted features three times
    X = np.concatenate([f32, f32, f32[:200_000]])
    p = c["hf"].predict(X)
    fod = {k: getattr(c["fo"], k) for k in FlatForest.FIELDS}
    sel2 = np.sort(rng.choice(X.shape[0], 3000, replace=False))
    assert np.array_equal(gio.bits(p[sel2]), gio.bits(onp.predict(fod, X[sel2])))
    assert np.array_equal(gio.bits(p[:keep.size]), gio.bits(p[keep.size:2 * keep.size]))


def test_clean_and_general_extractor_agree_on_config2(config2):
    """Config 2 qualifies for the pre-divided band; switching the shortcut off must not
    change a single reported pixel, probability or survivor flag."""
    c = config2
    out = {}
    for clean in (1, 0):
        M = c["Mf"]
        hm = _lib.HipMatrix(M.indptr, M.indices, M.data, M.shape[0], c["e"], -2 * c["w"] + 1,
                            200 + 2 * c["w"] - 1, options={"extract_clean": clean})
        cd = _lib.HipCands(c["x"], c["y"])
        cd.run(hm, c["hf"], c["w"], 0.5)
        out[clean] = digest(*cd.fetch(), *cd.fetch_all())
    assert out[0] == out[1]


@pytest.mark.parametrize("w", [5, 6])
def test_clean_extractor_bitwise_equals_general_on_balanced_values(hip_lib, config2, w):
    """The clean kernel's shortcuts (pre-divided band, shared reciprocal of the min-max
    division, no NaN handling) against the general kernel's per-cell IEEE divisions:
    float64 features of ~600 k windows of a BALANCED (non-integer, ~1e-3) matrix, bit for
    bit -- 10^8 quotients that must round identically."""
    c = config2
    M = c["Mf"].tocoo()
    rng = np.random.default_rng(77 + w)
    wts = 1.0 / np.sqrt(200.0 * rng.uniform(0.7, 1.3, M.shape[0]))
    from scipy import sparse
    B = sparse.csr_matrix((M.data * wts[M.row] * wts[M.col], (M.row, M.col)), shape=M.shape)
    B = utils.canonical_csr(B)
    e = c["e"] * float(np.mean(wts)) ** 2
    sel = np.sort(rng.choice(c["x"].size, 600_000, replace=False))
    x, y = c["x"][sel], c["y"][sel]
    keep_band = (y - x >= 2 * w) & (y - x <= 200 - 2 * w)
    x, y = x[keep_band], y[keep_band]
    L = _lib.load()
    got = {}
    for clean in (1, 0):
        before = L.pk_get_option(b"stat_extract_clean")
        hm = _lib.HipMatrix(B.indptr, B.indices, B.data, B.shape[0], e, -2 * w + 1, 200 + 2 * w - 1,
                            options={"extract_clean": clean})
        parts = []
        for s0 in range(0, x.size, 150_000):          # bounded host memory
            f64, _, keep = hm.extract(w, x[s0:s0 + 150_000], y[s0:s0 + 150_000])
            parts.append((hashlib.sha256(f64.tobytes()).hexdigest(), keep.size,
                          hashlib.sha256(keep.tobytes()).hexdigest()))
        got[clean] = parts
        assert (L.pk_get_option(b"stat_extract_clean") > before) == (clean == 1)
    assert got[0] == got[1]
    assert sum(p[1] for p in got[1]) > 300_000


@pytest.mark.skipif(os.environ.get("PK_TEST_HUGE") != "1",
                    reason="opt-in (PK_TEST_HUGE=1): 2.15e9 candidates, ~90 GB of HBM and ~20 GB of host memory")
def test_more_candidates_than_int32_holds(config2):
    """Maximum sizes: a candidate list of more than 2^31 entries (the configs[1] list 387 times over:
    2 150 856 603 candidates, 88 GB of device arrays) through pk_score_run.  The list is periodic, so
    is the result: every period scores the same 7 115 pixels with the same bits (no batch of 100 000
    has fewer than two survivors here, so the batch rule never bites), in candidate order; the
    per-candidate status / probability of the LAST period equal the first period's.  Run by hand
    (profiles/r04_huge_list.log); indices, block counts and offsets beyond int32 are what it checks."""
    c = config2
    w, x, y = c["w"], c["x"], c["y"]
    reps = (1 << 31) // x.size + 1
    N = reps * x.size
    assert N > (1 << 31)
    one = _lib.HipCands(x, y)
    n1 = one.run(c["hm"], c["hf"], w, 0.5)
    ox1, oy1, op1, os1 = one.fetch()
    st1, pr1 = one.fetch_all()
    del one
    X, Y = np.tile(x, reps), np.tile(y, reps)
    # pk_score with host buffers: the first call uploads the list whole, the second streams it chunk by
    # chunk (1 027 launches) -- offsets beyond int32 in the upload schedule as well
    for _ in range(2):
        sx, sy, sp, ss = c["hm"].score(c["hf"], w, 0.5, X, Y)
        assert sx.size == reps * n1
        assert (sx.reshape(reps, n1) == ox1).all() and (sy.reshape(reps, n1) == oy1).all()
        assert (gio.bits(sp).reshape(reps, n1) == gio.bits(op1)).all() and (gio.bits(ss).reshape(reps, n1) == gio.bits(os1)).all()
    del sx, sy, sp, ss
    c["hm"].__dict__.pop("_score_out", None)   # (the wrapper's N-sized result buffers)
    cd = _lib.HipCands(X, Y)
    del X, Y
    n = cd.run(c["hm"], c["hf"], w, 0.5)
    assert n == reps * n1
    ox, oy, op, osig = cd.fetch()
    for got, want in ((ox, ox1), (oy, oy1), (op, op1), (osig, os1)):
        got = got.reshape(reps, n1)
        assert np.array_equal(gio.bits(got[0]) if got.dtype == np.float64 else got[0],
                              gio.bits(want) if want.dtype == np.float64 else want)
        assert (got == got[0]).all()
    st, pr = cd.fetch_all()
    assert np.array_equal(st[-x.size:], st1) and np.array_equal(gio.bits(pr[-x.size:]), gio.bits(pr1))
    assert np.array_equal(st[:x.size], st1) and np.array_equal(gio.bits(pr[:x.size]), gio.bits(pr1))


def _wide_band_csr(n, B, seed):
    """Symmetric band matrix with every cell |col - row| <= B stored (small integer counts, a third
    of them explicit zeros), built row block by row block: 536 M stored cells at n = 120 000, B = 2 226."""
    from scipy import sparse
    rng = np.random.default_rng(seed)
    V = rng.integers(0, 6, size=(B + 1, n), dtype=np.uint8)   # V[d, k] = M[k, k + d]
    V[V == 5] = 0
    V[V == 4] = 0
    rows = np.arange(n, dtype=np.int64)
    lo, hi = np.maximum(rows - B, 0), np.minimum(rows + B, n - 1)
    indptr = np.concatenate([[0], np.cumsum(hi - lo + 1)])
    assert indptr[-1] < 2 ** 31
    indices = np.empty(indptr[-1], np.int32)
    data = np.empty(indptr[-1], np.float64)
    off = np.arange(-B, B + 1, dtype=np.int64)
    for r0 in range(0, n, 512):
        r = rows[r0:r0 + 512]
        cols = r[:, None] + off[None, :]
        ok = (cols >= 0) & (cols < n)
        d = np.abs(off)[None, :].repeat(r.size, 0)[ok]
        cc = cols[ok]
        rr = np.broadcast_to(r[:, None], cols.shape)[ok]
        sl = slice(indptr[r0], indptr[min(r0 + 512, n)])
        indices[sl] = cc
        data[sl] = V[d, np.minimum(rr, cc)]
    return sparse.csr_matrix((data, indices, indptr.astype(np.int32)), shape=(n, n))


@pytest.mark.skipif(os.environ.get("PK_TEST_NO_BIG") == "1",
                    reason="PK_TEST_NO_BIG=1: skips the 4.3 GB band (~12 GB of host memory, ~10 s)")
def test_bands_on_either_side_of_the_32_bit_offsets(hip_lib):
    """Maximum sizes: the clean extractor addresses the raw band and the quotient band behind it with
    32-bit byte offsets (csrc/pk_api.hip: both must end below 4 GiB - 4 KiB).  One matrix, two bands:
    2 236 diagonals x 120 000 columns x 8 B x 2 = 4 293 120 000 B (the clean kernel, offsets up to the
    limit) and 2 237 diagonals (over: the general kernel with 64-bit addresses) -- 1.5 M candidates
    spread over all diagonals incl. the last one and the last rows, against the oracle."""
    # (B = the first band's last diagonal: the matrix holds nothing beyond what Chromosome's band filter,
    # scoreUtils.py:30-33, would keep, so the oracle and the device read the same cells)
    n, B, w = 120_000, 2_217 + 2 * 5 - 1, 5
    M = _wide_band_csr(n, B, 11)
    fo = FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", "forest_w5_t100.npz"))
    hf = _lib.HipForest(fo)
    rng = np.random.default_rng(5)
    L = hip_lib
    for upper, want_clean in ((2_217, True), (2_218, False)):
        assert ((upper + 4 * w - 1) * n * 16 < (1 << 32) - 4096) == want_clean
        e = 3.0 / np.sqrt(1.0 + np.arange(upper + 2 * w + 1))
        k = 1_500_000
        d = rng.integers(6, upper + 1, k)
        d[:2000] = upper                      # the last diagonal of the band
        x = rng.integers(0, n, k)
        x[2000:4000] = n - 1 - w - d[2000:4000]   # windows that end in the last rows / columns
        x = np.minimum(x, n - 1 - d)
        y = x + d
        x, y = x.astype(np.int32), y.astype(np.int32)
        hm = _lib.HipMatrix(M.indptr, M.indices, M.data, n, e, -2 * w + 1, upper + 2 * w - 1)
        before = L.pk_get_option(b"stat_extract_clean"), L.pk_get_option(b"stat_extract_general")
        cd = _lib.HipCands(x, y)
        cd.run(hm, hf, w, 0.5)
        after = L.pk_get_option(b"stat_extract_clean"), L.pk_get_option(b"stat_extract_general")
        assert (after[0] > before[0], after[1] > before[1]) == (want_clean, not want_clean)
        got, (st, pr) = cd.fetch(), cd.fetch_all()
        want, st_ref, pr_ref = onp.score_all(M, e, w, {f: getattr(fo, f) for f in FlatForest.FIELDS}, 0.5, x, y)
        # every candidate's status and probability (most windows of this noise pass the filters and
        # score low), and the few scored pixels
        assert np.array_equal(st != 0, st_ref != 0) and (st != 0).mean() > 0.3
        assert np.array_equal(gio.bits(pr), gio.bits(pr_ref))
        assert got[0].size == want[0].size
        for a, b in zip(got, want):
            a, b = np.asarray(a), np.asarray(b)
            assert np.array_equal(gio.bits(a.astype(np.float64)) if a.dtype.kind == "f" else a.astype(np.int64),
                                  gio.bits(b.astype(np.float64)) if b.dtype.kind == "f" else b.astype(np.int64))
        del cd
        del hm


@pytest.mark.skipif(os.environ.get("PK_TEST_NO_BIG") == "1",
                    reason="PK_TEST_NO_BIG=1: skips the matrices of more than a million bins (~5 s)")
def test_more_bins_than_the_clean_extractors_24_bit_products_hold(hip_lib):
    """Maximum sizes: the clean extractor multiplies (column offset) x (8 x leading dimension) with
    v_mad_i32_i24 and is only used for ld < 2^20 (csrc/pk_extract.hip: `m->ld < (1 << 20)`).  A thin
    band over 1 048 000 bins (ld = 1 048 000: the clean kernel, products up to 2^23 - 4 608) and over
    1 100 000 bins (the general kernel), 2 M candidates each incl. the last rows, against the oracle."""
    w, upper = 5, 20
    fo = FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", "forest_w5_t100.npz"))
    hf = _lib.HipForest(fo)
    rng = np.random.default_rng(9)
    L = hip_lib
    e = 3.0 / np.sqrt(1.0 + np.arange(upper + 2 * w + 1))
    for n, want_clean in ((1_048_000, True), (1_100_000, False)):
        assert (((n + 63) // 64 * 64) < (1 << 20)) == want_clean
        M = _wide_band_csr(n, upper + 2 * w - 1, 3)
        k = 2_000_000
        d = rng.integers(6, upper + 1, k)
        x = rng.integers(0, n, k)
        x[:5000] = n - 1 - w - d[:5000] - rng.integers(0, 3, 5000)   # windows in the last rows / columns
        x = np.clip(x, 0, n - 1 - d)
        x, y = x.astype(np.int32), (x + d).astype(np.int32)
        hm = _lib.HipMatrix(M.indptr, M.indices, M.data, n, e, -2 * w + 1, upper + 2 * w - 1)
        before = L.pk_get_option(b"stat_extract_clean"), L.pk_get_option(b"stat_extract_general")
        cd = _lib.HipCands(x, y)
        cd.run(hm, hf, w, 0.5)
        after = L.pk_get_option(b"stat_extract_clean"), L.pk_get_option(b"stat_extract_general")
        assert (after[0] > before[0], after[1] > before[1]) == (want_clean, not want_clean)
        got, (st, pr) = cd.fetch(), cd.fetch_all()
        want, st_ref, pr_ref = onp.score_all(M, e, w, {f: getattr(fo, f) for f in FlatForest.FIELDS}, 0.5, x, y)
        assert np.array_equal(st != 0, st_ref != 0) and (st != 0).mean() > 0.3
        assert np.array_equal(gio.bits(pr), gio.bits(pr_ref))
        assert got[0].size == want[0].size
        for a, b in zip(got, want):
            a, b = np.asarray(a), np.asarray(b)
            assert np.array_equal(gio.bits(a.astype(np.float64)) if a.dtype.kind == "f" else a.astype(np.int64),
                                  gio.bits(b.astype(np.float64)) if b.dtype.kind == "f" else b.astype(np.int64))
        del cd, hm, M
