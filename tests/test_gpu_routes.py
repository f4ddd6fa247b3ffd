"""Which kernel family walks which model (DESIGN.md "Forest routes"): one model per row of that
table, built here, walked with the library's default options, and the family that ran read back
(read-only option `stat_family`, `stat_q_mode` for the node word of the rank image) -- bit-exact
against the oracle on every route (model.predict_proba, peakachu/scoreUtils.py:109).  A family no
row reaches has no business in the library: forest_pipe_kernel and forest_l2_kernel went that way
in round 5."""
import os

import numpy as np
import pytest

from oracle import oracle_np as onp
from peakachu_amd import _lib
from peakachu_amd.forest import FlatForest
from tools.forest_routes import FAMILY, replicated_forest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
QR, Q, Q2, IMG, LDS, GMEM = 1, 2, 3, 4, 6, 7
NARROW, WIDE, NARROW12 = 0, 1, 2


def committed(name):
    return lambda: FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", name))


def rep(F, T, nodes, depth):
    return lambda: replicated_forest(F, T, nodes, depth, seed=F + T + nodes)


ROUTES = [
    # (what the model is, how it is made, family, node word of the rank image or None)
    ("the benchmark forests' shape, w = 5 (<= 192 features, groups of <= 80 KiB)", committed("forest_w5_t100.npz"), QR, NARROW),
    ("w = 6: the released 5 / 10 kb models' window", committed("forest_w6_t100.npz"), QR, NARROW),
    ("more than 2 047 thresholds per feature (a model fitted on more windows): the 12-bit rank word keeps one row "
     "per feature", rep(121, 300, 2501, 22), QR, NARROW12),
    ("trees of 5 000 .. 8 000 nodes each: a group of them exceeds five 16-KiB rows -- the generic rank kernel",
     rep(121, 100, 8001, 30), Q, NARROW12),
    ("tens of thousands of stumps", rep(121, 20000, 3, 1), QR, NARROW),
    ("trees of 20 000 nodes: two rows per feature, 128-candidate workgroups", rep(121, 60, 20001, 40), Q, NARROW12),
    ("193 .. 255 features (w = 7): 128-candidate workgroups", rep(225, 100, 2501, 22), Q, NARROW),
    ("256 .. 639 rank-tile rows with groups <= 96 KiB: the wide word, two 64-candidate tiles per trip (w = 11, "
     "configs[4])", committed("forest_w11_t500.npz"), Q2, WIDE),
    ("more than 639 rows (w = 13 .. 15): the wide word, one tile per trip", rep(961, 50, 2001, 20), Q, WIDE),
    ("more thresholds than 1 023 rank-tile rows hold, trees that fit the LDS: the 8-byte LDS image", rep(121, 3000, 2501, 22),
     IMG, None),
    ("the same with trees too large for the LDS image: preorder nodes streamed through LDS", rep(121, 300, 40001, 40), LDS, None),
    ("too many thresholds AND a feature tile that leaves the LDS no room for trees (> 255 features): no LDS at all",
     rep(529, 3000, 2501, 22), GMEM, None),
    ("1 024 features (the most the library takes)", rep(1024, 50, 2001, 20), GMEM, None),
]


@pytest.mark.parametrize("what,make,family,word", ROUTES, ids=[r[0][:40] for r in ROUTES])
def test_model_reaches_its_kernel_family(hip_lib, what, make, family, word):
    flat = make()
    hf = _lib.HipForest(flat)
    rng = np.random.default_rng(5)
    X = rng.random((700, flat.F)).astype(np.float32)
    X[3, :] = np.nan
    p = hf.predict(X)
    fod = {k: getattr(flat, k) for k in FlatForest.FIELDS}
    assert np.array_equal(p.view(np.uint64), onp.predict(fod, X).view(np.uint64))
    assert FAMILY[hf.get_option("stat_family")] == FAMILY[family], what
    assert hf.get_option("stat_q_mode") == (-1 if word is None else word), what
    hf.close()


CUT_ROUTES = [r for r in ROUTES if r[2] in (QR, Q, Q2)]


@pytest.mark.parametrize("what,make,family,word", CUT_ROUTES, ids=[r[0][:40] for r in CUT_ROUTES])
def test_rank_kernel_families_cut_in_two(hip_lib, what, make, family, word):
    """The forest cut in two (head over every candidate, the open ones parked, tail over those) on every
    shape of the rank kernels -- forest_qr_kernel, the generic forest_q_kernel with 256 / 128 / 64
    candidates per workgroup, the two-tile forest_q2_kernel: scoring a band's candidates with the cut
    forced in front of groups 1 and 2 (and wherever the library puts it) gives the one-launch run's pixels
    bit for bit (Chromosome.score's body, peakachu/scoreUtils.py:95-125)."""
    import hashlib
    from peakachu_amd import synth, utils
    flat = make()
    w = int(round((flat.F ** 0.5 - 1) / 2))
    assert (2 * w + 1) ** 2 == flat.F
    n, band, upper = 2500, 120, 100
    M, _ = synth.synth_band(n, band, seed=3)
    e = utils.calculate_expected(M, upper + 2 * w, raw=True)
    Mf = utils.band_filter(M, w, upper)
    x, y = synth.all_band_pixels(Mf, w + 1, upper)
    hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, n, e, -2 * w + 1, upper + 2 * w - 1)
    hf = _lib.HipForest(flat, options={"forest_split_min": 1})

    def dig(cd):
        h = hashlib.sha256()
        for a in cd.fetch():
            h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()

    thre = 0.5
    cd = _lib.HipCands(x, y)
    n0 = cd.run(hm, hf, w, thre)
    base = dig(cd)
    st0, pr0 = cd.fetch_all()
    assert FAMILY[hf.get_option("stat_family")] == FAMILY[family], what
    assert hf.get_option("stat_split_group") == 0
    n_grp = hf.get_option("stat_q_groups")
    for at in (1, 2, 0):
        if at >= n_grp:
            continue
        hf.set_option("forest_split_at", at)
        cd2 = _lib.HipCands(x, y)
        cd2.set_prune(True)
        assert cd2.run(hm, hf, w, thre) == n0 and dig(cd2) == base, (what, at)
        if at:
            # (the parked candidates live in the chunk's dead float tiles, 4 F bytes per candidate: a rank tile
            # of two rows per feature -- 20 000-node trees -- does not fit there and the launch stays whole)
            fits = hf.get_option("stat_q_rows") * 2 + 16 < flat.F * 4
            assert hf.get_option("stat_split_group") == (at if fits else 0), (what, at)
        st, pr = cd2.fetch_all()
        same = pr.view(np.uint64) == pr0.view(np.uint64)
        assert np.array_equal(st, st0) and np.all(same | (pr == 0.0)) and np.all(pr0[~same] <= thre), (what, at)
        cd2.close()
    hf.close()
    hm.close()


def test_every_family_is_reached():
    assert {r[2] for r in ROUTES} == set(FAMILY) - {0}
