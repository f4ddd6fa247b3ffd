"""The forest's LDS image (peakachu_amd/csrc/pk_image.hip), checked on the CPU.

The image is what forest_img_kernel copies into LDS and walks
(model.predict_proba at peakachu/scoreUtils.py:109).  Here a numpy model of the
160 KiB LDS receives each group's image exactly as the kernel stages it
(16-byte units into the two regions the feature tile leaves free) and walks
it with the kernel's rules: feature address = half tile + feature*256 +
lane*4, child pair at (word.y & 0x3fff8), `x <= thr` picks the left word (NaN
goes left only where bit 0 says so), a fixed number of levels per tree, value
read through the final word.  The result must equal the oracle's
predict_proba bit for bit, i.e. the golden sklearn outputs.
No GPU needed: pk_debug_forest_image is host code.
"""
import ctypes as C
import os

import numpy as np
import pytest

import golden_io as gio
from oracle import oracle_np as onp
from peakachu_amd import _lib
from peakachu_amd.forest import FlatForest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LDS_BYTES = 163840


def build_image(fo, F, slots):
    L = _lib.load()
    T = int(len(fo["tree_off"]) - 1)
    lay = np.zeros(8, np.int32)
    cap_words = 4 * int(fo["tree_off"][-1]) + 64 * T + 4096
    words = np.zeros(cap_words, np.uint64)
    nw = C.c_int64()
    gtab = np.zeros(4 * (T + 4), np.int32)
    ng = C.c_int32()
    troot = np.zeros(T, np.uint64)
    tdepth = np.zeros(T, np.int32)
    rc = L.pk_debug_forest_image(
        T, F, np.ascontiguousarray(fo["tree_off"], np.int32),
        np.ascontiguousarray(fo["left"], np.int32), np.ascontiguousarray(fo["right"], np.int32),
        np.ascontiguousarray(fo["feat"], np.int32), np.ascontiguousarray(fo["thr"], np.float64),
        np.ascontiguousarray(fo["miss_left"], np.uint8), np.ascontiguousarray(fo["p1"], np.float64),
        slots, lay, cap_words, words, C.byref(nw), T + 4, gtab, C.byref(ng), troot, tdepth)
    if rc != 0:
        return rc, _lib.last_error()
    return 0, dict(lay=lay, words=words[:nw.value], gtab=gtab[:4 * (ng.value + 2)].reshape(-1, 4),
                   n_grp=ng.value, troot=troot, tdepth=tdepth)


def walk_image(img, X, T):
    """predict_proba[:,1] of float32 rows X by walking the image like the kernel does."""
    HB, lenA, B0, val_off, dec_off, cap, slots, F = [int(v) for v in img["lay"]]
    N = X.shape[0]
    acc = np.zeros(N, np.float64)
    with np.errstate(invalid="ignore"):
        for g in range(img["n_grp"]):
            t0, nt, off, nu = [int(v) for v in img["gtab"][g]]
            assert 0 < nt <= slots and nu * 16 <= cap
            lds = np.zeros(LDS_BYTES // 8, np.uint64)
            # the tile regions hold features, not words: poison them so a stray read shows
            lds[: HB // 8] = np.uint64(0xDEADBEEFDEADBEEF)
            lds[65536 // 8: (65536 + HB) // 8] = np.uint64(0xDEADBEEFDEADBEEF)
            w = img["words"][2 * off: 2 * (off + nu)]
            vo = np.arange(w.size) * 8
            phys = np.where(vo < lenA, HB + vo, B0 + (vo - lenA))
            assert phys.max() + 8 <= LDS_BYTES
            lds[phys // 8] = w
            for t in range(t0, t0 + nt):
                cur = np.full(N, img["troot"][t], np.uint64)
                for _ in range(int(img["tdepth"][t])):
                    thr = (cur & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.float32)
                    pk = (cur >> np.uint64(32)).astype(np.uint32)
                    f = (pk >> 24).astype(np.int64)
                    assert f.max() < F
                    x = X[np.arange(N), f]
                    ca = (pk & np.uint32(0x3FFF8)).astype(np.int64)
                    # a pair never touches a feature tile
                    assert ((ca >= HB) & (ca + 16 <= 65536) | (ca >= 65536 + HB) & (ca + 16 <= LDS_BYTES)).all()
                    lw, rw = lds[ca // 8], lds[ca // 8 + 1]
                    gl = (x <= thr) | (np.isnan(x) & ((pk & 1) != 0))
                    cur = np.where(gl, lw, rw)
                va = (cur & np.uint64(0x3FFFF)).astype(np.int64)
                assert (va % 8 == 0).all()
                acc += lds[va // 8].view(np.float64)  # tree order: sklearn's sequential sum
    return acc / float(T)


@pytest.mark.parametrize("slots", [2, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("tag", ["plain", "balanced", "subsample"])
def test_image_walk_equals_sklearn_golden(tag, slots):
    z = gio.load("g2_forest_%s.npz" % tag)
    X = np.ascontiguousarray(gio.load("g2_forest_plain.npz")["X"], np.float32)
    assert np.isnan(X).any()
    fo = gio.forest(z)
    F = X.shape[1]
    rc, img = build_image(fo, F, slots)
    assert rc == 0, img
    T = len(fo["tree_off"]) - 1
    p = walk_image(img, X, T)
    assert np.array_equal(gio.bits(p), gio.bits(z["p"]))


@pytest.mark.parametrize("name,slots", [("forest_w5_t100.npz", 7), ("forest_w5_t100.npz", 8),
                                        ("forest_w6_t100.npz", 6)])
def test_image_of_benchmark_forests(name, slots):
    ff = FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", name))
    fo = {k: getattr(ff, k) for k in FlatForest.FIELDS}
    rc, img = build_image(fo, ff.F, slots)
    assert rc == 0, img
    rng = np.random.default_rng(5)
    X = rng.random((600, ff.F)).astype(np.float32)
    X[::7] = (X[::7] > 0.5).astype(np.float32)      # exact 0 / 1 features (min-max scaling makes them)
    X[5, 3] = np.nan
    X[11, :] = np.nan
    p = walk_image(img, X, ff.T)
    ref = onp.predict(fo, X)
    assert np.array_equal(gio.bits(p), gio.bits(ref))
    # groups are consecutive, cover every tree once, and are well filled
    gt = img["gtab"][: img["n_grp"]]
    assert gt[0, 0] == 0 and (gt[1:, 0] == gt[:-1, 0] + gt[:-1, 1]).all()
    assert gt[-1, 0] + gt[-1, 1] == ff.T
    assert (img["tdepth"] == 20).all()


def test_image_degenerate_trees():
    """One-leaf trees, a stump with two pure leaves, equal-valued pure siblings, stored leaves."""
    # tree 0: single leaf 0.25; tree 1: stump -> (0.0, 1.0); tree 2: stump -> (1.0, 1.0);
    # tree 3: depth 2, left child interior with (0.5, 0.0), right child pure 0.0; tree 4: single leaf 1.0
    left = [-1, 1, -1, -1, 1, -1, -1, 1, 3, -1, -1, -1, -1]
    right = [-1, 2, -1, -1, 2, -1, -1, 2, 4, -1, -1, -1, -1]
    feat = [0, 2, 0, 0, 1, 0, 0, 0, 2, 0, 0, 0, 0]
    thr = [0, 0.5, 0, 0, 0.25, 0, 0, 0.75, 0.125, 0, 0, 0, 0]
    p1 = [0.25, 0, 0.0, 1.0, 0, 1.0, 1.0, 0, 0, 0.5, 0.0, 0.0, 1.0]
    # node layout: t0 = [0]; t1 = [1,2,3]; t2 = [4,5,6]; t3 = [7..11]; t4 = [12]
    tree_off = [0, 1, 4, 7, 12, 13]
    # children are relative to the tree's first node
    left = np.array(left, np.int32)
    right = np.array(right, np.int32)
    rel = np.zeros(13, np.int32)
    for t in range(5):
        rel[tree_off[t]:tree_off[t + 1]] = tree_off[t]
    lrel = np.where(left >= 0, left, -1)
    rrel = np.where(right >= 0, right, -1)
    # the literals above are already tree-relative except tree 3 (written relative too)
    fo = dict(tree_off=np.array(tree_off, np.int32), left=lrel, right=rrel,
              feat=np.array(feat, np.int32), thr=np.array(thr, np.float64),
              miss_left=np.zeros(13, np.uint8), p1=np.array(p1, np.float64))
    fo["miss_left"][7] = 1
    F = 3
    rng = np.random.default_rng(2)
    X = rng.random((200, F)).astype(np.float32)
    X[3, 0] = np.nan
    X[4, 2] = np.nan
    ref = onp.predict(fo, X)
    for slots in (2, 4, 8):
        rc, img = build_image(fo, F, slots)
        assert rc == 0, img
        assert list(img["tdepth"]) == [0, 1, 1, 2, 0]
        p = walk_image(img, X, 5)
        assert np.array_equal(gio.bits(p), gio.bits(ref))


def test_image_rejects_bad_forests():
    fo = dict(tree_off=np.array([0, 3], np.int32), left=np.array([1, 0, -1], np.int32),
              right=np.array([2, 2, -1], np.int32), feat=np.zeros(3, np.int32),
              thr=np.zeros(3), miss_left=np.zeros(3, np.uint8), p1=np.zeros(3))
    rc, msg = build_image(fo, 4, 4)     # node 1 points back at the root
    assert rc != 0 and "malformed" in msg
    fo["left"] = np.array([1, -1, -1], np.int32)
    fo["feat"] = np.array([9, 0, 0], np.int32)
    rc, msg = build_image(fo, 4, 4)     # feature index out of range
    assert rc != 0
