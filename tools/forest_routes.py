#!/usr/bin/env python
"""GPU-box helper: which kernel family walks which model (read-only option stat_family) --
the table of DESIGN.md "Forest routes" is this script's output; tests/test_gpu_routes.py
asserts its rows."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
from peakachu_amd import _lib
from peakachu_amd.forest import FlatForest
from oracle import oracle_np as onp
from fuzz_forest import random_forest

FAMILY = {0: "-", 1: "forest_qr_kernel", 2: "forest_q_kernel", 3: "forest_q2_kernel", 4: "forest_img_kernel",
          6: "forest_lds_kernel", 7: "forest_gmem_kernel"}


def replicated_forest(F, T, nodes, depth, seed):
    """T trees of one random shape (nodes, depth) with their own random features, thresholds and
    leaf values: what decides a model's route -- features, trees, nodes per tree, distinct
    thresholds per feature -- without a Python loop per node."""
    rng = np.random.default_rng(seed)
    one = random_forest(rng, F, 1, nodes, depth, 0.0)
    n = one["left"].size
    inner = np.tile(one["left"] >= 0, T)
    fo = {k: np.tile(one[k], T) for k in ("left", "right", "miss_left")}
    feat, thr, p1 = np.tile(one["feat"], T), np.tile(one["thr"], T), np.tile(one["p1"], T)
    k = int(inner.sum())
    feat[inner] = rng.integers(0, F, k)
    thr[inner] = rng.random(k)
    p1[~inner] = rng.integers(0, 2, inner.size - k)
    fo.update(feat=feat.astype(np.int32), thr=thr, p1=p1, tree_off=(np.arange(T + 1) * n).astype(np.int32))
    return FlatForest(F, *[fo[k] for k in FlatForest.FIELDS])


def committed(name):
    return FlatForest.load(os.path.join(ROOT, "peakachu_amd", "data", name))


def models():
    rng = np.random.default_rng(7)
    yield "benchmark w=5 (121 features, 100 x 2 483 nodes)", committed("forest_w5_t100.npz")
    yield "benchmark w=6 (169 features, 100 trees)", committed("forest_w6_t100.npz")
    yield "benchmark w=11 (529 features, 500 x 1 719 nodes)", committed("forest_w11_t500.npz")
    for label, F, T, nodes, depth in [
            ("121 features, 100 x 8 000 nodes (a model fitted on 139 000 windows)", 121, 100, 8001, 30),
            ("121 features, 60 x 20 001 nodes", 121, 60, 20001, 40),
            ("225 features (w=7), 100 x 2 501 nodes", 225, 100, 2501, 22),
            ("255 features, 100 x 2 501 nodes", 255, 100, 2501, 22),
            ("289 features (w=8), 100 x 2 501 nodes", 289, 100, 2501, 22),
            ("529 features, 100 x 6 001 nodes", 529, 100, 6001, 26),
            ("639 features, 50 x 2 001 nodes", 639, 50, 2001, 20),
            ("700 features, 50 x 2 001 nodes", 700, 50, 2001, 20),
            ("961 features (w=15), 50 x 2 001 nodes", 961, 50, 2001, 20),
            ("1 024 features, 50 x 2 001 nodes", 1024, 50, 2001, 20),
            ("121 features, 3 000 x 2 501 nodes", 121, 3000, 2501, 22),
            ("121 features, 20 000 stumps", 121, 20000, 3, 1),
            ("900 features, 1 500 x 301 nodes", 900, 1500, 301, 12),
            ("121 features, 5 x 31 nodes + one tree of 60 001 nodes", 121, 6, -1, 0)]:
        if nodes == -1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from test_gpu_parity import big_tree_forest
            fo = big_tree_forest(121, 3)
        else:
            fo = random_forest(rng, F, T, nodes, depth, 0.0)
        yield label, FlatForest(F, *[fo[k] for k in FlatForest.FIELDS])


def main():
    _lib.require_device()
    rng = np.random.default_rng(1)
    for label, fo in models():
        flat = fo
        try:
            hf = _lib.HipForest(flat)
        except _lib.PeakachuHipError as e:
            print("%-70s refused: %s" % (label, str(e)[:60]))
            continue
        X = rng.random((1000, flat.F)).astype(np.float32)
        p = hf.predict(X)
        fod = {k: getattr(flat, k) for k in FlatForest.FIELDS}
        ok = np.array_equal(p.view(np.uint64), onp.predict(fod, X).view(np.uint64))
        fam = hf.get_option("stat_family")
        print("%-70s %-20s q_mode %2d rows %4d shape %2d trees %5d  %s" % (
            label, FAMILY.get(fam, fam), hf.get_option("stat_q_mode"), hf.get_option("stat_q_rows"),
            hf.get_option("stat_q_shape"), hf.get_option("stat_q_trees"), "bit-exact" if ok else "MISMATCH"))
        hf.close()


if __name__ == "__main__":
    main()
