#!/usr/bin/env python
"""Train the synthetic-workload forest committed as
peakachu_amd/data/forest_w{w}_t{T}.npz.

No pre-trained Peakachu model exists offline, so the benchmark forest is
fitted here the way `peakachu train` fits one (peakachu/trainUtils.py:46-63:
RandomForestClassifier, 100 trees, max_depth 20, max_features='sqrt',
n_jobs=1) on features of a seeded synthetic band matrix: planted loop pixels
(and their 8 neighbours) against random band pixels, with 10 % label noise so
that the trees reach the size of trees grown on real, overlapping classes
(thousands of nodes, depth 20).  Features come from the reference's own
trainUtils.buildmatrix (numba.njit as identity, see tools/make_golden.py).
Deterministic: fixed seeds, n_jobs=1.
"""
import argparse
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_nb = types.ModuleType("numba")
_nb.njit = lambda f=None, *a, **k: f if callable(f) else (lambda g: g)
sys.modules["numba"] = _nb
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

from sklearn.ensemble import RandomForestClassifier  # noqa: E402

from peakachu import trainUtils  # noqa: E402
from peakachu_amd import synth  # noqa: E402
from peakachu_amd.forest import FlatForest  # noqa: E402

np.seterr(divide="ignore", invalid="ignore")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-w", type=int, default=5)
    ap.add_argument("-T", type=int, default=100)
    ap.add_argument("--n", type=int, default=12000)
    ap.add_argument("--loops", type=int, default=2500)
    ap.add_argument("--neg", type=int, default=22000)
    ap.add_argument("--noise", type=float, default=0.10)
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("-o", default=None)
    a = ap.parse_args()
    w = a.w
    t0 = time.time()
    M, loops = synth.synth_band(a.n, 200, seed=1000 + a.seed, loops=a.loops)
    rng = np.random.default_rng(a.seed)
    pos = set()
    for x, y in loops:
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                pos.add((int(x + dx), int(y + dy)))
    pos = sorted(pos)
    nx = rng.integers(w, a.n - 200 - w, size=a.neg)
    nd = rng.integers(w + 1, 200, size=a.neg)
    neg = [(int(p), int(p + q)) for p, q in zip(nx, nd) if (int(p), int(p + q)) not in set(pos)]
    fp = trainUtils.buildmatrix(M, pos, w=w)
    fn = trainUtils.buildmatrix(M, neg, w=w)
    X = np.r_[fp, fn]
    y = np.r_[np.ones(len(fp)), np.zeros(len(fn))]
    flip = rng.random(y.size) < a.noise
    y = np.where(flip, 1 - y, y)
    print("features: %d pos %d neg in %.1fs" % (len(fp), len(fn), time.time() - t0))
    t0 = time.time()
    rf = RandomForestClassifier(n_estimators=a.T, max_depth=a.depth, max_features="sqrt",
                                n_jobs=1, random_state=a.seed)
    rf.fit(X, y)
    ff = FlatForest.from_sklearn(rf)
    depth = [e.tree_.max_depth for e in rf.estimators_]
    print("fit %.1fs: %s, depth max %d mean %.1f" % (time.time() - t0, ff.stats(), max(depth),
                                                     np.mean(depth)))
    out = a.o or os.path.join(ROOT, "peakachu_amd", "data", "forest_w%d_t%d.npz" % (w, a.T))
    ff.save(out)
    print(out, "%.1f KB" % (os.path.getsize(out) / 1024))


if __name__ == "__main__":
    main()
