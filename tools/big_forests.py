#!/usr/bin/env python
"""GPU-box helper: forests far larger than the fuzzer draws (thousands of trees, tens of thousands of
stumps) through HipForest.predict against the oracle -- group tables, image offsets and tree counts
beyond the sizes of the bench models."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
from oracle import oracle_np as onp
from peakachu_amd import _lib
from peakachu_amd.forest import FlatForest
from fuzz_forest import random_forest

L = _lib.require_device()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for F, T, max_nodes, depth, N in ((121, 3000, 2501, 20, 20000), (529, 2000, 1501, 20, 6000), (121, 20000, 31, 5, 20000),
                                  (25, 60000, 3, 1, 30000), (169, 1200, 9001, 25, 10000), (900, 1500, 301, 12, 4000)):
    t0 = time.time()
    fo = random_forest(rng, F, T, max_nodes, depth, 0.1, None, False)
    X = (rng.random((N, F)) * 2 - 0.5).astype(np.float32)
    X[N // 2, :] = np.nan
    ref = onp.predict(fo, X)
    ff = FlatForest(F, *(fo[k] for k in FlatForest.FIELDS))
    L.pk_prof_enable(1); L.pk_prof_reset()
    p = _lib.HipForest(ff).predict(X)
    rank = _lib.prof_get("quant")[1] > 0
    L.pk_prof_enable(0)
    ok = np.array_equal(p.view(np.uint64), ref.view(np.uint64))
    print("F=%4d T=%6d nodes<=%5d depth=%2d N=%6d (%d nodes): %s kernels, %s, %.1f s" % (
        F, T, max_nodes, depth, N, fo["left"].size, "rank" if rank else "float", "bit-exact" if ok else "MISMATCH", time.time() - t0), flush=True)
    if not ok:
        sys.exit(1)
