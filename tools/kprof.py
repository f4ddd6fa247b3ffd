#!/usr/bin/env python
"""GPU-box helper: device time per kernel class of config 2 (library HIP events), for
the option sets given as arguments ("opt=val,opt=val" ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from peakachu_amd import _lib
from peakachu_amd.forest import FlatForest

w = int(os.environ.get("PK_W", "5"))
Mf, e, x, y, upper = bench.build_workload(0, 30000, 200 if w == 5 else 300, w, 6, 200 if w == 5 else 300)
fo = FlatForest.load("peakachu_amd/data/forest_w%d_t100.npz" % w)
L = _lib.require_device()
hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, Mf.shape[0], e, -2 * w + 1, upper + 2 * w - 1)
hf = _lib.HipForest(fo)
xb = int(os.environ.get("PK_XBLOCK", "0"))
if xb:  # experiment: candidates tiled by x-range (x-block major, then diagonal, then x)
    import numpy as np
    o = np.lexsort((x, y - x, x // xb))
    x, y = x[o].copy(), y[o].copy()
cd = _lib.HipCands(x, y)
for opts in (sys.argv[1:] or [""]):
    old = {}
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("=")
        old[k] = L.pk_get_option(k.encode())   # (the process default: what the handles go back to)
        for h in (hm, hf, cd):                  # options are per handle: each takes what concerns it
            h.set_options({k: int(v)})
    cd.run(hm, hf, w, 0.5)
    L.pk_prof_enable(1); L.pk_prof_reset()
    steps = 5
    for _ in range(steps):
        cd.run(hm, hf, w, 0.5)
    r = {k: _lib.prof_get(k)[0] / steps for k in ("extract", "quant", "forest", "compact")}
    L.pk_prof_enable(0)
    print("%-40s" % opts, " ".join("%s %.3f" % kv for kv in r.items()), " total %.3f ms  -> %.0f M/s" % (sum(r.values()), x.size / sum(r.values()) / 1e3))
    for h in (hm, hf, cd):
        h.set_options(old)
