#!/usr/bin/env python
"""GPU-box helper: how many trees could an exact early-termination rule skip on
config 2?  (Upper bound estimated from the final probabilities.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from peakachu_amd import _lib
from peakachu_amd.forest import FlatForest
Mf, e, x, y, upper = bench.build_workload(0, 30000, 200, 5, 6, 200)
fo = FlatForest.load("peakachu_amd/data/forest_w5_t100.npz")
hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, Mf.shape[0], e, -9, upper + 9)
hf = _lib.HipForest(fo); cd = _lib.HipCands(x, y)
cd.run(hm, hf, 5, 0.5)
st, p = cd.fetch_all()
T, thre = 100, 0.5
print("candidates", p.size, "survivors", int(st.sum()), "p>0.5:", int((p > thre).sum()))
print("p quantiles 50/90/99/99.9:", np.quantile(p, [0.5, 0.9, 0.99, 0.999]))
for blk in (64, 128):
    n = p.size // blk * blk
    pm = p[:n].reshape(-1, blk).max(1)
    # trees needed until p*t + (T - t) < thre*T for the block's worst candidate
    t_exit = np.where(pm < thre, np.ceil((T - thre * T) / (1 - np.minimum(pm, 0.499999))), T)
    t_exit = np.minimum(t_exit, T)
    print("block %3d: mean trees needed %.1f of %d -> forest time x%.2f ; blocks with a hit %.1f%%" %
          (blk, t_exit.mean(), T, t_exit.mean() / T, 100 * (pm >= thre).mean()))
