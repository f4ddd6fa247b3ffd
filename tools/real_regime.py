#!/usr/bin/env python
"""GPU-box helper: the candidate regime the CLI runs in (bench.real_regime) on the config-2
matrix, printed as a table.  `python tools/real_regime.py [bins band w]`."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from peakachu_amd import _lib, synth, utils  # noqa: E402

n, band, w = [int(v) for v in (sys.argv[1:4] + ["30000", "200", "5"][len(sys.argv[1:4]):])]
L = _lib.require_device()
F = (2 * w + 1) ** 2
fo = bench.load_forest(None, w, F)
M, _ = synth.synth_band(n, band, seed=0)
upper = min(band, n - 2 * w)
exp_arr = utils.calculate_expected(M, upper + 2 * w, raw=True)
Mf = utils.band_filter(M, w, upper)
x, y = synth.all_band_pixels(Mf, max(6, w + 1), upper)
hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, n, exp_arr, -2 * w + 1, upper + 2 * w - 1)
hf = _lib.HipForest(fo)
# PK_RR_POISSON_ONLY=1: only the list get_candidate makes (the leg whose rocprofv3 --stats summary is kept)
rr = bench.real_regime(L, 0, M, fo, w, 6, upper, 0.5, 100000, x, y, hm, hf,
                       strided=os.environ.get("PK_RR_POISSON_ONLY") != "1")
for p in rr["chromosome_cold"]["passes"]:
    print("cold: construct %.2f ms, score %.2f ms, %d candidates, %d pixels"
          % (p["construct_ms"], p["score_ms"], p["candidates"], p["scored_pixels"]))
for g in rr["legs"]:
    k = g["kernel_us_per_call"]
    print("%-40s %9d cand  %9.1f us/call  %8.1f M/s  frac %.3f | extract %7.1f quant %7.1f forest %7.1f compact %6.1f us"
          % (g["list"], g["candidates"], g["us_per_call"], g["value"] / 1e6, g["whole_path_frac"],
             k["extract"], k["quant"], k["forest"], k["compact"]))
if os.environ.get("PK_RR_JSON"):
    json.dump(rr, open(os.environ["PK_RR_JSON"], "w"))
