#!/usr/bin/env python
"""GPU-box diagnostic: where does a tree group's time go inside the LDS forest
kernel?  Runs config 2 with in-kernel s_memtime stamps (workgroup 0 of the last
launch) and prints per-phase cycle shares.  Shares, not absolute times."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from peakachu_amd import _lib
from peakachu_amd.forest import FlatForest

# usage: stamps.py [bins [w [forest spec]]]   (defaults: config 2's shape; `11 random:500:20` = config 5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 5
Mf, e, x, y, upper = bench.build_workload(0, n, 200, w, 6, 200)
fo = bench.load_forest(sys.argv[3] if len(sys.argv) > 3 else None, w, (2 * w + 1) ** 2)
L = _lib.require_device()
dbg_extra = 0
for kv in filter(None, os.environ.get("PK_OPTS", "").split(",")):
    k, v = kv.split("=")
    if k == "forest_dbg":
        dbg_extra = int(v)
    else:
        _lib.set_option(k, int(v))
hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, Mf.shape[0], e, -2 * w + 1, upper + 2 * w - 1)
hf = _lib.HipForest(fo)
cd = _lib.HipCands(x, y)
if os.environ.get("PK_STAMP_CUT"):   # the head of the cut forest (PK_STAMP_CUT = group to cut in front of; 0 = the library's choice)
    cd.set_prune(True)
    hf.set_option("forest_split_min", 1)
    hf.set_option("forest_split_at", int(os.environ["PK_STAMP_CUT"]))
cd.run(hm, hf, w, 0.5)
hf.set_option("forest_dbg", 16 | dbg_extra)   # (in-kernel stamps: builds with -DPK_QR_STAMPS, tools/build_variant.sh)
cd.run(hm, hf, w, 0.5)
hf.set_option("forest_dbg", 0)
if (2 * w + 1) ** 2 > 255 and not os.environ.get("PK_STAMP_Q1"):
    # the two-tile kernel of the wide format (forest_q2_kernel): eight stamps per group
    buf = np.zeros(16 * 32 * 8, np.int64)
    _lib.check(L.pk_debug_read(0, buf, buf.size), "dbg")
    st = buf.reshape(16, 32, 8)
    ng = int((st[0, :, 0] != 0).sum())
    st = st[:, :ng, :].astype(np.float64)
    walkers = np.arange(16) < int(os.environ.get("PK_STAMP_WALKERS", "7"))  # waves with a tree
    names = ["issue loads + walk tile 1", "barrier", "issue loads + ordered sum + tile swap", "barrier",
             "walk tile 2", "barrier", "ordered sum + wait for the loads", "commit + barrier"]
    st = np.concatenate([st, np.concatenate([st[:, 1:, :1], st[:, -1:, 7:8]], axis=1)], axis=2)  # end = next start
    print("forest_q2_kernel: groups", ng, "(first 32 at most); cycles per group, mean over the %d walking waves | all 16"
          % walkers.sum())
    for k, nm in enumerate(names):
        d = st[:, :, k + 1] - st[:, :, k]
        d = d[:, :-1]
        print("  %-38s %8.0f | %8.0f" % (nm, d[walkers].mean(), d.mean()))
    print("  %-32s %8.0f" % ("total per group", (st[:, :-1, 8] - st[:, :-1, 0]).mean()))
    for k, nm in enumerate(names):
        print("  per wave, %-38s" % nm, np.round((st[:, :-1, k + 1] - st[:, :-1, k]).mean(1)).astype(int).tolist())
    sys.exit(0)
buf = np.zeros(16 * 32 * 5, np.int64)
_lib.check(L.pk_debug_read(0, buf, buf.size), "dbg")
st = buf.reshape(16, 32, 5)
tile = st[:, 31, :].astype(np.float64)   # forest_qr_kernel: stamps of the tile change
st = st[:, :31, :]
ng = int((st[0, :, 0] != 0).sum())
nw = int((st[:, 0, 0] != 0).sum())  # waves of the workgroup (2 per tree slot)
st = st[:nw, :ng, :].astype(np.float64)
walk = st[:, :, 1] - st[:, :, 0]
bar1 = st[:, :, 2] - st[:, :, 1]
commit = st[:, :, 3] - st[:, :, 2]
bar2 = st[:, :, 4] - st[:, :, 3]
tot = st[:, -1, 4] - st[:, 0, 0]
print("groups", ng, "(first 32 at most) cycles per group (mean over %d waves):" % nw)
print("  prefetch-issue+walk %8.0f  (min wave %.0f max wave %.0f)" % (walk.mean(), walk.mean(1).min(), walk.mean(1).max()))
print("  wait at barrier 1   %8.0f" % bar1.mean())
print("  commit+accumulate   %8.0f" % commit.mean())
print("  wait at barrier 2   %8.0f" % bar2.mean())
print("  total per group     %8.0f ; whole workgroup %.0f cycles" % ((st[:, :, 4] - st[:, :, 0]).mean(), tot.mean()))
print("per-wave walk means:", np.round(walk.mean(1)).astype(int).tolist())
park = buf.reshape(16, 32, 5)[:, 30, :].astype(np.float64)   # the head of the cut forest: phases of the parking
if os.environ.get("PK_STAMP_CUT") and park[:, 0].all() and park[:, 4].all():
    d = np.diff(park, axis=1)
    print("parking (head of the cut forest), cycles per wave: owners' list %s | wait at barrier 1 %s | copy %s | wait at barrier 2 %s"
          % tuple(np.round(d[:, k]).astype(int).tolist() for k in range(4)))
    p2 = buf.reshape(16, 32, 5)[:, 29, :].astype(np.float64)
    if p2[:, 0].all() and p2[:, 2].all():
        print("   inside `copy`: count read + block %s | owners' records %s | codes %s | rest %s" % (
            np.round(p2[:, 0] - park[:, 2]).astype(int).tolist(), np.round(p2[:, 1] - p2[:, 0]).astype(int).tolist(),
            np.round(p2[:, 2] - p2[:, 1]).astype(int).tolist(), np.round(park[:, 3] - p2[:, 2]).astype(int).tolist()))
if tile[:, 0].all() and tile[:, 3].all():
    # (stamp 4 is taken at the START of a trip: the one stored last belongs to the trip whose end
    # stamps 0-3 describe only when that trip was not the workgroup's last one)
    print("tile change (forest_qr_kernel), cycles, mean over waves: tile/group stores + barrier %.0f, last sum + prob store %.0f, "
          "cold first group %.0f | last group's end -> tile end %.0f"
          % ((tile[:, 1] - tile[:, 0]).mean(), (tile[:, 2] - tile[:, 1]).mean(), (tile[:, 3] - tile[:, 2]).mean(),
             (tile[:, 3] - st[:nw, ng - 1, 4]).mean() if ng else 0))
if os.environ.get("PK_STAMP_MATRIX"):
    np.set_printoptions(linewidth=250)
    print("walk cycles / 10 per group (rows) and wave (columns); * = slowest")
    for g in range(ng):
        row = walk[:, g]
        print("g%02d" % g, " ".join(("%4d%s" % (v / 10, "*" if v == row.max() else " ")) for v in row),
              " release-mean %5d" % (st[:, g, 2].mean() - st[:, g, 0].mean()))
    nodes = np.diff(fo.tree_off)
    print("nodes per tree:", nodes.tolist())
    print("tree depth:", [int(d) for d in fo.tree_depths()] if hasattr(fo, "tree_depths") else "n/a")
