#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/pmc.sh.

HBM-side bytes per launch of the two hot kernels: FETCH_SIZE and WRITE_SIZE
(rocprofv3 --pmc, separate passes, values in KiB summed over the dimensions
rocprofv3 reports), FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md
prescribes for gfx950.  Only the full-evaluation forest instantiation
(forest_lds_kernel<SLOTS, false>) is counted, not the early-exit extra pass.

usage: tools/make_traffic.py <pmc dir> <candidates in the workload> <launches per step> > traffic.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, n_cand, launches = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
acc = defaultdict(lambda: defaultdict(float))
ids = defaultdict(lambda: defaultdict(set))


def kclass(name):
    if "forest_lds_kernel" in name and "false>" in name.replace(" ", ""):
        return "forest"
    if "extract_pair_clean_kernel" in name or "extract_pair_kernel" in name:
        return "extract"
    return None


for path in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            k = kclass(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                ids[k][r["Counter_Name"]].add(r["Dispatch_Id"])
out = {}
for k in ("forest", "extract"):
    f = acc[k]["FETCH_SIZE"] / max(1, len(ids[k]["FETCH_SIZE"]))
    w = acc[k]["WRITE_SIZE"] / max(1, len(ids[k]["WRITE_SIZE"]))
    b = (2.0 * f + w) * 1024.0
    cpl = n_cand / launches
    out[k] = {"fetch_size_kib": round(f, 1), "write_size_kib": round(w, 1), "bytes_per_launch": b,
              "dispatches_seen": len(ids[k]["FETCH_SIZE"]), "candidates_per_launch": cpl,
              "bytes_per_candidate": b / cpl}
out["_note"] = ("HBM-side bytes per kernel launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate "
                "passes, tools/pmc.sh), FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; "
                "%d candidates in %d launches per step; made by tools/make_traffic.py" % (n_cand, launches))
print(json.dumps(out, indent=1))
