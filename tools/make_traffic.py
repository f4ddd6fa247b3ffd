#!/usr/bin/env python3
"""profiles/pmc.json from the PMC passes of tools/pmc.sh.

Per kernel class (extract / quant / forest), the mean per launch of every counter the
passes collected, plus what bench.py derives its roofline blocks from:
  * HBM-side bytes per launch: FETCH_SIZE and WRITE_SIZE (rocprofv3 --pmc, separate
    passes, KiB), FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes
    for gfx950;
  * cycles per launch: GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs);
  * SQ_INSTS_VALU, SQ_INSTS_LDS (wave instructions), SQ_LDS_IDX_ACTIVE,
    SQ_LDS_BANK_CONFLICT (LDS array cycles, summed over the CUs).
Only full-evaluation instantiations are counted (not the early-exit extra pass of
bench.py).  The file carries a hash of the kernel sources it was measured on; bench.py
reports the derived figures as stale when the sources have changed since.

usage: tools/make_traffic.py <pmc dir> <candidates in the workload> > pmc.json
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


sys.path.insert(0, ROOT)
from tools.srchash import source_sha  # noqa: E402  (token-stream hash: comments do not change it)


def kclass(name):
    n = name.replace(" ", "")
    if "forest_qr_kernel" in n:
        # forest_qr_kernel<HALF1, PRUNE, NR, SPLIT>: PRUNE = true is the in-kernel early exit (not counted);
        # SPLIT 1 = the head of the cut forest (what the headline runs: the dominant kernel), 2 = its tail,
        # 0 = the one-launch kernel of the full-evaluation leg (class "forest" only when nothing was cut)
        m = re.search(r"forest_qr_kernel<\d+,(true|false),\d+,(\d)>", n)
        if not m or m.group(1) == "true":
            return None
        return {"0": "forest_full", "1": "forest", "2": "forest_tail"}[m.group(2)]
    if "forest_q2_kernel" in n:   # forest_q2_kernel<SPLIT>
        m = re.search(r"forest_q2_kernel<(\d)>", n)
        return {"0": "forest_full", "1": "forest", "2": "forest_tail"}[m.group(1)] if m else "forest"
    if "forest_q_kernel" in n:
        # forest_q_kernel<CH, WPT, HALF1, PRUNE, EARLY>: the full evaluation is PRUNE = false AND
        # EARLY = false (`<..., true, false>` -- the early-exit extra pass -- also ends in "false>")
        return "forest" if re.search(r"forest_q_kernel<[^>]*,false,false>", n) else None
    if "forest_img_kernel" in n or "forest_img2_kernel" in n or "forest_lds_kernel" in n:
        return "forest" if "false>" in n else None
    if "quantize_tiles_kernel" in n:
        return "quant"
    if "extract_" in n and "_kernel" in n:
        return "extract"
    return None


def main():
    root, n_cand = sys.argv[1], int(sys.argv[2])
    acc = defaultdict(lambda: defaultdict(float))
    ids = defaultdict(lambda: defaultdict(set))
    names = defaultdict(set)
    for path in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                k = kclass(r["Kernel_Name"])
                if k:
                    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    ids[k][r["Counter_Name"]].add(r["Dispatch_Id"])
                    names[k].add(r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0])
    if "forest" not in acc and "forest_full" in acc:   # nothing was cut: the one-launch kernel is THE forest kernel
        acc["forest"], ids["forest"], names["forest"] = acc.pop("forest_full"), ids.pop("forest_full"), names.pop("forest_full")
    out = {"source_sha": source_sha(), "candidates": n_cand}
    for k in ("extract", "quant", "forest", "forest_tail", "forest_full"):
        if k not in acc:
            continue
        per = {c: acc[k][c] / max(1, len(ids[k][c])) for c in acc[k]}
        n_disp = max(len(v) for v in ids[k].values())
        # launches per step: dispatches seen in one pass / steps of that pass (bench --steps 2 --warmup 1 = 3 passes
        # over the candidate list, plus the early-exit extra which kclass() filters for the forest)
        d = {"kernels": sorted(names[k]), "counters_per_launch": {c: round(v, 1) for c, v in sorted(per.items())},
             "dispatches_seen": n_disp}
        if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
            d["hbm_bytes_per_launch"] = (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0
        if "GRBM_GUI_ACTIVE" in per:
            d["cycles_per_launch"] = per["GRBM_GUI_ACTIVE"] / 8.0
        out[k] = d
    out["_note"] = ("means per kernel launch from rocprofv3 --pmc passes (tools/pmc.sh: separate passes, "
                    "--kernel-trace only); hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per "
                    "MI355X_MICROARCH.md (gfx950 reports half of streamed reads); cycles = GRBM_GUI_ACTIVE / 8 XCDs")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
