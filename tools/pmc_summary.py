#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes: per kernel, per counter, the mean
value per dispatch (summed over the dimensions rocprofv3 reports)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(set))
for path in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            k = k.split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n = max(1, len(cnt[k][c]))
        print("   %-24s %16.1f per dispatch  (%d dispatches)" % (c, acc[k][c] / n, n))
