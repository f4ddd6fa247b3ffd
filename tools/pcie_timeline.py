#!/usr/bin/env python
"""GPU-box helper: where does `pk_score` (host coordinate / result buffers, bench.py's
`pcie_inclusive` leg) lose time against the device-resident pass?

  run:     rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o t -- \
               python3 tools/pcie_timeline.py run [calls]
  report:  python3 tools/pcie_timeline.py report DIR

`run` scores the headline workload `calls` times through pk_score and prints the host's wall time
per call; `report` lays the LAST call's kernels and copies on one time axis: per upload chunk the
copy's span, the kernels' span, and the idle gaps of the device between consecutive kernels."""
import csv
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(calls):
    import bench
    from peakachu_amd import _lib
    w, n, band = 5, 30000, 200
    Mf, e, x, y, upper = bench.build_workload(0, n, band, w, 6, band)
    fo = bench.load_forest(None, w, (2 * w + 1) ** 2)
    hm = _lib.HipMatrix(Mf.indptr, Mf.indices, Mf.data, Mf.shape[0], e, -2 * w + 1, upper + 2 * w - 1)
    hf = _lib.HipForest(fo)
    L = _lib.load()
    hm.score(hf, w, 0.5, x, y)
    L.pk_device_synchronize(0)
    for _ in range(calls):
        t0 = time.perf_counter()
        r = hm.score(hf, w, 0.5, x, y)
        L.pk_device_synchronize(0)
        print("call %.3f ms, %d pixels" % ((time.perf_counter() - t0) * 1e3, r[0].size), flush=True)
    t0 = time.perf_counter()
    for _ in range(20):
        r = hm.score(hf, w, 0.5, x, y)
    print("20 calls back to back: %.3f ms each" % ((time.perf_counter() - t0) * 1e3 / 20), flush=True)
    cd = _lib.HipCands(x, y)
    cd.run(hm, hf, w, 0.5)
    L.pk_device_synchronize(0)
    for _ in range(3):
        t0 = time.perf_counter()
        cd.run(hm, hf, w, 0.5)
        L.pk_device_synchronize(0)
        print("resident %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)


def report(d):
    ev = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:60]))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
    ev.sort()
    # the calls are separated by device idle time > 300 us with a resident run at the end: take the
    # groups, print the last pk_score group (it holds host-to-device copies)
    groups, cur = [], []
    for e in ev:
        if cur and e[0] - max(c[1] for c in cur) > 300000:
            groups.append(cur)
            cur = []
        cur.append(e)
    if cur:
        groups.append(cur)
    score = [g for g in groups if sum(1 for e in g if e[2].startswith("C ") and "HOST_TO_DEVICE" in e[2].upper().replace(" ", "_")) >= 4]
    if not score:
        score = groups
    g = score[-1]
    t0 = g[0][0]
    busy_end = t0
    idle = 0
    print("%d events, span %.3f ms" % (len(g), (max(e[1] for e in g) - t0) / 1e6))
    for s, e, name in g:
        gap = ""
        if name.startswith("K "):
            if s > busy_end:
                idle += s - busy_end
                gap = "  (device idle %.1f us before)" % ((s - busy_end) / 1e3)
            busy_end = max(busy_end, e)
        print("%9.1f %9.1f %8.1f us  %s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, name, gap))
    print("kernel-idle total %.1f us" % (idle / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    else:
        report(sys.argv[2])
