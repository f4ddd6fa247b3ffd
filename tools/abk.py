#!/usr/bin/env python
"""GPU-box helper: A/B of library builds (tools/ab/*.so, tools/build_variant.sh) on ONE box through
the library's own HIP-event timers, config 2 (PK_W=6: the w = 6 map; PK_W=11 PK_FOREST=<file or
random:500:20>: configs[4]; PK_BINS / PK_BAND: another map).  The workload is built once
and handed to one child process per build and repetition (PEAKACHU_HIP_LIB selects the build);
every child also prints the number of scored pixels and a checksum of all probabilities, so a
variant that changes a result shows.
usage: tools/abk.py [--reps 2] [--steps 10] [--opts "name=val,..."] a.so b.so:opt=val,... ...   ("-" = the product's library; ":opts" = library options of that entry)"""
import os, subprocess, sys, tempfile, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def child(path, steps, opts):
    from peakachu_amd import _lib
    from peakachu_amd.forest import FlatForest
    d = np.load(path)
    w = int(d["w"])
    import bench
    fo = bench.load_forest(os.environ.get("PK_FOREST") or None, w, (2 * w + 1) ** 2)   # (default: forest_w<w>_t100)
    L = _lib.require_device()
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    hm = _lib.HipMatrix(d["indptr"], d["indices"], d["data"], int(d["n"]), d["e"], -2 * w + 1, int(d["upper"]) + 2 * w - 1)
    hf = _lib.HipForest(fo)
    cd = _lib.HipCands(d["x"], d["y"])
    for _ in range(3):
        out = cd.run(hm, hf, w, 0.5)
    L.pk_prof_enable(1); L.pk_prof_reset()
    for _ in range(steps):
        cd.run(hm, hf, w, 0.5)
    r = {k: _lib.prof_get(k)[0] / steps for k in ("extract", "quant", "forest", "compact")}
    L.pk_prof_enable(0)
    import time
    L.pk_device_synchronize(0)
    t0 = time.perf_counter()
    for _ in range(steps):
        cd.run(hm, hf, w, 0.5)
    L.pk_device_synchronize(0)
    wall = (time.perf_counter() - t0) / steps * 1e3
    st, pr = cd.fetch_all()
    crc = zlib.crc32(pr.tobytes()) ^ zlib.crc32(st.tobytes())
    print(" ".join("%s %.3f" % kv for kv in r.items()), " total %.3f ms -> %.0f M/s  wall %.3f ms  pixels %d crc %08x"
          % (sum(r.values()), d["x"].size / sum(r.values()) / 1e3, wall, out, crc), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--child":
        child(a[1], int(a[2]), a[3])
        sys.exit(0)
    reps, steps, opts = 2, 10, ""
    while a and a[0].startswith("--"):
        if a[0] == "--reps": reps = int(a[1])
        elif a[0] == "--steps": steps = int(a[1])
        elif a[0] == "--opts": opts = a[1]
        a = a[2:]
    import bench
    w = int(os.environ.get("PK_W", "5"))
    band = int(os.environ.get("PK_BAND", 300 if w == 6 else 200))
    Mf, e, x, y, upper = bench.build_workload(0, int(os.environ.get("PK_BINS", 8000 if w == 11 else 30000)), band, w, 6, band)
    path = os.path.join(tempfile.gettempdir(), "pk_abk_w%d.npz" % w)
    np.savez(path, indptr=Mf.indptr, indices=Mf.indices, data=Mf.data, n=Mf.shape[0], e=e, x=x, y=y, upper=upper, w=w)
    for rep in range(reps):
        for ent in a:
            so, _, o2 = ent.partition(":")   # "build.so:opt=val,opt=val": options of this entry only
            env = dict(os.environ)
            if so != "-":
                env["PEAKACHU_HIP_LIB"] = os.path.join(ROOT, "tools", "ab", so)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path, str(steps), ",".join(filter(None, (opts, o2)))],
                               env=env, capture_output=True, text=True, timeout=300)
            print("%-24s %s" % (ent, p.stdout.strip().splitlines()[-1] if p.returncode == 0 and p.stdout.strip() else "FAILED rc=%d %s" % (p.returncode, p.stderr[-300:])), flush=True)
