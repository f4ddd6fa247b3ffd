#!/usr/bin/env python
"""GPU-box helper: wall-clock profile of `score_genome` on a synthetic genome
(.pkmap.npz container, raw mode and balanced mode), per stage."""
import cProfile, io, os, pstats, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peakachu_amd import cli, io as pkio, synth

sizes = [int(a) for a in (sys.argv[1:] or ["24000", "18000", "13000", "8000", "4800"])]
tmp = tempfile.mkdtemp()
cont = os.path.join(tmp, "genome.pkmap.npz")
t0 = time.time()
chroms = {}
for i, n in enumerate(sizes):
    M, _ = synth.synth_band(n, 300, seed=i)
    chroms["chr%d" % (i + 1)] = (M, synth.synth_weights(n, i, n_nan=5))
print("synthesised %d chromosomes (%d bins) in %.1f s" % (len(sizes), sum(sizes), time.time() - t0))
t0 = time.time()
pkio.write_pkmap(cont, chroms)
print("container written in %.1f s (%.0f MB)" % (time.time() - t0, os.path.getsize(cont) / 1e6))
model = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "peakachu_amd", "data", "forest_w5_t100.npz")
for wname in ("raw", "weight"):
    out = os.path.join(tmp, wname + ".bedpe")
    argv = ["score_genome", "-p", cont, "-m", model, "-O", out, "--clr-weight-name", wname, "-u", "300"]
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    buf = io.StringIO(); old = sys.stdout; sys.stdout = buf
    try:
        cli.run(argv)
    finally:
        sys.stdout = old
    pr.disable()
    dt = time.time() - t0
    lines = sum(1 for _ in open(out)) if os.path.exists(out) else 0
    print("== score_genome --clr-weight-name %s: %.2f s wall, %d scored pixels" % (wname, dt, lines))
    st = io.StringIO()
    pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(14)
    for l in st.getvalue().splitlines():
        if "/peakachu_amd/" in l or "scipy" in l or "numpy" in l or "method" in l:
            print("   ", l.strip()[:150])
