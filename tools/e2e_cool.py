#!/usr/bin/env python
"""GPU-box helper: `score_genome` end to end on a `.cool` file of realistic size, read by the
built-in reader (no cooler / h5py in the scoring interpreter).  The file itself is written
first by the image's Anaconda interpreter (h5py), in cooler's layout:
    /opt/conda/bin/python3.9 tools/e2e_cool.py make /tmp/e2e.cool
    python tools/e2e_cool.py run /tmp/e2e.cool"""
import os, sys, time
import numpy as np


def make(path):
    import h5py
    rng = np.random.RandomState(1)
    nb = [25000, 18000, 12000]
    off = np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
    N = int(off[-1])
    b1, b2, c = [], [], []
    for ci, n in enumerate(nb):
        for d in range(0, 320):
            i = np.arange(0, n - d, dtype=np.int64)
            lam = 200.0 / (1 + d) ** 0.9 + 0.3
            v = rng.poisson(lam, i.size).astype(np.int32)
            keep = v > 0
            b1.append(i[keep] + off[ci]); b2.append(i[keep] + d + off[ci]); c.append(v[keep])
    b1 = np.concatenate(b1); b2 = np.concatenate(b2); c = np.concatenate(c)
    o = np.lexsort((b2, b1)); b1, b2, c = b1[o], b2[o], c[o]
    with h5py.File(path, "w") as f:
        kw = dict(compression="gzip", compression_opts=6, shuffle=True)
        g = f.create_group("chroms")
        g.create_dataset("name", data=np.array([b"chr1", b"chr2", b"chr3"])); g.create_dataset("length", data=np.array(nb, np.int32) * 10000)
        g = f.create_group("bins")
        st = np.concatenate([np.arange(n) for n in nb]).astype(np.int32) * 10000
        g.create_dataset("start", data=st, **kw); g.create_dataset("end", data=st + 10000, **kw)
        g.create_dataset("weight", data=1.0 / np.sqrt(rng.uniform(150, 250, N)), **kw)
        g = f.create_group("pixels")
        for name, a in (("bin1_id", b1), ("bin2_id", b2), ("count", c)):
            g.create_dataset(name, data=a, chunks=(1 << 20,), maxshape=(None,), **kw)
        g = f.create_group("indexes")
        g.create_dataset("chrom_offset", data=off, **kw)
        g.create_dataset("bin1_offset", data=np.searchsorted(b1, np.arange(N + 1)).astype(np.int64), **kw)
        f.attrs["bin-size"] = 10000; f.attrs["storage-mode"] = "symmetric-upper"; f.attrs["format"] = "HDF5::Cooler"
    print("wrote %s: %d bins, %d pixels, %.0f MB" % (path, N, b1.size, os.path.getsize(path) / 1e6))


def run(path):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from peakachu_amd import cli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = os.path.join(root, "peakachu_amd", "data", "forest_w5_t100.npz")
    for wname in ("raw", "weight", "raw"):
        out = "/tmp/e2e_cool_%s.bedpe" % wname
        t0 = time.time()
        cli.run(["score_genome", "-p", path, "-m", model, "-O", out, "--clr-weight-name", wname, "-r", "10000",
                 "-C"])
        dt = time.time() - t0
        n = sum(1 for _ in open(out)) if os.path.exists(out) else 0
        print("score_genome -p %s --clr-weight-name %s: %.2f s wall, %d scored pixels" % (os.path.basename(path), wname, dt, n))
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    cli.run(["score_genome", "-p", path, "-m", model, "-O", "/tmp/e2e_cool_p.bedpe", "--clr-weight-name", "weight",
             "-r", "10000", "-C"])
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)


if __name__ == "__main__":
    (make if sys.argv[1] == "make" else run)(sys.argv[2])
