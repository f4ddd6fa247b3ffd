#!/opt/conda/bin/python3.9
"""Writes the pieces tools/genome_standin.py synthesised as ONE .cool with the genuine HDF5
library (h5py / libhdf5 of the image's Anaconda interpreter), in the layout cooler gives its
files (schema v3, see tools/make_cool_fixture.py): resizable chunked datasets, gzip + shuffle.
The pixel table is appended chromosome by chromosome, so a genome of several 1e8 pixels never
sits in memory whole.
usage: /opt/conda/bin/python3.9 tools/write_cool_h5py.py <workdir> <out.cool> [gzip level] [chunk]
(chunk = pixels per HDF5 chunk; 0 = the library's own guess)"""
import json
import os
import sys
import time

import h5py
import numpy as np

work, out = sys.argv[1], sys.argv[2]
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 20
man = json.load(open(os.path.join(work, "manifest.json")))
binsize = int(man["binsize"])
chroms = man["chroms"]  # [{name, length, bins, offset, pixels}]
N = sum(c["bins"] for c in chroms)
kw = dict(compression="gzip", compression_opts=level, shuffle=True) if level > 0 else {}
t0 = time.time()
with h5py.File(out, "w") as f:
    g = f.create_group("chroms")
    g.create_dataset("name", data=np.array([c["name"] for c in chroms], dtype="S"), **kw)
    g.create_dataset("length", data=np.array([c["length"] for c in chroms], np.int32), **kw)
    g = f.create_group("bins")
    enum = h5py.enum_dtype({c["name"]: i for i, c in enumerate(chroms)}, basetype="i4")
    g.create_dataset("chrom", data=np.repeat(np.arange(len(chroms)), [c["bins"] for c in chroms]).astype(np.int32),
                     dtype=enum, **kw)
    st = np.concatenate([np.arange(c["bins"], dtype=np.int64) * binsize for c in chroms])
    en = np.concatenate([np.minimum((np.arange(c["bins"], dtype=np.int64) + 1) * binsize, c["length"])
                         for c in chroms])
    g.create_dataset("start", data=st.astype(np.int32), **kw)
    g.create_dataset("end", data=en.astype(np.int32), **kw)
    w = np.concatenate([np.load(os.path.join(work, "weights_%d.npy" % i)) for i in range(len(chroms))])
    d = g.create_dataset("weight", data=w, **kw)
    d.attrs["ignore_diags"] = 2
    d.attrs["converged"] = True
    g = f.create_group("pixels")
    ck = dict(chunks=(chunk,)) if chunk > 0 else dict(chunks=True)
    ds = {name: g.create_dataset(name, shape=(0,), dtype=dt, maxshape=(None,), **ck, **kw)
          for name, dt in (("bin1_id", np.int64), ("bin2_id", np.int64), ("count", np.int32))}
    bin1_offset = np.zeros(N + 1, np.int64)
    total, csum = 0, 0
    for i, c in enumerate(chroms):
        z = np.load(os.path.join(work, "pixels_%d.npz" % i))
        b1, b2, cnt = z["bin1"], z["bin2"], z["count"]
        for name, a in (("bin1_id", b1), ("bin2_id", b2), ("count", cnt)):
            ds[name].resize((total + a.size,))
            ds[name][total:total + a.size] = a
        lo, hi = c["offset"], c["offset"] + c["bins"]
        bin1_offset[lo:hi] = total + np.searchsorted(b1, np.arange(lo, hi))
        total += b1.size
        csum += int(cnt.sum())
        print("  %-6s %6d bins %10d pixels (%.0f s)" % (c["name"], c["bins"], b1.size, time.time() - t0), flush=True)
    bin1_offset[N] = total
    g = f.create_group("indexes")
    g.create_dataset("chrom_offset", data=np.array([c["offset"] for c in chroms] + [N], np.int64), **kw)
    g.create_dataset("bin1_offset", data=bin1_offset, **kw)
    f.attrs["format"] = "HDF5::Cooler"
    f.attrs["format-version"] = 3
    f.attrs["bin-type"] = "fixed"
    f.attrs["bin-size"] = binsize
    f.attrs["storage-mode"] = "symmetric-upper"
    f.attrs["nbins"] = N
    f.attrs["nchroms"] = len(chroms)
    f.attrs["nnz"] = total
    f.attrs["sum"] = csum
    f.attrs["genome-assembly"] = "hg19 (synthetic stand-in)"
print("wrote %s: %d bins, %d pixels, %.0f MB, gzip %d, chunk %s, %.0f s (h5py %s, hdf5 %s)"
      % (out, N, total, os.path.getsize(out) / 1e6, level, chunk or "auto", time.time() - t0, h5py.__version__,
         h5py.version.hdf5_version))
