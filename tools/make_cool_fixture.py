#!/opt/conda/bin/python3.9
"""Writes tests/golden/cool_*.cool / .mcool with the REAL HDF5 library (h5py 3.3.0 / HDF5 1.10.6
of this image's Anaconda interpreter) in the layout `cooler` gives its files (schema v3:
chroms/{name,length}, bins/{chrom,start,end,weight}, pixels/{bin1_id,bin2_id,count},
indexes/{chrom_offset,bin1_offset}; chunked, gzip 6 + shuffle; root attributes), and next to
them the matrices `cooler.Cooler(path).matrix(balance=..., sparse=True).fetch(chrom)` returns
for such a file (upper-triangle pixels mirrored, trans pixels dropped, (w_i * w_j) * count), as
computed here from the same arrays with numpy.  `cooler` itself is not installed anywhere in
this image: the container bytes are genuine HDF5, the schema is restated from cooler's
documentation.  Run:  /opt/conda/bin/python3.9 tools/make_cool_fixture.py"""
import os
import numpy as np
import h5py

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
rng = np.random.RandomState(424242)
binsize = 10000
chroms = [("chr1", 3_004_000), ("chr2", 1_999_999), ("chrX", 1_210_000)]
nb = [(l + binsize - 1) // binsize for _, l in chroms]
off = np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
N = int(off[-1])
# upper-triangle pixels: a band of 60 bins around the diagonal, some far cis pixels, some trans
b1, b2, cnt = [], [], []
for c in range(len(chroms)):
    for i in range(off[c], off[c + 1]):
        for j in range(i, min(i + 60, off[c + 1])):
            lam = 40.0 / (1 + (j - i)) ** 0.9
            k = rng.poisson(lam)
            if k and rng.rand() < 0.92:
                b1.append(i); b2.append(j); cnt.append(k)
        for _ in range(2):  # far cis / trans
            j = rng.randint(i, N)
            if j - i >= 60:
                b1.append(i); b2.append(j); cnt.append(1 + rng.poisson(0.3))
order = np.lexsort((b2, b1))
b1 = np.asarray(b1, np.int64)[order]; b2 = np.asarray(b2, np.int64)[order]; cnt = np.asarray(cnt, np.int32)[order]
keep = np.concatenate([[True], (np.diff(b1) != 0) | (np.diff(b2) != 0)])
b1, b2, cnt = b1[keep], b2[keep], cnt[keep]
bin1_offset = np.searchsorted(b1, np.arange(N + 1), side="left").astype(np.int64)
weight = 1.0 / np.sqrt(rng.uniform(200, 900, N))
weight[rng.choice(N, 25, replace=False)] = np.nan
weight[off[1]:off[1] + 3] = np.nan
kr = rng.uniform(0.5, 2.0, N)

bin_chrom = np.repeat(np.arange(len(chroms)), nb).astype(np.int32)
bin_start = np.concatenate([np.arange(n) * binsize for n in nb]).astype(np.int32)
bin_end = np.concatenate([np.minimum((np.arange(n) + 1) * binsize, l) for n, (_, l) in zip(nb, chroms)]).astype(np.int32)


def write_cool(grp, bs, scale=1):
    kw = dict(compression="gzip", compression_opts=6, shuffle=True)
    names = np.array([c for c, _ in chroms], dtype="S")
    g = grp.create_group("chroms")
    g.create_dataset("name", data=names, chunks=(len(names),), **kw)
    g.create_dataset("length", data=np.array([l for _, l in chroms], np.int32), chunks=(len(names),), **kw)
    g = grp.create_group("bins")
    enum = h5py.enum_dtype({c: i for i, (c, _) in enumerate(chroms)}, basetype="i4")
    g.create_dataset("chrom", data=bin_chrom, dtype=enum, chunks=(min(N, 512),), **kw)
    g.create_dataset("start", data=bin_start * scale, chunks=(min(N, 512),), **kw)
    g.create_dataset("end", data=bin_end * scale, chunks=(min(N, 512),), **kw)
    d = g.create_dataset("weight", data=weight, chunks=(min(N, 512),), **kw)
    d.attrs["ignore_diags"] = 2
    d.attrs["converged"] = True
    g.create_dataset("KR", data=kr, chunks=(min(N, 512),), **kw)
    d = g.create_dataset("DIV", data=kr, chunks=(min(N, 512),), **kw)   # divisive by attribute, not by name
    d.attrs["divisive_weights"] = True
    d = g.create_dataset("VC", data=weight, chunks=(min(N, 512),), **kw)  # multiplicative although named VC
    d.attrs["divisive_weights"] = False
    g = grp.create_group("pixels")
    g.create_dataset("bin1_id", data=b1, chunks=(4096,), maxshape=(None,), **kw)
    g.create_dataset("bin2_id", data=b2, chunks=(4096,), maxshape=(None,), **kw)
    g.create_dataset("count", data=cnt, chunks=(4096,), maxshape=(None,), **kw)
    g = grp.create_group("indexes")
    g.create_dataset("chrom_offset", data=off, chunks=(len(off),), **kw)
    g.create_dataset("bin1_offset", data=bin1_offset, chunks=(min(N + 1, 512),), **kw)
    grp.attrs["format"] = "HDF5::Cooler"
    grp.attrs["format-version"] = 3
    grp.attrs["bin-type"] = "fixed"
    grp.attrs["bin-size"] = bs
    grp.attrs["storage-mode"] = "symmetric-upper"
    grp.attrs["nbins"] = N
    grp.attrs["nchroms"] = len(chroms)
    grp.attrs["nnz"] = len(cnt)
    grp.attrs["sum"] = int(cnt.sum())
    grp.attrs["genome-assembly"] = "unknown"


with h5py.File(os.path.join(out, "cool_small.cool"), "w") as f:
    write_cool(f, binsize)
with h5py.File(os.path.join(out, "cool_small.mcool"), "w") as f:
    r = f.create_group("resolutions")
    write_cool(r.create_group(str(binsize)), binsize)
    # a second, coarser entry (same arrays, another bin size on the label only: never read by the test)
    g2 = r.create_group(str(binsize * 2))
    g2.attrs["bin-size"] = binsize * 2
with h5py.File(os.path.join(out, "cool_small_latest.cool"), "w", libver="latest") as f:
    write_cool(f, binsize)

# what cooler's matrix(balance, sparse=True).fetch(chrom) holds, as CSR
exp = {"chromnames": np.array([c for c, _ in chroms]), "binsize": np.int64(binsize),
       "chromsizes": np.array([l for _, l in chroms], np.int64)}
for c, (name, _) in enumerate(chroms):
    lo, hi = off[c], off[c + 1]
    m = (b1 >= lo) & (b1 < hi) & (b2 >= lo) & (b2 < hi)
    i, j, v = b1[m] - lo, b2[m] - lo, cnt[m].astype(np.float64)
    n = hi - lo
    for tag, w in (("raw", None), ("weight", weight[lo:hi]), ("KR", kr[lo:hi])):
        # cooler.api.matrix: `mat.data = bias1[mat.row] * bias2[mat.col] * mat.data` -- the two
        # weights are multiplied first, so the mirrored entry gets the identical value
        # columns named KR / VC / SQRT_VC are divisive in cooler: the biases are inverted first
        # (`bias = 1 / bias`) and then applied like multiplicative ones
        if tag == "KR":
            w = 1.0 / w
        vv = v if w is None else w[i] * w[j] * v
        offd = i != j
        rows = np.concatenate([i, j[offd]]); cols = np.concatenate([j, i[offd]]); vals = np.concatenate([vv, vv[offd]])
        o = np.lexsort((cols, rows))
        rows, cols, vals = rows[o], cols[o], vals[o]
        indptr = np.searchsorted(rows, np.arange(n + 1)).astype(np.int64)
        exp["%s/%s/indptr" % (name, tag)] = indptr
        exp["%s/%s/indices" % (name, tag)] = cols.astype(np.int32)
        exp["%s/%s/data" % (name, tag)] = vals
    exp[name + "/weight"] = weight[lo:hi]
    exp[name + "/KR"] = kr[lo:hi]
    exp[name + "/n"] = np.int64(n)
np.savez_compressed(os.path.join(out, "cool_small_expected.npz"), **exp)
for fn in sorted(os.listdir(out)):
    if fn.startswith("cool_"):
        print(fn, os.path.getsize(os.path.join(out, fn)))
print("h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version, "pixels", len(cnt), "bins", N)

# ---- a second file that exercises the rest of what h5lite reads (not cooler-shaped)
exp2 = {}
with h5py.File(os.path.join(out, "h5lite_types.h5"), "w") as f:
    r2 = np.random.RandomState(7)
    def put(g, name, arr, **kw):
        g.create_dataset(name, data=arr, **kw)
        exp2[(g.name.rstrip("/") + "/" + name).lstrip("/")] = np.asarray(arr)
    put(f, "contiguous_f64", r2.rand(100))
    put(f, "chunked2d_i16", r2.randint(-3000, 3000, (37, 23)).astype(np.int16), chunks=(8, 8), compression="gzip")
    put(f, "chunked3d_u8", r2.randint(0, 255, (5, 6, 7)).astype(np.uint8), chunks=(2, 3, 4), shuffle=True,
        compression="gzip", compression_opts=1)
    put(f, "bigendian_i4", np.arange(-5, 20, dtype=">i4"))
    put(f, "f32", r2.rand(33).astype(np.float32), chunks=(10,))
    put(f, "u64", (r2.randint(0, 2 ** 62, 17).astype(np.uint64) * 3), chunks=(4,), shuffle=True)
    put(f, "strings", np.array([b"alpha", b"be", b"gamma77"], dtype="S7"))
    put(f, "checksummed", r2.randint(0, 100000, 3000).astype(np.int32), chunks=(512,), fletcher32=True,
        shuffle=True, compression="gzip")
    put(f, "empty", np.zeros(0, np.int64), chunks=(16,), maxshape=(None,))
    f.create_dataset("scalar", data=np.float64(2.5)); exp2["scalar"] = np.float64(2.5)
    f.create_dataset("unwritten", shape=(10,), dtype="i4", chunks=(4,)); exp2["unwritten"] = np.zeros(10, np.int32)
    g = f.create_group("a/b/c")
    put(g, "deep", np.arange(7))
    many = f.create_group("many")
    for k in range(70):  # more than one symbol-table node
        put(many, "d%03d" % k, np.arange(k, k + 3))
    f.attrs["float"] = 1.5
    f.attrs["ints"] = np.array([1, 2, 3], np.int16)
    f.attrs["fixed"] = np.string_("fixed-length")
    f.attrs["text"] = "variable length é"
    f["f32"].attrs["unit"] = "m"
np.savez_compressed(os.path.join(out, "h5lite_types_expected.npz"), **{k.replace("/", "|"): v for k, v in exp2.items()})
print("h5lite_types.h5", os.path.getsize(os.path.join(out, "h5lite_types.h5")))

# ---- libver='latest' with MANY chunks: deep extensible arrays (super blocks, paged data blocks)
with h5py.File(os.path.join(out, "h5lite_latest_many.h5"), "w", libver="latest") as f:
    def ramp(n):
        return (np.arange(n, dtype=np.int64) * 7919 % 30011).astype(np.int16)
    f.create_dataset("ea_3000_chunks", data=ramp(3000 * 4 + 3), chunks=(4,), maxshape=(None,))
    f.create_dataset("ea_20000_chunks", data=ramp(20000 * 2), chunks=(2,), maxshape=(None,))
    f.create_dataset("ea_filtered_5000_chunks", data=ramp(5000 * 16 + 1), chunks=(16,), maxshape=(None,),
                     compression="gzip", compression_opts=1, shuffle=True)
    f.create_dataset("fa_5000_chunks", data=ramp(5000 * 4), chunks=(4,))  # fixed array, paged
    f.create_dataset("fa_filtered_3000_chunks", data=ramp(3000 * 16), chunks=(16,), compression="gzip",
                     compression_opts=1)
    d = f.create_dataset("ea_sparse", shape=(64 * 300,), dtype="i2", chunks=(64,), maxshape=(None,))
    d[64 * 250:64 * 251] = ramp(64)  # only one chunk, far into the array, was ever written
print("h5lite_latest_many.h5", os.path.getsize(os.path.join(out, "h5lite_latest_many.h5")))
