#!/usr/bin/env python
"""`.cool` reader fuzz, two stages (build container only):
  1. /opt/conda/bin/python3.9 tests/fuzz/fuzz_cool.py make <dir> [n]  -- h5py 3.3 / libhdf5 1.10.6 write
     n random coolers (1-6 chromosomes incl. empty ones, 1-3000 bins, random chunk sizes,
     gzip levels / no compression / shuffle on-off, int32 / int64 / float64 counts, int8 /
     int32 enum base, libver earliest / latest, nested mcool groups, extra weight columns) and
     the matrices cooler's matrix(balance, sparse=True).fetch(chrom) would return;
  2. python tests/fuzz/fuzz_cool.py check <dir>  -- peakachu_amd.cool.CoolFile (pure Python) must
     return them bit for bit."""
import os, sys
import numpy as np


def make(out, n):
    import h5py
    os.makedirs(out, exist_ok=True)
    rng = np.random.RandomState(777)
    for k in range(n):
        nch = int(rng.randint(1, 7))
        binsize = int(rng.choice([1000, 5000, 10000, 25000]))
        nb = [int(rng.choice([0, 1, 2, 17, 300, 3000]) if rng.rand() < 0.3 else rng.randint(1, 400)) for _ in range(nch)]
        names = ["chr%d" % (i + 1) if rng.rand() < 0.8 else "scaffold_%d_random" % i for i in range(nch)]
        off = np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
        N = int(off[-1])
        b1, b2 = [], []
        for c in range(nch):
            if rng.rand() < 0.15:
                continue  # a chromosome without any pixel
            for i in range(off[c], off[c + 1]):
                m = rng.randint(0, 12)
                js = np.unique(np.minimum(i + rng.randint(0, 80, m), N - 1)) if N else np.zeros(0, int)
                for j in js:
                    b1.append(i); b2.append(int(j))
        b1 = np.asarray(b1, np.int64); b2 = np.asarray(b2, np.int64)
        o = np.lexsort((b2, b1)); b1, b2 = b1[o], b2[o]
        cdt = [np.int32, np.int64, np.float64][rng.randint(3)]
        cnt = (rng.poisson(5, b1.size) + 1).astype(cdt)
        if cdt is np.float64:
            cnt = cnt * 0.5
        bin1_offset = np.searchsorted(b1, np.arange(N + 1), side="left").astype(np.int64)
        weight = 1.0 / np.sqrt(rng.uniform(100, 900, N)) if N else np.zeros(0)
        if N > 3:
            weight[rng.choice(N, max(1, N // 40), replace=False)] = np.nan
        kr = rng.uniform(0.5, 2.0, N)
        latest = rng.rand() < 0.25
        nested = rng.rand() < 0.3
        path = os.path.join(out, "c%03d.%s" % (k, "mcool" if nested else "cool"))
        with h5py.File(path, "w", libver="latest" if latest else "earliest") as f:
            grp = f.create_group("resolutions/%d" % binsize) if nested else f
            comp = [dict(), dict(compression="gzip", compression_opts=int(rng.randint(1, 10))),
                    dict(compression="gzip", compression_opts=6, shuffle=True)][rng.randint(3)]
            def ds(g, name, data, resizable=False, **extra):
                data = np.asarray(data)
                kw = dict(comp)
                if data.size == 0 and not resizable:
                    kw = {}
                else:
                    kw["chunks"] = (int(max(1, min(max(data.size, 1), rng.choice([1, 7, 64, 1000, 100000])))),)
                if resizable:
                    kw["maxshape"] = (None,)
                kw.update(extra)
                return g.create_dataset(name, data=data, **kw)
            g = grp.create_group("chroms")
            ds(g, "name", np.array(names, dtype="S"))
            ds(g, "length", np.array([max(1, b) * binsize - int(rng.randint(0, binsize)) if b else 0 for b in nb], np.int32))
            g = grp.create_group("bins")
            base = "i1" if nch < 100 and rng.rand() < 0.5 else "i4"
            enum = h5py.enum_dtype({nm: i for i, nm in enumerate(names)}, basetype=base)
            ds(g, "chrom", np.repeat(np.arange(nch), nb).astype(base), dtype=enum)
            st = np.concatenate([np.arange(b) for b in nb]).astype(np.int32) * binsize if N else np.zeros(0, np.int32)
            ds(g, "start", st)
            ds(g, "end", st + binsize)
            ds(g, "weight", weight)
            ds(g, "KR", kr)
            for extra in range(int(rng.randint(0, 3))):
                ds(g, "VC%d" % extra, rng.rand(N))
            g = grp.create_group("pixels")
            ds(g, "bin1_id", b1, resizable=True)
            ds(g, "bin2_id", b2, resizable=True)
            ds(g, "count", cnt, resizable=True)
            g = grp.create_group("indexes")
            ds(g, "chrom_offset", off)
            ds(g, "bin1_offset", bin1_offset)
            grp.attrs["format"] = "HDF5::Cooler"
            grp.attrs["bin-size"] = binsize
            grp.attrs["storage-mode"] = "symmetric-upper"
            grp.attrs["nbins"] = N
        exp = {"names": np.array(names), "binsize": np.int64(binsize), "uri": np.array(
            os.path.basename(path) + ("::/resolutions/%d" % binsize if nested else ""))}
        for c, name in enumerate(names):
            lo, hi = off[c], off[c + 1]
            m = (b1 >= lo) & (b1 < hi) & (b2 >= lo) & (b2 < hi)
            i, j, v = b1[m] - lo, b2[m] - lo, cnt[m]
            nbin = hi - lo
            for tag, w in (("raw", None), ("weight", weight[lo:hi]), ("KR", kr[lo:hi])):
                # (divisive columns: the biases inverted first, then applied like multiplicative ones --
                # the order peakachu_amd.cool uses; unverified against cooler itself, see cool.py)
                vv = v if w is None else ((1.0 / w)[i] * (1.0 / w)[j] * v if tag == "KR" else w[i] * w[j] * v)
                od = i != j
                rows = np.concatenate([i, j[od]]); cols = np.concatenate([j, i[od]]); vals = np.concatenate([vv, vv[od]])
                oo = np.lexsort((cols, rows))
                exp["%d/%s/rows" % (c, tag)] = rows[oo]
                exp["%d/%s/cols" % (c, tag)] = cols[oo]
                exp["%d/%s/vals" % (c, tag)] = vals[oo].astype(np.float64)
            exp["%d/n" % c] = np.int64(nbin)
            exp["%d/w" % c] = weight[lo:hi]
        np.savez_compressed(os.path.join(out, "c%03d.npz" % k), **exp)
    print("wrote %d coolers with h5py %s / hdf5 %s" % (n, h5py.__version__, h5py.version.hdf5_version))


def check(out):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from peakachu_amd import cool, h5lite
    names = sorted(f[:-4] for f in os.listdir(out) if f.endswith(".npz"))
    nmat = refused = 0
    for nm in names:
        z = np.load(os.path.join(out, nm + ".npz"))
        uri = os.path.join(out, str(z["uri"]))
        try:
            c = cool.CoolFile(uri)
        except h5lite.H5Unsupported as e:
            print("%s: refused (%s)" % (nm, str(e)[:80]))
            refused += 1
            continue
        assert c.chromnames == [str(s) for s in z["names"]] and c.binsize == int(z["binsize"]), nm
        for ci, name in enumerate(c.chromnames):
            n = int(z["%d/n" % ci])
            assert c.chrom_bins(name) == n
            for tag, bal in (("raw", False), ("weight", "weight"), ("KR", "KR")):
                M = c.matrix(balance=bal, sparse=True).fetch(name)
                o = np.lexsort((M.col, M.row))
                ok = (M.shape == (n, n) and np.array_equal(M.row[o], z["%d/%s/rows" % (ci, tag)])
                      and np.array_equal(M.col[o], z["%d/%s/cols" % (ci, tag)])
                      and np.array_equal(M.data[o].astype(np.float64).view(np.uint64), z["%d/%s/vals" % (ci, tag)].view(np.uint64)))
                if not ok:
                    print("%s %s %s: MISMATCH" % (nm, name, tag))
                    sys.exit(1)
                nmat += 1
            assert np.array_equal(c.bins().fetch(name)["weight"].values, z["%d/w" % ci], equal_nan=True)
        c.close()
        print("%s %s: %d chromosomes ok" % (nm, str(z["uri"]), len(z["names"])))
    print("all %d files: %d matrices bit-identical (%d files refused with a message)" % (len(names), nmat, refused))


if __name__ == "__main__":
    if sys.argv[1] == "make":
        make(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40)
    else:
        check(sys.argv[2])
