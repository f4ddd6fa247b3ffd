"""Contact-map access for the scoring drivers.

The reference reads `.cool` through `cooler` and `.hic` through `straw`
(peakachu/score_genome.py:26-35,55-57; peakachu/utils.py:17-58).  Neither
package (nor a contact map) is available in the build environment, so the
drivers accept
  * a `.cool` / `.mcool::/resolutions/N` URI -- through `cooler` when it is importable (same
    calls as the reference), otherwise through `cool.CoolFile`, which reads the HDF5
    container itself (`h5lite`),
  * a `.pkmap.npz` container written by `write_pkmap` -- per chromosome the
    symmetric raw-count CSR and, optionally, balancing weights -- which is
    what the synthetic configurations use.
Both are served through the three calls the reference makes on a Cooler:
`chromnames`, `matrix(balance=..., sparse=True).fetch(chrom)` (a symmetric
COO, NaN where a weight is NaN) and `bins().fetch(chrom)[name].values`.
"""
import numpy as np
from scipy import sparse


def write_pkmap(path, chroms, resolution=10000, weight_name="weight", compress=False):
    """chroms: ordered dict name -> (raw symmetric CSR, weights or None).
    Uncompressed by default: inflating a compressed container costs more host time per
    chromosome than the whole GPU scoring of it."""
    out = {"chromnames": np.array(list(chroms.keys())), "resolution": np.int64(resolution),
           "weight_name": np.array(weight_name)}
    for name, (M, w) in chroms.items():
        M = sparse.csr_matrix(M, dtype=np.float64)
        M.sum_duplicates()
        M.sort_indices()
        out[name + "/indptr"] = M.indptr.astype(np.int64)
        out[name + "/indices"] = M.indices.astype(np.int32)
        out[name + "/data"] = M.data
        out[name + "/n"] = np.int64(M.shape[0])
        if w is not None:
            out[name + "/weights"] = np.asarray(w, np.float64)
    (np.savez_compressed if compress else np.savez)(path, **out)


class _Selector:
    def __init__(self, fn):
        self._fn = fn

    def fetch(self, chrom):
        return self._fn(chrom)


class _Bins(dict):
    pass


class PkMap:
    """Cooler-like view of a .pkmap.npz container."""

    def __init__(self, path):
        self._z = np.load(path, allow_pickle=False)
        self.chromnames = [str(c) for c in self._z["chromnames"]]
        self.binsize = int(self._z["resolution"])
        self.weight_name = str(self._z["weight_name"])

    def chrom_bins(self, chrom):
        """Number of bins of a chromosome, from the container's metadata (no matrix is
        decompressed for it)."""
        return int(self._z[chrom + "/n"])

    def _raw(self, chrom):
        # score_genome fetches a chromosome twice in balanced mode (raw counts for the
        # Poisson candidates, balanced values for the windows): decompress it once
        if getattr(self, "_last", (None, None))[0] != chrom:
            z = self._z
            n = int(z[chrom + "/n"])
            self._last = (chrom, sparse.csr_matrix((z[chrom + "/data"], z[chrom + "/indices"],
                                                    z[chrom + "/indptr"]), shape=(n, n)))
        return self._last[1]

    def _weights(self, chrom):
        key = chrom + "/weights"
        if key not in self._z.files:
            raise KeyError("no balancing weights stored for %s" % chrom)
        return self._z[key]

    def matrix(self, balance=False, sparse=True):
        def fetch(chrom):
            M = self._raw(chrom).tocoo()
            if balance:
                w = self._weights(chrom)
                M.data = M.data * w[M.row] * w[M.col]
            return M
        return _Selector(fetch)

    def bins(self):
        def fetch(chrom):
            class _Col:
                def __init__(self, v):
                    self.values = v
            b = _Bins()
            b[self.weight_name] = _Col(self._weights(chrom))
            return b
        return _Selector(fetch)


def chrom_bins(lib, chrom):
    """Bins of `chrom` without fetching its matrix: PkMap metadata, or cooler's
    chromsizes / binsize (what cooler itself derives the bin table from)."""
    if hasattr(lib, "chrom_bins"):
        return lib.chrom_bins(chrom)
    size, binsize = int(lib.chromsizes[chrom]), int(lib.binsize)
    return (size + binsize - 1) // binsize


def open_map(path):
    """`-p/--path`: a .pkmap.npz container, or a .cool / .mcool::/resolutions/N URI -- through
    `cooler` when it is installed (the reference's own reader), else through the built-in
    reader (`cool.CoolFile`: pure Python, no h5py)."""
    if str(path).endswith(".npz"):
        return PkMap(path)
    try:
        import cooler
    except ImportError:
        from . import cool
        if str(path).partition("::")[0].endswith(".hic"):
            raise ImportError("reading %s needs `hic-straw` (not installed); convert it with "
                              "`hic2cool` and pass the .cool / .mcool" % path)
        return cool.CoolFile(path)
    return cooler.Cooler(path)
