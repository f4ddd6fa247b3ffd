"""Rebuild inputs from the compact golden fixtures (tests/golden/*.npz).

Fixtures store symmetric count matrices as upper-triangle COO of small ints
(-1 marks a NaN cell); derived forms (balanced, .hic-style, band-filtered)
are recomputed here exactly as tools/make_golden.py computed them and are
pinned by the sha256 digests the fixtures carry.
"""
import hashlib
import os

import numpy as np
from scipy import sparse

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def sym_matrix(z, prefix):
    n = int(z[prefix + "_n"])
    r = z[prefix + "_urow"].astype(np.int64)
    c = z[prefix + "_ucol"].astype(np.int64)
    v = z[prefix + "_uval"].astype(np.float64)
    v[z[prefix + "_uval"] == -1] = np.nan
    off = r != c
    R = np.r_[r, c[off]]
    C = np.r_[c, r[off]]
    V = np.r_[v, v[off]]
    M = sparse.csr_matrix((V, (R, C)), shape=(n, n), dtype=np.float64)
    M.sum_duplicates()
    M.sort_indices()
    return M


def balance(M, weights):
    coo = M.tocoo()
    data = coo.data * weights[coo.row] * weights[coo.col]
    B = sparse.csr_matrix((data, (coo.row, coo.col)), shape=M.shape, dtype=np.float64)
    B.sort_indices()
    return B


def hicstyle(M, weights):
    B = sparse.csr_matrix(balance(M, weights))
    B.data = np.where(np.isfinite(B.data), B.data, 0.0)
    B.eliminate_zeros()
    return B * 200.0


def digest(M):
    M = sparse.csr_matrix(M, dtype=np.float64)
    M.sum_duplicates()
    M.sort_indices()
    h = hashlib.sha256()
    h.update(M.indptr.astype(np.int32).tobytes())
    h.update(M.indices.astype(np.int32).tobytes())
    h.update(M.data.astype(np.float64).tobytes())
    return h.hexdigest()


def forest(z_or_name):
    z = load(z_or_name) if isinstance(z_or_name, str) else z_or_name
    return {k[3:]: z[k] for k in z.files if k.startswith("fo_")}


def bits(a):
    """View float64 as uint64 so that comparisons are bit-exact (NaN-safe)."""
    return np.ascontiguousarray(a, np.float64).view(np.uint64)
