"""Synthetic Hi-C-like band matrices.

There is no contact map in the build environment, so every measured
configuration runs on seeded synthetic matrices of the shapes
BASELINE.json names: a symmetric matrix whose counts live on the
diagonals 0..band, Poisson-distributed around a power-law distance decay,
with planted 3x3 "loop" bumps.  Host-side only (numpy/scipy).
"""
import numpy as np
from scipy import sparse


def synth_band(n, band, seed=0, loops=None, bump=30.0):
    """Symmetric raw-count CSR (float64, canonical) of side `n`.

    count[i, i+d] ~ Poisson(200/(1+d)**0.9 + 0.3) for d in 0..band, plus
    `loops` 3x3 bumps of height `bump` at uniformly drawn (a, a+d),
    d in [8, band-2].  Returns (M, loop_coords[int64 (L,2)]).
    """
    rng = np.random.default_rng(seed)
    if loops is None:
        loops = max(1, n // 40)
    d = np.arange(band + 1)
    lam = 200.0 / (1.0 + d) ** 0.9 + 0.3
    cnt = rng.poisson(lam[:, None], size=(band + 1, n)).astype(np.float64)
    # planted loops
    la = rng.integers(2, max(3, n - band - 3), size=loops)
    ld = rng.integers(8, max(9, band - 2), size=loops)
    for a, dd in zip(la, ld):
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                i, j = a + di, a + dd + dj
                if 0 <= i < n and 0 <= j < n and 0 <= j - i <= band:
                    cnt[j - i, i] += bump
    ii = np.broadcast_to(np.arange(n), cnt.shape)
    dd = np.broadcast_to(d[:, None], cnt.shape)
    ok = (ii + dd < n) & (cnt != 0)
    r = ii[ok].astype(np.int64)
    c = r + dd[ok]
    v = cnt[ok]
    off = r != c
    R = np.concatenate([r, c[off]])
    C = np.concatenate([c, r[off]])
    V = np.concatenate([v, v[off]])
    M = sparse.csr_matrix((V, (R, C)), shape=(n, n), dtype=np.float64)
    M.sum_duplicates()
    M.sort_indices()
    return M, np.stack([la, la + ld], axis=1).astype(np.int64)


def synth_weights(n, seed=0, n_nan=5):
    """Balancing weights w_i = 1/sqrt(200*U(0.7,1.3)) with `n_nan` NaNs."""
    rng = np.random.default_rng(seed + 7919)
    w = 1.0 / np.sqrt(200.0 * rng.uniform(0.7, 1.3, size=n))
    if n_nan:
        w[rng.choice(n, size=min(n_nan, n), replace=False)] = np.nan
    return w


def balance(M, weights):
    """What cooler's matrix(balance=name) returns: raw[i,j]*w_i*w_j, NaN
    where either weight is NaN (entries kept, value NaN)."""
    coo = M.tocoo()
    data = coo.data * weights[coo.row] * weights[coo.col]
    B = sparse.csr_matrix((data, (coo.row, coo.col)), shape=M.shape, dtype=np.float64)
    B.sort_indices()
    return B


def all_band_pixels(M, lower, upper):
    """Every non-zero pixel with lower <= col-row <= upper, in the
    reference's candidate order (diagonal ascending, then row ascending;
    peakachu/scoreUtils.py:46-68).  Returns int32 x, y."""
    coo = M.tocoo()
    d = coo.col.astype(np.int64) - coo.row.astype(np.int64)
    ok = (d >= lower) & (d <= upper) & (coo.data != 0) & np.isfinite(coo.data)
    r, dd = coo.row[ok].astype(np.int64), d[ok]
    order = np.lexsort((r, dd))
    r, dd = r[order], dd[order]
    return r.astype(np.int32), (r + dd).astype(np.int32)
