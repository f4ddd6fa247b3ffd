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


# ---------------------------------------------------------------- genome-shaped stand-ins
# hg19 chromosome lengths (UCSC hg19.chrom.sizes): the assembly of the reference's example map
# (README.md:57, Rao 2014 GM12878) -- BASELINE.json configs[0] / configs[2] run on it.  The map
# itself cannot be had offline; these are the SHAPES a stand-in is synthesised in.
HG19_CHROMS = (
    ("chr1", 249250621), ("chr2", 243199373), ("chr3", 198022430), ("chr4", 191154276),
    ("chr5", 180915260), ("chr6", 171115067), ("chr7", 159138663), ("chr8", 146364022),
    ("chr9", 141213431), ("chr10", 135534747), ("chr11", 135006516), ("chr12", 133851895),
    ("chr13", 115169878), ("chr14", 107349540), ("chr15", 102531392), ("chr16", 90354753),
    ("chr17", 81195210), ("chr18", 78077248), ("chr19", 59128983), ("chr20", 63025520),
    ("chr21", 48129895), ("chr22", 51304566), ("chrX", 155270560), ("chrY", 59373566),
    ("chrM", 16571))


def band_counts(n, band, seed=0, loops=None, bump=30):
    """Upper-band counts of one synthetic chromosome as a dense [n, band+1] int32 array:
    cnt[i, d] = count(i, i+d) ~ Poisson(200/(1+d)**0.9 + 0.3) (synth_band's law), 0 where
    i+d >= n, plus `loops` 3x3 bumps.  Row-major order IS the (bin1, bin2) order of a .cool's
    pixel table, so a genome of these is written / converted without a sort."""
    rng = np.random.default_rng(seed)
    band = int(min(band, max(n - 1, 0)))
    d = np.arange(band + 1)
    lam = 200.0 / (1.0 + d) ** 0.9 + 0.3
    cnt = rng.poisson(np.broadcast_to(lam, (n, band + 1))).astype(np.int32)
    if loops is None:
        loops = max(1, n // 40)
    if n > band + 6 and band > 12:
        la = rng.integers(2, n - band - 3, size=loops)
        ld = rng.integers(8, band - 2, size=loops)
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                np.add.at(cnt, (la + di, ld + dj - di), bump)
    i = np.arange(n)[:, None]
    cnt[i + d[None, :] >= n] = 0
    return cnt


def band_counts_to_pixels(cnt, offset=0):
    """(bin1_id, bin2_id, count) of the non-zero cells, sorted by (bin1, bin2)."""
    i, d = np.nonzero(cnt)
    return (i + offset).astype(np.int64), (i + d + offset).astype(np.int64), cnt[i, d]


def band_counts_to_csr(cnt, dtype=np.float64):
    """The symmetric canonical CSR of the chromosome (what cooler's matrix(balance=False).fetch
    gives, as utils.tocsr leaves it), assembled from the band without a sort: row i = the cells
    (i, i-band .. i-1) taken from the rows above, then (i, i .. i+band)."""
    n, b1 = cnt.shape
    band = b1 - 1
    # diagonal-major first (contiguous row copies), then one transposition to row-major
    cnt_t = np.ascontiguousarray(cnt.T)
    full_t = np.zeros((2 * band + 1, n), cnt.dtype)
    full_t[band:] = cnt_t
    for d in range(1, band + 1):  # M[i, i-d] = cnt[i-d, d]
        full_t[band - d, d:] = cnt_t[d, :n - d]
    full = np.ascontiguousarray(full_t.T)
    mask = full != 0
    cols = np.arange(n, dtype=np.int32)[:, None] + np.arange(-band, band + 1, dtype=np.int32)[None, :]
    indptr = np.zeros(n + 1, np.int64)
    np.cumsum(mask.sum(axis=1), out=indptr[1:])
    M = sparse.csr_matrix((full[mask].astype(dtype), cols[mask], indptr.astype(np.int32)), shape=(n, n))
    M.has_sorted_indices = True
    M.has_canonical_format = True
    return M
