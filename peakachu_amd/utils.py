"""Host-side helpers around the hot path (numpy / scipy).

`calculate_expected` and `candidates` restate peakachu/utils.py:139-178 and
peakachu/scoreUtils.py:40-68 (SURVEY.md §8f "next" rows): they produce the
inputs of the GPU path (exp_arr, the candidate list) once per chromosome.
They are vectorised over the band instead of looping `M.diagonal(i)`, but
feed numpy / scipy / sklearn the same operands in the same order, so their
results are bit-identical to the reference's (tests/test_host_golden.py).
"""
import numpy as np
from scipy import sparse, stats


def tocsr(X):
    """peakachu/utils.py:10-15."""
    return sparse.csr_matrix((X.data, (X.row, X.col)), shape=X.shape, dtype=float)


def canonical_csr(M):
    M = sparse.csr_matrix(M, dtype=np.float64)
    if not M.has_canonical_format:
        M = M.copy()
        M.sum_duplicates()
    M.sort_indices()
    return M


def band_filter(M, width, upper):
    """peakachu/scoreUtils.py:30-33: keep finite entries with
    -2w < col-row < upper+2w (both strict)."""
    coo = sparse.csr_matrix(M).tocoo()
    R, C, data = coo.row.astype(np.int64), coo.col.astype(np.int64), coo.data
    ok = (data != 0) & np.isfinite(data) & (C - R > -2 * width) & (C - R < upper + 2 * width)
    out = sparse.csr_matrix((data[ok], (R[ok], C[ok])), shape=M.shape, dtype=np.float64)
    out.sum_duplicates()
    out.sort_indices()
    return out


def _dense_diagonals(R, C, data, n, maxdis):
    """D[k, r] = M[r, r+k] for 0 <= k <= maxdis (upper band, dense)."""
    k = C - R
    ok = (k >= 0) & (k <= maxdis)
    D = np.zeros((maxdis + 1, n), np.float64)
    D[k[ok], R[ok]] = data[ok]
    return D


def calculate_expected(M, maxdis, raw=False):
    """peakachu/utils.py:139-178: mean of each diagonal over valid bins, then
    a non-increasing isotonic fit (sklearn IsotonicRegression)."""
    from sklearn.isotonic import IsotonicRegression

    M = canonical_csr(M)
    n = M.shape[0]
    coo = M.tocoo()
    R, C, data = coo.row.astype(np.int64), coo.col.astype(np.int64), coo.data
    nz = data != 0  # M.nonzero() semantics
    R, C, data = R[nz], C[nz], data[nz]
    finite = np.isfinite(data)
    if raw:
        R, C, data = R[finite], C[finite], data[finite]
        marg = np.zeros(n)
        np.add.at(marg, C, data)
        valid_cols = marg > 0
    else:
        valid_cols = np.zeros(n, dtype=bool)
        valid_cols[R[finite]] = True
        valid_cols[C[finite]] = True
    maxdis = int(maxdis)
    D = _dense_diagonals(R, C, data, n, min(maxdis, n - 1))
    exp_arr = np.zeros(maxdis + 1)
    for i in range(min(maxdis, n - 1) + 1):
        valid = valid_cols if i == 0 else valid_cols[:-i] * valid_cols[i:]
        diag = D[i, :n - i][valid]
        if diag.size > 10:
            exp_arr[i] = diag.mean()
    IR = IsotonicRegression(increasing=False, out_of_bounds="clip")
    _d = np.where(exp_arr > 0)[0]
    IR.fit(_d, exp_arr[_d])
    return IR.predict(list(range(maxdis + 1)))


def candidates(raw_M, background, weights, lower, upper):
    """peakachu/scoreUtils.py:40-68: Poisson survival p-value of every
    non-zero raw pixel on diagonals lower..upper against the expected count
    (divided by the two bin weights in balanced mode); keep p < 0.01.
    Order: diagonal ascending, then row ascending."""
    raw_M = canonical_csr(raw_M)
    n = raw_M.shape[0]
    coo = raw_M.tocoo()
    R, C, data = coo.row.astype(np.int64), coo.col.astype(np.int64), coo.data
    k = C - R
    e = np.asarray(background, np.float64)
    hi = min(int(upper), e.size - 1, n - 1)
    ok = (k >= int(lower)) & (k <= hi)
    R, k, data = R[ok], k[ok], data[ok]
    ok = e[k] > 0
    R, k, data = R[ok], k[ok], data[ok]
    order = np.lexsort((R, k))
    R, k, data = R[order], k[order], data[order]
    if weights is None:
        mu = np.ones(R.size, dtype=float) * e[k]
    else:
        w = np.asarray(weights, np.float64)
        mu = np.ones(R.size, dtype=float) * e[k] / (w[R] * w[R + k])
    with np.errstate(all="ignore"):
        p = stats.poisson.sf(data, mu)
    mask = (data > 0) & np.isfinite(p)
    mask &= p < 0.01
    x = R[mask]
    return x.astype(np.int64), (x + k[mask]).astype(np.int64)
