"""Host-side helpers around the hot path (numpy / scipy).

`calculate_expected` and `candidates` restate peakachu/utils.py:139-178 and
peakachu/scoreUtils.py:40-68 (SURVEY.md §8f "next" rows): they produce the
inputs of the GPU path (exp_arr, the candidate list) once per chromosome.
They are vectorised over the band instead of looping `M.diagonal(i)`, but
feed numpy / scipy / sklearn the same operands in the same order, so their
results are bit-identical to the reference's (tests/test_host_golden.py).
"""
import numpy as np
from scipy import sparse


class _LazyStats:
    """scipy.stats on first use: its import costs a quarter of a second, which the scoring commands
    would otherwise pay before the first chromosome is even read (score_genome.warm_imports starts
    it on a thread instead)."""

    def __getattr__(self, name):
        import scipy.stats
        return getattr(scipy.stats, name)


stats = _LazyStats()


def tocsr(X):
    """peakachu/utils.py:10-15.  A matrix from this package's own `.cool` reader arrives in
    canonical order with its row pointer attached (cool.sparse_coo, which has CHECKED that order):
    the CSR is then assembled from the parts -- as long as the COO is still the object the reader
    made (same row / col / data arrays, same shape).  The CSR's `indices` ARE the reader's cached,
    read-only column array (a chromosome's balanced and raw matrices share it): an in-place
    structural change of the result raises "read-only" instead of reaching the other matrix; the
    scoring path never makes one (the reference's CSR is a private copy)."""
    parts = getattr(X, "_pk_csr_parts", None)
    if (parts is not None and len(parts) == 5 and parts[1] is X.col and parts[2] is X.data
            and parts[3] is X.row and tuple(parts[4]) == tuple(X.shape)
            and parts[0].size == X.shape[0] + 1 and int(parts[0][-1]) == X.col.size):
        indptr, col, data = parts[:3]
        out = sparse.csr_matrix((data.astype(float, copy=False), col, indptr), shape=X.shape)
        out.has_sorted_indices = True
        out.has_canonical_format = True
        return out
    return sparse.csr_matrix((X.data, (X.row, X.col)), shape=X.shape, dtype=float)


def mirror_upper(i, j, v, n):
    """(row, col, value, indptr) of the symmetric matrix whose upper triangle is the pixel list
    (i, j, v) -- sorted by (i, j), j >= i, no duplicates: a conforming .cool's -- in CANONICAL
    order (row-major, columns ascending) without a sort of the list: the upper part U is a
    canonical CSR as it stands, the lower part the transpose of U without its diagonal (scipy's
    csc -> csr, a counting sort), rows merged by scipy's sorted-row addition (which drops explicit
    zeros: a table that holds any takes the numpy route)."""
    if v.size and not np.all(v != 0):
        return _mirror_by_scatter(i, j, v, n)
    indptr_u = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(i, minlength=n), out=indptr_u[1:])
    U = sparse.csr_matrix((v, j, indptr_u.astype(np.int32)), shape=(n, n))
    off_diag = i != j
    indptr_s = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(i[off_diag], minlength=n), out=indptr_s[1:])
    Us = sparse.csr_matrix((v[off_diag], j[off_diag], indptr_s.astype(np.int32)), shape=(n, n))
    M = Us.T.tocsr() + U
    row = np.repeat(np.arange(n, dtype=np.int32), np.diff(M.indptr))
    return row, M.indices, M.data, M.indptr


def _mirror_by_scatter(i, j, v, n):
    """The same canonical (row, col, value, indptr) with numpy alone: row r = its lower entries
    (the pixels of column r above the diagonal, ordered by their row = a stable sort by
    column) followed by its upper entries (already in order)."""
    strict = np.flatnonzero(i != j)
    lo_order = strict[np.argsort(j[strict], kind="stable")]
    cnt_u = np.bincount(i, minlength=n)
    cnt_l = np.bincount(j[strict], minlength=n)
    indptr = np.zeros(n + 1, np.int64)
    np.cumsum(cnt_u + cnt_l, out=indptr[1:])
    start_l = np.cumsum(cnt_l) - cnt_l
    start_u = np.cumsum(cnt_u) - cnt_u
    total = int(indptr[-1])
    row = np.empty(total, np.int32)
    col = np.empty(total, np.int32)
    val = np.empty(total, v.dtype)
    rl = j[lo_order]                       # row of a lower entry = the pixel's column
    pos_l = indptr[:-1][rl] + (np.arange(rl.size, dtype=np.int64) - start_l[rl])
    row[pos_l], col[pos_l], val[pos_l] = rl, i[lo_order], v[lo_order]
    pos_u = indptr[:-1][i] + cnt_l[i] + (np.arange(i.size, dtype=np.int64) - start_u[i])
    row[pos_u], col[pos_u], val[pos_u] = i, j, v
    return row, col, val, indptr.astype(np.int32)


def is_canonical(row, col):
    """Rows in order, columns strictly ascending inside a row (no duplicates)."""
    if col.size < 2:
        return True
    return bool(np.all(row[1:] >= row[:-1])) and bool(np.all((col[1:] > col[:-1]) | (row[1:] != row[:-1])))


class UpperPixels:
    """A chromosome as a contact-map file stores it: the upper triangle, pixels sorted by
    (bin1, bin2).  `indptr[n+1]` = first pixel of each bin1, `cols` = bin2 relative to the
    chromosome's first bin -- entries with cols >= n are pixels of OTHER chromosomes that share
    the rows (a .cool keeps a row's trans pixels behind its cis ones) and are not part of the
    matrix -- `counts` = the stored counts.  The reference has cooler mirror (and balance) these
    on the host (peakachu/score_genome.py:55-57); `Chromosome.from_upper` sends them to the
    device as they are, and `symmetric()` makes the host matrix for whoever asks."""

    def __init__(self, n, indptr, cols, counts):
        self.n = int(n)
        self.indptr, self.cols, self.counts = indptr, cols, counts
        self.shape = (self.n, self.n)

    @property
    def nnz(self):
        return int(self.indptr[-1])

    def cis(self):
        """(row, col, count) of the pixels inside the chromosome."""
        i = np.repeat(np.arange(self.n, dtype=np.int32), np.diff(self.indptr))
        j, v = self.cols, self.counts
        inside = j < self.n
        if not inside.all():
            i, j, v = i[inside], j[inside], v[inside]
        return i, j, v

    def symmetric(self, bias=None):
        """The CSR the reference's driver would hold: cooler's mirrored matrix through
        utils.tocsr; with `bias` the values are (bias[row] * bias[col]) * count."""
        i, j, v = self.cis()
        if is_canonical(i, j):
            row, col, data, indptr = mirror_upper(i, j, v, self.n)
        else:  # not a conforming table: sort and sum like the reference's conversion
            off = i != j
            coo = sparse.coo_matrix((np.concatenate([v, v[off]]), (np.concatenate([i, j[off]]),
                                                                  np.concatenate([j, i[off]]))), shape=self.shape)
            M = sparse.csr_matrix(coo, dtype=float)
            M.sort_indices()
            row = np.repeat(np.arange(self.n, dtype=np.int32), np.diff(M.indptr))
            col, data, indptr = M.indices, M.data, M.indptr
        if bias is not None:
            f = np.repeat(np.asarray(bias, np.float64), np.diff(indptr))
            f *= np.take(bias, col)
            f *= data
            data = f
        out = sparse.csr_matrix((np.asarray(data, np.float64), col, indptr), shape=self.shape)
        out.has_sorted_indices = True
        out.has_canonical_format = True
        return out


def canonical_csr(M):
    M = sparse.csr_matrix(M, dtype=np.float64)
    if not M.has_canonical_format:
        M = M.copy()
        M.sum_duplicates()
    M.sort_indices()
    return M


def band_filter(M, width, upper):
    """peakachu/scoreUtils.py:30-33: keep finite entries with
    -2w < col-row < upper+2w (both strict).  Works on the CSR arrays directly
    (no COO round trip): the result is canonical because the input is."""
    M = canonical_csr(M)
    n = M.shape[0]
    indices = M.indices.astype(np.int32, copy=False)
    R = np.repeat(np.arange(n, dtype=np.int32), np.diff(M.indptr))
    k = indices - R
    data = M.data
    ok = (data != 0) & np.isfinite(data) & (k > -2 * width) & (k < upper + 2 * width)
    indptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(R[ok], minlength=n), out=indptr[1:])
    out = sparse.csr_matrix((data[ok], indices[ok], indptr.astype(np.int32)), shape=M.shape)
    out.has_sorted_indices = True
    return out


def _dense_diagonals(indptr, indices, data, keep, n, maxdis):
    """D[k, r] = M[r, r+k] for 0 <= k <= maxdis over the entries flagged in
    `keep` (upper band, dense; one gather + one scatter, int32 arithmetic)."""
    R = np.repeat(np.arange(n, dtype=np.int32), np.diff(indptr))
    k = indices - R
    sel = np.flatnonzero(keep & (k >= 0) & (k <= maxdis))
    D = np.zeros((maxdis + 1) * n, np.float64)
    D[k[sel].astype(np.int64) * n + R[sel]] = data[sel]
    return D.reshape(maxdis + 1, n)


def calculate_expected(M, maxdis, raw=False, device=None):
    """peakachu/utils.py:139-178: mean of each diagonal over valid bins, then
    a non-increasing isotonic fit (sklearn IsotonicRegression).

    With `device` set, the diagonal means come from the GPU (pk_expected_means:
    dense diagonals built on the device, summed in numpy's own order, so they are
    bit-identical to the host path below); validity flags and the isotonic fit
    stay on the host."""
    M = canonical_csr(M)
    n = M.shape[0]
    indices = M.indices.astype(np.int32, copy=False)
    data = M.data
    nz = data != 0  # M.nonzero() semantics
    finite = np.isfinite(data) & nz
    if raw:
        # column sums of the finite entries; only their sign is used
        marg = np.bincount(indices[finite], weights=data[finite], minlength=n)
        valid_cols = marg > 0
        keep = finite
    else:
        # bins that occur as a row or a column of a finite entry (utils.py:150-155)
        rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(M.indptr))
        valid_cols = np.zeros(n, dtype=bool)
        valid_cols[rows[finite]] = True
        valid_cols[indices[finite]] = True
        keep = nz  # the reference keeps NaN entries in the diagonals here
    maxdis = int(maxdis)
    top = min(maxdis, n - 1)
    exp_arr = np.zeros(maxdis + 1)
    if device is not None:
        exp_arr[:top + 1] = _diagonal_means_device(M, keep, n, top, valid_cols, device)
    else:
        D = _dense_diagonals(M.indptr, indices, data, keep, n, top)
        for i in range(top + 1):
            valid = valid_cols if i == 0 else valid_cols[:-i] * valid_cols[i:]
            diag = D[i, :n - i][valid]
            if diag.size > 10:
                exp_arr[i] = diag.mean()
    return isotonic_expected(exp_arr)


def _pava_increasing(y):
    """Pool-adjacent-violators on float64 `y` with unit weights, in place: the steps and the
    arithmetic of scikit-learn's `_inplace_contiguous_isotonic_regression` (block sums grow
    term by term from the left, a block's value is sum / count, backtrack after every merge)."""
    n = y.size
    w = np.ones(n, np.float64)
    target = np.arange(n)
    i = 0
    while i < n:
        k = target[i] + 1
        if k == n:
            break
        if y[i] < y[k]:
            i = k
            continue
        sum_wy = w[i] * y[i]
        sum_w = w[i]
        while True:
            prev_y = y[k]
            sum_wy += w[k] * y[k]
            sum_w += w[k]
            k = target[k] + 1
            if k == n or prev_y < y[k]:
                y[i] = sum_wy / sum_w
                w[i] = sum_w
                target[i] = k - 1
                target[k - 1] = i
                if i > 0:
                    i = target[i - 1]
                break
    i = 0
    while i < n:
        k = target[i] + 1
        y[i + 1:k] = y[i]
        i = k
    return y


def isotonic_expected(exp_arr):
    """peakachu/utils.py:173-178: `IsotonicRegression(increasing=False, out_of_bounds='clip')`
    fitted through the positive diagonal means and evaluated at every distance (tiny: <=
    upper + 2w + 1 points).  The fit's last bits depend on the installation: scikit-learn up
    to 1.3 (the reference's pin is 1.1.2) pools with its own Cython PAVA, later releases hand
    the job to scipy >= 1.12's, which sums in another order.  So, like the reference, this
    calls the scikit-learn that is installed; without scikit-learn the restatement of the
    pinned-era algorithm below is used."""
    try:
        from sklearn.isotonic import IsotonicRegression
    except ImportError:
        return isotonic_expected_restated(exp_arr)
    IR = IsotonicRegression(increasing=False, out_of_bounds="clip")
    _d = np.where(exp_arr > 0)[0]
    IR.fit(_d, exp_arr[_d])
    return IR.predict(list(range(exp_arr.size)))


def isotonic_expected_restated(exp_arr):
    """The same without scikit-learn, as scikit-learn <= 1.3 computes it (bit for bit:
    tests/test_host_golden.py against the Cython routine itself and against curves fitted by
    scikit-learn 0.24.2):
      * a decreasing fit = the increasing PAVA on the reversed sequence;
      * fit() drops every point whose value equals both neighbours' (flat interior);
      * predict() clips the abscissae to the fitted range and interpolates linearly --
        scipy's interp1d hands float64 data to numpy.interp, which is called here directly."""
    exp_arr = np.asarray(exp_arr, np.float64)
    d = np.flatnonzero(exp_arr > 0)
    if d.size == 0:
        raise ValueError("Found array with 0 sample(s) (shape=(0,)) while a minimum of 1 is required.")
    x = d.astype(np.float64)
    y = _pava_increasing(exp_arr[d][::-1].copy())[::-1].copy()
    t = np.clip(np.arange(exp_arr.size, dtype=np.float64), x[0], x[-1])
    if y.size == 1:
        return np.repeat(y, t.size)
    keep = np.ones(y.size, bool)
    keep[1:-1] = (y[1:-1] != y[:-2]) | (y[1:-1] != y[2:])
    return np.interp(t, x[keep], y[keep])


def _diagonal_means_device(M, keep, n, top, valid_cols, device):
    from . import _lib
    if keep.all():
        indptr, indices, data = M.indptr, M.indices, M.data
    else:  # drop the entries calculate_expected ignores (non-finite ones in raw mode)
        rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(M.indptr))
        indptr = np.zeros(n + 1, np.int64)
        np.cumsum(np.bincount(rows[keep], minlength=n), out=indptr[1:])
        indices, data = M.indices[keep], M.data[keep]
    hm = _lib.HipMatrix(indptr, indices, data, n, np.ones(1), 0, top, device=device)
    try:
        return hm.expected_means(top, valid_cols)
    finally:
        hm.close()


def _poisson_count_thresholds(mu):
    """For each expected count mu[i] > 0 the smallest integer k >= 1 with
    scipy.stats.poisson.sf(k, mu[i]) < 0.01, found with scipy's own sf so the
    decision is the reference's (sf is non-increasing in k).  One sf call evaluates the
    counts 1..mu + 10 sqrt(mu) + 30 of every diagonal at once (a chromosome has ~300
    diagonals; one call per diagonal cost 7-30 ms per chromosome)."""
    mu = np.asarray(mu, np.float64)
    out = np.full(mu.size, np.iinfo(np.int64).max, np.int64)
    pending = np.flatnonzero(np.isfinite(mu) & (mu > 0))
    hi = (mu[pending] + 10.0 * np.sqrt(mu[pending]) + 30).astype(np.int64)
    while pending.size:
        # bounded work per call: diagonals in groups of at most ~4 M (count, mu) pairs
        cut = max(1, int(np.searchsorted(np.cumsum(hi), 1 << 22, side="right")))
        grp, glen = pending[:cut], hi[:cut]
        starts = np.cumsum(glen) - glen
        ks = (np.arange(int(glen.sum()), dtype=np.int64) - np.repeat(starts, glen) + 1).astype(np.float64)
        with np.errstate(all="ignore"):
            p = stats.poisson.sf(ks, np.repeat(mu[grp], glen))
        first = np.minimum.reduceat(np.where(p < 0.01, ks, np.inf), starts)
        found = np.isfinite(first)
        out[grp[found]] = first[found].astype(np.int64)
        again = glen[~found] * 2   # no such count below the bound: look further
        ok = again <= 1 << 24
        pending = np.concatenate([grp[~found][ok], pending[cut:]])
        hi = np.concatenate([again[ok], hi[cut:]])
    return out


_MUSTAR = np.zeros(0)


def poisson_mu_thresholds(kmax):
    """mustar[k] (k = 0..kmax): the smallest float64 mu for which scipy's
    poisson.sf(k, mu) is NOT < 0.01, found by bisection on scipy's own sf (it is
    increasing in mu), so that  sf(k, mu) < 0.01  <=>  mu < mustar[k]  up to the
    last-bit wiggle the GPU kernel's 1e-9 guard band leaves to scipy.  Cached."""
    global _MUSTAR
    kmax = int(kmax)
    if _MUSTAR.size > kmax:
        return _MUSTAR[:kmax + 1]
    k = np.arange(kmax + 1, dtype=np.float64)
    lo = np.zeros(kmax + 1)
    hi = k + 12.0 * np.sqrt(k + 1.0) + 20.0
    with np.errstate(all="ignore"):
        assert np.all(stats.poisson.sf(k, hi) >= 0.01)
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            below = stats.poisson.sf(k, mid) < 0.01
            lo = np.where(below, mid, lo)
            hi = np.where(below, hi, mid)
            if np.all((hi - lo) <= np.spacing(lo)):
                break
    _MUSTAR = hi
    return _MUSTAR


def candidates(raw_M, background, weights, lower, upper):
    """peakachu/scoreUtils.py:40-68: Poisson survival p-value of every
    non-zero raw pixel on diagonals lower..upper against the expected count
    (divided by the two bin weights in balanced mode); keep p < 0.01.
    Order: diagonal ascending, then row ascending."""
    raw_M = canonical_csr(raw_M)
    n = raw_M.shape[0]
    R = np.repeat(np.arange(n, dtype=np.int64), np.diff(raw_M.indptr))
    C = raw_M.indices.astype(np.int64)
    data = raw_M.data
    k = C - R
    e = np.asarray(background, np.float64)
    hi = min(int(upper), e.size - 1, n - 1)
    ok = (k >= int(lower)) & (k <= hi) & (data > 0)
    R, k, data = R[ok], k[ok], data[ok]
    ok = e[k] > 0
    R, k, data = R[ok], k[ok], data[ok]
    if weights is None and np.all(data == np.floor(data)):
        # raw mode: the expected count is one number per diagonal, so
        # sf(count, mu_d) < 0.01  <=>  count >= kstar[d]
        kstar = _poisson_count_thresholds(e[:hi + 1])
        mask = data >= kstar[k]
    else:
        if weights is None:
            mu = np.ones(R.size, dtype=float) * e[k]
        else:
            w = np.asarray(weights, np.float64)
            mu = np.ones(R.size, dtype=float) * e[k] / (w[R] * w[R + k])
        with np.errstate(all="ignore"):
            p = stats.poisson.sf(data, mu)
        mask = np.isfinite(p) & (p < 0.01)
    x, kk = R[mask], k[mask]
    order = np.lexsort((x, kk))
    x, kk = x[order], kk[order]
    return x.astype(np.int64), (x + kk).astype(np.int64)
