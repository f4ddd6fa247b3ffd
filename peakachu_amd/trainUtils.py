"""Mirror of the feature builder in peakachu/trainUtils.py:12-44.

Only `buildmatrix` -- the training-side twin of Chromosome.getwindow -- is
on the hot path; forest training itself is out of scope.
"""
import numpy as np

from . import _lib, utils


def buildmatrix(Matrix, coords, w=5, device=0):
    """peakachu/trainUtils.py:12-44 -> list of [F] float64 feature vectors,
    or None when fewer than 10 coordinates pass the pre-filter."""
    coords = np.r_[coords]
    xi, yi = coords[:, 0].astype(np.int64), coords[:, 1].astype(np.int64)
    mask = (xi - w >= 0) & (yi + w + 1 <= Matrix.shape[0]) & (yi - xi > w)
    xi, yi = xi[mask], yi[mask]
    if xi.size < 10:
        return
    maxdis = int(np.abs(xi - yi).max()) + 2 * w
    exp_arr = utils.calculate_expected(Matrix, maxdis)
    M = utils.canonical_csr(Matrix)
    d = yi - xi
    hm = _lib.HipMatrix(M.indptr, M.indices, M.data, M.shape[0], exp_arr,
                        int(d.min()) - 2 * w, int(d.max()) + 2 * w, device=device)
    try:
        fea, _, _ = hm.extract(w, xi, yi)
    finally:
        hm.close()
    return [row for row in fea]
