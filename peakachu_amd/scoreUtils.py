"""Drop-in mirror of peakachu/scoreUtils.py for the MI355X path.

`Chromosome` keeps the reference's constructor, `getwindow`, `score` and
`writeBed` signatures and return values (peakachu/scoreUtils.py:9-135); the
per-candidate work (window gather, distance normalisation, blur, min-max,
forest, threshold) runs in the HIP library behind include/peakachu_hip.h.
"""
import numpy as np
from scipy import sparse

from . import _lib, utils
from .forest import as_flat_forest


# The device copy of a model (pk_forest: packed trees, rank tables, LDS images) is built
# once per model object and device, not once per chromosome: score_genome constructs a
# Chromosome per chromosome with the same model (peakachu/score_genome.py:58).
_FOREST_CACHE = []  # [(model, device, HipForest)], most recent last, at most 4


def _device_forest(model, device):
    for i, (m, d, hf) in enumerate(_FOREST_CACHE):
        if m is model and d == device and hf.h:
            _FOREST_CACHE.append(_FOREST_CACHE.pop(i))
            return hf
    hf = _lib.HipForest(as_flat_forest(model), device=device)
    _FOREST_CACHE.append((model, device, hf))
    while len(_FOREST_CACHE) > 4:
        _FOREST_CACHE.pop(0)[2].close()
    return hf


class Chromosome():
    def __init__(self, M, model, raw_M=None, weights=None,
                 lower=6, upper=300, cname='chrm', res=10000, width=5, device=0):
        # peakachu/scoreUtils.py:13-14
        lower = max(lower, width + 1)
        upper = min(upper, M.shape[0] - 2 * width)
        # expected values (peakachu/scoreUtils.py:16-24)
        # (diagonal means on the device, isotonic fit on the host)
        if weights is None:
            self.exp_arr = utils.calculate_expected(M, upper + 2 * width, raw=True, device=device)
            if M is raw_M:
                self.background = self.exp_arr
            else:
                self.background = utils.calculate_expected(raw_M, upper + 2 * width, raw=True,
                                                           device=device)
        else:
            self.exp_arr = utils.calculate_expected(M, upper + 2 * width, raw=False, device=device)
            self.background = self.exp_arr

        self.raw_M = raw_M
        self.weights = weights
        self._raw_is_M = M is raw_M
        # peakachu/scoreUtils.py:30-33
        self.M = utils.band_filter(M, width, upper)
        self.chromname = cname
        self.r = res
        self.w = width
        self.model = model
        self.lower, self.upper = lower, upper
        self.device = device
        self._hm = None
        self._hf = None
        self._cands = None
        self.get_candidate(lower, upper)

    # ----------------------------------------------------------- host side
    def get_candidate(self, lower, upper):
        """peakachu/scoreUtils.py:40-68.  On the device when the raw counts are
        integers (pk_candidates_create: the Poisson test runs against tables made
        with scipy); otherwise, or when a pixel falls inside the table's guard
        band, on the host with scipy directly."""
        self._cands = None
        got = self._candidates_on_device(lower, upper)  # device errors propagate
        if got is None:
            self.ridx, self.cidx = utils.candidates(self.raw_M, self.background, self.weights,
                                                    lower, upper)
        else:
            self._cands = got
            x, y = got.coords()
            self.ridx, self.cidx = x.astype(np.int64), y.astype(np.int64)
        self._cands_key = (self.ridx, self.cidx)

    def _candidates_on_device(self, lower, upper):
        raw = utils.canonical_csr(self.raw_M)
        n = raw.shape[0]
        hi = min(int(upper), self.background.size - 1, n - 1)
        if hi < lower or raw.nnz == 0:
            return None
        if not np.all(raw.data[np.isfinite(raw.data)] == np.floor(raw.data[np.isfinite(raw.data)])):
            return None  # the tables assume integer counts
        if self.weights is None and self._raw_is_M:
            rawm = self._matrix()  # the scoring band already holds these counts
            own = False
        else:
            rawm = _lib.HipMatrix(raw.indptr, raw.indices, raw.data, n, self.background,
                                  lower, hi, device=self.device)
            own = True
        try:
            bg = np.ascontiguousarray(self.background[:hi + 1], np.float64)
            if self.weights is None:
                kstar = utils._poisson_count_thresholds(bg)
                cands, amb = _lib.HipCands.from_band(rawm, lower, hi, bg, kstar=kstar)
            else:
                kmax = int(np.nanmax(raw.data)) if raw.nnz else 0
                cands, amb = _lib.HipCands.from_band(rawm, lower, hi, bg,
                                                     weights=np.asarray(self.weights, np.float64),
                                                     mustar=utils.poisson_mu_thresholds(kmax))
        finally:
            if own:
                rawm.close()
        if amb:
            cands.close()
            return None
        return cands

    # --------------------------------------------------------- device side
    def _matrix(self):
        if self._hm is None:
            M = utils.canonical_csr(self.M)
            dlo, dhi = -2 * self.w + 1, self.upper + 2 * self.w - 1
            self._hm = _lib.HipMatrix(M.indptr, M.indices, M.data, M.shape[0], self.exp_arr,
                                      dlo, max(dhi, dlo), device=self.device)
        return self._hm

    def _forest(self):
        if self._hf is None:
            self._hf = _device_forest(self.model, self.device)
        return self._hf

    def getwindow(self, coords):
        """peakachu/scoreUtils.py:70-93 -> (fea [N', F] float64, clist [N', 2])."""
        coords = np.r_[coords]
        if coords.size == 0:
            return np.r_[[]], np.r_[[]]
        xi, yi = coords[:, 0].astype(np.int64), coords[:, 1].astype(np.int64)
        if np.any(xi > yi):
            raise ValueError("peakachu_amd.getwindow needs upper-triangle coords (x <= y)")
        fea, _, keep = self._matrix().extract(self.w, xi, yi)
        if keep.size == 0:
            return np.r_[[]], np.r_[[]]
        clist = np.stack([xi[keep], yi[keep]], axis=1)
        return fea, clist

    def score(self, thre=0.5):
        """peakachu/scoreUtils.py:95-125."""
        print('scoring matrix {}'.format(self.chromname))
        print('number of candidates {}'.format(self.ridx.size))
        # the device-resident candidate list is reused unless ridx / cidx were replaced
        same = (self._cands is not None and self._cands_key[0] is self.ridx
                and self._cands_key[1] is self.cidx)
        cd = self._cands if same else _lib.HipCands(self.ridx, self.cidx, device=self.device)
        # exact early termination, a property of this candidate list (no process-wide
        # option is touched): a candidate stops once its sum can no longer exceed thre*T;
        # the reported pixels are identical (tests/test_gpu_fullsize.py)
        cd.set_prune(True)
        cd.run(self._matrix(), self._forest(), self.w, thre, batch=100000)
        ri, ci, prob_pool, signal = cd.fetch()
        ri = ri.astype(int)
        ci = ci.astype(int)
        result = sparse.csr_matrix((prob_pool, (ri, ci)), shape=self.M.shape)
        if ri.size > 0:
            self.M = sparse.csr_matrix((signal, (ri, ci)), shape=self.M.shape)
        else:
            self.M = result
        self._hm = None  # self.M changed, as in the reference
        return result, self.M

    def writeBed(self, outfil, prob_csr, raw_csr):
        """peakachu/scoreUtils.py:127-135: 8 tab-separated columns, appended."""
        write_bedpe(outfil, self.chromname, self.r, prob_csr, raw_csr)


def write_bedpe(outfil, chromname, res, prob_csr, raw_csr):
    with open(outfil, 'a') as out:
        r, c = prob_csr.nonzero()
        if r.size == 0:
            return
        p = np.asarray(prob_csr[r, c]).ravel()
        s = np.asarray(raw_csr[r, c]).ravel()
        lines = []
        for i in range(r.size):
            ri, ci = int(r[i]), int(c[i])
            lines.append('\t'.join((chromname, str(ri * res), str((ri + 1) * res),
                                    chromname, str(ci * res), str((ci + 1) * res),
                                    repr(float(p[i])), repr(float(s[i])))))
        out.write('\n'.join(lines) + '\n')
