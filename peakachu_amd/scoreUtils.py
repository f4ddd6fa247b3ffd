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


class Chromosome():
    def __init__(self, M, model, raw_M=None, weights=None,
                 lower=6, upper=300, cname='chrm', res=10000, width=5, device=0):
        # peakachu/scoreUtils.py:13-14
        lower = max(lower, width + 1)
        upper = min(upper, M.shape[0] - 2 * width)
        # expected values (peakachu/scoreUtils.py:16-24)
        if weights is None:
            self.exp_arr = utils.calculate_expected(M, upper + 2 * width, raw=True)
            if M is raw_M:
                self.background = self.exp_arr
            else:
                self.background = utils.calculate_expected(raw_M, upper + 2 * width, raw=True)
        else:
            self.exp_arr = utils.calculate_expected(M, upper + 2 * width, raw=False)
            self.background = self.exp_arr

        self.raw_M = raw_M
        self.weights = weights
        # peakachu/scoreUtils.py:30-33
        self.M = utils.band_filter(M, width, upper)
        self.get_candidate(lower, upper)
        self.chromname = cname
        self.r = res
        self.w = width
        self.model = model
        self.lower, self.upper = lower, upper
        self.device = device
        self._hm = None
        self._hf = None

    # ----------------------------------------------------------- host side
    def get_candidate(self, lower, upper):
        """peakachu/scoreUtils.py:40-68."""
        self.ridx, self.cidx = utils.candidates(self.raw_M, self.background, self.weights,
                                                lower, upper)

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
            self._hf = _lib.HipForest(as_flat_forest(self.model), device=self.device)
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
        ri, ci, prob_pool, signal = self._matrix().score(self._forest(), self.w, thre,
                                                         self.ridx, self.cidx, batch=100000)
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
