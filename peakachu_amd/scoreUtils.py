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
from .stagetime import stage


# The device copy of a model (pk_forest: packed trees, rank tables, LDS images) is built
# once per model object and device, not once per chromosome: score_genome constructs a
# Chromosome per chromosome with the same model (peakachu/score_genome.py:58).
_FOREST_CACHE = []  # [(model, device, HipForest)], most recent last, at most 4


def _device_forest(model, device):
    for i, (m, d, hf) in enumerate(_FOREST_CACHE):
        if m is model and d == device and hf.h:
            _FOREST_CACHE.append(_FOREST_CACHE.pop(i))
            return hf
    with stage("forest: upload + device image (once per model)"):
        hf = _lib.HipForest(as_flat_forest(model), device=device)
    _FOREST_CACHE.append((model, device, hf))
    while len(_FOREST_CACHE) > 4:
        # only the cache's reference goes: a Chromosome that still holds the evicted forest
        # keeps it alive, and HipForest.__del__ frees the device copy with the last holder
        _FOREST_CACHE.pop(0)
    return hf


class Chromosome():
    def __init__(self, M, model, raw_M=None, weights=None,
                 lower=6, upper=300, cname='chrm', res=10000, width=5, device=0):
        self._common(M.shape, model, weights, lower, upper, cname, res, width, device)
        self._raw_val = raw_M
        self._raw_is_M = M is raw_M
        self._M_src, self._M_val = M, None   # self.M (the band-filtered CSR) is made on demand
        self._pixels = self._bias = None
        self._prepare()

    @classmethod
    def from_upper(cls, pixels, model, bias=None, weights=None,
                   lower=6, upper=300, cname='chrm', res=10000, width=5, device=0):
        """The same object from a chromosome AS THE CONTACT-MAP FILE STORES IT (utils.UpperPixels:
        the upper triangle, pixels sorted by bin1, bin2) instead of the mirrored, balanced
        matrices the reference has cooler make on the host first (peakachu/score_genome.py:55-57):
        one upload of the pixel table, mirrored and balanced on the device
        (pk_csr_upload_upper / pk_csr_view).  `bias` = the vector cooler multiplies the counts with
        (None: raw mode, M is raw_M), `weights` = the bin-weight column handed to get_candidate.
        `M` / `raw_M` are made on the host only when somebody reads them."""
        self = cls.__new__(cls)
        self._common(pixels.shape, model, weights, lower, upper, cname, res, width, device)
        self._pixels, self._bias = pixels, (None if bias is None else np.ascontiguousarray(bias, np.float64))
        self._raw_is_M = bias is None
        self._raw_val = self._M_src = self._M_val = None
        self._prepare()
        return self

    def _common(self, shape, model, weights, lower, upper, cname, res, width, device):
        # peakachu/scoreUtils.py:13-14
        self.lower = max(lower, width + 1)
        self.upper = min(upper, shape[0] - 2 * width)
        self.weights = weights
        self.chromname = cname
        self.r = res
        self.w = width
        self.model = model
        self.device = device
        self._hm = None
        self._hf = None
        self._cands = None
        self._shape = shape
        self._raw_facts = None

    def _prepare(self):
        upper, width, weights, device = self.upper, self.w, self.weights, self.device
        # expected values (peakachu/scoreUtils.py:16-24), band filter (:30-33): on the device
        # from one upload of the matrix when its values allow it, else on the host
        if not self._prepare_on_device(upper, width):
            M, raw_M = self._source(), self.raw_M
            if weights is None:
                self.exp_arr = utils.calculate_expected(M, upper + 2 * width, raw=True, device=device)
                if self._raw_is_M:
                    self.background = self.exp_arr
                else:
                    self.background = utils.calculate_expected(raw_M, upper + 2 * width, raw=True,
                                                               device=device)
            else:
                self.exp_arr = utils.calculate_expected(M, upper + 2 * width, raw=False, device=device)
                self.background = self.exp_arr
        self.get_candidate(self.lower, self.upper)

    # raw_M: what the caller handed in; from_upper makes it when somebody reads it
    @property
    def raw_M(self):
        if self._raw_val is None and self._pixels is not None:
            with stage("host copy of the mirrored matrix (on demand)"):
                self._raw_val = self._pixels.symmetric()
        return self._raw_val

    @raw_M.setter
    def raw_M(self, value):
        self._raw_val = value

    def _source(self):
        """The unfiltered matrix M of the constructor (None once self.M has been made from it)."""
        if self._M_src is None and self._M_val is None and self._pixels is not None:
            if self._bias is None:
                self._M_src = self.raw_M
            else:
                with stage("host copy of the mirrored matrix (on demand)"):
                    self._M_src = self._pixels.symmetric(self._bias)
        return self._M_src

    # self.M: peakachu/scoreUtils.py:30-33 (finite entries with -2w < col-row < upper+2w).
    # The device builds its band straight from the unfiltered matrix with the same filter,
    # so the host copy is only made when somebody reads it.
    @property
    def M(self):
        if self._M_val is None:
            self._M_val = utils.band_filter(self._source(), self.w, self.upper)
            self._M_src = None  # the reference keeps the filtered copy only (scoreUtils.py:29)
        return self._M_val

    @M.setter
    def M(self, value):
        self._M_val = value
        self._M_src = None

    def _expected_on_device(self, csr, band, maxdis, balanced):
        """utils.calculate_expected with the diagonal means taken from device bands of the
        uploaded matrix.  `band` (finite entries, raw mode only) may cover a prefix of the
        diagonals; the rest comes from a band built for the purpose."""
        n = csr.n
        top = min(int(maxdis), n - 1)
        exp_arr = np.zeros(int(maxdis) + 1)
        done = -1
        if not balanced and band is not None and band.dlo <= 0:
            done = min(top, band.dhi)
            exp_arr[:done + 1] = csr.expected_means(band, 0, done, False)
        if done < top:
            # balanced mode keeps NaN entries in the diagonals (peakachu/utils.py:156)
            extra = csr.band(done + 1, top, keep_nan=balanced)
            try:
                exp_arr[done + 1:top + 1] = csr.expected_means(extra, done + 1, top, balanced)
            finally:
                extra.close()
        return utils.isotonic_expected(exp_arr)

    def _open_on_device(self):
        """(csr of M, csr of raw_M or None when M is raw_M, stored entries of M / of raw_M)."""
        if self._pixels is not None:
            px = self._pixels
            try:
                with stage("prepare: H2D of the pixel table + facts (pk_csr_upload_upper)"):
                    base = _lib.HipCsr.from_upper(px.n, px.indptr, px.cols, px.counts, device=self.device)
            except _lib.PeakachuHipError as e:
                if "out of order" not in str(e):
                    raise
                # not the table a conforming file holds (unsorted or duplicate pixels): mirror it on
                # the host, where the conversion sorts and sums like the reference's (utils.py:10-15)
                self._raw_val = px.symmetric()
                self._M_src = self._raw_val if self._bias is None else px.symmetric(self._bias)
                self._pixels = None
                return self._open_on_device()
            if self._bias is None:
                return base, None, px.nnz, px.nnz
            try:
                with stage("prepare: balanced view of the pixels (pk_csr_view)"):
                    return base.view(self._bias), base, px.nnz, px.nnz
            except Exception:
                base.close()
                raise
        with stage("prepare: canonical_csr"):
            Mc = utils.canonical_csr(self._M_src)
        if Mc.nnz == 0:
            return None, None, 0, 0
        with stage("prepare: H2D of the CSR + facts (pk_csr_upload)"):
            csr = _lib.HipCsr(Mc, device=self.device)
        if self._raw_is_M:
            return csr, None, Mc.nnz, Mc.nnz
        try:
            with stage("prepare: canonical_csr"):
                Rc = utils.canonical_csr(self.raw_M)
            if Rc.nnz == 0:
                return csr, None, Mc.nnz, 0
            with stage("prepare: H2D of the CSR + facts (pk_csr_upload)"):
                return csr, _lib.HipCsr(Rc, device=self.device), Mc.nnz, Rc.nnz
        except Exception:
            csr.close()
            raise

    def _prepare_on_device(self, upper, width):
        weights = self.weights
        n = self._shape[0]
        if n <= 2 * width or (self._pixels is not None and self._pixels.nnz == 0):
            return False
        maxdis = upper + 2 * width
        dlo, dhi = -2 * width + 1, max(upper + 2 * width - 1, -2 * width + 1)
        csr = rcsr = None
        try:
            csr, rcsr, nnz_m, nnz_r = self._open_on_device()
            if csr is None:
                return False
            balanced = weights is not None
            if not balanced and csr.n_negative:
                return False  # the validity test is a sign of a column sum: host path
            with stage("prepare: band build (pk_matrix_from_csr)"):
                band = csr.band(dlo, dhi)
                band.dlo, band.dhi = dlo, dhi
            with stage("prepare: expected curve (device means + isotonic fit)"):
                self.exp_arr = self._expected_on_device(csr, band, maxdis, balanced)
                band.set_expected(self.exp_arr)
            self._hm = band
            if balanced or self._raw_is_M:
                self.background = self.exp_arr
            if self._raw_is_M:
                self._raw_facts = dict(integer=csr.n_noninteger == 0, vmax=csr.vmax, nnz=nnz_m, band=band)
            else:
                if nnz_r == 0:
                    if not balanced:
                        self._hm = None
                        return False
                    self._raw_facts = dict(integer=True, vmax=0.0, nnz=0, band=None)
                    return True
                if not balanced:
                    # .hic style: M holds normalised values, the background comes from the raw counts
                    if rcsr.n_negative:
                        self._hm = None
                        return False
                    self.background = self._expected_on_device(rcsr, None, maxdis, False)
                hi = min(int(upper), self.background.size - 1, n - 1)
                rband = None
                if hi >= self.lower:
                    with stage("prepare: band build (pk_matrix_from_csr)"):
                        rband = rcsr.band(self.lower, hi)
                        rband.dlo, rband.dhi = self.lower, hi
                self._raw_facts = dict(integer=rcsr.n_noninteger == 0, vmax=rcsr.vmax, nnz=nnz_r,
                                       band=rband, own=True)
            return True
        finally:
            for c in (csr, rcsr):
                if c is not None:
                    c.close()

    # ----------------------------------------------------------- host side
    def get_candidate(self, lower, upper):
        """peakachu/scoreUtils.py:40-68.  On the device when the raw counts are
        integers (pk_candidates_create: the Poisson test runs against tables made
        with scipy); otherwise, or when a pixel falls inside the table's guard
        band, on the host with scipy directly."""
        self._cands = None
        with stage("candidates: Poisson tables (scipy) + device scan"):
            got = self._candidates_on_device(lower, upper)  # device errors propagate
        if got is None:
            with stage("candidates: host path (scipy per pixel)"):
                self.ridx, self.cidx = utils.candidates(self.raw_M, self.background, self.weights,
                                                        lower, upper)
        else:
            self._cands = got
            with stage("candidates: D2H of ridx / cidx"):
                x, y = got.coords()
                self.ridx, self.cidx = x.astype(np.int64), y.astype(np.int64)
        self._cands_key = (self.ridx, self.cidx)

    def _candidates_on_device(self, lower, upper):
        facts = self._raw_facts
        self._raw_facts = None
        own = False
        if facts is not None and (lower, upper) == (self.lower, self.upper):
            # everything get_candidate needs to know about the raw counts came with the upload
            n = self._shape[0]
            hi = min(int(upper), self.background.size - 1, n - 1)
            if hi < lower or facts["nnz"] == 0 or not facts["integer"] or facts["band"] is None:
                if facts.get("own") and facts["band"] is not None:
                    facts["band"].close()
                return None
            rawm, own, kmax = facts["band"], bool(facts.get("own")), int(facts["vmax"])
        else:
            raw = utils.canonical_csr(self.raw_M)
            n = raw.shape[0]
            hi = min(int(upper), self.background.size - 1, n - 1)
            if hi < lower or raw.nnz == 0:
                return None
            fin = raw.data[np.isfinite(raw.data)]
            if not np.all(fin == np.floor(fin)):
                return None  # the tables assume integer counts
            kmax = int(fin.max()) if fin.size else 0
            if self.weights is None and self._raw_is_M:
                rawm = self._matrix()  # the scoring band already holds these counts
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
                cands, amb = _lib.HipCands.from_band(rawm, lower, hi, bg,
                                                     weights=np.asarray(self.weights, np.float64),
                                                     mustar=utils.poisson_mu_thresholds(max(kmax, 0)))
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
        n, w = self._shape[0], self.w
        # the reference masks `x-w >= 0 and y+w+1 <= n` (:75) and lets scipy's fancy indexing
        # judge the rest: a row x+w >= n or a column y-w < -n raises (only possible with x > y)
        passes = (xi - w >= 0) & (yi + w + 1 <= n)
        bad = passes & ((xi + w >= n) | (yi - w < -n))
        if np.any(bad):
            k = int(np.flatnonzero(bad)[0])
            raise IndexError("index out of range: window of (%d, %d) leaves the %d-bin matrix"
                             % (xi[k], yi[k], n))
        # (dropped coordinates only need to stay dropped when squeezed into 32 bits)
        lim = n + 2 * w + 2
        fea, _, keep = self._matrix().extract(self.w, np.clip(xi, -1, lim), np.clip(yi, -lim, lim))
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
        hm, hf = self._matrix(), self._forest()
        with stage("score: pk_score_run (extract, forest, compact)"):
            cd.run(hm, hf, self.w, thre, batch=100000)
        with stage("score: D2H of the scored pixels"):
            ri, ci, prob_pool, signal = cd.fetch()
        with stage("score: result CSR matrices"):
            ri = ri.astype(int)
            ci = ci.astype(int)
            result = sparse.csr_matrix((prob_pool, (ri, ci)), shape=self._shape)
            if ri.size > 0:
                self.M = sparse.csr_matrix((signal, (ri, ci)), shape=self._shape)
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
