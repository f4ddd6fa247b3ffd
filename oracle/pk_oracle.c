/*
 * pk_oracle.c -- CPU restatement of Peakachu's scoring hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported baseline.  The
 * product path (peakachu_amd + libpeakachu_hip.so) never links or calls it.
 *
 * Parity status: PINNED.  tools/make_golden.py imports the reference's own
 * Python (from /root/reference, in the build container only) and stores its
 * inputs/outputs under tests/golden/; tests/test_oracle_golden.py requires
 * this file to reproduce them bit-for-bit.
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference).  Third-party arithmetic restated from published
 * behaviour: scipy 1.15.3 ndimage.gaussian_filter (correlate1d, symmetric
 * kernel, mode='reflect'), scipy.sparse CSR sampling, scikit-learn 1.7.2
 * RandomForestClassifier.predict_proba.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PKO_MAX_S 64 /* window side limit: w <= 31 */

/* ---- scipy.sparse csr_matrix.__getitem__(rows, cols) -> csr_sample_values
 * (called at peakachu/scoreUtils.py:81,120 and peakachu/trainUtils.py:36).
 * For a canonical CSR (sorted, no duplicates) this is a binary search per
 * requested cell; an absent cell reads 0.0. */
static double csr_get(const int32_t *indptr, const int32_t *indices,
                      const double *data, int64_t r, int64_t c)
{
    int32_t lo = indptr[r], hi = indptr[r + 1];
    while (lo < hi) {
        int32_t mid = lo + (hi - lo) / 2;
        if (indices[mid] < c)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo < indptr[r + 1] && indices[lo] == c)
        return data[lo];
    return 0.0;
}

double pko_csr_get(const int32_t *indptr, const int32_t *indices,
                   const double *data, int64_t r, int64_t c)
{
    return csr_get(indptr, indices, data, r, c);
}

/* ---- scipy.ndimage._filters._gaussian_kernel1d(sigma=1, order=0, radius=4)
 * = exp(-0.5 x^2) / sum, x = -4..4 (gaussian_filter truncate=4.0 ->
 * radius = int(4.0*1 + 0.5) = 4).  Hex constants so no libm exp() rounding
 * can creep in; tests/test_oracle_golden.py checks them against scipy. */
static double GK[5] = {
    0x1.9884a307594fbp-2,  /* k0 (centre) */
    0x1.ef8eb9ad499bap-3,  /* k1 */
    0x1.ba4b99d1799abp-5,  /* k2 */
    0x1.22724cb7eb269p-8,  /* k3 */
    0x1.18a9c4fd536c6p-13, /* k4 */
};

const double *pko_gauss_taps(void) { return GK; }
/* tests only: the taps of another numpy (see tests/test_oracle_golden.py) */
void pko_set_gauss_taps(const double *k) { for (int i = 0; i < 5; i++) GK[i] = k[i]; }

/* scipy 'reflect' boundary: (d c b a | a b c d | d c b a). */
static inline int reflect_idx(int i, int n)
{
    if (n == 1)
        return 0;
    while (i < 0 || i >= n) {
        if (i < 0)
            i = -i - 1;
        else
            i = 2 * n - 1 - i;
    }
    return i;
}

/* ---- scipy.ndimage.gaussian_filter(arr, sigma=1, order=0) as called at
 * peakachu/scoreUtils.py:86 and peakachu/trainUtils.py:41.
 * Separable: correlate1d along axis 0 (rows) then axis 1, float64.
 * NI_Correlate1D's symmetric-kernel branch computes, per output sample,
 *     acc = in[c]*k0;  for j = 4,3,2,1: acc += (in[c-j] + in[c+j]) * k_j
 * (outermost taps first), with the line extended by 'reflect'. */
static void gauss2d(const double *in, double *out, int S)
{
    double tmp[PKO_MAX_S * PKO_MAX_S];
    for (int j = 0; j < S; j++) {          /* axis 0: filter down each column */
        for (int i = 0; i < S; i++) {
            double acc = in[i * S + j] * GK[0];
            for (int k = 4; k >= 1; k--) {
                double a = in[reflect_idx(i - k, S) * S + j];
                double b = in[reflect_idx(i + k, S) * S + j];
                acc += (a + b) * GK[k];
            }
            tmp[i * S + j] = acc;
        }
    }
    for (int i = 0; i < S; i++) {          /* axis 1: filter along each row */
        for (int j = 0; j < S; j++) {
            double acc = tmp[i * S + j] * GK[0];
            for (int k = 4; k >= 1; k--) {
                double a = tmp[i * S + reflect_idx(j - k, S)];
                double b = tmp[i * S + reflect_idx(j + k, S)];
                acc += (a + b) * GK[k];
            }
            out[i * S + j] = acc;
        }
    }
}

void pko_gauss2d(const double *in, double *out, int S) { gauss2d(in, out, S); }

/* ---- peakachu/utils.py:204-209 image_normalize:
 * (a - a.min()) / (a.max() - a.min()); numpy min/max propagate NaN. */
static void image_normalize(double *a, int n)
{
    double mn = a[0], mx = a[0];
    int has_nan = isnan(a[0]);
    for (int i = 1; i < n; i++) {
        double v = a[i];
        if (isnan(v))
            has_nan = 1;
        if (v < mn)
            mn = v;
        if (v > mx)
            mx = v;
    }
    if (has_nan)
        mn = mx = NAN;
    double den = mx - mn;
    for (int i = 0; i < n; i++)
        a[i] = (a[i] - mn) / den;
}

/* ---- one candidate through peakachu/utils.py:211-237 (distance_normalize),
 * :180-202 (distance_normaize_core), then scoreUtils.py:85-88 (blur, scale,
 * ravel).  Returns 1 and fills fea[F] if the window survives the filters. */
static int one_window(int32_t n, const int32_t *indptr, const int32_t *indices,
                      const double *data, const double *exp_arr,
                      int64_t exp_len, int w, int64_t x, int64_t y, double *fea)
{
    const int S = 2 * w + 1, F = S * S;
    double win[PKO_MAX_S * PKO_MAX_S];
    /* scoreUtils.py:77-82: S x S gather, row offset i, col offset j.  scipy's fancy
     * indexing counts a negative column from the far end (possible only when x > y;
     * rows and columns beyond that raise there: see pko_extract). */
    for (int i = 0; i < S; i++)
        for (int j = 0; j < S; j++) {
            int64_t col = y - w + j;
            if (col < 0)
                col += n;
            win[i * S + j] = csr_get(indptr, indices, data, x - w + i, col);
        }
    /* utils.py:221-223: NaN -> 0 */
    int nnz = 0;
    for (int k = 0; k < F; k++) {
        if (isnan(win[k]))
            win[k] = 0.0;
        if (win[k] != 0.0)
            nnz++;
    }
    /* utils.py:225: count_nonzero(window) < window.size*0.1 -> skip */
    if ((double)nnz < (double)F * 0.1)
        return 0;
    /* utils.py:228: window[:w,:w].mean(); numba's mean accumulates
     * sequentially in C order, then divides by the size. */
    double acc = 0.0;
    for (int i = 0; i < w; i++)
        for (int j = 0; j < w; j++)
            acc += win[i * S + j];
    double ll_mean = acc / (double)(w * w);
    if (!(ll_mean > 0.0))
        return 0;
    double p2ll = win[w * S + w] / ll_mean; /* utils.py:230-232 */
    if (!(p2ll > 0.1))
        return 0;
    /* utils.py:180-202: divide by expected at |col-row| unless the largest
     * distance in the window is outside exp_arr (then left unnormalised). */
    int64_t dmax = 0;
    for (int i = 0; i < S; i++)
        for (int j = 0; j < S; j++) {
            int64_t d = llabs((y - w + j) - (x - w + i));
            if (d > dmax)
                dmax = d;
        }
    if (dmax < exp_len) {
        for (int i = 0; i < S; i++)
            for (int j = 0; j < S; j++) {
                int64_t d = llabs((y - w + j) - (x - w + i));
                win[i * S + j] = win[i * S + j] / exp_arr[d];
            }
    }
    gauss2d(win, fea, S);
    image_normalize(fea, F);
    return 1;
}

/* ---- Chromosome.getwindow (peakachu/scoreUtils.py:70-93) and, with
 * train_mask != 0, the pre-filter of trainUtils.buildmatrix
 * (peakachu/trainUtils.py:22: additionally yi - xi > w).
 * keep[k] = index into the input of survivor k (input order);
 * fea64 is [n_keep, F] row-major (may be NULL). Returns n_keep. */
int64_t pko_extract(int32_t n, const int32_t *indptr, const int32_t *indices,
                    const double *data, const double *exp_arr, int64_t exp_len,
                    int w, int64_t N, const int64_t *x, const int64_t *y,
                    int train_mask, double *fea64, int64_t *keep)
{
    const int S = 2 * w + 1, F = S * S;
    double fea[PKO_MAX_S * PKO_MAX_S];
    int64_t nk = 0;
    for (int64_t c = 0; c < N; c++) {
        int64_t xi = x[c], yi = y[c];
        /* scoreUtils.py:75 */
        if (!(xi - w >= 0 && yi + w + 1 <= n))
            continue;
        if (train_mask && !(yi - xi > w))
            continue;
        /* scoreUtils.py:81: M[rows, cols] raises IndexError for a row >= n or a column
         * < -n (the mask above admits such windows only when x > y) */
        if (xi + w >= n || yi - w < -(int64_t)n)
            return -1;
        if (!one_window(n, indptr, indices, data, exp_arr, exp_len, w, xi, yi, fea))
            continue;
        if (fea64)
            memcpy(fea64 + nk * F, fea, sizeof(double) * F);
        keep[nk++] = c;
    }
    return nk;
}

/* ---- sklearn 1.7.2 ForestClassifier.predict_proba(X)[:, 1] as called at
 * peakachu/scoreUtils.py:109.  X is cast to float32 by sklearn's input
 * validation; Tree._apply_dense walks each tree: at an internal node
 * (left != -1) a NaN goes missing_go_to_left ? left : right, otherwise
 * (double)x[feature] <= threshold ? left : right.  Each tree contributes
 * value[leaf, 0, 1] (a class fraction); contributions are added in tree
 * order t = 0..T-1 in float64 (n_jobs = 1, peakachu/trainUtils.py:51) and
 * the sum divided by T.  Arrays are sklearn's own per-tree node arrays laid
 * end to end; tree t owns nodes [tree_off[t], tree_off[t+1]) and child
 * indices are relative to tree_off[t]. */
void pko_predict(int T, const int32_t *tree_off, const int32_t *left,
                 const int32_t *right, const int32_t *feat, const double *thr,
                 const uint8_t *miss_left, const double *p1, int F, int64_t N,
                 const float *fea32, double *out)
{
    for (int64_t c = 0; c < N; c++) {
        const float *xrow = fea32 + c * F;
        double acc = 0.0;
        for (int t = 0; t < T; t++) {
            int32_t base = tree_off[t];
            int32_t node = 0;
            while (left[base + node] != -1) {
                float xv = xrow[feat[base + node]];
                int go_left;
                if (isnan(xv))
                    go_left = miss_left[base + node] != 0;
                else
                    go_left = (double)xv <= thr[base + node];
                node = go_left ? left[base + node] : right[base + node];
            }
            acc += p1[base + node];
        }
        out[c] = acc / (double)T;
    }
}

/* ---- Chromosome.score (peakachu/scoreUtils.py:95-125) up to the point
 * where the two CSR results are built: candidates in input order, batches of
 * `batch` (100000 at :104), getwindow per batch, the forest only when more
 * than one window survives the batch (:108), keep p > thre (:110, strict),
 * signal = M[ri, ci] (:120).  Outputs are in candidate order. */
int64_t pko_score(int32_t n, const int32_t *indptr, const int32_t *indices,
                  const double *data, const double *exp_arr, int64_t exp_len,
                  int w, int T, const int32_t *tree_off, const int32_t *left,
                  const int32_t *right, const int32_t *feat, const double *thr,
                  const uint8_t *miss_left, const double *p1, double thre,
                  int64_t batch, int64_t N, const int64_t *x, const int64_t *y,
                  int64_t *ox, int64_t *oy, double *op, double *osig)
{
    const int S = 2 * w + 1, F = S * S;
    int64_t n_out = 0;
    if (batch <= 0)
        batch = 100000;
    int64_t cap = batch < N ? batch : N;
    double *fea64 = (double *)malloc(sizeof(double) * (size_t)(cap > 0 ? cap : 1) * F);
    float *fea32 = (float *)malloc(sizeof(float) * (size_t)(cap > 0 ? cap : 1) * F);
    int64_t *keep = (int64_t *)malloc(sizeof(int64_t) * (size_t)(cap > 0 ? cap : 1));
    double *p = (double *)malloc(sizeof(double) * (size_t)(cap > 0 ? cap : 1));
    for (int64_t t0 = 0; t0 < N; t0 += batch) {
        int64_t nb = N - t0 < batch ? N - t0 : batch;
        int64_t nk = pko_extract(n, indptr, indices, data, exp_arr, exp_len, w,
                                 nb, x + t0, y + t0, 0, fea64, keep);
        if (nk > 1) {
            for (int64_t k = 0; k < nk * F; k++)
                fea32[k] = (float)fea64[k]; /* sklearn check_array -> float32 */
            pko_predict(T, tree_off, left, right, feat, thr, miss_left, p1, F,
                        nk, fea32, p);
            for (int64_t k = 0; k < nk; k++) {
                if (p[k] > thre) {
                    int64_t c = t0 + keep[k];
                    ox[n_out] = x[c];
                    oy[n_out] = y[c];
                    op[n_out] = p[k];
                    osig[n_out] = csr_get(indptr, indices, data, x[c], y[c]);
                    n_out++;
                }
            }
        }
    }
    free(fea64);
    free(fea32);
    free(keep);
    free(p);
    return n_out;
}

/* ---- All-core variant of pko_score for the bench's cpu_baseline leg: the
 * same per-candidate arithmetic (one_window + one forest walk), OpenMP over
 * candidates, then a serial pass that applies the reference's per-batch rule
 * (scoreUtils.py:108) and threshold.  Results are identical to pko_score. */
#ifdef _OPENMP
#include <omp.h>
#endif
int pko_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* all_status / all_prob (either may be NULL): per candidate, whether its window
 * survived getwindow's filters and predict_proba[:,1] of its features (0 where
 * it did not) -- what pk_score_fetch_all reports, for full-size parity tests. */
int64_t pko_score_all_mt(int nthreads, int32_t n, const int32_t *indptr,
                         const int32_t *indices, const double *data,
                         const double *exp_arr, int64_t exp_len, int w, int T,
                         const int32_t *tree_off, const int32_t *left,
                         const int32_t *right, const int32_t *feat,
                         const double *thr, const uint8_t *miss_left,
                         const double *p1, double thre, int64_t batch, int64_t N,
                         const int64_t *x, const int64_t *y, int64_t *ox,
                         int64_t *oy, double *op, double *osig,
                         uint8_t *all_status, double *all_prob)
{
    const int S = 2 * w + 1, F = S * S;
    if (batch <= 0)
        batch = 100000;
    uint8_t *ok = (uint8_t *)calloc((size_t)(N > 0 ? N : 1), 1);
    double *p = (double *)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
#ifdef _OPENMP
    if (nthreads > 0)
        omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 256)
#endif
    for (int64_t c = 0; c < N; c++) {
        double fea[PKO_MAX_S * PKO_MAX_S];
        float f32[PKO_MAX_S * PKO_MAX_S];
        int64_t xi = x[c], yi = y[c];
        if (!(xi - w >= 0 && yi + w + 1 <= n))
            continue;
        if (!one_window(n, indptr, indices, data, exp_arr, exp_len, w, xi, yi, fea))
            continue;
        for (int k = 0; k < F; k++)
            f32[k] = (float)fea[k];
        pko_predict(T, tree_off, left, right, feat, thr, miss_left, p1, F, 1,
                    f32, &p[c]);
        ok[c] = 1;
    }
    int64_t n_out = 0;
    for (int64_t t0 = 0; t0 < N; t0 += batch) {
        int64_t nb = N - t0 < batch ? N - t0 : batch, nk = 0;
        for (int64_t c = t0; c < t0 + nb; c++)
            nk += ok[c];
        if (nk <= 1)
            continue;
        for (int64_t c = t0; c < t0 + nb; c++)
            if (ok[c] && p[c] > thre) {
                ox[n_out] = x[c];
                oy[n_out] = y[c];
                op[n_out] = p[c];
                osig[n_out] = csr_get(indptr, indices, data, x[c], y[c]);
                n_out++;
            }
    }
    for (int64_t c = 0; c < N; c++) {
        if (all_status)
            all_status[c] = ok[c];
        if (all_prob)
            all_prob[c] = ok[c] ? p[c] : 0.0;
    }
    free(ok);
    free(p);
    return n_out;
}

int64_t pko_score_mt(int nthreads, int32_t n, const int32_t *indptr,
                     const int32_t *indices, const double *data,
                     const double *exp_arr, int64_t exp_len, int w, int T,
                     const int32_t *tree_off, const int32_t *left,
                     const int32_t *right, const int32_t *feat,
                     const double *thr, const uint8_t *miss_left,
                     const double *p1, double thre, int64_t batch, int64_t N,
                     const int64_t *x, const int64_t *y, int64_t *ox,
                     int64_t *oy, double *op, double *osig)
{
    return pko_score_all_mt(nthreads, n, indptr, indices, data, exp_arr, exp_len, w, T,
                            tree_off, left, right, feat, thr, miss_left, p1, thre, batch,
                            N, x, y, ox, oy, op, osig, NULL, NULL);
}
