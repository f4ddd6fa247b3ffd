// pk_compact.hip -- band construction and the tail of Chromosome.score for
// gfx950: the reference's batch rule, the p > thre filter, deterministic
// stream compaction in candidate order and the signal re-gather
// (peakachu/scoreUtils.py:104-121).  Integer / byte work, HBM-bound, tiny
// next to the extract and forest kernels.
#include "pk_common.h"

namespace {

// row of CSR entry e: largest r with indptr[r] <= e
__device__ __forceinline__ int csr_row_of(const int32_t *__restrict__ indptr, int n, int64_t e)
{
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)indptr[mid] <= e) lo = mid;
        else hi = mid;
    }
    return lo;
}

// ---- CSR -> diagonal-major band: one thread per stored entry ------------
// filter: 0 = every stored entry (the host has filtered already), 1 = finite non-zero
// entries only (the band filter of peakachu/scoreUtils.py:30-33 done here), 2 = non-zero
// entries including NaN (what utils.calculate_expected keeps in balanced mode).
// bias != nullptr: the value of an entry is (bias[row] * bias[col]) * data -- what cooler's
// matrix(balance=name) hands peakachu/score_genome.py:55, the two biases multiplied first.
// upper: the arrays hold the UPPER triangle only, as a .cool stores a chromosome (col >= row;
// entries with col >= n are pixels of other chromosomes and are skipped); the matrix is its
// mirror image, so an entry off the diagonal is written twice.
__global__ void band_build_kernel(const int32_t *__restrict__ indptr,
                                  const int32_t *__restrict__ indices,
                                  const double *__restrict__ data, int64_t nnz, int n, int dlo,
                                  int dhi, int64_t ld, double *__restrict__ band, int filter,
                                  const double *__restrict__ bias, int upper)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int col = indices[e];
    double v = data[e];
    if (col < 0 || col >= n) return;
    const int r = csr_row_of(indptr, n, e);
    if (bias) v = (bias[r] * bias[col]) * v;
    if (filter && (v == 0.0 || (filter == 1 && !(v - v == 0.0)))) return;  // (v - v != 0: inf or NaN)
    const int k = col - r;
    if (k >= dlo && k <= dhi) band[(int64_t)(k - dlo) * ld + r] = v;
    if (upper && k > 0 && -k >= dlo && -k <= dhi) band[(int64_t)(-k - dlo) * ld + col] = v;
}

// ---- facts about the stored values, and the bins calculate_expected calls valid -------
// info[0] finite non-zero entries, [1] non-finite entries, [2] finite entries that are not
// non-negative integers, [3] finite negative entries; vmax_bits = largest finite value
// (as ordered bits of a non-negative double).  valid_raw[c] = column c holds a positive
// finite entry; valid_bal[r] = valid_bal[c] = 1 for every finite non-zero entry
// (peakachu/utils.py:145-155).
__global__ void __launch_bounds__(256) csr_info_kernel(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                       const double *__restrict__ data, int64_t nnz, int n,
                                                       unsigned long long *__restrict__ info,
                                                       unsigned long long *__restrict__ vmax_bits,
                                                       uint8_t *__restrict__ valid_raw, uint8_t *__restrict__ valid_bal,
                                                       const double *__restrict__ bias, int upper,
                                                       unsigned long long *__restrict__ bad_order)
{
    // A fixed grid strides over the entries and every thread keeps its own counts: one atomic per
    // WORKGROUP and counter at the end.  (Rounds 2-4 issued them per wave: 380 000 atomics on two
    // addresses for a 12 M-entry chromosome, serialised in the L2 -- 4.2 ms for a scan that takes 0.2;
    // profiles/r05_real_regime_kernel_stats.csv shows it as the longest kernel of a chromosome.)
    unsigned n_fin = 0, n_non = 0, n_frac = 0, n_neg = 0, n_bad = 0;
    unsigned long long mx = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
        double v = data[e];
        const int col = indices[e];
        const bool inside = col >= 0 && col < n;
        int r = -1;
        if (upper || bias) {
            r = csr_row_of(indptr, n, e);
            // an upper-triangle table must be what a .cool holds: columns strictly ascending
            // inside a row, none left of the diagonal (a duplicate pixel would be written, not summed)
            if (upper && (col < r || (e > (int64_t)indptr[r] && indices[e - 1] >= col))) n_bad++;
            if (bias && inside) v = (bias[r] * bias[col]) * v;
        }
        // (an off-diagonal entry of an upper-triangle table stands for two entries of the matrix)
        const unsigned weight = (upper && col != r) ? 2u : 1u;
        if (upper && !inside) {
            // a pixel of another chromosome: not part of this matrix
        } else if (!(v - v == 0.0)) {
            n_non += weight;
        } else if (v != 0.0) {
            n_fin += weight;
            if (v < 0.0) n_neg += weight;
            if (v < 0.0 || v != __builtin_floor(v)) n_frac += weight;
            if (r < 0 && inside) r = csr_row_of(indptr, n, e);
            if (v > 0.0) {
                const unsigned long long b = (unsigned long long)__double_as_longlong(v);
                mx = b > mx ? b : mx;
                if (inside && !valid_raw[col]) valid_raw[col] = 1;
                if (upper && !valid_raw[r]) valid_raw[r] = 1;  // (the mirrored entry sits in column r)
            }
            if (inside) {
                if (!valid_bal[col]) valid_bal[col] = 1;
                if (!valid_bal[r]) valid_bal[r] = 1;
            }
        }
    }
    __shared__ unsigned long long part[4][6];
    unsigned long long v6[6] = {n_fin, n_non, n_frac, n_neg, n_bad, mx};
#pragma unroll
    for (int k = 0; k < 6; k++)
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(v6[k], o);
            v6[k] = k == 5 ? (other > v6[k] ? other : v6[k]) : v6[k] + other;
        }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 6; k++) part[threadIdx.x >> 6][k] = v6[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        unsigned long long t = part[0][k];
        for (int w = 1; w < 4; w++) t = k == 5 ? (part[w][k] > t ? part[w][k] : t) : t + part[w][k];
        if (t) {
            if (k < 4) atomicAdd(&info[k], t);
            else if (k == 4) atomicAdd(bad_order, t);
            else atomicMax(vmax_bits, t);
        }
    }
}

// int32 counts of a pixel table -> the float64 values every other kernel reads
__global__ void counts_to_f64_kernel(const int32_t *__restrict__ src, double *__restrict__ dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}

// ---- survivors per reference batch (scoreUtils.py:104-108) --------------
// One workgroup counts 4096 consecutive candidates: wave ballots, one LDS
// add per wave, one global atomic per workgroup and batch it touches (a
// workgroup straddles at most two batches when batch >= 4096; smaller
// batches fall back to per-candidate atomics).
constexpr int BC_ITEMS = 16;
__global__ void __launch_bounds__(256) batch_count_kernel(const uint8_t *__restrict__ status, int64_t N, int64_t batch,
                                                          int32_t *__restrict__ batch_cnt)
{
    __shared__ int cnt[2];
    const int64_t base = (int64_t)blockIdx.x * (256 * BC_ITEMS);
    const int64_t b0 = base / batch;
    if (batch < 256 * BC_ITEMS) {  // (a block may span more than two batches)
        for (int i = 0; i < BC_ITEMS; i++) {
            const int64_t c = base + (int64_t)i * 256 + threadIdx.x;
            if (c < N && status[c]) atomicAdd(&batch_cnt[c / batch], 1);
        }
        return;
    }
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    // thread t takes the BC_ITEMS = 16 consecutive flags base + 16 t ..: one 16-byte load (the list is
    // allocated by hipMalloc and `base` is a multiple of 4 096); a flag counts when it is not zero,
    // whatever its value (the byte-wise test below is exact for all 256 values, like the slow paths)
    static_assert(BC_ITEMS == 16, "one uint4 of flags per thread");
    int mine0 = 0, mine1 = 0;
    const int64_t c0 = base + (int64_t)threadIdx.x * BC_ITEMS;
    if (c0 + BC_ITEMS <= N) {
        const uint4 v = *reinterpret_cast<const uint4 *>(status + c0);
        const unsigned w4[4] = {v.x, v.y, v.z, v.w};
        if (c0 / batch == (c0 + BC_ITEMS - 1) / batch) {
            int k = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)   // bit 7 of a byte = "the byte is not zero"
                k += __popc((((w4[j] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w4[j]) & 0x80808080u);
            if (c0 / batch == b0) mine0 = k;
            else mine1 = k;
        } else {
#pragma unroll
            for (int j = 0; j < BC_ITEMS; j++)
                if ((w4[j >> 2] >> ((j & 3) * 8)) & 0xFFu) {
                    if ((c0 + j) / batch == b0) mine0++;
                    else mine1++;
                }
        }
    } else {
        for (int j = 0; j < BC_ITEMS; j++) {
            const int64_t c = c0 + j;
            if (c < N && status[c]) {
                if (c / batch == b0) mine0++;
                else mine1++;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mine0 += __shfl_xor(mine0, o);
        mine1 += __shfl_xor(mine1, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (mine0) atomicAdd(&cnt[0], mine0);
        if (mine1) atomicAdd(&cnt[1], mine1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (cnt[0]) atomicAdd(&batch_cnt[b0], cnt[0]);
        if (cnt[1]) atomicAdd(&batch_cnt[b0 + 1], cnt[1]);
    }
}

constexpr int CB = 256;          // threads per compaction block
constexpr int CITEMS = 4;        // candidates per thread
constexpr int CTILE = CB * CITEMS;

__device__ __forceinline__ bool keep_flag(const uint8_t *status, const double *prob,
                                          const int32_t *batch_cnt, int64_t batch, double thre,
                                          int64_t c, int64_t N)
{
    // fea.shape[0] > 1 (scoreUtils.py:108) and p > thre (scoreUtils.py:110)
    return c < N && status[c] && batch_cnt[c / batch] > 1 && prob[c] > thre;
}

__global__ void compact_count_kernel(const uint8_t *__restrict__ status,
                                     const double *__restrict__ prob,
                                     const int32_t *__restrict__ batch_cnt, int64_t batch,
                                     double thre, int64_t N, int64_t *__restrict__ block_cnt)
{
    __shared__ int wsum[CB / 64];
    const int64_t base = (int64_t)blockIdx.x * CTILE + (int64_t)threadIdx.x * CITEMS;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < CITEMS; i++) cnt += keep_flag(status, prob, batch_cnt, batch, thre, base + i, N);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int i = 0; i < CB / 64; i++) s += wsum[i];
        block_cnt[blockIdx.x] = s;
    }
}

// exclusive scan of block_cnt in place by one workgroup; total -> *n_out
// (thread t owns a contiguous run of blocks; the 1024 partial sums are scanned by wave shuffles --
// a serial pass of one thread over them was half of this kernel's 19 us)
__global__ void __launch_bounds__(1024) compact_scan_kernel(int64_t *__restrict__ block_cnt, int64_t nblocks,
                                                            int64_t *__restrict__ n_out)
{
    __shared__ int64_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = (nblocks + 1023) / 1024;
    const int64_t b0 = (int64_t)tid * per;
    const int64_t b1 = b0 + per < nblocks ? b0 + per : nblocks;
    int64_t s = 0;
    for (int64_t b = b0; b < b1; b++) s += block_cnt[b];
    int64_t incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int64_t woff = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int64_t v = wsum[i];
        if (i < wave) woff += v;
        total += v;
    }
    if (tid == 0) *n_out = total;
    int64_t run = woff + incl - s;
    for (int64_t b = b0; b < b1; b++) {
        const int64_t v = block_cnt[b];
        block_cnt[b] = run;
        run += v;
    }
}

__global__ void compact_scatter_kernel(const uint8_t *__restrict__ status,
                                       const double *__restrict__ prob,
                                       const int32_t *__restrict__ batch_cnt, int64_t batch,
                                       double thre, int64_t N, const int64_t *__restrict__ block_off,
                                       const int32_t *__restrict__ xs, const int32_t *__restrict__ ys,
                                       const double *__restrict__ band, int64_t ld, int dlo, int dhi,
                                       int32_t *__restrict__ ox, int32_t *__restrict__ oy,
                                       double *__restrict__ op, double *__restrict__ osig)
{
    __shared__ int wsum[CB / 64];
    const int64_t base = (int64_t)blockIdx.x * CTILE + (int64_t)threadIdx.x * CITEMS;
    bool k[CITEMS];
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < CITEMS; i++) {
        k[i] = keep_flag(status, prob, batch_cnt, batch, thre, base + i, N);
        cnt += k[i];
    }
    // exclusive scan of cnt over the workgroup, thread order
    int incl = cnt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < wave; i++) woff += wsum[i];
    int64_t pos = block_off[blockIdx.x] + woff + incl - cnt;
#pragma unroll
    for (int i = 0; i < CITEMS; i++) {
        if (!k[i]) continue;
        const int64_t c = base + i;
        const int x = xs[c], y = ys[c];
        ox[pos] = x;
        oy[pos] = y;
        op[pos] = prob[c];
        const int d = y - x;
        // signal = M[row, col] (scoreUtils.py:120)
        osig[pos] = (d >= dlo && d <= dhi) ? band[(int64_t)(d - dlo) * ld + x] : 0.0;
        pos++;
    }
}

// ---- get_candidate: flag every band pixel, then ordered compaction -----------
// index space: q = (k - lower) * n + r, k = diagonal, r = row; flag 1 = candidate,
// 2 = inside the guard band of the mustar table (host decides)
__device__ __forceinline__ int cand_flag(const double *__restrict__ band, int64_t ld, int dlo,
                                         int n, int lower, int64_t q, const int64_t *kstar,
                                         const double *bg, const double *w, const double *mustar,
                                         int64_t n_mustar)
{
    const int k = lower + (int)(q / n);
    const int r = (int)(q - (int64_t)(k - lower) * n);
    if (r + k >= n) return 0;
    const double v = band[(int64_t)(k - dlo) * ld + r];
    if (!(v > 0.0) || !(bg[k] > 0.0)) return 0;  // diag > 0, e > 0 (scoreUtils.py:49,61)
    const double fv = floor(v);
    if (!w) return (fv >= (double)kstar[k]) ? 1 : 0;
    // exp = 1.0 * e / (b1 * b2)   (scoreUtils.py:57)
    const double mu = (1.0 * bg[k]) / (w[r] * w[r + k]);
    if (!(mu > 0.0) || mu > 1.7e308) return 0;  // NaN / non-positive mu: p is NaN, dropped at :61
    if (!(fv < (double)n_mustar)) return 2;
    const double ms = mustar[(int64_t)fv];
    if (mu < ms * (1.0 - 1e-9)) return 1;
    if (mu > ms * (1.0 + 1e-9)) return 0;
    return 2;
}

__global__ void cand_count_kernel(const double *__restrict__ band, int64_t ld, int dlo, int n,
                                  int lower, int64_t Q, const int64_t *__restrict__ kstar,
                                  const double *__restrict__ bg, const double *__restrict__ w,
                                  const double *__restrict__ mustar, int64_t n_mustar,
                                  int64_t *__restrict__ block_cnt, int64_t *__restrict__ n_amb)
{
    __shared__ int wsum[CB / 64];
    const int64_t base = (int64_t)blockIdx.x * CTILE + (int64_t)threadIdx.x * CITEMS;
    int cnt = 0, amb = 0;
#pragma unroll
    for (int i = 0; i < CITEMS; i++) {
        const int64_t q = base + i;
        const int f = q < Q ? cand_flag(band, ld, dlo, n, lower, q, kstar, bg, w, mustar, n_mustar) : 0;
        cnt += f == 1;
        amb += f == 2;
    }
    if (amb) atomicAdd((unsigned long long *)n_amb, (unsigned long long)amb);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int i = 0; i < CB / 64; i++) s += wsum[i];
        block_cnt[blockIdx.x] = s;
    }
}

__global__ void cand_scatter_kernel(const double *__restrict__ band, int64_t ld, int dlo, int n,
                                    int lower, int64_t Q, const int64_t *__restrict__ kstar,
                                    const double *__restrict__ bg, const double *__restrict__ w,
                                    const double *__restrict__ mustar, int64_t n_mustar,
                                    const int64_t *__restrict__ block_off, int32_t *__restrict__ ox,
                                    int32_t *__restrict__ oy)
{
    __shared__ int wsum[CB / 64];
    const int64_t base = (int64_t)blockIdx.x * CTILE + (int64_t)threadIdx.x * CITEMS;
    bool kf[CITEMS];
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < CITEMS; i++) {
        const int64_t q = base + i;
        kf[i] = q < Q && cand_flag(band, ld, dlo, n, lower, q, kstar, bg, w, mustar, n_mustar) == 1;
        cnt += kf[i];
    }
    int incl = cnt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < wave; i++) woff += wsum[i];
    int64_t pos = block_off[blockIdx.x] + woff + incl - cnt;
#pragma unroll
    for (int i = 0; i < CITEMS; i++) {
        if (!kf[i]) continue;
        const int64_t q = base + i;
        const int k = lower + (int)(q / n);
        const int r = (int)(q - (int64_t)(k - lower) * n);
        ox[pos] = r;
        oy[pos] = r + k;
        pos++;
    }
}

// ---- calculate_expected (peakachu/utils.py:139-178): mean of every diagonal over
// the valid bins, summed exactly as numpy does ---------------------------------
// numpy's add.reduce walks a contiguous float64 array in buffers of 8192 elements;
// each buffer is summed pairwise (blocks of <= 128 elements with 8 interleaved
// accumulators, split at n/2 rounded down to a multiple of 8) and the buffer sums are
// added one after the other (checked against np.sum / ndarray.mean in
// tests/test_host_golden.py::test_numpy_sum_model).
__device__ double np_block_sum(const double *a, int n)  // n <= 128
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}

constexpr int EXP_THREADS = 256;
constexpr int EXP_MAXLEAF = 128;  // an 8192-element buffer splits into at most 128 blocks

// one workgroup per diagonal
__global__ __launch_bounds__(EXP_THREADS) void expected_means_kernel(
    const double *__restrict__ band, int64_t ld, int dlo, int n, int first, int top,
    const uint8_t *__restrict__ valid, double *__restrict__ scratch, double *__restrict__ means)
{
    __shared__ int cnt[EXP_THREADS + 1];
    __shared__ int leaf_lo[EXP_MAXLEAF], leaf_n[EXP_MAXLEAF];
    __shared__ double leaf_sum[EXP_MAXLEAF];
    __shared__ int n_leaf;
    __shared__ double total;
    const int i = first + blockIdx.x;  // diagonal
    if (i > top) return;
    const int tid = threadIdx.x;
    const int len = n - i;
    const double *diag = band + (int64_t)(i - dlo) * ld;
    double *vals = scratch + (int64_t)(i - first) * ld;
    // 1. ordered compaction of diag[r] over valid[r] & valid[r+i] (thread t owns a
    //    contiguous run of rows, so thread order = row order)
    const int seg = (len + EXP_THREADS - 1) / EXP_THREADS;
    const int r0 = tid * seg, r1 = min(len, r0 + seg);
    int c = 0;
    for (int r = r0; r < r1; r++) c += (valid[r] && valid[r + i]) ? 1 : 0;
    cnt[tid + 1] = c;
    if (tid == 0) cnt[0] = 0;
    __syncthreads();
    if (tid == 0)
        for (int t = 1; t <= EXP_THREADS; t++) cnt[t] += cnt[t - 1];
    __syncthreads();
    int pos = cnt[tid];
    for (int r = r0; r < r1; r++)
        if (valid[r] && valid[r + i]) vals[pos++] = diag[r];
    const int nv = cnt[EXP_THREADS];
    __syncthreads();
    if (nv <= 10) {  // utils.py:168: only diagonals with more than 10 valid pixels
        if (tid == 0) means[i] = 0.0;
        return;
    }
    // 2. numpy-order sum: buffers of 8192, pairwise inside
    if (tid == 0) total = 0.0;
    for (int b0 = 0; b0 < nv; b0 += 8192) {
        const int bn = min(8192, nv - b0);
        __syncthreads();
        if (tid == 0) {
            // enumerate the blocks of the pairwise recursion in order (explicit stack)
            int slo[16], sn[16], sp = 0, nl = 0;
            slo[0] = 0; sn[0] = bn; sp = 1;
            while (sp > 0) {
                sp--;
                const int lo = slo[sp], m = sn[sp];
                if (m <= 128) {
                    leaf_lo[nl] = lo; leaf_n[nl] = m; nl++;
                } else {
                    int h = m / 2;
                    h -= h % 8;
                    slo[sp] = lo + h; sn[sp] = m - h; sp++;   // right, visited second
                    slo[sp] = lo; sn[sp] = h; sp++;           // left first
                }
            }
            n_leaf = nl;
        }
        __syncthreads();
        for (int l = tid; l < n_leaf; l += EXP_THREADS)
            leaf_sum[l] = np_block_sum(vals + b0 + leaf_lo[l], leaf_n[l]);
        __syncthreads();
        if (tid == 0) {
            // combine in recursion order: post-order walk with a value stack
            // frame: (n, state) ; leaves are consumed in the order they were listed
            int fn[16], fs[16], sp = 0, nextleaf = 0;
            double val[16];
            fn[0] = bn; fs[0] = 0; sp = 1;
            double ret = 0.0;
            while (sp > 0) {
                const int m = fn[sp - 1];
                if (m <= 128) {
                    ret = leaf_sum[nextleaf++];
                    sp--;
                    // hand the value to the parent
                    while (sp > 0) {
                        if (fs[sp - 1] == 1) {  // parent waits for its left value
                            val[sp - 1] = ret;
                            fs[sp - 1] = 2;
                            int h = fn[sp - 1] / 2;
                            h -= h % 8;
                            fn[sp] = fn[sp - 1] - h; fs[sp] = 0; sp++;
                            break;
                        } else {  // fs == 2: right value arrived
                            ret = val[sp - 1] + ret;
                            sp--;
                        }
                    }
                } else {
                    int h = m / 2;
                    h -= h % 8;
                    fs[sp - 1] = 1;
                    fn[sp] = h; fs[sp] = 0; sp++;
                }
            }
            total = (b0 == 0) ? ret : total + ret;
        }
    }
    __syncthreads();
    if (tid == 0) means[i] = total / (double)nv;  // ndarray.mean: sum / count
}

}  // namespace

// diagonal means first..top for calculate_expected; `m` is a band with dlo <= first,
// dhi >= top; d_means and d_scratch are indexed from diagonal `first`
int pk_launch_expected_means(pk_device_ctx *ctx, const pk_matrix *m, int first, int top,
                             const uint8_t *d_valid, double *d_scratch, double *d_means)
{
    pk_prof_scope prof(ctx, PK_K_BAND);
    hipLaunchKernelGGL(expected_means_kernel, dim3((unsigned)(top - first + 1)), dim3(EXP_THREADS), 0,
                       ctx->stream, m->band, m->ld, m->dlo, m->n, first, top, d_valid, d_scratch,
                       d_means - first);
    PK_HIP(hipGetLastError());
    return PK_OK;
}

int pk_launch_csr_info(pk_device_ctx *ctx, const int32_t *d_indptr, const int32_t *d_indices,
                       const double *d_data, int64_t nnz, int n, unsigned long long *d_info6,
                       uint8_t *d_valid_raw, uint8_t *d_valid_bal, const double *d_bias, int upper)
{
    pk_prof_scope prof(ctx, PK_K_BAND);
    PK_HIP(hipMemsetAsync(d_info6, 0, 6 * sizeof(unsigned long long), ctx->stream));
    PK_HIP(hipMemsetAsync(d_valid_raw, 0, (size_t)n, ctx->stream));
    PK_HIP(hipMemsetAsync(d_valid_bal, 0, (size_t)n, ctx->stream));
    if (nnz > 0) {
        const int64_t blocks = (nnz + 255) / 256;
        hipLaunchKernelGGL(csr_info_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, ctx->stream,
                           d_indptr, d_indices, d_data, nnz, n, d_info6, d_info6 + 4, d_valid_raw,
                           d_valid_bal, d_bias, upper, d_info6 + 5);
        PK_HIP(hipGetLastError());
    }
    return PK_OK;
}

int pk_launch_counts_to_f64(pk_device_ctx *ctx, const int32_t *d_src, double *d_dst, int64_t n)
{
    if (n > 0) {
        hipLaunchKernelGGL(counts_to_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_src,
                           d_dst, n);
        PK_HIP(hipGetLastError());
    }
    return PK_OK;
}

// two-pass ordered compaction of the candidate flags; fills *total and, on the
// second call (ox != nullptr), the coordinate arrays
int pk_launch_candidates(pk_device_ctx *ctx, const pk_matrix *raw, int lower, int upper,
                         const int64_t *d_kstar, const double *d_bg, const double *d_w,
                         const double *d_mustar, int64_t n_mustar, int64_t *d_total,
                         int64_t *d_amb, int32_t *ox, int32_t *oy)
{
    const int64_t Q = (int64_t)(upper - lower + 1) * raw->n;
    if (Q <= 0) return PK_OK;
    pk_prof_scope prof(ctx, PK_K_COMPACT);
    const int64_t nblocks = (Q + CTILE - 1) / CTILE;
    int rc = pk_ctx_reserve_scan(ctx, sizeof(int64_t) * (size_t)nblocks);
    if (rc) return rc;
    if (!ox) {
        hipLaunchKernelGGL(cand_count_kernel, dim3((unsigned)nblocks), dim3(CB), 0, ctx->stream,
                           raw->band, raw->ld, raw->dlo, raw->n, lower, Q, d_kstar, d_bg, d_w,
                           d_mustar, n_mustar, ctx->scan_scratch, d_amb);
        hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream,
                           ctx->scan_scratch, nblocks, d_total);
    } else {
        hipLaunchKernelGGL(cand_scatter_kernel, dim3((unsigned)nblocks), dim3(CB), 0, ctx->stream,
                           raw->band, raw->ld, raw->dlo, raw->n, lower, Q, d_kstar, d_bg, d_w,
                           d_mustar, n_mustar, ctx->scan_scratch, ox, oy);
    }
    PK_HIP(hipGetLastError());
    return PK_OK;
}

int pk_launch_band_build(pk_device_ctx *ctx, pk_matrix *m, const int32_t *d_indptr,
                         const int32_t *d_indices, const double *d_data, int64_t nnz, int filter,
                         const double *d_bias, int upper)
{
    pk_prof_scope prof(ctx, PK_K_BAND);
    const size_t bytes = (size_t)(m->dhi - m->dlo + 1) * m->ld * sizeof(double);
    PK_HIP(hipMemsetAsync(m->band, 0, bytes, ctx->stream));
    if (nnz > 0) {
        hipLaunchKernelGGL(band_build_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0,
                           ctx->stream, d_indptr, d_indices, d_data, nnz, m->n, m->dlo, m->dhi,
                           m->ld, m->band, filter, d_bias, upper);
        PK_HIP(hipGetLastError());
    }
    return PK_OK;
}

// ---- short lists: everything behind the forest in ONE launch ------------------------------------
// The regime the CLI runs in -- 10^3 .. 10^5 candidates per chromosome -- pays for launches, not for
// work: behind the forest kernel stood a memset, four kernels (batch counts, survivors per block, scan,
// scatter) and the packing of the reply, each a few microseconds of device time behind ~5 us of
// dependent-launch latency (round 5: 20 us of kernels + ~25 us of gaps in a 123-us call on 1 001
// candidates).  One workgroup does all of it for lists of up to 2^14 candidates (measured: 1 001
// candidates 20 -> 8 us of kernels and four launches fewer, 123 -> 100 us per call; one workgroup has nobody
// to hide its loads' latency behind -- on 10^5 candidates it took 170 us where the four kernels take 24,
// so longer lists keep those):
//   1. groups of 64 candidates, wave by wave (coalesced): the survivors' mask of a group (ballot) parked
//      in LDS, the batch counts by one LDS add per group and batch it touches (scoreUtils.py:104-108);
//   2. the keep mask of every group: survivor, batch with more than one survivor, p > thre (:108-110);
//   3. an exclusive scan of the groups' counts (thread t owns four groups; wave shuffles);
//   4. the kept candidates written in candidate order (x, y, p, M[x, y]: :118-121) -- and, when they fit,
//      into the reply's inline records as well -- with the reply's header words.
constexpr int CS_MAX_N = 1 << 14;
constexpr int CS_GROUPS = CS_MAX_N / 64;   // 256
constexpr int CS_MAX_BATCHES = 1024;
constexpr size_t CS_LDS = (size_t)CS_GROUPS * (8 + 8 + 4) + (size_t)CS_MAX_BATCHES * 4 + 16 * 8 + 64;
__global__ void __launch_bounds__(1024) compact_small_kernel(
    const uint8_t *__restrict__ status, const double *__restrict__ prob, int64_t batch, double thre, int N,
    const int32_t *__restrict__ xs, const int32_t *__restrict__ ys, const double *__restrict__ band, int64_t ld,
    int dlo, int dhi, int32_t *__restrict__ ox, int32_t *__restrict__ oy, double *__restrict__ op,
    double *__restrict__ osig, int64_t *__restrict__ n_out, int32_t *__restrict__ batch_cnt_out,
    const long long *__restrict__ dbg3, char *__restrict__ ret, int with_records,
    const unsigned *__restrict__ split_cnt, int split_k)
{
    extern __shared__ __attribute__((aligned(16))) char cs_lds[];
    unsigned long long *stmask = reinterpret_cast<unsigned long long *>(cs_lds);
    unsigned long long *kmask = stmask + CS_GROUPS;
    int *koff = reinterpret_cast<int *>(kmask + CS_GROUPS);
    int *bcnt = koff + CS_GROUPS;
    long long *wsum = reinterpret_cast<long long *>(bcnt + CS_MAX_BATCHES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = (N + 63) >> 6;
    const unsigned ubatch = batch > 0x40000000 ? 0x40000000u : (unsigned)batch;   // (N < 2^14: 32-bit divisions)
    const int nb = (int)(((unsigned)N + ubatch - 1u) / ubatch);
    for (int b = tid; b < nb; b += 1024) bcnt[b] = 0;
    __syncthreads();
    // 1. survivors.  (Eight groups of a wave per trip, their loads in flight together: one workgroup has
    // nobody else to hide a load's latency behind -- one group per trip took 180 us on 10^5 candidates.)
    constexpr int U = 8;
    for (int g0 = wave; g0 < G; g0 += 16 * U) {
        uint8_t stv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int c = ((g0 + 16 * u) << 6) + lane;
            stv[u] = c < N ? status[c] : (uint8_t)0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int g = g0 + 16 * u;
            const unsigned long long m = __ballot(stv[u] != 0);
            if (lane == 0 && g < G) {
                stmask[g] = m;
                if (m) {
                    const unsigned c0 = (unsigned)g << 6;
                    const int b0 = (int)(c0 / ubatch);
                    // (batch >= 64: a group touches two batches at most)
                    const int64_t edge = (int64_t)(b0 + 1) * ubatch - c0;   // candidates of the group in batch b0
                    const unsigned long long lo = edge >= 64 ? m : (m & ((1ull << edge) - 1ull));
                    if (lo) atomicAdd(&bcnt[b0], __popcll(lo));
                    if (m ^ lo) atomicAdd(&bcnt[b0 + 1], __popcll(m ^ lo));
                }
            }
        }
    }
    __syncthreads();
    for (int b = tid; b < nb; b += 1024) batch_cnt_out[b] = bcnt[b];
    // 2. keep masks
    for (int g0 = wave; g0 < G; g0 += 16 * U) {
        double pv[U];
        bool stb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int g = g0 + 16 * u;
            const int c = (g << 6) + lane;
            stb[u] = g < G && ((stmask[g < G ? g : 0] >> lane) & 1ull);
            pv[u] = stb[u] ? prob[c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int g = g0 + 16 * u;
            const int c = (g << 6) + lane;
            const bool keep = stb[u] && bcnt[(unsigned)c / ubatch] > 1 && pv[u] > thre;
            const unsigned long long km = __ballot(keep);
            if (lane == 0 && g < G) {
                kmask[g] = km;
                koff[g] = __popcll(km);
            }
        }
    }
    __syncthreads();
    // 3. exclusive scan of koff[0 .. G): thread t owns groups 4t .. 4t + 3
    int own[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        own[j] = 4 * tid + j < G ? koff[4 * tid + j] : 0;
        s += own[j];
    }
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int v = (int)wsum[i];
        if (i < wave) woff += v;
        total += v;
    }
    int run = woff + incl - s;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (4 * tid + j < G) koff[4 * tid + j] = run;
        run += own[j];
    }
    __syncthreads();
    // 4. the kept candidates, in candidate order; the reply
    const bool inl = with_records && total <= PK_RET_INLINE;
    int32_t *rx = reinterpret_cast<int32_t *>(ret + PK_RET_HEAD), *ry = rx + PK_RET_INLINE;
    double *rp = reinterpret_cast<double *>(ry + PK_RET_INLINE), *rs = rp + PK_RET_INLINE;
    for (int g = wave; g < G; g += 16) {
        const unsigned long long km = kmask[g];
        if (km == 0ull) continue;   // (uniform)
        if ((km >> lane) & 1ull) {
            const int c = (g << 6) + lane;
            const int pos = koff[g] + __popcll(km & ((1ull << lane) - 1ull));
            const int x = xs[c], y = ys[c];
            const double p = prob[c];
            const int d = y - x;
            // signal = M[row, col] (scoreUtils.py:120)
            const double sg = (d >= dlo && d <= dhi) ? band[(int64_t)(d - dlo) * ld + x] : 0.0;
            ox[pos] = x;
            oy[pos] = y;
            op[pos] = p;
            osig[pos] = sg;
            if (inl) {
                rx[pos] = x;
                ry[pos] = y;
                rp[pos] = p;
                rs[pos] = sg;
            }
        }
    }
    if (tid == 0) {
        *n_out = total;
        reinterpret_cast<long long *>(ret)[0] = total;
    }
    if (tid >= 1 && tid < 4) reinterpret_cast<long long *>(ret)[tid] = dbg3[tid - 1];
    if (tid == 4 && split_k > 0) {
        long long parked = 0;
        for (int i = 0; i < split_k; i++) parked += split_cnt[i];
        reinterpret_cast<long long *>(ret)[4] = parked;
    }
}

int pk_launch_compact(pk_device_ctx *ctx, const pk_matrix *m, pk_cands *cd, double thre,
                      int64_t batch, int with_records, bool *reply_packed)
{
    const int64_t N = cd->N;
    if (reply_packed) *reply_packed = false;
    if (N == 0) {
        PK_HIP(hipMemsetAsync(cd->n_out_dev, 0, sizeof(int64_t), ctx->stream));
        return PK_OK;
    }
    pk_prof_scope prof(ctx, PK_K_COMPACT);
    if (reply_packed && N <= CS_MAX_N && batch >= 64 && (N + batch - 1) / batch <= CS_MAX_BATCHES && cd->opt.compact_small) {
        PK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(compact_small_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)CS_LDS));
        hipLaunchKernelGGL(compact_small_kernel, dim3(1), dim3(1024), CS_LDS, ctx->stream, cd->status, cd->prob, batch,
                           thre, (int)N, cd->x, cd->y, m->band, m->ld, m->dlo, m->dhi, cd->ox, cd->oy, cd->op, cd->osig,
                           cd->n_out_dev, cd->batch_cnt, ctx->dbg_buf + 65533, ctx->d_ret, with_records,
                           ctx->split_cnt, ctx->split_cnt ? ctx->split_k : 0);
        PK_HIP(hipGetLastError());
        *reply_packed = true;
        return PK_OK;
    }
    const int64_t nb = (N + batch - 1) / batch;
    PK_HIP(hipMemsetAsync(cd->batch_cnt, 0, sizeof(int32_t) * (size_t)nb, ctx->stream));
    hipLaunchKernelGGL(batch_count_kernel,
                       dim3((unsigned)((N + 256 * BC_ITEMS - 1) / (256 * BC_ITEMS))), dim3(256), 0,
                       ctx->stream, cd->status, N, batch, cd->batch_cnt);
    const int64_t nblocks = (N + CTILE - 1) / CTILE;
    int rc = pk_ctx_reserve_scan(ctx, sizeof(int64_t) * (size_t)nblocks);
    if (rc) return rc;
    hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)nblocks), dim3(CB), 0, ctx->stream,
                       cd->status, cd->prob, cd->batch_cnt, batch, thre, N, ctx->scan_scratch);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream,
                       ctx->scan_scratch, nblocks, cd->n_out_dev);
    hipLaunchKernelGGL(compact_scatter_kernel, dim3((unsigned)nblocks), dim3(CB), 0, ctx->stream,
                       cd->status, cd->prob, cd->batch_cnt, batch, thre, N, ctx->scan_scratch,
                       cd->x, cd->y, m->band, m->ld, m->dlo, m->dhi, cd->ox, cd->oy, cd->op,
                       cd->osig);
    PK_HIP(hipGetLastError());
    return PK_OK;
}
