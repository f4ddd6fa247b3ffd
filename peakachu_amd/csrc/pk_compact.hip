// pk_compact.hip -- band construction and the tail of Chromosome.score for
// gfx950: the reference's batch rule, the p > thre filter, deterministic
// stream compaction in candidate order and the signal re-gather
// (peakachu/scoreUtils.py:104-121).  Integer / byte work, HBM-bound, tiny
// next to the extract and forest kernels.
#include "pk_common.h"

namespace {

// ---- CSR -> diagonal-major band: one thread per stored entry ------------
__global__ void band_build_kernel(const int32_t *__restrict__ indptr,
                                  const int32_t *__restrict__ indices,
                                  const double *__restrict__ data, int64_t nnz, int n, int dlo,
                                  int dhi, int64_t ld, double *__restrict__ band)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    // row of entry e: largest r with indptr[r] <= e
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)indptr[mid] <= e) lo = mid;
        else hi = mid;
    }
    const int r = lo;
    const int col = indices[e];
    const int k = col - r;
    if (k < dlo || k > dhi || col < 0 || col >= n) return;
    band[(int64_t)(k - dlo) * ld + r] = data[e];
}

// ---- survivors per reference batch (scoreUtils.py:104-108) --------------
// One workgroup counts 4096 consecutive candidates: wave ballots, one LDS
// add per wave, one global atomic per workgroup and batch it touches (a
// workgroup straddles at most two batches when batch >= 4096; smaller
// batches fall back to per-candidate atomics).
constexpr int BC_ITEMS = 16;
__global__ void batch_count_kernel(const uint8_t *__restrict__ status, int64_t N, int64_t batch,
                                   int32_t *__restrict__ batch_cnt)
{
    __shared__ int cnt[2];
    const int64_t base = (int64_t)blockIdx.x * (256 * BC_ITEMS);
    const int64_t b0 = base / batch;
    if (batch < 256 * BC_ITEMS) {
        for (int i = 0; i < BC_ITEMS; i++) {
            const int64_t c = base + (int64_t)i * 256 + threadIdx.x;
            if (c < N && status[c]) atomicAdd(&batch_cnt[c / batch], 1);
        }
        return;
    }
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    int mine0 = 0, mine1 = 0;
    for (int i = 0; i < BC_ITEMS; i++) {
        const int64_t c = base + (int64_t)i * 256 + threadIdx.x;
        if (c < N && status[c]) {
            if (c / batch == b0) mine0++;
            else mine1++;
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
__global__ void compact_scan_kernel(int64_t *__restrict__ block_cnt, int64_t nblocks,
                                    int64_t *__restrict__ n_out)
{
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (nblocks + blockDim.x - 1) / blockDim.x;
    const int64_t b0 = (int64_t)tid * per;
    const int64_t b1 = b0 + per < nblocks ? b0 + per : nblocks;
    int64_t s = 0;
    for (int64_t b = b0; b < b1; b++) s += block_cnt[b];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int64_t run = 0;
        for (int i = 0; i < (int)blockDim.x; i++) {
            const int64_t v = part[i];
            part[i] = run;
            run += v;
        }
        *n_out = run;
    }
    __syncthreads();
    int64_t run = part[tid];
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

}  // namespace

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
                         const int32_t *d_indices, const double *d_data, int64_t nnz)
{
    pk_prof_scope prof(ctx, PK_K_BAND);
    const size_t bytes = (size_t)(m->dhi - m->dlo + 1) * m->ld * sizeof(double);
    PK_HIP(hipMemsetAsync(m->band, 0, bytes, ctx->stream));
    if (nnz > 0) {
        hipLaunchKernelGGL(band_build_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0,
                           ctx->stream, d_indptr, d_indices, d_data, nnz, m->n, m->dlo, m->dhi,
                           m->ld, m->band);
        PK_HIP(hipGetLastError());
    }
    return PK_OK;
}

int pk_launch_compact(pk_device_ctx *ctx, const pk_matrix *m, pk_cands *cd, double thre,
                      int64_t batch)
{
    const int64_t N = cd->N;
    if (N == 0) {
        PK_HIP(hipMemsetAsync(cd->n_out_dev, 0, sizeof(int64_t), ctx->stream));
        return PK_OK;
    }
    pk_prof_scope prof(ctx, PK_K_COMPACT);
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
