// pk_forest.hip -- Random-Forest predict_proba[:,1] for gfx950 (CDNA4).
//
// Replaces model.predict_proba(fea)[:, 1] at peakachu/scoreUtils.py:109
// (sklearn ForestClassifier.predict_proba -> Tree._apply_dense): float32
// features, axis-aligned splits `x[feature] <= threshold` (NaN routed by
// missing_go_to_left), leaf class-1 fractions added in tree order in float64
// and divided by T.
//
// Design:
//  * one candidate per lane; the candidate's F float32 features sit in LDS
//    as a [F][BLK] tile, so the per-node feature fetch `fea[f*BLK + lane]`
//    is bank-conflict-free whatever f each lane asks for;
//  * nodes are 8-byte words in preorder (left child = next word), thresholds
//    pre-rounded down to float32 (identical decisions for float32 x); the
//    forest (a few MB) is served from L2 / Infinity Cache;
//  * a lane walks ILP trees at once: ILP independent load chains in flight
//    hide the L2 latency at the 4-waves-per-CU occupancy the feature tile
//    allows; leaf values are then added in tree order, so the float64 sum is
//    the same sequential sum sklearn computes;
//  * optional LDS staging of the trees (forest_lds): a workgroup streams
//    groups of trees through LDS, all lanes walk the staged trees, which
//    turns the dependent L2 loads into LDS reads.
#include "pk_common.h"

namespace {

__device__ __forceinline__ double node_as_double(uint2 n)
{
    return __longlong_as_double(((long long)n.y << 32) | (long long)n.x);
}

// one step of one chain: `n` is the interior node at `idx`
__device__ __forceinline__ void step(uint2 n, const float *fea, int blk, int lane, int &idx,
                                     bool &at_leaf)
{
    const unsigned pk = n.y;
    const float x = fea[(pk & ((1u << PK_NODE_FEAT_BITS) - 1)) * blk + lane];
    const float thr = __uint_as_float(n.x);
    const bool miss = (pk >> PK_NODE_MISS_BIT) & 1u;
    const bool go_left = (x <= thr) || ((x != x) && miss);
    at_leaf = go_left ? ((pk >> PK_NODE_LLEAF_BIT) & 1u) : ((pk >> PK_NODE_RLEAF_BIT) & 1u);
    idx += go_left ? 1 : (int)(pk >> PK_NODE_ROFF_SHIFT);
}

// ------------------------------------------------------------------------
// v1: nodes read through the cache hierarchy, ILP trees in flight per lane.
// ------------------------------------------------------------------------
template <int ILP>
__global__ void forest_l2_kernel(const uint2 *__restrict__ nodes, const int32_t *__restrict__ root,
                                 int T, int F, const float *__restrict__ tiles,
                                 const uint8_t *__restrict__ status, int64_t c0, int64_t cn,
                                 double *__restrict__ prob)
{
    extern __shared__ __attribute__((aligned(16))) float fea[];  // [F][blk]
    const int blk = blockDim.x;
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    {
        const float4 *src = reinterpret_cast<const float4 *>(tiles + (size_t)tile * F * blk);
        float4 *dst = reinterpret_cast<float4 *>(fea);
        const int nvec = F * blk / 4;
        for (int i = lane; i < nvec; i += blk) dst[i] = src[i];
    }
    __syncthreads();
    const int64_t local = tile * blk + lane;
    if (local >= cn) return;
    const int64_t c = c0 + local;
    if (!status[c]) {
        prob[c] = 0.0;
        return;
    }
    double acc = 0.0;
    for (int t = 0; t < T; t += ILP) {
        int idx[ILP];
        bool leaf[ILP];
#pragma unroll
        for (int k = 0; k < ILP; k++) {
            idx[k] = root[min(t + k, T - 1)];
            leaf[k] = false;
        }
        bool all_done = false;
        while (!all_done) {
            uint2 nd[ILP];
#pragma unroll
            for (int k = 0; k < ILP; k++) nd[k] = nodes[idx[k]];
            all_done = true;
#pragma unroll
            for (int k = 0; k < ILP; k++) {
                if (!leaf[k]) step(nd[k], fea, blk, lane, idx[k], leaf[k]);
                all_done = all_done && leaf[k];
            }
        }
        double v[ILP];
#pragma unroll
        for (int k = 0; k < ILP; k++) v[k] = node_as_double(nodes[idx[k]]);
#pragma unroll
        for (int k = 0; k < ILP; k++)
            if (t + k < T) acc += v[k];  // tree order: sklearn's sequential sum
    }
    prob[c] = acc / (double)T;
}

// ------------------------------------------------------------------------
// v2: trees staged through LDS.  The workgroup copies a group of whole trees
// (as many as fit `tree_lds_words`) into LDS, every lane walks them, repeat.
// Trees larger than the staging buffer are walked from global memory.
// ------------------------------------------------------------------------
template <int ILP>
__global__ void forest_lds_kernel(const uint2 *__restrict__ nodes, const int32_t *__restrict__ root,
                                  int T, int F, const float *__restrict__ tiles,
                                  const uint8_t *__restrict__ status, int64_t c0, int64_t cn,
                                  double *__restrict__ prob, int tree_lds_words)
{
    extern __shared__ __attribute__((aligned(16))) float fea[];  // [F][blk] then tree buffer
    const int blk = blockDim.x;
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    uint2 *tbuf = reinterpret_cast<uint2 *>(fea + (size_t)F * blk);
    {
        const float4 *src = reinterpret_cast<const float4 *>(tiles + (size_t)tile * F * blk);
        float4 *dst = reinterpret_cast<float4 *>(fea);
        const int nvec = F * blk / 4;
        for (int i = lane; i < nvec; i += blk) dst[i] = src[i];
    }
    const int64_t local = tile * blk + lane;
    const bool valid = local < cn;
    const int64_t c = c0 + (valid ? local : 0);
    const bool active = valid && status[c];
    double acc = 0.0;
    int t = 0;
    while (t < T) {  // T, root[] are uniform: every thread takes the same trips
        // group = trees t..t1-1 whose nodes [root[t], root[t1]) fit the buffer
        const int g0 = root[t];
        int t1 = t + 1;
        while (t1 < T && root[t1 + 1] - g0 <= tree_lds_words) t1++;
        const int gwords = root[t1] - g0;
        const bool staged = gwords <= tree_lds_words;
        __syncthreads();  // previous group fully walked (and feature tile landed)
        if (staged) {
            for (int i = lane; i < gwords; i += blk) tbuf[i] = nodes[g0 + i];
        }
        __syncthreads();
        if (active) {
            const uint2 *base = staged ? (const uint2 *)tbuf : nodes + g0;
            for (int tt = t; tt < t1; tt += ILP) {
                int idx[ILP];
                bool leaf[ILP];
#pragma unroll
                for (int k = 0; k < ILP; k++) {
                    idx[k] = root[min(tt + k, t1 - 1)] - g0;
                    leaf[k] = false;
                }
                bool all_done = false;
                while (!all_done) {
                    uint2 nd[ILP];
#pragma unroll
                    for (int k = 0; k < ILP; k++) nd[k] = base[idx[k]];
                    all_done = true;
#pragma unroll
                    for (int k = 0; k < ILP; k++) {
                        if (!leaf[k]) step(nd[k], fea, blk, lane, idx[k], leaf[k]);
                        all_done = all_done && leaf[k];
                    }
                }
                double v[ILP];
#pragma unroll
                for (int k = 0; k < ILP; k++) v[k] = node_as_double(base[idx[k]]);
#pragma unroll
                for (int k = 0; k < ILP; k++)
                    if (tt + k < t1) acc += v[k];
            }
        }
        t = t1;
    }
    if (valid) prob[c] = active ? acc / (double)T : 0.0;
}

// row-major [N][F] float32 -> [tile][F][blk] tiles (pk_predict's input path)
__global__ void tile_rows_kernel(const float *__restrict__ rows, int64_t N, int F,
                                 float *__restrict__ tiles, int blk)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * F) return;
    const int64_t c = i / F;
    const int f = (int)(i - c * F);
    const int64_t tile = c / blk;
    const int lane = (int)(c - tile * blk);
    tiles[((size_t)tile * F + f) * blk + lane] = rows[i];
}

template <typename KernelT>
int set_max_lds(KernelT k, size_t bytes)
{
    PK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PK_OK;
}

}  // namespace

// Candidates per feature tile: as many 64-lane waves as fit F*4 bytes each in
// the 160 KiB LDS of a CU (keeping 4 KiB spare), at most 256.
int pk_forest_tile_width(int F)
{
    int blk = (int)((156 * 1024) / ((size_t)F * 4)) / 64 * 64;
    if (blk > 256) blk = 256;
    if (blk < 64) blk = 0;  // F too large for an LDS-resident tile
    return blk;
}

int pk_launch_tile_rows(pk_device_ctx *ctx, const float *d_rows, int64_t N, int F, float *tiles,
                        int blk)
{
    if (N <= 0) return PK_OK;
    const int64_t total = N * F;
    hipLaunchKernelGGL(tile_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       ctx->stream, d_rows, N, F, tiles, blk);
    PK_HIP(hipGetLastError());
    return PK_OK;
}

#define PK_LAUNCH_L2(ILP)                                                                     \
    do {                                                                                      \
        int rc__ = set_max_lds(forest_l2_kernel<ILP>, lds);                                   \
        if (rc__) return rc__;                                                                \
        hipLaunchKernelGGL(forest_l2_kernel<ILP>, dim3(grid), dim3(blk), lds, ctx->stream,    \
                           f->nodes, f->root, f->T, f->F, tiles, d_status, c0, cn, d_prob);   \
    } while (0)

#define PK_LAUNCH_LDS(ILP)                                                                    \
    do {                                                                                      \
        int rc__ = set_max_lds(forest_lds_kernel<ILP>, lds);                                  \
        if (rc__) return rc__;                                                                \
        hipLaunchKernelGGL(forest_lds_kernel<ILP>, dim3(grid), dim3(blk), lds, ctx->stream,   \
                           f->nodes, f->root, f->T, f->F, tiles, d_status, c0, cn, d_prob,    \
                           tree_words);                                                       \
    } while (0)

int pk_launch_forest(pk_device_ctx *ctx, const pk_forest *f, const float *tiles, int blk,
                     const uint8_t *d_status, int64_t c0, int64_t cn, double *d_prob)
{
    if (cn <= 0) return PK_OK;
    pk_prof_scope prof(ctx, PK_K_FOREST);
    const unsigned grid = (unsigned)((cn + blk - 1) / blk);
    const size_t fea_bytes = (size_t)f->F * blk * sizeof(float);
    const int ilp = (int)g_opt.forest_ilp;
    if (g_opt.forest_lds > 0) {
        // whatever LDS the feature tile leaves (one workgroup per CU)
        size_t room = (size_t)160 * 1024 - fea_bytes;
        if ((size_t)g_opt.forest_lds * 1024 < room) room = (size_t)g_opt.forest_lds * 1024;
        const int tree_words = (int)(room / sizeof(uint2));
        const size_t lds = fea_bytes + (size_t)tree_words * sizeof(uint2);
        switch (ilp) {
        case 1: PK_LAUNCH_LDS(1); break;
        case 2: PK_LAUNCH_LDS(2); break;
        case 8: PK_LAUNCH_LDS(8); break;
        default: PK_LAUNCH_LDS(4); break;
        }
    } else {
        const size_t lds = fea_bytes;
        switch (ilp) {
        case 1: PK_LAUNCH_L2(1); break;
        case 2: PK_LAUNCH_L2(2); break;
        case 8: PK_LAUNCH_L2(8); break;
        default: PK_LAUNCH_L2(4); break;
        }
    }
    PK_HIP(hipGetLastError());
    return PK_OK;
}
