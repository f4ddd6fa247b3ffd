// pk_forest.hip -- Random-Forest predict_proba[:,1] for gfx950 (CDNA4).
//
// Replaces model.predict_proba(fea)[:, 1] at peakachu/scoreUtils.py:109
// (sklearn ForestClassifier.predict_proba -> Tree._apply_dense): float32
// features, axis-aligned splits `x[feature] <= threshold` (NaN routed by
// missing_go_to_left), leaf class-1 fractions added in tree order in float64
// and divided by T.
//
// Design (the measurements: EXPERIMENTS.md §4.2):
//  * one candidate per lane; the candidate's F float32 features sit in LDS as a
//    [F][128] tile, so the per-node feature fetch `fea[f*128 + lane]` is
//    bank-conflict-free whatever f each lane asks for;
//  * nodes are 8-byte words in preorder (left child = next word), thresholds
//    pre-rounded down to float32 (identical decisions for float32 x), leaves worth
//    exactly 0 or 1 folded into their parent; the forest (1-2 MB) lives in L2;
//  * leaf values are added in tree order, so the float64 sum is the sequential sum
//    sklearn computes.
// Kernels in this file:
//    forest_lds_kernel<SLOTS, PRUNE>  (default) trees streamed through LDS in groups,
//        16 waves per CU, two barriers per group; PRUNE = exact early termination
//    forest_gmem_kernel<ILP>          no LDS at all, for feature counts whose tile
//        cannot share the LDS with trees (w = 11 ... 15), and for option forest_lds = 0
// (rounds 1-4 also carried forest_pipe_kernel, a barrier-free variant of the LDS kernel, and
// forest_l2_kernel, the LDS-tile / L2-node variant: no model reached either by default --
// DESIGN.md "Forest routes" -- and both were removed in round 5; EXPERIMENTS.md keeps their numbers)
#include "pk_common.h"

namespace {

__device__ __forceinline__ double word_as_double(uint2 n)
{
    return __longlong_as_double(((long long)n.y << 32) | (long long)n.x);
}

// value of a finished chain: `kind` of the child it stepped to, `w` that
// child's word (only meaningful for a stored leaf)
__device__ __forceinline__ double leaf_value(unsigned kind, uint2 w)
{
    return kind == PK_KIND_LEAF ? word_as_double(w) : (kind == PK_KIND_ONE ? 1.0 : 0.0);
}

// sklearn's split: x <= thr goes left; NaN goes where missing_go_to_left says.
// Bitwise, so no branch is generated for the (rare) NaN case.
__device__ __forceinline__ bool goes_left(float x, uint2 n)
{
    const bool le = x <= __uint_as_float(n.x);
    const bool nan_left = (x != x) & (((n.y >> PK_NODE_MISS_BIT) & 1u) != 0);
    return le | nan_left;
}

__device__ __forceinline__ unsigned node_feat(uint2 n)
{
    return (n.y & PK_NODE_FEAT_MASK) >> PK_NODE_FEAT_SHIFT;
}
// right-child offset of the word at absolute index `idx`
__device__ __forceinline__ int node_roff(uint2 n, const int32_t *__restrict__ big_roff, int idx)
{
    const int r = (int)(n.y >> PK_NODE_ROFF_SHIFT);
    return (r == PK_NODE_ROFF_BIG && big_roff) ? big_roff[idx] : r;
}
__device__ __forceinline__ unsigned node_kind(uint2 n, bool left)
{
    return (n.y >> (left ? PK_NODE_LKIND_SHIFT : PK_NODE_RKIND_SHIFT)) & 3u;
}

// ------------------------------------------------------------------------
// v1b: no LDS at all, for feature counts whose tile leaves no room for trees
// (w = 11: a [529][64] tile is 135 KB, which would pin the LDS kernels to one
// wave per CU).  Features are read from the tile in global memory (it stays
// in L2 / Infinity Cache; lanes agree on the feature near the root, so those
// reads coalesce), nodes from L2; 256-thread workgroups at full occupancy hide
// the latency, ILP chains per lane add memory-level parallelism.
// ------------------------------------------------------------------------
template <int ILP>
__global__ __launch_bounds__(256) void forest_gmem_kernel(
    const uint2 *__restrict__ nodes, const int32_t *__restrict__ root,
    const int32_t *__restrict__ big_roff, int T, int F, const float *__restrict__ tiles, int blk,
    const uint8_t *__restrict__ status, int64_t c0, int64_t cn, double *__restrict__ prob)
{
    const int64_t local = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (local >= cn) return;
    const int64_t c = c0 + local;
    if (!status[c]) {
        prob[c] = 0.0;
        return;
    }
    const int64_t tile = local / blk;
    const float *fea = tiles + (size_t)tile * F * blk + (local - tile * blk);  // + f * blk
    double acc = 0.0;
    for (int t = 0; t < T; t += ILP) {
        int idx[ILP];
        unsigned kind[ILP];
#pragma unroll
        for (int k = 0; k < ILP; k++) {
            idx[k] = root[min(t + k, T - 1)];
            kind[k] = PK_KIND_NODE;
        }
        bool all_done = false;
        while (!all_done) {
            uint2 nd[ILP];
            float x[ILP];
#pragma unroll
            for (int k = 0; k < ILP; k++) nd[k] = nodes[idx[k]];
#pragma unroll
            for (int k = 0; k < ILP; k++) {
                // a finished chain sits on a leaf word (arbitrary bits): it must not
                // be decoded into a feature index -- this is GLOBAL memory
                const unsigned f = kind[k] == PK_KIND_NODE ? node_feat(nd[k]) : 0u;
                x[k] = fea[(size_t)f * blk];
            }
            all_done = true;
#pragma unroll
            for (int k = 0; k < ILP; k++) {
                if (kind[k] == PK_KIND_NODE) {
                    const bool gl = goes_left(x[k], nd[k]);
                    kind[k] = node_kind(nd[k], gl);
                    idx[k] += gl ? 1 : node_roff(nd[k], big_roff, idx[k]);
                }
                all_done = all_done && (kind[k] != PK_KIND_NODE);
            }
        }
        double v[ILP];
#pragma unroll
        for (int k = 0; k < ILP; k++) v[k] = leaf_value(kind[k], nodes[idx[k]]);
#pragma unroll
        for (int k = 0; k < ILP; k++)
            if (t + k < T) acc += v[k];  // tree order: sklearn's sequential sum
    }
    prob[c] = acc / (double)T;
}

// ------------------------------------------------------------------------
// v2: trees streamed through LDS.
//
// A workgroup owns LDS_C = 128 candidates (feature tile [F][128], 62 KB at
// F = 121) and 2*SLOTS waves.  Wave pair s (= wave >> 1) is "tree slot" s: its
// two waves cover the 128 candidates and walk the same tree, so node reads
// near the root are LDS broadcasts.  Trees are copied into LDS in groups of up
// to SLOTS whole trees; slot s walks tree s of the group, leaf values are
// parked in LDS and the slot-0 lanes add them in tree order -- the same
// sequential float64 sum sklearn computes.
//
// One LDS round trip per tree level: as soon as a node word is known, the
// lane issues the feature read AND the reads of both child words (left =
// next word, right = roff words on); when they return it picks the child, so
// the next level's node never costs a second dependent LDS access.
//
// The walk is instruction-issue bound (about 16 VALU per level), so
// parallelism comes from waves (SLOTS up to 8 = 16 waves per CU), not from
// several chains per lane, and the NaN routing (missing_go_to_left) is only
// compiled into a slow path taken by waves that actually hold a NaN feature.
//
// Staging overlaps the walk: each thread loads its share of the NEXT group
// into registers before walking the current one and writes it to LDS after
// the barrier that ends the walk.
// ------------------------------------------------------------------------
constexpr int LDS_C = 128;

// Both child words are fetched together with the feature: one LDS round trip
// per level (20 bytes read).  Fetching only the chosen child afterwards (two
// dependent round trips, 12 bytes) was measured 8-10 % slower at 16 waves/CU.
typedef __attribute__((address_space(3))) const float pk_lds_f32;

template <bool WITH_NAN>
__device__ __forceinline__ double walk_tree_lds(const char *fea_b, int cl4, const char *a)
{
    // a: LDS byte address of the current node word
    uint2 cur = *reinterpret_cast<const uint2 *>(a);
    uint2 nxt;
    unsigned kind;
    do {
        const unsigned pk = cur.y;
        // feature row offset is stored in place and the tile starts at LDS offset 0
        // (checked by the kernel): the address is one v_and_or_b32
        const float x = *reinterpret_cast<pk_lds_f32 *>(
            (__UINTPTR_TYPE__)((pk & PK_NODE_FEAT_MASK) | (unsigned)cl4));
        const char *ra = a + ((pk >> PK_NODE_ROFF_SHIFT) << 3);
        // (predicating these reads on the child having a word -- 46 % are pure
        // leaves -- was measured slower: the exec-mask code costs more than the
        // LDS cycles it saves)
        const uint2 lw = *reinterpret_cast<const uint2 *>(a + 8);
        const uint2 rw = *reinterpret_cast<const uint2 *>(ra);
        bool gl = x <= __uint_as_float(cur.x);
        if (WITH_NAN) gl = gl | ((x != x) & (((pk >> PK_NODE_MISS_BIT) & 1u) != 0));
        kind = (pk >> (gl ? PK_NODE_LKIND_SHIFT : PK_NODE_RKIND_SHIFT)) & 3u;
        nxt.x = gl ? lw.x : rw.x;
        nxt.y = gl ? lw.y : rw.y;
        a = gl ? a + 8 : ra;
        cur = nxt;
    } while (kind == PK_KIND_NODE);
    return leaf_value(kind, nxt);
}

// generic (global-memory) walk of one tree, for trees too large to stage
__device__ __forceinline__ double walk_tree_global(const uint2 *__restrict__ base,
                                                   const int32_t *__restrict__ big_roff, int idx,
                                                   const float *fea, int cl)
{
    unsigned kind;
    uint2 nd;
    do {
        nd = base[idx];
        const float x = fea[node_feat(nd) * LDS_C + cl];
        const bool gl = goes_left(x, nd);
        kind = node_kind(nd, gl);
        idx += gl ? 1 : node_roff(nd, big_roff, idx);
    } while (kind == PK_KIND_NODE);
    return leaf_value(kind, base[idx]);
}

// The next tree (group) travels global -> registers (prefetch, issued before
// the current one is walked) -> LDS (commit, afterwards).  The registers are
// individually named (X-macros), not an array: the compiler demoted arrays to
// scratch memory here.  The loads are unconditional with a clamped index:
// nothing may consume a loaded value before the walk, or the compiler waits
// for the load on the spot.
#define PK_PF6(X) X(0) X(1) X(2) X(3) X(4) X(5)
#define PK_PF8(X) PK_PF6(X) X(6) X(7)
#define PK_PF_DECL(q) uint4 pf##q;
#define PK_PF_LOAD(q) pf##q = pf_src[min(pf_tid + (q) * pf_stride, pf_nv - 1)];
#define PK_PF_STORE(q) \
    if (pf_tid + (q) * pf_stride < pf_nv) pf_dst[pf_tid + (q) * pf_stride] = pf##q;

template <int SLOTS, bool PRUNE>
__global__ __launch_bounds__(LDS_C *SLOTS) void forest_lds_kernel(
    const uint2 *__restrict__ nodes, const int32_t *__restrict__ root,
    const int32_t *__restrict__ big_roff, const int32_t *__restrict__ grp, int n_grp, int T, int F,
    const float *__restrict__ tiles, const uint8_t *__restrict__ status, int64_t c0, int64_t cn,
    double *__restrict__ prob, int tree_words, int dbg, long long *__restrict__ stamps,
    double prune_sum, int warm_ahead)
{
    constexpr int THREADS = LDS_C * SLOTS;
    // 6 x uint4 registers per thread hold the prefetched group (the launcher keeps
    // tree_words <= THREADS * 6 * 2); six, not eight, keeps the kernel at <= 72
    // VGPRs so that one extractor wave (216) fits beside four of its waves on a SIMD
    extern __shared__ __attribute__((aligned(16))) float fea[];  // [F][128] | val | trees
    double *val = reinterpret_cast<double *>(fea + (size_t)F * LDS_C);  // [SLOTS][128]
    // (the leaf-value area has SLOTS+1 rows: the last one holds the "decided" flags
    // of the optional early termination, one int per candidate)
    int *decided = reinterpret_cast<int *>(val + SLOTS * LDS_C);
    uint2 *tbuf = reinterpret_cast<uint2 *>(val + (SLOTS + (PRUNE ? 1 : 0)) * LDS_C);  // tree_words + 2 pad
    const int tid = threadIdx.x;
    const int cl = tid & (LDS_C - 1);
    const int slot = __builtin_amdgcn_readfirstlane(tid >> 7);  // wave-uniform -> SGPR
    constexpr bool prune = PRUNE;  // early termination compiled in (separate instantiation)
    // decided[0..127]: per-candidate flags; decided[128..130]: three rotating vote words
    // (a vote word is set before barrier g, read after it, and cleared two groups ahead,
    // so a clear and a set of the same word are always separated by a barrier).
    // No __syncthreads_and here: it would add static LDS in front of the feature tile.
    if (PRUNE && tid < LDS_C + 3) decided[tid] = 0;
    const int64_t tile = blockIdx.x;
    {
        const float4 *src = reinterpret_cast<const float4 *>(tiles + (size_t)tile * F * LDS_C);
        float4 *dst = reinterpret_cast<float4 *>(fea);
        const int nvec = F * LDS_C / 4;
        for (int i = tid; i < nvec; i += THREADS) dst[i] = src[i];
    }
    const int64_t local = tile * LDS_C + cl;
    const bool valid = local < cn;
    const int64_t c = c0 + (valid ? local : 0);
    const unsigned st = valid ? status[c] : 0;
    // walk_tree_lds addresses the tile as LDS offset 0 (no static LDS in this kernel)
    const bool tile_at_zero =
        (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const void *)fea == 0u;
    if (!tile_at_zero && tid == 0 && stamps) stamps[65535] = 2;
    const bool active = st != 0 && tile_at_zero;
    // a wave holding a candidate with NaN features takes the slow walk
    const bool wave_nan = __any(st == 2);

    // group g = trees [grp[g], grp[g+1]); grp[n_grp+2+g] = staged in LDS or not
    const int32_t *gstaged = grp + n_grp + 2;
    PK_PF6(PK_PF_DECL)  // the next group, in flight / parked in VGPRs
    const int pf_tid = tid, pf_stride = THREADS;
    const uint4 *pf_src;
    uint4 *const pf_dst = reinterpret_cast<uint4 *>(tbuf);
    int pf_nv;
    int t = grp[0], t1 = grp[1];
    if (gstaged[0]) {
        pf_src = reinterpret_cast<const uint4 *>(nodes + root[t]);
        pf_nv = (root[t1] - root[t]) >> 1;
        PK_PF6(PK_PF_LOAD)
        PK_PF6(PK_PF_STORE)
    }
    __syncthreads();  // feature tile and first group are in LDS

    // diagnostic build only (dbg bit 4): cycle stamps of workgroup 0 go to a
    // buffer nothing else reads; [wave][group][5]
#define PK_STAMP(slot_)                                                                  \
    do {                                                                                 \
        if ((dbg & 16) && stamps && blockIdx.x == 0 && (tid & 63) == 0 && g < 32)        \
            stamps[((tid >> 6) * 32 + g) * 5 + (slot_)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
    double acc = 0.0;
    float warm_sink = 0.f;
    const char *fea_b = reinterpret_cast<const char *>(fea);
    const int cl4 = cl << 2;
    for (int g = 0; g < n_grp; g++) {  // uniform: every thread takes the same trips
        const int g0 = root[t];
        const int gt = t1 - t;
        const bool staged = gstaged[g] != 0;
        const int tn = t1, tn1 = grp[g + 2];
        const bool next_staged = (g + 1 < n_grp) && gstaged[g + 1] != 0;
        PK_STAMP(0);
        if (next_staged) {  // loads fly while this group is walked
            pf_src = reinterpret_cast<const uint4 *>(nodes + root[tn]);
            pf_nv = (root[tn1] - root[tn]) >> 1;
            PK_PF6(PK_PF_LOAD)
        } else if (g + 1 == n_grp && warm_ahead > 0) {
            // last group: pull the tile of the workgroup that will follow this one on this
            // XCD (ids are dealt round-robin over the 8 XCDs, each with its own L2) into
            // the L2, one dword per 128-byte line; the result is never used
            const int64_t ahead = tile + warm_ahead;
            const int line = tid;
            // (an ordinary load with a use the compiler cannot remove: an asm load whose
            // result register the compiler thinks is free gets overwritten late)
            if (ahead * LDS_C < cn && line < F * LDS_C / 32)
                warm_sink = tiles[(size_t)ahead * F * LDS_C + (size_t)line * 32];
        }
        // early termination: a candidate whose sum can no longer reach thre*T is not
        // walked any more (its flag was set by slot 0 before the last barrier)
        const bool undecided = !prune || decided[cl] == 0;
        if (active && undecided && slot < gt && !(dbg & 2)) {
            double v;
            if (staged) {
                const char *a = reinterpret_cast<const char *>(tbuf + (root[t + slot] - g0));
                v = wave_nan ? walk_tree_lds<true>(fea_b, cl4, a) : walk_tree_lds<false>(fea_b, cl4, a);
            } else {
                v = walk_tree_global(nodes, big_roff, root[t + slot], fea, cl);
            }
            val[slot * LDS_C + cl] = v;
        }
        PK_STAMP(1);
        __syncthreads();  // every walk of the group is done: tbuf may be overwritten
        PK_STAMP(2);
        if (next_staged) { PK_PF6(PK_PF_STORE) }
        if (slot == 0 && active && undecided) {
            for (int j = 0; j < gt; j++) acc += val[j * LDS_C + cl];  // tree order
            if (prune) {
                // every remaining tree adds at most 1.0: if even that cannot lift the
                // sum to the bound (pk_prune_bound, pk_common.h: thre*T less a proven rounding margin), the
                // final p is <= thre and the pixel is not reported -- stop walking it
                const bool out = (acc + (double)(T - tn)) < prune_sum;
                if (out) {
                    decided[cl] = 1;
                    acc = 0.0;  // reported probability of a pruned candidate: 0
                } else {
                    decided[LDS_C + (g % 3)] = 1;  // this workgroup still has an open candidate
                }
            }
        }
        PK_STAMP(3);
        __syncthreads();  // next group staged; val consumed; votes cast
        bool all_done = false;
        if (prune) {
            all_done = decided[LDS_C + (g % 3)] == 0;
            if (tid == 0) decided[LDS_C + ((g + 2) % 3)] = 0;
        }
        PK_STAMP(4);
        t = tn;
        t1 = tn1;
        if (all_done) break;
    }
    if (slot == 0 && valid) prob[c] = active ? acc / (double)T : 0.0;
    // keeps the warm-ahead load alive (features are never -inf ... practically never taken)
    if (warm_sink == -__builtin_inff() && stamps) stamps[65534] = 1;
#undef PK_STAMP
}

// row-major [N][F] float32 -> [tile][F][blk] tiles (pk_predict's input path):
// one wave per row, which also notes whether the row holds a NaN.
__global__ void tile_rows_kernel(const float *__restrict__ rows, int64_t N, int F,
                                 float *__restrict__ tiles, int blk, uint8_t *__restrict__ status)
{
    const int64_t c = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= N) return;
    const int64_t tile = c / blk;
    const int col = (int)(c - tile * blk);
    bool nan = false;
    for (int f = lane; f < F; f += 64) {
        const float v = rows[c * F + f];
        nan = nan || (v != v);
        tiles[((size_t)tile * F + f) * blk + col] = v;
    }
    nan = __any(nan);
    if (lane == 0) status[c] = nan ? 2 : 1;
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
int pk_forest_tile_width(int F, const pk_options &o)
{
    // LDS-streamed trees: 128-candidate tiles, provided a useful tree buffer
    // (>= 32 KiB) is left beside the tile
    if (o.forest_lds > 0 && (size_t)F * 4 * LDS_C + 8192 + 32768 <= (size_t)160 * 1024)
        return LDS_C;
    int blk = (int)((156 * 1024) / ((size_t)F * 4)) / 64 * 64;
    if (blk > 256) blk = 256;
    if (blk < 64) blk = 64;  // no LDS-resident tile possible: the no-LDS kernel reads tiles from L2
    return blk;
}

int pk_launch_tile_rows(pk_device_ctx *ctx, const float *d_rows, int64_t N, int F, float *tiles,
                        int blk, uint8_t *d_status)
{
    if (N <= 0) return PK_OK;
    hipLaunchKernelGGL(tile_rows_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, ctx->stream,
                       d_rows, N, F, tiles, blk, d_status);
    PK_HIP(hipGetLastError());
    return PK_OK;
}

#define PK_LAUNCH_LDS_P(SLOTS, PRUNE)                                                         \
    do {                                                                                      \
        const size_t val_bytes = (size_t)((SLOTS) + ((PRUNE) ? 1 : 0)) * LDS_C * sizeof(double); \
        size_t room = (size_t)160 * 1024 - fea_bytes - val_bytes - 2 * sizeof(uint2);        \
        if ((size_t)f->opt.forest_lds * 1024 < room) room = (size_t)f->opt.forest_lds * 1024;   \
        const size_t pf_cap = (size_t)LDS_C * (SLOTS) * 6 * 16; /* THREADS * PF * 16 B */      \
        if (room > pf_cap) room = pf_cap;                                                     \
        const int tree_words = (int)(room / sizeof(uint2)) & ~1;                              \
        const size_t lds = fea_bytes + val_bytes + (size_t)(tree_words + 2) * sizeof(uint2);  \
        int rc__ = set_max_lds(forest_lds_kernel<SLOTS, PRUNE>, lds);                         \
        if (rc__) return rc__;                                                                \
        rc__ = pk_forest_groups(f, tree_words, SLOTS);                                        \
        if (rc__) return rc__;                                                                \
        hipLaunchKernelGGL((forest_lds_kernel<SLOTS, PRUNE>), dim3(grid), dim3(LDS_C *(SLOTS)), \
                           lds, ctx->stream, f->nodes, f->root, f->big_roff, f->grp, f->n_grp, \
                           f->T, f->F, tiles, d_status, c0, cn, d_prob, tree_words,           \
                           (int)f->opt.forest_dbg, ctx->dbg_buf, prune_sum,                    \
                           f->opt.forest_warm == 1 ? ctx->cu_count : (int)f->opt.forest_warm);  \
    } while (0)
#define PK_LAUNCH_LDS(SLOTS)                                                                  \
    do {                                                                                      \
        if (prune_sum > -1e300) PK_LAUNCH_LDS_P(SLOTS, true);                                 \
        else PK_LAUNCH_LDS_P(SLOTS, false);                                                   \
    } while (0)

int pk_launch_forest(pk_device_ctx *ctx, pk_forest *f, const float *tiles, int blk,
                     const uint8_t *d_status, int64_t c0, int64_t cn, double *d_prob,
                     double prune_sum, double split_sum)
{
    if (cn <= 0) return PK_OK;
    // pk_forest_plan_blk decided how this forest is evaluated (and which tile width that needs)
    if (f->plan_kind == 2 && blk == (f->q_ch == 1 ? 128 : 128 * PK_Q_FTILE) && f->q_state == 1)
        return pk_launch_forest_q(ctx, f, tiles, d_status, c0, cn, d_prob, prune_sum, split_sum);
    if (f->plan_kind == 1 && blk == 64 && f->img_state == 1)
        return pk_launch_forest_img(ctx, f, tiles, d_status, c0, cn, d_prob, prune_sum);
    pk_prof_scope prof(ctx, PK_K_FOREST);
    const unsigned grid = (unsigned)((cn + blk - 1) / blk);
    const size_t fea_bytes = (size_t)f->F * blk * sizeof(float);
    const int ilp = (int)f->opt.forest_ilp;
    if (blk == LDS_C && f->opt.forest_lds > 0) {
        f->last_family = 6;
        int slots = (int)f->opt.forest_slots;
        if (slots == 0) {
            // auto: as many slots as average trees fit beside the tile (a slot without a tree
            // idles its two waves): 8 at F = 121, 7 at F = 169 (measured 1.5 % better than 8)
            const size_t room = (size_t)160 * 1024 - fea_bytes - 8 * LDS_C * sizeof(double) - 16;
            const double avg_words = (double)f->n_nodes / (double)f->T;
            const int fit = (int)((double)(room / sizeof(uint2)) / (avg_words > 1.0 ? avg_words : 1.0));
            slots = fit >= 8 ? 8 : fit == 7 ? 7 : fit >= 5 ? 6 : fit >= 3 ? 4 : 2;
        }
        switch (slots) {
        case 2: PK_LAUNCH_LDS(2); break;
        case 4: PK_LAUNCH_LDS(4); break;
        case 6: PK_LAUNCH_LDS(6); break;
        case 7: PK_LAUNCH_LDS(7); break;
        default: PK_LAUNCH_LDS(8); break;
        }
    } else {
        // large F (w = 11): no LDS, features from the L2-resident tile
        f->last_family = 7;
        const unsigned g2 = (unsigned)((cn + 255) / 256);
        switch (ilp) {
        case 1: hipLaunchKernelGGL(forest_gmem_kernel<1>, dim3(g2), dim3(256), 0, ctx->stream, f->nodes, f->root, f->big_roff, f->T, f->F, tiles, blk, d_status, c0, cn, d_prob); break;
        case 2: hipLaunchKernelGGL(forest_gmem_kernel<2>, dim3(g2), dim3(256), 0, ctx->stream, f->nodes, f->root, f->big_roff, f->T, f->F, tiles, blk, d_status, c0, cn, d_prob); break;
        case 8: hipLaunchKernelGGL(forest_gmem_kernel<8>, dim3(g2), dim3(256), 0, ctx->stream, f->nodes, f->root, f->big_roff, f->T, f->F, tiles, blk, d_status, c0, cn, d_prob); break;
        default: hipLaunchKernelGGL(forest_gmem_kernel<4>, dim3(g2), dim3(256), 0, ctx->stream, f->nodes, f->root, f->big_roff, f->T, f->F, tiles, blk, d_status, c0, cn, d_prob); break;
        }
    }
    PK_HIP(hipGetLastError());
    return PK_OK;
}
