// pk_forest_img.hip -- Random-Forest predict_proba[:,1] over an LDS image of
// the forest (gfx950 / CDNA4 only).
//
// Replaces model.predict_proba(fea)[:, 1] at peakachu/scoreUtils.py:109, like
// pk_forest.hip, with the same workgroup shape (128 candidates x SLOTS tree
// slots, trees streamed through LDS group by group, leaf values added in tree
// order in float64 = sklearn's sequential sum), but a different walk:
//
//  * the forest is stored as the image pk_image.hip builds: sibling words are
//    adjacent (one address serves both speculative child reads), addresses are
//    absolute LDS addresses (no per-level address arithmetic on the old one),
//    leaves are words that lead back to themselves, so every tree is walked
//    for a FIXED number of levels (its depth; a grown forest's trees all reach
//    max_depth) with no termination test, no exec masking and no per-level
//    bookkeeping of leaf kinds;
//  * the feature tile is [2][F][64] (two half tiles, 64 KiB apart in LDS), so
//    the feature address is one v_perm_b32: byte 0 = lane*4, byte 1 = the
//    node's feature byte, byte 2 = the half tile.
// Per level: v_perm_b32, v_and_b32, v_cmp_le_f32, 2 x v_cndmask_b32 (5 VALU,
// was 11), ds_read_b32 + 2 x ds_read_b64 (6 LDS cycles, as before), 3 SALU.
// The two child reads are `volatile` only to keep the compiler from fusing
// them into ds_read2_b64, which takes 8 LDS cycles instead of 2 + 2.
#include "pk_common.h"

namespace {

typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) int lds_i32;
typedef unsigned v4u __attribute__((ext_vector_type(4)));  // a plain 16-byte vector (uint4 is a class)
typedef __attribute__((address_space(3))) v4u lds_u4;
typedef unsigned long long u64;
typedef __attribute__((address_space(3))) u64 lds_u64;

#define LDS_AT(type, byte_addr) (reinterpret_cast<type *>((__UINTPTR_TYPE__)(unsigned)(byte_addr)))

constexpr int IMG_C = 128;          // candidates per workgroup
constexpr unsigned HALF1 = 65536u;  // LDS offset of the second half tile

// one tree, `depth` levels; cur = the root word; lanek = lane*4 | half << 16
// one level of one walk
template <bool WITH_NAN, bool ALL_LEFT>
__device__ __forceinline__ void img_step(uint2 &cur, unsigned lanek)
{
    const unsigned xa = __builtin_amdgcn_perm(cur.y, lanek, 0x0c020700u);
    const float x = *LDS_AT(const lds_f32, xa);
    const unsigned ca = cur.y & 0x3fff8u;
    const u64 lw = *LDS_AT(const volatile lds_u64, ca);
    const u64 rw = *LDS_AT(const volatile lds_u64, ca + 8);
    bool gl = x <= __uint_as_float(cur.x);
    if (WITH_NAN) gl = gl | ((x != x) & ((cur.y & 1u) != 0));
    if (ALL_LEFT) gl = gl | (x == x);  // timing ablation: every lane takes the same path
    cur.x = gl ? (unsigned)lw : (unsigned)rw;
    cur.y = gl ? (unsigned)(lw >> 32) : (unsigned)(rw >> 32);
}

// one tree, `depth` levels; cur = the root word; lanek = lane*4 | half << 16.
// Four levels per loop trip: a taken branch costs the wave an instruction refetch.
template <bool WITH_NAN, bool ALL_LEFT = false>
__device__ __forceinline__ double walk_img(uint2 cur, int depth, unsigned lanek)
{
    int d = depth;
    for (; d >= 4; d -= 4) {
        img_step<WITH_NAN, ALL_LEFT>(cur, lanek);
        img_step<WITH_NAN, ALL_LEFT>(cur, lanek);
        img_step<WITH_NAN, ALL_LEFT>(cur, lanek);
        img_step<WITH_NAN, ALL_LEFT>(cur, lanek);
    }
    for (; d > 0; d--) img_step<WITH_NAN, ALL_LEFT>(cur, lanek);
    return *LDS_AT(const lds_f64, cur.x & 0x3ffffu);
}

// staging registers: the next group's image travels global -> VGPR while the current
// one is walked and VGPR -> LDS after the barrier (named registers: arrays went to scratch)
#define IMG_PF12(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)
#define IMG_PF_DECL(q) v4u pf##q = {0u, 0u, 0u, 0u};
#define IMG_PF_LOAD(q) \
    if constexpr ((q) < PFN) pf##q = pf_src[min(tid + (q) * THREADS, pf_nu - 1)];
#define IMG_PF_STORE(q)                                                                   \
    if constexpr ((q) < PFN) {                                                            \
        const int u = tid + (q) * THREADS;                                                \
        if (u < pf_nu) {                                                                  \
            const int vo = u << 4;                                                        \
            *LDS_AT(lds_u4, vo < lenA ? HB + vo : B0 + (vo - lenA)) = pf##q;              \
        }                                                                                 \
    }

template <int SLOTS>
constexpr int img_pfn()
{
    return SLOTS >= 8 ? 6 : SLOTS == 7 ? 7 : SLOTS == 6 ? 8 : SLOTS == 5 ? 10 : 12;
}

template <int SLOTS, bool PRUNE>
__global__ __launch_bounds__(IMG_C *SLOTS) void forest_img_kernel(
    const v4u *__restrict__ img, const int4 *__restrict__ gtab, int n_grp,
    const uint2 *__restrict__ troot, const int32_t *__restrict__ tdepth, int T, int F, int lenA,
    int B0, const float *__restrict__ tiles, const uint8_t *__restrict__ status, int64_t c0,
    int64_t cn, double *__restrict__ prob, double prune_sum, int warm_ahead, int dbg,
    long long *__restrict__ stamps)
{
    constexpr int THREADS = IMG_C * SLOTS;
    constexpr int PFN = img_pfn<SLOTS>();
    extern __shared__ __attribute__((aligned(16))) char lds[];  // all of it: addressed absolutely
    const int tid = threadIdx.x;
    const int cl = tid & (IMG_C - 1);
    const int slot = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int HB = F * 256;
    const int val_off = (int)HALF1 + HB;            // [SLOTS][128] float64
    const int dec_off = val_off + SLOTS * 1024;     // 128 flags + 3 vote words
    const unsigned lanek = ((unsigned)(cl & 63) << 2) | ((cl & 64) ? HALF1 : 0u);
    // the image addresses LDS from offset 0 (this kernel has no static LDS)
    const bool lds_at_zero = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char *)lds == 0u;
    if (!lds_at_zero && tid == 0 && stamps) stamps[65535] = 2;

    if (PRUNE && tid < IMG_C + 3) *LDS_AT(lds_i32, dec_off + 4 * tid) = 0;
    const int64_t tile = blockIdx.x;  // 128 candidates = two 64-candidate feature tiles
    const int64_t local = tile * IMG_C + cl;
    {
        const v4u *src = reinterpret_cast<const v4u *>(tiles + (size_t)tile * 2 * F * 64);
        // the second half tile exists only if it holds a candidate (the buffer ends there)
        const int nu = (tile * IMG_C + 64 < cn ? 2 : 1) * (HB >> 4);
        for (int i = tid; i < nu; i += THREADS) {
            const int o = i << 4;
            *LDS_AT(lds_u4, o < HB ? o : o - HB + (int)HALF1) = src[i];
        }
    }
    const bool valid = local < cn;
    const int64_t c = c0 + (valid ? local : 0);
    const unsigned st = valid ? status[c] : 0;
    const bool active = st != 0 && lds_at_zero;
    const bool wave_nan = __any(st == 2);  // a wave holding NaN features takes the slow walk

    IMG_PF12(IMG_PF_DECL)
    const v4u *pf_src;
    int pf_nu;
    int4 g_cur = gtab[0];
    {
        pf_src = img + g_cur.z;
        pf_nu = g_cur.w;
        IMG_PF12(IMG_PF_LOAD)
        IMG_PF12(IMG_PF_STORE)
    }
    __syncthreads();  // feature tile and first group are in LDS

#define IMG_STAMP(slot_)                                                                 \
    do {                                                                                 \
        if ((dbg & 16) && stamps && blockIdx.x == 0 && (tid & 63) == 0 && g < 32)        \
            stamps[((tid >> 6) * 32 + g) * 5 + (slot_)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
    double acc = 0.0;
    float warm_sink = 0.f;
    for (int g = 0; g < n_grp; g++) {  // uniform: every thread takes the same trips
        const int t0 = g_cur.x, gt = g_cur.y;
        const int4 g_nxt = gtab[g + 1];
        IMG_STAMP(0);
        if (g + 1 < n_grp) {  // loads fly while this group is walked
            pf_src = img + g_nxt.z;
            pf_nu = g_nxt.w;
            IMG_PF12(IMG_PF_LOAD)
        } else if (warm_ahead > 0) {
            // last group: pull the tiles of the workgroup that follows this one on this XCD
            // into its L2, one dword per 128-byte line; the value is never used
            const int64_t ahead = tile + warm_ahead;
            if (ahead * IMG_C + 64 < cn && tid < F * 4)
                warm_sink = tiles[(size_t)ahead * 2 * F * 64 + (size_t)tid * 32];
        }
        const bool undecided = !PRUNE || *LDS_AT(lds_i32, dec_off + 4 * cl) == 0;
        if (active && undecided && slot < gt && !(dbg & 2)) {
            const int t = t0 + slot;
            const uint2 r = troot[t];
            const int depth = tdepth[t];
            const double v = (dbg & 8)   ? walk_img<false, true>(r, depth, lanek)  // wrong results
                             : wave_nan ? walk_img<true>(r, depth, lanek)
                                        : walk_img<false>(r, depth, lanek);
            *LDS_AT(lds_f64, val_off + (slot * IMG_C + cl) * 8) = v;
        }
        IMG_STAMP(1);
        __syncthreads();  // every walk of the group is done: the image may be overwritten
        IMG_STAMP(2);
        if (g + 1 < n_grp) { IMG_PF12(IMG_PF_STORE) }
        if (slot == 0 && active && undecided) {
            for (int j = 0; j < gt; j++) acc += *LDS_AT(lds_f64, val_off + (j * IMG_C + cl) * 8);  // tree order
            if (PRUNE) {
                // every remaining tree adds at most 1.0: if even that cannot lift the sum to
                // the bound (pk_prune_bound, pk_common.h: thre*T less a proven rounding margin) the final p is
                // <= thre and the pixel is not reported -- stop walking it
                const bool out = (acc + (double)(T - (t0 + gt))) < prune_sum;
                if (out) {
                    *LDS_AT(lds_i32, dec_off + 4 * cl) = 1;
                    acc = 0.0;  // reported probability of a pruned candidate: 0
                } else {
                    *LDS_AT(lds_i32, dec_off + 4 * (IMG_C + (g % 3))) = 1;  // still an open candidate
                }
            }
        }
        IMG_STAMP(3);
        __syncthreads();  // next group staged; values consumed; votes cast
        bool all_done = false;
        if (PRUNE) {
            // a vote word is set before this barrier, read after it and cleared two groups
            // ahead, so a clear and a set of the same word are always a barrier apart
            all_done = *LDS_AT(lds_i32, dec_off + 4 * (IMG_C + (g % 3))) == 0;
            if (tid == 0) *LDS_AT(lds_i32, dec_off + 4 * (IMG_C + ((g + 2) % 3))) = 0;
        }
        IMG_STAMP(4);
        g_cur = g_nxt;
        if (all_done) break;
    }
    if (slot == 0 && valid) prob[c] = active ? acc / (double)T : 0.0;
    // keeps the warm-ahead load alive (features are never -inf)
    if (warm_sink == -__builtin_inff() && stamps) stamps[65534] = 1;
#undef IMG_STAMP
}


template <typename KernelT>
int img_set_max_lds(KernelT k, size_t bytes)
{
    PK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PK_OK;
}

void img_free(pk_forest *f)
{
    if (f->img) hipFree(f->img);
    if (f->img_gtab) hipFree(f->img_gtab);
    if (f->img_troot) hipFree(f->img_troot);
    if (f->img_tdepth) hipFree(f->img_tdepth);
    f->img = nullptr;
    f->img_gtab = nullptr;
    f->img_troot = nullptr;
    f->img_tdepth = nullptr;
    delete f->img_layout;
    f->img_layout = nullptr;
}

}  // namespace

void pk_forest_img_release(pk_forest *f)
{
    img_free(f);
    f->img_state = 0;
}

// Builds the image for the largest slot count whose groups are (nearly) full:
// `slots` wave pairs need `slots` trees beside the tile; an empty slot idles.
static int img_plan(pk_forest *f)
{
    const int F = f->F, T = f->T;
    if (F > 255 || f->h_tree_off.empty()) return PK_E_UNSUPPORTED;
    pk_img_out best;
    pk_img_layout bestL;
    int best_slots = 0;
    double best_score = 0.0;
    const int forced = (int)f->opt.forest_slots;
    for (int slots = 2; slots <= 8; slots++) {
        if (slots == 3) continue;  // not instantiated
        if (forced && slots != forced) continue;
        pk_img_layout L;
        if (!pk_img_make_layout(F, slots, pk_img_stage_bytes(slots), &L)) continue;
        pk_img_out out;
        const int rc = pk_img_build(T, F, f->h_tree_off.data(), f->h_left.data(), f->h_right.data(),
                                    f->h_feat.data(), f->h_thr.data(),
                                    f->h_miss.empty() ? nullptr : f->h_miss.data(), f->h_p1.data(),
                                    L, &out);
        if (rc == PK_E_UNSUPPORTED) continue;
        if (rc) return rc;
        // a group costs about the same whatever it holds, so trees per group is the figure
        // of merit; a slot count that adds less than a quarter tree per group only adds
        // idle waves and is not taken
        const double score = (double)T / (double)out.n_grp;
        if (score > best_score + 0.24) {
            best_score = score;
            best = std::move(out);
            bestL = L;
            best_slots = slots;
        }
    }
    if (!best_slots) return PK_E_UNSUPPORTED;
    img_free(f);
    f->img_layout = new pk_img_layout(bestL);
    f->img_slots = best_slots;
    f->img_n_grp = best.n_grp;
    PK_HIP(hipMalloc((void **)&f->img, best.words.size() * sizeof(uint2)));
    PK_HIP(hipMalloc((void **)&f->img_gtab, best.gtab.size() * sizeof(int32_t)));
    PK_HIP(hipMalloc((void **)&f->img_troot, (size_t)T * sizeof(uint2)));
    PK_HIP(hipMalloc((void **)&f->img_tdepth, (size_t)T * sizeof(int32_t)));
    PK_HIP(hipMemcpy(f->img, best.words.data(), best.words.size() * sizeof(uint2), hipMemcpyHostToDevice));
    PK_HIP(hipMemcpy(f->img_gtab, best.gtab.data(), best.gtab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    PK_HIP(hipMemcpy(f->img_troot, best.troot.data(), (size_t)T * sizeof(uint2), hipMemcpyHostToDevice));
    PK_HIP(hipMemcpy(f->img_tdepth, best.tdepth.data(), (size_t)T * sizeof(int32_t), hipMemcpyHostToDevice));
    return PK_OK;
}

int pk_forest_plan_blk(pk_forest *f)
{
    f->plan_kind = 0;
    if (f->opt.forest_q && f->opt.forest_lds > 0) {
        const int rc = pk_forest_q_plan(f);
        if (rc == PK_OK) {
            f->plan_kind = 2;
            return f->q_ch == 1 ? 128 : 128 * PK_Q_FTILE;  // (wide forests: plain 128-candidate float tiles)
        }
        if (rc != PK_E_UNSUPPORTED) return 0;  // error already set
    }
    if (!f->opt.forest_img || f->opt.forest_lds <= 0) return pk_forest_tile_width(f->F, f->opt);
    if (f->img_state != 0 && f->img_opt_slots != f->opt.forest_slots) {
        img_free(f);
        f->img_state = 0;
    }
    if (f->img_state == 0) {
        f->img_opt_slots = f->opt.forest_slots;
        const int rc = img_plan(f);
        f->img_state = rc == PK_OK ? 1 : -1;
        if (rc != PK_OK) img_free(f);
    }
    if (f->img_state == 1) {
        f->plan_kind = 1;
        return 64;
    }
    return pk_forest_tile_width(f->F, f->opt);
}

#define IMG_LAUNCH_P(SLOTS, PRUNE)                                                             \
    do {                                                                                       \
        int rc__ = img_set_max_lds(forest_img_kernel<SLOTS, PRUNE>, 163840);                   \
        if (rc__) return rc__;                                                                 \
        hipLaunchKernelGGL((forest_img_kernel<SLOTS, PRUNE>), dim3(grid), dim3(IMG_C *(SLOTS)),  \
                           163840, ctx->stream, reinterpret_cast<const v4u *>(f->img), reinterpret_cast<const int4 *>(f->img_gtab), \
                           f->img_n_grp, f->img_troot, f->img_tdepth, f->T, f->F, L.lenA, L.B0, \
                           tiles, d_status, c0, cn, d_prob, prune_sum,                         \
                           f->opt.forest_warm == 1 ? ctx->cu_count : (int)f->opt.forest_warm,    \
                           (int)f->opt.forest_dbg, ctx->dbg_buf);                               \
    } while (0)
#define IMG_LAUNCH(SLOTS)                                                                      \
    do {                                                                                       \
        if (prune_sum > -1e300) IMG_LAUNCH_P(SLOTS, true);                                     \
        else IMG_LAUNCH_P(SLOTS, false);                                                       \
    } while (0)
int pk_launch_forest_img(pk_device_ctx *ctx, pk_forest *f, const float *tiles,
                         const uint8_t *d_status, int64_t c0, int64_t cn, double *d_prob,
                         double prune_sum)
{
    if (cn <= 0) return PK_OK;
    if (f->img_state != 1 || !f->img_layout) {
        pk_set_error("forest image kernel launched without an image (internal error)");
        return PK_E_INVALID;
    }
    pk_prof_scope prof(ctx, PK_K_FOREST);
    f->last_family = 4;
    const pk_img_layout &L = *f->img_layout;
    const unsigned grid = (unsigned)((cn + IMG_C - 1) / IMG_C);
    switch (f->img_slots) {
    case 2: IMG_LAUNCH(2); break;
    case 4: IMG_LAUNCH(4); break;
    case 5: IMG_LAUNCH(5); break;
    case 6: IMG_LAUNCH(6); break;
    case 7: IMG_LAUNCH(7); break;
    case 8: IMG_LAUNCH(8); break;
    default:
        pk_set_error("forest image: %d slots not instantiated", f->img_slots);
        return PK_E_INVALID;
    }
    PK_HIP(hipGetLastError());
    return PK_OK;
}
