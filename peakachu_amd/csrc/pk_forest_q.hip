// pk_forest_q.hip -- Random-Forest predict_proba[:,1] on rank codes (gfx950 only).
//
// Replaces model.predict_proba(fea)[:, 1] at peakachu/scoreUtils.py:109 (sklearn
// ForestClassifier.predict_proba -> Tree._apply_dense) exactly; pk_qimage.hip
// explains why ranks decide every split the same way as the float32 features.
//
// Why this shape (measured on MI355X, tools/micro/lds_chain.hip, EXPERIMENTS.md 4.2):
// a level of a tree walk is one dependent LDS round trip, and with >= 12 waves
// walking the LDS pipe is what binds: a wave-level costs the CU about 15 cycles
// when it reads a float32 feature and two 8-byte child words, and about 9 when
// it reads a 16-bit code and ONE 8-byte word that holds both children.  Hence
//   * quantize_tiles_kernel turns the extractor's float32 tiles into 16-bit
//     rank codes once per candidate (a streaming pass, table lookups in LDS);
//   * nodes are 4 bytes, sibling words adjacent; leaves lead back to themselves,
//     so a tree is walked for a fixed number of levels with no termination test;
//   * half the bytes per tree and per candidate tile put 8 whole trees and 256
//     candidates (F <= 192; else up to 16 trees and 128 candidates) into the
//     160 KiB of LDS at once: half as many tree groups (barriers, LDS commits,
//     L2 -> LDS traffic) per candidate as the float kernels need, and all 16
//     waves walk (two per tree, one for each 128-candidate rank tile).
// Per level and walk: v_perm_b32 (feature address), v_bfe_u32 + v_lshl_add_u32
// (pair address), v_cmp_le_u32_sdwa (code against the word's upper half),
// v_cndmask_b32; ds_read_u16 + ds_read_b64.  Leaf values are added in tree order
// in float64, the sequential sum sklearn computes.
#include <stdlib.h>

#include <type_traits>

#include "pk_common.h"

namespace {

typedef __attribute__((address_space(3))) unsigned short lds_u16;
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) int lds_i32;
typedef unsigned v4u __attribute__((ext_vector_type(4)));  // a plain 16-byte vector (uint4 is a class)
typedef __attribute__((address_space(3))) v4u lds_u4;
typedef unsigned long long u64;
typedef __attribute__((address_space(3))) u64 lds_u64;

#define LDS_AT(type, byte_addr) (reinterpret_cast<type *>((__UINTPTR_TYPE__)(unsigned)(byte_addr)))


// ------------------------------------------------------------------------
// float32 feature tiles [tile / 8][F][8 x 128] -> rank codes [tile][F][64][2] u16.
// (The extractor is asked for 1024-candidate float tiles: a feature's values of 8
// consecutive 128-candidate tiles are then 4 KiB of contiguous HBM for the block that
// converts this feature, instead of 8 pieces 60 KiB apart.)
// Block (f, s): the tables of feature f in LDS, every s-th group of tiles.
// code = r(x) << 5 with r(x) = number of the feature's distinct thresholds
// below x (exact: the lookup cell settles all thresholds but the few -- mostly
// none or one -- that share the cell, which are compared), NaN -> 0xFFFF.
// ------------------------------------------------------------------------
// Eight values at a time, in phases, so that the eight lookups are in flight together and
// no branch stands between them (NaN takes cell 0 like any small value and is overridden
// at the end; the rare lanes whose cell holds more than one threshold finish in a loop).
template <int N>
__device__ __forceinline__ void q_codes(const float (&x)[N], unsigned (&code)[N], const float *thr,
                                        const unsigned *lut, float lo, float inv, int shift)
{
    unsigned e[N], r[N];
    float th[N];
#pragma unroll
    for (int i = 0; i < N; i++) e[i] = lut[pk_q_cell(x[i], lo, inv)];
#pragma unroll
    for (int i = 0; i < N; i++) {
        r[i] = e[i] & 0xFFFFu;  // thresholds in lower cells: all below x
        th[i] = thr[r[i]];      // (thr[] is padded: always readable)
    }
    bool more = false;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const unsigned k = e[i] >> 16;  // thresholds of x's own cell (ascending): mostly none or one
        const bool first = (k != 0) & (th[i] < x[i]);
        r[i] += first;
        more = more | (first & (k > 1));
    }
    if (__any(more)) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            unsigned k = e[i] >> 16;
            if (k > 1 && r[i] != (e[i] & 0xFFFFu))
                for (k--; k != 0 && thr[r[i]] < x[i]; k--) r[i]++;
        }
    }
#pragma unroll
    for (int i = 0; i < N; i++) code[i] = x[i] != x[i] ? 0xFFFFu : r[i] << shift;  // (5; 4 for the 12-bit rank field)
}

// the cell of every threshold, computed where the quantizer computes the cells of the features
__global__ void q_cells_kernel(const float *__restrict__ qthr, const int32_t *__restrict__ qoff,
                               const float *__restrict__ qpar, int F, int32_t *__restrict__ cells)
{
    const int f = blockIdx.x;
    for (int i = qoff[f] + threadIdx.x; i < qoff[f + 1]; i += blockDim.x)
        cells[i] = pk_q_cell(qthr[i], qpar[2 * f], qpar[2 * f + 1]);
}

// TP = tiles a wave converts per trip (2 * TP loads in flight per lane); the block has blockDim.x
// threads (the tables of one feature: 24 KB of LDS per block)
template <int TP>
__global__ __launch_bounds__(1024) void quantize_tiles_kernel(
    const float *__restrict__ tiles, int64_t n_tiles, int Fs, int F, const int32_t *__restrict__ qsrc,
    const float *__restrict__ qthr, const int32_t *__restrict__ qoff, const unsigned *__restrict__ qlut,
    const float *__restrict__ qpar, unsigned short *__restrict__ qtiles, int tile64, int shift)
{
    // Fs rows in a float tile, F >= Fs rows in a rank tile: row f is made from float feature qsrc[f]
    // (a feature with more than 2 047 thresholds has virtual features behind the real ones)
    // shift = 5: 11-bit ranks, n <= 2047 thresholds per row; 4: the 12-bit rank field, n <= 4095
    __shared__ float thr[4096];  // n entries + padding (filled as far as the rank field counts)
    __shared__ unsigned lut[PK_Q_CELLS];
    const int f = blockIdx.x, fs = qsrc[f];
    const int o = qoff[f], n = qoff[f + 1] - o;
    const int thr_fill = shift == 4 ? 4096 : 2048;
    for (int i = threadIdx.x; i < thr_fill; i += blockDim.x) thr[i] = i < n ? qthr[o + i] : __builtin_inff();
    for (int i = threadIdx.x; i < PK_Q_CELLS; i += blockDim.x) lut[i] = qlut[(size_t)f * PK_Q_CELLS + i];
    const float lo = qpar[2 * f], inv = qpar[2 * f + 1];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // a wave converts four rows per trip: the loads of all four are in flight together
    // (one row per trip left the kernel waiting for HBM: 1.55 ms per 5.6 M candidates)
    // (a wave takes four consecutive tiles, the block sixteen: 8 KiB of this feature's values)
    // (four consecutive tiles starting at a multiple of four lie in ONE group of eight: their
    // rows are 512 B apart -- immediate offsets of one address; the float buffer is whole
    // groups, so a tile past n_tiles is readable and merely not converted)
    const size_t out_stride = (size_t)F * 64;  // dwords between the rows of consecutive tiles
    static_assert(TP == 4 || TP == 8, "a wave's tiles of a trip lie in one group of eight");
    const int WPB = blockDim.x >> 6;  // waves per block
    for (int64_t t = ((int64_t)blockIdx.y * WPB + wave) * TP; t < n_tiles; t += (int64_t)gridDim.y * WPB * TP) {
        // float tiles: groups of 8 (1024 candidates per feature row) or, tile64 (wide forests,
        // whose one-window-per-wave extractor stores a feature at a time), single 128-tiles
        const float *src = tile64 ? tiles + ((size_t)t * Fs + fs) * 128 + lane
                                  : tiles + ((size_t)(t >> 3) * Fs + fs) * (128 * PK_Q_FTILE) + (size_t)(t & 7) * 128 + lane;
        const int kstride = tile64 ? Fs * 128 : 128;  // floats between consecutive tiles of this feature
        float x[2 * TP];
        unsigned code[2 * TP];
#pragma unroll
        for (int k = 0; k < TP; k++) {
            const int kk = (tile64 && t + k >= n_tiles) ? 0 : k;  // (single tiles: nothing behind the last one)
            // (non-temporal: the float tile is read once and the rank tile written once -- 1.5 GB per chunk that
            // would otherwise push the lookup tables and the next kernel's working set out of the L2; with the
            // extractor's stores marked the same way the quantizer runs 0.81 -> 0.71 ms per step of config 2:
            // profiles/r06_ab_nontemporal.log)
            x[2 * k] = __builtin_nontemporal_load(src + (size_t)kk * kstride);
            x[2 * k + 1] = __builtin_nontemporal_load(src + (size_t)kk * kstride + 64);
        }
        q_codes<2 * TP>(x, code, thr, lut, lo, inv, shift);
        // dword j of a row holds the codes of candidates j (low half) and j + 64 (high
        // half): the two walks of a lane read the same LDS bank, different lanes different banks
        if (tile64) {
            // 64-candidate tiles [tile][F][64] u16 (wide forests): candidates lane and lane + 64
            // of a 128-tile go to two consecutive 64-tiles
#pragma unroll
            for (int k = 0; k < TP; k++) {
                if (t + k >= n_tiles) break;
                unsigned short *d16 = qtiles + ((size_t)(t + k) * 2 * F + f) * 64 + lane;
                __builtin_nontemporal_store((unsigned short)code[2 * k], d16);
                __builtin_nontemporal_store((unsigned short)code[2 * k + 1], d16 + (size_t)F * 64);
            }
            continue;
        }
        unsigned *dst = reinterpret_cast<unsigned *>(qtiles) + ((size_t)t * F + f) * 64 + lane;
#pragma unroll
        for (int k = 0; k < TP; k++) {
            if (t + k >= n_tiles) break;
            __builtin_nontemporal_store(code[2 * k] | (code[2 * k + 1] << 16), dst + k * out_stride);
        }
    }
}

// ------------------------------------------------------------------------
// the walk
// ------------------------------------------------------------------------
template <bool WIDE = false>
__device__ __forceinline__ unsigned q_pair_index(unsigned w)
{
    unsigned t;  // (the compiler turns a C bit-field extract into shift + and + add: 3 VALU, not 2)
    if (WIDE) asm("v_bfe_u32 %0, %1, 10, 11" : "=v"(t) : "v"(w));
    else asm("v_bfe_u32 %0, %1, 8, 12" : "=v"(t) : "v"(w));
    return t;
}

// one level of CH walks of one tree.  Walk c belongs to candidate lane + 64 c of the
// workgroup: code address = (c >> 1) * 32 KiB + feature * 256 + lane * 4 + (c & 1) * 2
// X0: LDS offset of the rank tile walks 0 and 1 read; walks 2 and 3 read the one HALF1
// bytes further (both are immediate offsets of the ds_read)
// nsplit (narrow word, WITH_NAN only): 0 = "NaN goes left" is bit 20 of the word; otherwise the 12-bit rank
// form, whose bit 20 belongs to the rank: the node's pair lies at or beyond the tree's split (pk_qimage.hip)
template <int CH, int X0, int HALF1, bool WITH_NAN, bool ALL_LEFT>
__device__ __forceinline__ void q_level(unsigned (&w)[CH], unsigned tbase, unsigned lk0, unsigned lk1, unsigned nsplit = 0u)
{
    unsigned xv[CH];
    u64 pr[CH];
    // one walk per lane = the 64-candidate tile [F][64] u16 and the wide node word (10-bit
    // feature, 11-bit pair index; "NaN goes left" = the pair lies at or beyond the tree's split, lk1)
    constexpr bool WIDE = CH == 1;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        // byte 0 <- lane constant, byte 1 <- the word's feature byte
        const unsigned xa = WIDE ? ((w[c] & 0x3FFu) << 7) + lk0
                                 : __builtin_amdgcn_perm(w[c], (c & 1) ? lk1 : lk0, 0x0c0c0400u);
        xv[c] = *LDS_AT(const lds_u16, xa + X0 + (c >> 1) * HALF1);
        pr[c] = *LDS_AT(const lds_u64, tbase + (q_pair_index<WIDE>(w[c]) << 3));
    }
#pragma unroll
    for (int c = 0; c < CH; c++) {
        bool gl = xv[c] <= (w[c] >> 16);  // rank(x) <= rank(threshold); the bits below cannot flip it
        // NaN goes left where the node says so: bit 20 of the narrow word; the wide word has no bit to
        // spare: its nodes that send NaN left have their child pairs at or beyond the tree's `split`
        // (pk_qimage.hip), handed in through lk1 (a one-walk lane has no second code offset)
        if (WITH_NAN)
            gl = gl | ((xv[c] == 0xFFFFu) & (WIDE     ? ((w[c] >> 10) & 0x7FFu) >= lk1
                                             : nsplit ? ((w[c] >> 8) & 0xFFFu) >= nsplit
                                                      : (w[c] & (1u << 20)) != 0));
        if (ALL_LEFT) gl = gl | (xv[c] < 0x10000u);  // timing ablation: every lane takes the same path
        w[c] = gl ? (unsigned)pr[c] : (unsigned)(pr[c] >> 32);
    }
}

// POS >= 0: the wave's issue priority rotates every four levels, starting from POS (its
// position among the four waves of its SIMD).  At equal priority the SIMD issues oldest
// first: the walks of a group then finish in wave order, the last quarter about 25 % after
// the first, and everybody waits for them at the barrier.
template <int CH, int X0, int HALF1, bool WITH_NAN, bool ALL_LEFT = false, int POS = -1>
__device__ __forceinline__ void q_walk(unsigned root, int depth, unsigned tbase, unsigned lk0,
                                       unsigned lk1, double (&v)[CH], unsigned nsplit = 0u)
{
    unsigned w[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) w[c] = root;
    int d = depth;
#define Q_TWO_LEVELS()                                                  \
    do {                                                                \
        q_level<CH, X0, HALF1, WITH_NAN, ALL_LEFT>(w, tbase, lk0, lk1, nsplit); \
        q_level<CH, X0, HALF1, WITH_NAN, ALL_LEFT>(w, tbase, lk0, lk1, nsplit); \
    } while (0)
    if (POS >= 0) {
        // (round 2: four levels per priority measured best: 3.85 ms; two 3.91, one 3.93, none 4.01)
        // The priority of a wave by its position POS (among the four waves of its SIMD) and the block
        // of levels: four blocks of four, then two of two (depth 20).  Round 2 stepped UP,
        // (POS + {0,1,2,3}) & 3 and one level per priority at the end; stepping DOWN after the first
        // block, (POS + {0,3,2,1}) & 3 and POS, POS + 3 at the end, is 3 % faster on the same box,
        // 3.66 -> 3.55 ms: the waves of the highest position start last (the load issue in front of
        // the walk is served in wave order) and this order lets them finish with the others.
        // Schedules measured: profiles/r03_prio_schedules.log; -DPK_Q_PRIO_TAB=... builds another.
#ifndef PK_Q_PRIO_TAB
#define PK_Q_PRIO_TAB {0, 3, 2, 1, 0, 3}, {1, 0, 3, 2, 1, 0}, {2, 1, 0, 3, 2, 1}, {3, 2, 1, 0, 3, 2}
#endif
        constexpr int PT[4][6] = {PK_Q_PRIO_TAB};
        constexpr int PP = POS & 3;
        for (; d >= 16; d -= 16) {
            __builtin_amdgcn_s_setprio(PT[PP][0]);
            Q_TWO_LEVELS();
            Q_TWO_LEVELS();
            __builtin_amdgcn_s_setprio(PT[PP][1]);
            Q_TWO_LEVELS();
            Q_TWO_LEVELS();
            __builtin_amdgcn_s_setprio(PT[PP][2]);
            Q_TWO_LEVELS();
            Q_TWO_LEVELS();
            __builtin_amdgcn_s_setprio(PT[PP][3]);
            Q_TWO_LEVELS();
            Q_TWO_LEVELS();
        }
        for (; d >= 4; d -= 4) {  // what is left of the depth
            __builtin_amdgcn_s_setprio(PT[PP][4]);
            Q_TWO_LEVELS();
            __builtin_amdgcn_s_setprio(PT[PP][5]);
            Q_TWO_LEVELS();
        }
        for (; d >= 2; d -= 2) Q_TWO_LEVELS();
    } else {
        for (; d >= 2; d -= 2) Q_TWO_LEVELS();  // two levels per trip: a taken branch costs an instruction refetch
    }
#undef Q_TWO_LEVELS
    if (d) q_level<CH, X0, HALF1, WITH_NAN, ALL_LEFT>(w, tbase, lk0, lk1, nsplit);
    if (POS >= 0) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int c = 0; c < CH; c++)  // the leaf's float64 value follows its pair
        v[c] = *LDS_AT(const lds_f64, tbase + ((q_pair_index<CH == 1>(w[c]) + 1) << 3));
}

// One chain per lane (the 64-candidate shape), no priorities: sixteen levels straight, then blocks
// of four (a walk that has reached its leaf stays there, so up to three surplus levels change
// nothing) -- every loop test is an instruction in a stream whose length bounds the walk
// (forest_qr_kernel's finding, round 4).  `split`: see q_level (the wide word's NaN rule).
template <bool WITH_NAN>
__device__ __forceinline__ double q_walk_one(unsigned root, int depth, unsigned tbase, unsigned lk0, unsigned split)
{
    unsigned w[1] = {root};
#define Q1_FOUR()                                                  \
    do {                                                           \
        q_level<1, 0, 32768, WITH_NAN, false>(w, tbase, lk0, split); \
        q_level<1, 0, 32768, WITH_NAN, false>(w, tbase, lk0, split); \
        q_level<1, 0, 32768, WITH_NAN, false>(w, tbase, lk0, split); \
        q_level<1, 0, 32768, WITH_NAN, false>(w, tbase, lk0, split); \
    } while (0)
    int n4 = (depth + 3) >> 2;
    if (__builtin_expect(n4 >= 4, 1)) {
        Q1_FOUR();
        Q1_FOUR();
        Q1_FOUR();
        Q1_FOUR();
        n4 -= 4;
    }
    for (; n4 > 0; n4--) Q1_FOUR();
#undef Q1_FOUR
    return *LDS_AT(const lds_f64, tbase + ((q_pair_index<true>(w[0]) + 1) << 3));
}

#define Q_PF16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define Q_PF_DECL(q) v4u pf##q = {0u, 0u, 0u, 0u};
// (register q is only moved when the piece reaches it: a wave-uniform test).  The piece
// is pf_nu 16-byte units at pf_src, destined for LDS offset pf_dst; thread i0 + q * stride
// moves unit q of its column (all threads of the workgroup, or the lanes of one wave).
#define Q_PF_LOAD(q)                                                           \
    if constexpr ((q) < PFN)                                                   \
        if ((q) * pf_stride < pf_nu) pf##q = pf_src[min(pf_i0 + (q) * pf_stride, pf_nu - 1)];
#define Q_PF_LOADX(q)                                                                           \
    if constexpr ((q) < PFN) {                                                                  \
        if ((q) < 3) {                                                                          \
            if ((q) * ld_stride < ld_lo) pf##q = __builtin_nontemporal_load(pf_src + min(ld_i0 + (q) * ld_stride, ld_lo - 1)); \
        } else {                                                                                \
            const int r_ = (q) * ld_stride - ld_shift;                                          \
            if (r_ < ld_hi) pf##q = __builtin_nontemporal_load(pf_src + ld_base + min(ld_i0 + r_, ld_hi - 1)); \
        }                                                                                       \
    }
#define Q_PF_STORE(q)                                                          \
    if constexpr ((q) < PFN) {                                                 \
        const int u = pf_i0 + (q) * pf_stride;                                 \
        if ((q) * pf_stride < pf_nu && u < pf_nu) *LDS_AT(lds_u4, pf_dst + (u << 4)) = pf##q; \
    }

// One workgroup = 64 * CH candidates and 16 waves.  WPT = 1: wave s walks tree s of the
// group for all candidates (CH walks per lane).  WPT = 2 (CH = 4, at most 8 trees per
// group): waves 2s and 2s+1 walk tree s, one for each rank tile (2 walks per lane), so
// that all 16 waves walk.  Waves without a tree only help to move data (tile load,
// staging of the next group: the LDS store path wants all SIMDs busy).  Trees arrive
// group by group: the next group travels global -> VGPR during the walk and VGPR -> LDS
// behind the barrier.
constexpr int Q_THREADS = 1024;
// EARLY (WPT = 2 only): every tree has a fixed slot of slot_bytes in LDS, and the two
// waves that walk tree s stage the NEXT group's tree s themselves -- each its half, from
// its own registers -- as soon as both are done with the current one (an LDS counter per
// slot, not a workgroup barrier).  The commit of a group then overlaps the walks of the
// waves that are still busy instead of standing between two barriers (it was 11 % of the
// kernel: 4.21 -> 3.76 ms with the stores switched off).  MEASURED: slower, 4.51 vs 4.09 ms
// -- LDS stores issued by two waves alone run at half rate (the store path wants waves on
// all four SIMDs) and delay the reads of the waves still walking; option forest_q_early,
// off by default.
struct q_slot_table {
    int off[16];  // EARLY: LDS offset (relative to img_off) of the tree slot of each wave pair
};
struct qr_split_args {
    unsigned *cnt;
    int32_t *idx;
    double *acc;
    uint8_t *st;
    unsigned short *tiles;
    double rem;
};

// The parking of a cut forest's head (see forest_qr_kernel, where the same steps are written out between
// its stamps -- and stay written out: with this function called from there instead, same logic, the default
// kernel's head took 2.26 ms per step of config 2 against 2.10, same-box A/B, round 5) for the generic shapes: C = 64 * CH candidates per workgroup, rank tiles of 128 candidates
// (CH = 2, 4) or 64 (CH = 1).  Called by every thread behind the barrier that follows the last group's
// sums; `list` = C + 2 ints of LDS nobody else uses (the early-exit flags' place); leaves behind a
// barrier after which the tile may be overwritten.
template <int CH, int HALF1>
__device__ __forceinline__ void q_park_tile(const int tid, const int lane, const int wave, const bool owner,
                                            const bool valid, const bool active, const double acc, const unsigned st,
                                            const int64_t local, const int64_t c0, const int F, const unsigned list,
                                            const qr_split_args &sp, const double prune_sum, double *__restrict__ prob,
                                            unsigned &pk_at, unsigned &pk_end)
{
    constexpr int C = 64 * CH;
    const unsigned pk_cnt = list + 4u * C, pk_new = pk_cnt + 4u;
    bool open = false;
    unsigned my_i = 0;
    if (owner) {  // (whole waves: C is a multiple of 64)
        open = valid && active && !((acc + sp.rem) < prune_sum);
        const unsigned long long m = __ballot(open);
        if (m != 0ull) {
            unsigned pos = 0;
            if (lane == 0) pos = __hip_atomic_fetch_add(LDS_AT(lds_i32, pk_cnt), (int)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pos = (unsigned)__builtin_amdgcn_readfirstlane((int)pos);
            my_i = pos + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (open) *LDS_AT(lds_i32, list + 4u * my_i) = tid;  // its place in the workgroup's tile(s)
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (LDS only: see forest_qr_kernel)
    const unsigned P = (unsigned)__builtin_amdgcn_readfirstlane(*LDS_AT(const lds_i32, pk_cnt));
    if (P != 0u) {  // (uniform)
        const unsigned room = pk_end - pk_at;
        unsigned new_base = 0;
        if (P > room) {  // (uniform) a new block of 256 slots
            if (tid == 0) *LDS_AT(lds_i32, pk_new) = (int)atomicAdd(sp.cnt, 256u);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            new_base = (unsigned)__builtin_amdgcn_readfirstlane(*LDS_AT(const lds_i32, pk_new));
        }
        if (open) {
            const unsigned s = my_i < room ? pk_at + my_i : new_base + (my_i - room);
            sp.idx[s] = (int32_t)local;
            sp.acc[s] = acc;
            sp.st[s] = (uint8_t)st;
        }
        for (unsigned i0 = 0; i0 < P; i0 += 64u) {
            const unsigned pc = P - i0 < 64u ? P - i0 : 64u;
            const unsigned sh = pc <= 1u ? 0u : 32u - (unsigned)__builtin_clz(pc - 1u);
            const unsigned ci = (unsigned)lane & ((1u << sh) - 1u), rsub = (unsigned)lane >> sh;
            const int rows = 64 >> sh;
            if (ci < pc) {
                const unsigned i = i0 + ci;
                const int cand = *LDS_AT(const lds_i32, list + 4u * i);
                const unsigned s = i < room ? pk_at + i : new_base + (i - room);
                unsigned src, rstride;
                unsigned short *dst;
                size_t dstride;
                if (CH == 1) {  // [F][64] u16
                    src = (unsigned)(cand & 63) << 1;
                    rstride = 128u;
                    dst = sp.tiles + (size_t)(s >> 6) * (size_t)F * 64u + (s & 63u);
                    dstride = 64u;
                } else {        // [F][64][2] u16, the second 128 candidates at HALF1
                    src = (cand >= 128 ? (unsigned)HALF1 : 0u) + ((unsigned)(cand & 63) << 2) + ((unsigned)((cand >> 6) & 1) << 1);
                    rstride = 256u;
                    dst = sp.tiles + (size_t)(s >> 7) * (size_t)F * 128u + ((s & 63u) << 1) + ((s >> 6) & 1u);
                    dstride = 128u;
                }
                for (int r0 = wave * rows; r0 < F; r0 += 16 * rows) {
                    const int r = r0 + (int)rsub;
                    if (r < F) dst[(size_t)r * dstride] = *LDS_AT(const lds_u16, src + (unsigned)r * rstride);
                }
            }
        }
        if (P > room) {
            pk_at = new_base + (P - room);
            pk_end = new_base + 256u;
        } else {
            pk_at += P;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the codes are read
    if (tid == 0) *LDS_AT(lds_i32, pk_cnt) = 0;
    if (valid && !open) prob[c0 + local] = 0.0;
}

template <int CH, int WPT, int HALF1, bool PRUNE, bool EARLY, int SPLIT = 0>
__global__ __launch_bounds__(Q_THREADS) void forest_q_kernel(
    const v4u *__restrict__ img, const int4 *__restrict__ gtab, int n_grp,
    const int4 *__restrict__ ttab, int T, int t_div, int F, int dec_off, int val_off, int img_off,
    const q_slot_table slots_at,
    const unsigned short *__restrict__ qtiles, const uint8_t *__restrict__ status, int64_t c0,
    int64_t cn_arg, double *__restrict__ prob, double prune_sum, int warm_ahead, int dbg,
    long long *__restrict__ stamps, qr_split_args sp)
{
    static_assert(SPLIT == 0 || (!PRUNE && !EARLY), "the cut forest decides at the cut, not inside the kernel");
    constexpr int THREADS = Q_THREADS;
    // (SPLIT, the forest cut in two -- see forest_qr_kernel: 1 = head, parks the open candidates behind its
    // last group; 2 = tail over the parked ones)
    const int64_t cn = SPLIT == 2 ? (int64_t)__builtin_amdgcn_readfirstlane((int)*sp.cnt) : cn_arg;
    const int64_t sc0 = SPLIT == 2 ? 0 : c0;
    [[maybe_unused]] unsigned pk_at = 0, pk_end = 0;
    constexpr int C = 64 * CH;
    constexpr int PFN = 8;  // staging registers: 8 x 1024 x 16 B = 128 KiB per group (pk_q_stage_regs)
    constexpr int NCH = CH / WPT;  // walks per lane of one wave
    static_assert(THREADS >= C, "one thread per candidate owns the ordered sum");
    static_assert(WPT == 1 || (WPT == 2 && CH == 4), "two waves per tree = one per rank tile");
    static_assert(CH == 1 || CH == 2 || CH == 4, "64, 128 or 256 candidates per workgroup");
    static_assert(!EARLY || WPT == 2, "early staging is written for two waves per tree");
    const int done_off = dec_off + 4 * (C + 4);  // EARLY: per tree slot, waves done with it (counts up)
    extern __shared__ __attribute__((aligned(16))) char lds[];  // addressed absolutely from 0
    const int HB = CH == 1 ? F * 128 : F * 256;  // bytes of a rank tile ([F][64] u16 when CH = 1)
    const bool lds_at_zero = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char *)lds == 0u;
    if (!lds_at_zero && threadIdx.x == 0 && stamps) stamps[65535] = 2;

    // Persistent launch (gridDim.x < tiles): a workgroup takes tiles blockIdx.x, blockIdx.x +
    // gridDim.x, ...  While it walks the LAST tree group of a tile -- the staging registers are
    // idle then -- it fetches the next tile's rank codes and status bytes into registers and
    // commits them to LDS right after the group: a tile change then costs about what a
    // group change costs, not a cold start (two dependent memory latencies).
    const int64_t n_wg = (cn + C - 1) / C;
    unsigned stc_n[NCH] = {}, st_n = 0;
    bool tile_ready = false;  // (uniform) the tile of this trip is in LDS already
    for (int64_t wg = blockIdx.x; wg < n_wg; wg += gridDim.x) {
        // (the thread index is opaque per trip: otherwise every address derived from it that the
        // prologue and the epilogue of a tile use is hoisted out of this loop and stays in a
        // register through all the walks -- 28 more VGPRs, and the tile prefetch then spills)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int slot = wave / WPT, sub = wave % WPT;  // tree slot; which part of the candidates
        const unsigned lk0 = (unsigned)lane << (CH == 1 ? 1 : 2), lk1 = lk0 + 2u;
        if (wg != (int64_t)blockIdx.x) __syncthreads();  // nobody reads the previous trip's flags any more
        if (PRUNE)
            for (int i = tid; i < C + 3; i += THREADS) *LDS_AT(lds_i32, dec_off + 4 * i) = 0;
        if (EARLY && tid < 8) *LDS_AT(lds_i32, done_off + 4 * tid) = 0;
        if (SPLIT == 1 && wg == (int64_t)blockIdx.x && tid == 0) *LDS_AT(lds_i32, dec_off + 4 * C) = 0;  // parked in this tile
        const int64_t cbase = wg * C;  // first candidate of this workgroup (relative to c0)
        if (!tile_ready) {
            // rank tiles of 128 candidates each, consecutive in memory; the second one exists
            // only if it holds a candidate (the buffer ends with the last tile in use)
            const v4u *src = reinterpret_cast<const v4u *>(qtiles + (size_t)cbase * F);
            const int halves = (CH == 4 && cbase + 128 < cn) ? 2 : 1;
            const int nu = halves * (HB >> 4);
            for (int i = tid; i < nu; i += THREADS) {
                const int o = i << 4;
                *LDS_AT(lds_u4, o < HB ? o : o - HB + HALF1) = __builtin_nontemporal_load(src + i);
            }
        }
        // walk c of a lane = candidate lane + 64 (NCH * sub + c) of the workgroup
        unsigned stc[NCH];
        bool act[NCH];
        bool any_nan = false;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int64_t loc = cbase + lane + 64 * (NCH * sub + c);
            stc[c] = tile_ready ? stc_n[c] : (loc < cn ? status[sc0 + loc] : 0);
            act[c] = stc[c] != 0 && lds_at_zero;
            any_nan = any_nan || stc[c] == 2;
        }
        const bool wave_nan = __any(any_nan);  // a wave holding NaN features takes the slow walk
        // threads 0 .. C-1 also own one candidate each for the ordered sum
        const bool owner = tid < C;
        const int64_t local = cbase + tid;
        const bool valid = owner && local < cn;
        const unsigned st = tile_ready ? st_n : (valid ? status[sc0 + local] : 0);
        const bool active = st != 0 && lds_at_zero;
        const int64_t wg_next = wg + gridDim.x;
        bool fetched = false;  // (uniform) tile_u / stc_n / st_n hold the next tile

        Q_PF16(Q_PF_DECL)
        const v4u *pf_src;
        int pf_nu, pf_dst;
        const int pf_i0 = EARLY ? lane : tid, pf_stride = EARLY ? 64 : THREADS;
        int4 g_cur = gtab[0];
        int4 tt = ttab[min(g_cur.x + slot, T - 1)];  // this wave's tree of the group: offset, units, depth, root
        // what this thread stages of group `gi`: the whole group (all threads together), or,
        // EARLY, this wave's half of the tree of its slot
        auto stage_of = [&](const int4 gi, const int4 ti) {
            if (EARLY) {
                const int tu = ti.w, half = (tu + 1) >> 1, u0 = sub * half;
                pf_src = img + gi.z + (ti.x >> 4) + u0;
                pf_nu = slot < gi.y ? max(0, min(tu, u0 + half) - u0) : 0;
                pf_dst = img_off + slots_at.off[min(slot, 15)] + (u0 << 4);
            } else {
                pf_src = img + gi.z;
                pf_nu = gi.w;
                pf_dst = img_off;
            }
        };
        {
            stage_of(g_cur, tt);
            Q_PF16(Q_PF_LOAD)
            Q_PF16(Q_PF_STORE)
        }
        __syncthreads();  // rank tiles and first group are in LDS

#define Q_STAMP(slot_)                                                                   \
    do {                                                                                 \
        if ((dbg & 16) && stamps && blockIdx.x == 0 && lane == 0 && g < 32)              \
            stamps[((tid >> 6) * 32 + g) * 5 + (slot_)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
        double acc = 0.0;
        if constexpr (SPLIT == 2)
            if (valid) acc = sp.acc[local];  // the head's trees, already added in order
        unsigned warm_sink = 0;
        for (int g = 0; g < n_grp; g++) {  // uniform: every thread takes the same trips
            const int t0 = g_cur.x, gt = g_cur.y;
            const int4 g_nxt = gtab[g + 1];
            const int4 tt_nxt = ttab[min(g_nxt.x + slot, T - 1)];  // (scalar loads: in flight during the walk)
            Q_STAMP(0);
            // What the staging registers fetch during this walk: the next group, or (last group) the
            // next tile -- through ONE set of load instructions.  (With a second set for the tile the
            // compiler loads it into copies of the staging registers, registers the walk uses as
            // temporaries: the s_waitcnt vmcnt(0) that protects them stood in front of EVERY group's
            // walk and made each wave wait for its eight staging loads to return before walking.)
            // register q < 3: units ld_i0 + q * ld_stride of the first ld_lo; q >= 3: units
            // ld_i0 + q * ld_stride - ld_shift of the ld_hi that follow ld_base
            bool ld_on = false;
            int ld_i0 = pf_i0, ld_stride = pf_stride, ld_lo = 0, ld_hi = 0, ld_base = 0, ld_shift = 0;
            if (g + 1 < n_grp) {  // loads fly while this group is walked
                stage_of(g_nxt, tt_nxt);
                ld_lo = ld_hi = pf_nu;
                ld_on = true;
            } else if (wg_next < n_wg && (HB >> 4) <= (CH == 4 ? 3 : 6) * THREADS) {
                // last group of this tile: the next tile travels global -> VGPR during the walk
                const int64_t cb = wg_next * C;
                // (into the staging registers, which carry nothing during the last group: registers
                // 0-2 the first rank tile, 3-5 the second; <= 3 x 1024 units of 16 B per tile; CH < 4:
                // one tile of <= 6 x 1024 units)
                pf_src = reinterpret_cast<const v4u *>(qtiles + (size_t)cb * F);
                const int upt = HB >> 4;
                ld_i0 = tid;
                ld_stride = THREADS;
                ld_lo = upt;
                if (CH == 4) {
                    ld_hi = cb + 128 < cn ? upt : 0;
                    ld_base = upt;
                    ld_shift = 3 * THREADS;
                } else {
                    ld_hi = upt;
                }
                ld_on = true;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const int64_t loc = cb + lane + 64 * (NCH * sub + c);
                    stc_n[c] = loc < cn ? status[sc0 + loc] : 0;
                }
                st_n = (owner && cb + tid < cn) ? status[sc0 + cb + tid] : 0;
                fetched = true;
            } else if (warm_ahead > 0 && gridDim.x >= n_wg) {
                // last group: pull the tiles of the workgroup that follows this one on this XCD
                // into its L2, one dword per 128-byte line; the value is never used
                const int64_t ahead = wg + warm_ahead;
                if ((ahead + 1) * C <= cn)
                    for (int line = tid; line < F * CH; line += THREADS)
                        warm_sink += reinterpret_cast<const unsigned *>(qtiles + (size_t)ahead * C * F)[line * 32];
            }
            if (ld_on) { Q_PF16(Q_PF_LOADX) }
            bool walk[NCH];
            bool any_walk = false;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                walk[c] = act[c] &&
                          (!PRUNE || *LDS_AT(lds_i32, dec_off + 4 * (lane + 64 * (NCH * sub + c))) == 0);
                any_walk = any_walk || walk[c];
            }
            if (slot < gt && !(dbg & 2) && __any(any_walk)) {
                double v[NCH];
                const unsigned tbase = (unsigned)(EARLY ? img_off + slots_at.off[min(slot, 15)] : img_off + tt.x);
                const unsigned root = (unsigned)tt.z;
                // lanes without a live candidate walk along (their values are not stored)
#define Q_WALK_POS(X0_, NAN_, POS_) \
    q_walk<NCH, X0_, HALF1, NAN_, false, POS_>(root, tt.y & 0xFFFF, tbase, lk0, CH == 1 ? (unsigned)tt.y >> 16 : lk1, v, \
                                               CH == 1 ? 0u : (unsigned)tt.y >> 16)
#define Q_WALK(X0_, NAN_)                              \
    do {                                               \
        if (dbg & 32) Q_WALK_POS(X0_, NAN_, -1);       \
        else if ((wave >> 2) == 0) Q_WALK_POS(X0_, NAN_, 0); \
        else if ((wave >> 2) == 1) Q_WALK_POS(X0_, NAN_, 1); \
        else if ((wave >> 2) == 2) Q_WALK_POS(X0_, NAN_, 2); \
        else Q_WALK_POS(X0_, NAN_, 3);                 \
    } while (0)
                if (WPT == 2 && sub) {  // the second rank tile (the tile index is an immediate offset)
                    if (wave_nan) Q_WALK(HALF1, true);
                    else Q_WALK(HALF1, false);
                } else {
                    if (dbg & 8) q_walk<NCH, 0, HALF1, false, true>(root, tt.y & 0xFFFF, tbase, lk0, lk1, v);  // wrong results
                    else if (wave_nan) Q_WALK(0, true);
                    else Q_WALK(0, false);
                }
#undef Q_WALK
#undef Q_WALK_POS
#pragma unroll
                for (int c = 0; c < NCH; c++)
                    if (walk[c])
                        *LDS_AT(lds_f64, val_off + (slot * C + lane + 64 * (NCH * sub + c)) * 8) = v[c];
            }
            Q_STAMP(1);
            if (EARLY) {
                // (the staging loads were issued in front of the walk: an explicit vmcnt(0) here costs
                // nothing and lets the compiler see that no load is pending on a staging register when
                // the next group's loads overwrite it -- without it every load of a group waits for
                // the one before, see forest_q2_kernel: 4.39 -> 4.10 ms, still behind 3.79)
                __builtin_amdgcn_s_waitcnt(0x0F70);
                // this wave no longer reads the tree of its slot; when its partner does not
                // either, each stages its half of the slot's next tree.  The partner waits for
                // nothing before it counts itself in, so the wait is bounded by its walk.
                if (lane == 0)
                    __hip_atomic_fetch_add(LDS_AT(lds_i32, done_off + 4 * slot), 1, __ATOMIC_RELEASE,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                if (g + 1 < n_grp && pf_nu > 0 && !(dbg & 4)) {
                    bool both = false;
                    for (int spin = 0; spin < (1 << 22) && !both; spin++) {
                        both = __hip_atomic_load(LDS_AT(lds_i32, done_off + 4 * slot), __ATOMIC_ACQUIRE,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP) >= 2 * (g + 1);
                        if (!both) __builtin_amdgcn_s_sleep(2);
                    }
                    if (!both && lane == 0 && stamps) stamps[65535] = 1;  // reported by the host as an error
                    Q_PF16(Q_PF_STORE)
                }
            }
            __syncthreads();  // every walk of the group is done (EARLY: and the next group staged)
            Q_STAMP(2);
            // Everything this wave has loaded (the next group, or the next tile and its status bytes)
            // has had the whole walk to arrive: an explicit vmcnt(0) here is free, and it is what
            // lets the compiler see that no load is pending on any register when the next iteration
            // starts.  Without it the status bytes the LAST group prefetches count as pending all
            // around the loop, their register is reused in front of the walk, and the s_waitcnt
            // vmcnt(0) the compiler puts there makes every wave wait for its eight staging loads to
            // RETURN before it walks (found in the assembly in round 3; it is the "12 % the staging
            // loads cost" of the ablation above).
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (!EARLY && g + 1 < n_grp && !(dbg & 4)) { Q_PF16(Q_PF_STORE) }  // (dbg 4: timing ablation, wrong results)
            const bool undecided = !PRUNE || (owner && *LDS_AT(lds_i32, dec_off + 4 * (tid & (C - 1))) == 0);
            if (owner && active && undecided) {
                for (int j = 0; j < gt; j++) acc += *LDS_AT(lds_f64, val_off + (j * C + tid) * 8);  // tree order
                if (PRUNE) {
                    // every remaining tree adds at most 1.0: if even that cannot lift the sum to
                    // the bound (pk_prune_bound, pk_common.h: thre*T less a proven rounding margin) the final p is
                    // <= thre and the pixel is not reported -- stop walking it
                    const bool out = (acc + (double)(T - (t0 + gt))) < prune_sum;
                    if (out) {
                        *LDS_AT(lds_i32, dec_off + 4 * tid) = 1;
                        acc = 0.0;  // reported probability of a pruned candidate: 0
                    } else {
                        *LDS_AT(lds_i32, dec_off + 4 * (C + (g % 3))) = 1;  // still an open candidate
                    }
                }
            }
            Q_STAMP(3);
            __syncthreads();  // next group staged; values consumed; votes cast
            bool all_done = false;
            if (PRUNE) {
                // a vote word is set before this barrier, read after it and cleared two groups
                // ahead, so a clear and a set of the same word are always a barrier apart
                all_done = *LDS_AT(lds_i32, dec_off + 4 * (C + (g % 3))) == 0;
                if (tid == 0) *LDS_AT(lds_i32, dec_off + 4 * (C + ((g + 2) % 3))) = 0;
            }
            Q_STAMP(4);
            g_cur = g_nxt;
            tt = tt_nxt;
            if (all_done) break;
        }
        if constexpr (SPLIT == 1) {
            q_park_tile<CH, HALF1>(tid, lane, wave, owner, valid, active, acc, st, local, c0, F, (unsigned)dec_off, sp,
                                   prune_sum, prob, pk_at, pk_end);
        } else if constexpr (SPLIT == 2) {
            if (valid && active) prob[c0 + sp.idx[local]] = acc / (double)t_div;  // (status 0: a slot nobody took)
        } else {
            if (valid) prob[c0 + local] = active ? acc / (double)t_div : 0.0;  // (t_div: the model's trees; T: the image's pieces)
        }
        if (warm_sink == 0x9e3779b9u && stamps) stamps[65534] = 1;  // keeps the warm-ahead loads alive
        tile_ready = fetched;
        if (fetched) {  // every walk of this tile is behind the last barrier: its rank tiles can go
            const int64_t cb = wg_next * C;
            const int upt = HB >> 4;
            const bool two = CH == 4 && cb + 128 < cn;
#define Q_TILE_STORE(q, half, j)                                                                        \
    if ((j) * THREADS < upt && tid + (j) * THREADS < upt && ((half) == 0 || two))                       \
        *LDS_AT(lds_u4, (half) * HALF1 + ((tid + (j) * THREADS) << 4)) = pf##q;
            if (CH == 4) {
                Q_TILE_STORE(0, 0, 0) Q_TILE_STORE(1, 0, 1) Q_TILE_STORE(2, 0, 2)
                Q_TILE_STORE(3, 1, 0) Q_TILE_STORE(4, 1, 1) Q_TILE_STORE(5, 1, 2)
            } else {
                Q_TILE_STORE(0, 0, 0) Q_TILE_STORE(1, 0, 1) Q_TILE_STORE(2, 0, 2)
                Q_TILE_STORE(3, 0, 3) Q_TILE_STORE(4, 0, 4) Q_TILE_STORE(5, 0, 5)
            }
#undef Q_TILE_STORE
        }
    }
    if constexpr (SPLIT == 1)  // what is left of the last block: nobody
        for (unsigned i = pk_at + threadIdx.x; i < pk_end; i += THREADS) sp.st[i] = 0;
#undef Q_STAMP
}

// ------------------------------------------------------------------------
// forest_qr_kernel (round 4) = forest_q_kernel<4, 2, HALF1, PRUNE, false> -- 256 candidates, eight
// trees, two waves per tree -- with the STAGING REGISTERS TAKEN AWAY FROM THE COMPILER and a
// walk path without scalar tests.
//
// Round 3 measured (EXPERIMENTS.md 4.2 iii): the 67 wave-loads of the next tree group, issued by all
// sixteen waves at once in front of the walk, queue up in the CU's memory pipeline and every wave
// stands still until its own are accepted; issued from INSIDE the walk, a few levels in and one
// SIMD position at a time, the stage is 7 % faster -- but every C++ form of that either moved the
// eight 16-byte staging registers to scratch, doubled the register demand or slowed the walk,
// because the compiler has to keep registers with loads in flight alive across the unrolled walk.
// Here the kernel is compiled for PK_QR_VGPRS (72) VGPRs (amdgpu_num_vgpr) and v72..v127 -- which a
// 1024-thread workgroup owns anyway -- are used by inline assembly only: global_load_dwordx4
// into fixed registers wherever the walk wants them, one explicit s_waitcnt vmcnt(0) behind the
// barrier, ds_write_b128 from the same registers.  The compiler never sees a value in them, so
// it neither waits for them nor moves them, and its own s_waitcnt arithmetic stays safe (extra
// loads in flight only make its vmcnt waits conservative: loads return in order).
// v96..v127 = G0..G7, 16-KiB rows of the next tree group; v72..v95 = T0..T5, the next tile's two
// rank tiles (three rows each).
//
// What the walk phase is bound by (round 4, same-box A/Bs, profiles/r04_ab_*.log): each wave's own
// instruction stream -- four waves per SIMD, two dependent chains each, an instruction of a wave
// issues every ~5 cycles at best.  Every scalar instruction in that stream costs like a vector one:
// three bit tests with (taken) branches for rows a group does not have cost 1.8 %, the five tests
// of the rows it has another 1.2 %, six more for the tile's rows 5 %.  Hence
//   * NR rows are loaded WITHOUT tests (NR = rows of the forest's largest group, 4 or 5; a shorter
//     group's surplus rows read what lies behind it -- the image and the rank-tile buffer are padded,
//     PK_Q_PAD_BYTES -- and are not stored);
//   * a wave's role (rank tile, position in its SIMD) is dispatched ONCE per kernel, not per walk:
//     the whole persistent loop exists eight times (qr_body);
//   * a tree of >= 16 levels is walked 16 + 4k levels (a walk that has reached its leaf stays there:
//     up to three idle levels instead of three more loop tests).
// ------------------------------------------------------------------------
#ifndef PK_QR_AT
#define PK_QR_AT(pos) ((pos) + 1)  // the staging loads start behind this many PAIRS of levels
#endif
#ifndef PK_QR_VGPRS
#define PK_QR_VGPRS 72
#endif
#ifndef PK_QR_SPREAD
#define PK_QR_SPREAD 1             // issue slots per pair of levels from there on (0: all at once)
#endif

// A barrier that orders LDS accesses only (the parking of the cut forest's head).  __syncthreads() also
// waits for every global access of the wave to be acknowledged (s_waitcnt vmcnt(0) in front of s_barrier):
// in the parking that is the tile's probability and record stores, which nobody in the workgroup reads --
// 4 000 cycles per tile.  (At the TOP of a tile the same exchange changed nothing: there the stores have had
// the tile's LDS commits to be acknowledged behind -- same-box A/B, 2.811 vs 2.809 ms -- and __syncthreads()
// stays.)
#ifndef PK_QR_RAWBAR
#define PK_QR_RAWBAR 1
#endif
#if PK_QR_RAWBAR
#define QR_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define QR_LDS_BARRIER() __syncthreads()
#endif
#define QR_LD_U(R, ptr) asm volatile("global_load_dwordx4 " R ", %0, %1" ::"v"(voff), "s"(ptr))
// a row behind a test of its bit (the tile's rows, once per tile): the test is inside the assembly --
// as a C++ `if` the compiler lays every conditional load out of line (two taken branches per load)
// (a rank tile is read once: fetched non-temporally it does not push the band -- which the next chunk's extractor
// reads again -- and the tree image out of the L2 / Infinity Cache: forest 2.11 -> 2.08 and the extractor behind it
// 1.04 -> 0.98 ms per step of config 2, profiles/r06_ab_nontemporal.log)
#ifndef PK_QR_TILE_PLAIN
#define QR_TILE_POLICY " nt"
#else
#define QR_TILE_POLICY ""
#endif
#define QR_LD_ASM(bit, R, ptr)                                                              \
    asm volatile("s_bitcmp1_b32 %0, " #bit "\n\t"                                           \
                 "s_cbranch_scc0 .Lqr" #bit "_%=\n\t"                                       \
                 "global_load_dwordx4 " R ", %1, %2" QR_TILE_POLICY "\n"                     \
                 ".Lqr" #bit "_%=:" ::"s"(mask),                                            \
                 "v"(voff), "s"(ptr)                                                        \
                 : "scc")
// issue slot S = row S of the group (rows >= NR do not exist)
#ifdef PK_QR_VOFFS
// (variant: the row's offset added to the lane offset in a VALU instruction instead of two scalar ones)
#define QR_LD_ROW(R, k) asm volatile("global_load_dwordx4 " R ", %0, %1" ::"v"(voff + (k) * 16384u), "s"(gb))
#else
#define QR_LD_ROW(R, k) QR_LD_U(R, gb + (k) * 16384)
#endif
template <int S, int NR>
__device__ __forceinline__ void qr_issue_slot(const char *gb, unsigned voff)
{
    if constexpr (S < NR) {
        if constexpr (S == 0) QR_LD_ROW("v[96:99]", 0);
        if constexpr (S == 1) QR_LD_ROW("v[100:103]", 1);
        if constexpr (S == 2) QR_LD_ROW("v[104:107]", 2);
        if constexpr (S == 3) QR_LD_ROW("v[108:111]", 3);
        if constexpr (S == 4) QR_LD_ROW("v[112:115]", 4);
        if constexpr (S == 5) QR_LD_ROW("v[116:119]", 5);
        if constexpr (S == 6) QR_LD_ROW("v[120:123]", 6);
        if constexpr (S == 7) QR_LD_ROW("v[124:127]", 7);
    }
}
template <int LO, int HI, int NR>
__device__ __forceinline__ void qr_issue(const char *gb, unsigned voff)
{
    if constexpr (LO < HI) {
        qr_issue_slot<LO, NR>(gb, voff);
        qr_issue<LO + 1, HI, NR>(gb, voff);
    }
}
// the next tile's rows (once per tile, in front of the last group's walk); mask bits 8..13
__device__ __forceinline__ void qr_issue_tile(unsigned tmask, const char *tb0, const char *tb1, unsigned voff)
{
    const unsigned mask = __builtin_amdgcn_readfirstlane(tmask);
    QR_LD_ASM(8, "v[72:75]", tb0);
    QR_LD_ASM(9, "v[76:79]", tb0 + 16384);
    QR_LD_ASM(10, "v[80:83]", tb0 + 2 * 16384);
    QR_LD_ASM(11, "v[84:87]", tb1);
    QR_LD_ASM(12, "v[88:91]", tb1 + 16384);
    QR_LD_ASM(13, "v[92:95]", tb1 + 2 * 16384);
}
#define QR_GREGS(X)                                                                         \
    X(0, "v[96:99]") X(1, "v[100:103]") X(2, "v[104:107]") X(3, "v[108:111]")               \
    X(4, "v[112:115]") X(5, "v[116:119]") X(6, "v[120:123]") X(7, "v[124:127]")
#define QR_TREGS(X)                                                                         \
    X(0, "v[72:75]") X(1, "v[76:79]") X(2, "v[80:83]") X(3, "v[84:87]") X(4, "v[88:91]") X(5, "v[92:95]")
#define QR_ST_GROUP(q, R)                                                                   \
    if ((q) < NR && (q) * 1024 < st_nu) {                                                   \
        if (utid + (q) * 1024 < st_nu)                                                      \
            asm volatile("ds_write_b128 %0, " R " offset:%1" ::"v"((q) < 4 ? st_a0 : st_a1), \
                         "n"(((q) & 3) * 16384)                                             \
                         : "memory");                                                       \
    }
#define QR_ST_TILE(q, R)                                                                    \
    if (((q) % 3) * 1024 < upt && ((q) < 3 || two)) {                                       \
        if (utid + ((q) % 3) * 1024 < upt)                                                  \
            asm volatile("ds_write_b128 %0, " R " offset:%1" ::"v"((q) < 3 ? tl_a0 : tl_a1), \
                         "n"(((q) % 3) * 16384)                                             \
                         : "memory");                                                       \
    }

// The walk of q_walk<2, X0, HALF1, WITH_NAN, false, POS> (same levels, same priorities) with the
// staging loads issued from inside: issue slot s goes out in front of pair PK_QR_AT(POS) +
// s / PK_QR_SPREAD of the first sixteen levels (what they do not reach: behind them;
// PK_QR_SPREAD = 0: all slots in front of pair PK_QR_AT(POS)).  Trees of fewer than 16 levels:
// everything in front of the walk.  Behind the first sixteen levels a tree is walked in blocks of
// four (fewer than 16: in pairs): a walk that has reached its leaf stays there (pk_qimage.hip), so
// the up to three surplus levels change nothing.
template <int X0, int HALF1, bool WITH_NAN, int POS, int NR>
__device__ __forceinline__ void qr_walk(unsigned root, int depth, unsigned tbase, unsigned lk0, unsigned lk1,
                                        double (&v)[2], const char *gb, unsigned voff, unsigned nsplit = 0u)
{
    unsigned w[2] = {root, root};
    constexpr int PT[4][6] = {PK_Q_PRIO_TAB};
    constexpr int PP = POS & 3;
    constexpr int AT = PK_QR_AT(PP);
    constexpr int SP = PK_QR_SPREAD;
#ifdef PK_QR_NOPRIO
#define QR_SETPRIO(p_) do {} while (0)
#else
#define QR_SETPRIO(p_) __builtin_amdgcn_s_setprio(p_)
#endif
    // slots [lo, hi) in front of pair k
#define QR_SLOTS_AT(k_)                                                                          \
    do {                                                                                         \
        if constexpr (SP == 0) {                                                                 \
            if constexpr (AT == (k_)) qr_issue<0, 8, NR>(gb, voff);                              \
        } else if constexpr ((k_) >= AT && ((k_) - AT) * SP < 8) {                               \
            qr_issue<((k_) - AT) * SP, (((k_) - AT + 1) * SP < 8 ? ((k_) - AT + 1) * SP : 8), NR>(gb, voff); \
        }                                                                                        \
    } while (0)
#define QR_TWO(k_)                                                     \
    do {                                                               \
        QR_SLOTS_AT(k_);                                               \
        q_level<2, X0, HALF1, WITH_NAN, false>(w, tbase, lk0, lk1, nsplit);    \
        q_level<2, X0, HALF1, WITH_NAN, false>(w, tbase, lk0, lk1, nsplit);    \
    } while (0)
    if (__builtin_expect(depth >= 16, 1)) {
        QR_SETPRIO(PT[PP][0]);
        QR_TWO(0);
        QR_TWO(1);
        QR_SETPRIO(PT[PP][1]);
        QR_TWO(2);
        QR_TWO(3);
        QR_SETPRIO(PT[PP][2]);
        QR_TWO(4);
        QR_TWO(5);
        QR_SETPRIO(PT[PP][3]);
        QR_TWO(6);
        QR_TWO(7);
        // what the sixteen levels did not reach
        constexpr int done = SP == 0 ? (AT < 8 ? 8 : 0) : (AT >= 8 ? 0 : ((8 - AT) * SP < 8 ? (8 - AT) * SP : 8));
        qr_issue<done, 8, NR>(gb, voff);
#undef QR_SLOTS_AT
#define QR_SLOTS_AT(k_) do {} while (0)
        for (int n4 = (depth - 16 + 3) >> 2; n4 > 0; n4--) {
            QR_SETPRIO(PT[PP][4]);
            QR_TWO(-1);
            QR_SETPRIO(PT[PP][5]);
            QR_TWO(-1);
        }
    } else {
        qr_issue<0, 8, NR>(gb, voff);  // (short trees)
        QR_SETPRIO(PT[PP][0]);
        for (int n2 = (depth + 1) >> 1; n2 > 0; n2--) QR_TWO(-1);
    }
#undef QR_TWO
#undef QR_SLOTS_AT
    QR_SETPRIO(0);
#pragma unroll
    for (int c = 0; c < 2; c++) v[c] = *LDS_AT(const lds_f64, tbase + ((q_pair_index<false>(w[c]) + 1) << 3));
}

struct qr_args {
    const char *img_b;
    const int4 *gtab, *ttab;
    int n_grp, T, t_div, F, dec_off, val_off, img_off;  // T: trees of the image (pieces); t_div: the model's trees
    const unsigned short *qtiles;
    const uint8_t *status;
    int64_t c0, cn;
    double *prob;
    double prune_sum;
    int opt, dbg;
    long long *stamps;
    // the forest cut in two at a group boundary (SPLIT, see forest_qr_kernel):
    // 1 = head: groups [0, n_grp) of the table handed in, then every candidate that is still OPEN (its sum
    //     plus one per remaining tree could exceed prune_sum) is parked -- partial sum, status, index and
    //     its column of rank codes, in tile format -- behind a device-side counter;
    // 2 = tail: the remaining groups over the parked candidates (their number is read from the device),
    //     sums continued in tree order, probabilities written back to the candidates' own places
    unsigned *s_cnt;            // 1: slots handed out so far; 2: candidates of this launch
    int32_t *s_idx;             // [slot] index of the candidate in its chunk
    double *s_acc;              // [slot] partial sum of the head's trees, tree order
    uint8_t *s_st;              // [slot] status byte
    unsigned short *s_tiles;    // 1: rank tiles of the parked candidates, slot order
    double split_rem;           // 1: trees (of the image) behind the cut
};

// the persistent loop of one wave role: X0 = LDS offset of its rank tile, POS = its position among
// the four waves of its SIMD.  opt bits: 1 PREF0 (the last group of a tile also fetches the next
// tile's first group: a tile change then has no exposed load)
template <int HALF1, bool PRUNE, int NR, int X0, int POS, int SPLIT>
__device__ __forceinline__ void qr_body(const qr_args &A)
{
    static_assert(SPLIT == 0 || !PRUNE, "the cut forest decides at the cut, not inside the kernel");
    constexpr int THREADS = Q_THREADS;
    constexpr int C = 256;
    constexpr int NCH = 2;  // walks per lane of one wave
    constexpr int sub = X0 != 0;
    const int n_grp = A.n_grp, T = A.T, F = A.F, dec_off = A.dec_off, val_off = A.val_off, img_off = A.img_off;
    const int dbg = A.dbg;
    [[maybe_unused]] long long *const stamps = A.stamps;  // (used by builds with -DPK_QR_STAMPS)
    const int64_t c0 = A.c0;
    const int64_t sc0 = SPLIT == 2 ? 0 : c0;  // (the tail's status bytes are the parked ones, in slot order)
    // (tail of a cut forest: as many candidates as the head parked)
    const int64_t cn = SPLIT == 2 ? (int64_t)__builtin_amdgcn_readfirstlane((int)*A.s_cnt) : A.cn;
    const uint8_t *const status = A.status;
    const int HB = F * 256;  // bytes of a rank tile [F][64][2] u16
    const int upt = HB >> 4;
    const bool lds_ok = A.opt >= 0;  // (cleared when the dynamic LDS does not start at address 0)
    const bool pref0 = (A.opt & 1) != 0;

    const int64_t n_wg = (cn + C - 1) / C;
    unsigned stc_n[NCH] = {}, st_n = 0;
    bool tile_ready = false;  // (uniform) the tile of this trip and its first tree group are in LDS already
    [[maybe_unused]] unsigned pk_at = 0, pk_end = 0;  // SPLIT 1: the next free parking slot and the end of this workgroup's block
    for (int64_t wg = blockIdx.x; wg < n_wg; wg += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // (see forest_q_kernel: keeps per-trip addresses inside the trip)
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int slot = wave >> 1;  // tree slot
        const unsigned lk0 = (unsigned)lane << 2, lk1 = lk0 + 2u;
        // this thread's 16 bytes of a 16-KiB row.  (PK_QR_REV: unit 1023 - tid, so that the partial last row
        // of a group is stored by the LAST waves, not by the four owner waves that also add the values)
#ifndef PK_QR_NO_REV
        const int utid = (THREADS - 1) - tid;
#else
        const int utid = tid;
#endif
        const unsigned voff = (unsigned)utid << 4;
        if (wg != (int64_t)blockIdx.x && !tile_ready) __syncthreads();  // nobody reads the previous trip's tile any more
        if (PRUNE)
            for (int i = tid; i < C + 3; i += THREADS) *LDS_AT(lds_i32, dec_off + 4 * i) = 0;
        if (SPLIT == 1 && wg == (int64_t)blockIdx.x && tid == 0) *LDS_AT(lds_i32, dec_off + 4 * 256) = 0;  // parked in this tile
        const int64_t cbase = wg * C;
        int4 g_cur = A.gtab[0];
        int4 tt = A.ttab[min(g_cur.x + slot, T - 1)];  // this wave's tree: offset, depth, root word, units
        int st_nu = g_cur.w;                            // units of 16 B of the group in the G registers
        const unsigned st_a0 = (unsigned)img_off + voff, st_a1 = st_a0 + 65536u;
        if (!tile_ready) {
            const v4u *src = reinterpret_cast<const v4u *>(A.qtiles + (size_t)cbase * F);
            const int halves = cbase + 128 < cn ? 2 : 1;
            const int nu = halves * upt;
            for (int i = tid; i < nu; i += THREADS) {
                const int o = i << 4;
                *LDS_AT(lds_u4, o < HB ? o : o - HB + HALF1) = __builtin_nontemporal_load(src + i);
            }
            // first group: global -> registers -> LDS
            qr_issue<0, 8, NR>(A.img_b + (size_t)g_cur.z * 16, voff);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            QR_GREGS(QR_ST_GROUP)
        }
        unsigned stc[NCH];
        bool act[NCH];
        bool any_nan = false;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int64_t loc = cbase + lane + 64 * (NCH * sub + c);
            stc[c] = tile_ready ? stc_n[c] : (loc < cn ? status[sc0 + loc] : 0);
            act[c] = stc[c] != 0 && lds_ok;
            any_nan = any_nan || stc[c] == 2;
        }
        const bool wave_nan = __any(any_nan);
        const bool owner = tid < C;
        const int64_t local = cbase + tid;
        const bool valid = owner && local < cn;
        const unsigned st = tile_ready ? st_n : (valid ? status[sc0 + local] : 0);
        const bool active = st != 0 && lds_ok;
        const int64_t wg_next = wg + gridDim.x;
        bool fetched = false;
        __syncthreads();  // rank tiles and first group are in LDS
        // in-kernel stamps (tools/stamps.py) exist in builds with -DPK_QR_STAMPS only: every test of a
        // debug flag is an instruction in the wave's stream, which is what bounds this kernel
#ifdef PK_QR_STAMPS
        if ((dbg & 16) && stamps && blockIdx.x == 0 && lane == 0 && wg != (int64_t)blockIdx.x)
            stamps[((tid >> 6) * 32 + 31) * 5 + 4] = (long long)__builtin_amdgcn_s_memtime();
#define QR_STAMP(slot_)                                                                  \
    do {                                                                                 \
        if ((dbg & 16) && stamps && blockIdx.x == 0 && lane == 0 && g < 31)              \
            stamps[((tid >> 6) * 32 + g) * 5 + (slot_)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define QR_STAMP(slot_) do {} while (0)
#endif
        double acc = 0.0;
        if constexpr (SPLIT == 2)
            if (valid) acc = A.s_acc[local];  // the head's trees, already added in order
        for (int g = 0; g < n_grp; g++) {  // uniform: every thread takes the same trips
            const int t0 = g_cur.x, gt = g_cur.y;
            const int4 g_nxt = A.gtab[g + 1];
            const int4 tt_nxt = A.ttab[min(g_nxt.x + slot, T - 1)];
            QR_STAMP(0);
            // what the G registers fetch during this walk: the next group -- or, in a tile's last group,
            // (PREF0) the first group again, for the next tile.  (Without PREF0, or without a next tile,
            // the rows loaded then are not used: the walk's loads carry no test.)
            const char *gb = A.img_b + (size_t)(g + 1 < n_grp ? g_nxt.z : A.gtab[0].z) * 16;
            if (g + 1 >= n_grp && wg_next < n_wg && upt <= 3 * THREADS) {
                // last group of this tile: the next tile travels global -> registers during the walk
                // (T0-2 its first rank tile, T3-5 the second) with its status bytes
                const int64_t cb = wg_next * C;
                const char *tb0 = reinterpret_cast<const char *>(A.qtiles + (size_t)cb * F);
                const unsigned rows = (1u << ((upt + 1023) >> 10)) - 1u;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const int64_t loc = cb + lane + 64 * (NCH * sub + c);
                    stc_n[c] = loc < cn ? status[sc0 + loc] : 0;
                }
                st_n = (owner && cb + tid < cn) ? status[sc0 + cb + tid] : 0;
                fetched = true;
                qr_issue_tile((rows | (cb + 128 < cn ? rows << 3 : 0u)) << 8, tb0, tb0 + HB, voff);
            }
            bool walk[NCH];
            bool any_walk = false;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                walk[c] = act[c] &&
                          (!PRUNE || *LDS_AT(lds_i32, dec_off + 4 * (lane + 64 * (NCH * sub + c))) == 0);
                any_walk = any_walk || walk[c];
            }
            if (__builtin_expect(slot < gt && !(dbg & 2) && __any(any_walk), 1)) {
                double v[NCH];
                const unsigned tbase = (unsigned)(img_off + tt.x);
                const unsigned root = (unsigned)tt.z;
                // (tt.y: levels to walk | the tree's split << 16, non-zero for the 12-bit rank word only)
                if (__builtin_expect(wave_nan, 0))
                    qr_walk<X0, HALF1, true, POS, NR>(root, tt.y & 0xFFFF, tbase, lk0, lk1, v, gb, voff, (unsigned)tt.y >> 16);
                else qr_walk<X0, HALF1, false, POS, NR>(root, tt.y & 0xFFFF, tbase, lk0, lk1, v, gb, voff);
#pragma unroll
                for (int c = 0; c < NCH; c++)
                    if (walk[c])
                        *LDS_AT(lds_f64, val_off + (slot * C + lane + 64 * (NCH * sub + c)) * 8) = v[c];
            } else {
                qr_issue<0, 8, NR>(gb, voff);  // a wave without a walk still moves its share
            }
            QR_STAMP(1);
            __syncthreads();  // every walk of the group is done
            QR_STAMP(2);
            // everything this wave asked for has had the whole walk to arrive
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PK_QR_SUM_LAST
            if (g + 1 < n_grp && !(dbg & 4)) {  // (dbg 4: timing ablation, wrong results)
                st_nu = g_nxt.w;
                QR_GREGS(QR_ST_GROUP)
            }
#endif
            const bool undecided = !PRUNE || (owner && *LDS_AT(lds_i32, dec_off + 4 * (tid & (C - 1))) == 0);
            if (owner && active && undecided) {
                // tree order (sklearn's sequential sum).  The group's parked values are read TOGETHER and then
                // added in order -- a slot without a tree adds +0.0, and x + 0.0 == x for the never-negative
                // sum.  (Rounds 2-4: a loop over the group's trees, every add behind its own LDS round trip.)
                double pv[8];
#pragma unroll
                for (int j = 0; j < 8; j++) pv[j] = *LDS_AT(lds_f64, val_off + (j * C + tid) * 8);
#pragma unroll
                for (int j = 0; j < 8; j++) acc += (j < gt) ? pv[j] : 0.0;
                if (PRUNE) {
                    // every remaining tree adds at most 1.0: if even that cannot lift the sum to the bound
                    // (pk_prune_bound: thre*T less a proven rounding margin) the pixel is not reported
                    const bool out = (acc + (double)(T - (t0 + gt))) < A.prune_sum;
                    if (out) {
                        *LDS_AT(lds_i32, dec_off + 4 * tid) = 1;
                        acc = 0.0;  // reported probability of a pruned candidate: 0
                    } else {
                        *LDS_AT(lds_i32, dec_off + 4 * (C + (g % 3))) = 1;  // still an open candidate
                    }
                }
            }
#ifndef PK_QR_SUM_LAST
            // (the owners add the parked values first, then everybody stores: 1.5 % over the other order,
            // with the partial last row stored by the last waves -- profiles/r04_ab_commit_order.log)
            if (g + 1 < n_grp && !(dbg & 4)) {
                st_nu = g_nxt.w;
                QR_GREGS(QR_ST_GROUP)
            }
#endif
            QR_STAMP(3);
            if (g + 1 < n_grp || !fetched) __syncthreads();  // next group staged; values consumed; votes cast
            bool all_done = false;
            if (PRUNE) {
                all_done = *LDS_AT(lds_i32, dec_off + 4 * (C + (g % 3))) == 0;
                if (tid == 0) *LDS_AT(lds_i32, dec_off + 4 * (C + ((g + 2) % 3))) = 0;
            }
            QR_STAMP(4);
            g_cur = g_nxt;
            tt = tt_nxt;
            if (all_done) break;
        }
#ifdef PK_QR_STAMPS
#define QR_TSTAMP(k_)                                                                    \
    do {                                                                                 \
        if ((dbg & 16) && stamps && blockIdx.x == 0 && lane == 0 && fetched)             \
            stamps[((tid >> 6) * 32 + 31) * 5 + (k_)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define QR_TSTAMP(k_) do {} while (0)
#endif
        QR_TSTAMP(0);
        if constexpr (SPLIT == 1) {
            // The cut.  A candidate whose sum cannot reach prune_sum even if every tree behind the cut adds
            // 1.0 is decided (the rule of the PRUNE instantiations; its reported probability is 0); the others
            // are parked for the tail: partial sum, status, index and their column of rank codes, in tile
            // format.  Slot order changes no bit of the result: every candidate is scored on its own.
            //  * The tile's open candidates are listed in LDS (the early-exit flags' place: this instantiation
            //    has no in-kernel exit); behind a barrier ALL sixteen waves copy the codes -- lanes = parked
            //    candidates (different columns of the tile: different banks), a wave takes every sixteenth
            //    row.  (First version: the four owner waves, lanes = rows.  A column of the [row][64][2] tile
            //    lies in ONE bank: 64-way conflicts, 1 us per parked candidate.)
            //  * Slots come in blocks of 256 that the workgroup reserves with one atomic (an atomic per tile
            //    and wave -- 87 000 on one address per launch -- costs 1-2 us under its own contention); what
            //    a tile parks beyond the end of the block goes to the front of the next one, and what is left
            //    of the last block at the end of the kernel gets status 0 (nobody).
            //  * The barriers order LDS only: __syncthreads() would also wait for the acknowledgement of this
            //    tile's global stores (4 000 cycles per tile, measured), which nobody in the workgroup reads.
            const unsigned pk_list = (unsigned)dec_off, pk_cnt = pk_list + 4u * 256u, pk_new = pk_cnt + 4u;
#ifdef PK_QR_STAMPS   // (row 30 of the stamp matrix: the phases of the parking, tools/stamps.py)
#define QR_PSTAMP(k_)                                                                    \
    do {                                                                                 \
        if ((dbg & 16) && stamps && blockIdx.x == 0 && lane == 0 && fetched)             \
            stamps[((tid >> 6) * 32 + 30) * 5 + (k_)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#define QR_PSTAMP2(k_)                                                                   \
    do {                                                                                 \
        if ((dbg & 16) && stamps && blockIdx.x == 0 && lane == 0 && fetched)             \
            stamps[((tid >> 6) * 32 + 29) * 5 + (k_)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define QR_PSTAMP(k_) do {} while (0)
#define QR_PSTAMP2(k_) do {} while (0)
#endif
            QR_PSTAMP(0);
            bool decided = false, open = false;
            unsigned my_i = 0;
            if (owner) {  // (waves 0-3, whole waves)
                open = valid && active && !((acc + A.split_rem) < A.prune_sum);
                decided = valid && !open;
                const unsigned long long m = __ballot(open);
                if (m != 0ull) {
                    unsigned pos = 0;
                    if (lane == 0) pos = __hip_atomic_fetch_add(LDS_AT(lds_i32, pk_cnt), (int)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    pos = (unsigned)__builtin_amdgcn_readfirstlane((int)pos);
                    my_i = pos + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                    if (open) *LDS_AT(lds_i32, pk_list + 4u * my_i) = tid;  // its place in the 256-candidate tile
                }
            }
            QR_PSTAMP(1);
            QR_LDS_BARRIER();  // list and count are published
            QR_PSTAMP(2);
            const unsigned P = (unsigned)__builtin_amdgcn_readfirstlane(*LDS_AT(const lds_i32, pk_cnt));
            if (P != 0u) {  // (uniform)
                const unsigned room = pk_end - pk_at;
                unsigned new_base = 0;
                if (P > room) {  // (uniform) a new block; the workgroup waits for ONE atomic, every ~25 tiles
                    if (tid == 0) *LDS_AT(lds_i32, pk_new) = (int)atomicAdd(A.s_cnt, 256u);
                    QR_LDS_BARRIER();
                    new_base = (unsigned)__builtin_amdgcn_readfirstlane(*LDS_AT(const lds_i32, pk_new));
                }
                QR_PSTAMP2(0);   // (P read, block reserved)
                if (open) {
                    const unsigned s = my_i < room ? pk_at + my_i : new_base + (my_i - room);
                    A.s_idx[s] = (int32_t)local;
                    A.s_acc[s] = acc;
                    A.s_st[s] = (uint8_t)st;
                }
                QR_PSTAMP2(1);   // (owners' stores issued)
                // A store INSTRUCTION costs the CU's address unit the same 16+ cycles whether ten of its lanes
                // are live or sixty-four (stamps, round 5: 128 of them per tile -- sixteen waves x eight rows,
                // ten parked candidates each -- took 4 000 cycles).  So the lanes are filled: the candidates of
                // a pass in the low bits of the lane number (different columns: different banks), as many ROWS
                // beside them as fit (the same column in several rows: that many lanes in one bank).
                for (unsigned i0 = 0; i0 < P; i0 += 64u) {
                    const unsigned pc = P - i0 < 64u ? P - i0 : 64u;          // candidates of this pass
                    const unsigned sh = pc <= 1u ? 0u : 32u - (unsigned)__builtin_clz(pc - 1u);  // lanes per row: 1 << sh
                    const unsigned ci = (unsigned)lane & ((1u << sh) - 1u), rsub = (unsigned)lane >> sh;
                    const int rows = 64 >> sh;                                 // rows per instruction
                    if (ci < pc) {
                        const unsigned i = i0 + ci;
                        const int cand = *LDS_AT(const lds_i32, pk_list + 4u * i);
                        const unsigned s = i < room ? pk_at + i : new_base + (i - room);
                        const unsigned src = (cand >= 128 ? (unsigned)HALF1 : 0u) + ((unsigned)(cand & 63) << 2) +
                                             ((unsigned)((cand >> 6) & 1) << 1);
                        unsigned short *dst = A.s_tiles + (size_t)(s >> 7) * (size_t)F * 128u + ((s & 63u) << 1) + ((s >> 6) & 1u);
                        for (int r0 = wave * rows; r0 < F; r0 += 16 * rows) {
                            const int r = r0 + (int)rsub;
                            if (r < F) dst[(size_t)r * 128u] = *LDS_AT(const lds_u16, src + (unsigned)r * 256u);
                        }
                    }
                }
                QR_PSTAMP2(2);   // (codes copied)
                if (P > room) {
                    pk_at = new_base + (P - room);
                    pk_end = new_base + 256u;
                } else {
                    pk_at += P;
                }
            }
            QR_PSTAMP(3);
            QR_LDS_BARRIER();  // the codes are read: the next tile may be stored over this one
            QR_PSTAMP(4);
            if (tid == 0) *LDS_AT(lds_i32, pk_cnt) = 0;  // (everybody has read it; the next tile adds to it many barriers later)
            if (decided) A.prob[c0 + local] = 0.0;
        } else if constexpr (SPLIT == 2) {
            if (valid && active) A.prob[c0 + A.s_idx[local]] = acc / (double)A.t_div;  // (status 0: a slot nobody took)
        } else {
            if (valid) A.prob[c0 + local] = active ? acc / (double)A.t_div : 0.0;
        }
        // the next tile and (PREF0) its first group: every walk of this tile is behind barrier 1 of
        // the last group and the sums are done, so its rank tiles and trees can go
        const bool grp0_here = fetched && pref0;
        tile_ready = fetched;
        if (fetched) {
            const int64_t cb = wg_next * C;
            const bool two = cb + 128 < cn;
            const unsigned tl_a0 = voff, tl_a1 = voff + (unsigned)HALF1;
            QR_TREGS(QR_ST_TILE)
            QR_TSTAMP(1);
            st_nu = A.gtab[0].w;
            if (!grp0_here) {  // without PREF0 the tile's first group is fetched now
                qr_issue<0, 8, NR>(A.img_b + (size_t)A.gtab[0].z * 16, voff);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            QR_TSTAMP(2);
            QR_GREGS(QR_ST_GROUP)
            // (the barrier at the top of the next trip publishes the stores; PRUNE: the flags of this
            // trip were last read before the last group's first barrier... its second: see below)
            if (PRUNE) __syncthreads();
        }
        QR_TSTAMP(3);
    }
    if constexpr (SPLIT == 1)  // what is left of the last block: nobody
        for (unsigned i = pk_at + threadIdx.x; i < pk_end; i += THREADS) A.s_st[i] = 0;
#undef QR_TSTAMP
#undef QR_STAMP
}

template <int HALF1, bool PRUNE, int NR, int SPLIT>
__global__ __launch_bounds__(Q_THREADS) __attribute__((amdgpu_num_vgpr(PK_QR_VGPRS))) void forest_qr_kernel(
    const v4u *__restrict__ img, const int4 *__restrict__ gtab, int n_grp, const int4 *__restrict__ ttab, int T,
    int t_div, int F, int dec_off, int val_off, int img_off, const unsigned short *__restrict__ qtiles,
    const uint8_t *__restrict__ status, int64_t c0, int64_t cn, double *__restrict__ prob, double prune_sum,
    int opt, int dbg, long long *__restrict__ stamps, qr_split_args sp)
{
    // v72..v127 belong to the inline assembly (the register count of the kernel covers them)
    asm volatile("" ::: "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83",
                 "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96",
                 "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108",
                 "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
                 "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    extern __shared__ __attribute__((aligned(16))) char lds[];  // addressed absolutely from 0
    const bool lds_at_zero = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char *)lds == 0u;
    if (!lds_at_zero && threadIdx.x == 0 && stamps) stamps[65535] = 2;
    qr_args A;
    A.img_b = reinterpret_cast<const char *>(img);
    A.gtab = gtab;
    A.ttab = ttab;
    A.n_grp = n_grp;
    A.T = T;
    A.t_div = t_div;
    A.F = F;
    A.dec_off = dec_off;
    A.val_off = val_off;
    A.img_off = img_off;
    A.qtiles = qtiles;
    A.status = status;
    A.c0 = c0;
    A.cn = cn;
    A.prob = prob;
    A.prune_sum = prune_sum;
    A.opt = lds_at_zero ? (opt & 0x7fffffff) : -1;
    A.dbg = dbg;
    A.stamps = stamps;
    A.s_cnt = sp.cnt;
    A.s_idx = sp.idx;
    A.s_acc = sp.acc;
    A.s_st = sp.st;
    A.s_tiles = sp.tiles;
    A.split_rem = sp.rem;
    // the role of this wave, once: rank tile = wave & 1, position among the waves of its SIMD = wave >> 2
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef PK_QR_FOLD_TEST   // (timing experiment, WRONG results: every wave runs the body of rank tile 0 -- four role copies)
    switch ((wave >> 2) << 1) {
#else
    switch (((wave >> 2) << 1) | (wave & 1)) {
#endif
    case 0: qr_body<HALF1, PRUNE, NR, 0, 0, SPLIT>(A); break;
    case 1: qr_body<HALF1, PRUNE, NR, HALF1, 0, SPLIT>(A); break;
    case 2: qr_body<HALF1, PRUNE, NR, 0, 1, SPLIT>(A); break;
    case 3: qr_body<HALF1, PRUNE, NR, HALF1, 1, SPLIT>(A); break;
    case 4: qr_body<HALF1, PRUNE, NR, 0, 2, SPLIT>(A); break;
    case 5: qr_body<HALF1, PRUNE, NR, HALF1, 2, SPLIT>(A); break;
    case 6: qr_body<HALF1, PRUNE, NR, 0, 3, SPLIT>(A); break;
    default: qr_body<HALF1, PRUNE, NR, HALF1, 3, SPLIT>(A); break;
    }
}

// ------------------------------------------------------------------------
// The 64-candidate shape (more than 255 features: the wide node word), TWO rank tiles per
// workgroup trip (round 3).
//
// Measured on configs[4] (529 features, 500 trees, 7 trees = 84 KB per group, 64 candidates = 68 KB
// per tile; forest_q_kernel<1,1,...>): a tree group costs 5 460 cycles of which the walk is 2 460
// (seven waves, one dependent chain each: latency-bound) -- the rest is staging: 84 KB per group
// and 64 candidates through the CU's L2 path alone take 2 420 cycles (35 B per cycle, the guide's
// L2-gather rate), the commit to LDS 1 500.  Candidates x trees resident is capped near 450 by
// the 160 KiB, so every staged byte serves a tenth of the walk work it serves at 121 features.
// The register file is the larger memory (512 KB per CU, 128 VGPRs x 1024 threads): a SECOND
// rank tile waits there (5 x 16 B per thread) while the first is walked, and is exchanged with
// it between the two walks of a group -- 68 KB out, 68 KB in, all sixteen waves -- so that every
// staged group serves 128 candidates: half the L2 traffic, half the commits, half the barriers
// per candidate for one exchange.  The tile left in LDS by a group is the one the next group
// walks first.  No next-tile prefetch (the staging registers and the spare tile leave no room;
// a trip is 72 groups long), no early termination (a run that allows pruning gets every candidate's
// full probability here: same scored pixels, and the one-tile kernel with its early exit is the slower of
// the two -- round 5), no issue priorities (latency-bound waves; rotating them was measured in round 5 on
// the fitted forest's 14-15-tree groups: +1.3 %, not kept).
// ------------------------------------------------------------------------
// SPLIT (the forest cut in two, see forest_qr_kernel / q_pick_cut): 1 = head -- after its groups every
// candidate that is still open is parked (partial sum, status, index, its column of the rank tile as a
// column of a 64-candidate tile in slot order) --, 2 = tail over the parked candidates.
template <int SPLIT>
__global__ __launch_bounds__(Q_THREADS) void forest_q2_kernel(
    const v4u *__restrict__ img, const int4 *__restrict__ gtab, int n_grp, const int4 *__restrict__ ttab, int T,
    int t_div, int F, int val_off, int img_off, const unsigned short *__restrict__ qtiles,
    const uint8_t *__restrict__ status, int64_t c0, int64_t cn_arg, double *__restrict__ prob,
    long long *__restrict__ stamps, int dbg, int late_below, qr_split_args sp, double prune_sum)
{
    constexpr int THREADS = Q_THREADS;
    const int64_t cn = SPLIT == 2 ? (int64_t)__builtin_amdgcn_readfirstlane((int)*sp.cnt) : cn_arg;
    const int64_t sc0 = SPLIT == 2 ? 0 : c0;  // (the tail's status bytes are the parked ones, in slot order)
    [[maybe_unused]] unsigned pk_at = 0, pk_end = 0;  // SPLIT 1: the next free parking slot, the end of this workgroup's block
    // six staging registers (groups of <= 96 KiB) and five for the spare tile (<= 5 x 1024 units of
    // 16 B: F <= 639) per thread.  Every thread stages its share of the next group, but the waves
    // that walk (slot < late_below: no group has more trees) issue their loads BEHIND the first
    // walk of a group, the others in front of it.  Measured (stamps, configs[4]): a wave cannot walk
    // before the memory pipeline has accepted its loads, and with all sixteen issuing at once that
    // took longer than the walk itself.  (Letting the idle waves stage everything -- twelve
    // registers each -- frees the walkers just as well, but a wave's LDS stores do not overlap:
    // ~230 cycles per ds_write_b128 and wave, so the commit of a group takes as long as the wave
    // with the most registers needs -- 2 900 cycles for twelve against 1 400 for six: 12.5 vs
    // 11.9 ms.)
    extern __shared__ __attribute__((aligned(16))) char lds[];  // addressed absolutely from 0
    const int HB = F * 128;  // bytes of a rank tile [F][64] u16
    const int upt = HB >> 4;
    const bool lds_at_zero = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char *)lds == 0u;
    if (!lds_at_zero && threadIdx.x == 0 && stamps) stamps[65535] = 2;
    const int64_t n_pair = (cn + 127) / 128;
    unsigned warm_sink = 0;
    for (int64_t pr = blockIdx.x; pr < n_pair; pr += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // (keeps per-trip address arithmetic inside the trip)
        const int lane = tid & 63;
        const int slot = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned lk0 = (unsigned)lane << 1;
        const int64_t cbase = pr * 128;  // tile A = candidates cbase .. +63, tile B = the next 64
        const bool has_b = cbase + 64 < cn;
        if (pr != (int64_t)blockIdx.x) __syncthreads();  // the previous trip's tile is no longer read
        // tile A -> LDS, tile B -> registers
        // (named registers, expanded by macros: arrays of them are moved to scratch or LDS)
#define Q2_TB5(X) X(0) X(1) X(2) X(3) X(4)
#define Q2_PF6(X) X(0) X(1) X(2) X(3) X(4) X(5)
#define Q2_TB_DECL(k) v4u tb##k = {0u, 0u, 0u, 0u};
        Q2_TB5(Q2_TB_DECL)
        {
            const v4u *src = reinterpret_cast<const v4u *>(qtiles + (size_t)cbase * F);
            // (all ten loads first, then the stores: a store behind every load is five round trips
            // to memory in a row)
#define Q2_TILES_LD(k)                                                       \
    v4u ta##k = {0u, 0u, 0u, 0u};                                            \
    if (tid + (k) * THREADS < upt) {                                         \
        ta##k = __builtin_nontemporal_load(src + tid + (k) * THREADS);       \
        if (has_b) tb##k = __builtin_nontemporal_load(src + upt + tid + (k) * THREADS); \
    }
#define Q2_TILES_ST(k) \
    if (tid + (k) * THREADS < upt) *LDS_AT(lds_u4, (tid + (k) * THREADS) << 4) = ta##k;
            Q2_TB5(Q2_TILES_LD)
            Q2_TB5(Q2_TILES_ST)
#undef Q2_TILES_LD
#undef Q2_TILES_ST
        }
        // lane's candidate in either tile; thread tid < 128 owns candidate cbase + tid (ordered sum)
        const int64_t la = cbase + lane, lb = cbase + 64 + lane;
        const unsigned st_a = la < cn ? status[sc0 + la] : 0, st_b = lb < cn ? status[sc0 + lb] : 0;
        const bool act_a = lds_at_zero && st_a != 0;
        const bool act_b = lds_at_zero && st_b != 0;
        // (uniform) a wave with a NaN feature among its candidates walks with the NaN rule
        const bool nan_a = __any(st_a == 2), nan_b = __any(st_b == 2);
        const bool owner = tid < 128;
        const bool valid = owner && cbase + tid < cn;
        const bool active = valid && lds_at_zero && status[sc0 + cbase + tid] != 0;
#define Q2_PF_DECL(q) v4u pf##q = {0u, 0u, 0u, 0u}, pl##q = {0u, 0u, 0u, 0u};
        Q2_PF6(Q2_PF_DECL)
        // Thread tid moves units tid, tid + 1024, ... of the group.  The loads in front of the walk
        // and the late ones use DIFFERENT registers (pf / pl; a wave uses one set), and an explicit
        // vmcnt(0) stands in front of the commit: with the same registers loaded at two places, or a
        // wait only inside the conditional stores, the compiler cannot rule out a load still pending
        // on a register it is about to overwrite (address arithmetic lands in the destination) and
        // waits for vmcnt(0) in front of EVERY load -- six round trips to the L2 in a row where six
        // loads in flight take one.
#define Q2_PF_LOAD(q) if ((q) * THREADS < pf_nu) pf##q = pf_src[min(tid + (q) * THREADS, pf_nu - 1)];
#define Q2_PL_LOAD(q) if ((q) * THREADS < pf_nu) pl##q = pf_src[min(tid + (q) * THREADS, pf_nu - 1)];
#define Q2_PF_STORE(q)                                                        \
    {                                                                         \
        const int u = tid + (q) * THREADS;                                    \
        if (u < pf_nu) *LDS_AT(lds_u4, img_off + (u << 4)) = pf##q;           \
    }
#define Q2_PL_STORE(q)                                                        \
    {                                                                         \
        const int u = tid + (q) * THREADS;                                    \
        if (u < pf_nu) *LDS_AT(lds_u4, img_off + (u << 4)) = pl##q;           \
    }
#define Q2_VMCNT0() __builtin_amdgcn_s_waitcnt(0x0F70)  // vmcnt(0); expcnt, lgkmcnt untouched
        int4 g_cur = gtab[0];
        int4 tt = ttab[min(g_cur.x + slot, T - 1)];
        const v4u *pf_src = img + g_cur.z;
        int pf_nu = g_cur.w;
        {   // first group: global -> VGPR -> LDS
            Q2_PF6(Q2_PF_LOAD)
            Q2_VMCNT0();
            Q2_PF6(Q2_PF_STORE)
        }
        const bool pf_late = has_b && slot < late_below;  // (uniform) this wave loads behind the first walk
        __syncthreads();  // tile A and the first group are in LDS
        double acc = 0.0;
        if constexpr (SPLIT == 2)
            if (valid) acc = sp.acc[cbase + tid];  // the head's trees, already added in order
        int cur = 0;  // (uniform) the tile that sits in LDS: 0 = A, 1 = B
#ifdef PK_QR_STAMPS
#define Q2_STAMP(k_)                                                                    \
    do {                                                                                \
        if ((dbg & 16) && stamps && blockIdx.x == 0 && lane == 0 && g < 32)             \
            stamps[((tid >> 6) * 32 + g) * 8 + (k_)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define Q2_STAMP(k_) do {} while (0)   // (every test of a debug flag is an instruction in the wave's stream)
#endif
        for (int g = 0; g < n_grp; g++) {
            const int gt = g_cur.y;
            Q2_STAMP(0);
            const int4 g_nxt = gtab[g + 1];
            const int4 tt_nxt = ttab[min(g_nxt.x + slot, T - 1)];
            unsigned warm_v = 0;
            if (slot == 15 && !(dbg & 128)) {
                // the group after the next one -> this XCD's L2 (the image is larger than it and is
                // walked in the same order by every workgroup): workgroup i runs on XCD i % 8; the 32
                // of an XCD share the lines of the group (<= 1 024: at most 32 each), one dword of
                // each 128-byte line is asked for and only looked at behind the commit's wait
                const int4 gw = gtab[g + 2 < n_grp ? g + 2 : g + 2 - n_grp < n_grp ? g + 2 - n_grp : 0];
                const int lines = (gw.w + 7) >> 3, per = (lines + 31) >> 5;
                const int i = (int)((blockIdx.x >> 3) & 31) * per + lane;
                if (lane < per && i < lines) warm_v = reinterpret_cast<const unsigned *>(img + gw.z)[i * 32];
            }
            if (g + 1 < n_grp) {  // the next group flies while this one is walked twice
                pf_src = img + g_nxt.z;
                pf_nu = g_nxt.w;
                if (!pf_late) { Q2_PF6(Q2_PF_LOAD) }
            }
            const unsigned tbase = (unsigned)(img_off + tt.x);
#pragma unroll
            for (int half = 0; half < 2; half++) {
                // walk the tile in LDS; thread tid owns candidate (tile `cur`, lane tid & 63)
                if (slot < gt) {  // (lanes without a live candidate walk along: their values are not stored)
                    // (tt.y: levels to walk | the tree's split << 16)
                    const double v0 = __builtin_expect(cur ? nan_b : nan_a, 0)
                                          ? q_walk_one<true>((unsigned)tt.z, tt.y & 0xFFFF, tbase, lk0, (unsigned)tt.y >> 16)
                                          : q_walk_one<false>((unsigned)tt.z, tt.y & 0xFFFF, tbase, lk0, 0u);
                    if (cur ? act_b : act_a) *LDS_AT(lds_f64, val_off + (slot * 64 + lane) * 8) = v0;
                }
                Q2_STAMP(half ? 5 : 1);
                __syncthreads();  // every walk of this tile is done, every value parked
                Q2_STAMP(half ? 6 : 2);
                if (half == 0 && g + 1 < n_grp && pf_late) {  // the walkers' loads: in flight during the
                    Q2_PF6(Q2_PL_LOAD)                          // exchange and the second walk
                }
                if (active && (tid >> 6) == cur) {  // tree order: sklearn's sequential float64 sum
                    // Eight parked values are read at once, then added in order (a slot without a tree adds
                    // +0.0: the sum is never negative, so x + 0.0 == x bit for bit).  Round 5: as a loop over
                    // the group's trees every add waited for its own LDS read -- 14 round trips, ~900 cycles
                    // per tile and group during which fifteen waves stood at the barrier
                    // (profiles/r05_stamps_w11_fitted.log); now two.
#pragma unroll
                    for (int j0 = 0; j0 < 16; j0 += 8) {
                        if (j0 < gt) {
                            double pv[8];
#pragma unroll
                            for (int j = 0; j < 8; j++) pv[j] = *LDS_AT(lds_f64, val_off + ((j0 + j) * 64 + (tid & 63)) * 8);
#pragma unroll
                            for (int j = 0; j < 8; j++) acc += (j0 + j < gt) ? pv[j] : 0.0;
                        }
                    }
                }
                if (half == 0) {
                    if (has_b) {  // exchange the tiles: this thread's units of the one in LDS against its spare
#define Q2_SWAP_RD(k) v4u t##k = tb##k; if (tid + (k) * THREADS < upt) t##k = *LDS_AT(lds_u4, (tid + (k) * THREADS) << 4);
#define Q2_SWAP_WR(k)                                                           \
    if (tid + (k) * THREADS < upt) {                                            \
        *LDS_AT(lds_u4, (tid + (k) * THREADS) << 4) = tb##k;                    \
        tb##k = t##k;                                                           \
    }
                        // (three units, then two: five exchange registers at once spill)
                        Q2_SWAP_RD(0) Q2_SWAP_RD(1) Q2_SWAP_RD(2)
                        Q2_SWAP_WR(0) Q2_SWAP_WR(1) Q2_SWAP_WR(2)
                        Q2_SWAP_RD(3) Q2_SWAP_RD(4)
                        Q2_SWAP_WR(3) Q2_SWAP_WR(4)
#undef Q2_SWAP_RD
#undef Q2_SWAP_WR
                        cur ^= 1;
                        Q2_STAMP(3);
                        __syncthreads();  // the other tile is in; the parked values are consumed
                        Q2_STAMP(4);
                    } else {
                        break;  // (uniform) a trip with one tile: one walk per group
                    }
                }
            }
            Q2_VMCNT0();
            Q2_STAMP(7);
            warm_sink += warm_v;
            if (g + 1 < n_grp) {  // commit the next group (every walk of this one is behind a barrier)
                if (pf_late) { Q2_PF6(Q2_PL_STORE) }
                else { Q2_PF6(Q2_PF_STORE) }
            }
            __syncthreads();  // next group staged; values consumed
            g_cur = g_nxt;
            tt = tt_nxt;
        }
        if constexpr (SPLIT == 1) {
            // The cut (see forest_qr_kernel): the open candidates of this trip's two tiles are listed in LDS
            // (the parked values' place: every sum is behind the last barrier), the spare tile joins the other
            // one in LDS (over the tree images: every walk is done), then all sixteen waves copy the columns --
            // lanes = parked candidates, a wave takes every sixteenth row.  Slots: blocks of 128 (a trip's two tiles) per workgroup.
            const unsigned pk_list = (unsigned)val_off, pk_cnt = pk_list + 4u * 128u, pk_new = pk_cnt + 4u;
            if (tid == 0) *LDS_AT(lds_i32, pk_cnt) = 0;
            if (has_b) {
#define Q2_SPARE_ST(k) if (tid + (k) * THREADS < upt) *LDS_AT(lds_u4, img_off + ((tid + (k) * THREADS) << 4)) = tb##k;
                Q2_TB5(Q2_SPARE_ST)
#undef Q2_SPARE_ST
            }
            QR_LDS_BARRIER();
            bool open = false;
            unsigned my_i = 0;
            if (owner) {  // (waves 0 and 1: tile A, tile B)
                open = active && !((acc + sp.rem) < prune_sum);
                const unsigned long long m = __ballot(open);
                if (m != 0ull) {
                    unsigned pos = 0;
                    if (lane == 0) pos = __hip_atomic_fetch_add(LDS_AT(lds_i32, pk_cnt), (int)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    pos = (unsigned)__builtin_amdgcn_readfirstlane((int)pos);
                    my_i = pos + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                    if (open) *LDS_AT(lds_i32, pk_list + 4u * my_i) = tid;  // tile << 6 | lane
                }
            }
            QR_LDS_BARRIER();  // list and count are published
            const unsigned P = (unsigned)__builtin_amdgcn_readfirstlane(*LDS_AT(const lds_i32, pk_cnt));
            if (P != 0u) {  // (uniform)
                const unsigned room = pk_end - pk_at;
                unsigned new_base = 0;
                if (P > room) {  // (uniform) a new block of slots: one atomic per ~128 parked candidates
                    if (tid == 0) *LDS_AT(lds_i32, pk_new) = (int)atomicAdd(sp.cnt, 128u);
                    QR_LDS_BARRIER();
                    new_base = (unsigned)__builtin_amdgcn_readfirstlane(*LDS_AT(const lds_i32, pk_new));
                }
                if (open) {
                    const unsigned s_ = my_i < room ? pk_at + my_i : new_base + (my_i - room);
                    sp.idx[s_] = (int32_t)(cbase + tid);
                    sp.acc[s_] = acc;
                    sp.st[s_] = (uint8_t)(tid < 64 ? st_a : st_b);
                }
                // (the lanes of a store instruction are filled -- candidates in the low bits of the lane number,
                // rows beside them -- as in forest_qr_kernel: an instruction costs the address unit the same
                // whether two of its lanes are live or sixty-four)
                for (unsigned i0 = 0; i0 < P; i0 += 64u) {
                    const unsigned pc = P - i0 < 64u ? P - i0 : 64u;
                    const unsigned sh = pc <= 1u ? 0u : 32u - (unsigned)__builtin_clz(pc - 1u);
                    const unsigned ci = (unsigned)lane & ((1u << sh) - 1u), rsub = (unsigned)lane >> sh;
                    const int rows = 64 >> sh;
                    if (ci < pc) {
                        const unsigned i = i0 + ci;
                        const int cand = *LDS_AT(const lds_i32, pk_list + 4u * i);
                        const unsigned s_ = i < room ? pk_at + i : new_base + (i - room);
                        // (the tile that sat in LDS at the end is at 0, the other one over the tree images)
                        const unsigned src = ((cand >> 6) == cur ? 0u : (unsigned)img_off) + ((unsigned)(cand & 63) << 1);
                        unsigned short *dst = sp.tiles + (size_t)(s_ >> 6) * (size_t)F * 64u + (s_ & 63u);
                        for (int r0 = slot * rows; r0 < F; r0 += 16 * rows) {
                            const int r = r0 + (int)rsub;
                            if (r < F) dst[(size_t)r * 64u] = *LDS_AT(const lds_u16, src + (unsigned)r * 128u);
                        }
                    }
                }
                if (P > room) {
                    pk_at = new_base + (P - room);
                    pk_end = new_base + 128u;
                } else {
                    pk_at += P;
                }
            }
            if (valid && !open) prob[c0 + cbase + tid] = 0.0;
            // (the barrier at the top of the next trip stands between these reads and the next tile's stores)
        } else if constexpr (SPLIT == 2) {
            if (valid && active) prob[c0 + sp.idx[cbase + tid]] = acc / (double)t_div;  // (status 0: a slot nobody took)
        } else {
            if (valid) prob[c0 + cbase + tid] = active ? acc / (double)t_div : 0.0;
        }
    }
    if constexpr (SPLIT == 1)  // what is left of the last block: nobody
        for (unsigned i = pk_at + threadIdx.x; i < pk_end; i += THREADS) sp.st[i] = 0;
    if (warm_sink == 0x9e3779b9u && stamps) stamps[65534] = 1;  // keeps the warm-up loads alive
#undef Q2_STAMP
#undef Q2_TB5
#undef Q2_PF6
#undef Q2_VMCNT0
#undef Q2_PL_LOAD
#undef Q2_PL_STORE
#undef Q2_TB_DECL
#undef Q2_PF_DECL
#undef Q2_PF_LOAD
#undef Q2_PF_STORE
}

template <typename KernelT>
int q_set_max_lds(KernelT k, size_t bytes)
{
    PK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PK_OK;
}

void q_free(pk_forest *f)
{
    void *ptrs[] = {f->q_img, f->q_gtab, f->q_ttab, f->q_off, f->q_thr, f->q_par, f->q_lut, f->q_src};
    for (void *p : ptrs)
        if (p) hipFree(p);
    f->q_img = nullptr;
    f->q_gtab = f->q_ttab = f->q_off = f->q_src = nullptr;
    f->q_thr = f->q_par = nullptr;
    f->q_lut = nullptr;
    delete f->q_layout;
    f->q_layout = nullptr;
    f->q_gtab_h.clear();
}

template <typename V>
int q_upload(void **dst, const V &v)
{
    PK_HIP(hipMalloc(dst, v.size() * sizeof(v[0])));
    PK_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice));
    return PK_OK;
}

}  // namespace

void pk_forest_q_release(pk_forest *f)
{
    q_free(f);
    f->q_state = 0;
}

// Picks walks per lane and the slot count, builds and uploads the rank image.
static int q_plan_build(pk_forest *f)
{
    const int F = f->F, T = f->T;
    if (F > 1023 || f->h_tree_off.empty()) return PK_E_UNSUPPORTED;
    // the rank tables first: they say how many ROWS the rank tile has (Fq = the features + the virtual
    // features of those with more than 2 047 thresholds, pk_qimage.hip), and the shape follows from that
    pk_q_out best;
    int rc = pk_q_tables(T, F, f->h_tree_off.data(), f->h_left.data(), f->h_feat.data(), f->h_thr.data(), PK_Q_MAX_RANK, &best);
    if (rc) return rc;
    int Fq = best.Fq;
    int mode = PK_Q_NARROW;
    // 256 candidates per workgroup while two rank tiles of 256 B per row fit 64 KiB of
    // offsets, 128 up to 255 rows (the narrow word's feature byte), 64 candidates and the
    // wide word beyond (w = 11: 529 features)
    auto shape_of = [](int rows) { return rows <= 192 ? 4 : rows <= 255 ? 2 : 1; };
    int ch = (int)f->opt.forest_q_ch;
    if (ch == 0) ch = shape_of(Fq);
    // Features with more than 2 047 thresholds cost rows (one per further 2 047).  When the rows push the
    // forest out of a shape that 12-bit ranks (4 095 per row) would keep -- a 100-tree forest fitted on
    // 90 000 windows: 2 154 thresholds per feature, 220 rows against 121 -- the 12-bit narrow word is used
    // (option forest_q_rank12: 0 never, 1 when it keeps a larger shape (default), 2 whenever rows are saved).
    if (f->opt.forest_q_rank12 != 0 && F <= 255 && Fq > F) {
        pk_q_out alt;
        if (pk_q_tables(T, F, f->h_tree_off.data(), f->h_left.data(), f->h_feat.data(), f->h_thr.data(), PK_Q_MAX_RANK12, &alt) == PK_OK &&
            alt.Fq < Fq && alt.Fq <= 255) {
            const int ch12 = f->opt.forest_q_ch ? (int)f->opt.forest_q_ch : shape_of(alt.Fq);
            const bool fits12 = !((ch12 == 4 && alt.Fq > 192) || (ch12 == 2 && alt.Fq > 255)) && ch12 != 1;
            if (fits12 && (f->opt.forest_q_rank12 == 2 || ch12 > shape_of(Fq) || Fq > 255) &&
                pk_q_trees(T, F, f->h_tree_off.data(), f->h_left.data(), f->h_right.data(), f->h_feat.data(), f->h_thr.data(),
                           f->h_miss.empty() ? nullptr : f->h_miss.data(), f->h_p1.data(), PK_Q_NARROW12, &alt) == PK_OK) {
                best = alt;
                Fq = best.Fq;
                ch = ch12;
                mode = PK_Q_NARROW12;
            }
        }
    }
    if ((ch == 4 && Fq > 192) || (ch == 2 && Fq > 255)) return PK_E_UNSUPPORTED;
    if (mode != PK_Q_NARROW12) {
        mode = ch == 1 ? PK_Q_WIDE : PK_Q_NARROW;
        rc = pk_q_trees(T, F, f->h_tree_off.data(), f->h_left.data(), f->h_right.data(), f->h_feat.data(),
                        f->h_thr.data(), f->h_miss.empty() ? nullptr : f->h_miss.data(), f->h_p1.data(), mode, &best);
        if (rc) return rc;  // the forest does not fit the format at all, or is malformed
    }
    // tables and trees exist once (they do not depend on the layout); only the grouping is tried
    // for every slot count
    pk_q_layout bestL;
    int best_slots = 0;
    double best_score = 0.0;
    const int forced = (int)f->opt.forest_slots;
    // (4 walks per lane: at most 8 trees per group, so that two waves can share a tree and
    // all 16 walk; measured faster than 9 trees on 9 of 16 waves)
    const int max_slots = (ch == 4 && f->opt.forest_q_wpt != 1) ? 8 : 16;
    std::vector<int32_t> best_gtab, best_ttab;
    int best_n_grp = 0;
    for (int slots = 2; slots <= max_slots; slots++) {  // slots = trees per group at most
        if (forced && slots != forced) continue;
        pk_q_layout L;
        if (!pk_q_make_layout(Fq, slots, ch, &L)) continue;
        rc = pk_q_group(&best, L);
        if (rc == PK_E_UNSUPPORTED) continue;
        if (rc) return rc;
        // a group costs about the same whatever it holds: trees per group is the figure of
        // merit; a slot that adds less than a quarter tree per group only adds an idle wave
        const double score = (double)T / (double)best.n_grp;
        if (score > best_score + 0.24) {
            best_score = score;
            best_gtab = best.gtab;
            best_ttab = best.ttab;
            best_n_grp = best.n_grp;
            bestL = L;
            best_slots = slots;
        }
    }
    if (best_slots) {
        best.gtab = best_gtab;
        best.ttab = best_ttab;
        best.n_grp = best_n_grp;
    }
    // early staging (two waves per tree): fixed tree slots, slot s as large as the largest
    // tree in position s of any group; each wave stages half a tree from 6 registers
    // (<= 6 KiB per half).  Only taken when it does not cost a tree per group.
    int slot_bytes = 0;
    if (best_slots && ch == 4 && best_slots <= 8 && f->opt.forest_q_wpt != 1 && f->opt.forest_q_early &&
        pk_q_max_tree_bytes(best) <= 2 * 6 * 64 * 16) {
        pk_q_layout L2;
        if (pk_q_make_layout(Fq, best_slots, ch, &L2)) {
            pk_q_fixed_slots(best, best_slots, &L2);
            if (L2.slot_bytes <= L2.cap && pk_q_group(&best, L2) == PK_OK) {
                bestL = L2;
                slot_bytes = L2.slot_bytes;
            } else {  // restore the packed grouping
                best.gtab = best_gtab;
                best.ttab = best_ttab;
                best.n_grp = best_n_grp;
            }
        }
    }
    if (!best_slots) return PK_E_UNSUPPORTED;
    q_free(f);
    f->q_layout = new pk_q_layout(bestL);
    f->q_slots = best_slots;
    f->q_slot_bytes = slot_bytes;
    f->q_ch = ch;
    f->q_mode = mode;
    f->q_T = (int)best.troot.size();  // (the model's trees, or more: trees cut into pieces)
    f->q_F = Fq;
    f->q_n_grp = best.n_grp;
    f->q_max_group_bytes = 0;
    for (int g = 0; g < best.n_grp; g++) f->q_max_group_bytes = std::max(f->q_max_group_bytes, best.gtab[4 * g + 3] * 16);
    // (+ pad: forest_qr_kernel's staging loads are whole 16-KiB rows, unclamped)
    best.pairs.resize(best.pairs.size() + PK_Q_PAD_BYTES / sizeof(best.pairs[0]), make_uint2(0, 0));
    rc = q_upload((void **)&f->q_img, best.pairs);
    if (!rc) rc = q_upload((void **)&f->q_src, best.qsrc);
    if (!rc) rc = q_upload((void **)&f->q_gtab, best.gtab);
    f->q_gtab_h = best.gtab;
    if (!rc) rc = q_upload((void **)&f->q_ttab, best.ttab);
    if (!rc) rc = q_upload((void **)&f->q_off, best.qoff);
    if (!rc) rc = q_upload((void **)&f->q_thr, best.qthr);
    if (!rc) rc = q_upload((void **)&f->q_par, best.qpar);
    if (rc) return rc;
    // the lookup table must agree with the cells the DEVICE computes: take the thresholds'
    // cells from the device (they equal the host's unless the two float units disagree)
    {
        int32_t *d_cells = nullptr;
        std::vector<int32_t> cells(best.qthr.size(), 0);
        PK_HIP(hipMalloc((void **)&d_cells, cells.size() * sizeof(int32_t)));
        PK_HIP(hipMemset(d_cells, 0, cells.size() * sizeof(int32_t)));
        hipLaunchKernelGGL(q_cells_kernel, dim3((unsigned)Fq), dim3(256), 0, 0, f->q_thr, f->q_off, f->q_par,
                           Fq, d_cells);
        hipError_t e = hipMemcpy(cells.data(), d_cells, cells.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
        hipFree(d_cells);
        if (e != hipSuccess) {
            pk_set_error("forest rank image: reading the lookup cells back failed: %s", hipGetErrorString(e));
            return PK_E_HIP;
        }
        pk_q_fill_lut(&best, Fq, cells);
    }
    return q_upload((void **)&f->q_lut, best.qlut);
}

int pk_forest_q_plan(pk_forest *f)
{
    if (f->q_state != 0 && (f->q_opt_slots != f->opt.forest_slots || f->q_opt_ch != f->opt.forest_q_ch ||
                            f->q_opt_wpt != f->opt.forest_q_wpt || f->q_opt_early != f->opt.forest_q_early ||
                            f->q_opt_rank12 != f->opt.forest_q_rank12)) {
        q_free(f);
        f->q_state = 0;
    }
    if (f->q_state == 0) {
        f->q_opt_slots = f->opt.forest_slots;
        f->q_opt_ch = f->opt.forest_q_ch;
        f->q_opt_wpt = f->opt.forest_q_wpt;
        f->q_opt_early = f->opt.forest_q_early;
        f->q_opt_rank12 = f->opt.forest_q_rank12;
        const int rc = q_plan_build(f);
        f->q_state = rc == PK_OK ? 1 : -1;
        if (rc != PK_OK) q_free(f);
        if (rc != PK_OK && rc != PK_E_UNSUPPORTED) return rc;
    }
    return f->q_state == 1 ? PK_OK : PK_E_UNSUPPORTED;
}

#define Q_LAUNCH_PS(CH, WPT, HALF1, PRUNE, EARLY, SPLIT_, GTAB_, NGRP_, TILES_, STATUS_, CN_, PSUM_)  \
    do {                                                                                       \
        int rc__ = q_set_max_lds(forest_q_kernel<CH, WPT, HALF1, PRUNE, EARLY, SPLIT_>, 163840); \
        if (rc__) return rc__;                                                                 \
        hipLaunchKernelGGL((forest_q_kernel<CH, WPT, HALF1, PRUNE, EARLY, SPLIT_>), dim3(grid), dim3(Q_THREADS), \
                           163840, ctx->stream, reinterpret_cast<const v4u *>(f->q_img),       \
                           GTAB_, NGRP_,                                                       \
                           reinterpret_cast<const int4 *>(f->q_ttab), f->q_T, f->T, f->q_F, L.dec_off,   \
                           L.val_off, L.img_off, slots_at, TILES_, STATUS_, c0,                \
                           CN_, d_prob,                                                        \
                           PSUM_,                                                              \
                           persist ? (int)grid : f->opt.forest_warm == 1 ? ctx->cu_count : (int)f->opt.forest_warm, \
                           (int)f->opt.forest_dbg | (f->opt.forest_q_prio ? 0 : 32), ctx->dbg_buf, gsp);  \
    } while (0)
#define Q_LAUNCH_P(CH, WPT, HALF1, PRUNE, EARLY)                                               \
    Q_LAUNCH_PS(CH, WPT, HALF1, PRUNE, EARLY, 0, reinterpret_cast<const int4 *>(f->q_gtab), f->q_n_grp, ctx->q_tiles, \
                d_status, cn, prune_sum)
#define Q_LAUNCH(CH, WPT, HALF1, EARLY)                                                        \
    do {                                                                                       \
        if (prune_sum > -1e300) Q_LAUNCH_P(CH, WPT, HALF1, true, EARLY);                       \
        else Q_LAUNCH_P(CH, WPT, HALF1, false, EARLY);                                         \
    } while (0)
// the generic shapes cut in two (never with early staging): head, then the tail over the parked candidates
#define Q_LAUNCH_CUT(CH, WPT, HALF1)                                                           \
    do {                                                                                       \
        Q_LAUNCH_PS(CH, WPT, HALF1, false, false, 1, reinterpret_cast<const int4 *>(f->q_gtab), gcut, ctx->q_tiles, \
                    d_status, cn, split_sum);                                                  \
        PK_HIP(hipGetLastError());                                                             \
        pk_prof_scope prof_tail(ctx, PK_K_FOREST_TAIL);                                        \
        Q_LAUNCH_PS(CH, WPT, HALF1, false, false, 2, reinterpret_cast<const int4 *>(f->q_gtab) + gcut, f->q_n_grp - gcut, \
                    gsp.tiles, gsp.st, (int64_t)0, split_sum);                                 \
    } while (0)

// Room for the rank tiles of `cn` candidates (scratch of the context; grows only).
int pk_forest_q_reserve(pk_device_ctx *ctx, pk_forest *f, int64_t cn)
{
    if (f->q_state != 1 || !f->q_layout) {
        pk_set_error("forest rank kernel launched without a rank image (internal error)");
        return PK_E_INVALID;
    }
    const int64_t n_tiles = (cn + 127) / 128;
    // (+ pad: forest_qr_kernel's unclamped 16-KiB register rows read past the last tile)
    const size_t qbytes = (size_t)n_tiles * f->q_F * 128 * sizeof(unsigned short) + PK_Q_PAD_BYTES;
    if (qbytes > ctx->q_tiles_bytes) {
        if (ctx->q_tiles) {
            PK_HIP(hipStreamSynchronize(ctx->stream));
            PK_HIP(hipFree(ctx->q_tiles));
            ctx->q_tiles = nullptr;
            ctx->q_tiles_bytes = 0;
        }
        PK_HIP(hipMalloc((void **)&ctx->q_tiles, qbytes));
        ctx->q_tiles_bytes = qbytes;
    }
    return PK_OK;
}

// float tiles of candidates [t0 * 128, t0 * 128 + cn) of a chunk -> their rank tiles.  `tiles`
// holds the floats of THOSE candidates from its start (option sub_chunk: the extractor of a
// piece writes from the start of the buffer every time), the rank tiles go to tile t0 onwards.
int pk_launch_quant_q(pk_device_ctx *ctx, hipStream_t st, pk_forest *f, const float *tiles, int64_t t0,
                      int64_t cn)
{
    if (cn <= 0) return PK_OK;
    const pk_q_layout &L = *f->q_layout;
    const int F = f->q_F;  // rows of a rank tile
    const int64_t n_tiles = (cn + 127) / 128;
    pk_prof_scope prof(ctx, PK_K_QUANT, st);
    // enough blocks per feature to fill the chip, few enough that the tables are
    // loaded for many tiles each
    int64_t split = (n_tiles + 255) / 256;
    if (split > 64) split = 64;
    if (split < 1) split = 1;
    // four tiles per wave and trip, 1024-thread blocks for the 128-candidate tiles (64 tiles = 32 KB of
    // one feature's floats per block and trip), 512 for the wide forests' single tiles -- round 4's
    // sweep (profiles/r04_ab_quant_shape.log): w = 5 0.96 -> 0.80 ms per step, w = 6 1.74 -> 1.52,
    // w = 11 0.90 -> 0.82; eight tiles per trip and 256 / 384 / 768 threads measured beside them
    const unsigned bt = L.ch == 1 ? 512u : 1024u;
    hipLaunchKernelGGL(quantize_tiles_kernel<4>, dim3((unsigned)F, (unsigned)split), dim3(bt), 0, st, tiles,
                       n_tiles, f->F, F, f->q_src, f->q_thr, f->q_off, f->q_lut, f->q_par,
                       ctx->q_tiles + (size_t)t0 * F * 128, L.ch == 1 ? 1 : 0, f->q_mode == PK_Q_NARROW12 ? 4 : 5);
    PK_HIP(hipGetLastError());
    return PK_OK;
}

int pk_launch_forest_q(pk_device_ctx *ctx, pk_forest *f, const float *tiles, const uint8_t *d_status,
                       int64_t c0, int64_t cn, double *d_prob, double prune_sum, double split_sum)
{
    if (cn <= 0) return PK_OK;
    int rc = pk_forest_q_reserve(ctx, f, cn);
    if (!rc) rc = pk_launch_quant_q(ctx, ctx->stream, f, tiles, 0, cn);
    // (behind the quantizer the float tiles of these candidates are dead: the cut forest's scratch)
    const int64_t blk = f->q_ch == 1 ? 128 : 128 * PK_Q_FTILE;
    if (!rc)
        rc = pk_launch_forest_q_walk(ctx, f, d_status, c0, cn, d_prob, prune_sum, split_sum, const_cast<float *>(tiles),
                                     (size_t)((cn + blk - 1) / blk * blk) * (size_t)f->F * sizeof(float));
    return rc;
}

// the walk over the rank tiles of candidates [c0, c0 + cn) (tile 0 = candidate c0)
// The forest cut in two (round 5).  Measured potential on config 2 (tests/fuzz/cut_potential.py: per-tree
// leaf values of 153 546 candidates, computed on the host): at the default threshold 0.5 a candidate's fate is sealed after
// ~60 of the 100 trees on average (its partial sum plus 1.0 per remaining tree can no longer exceed
// thre * T), but a 256-candidate TILE -- the unit the in-kernel early exit works on -- only after ~80,
// and leaving a tile early costs an exposed tile load: the in-kernel exit gains nothing below 0.55.  A
// cut at a group boundary with a compaction behind it works per CANDIDATE: the head walks the first
// `cut` groups over everybody, every candidate still open is parked (3.9 % of config 2's at 64 trees),
// the tail walks the remaining groups over the parked ones only and continues their sums in tree order.
// Returns the group to cut in front of (0: no cut): the first boundary behind which a candidate whose
// partial sum is at most `forest_split_frac` per mille of the trees walked is decided.
// What the runs themselves teach (pk_forest_cut_feedback, called with the number of candidates a call's cut
// launches parked): a cut that leaves more than 15 % open moves a group later for the next call (two when
// more than half stayed open), one that has nothing worth-while left behind it is given up for this
// threshold -- a forest of untrained random trees (p ~ 0.5 everywhere) parks everybody wherever it is cut.
static int q_pick_cut(pk_forest *f, double split_sum)
{
    if ((int)f->q_gtab_h.size() < 4 * (f->q_n_grp + 1)) return 0;
    if (f->opt.forest_split_at > 0) return f->opt.forest_split_at < f->q_n_grp ? (int)f->opt.forest_split_at : 0;
    if (f->cut_sum != split_sum) {  // another threshold: what was learnt does not apply
        f->cut_sum = split_sum;
        f->cut_shift = 0;
        f->cut_off = false;
    }
    if (f->cut_off) return 0;
    const double frac = (double)f->opt.forest_split_frac * 1e-3;
    for (int g = 1; g < f->q_n_grp; g++) {
        const double done = (double)f->q_gtab_h[4 * (size_t)g];  // trees in front of group g
        const double lim = split_sum - ((double)f->q_T - done);  // a sum up to here is decided
        if (lim >= frac * done && lim > 0.0) {
            g += f->cut_shift;
            // (a tail of less than a seventh of the forest does not pay for the head's parking and a launch)
            if (g * 7 > f->q_n_grp * 6) {
                f->cut_off = true;
                return 0;
            }
            return g;
        }
    }
    return 0;
}

void pk_forest_cut_feedback(pk_forest *f, int64_t candidates, int64_t parked, int64_t slack)
{
    if (!f || candidates <= 0 || f->last_cut <= 0 || f->opt.forest_split_at > 0) return;
    // (`parked` counts SLOTS: a workgroup reserves a block -- 256, the two-tile kernel 128 -- and leaves on
    // average half of its last one empty; `slack` = what the launches of this call booked for that: grid x
    // block / 2 each, with the grid and the block size each launch really had)
    parked -= slack;
    const double open = parked > 0 ? (double)parked / (double)candidates : 0.0;
    if (open > 0.5) f->cut_shift += 2;
    else if (open > 0.15) f->cut_shift += 1;
}

// (include/peakachu_hip.h: the cut's policy without a device, for the CPU tests)
extern "C" int pk_debug_cut_policy(const int32_t *trees_in_front, int n_groups, double split_sum, int frac_permille,
                                   const double *open_frac, int n_calls, int32_t *cuts)
{
    if (!trees_in_front || n_groups < 1 || !cuts || n_calls < 0 || (n_calls > 0 && !open_frac)) {
        pk_set_error("pk_debug_cut_policy: bad arguments");
        return PK_E_INVALID;
    }
    pk_forest f{};
    f.opt = pk_default_options();
    f.opt.forest_split_at = 0;
    f.opt.forest_split_frac = frac_permille;
    f.q_n_grp = n_groups;
    f.q_T = trees_in_front[n_groups];
    f.q_gtab_h.assign((size_t)4 * (n_groups + 1), 0);
    for (int g = 0; g <= n_groups; g++) f.q_gtab_h[(size_t)4 * g] = trees_in_front[g];
    const int64_t candidates = 1000000;
    for (int i = 0; i < n_calls; i++) {
        const int cut = q_pick_cut(&f, split_sum);
        cuts[i] = cut;
        f.last_cut = cut;
        // (one launch; the slots of the unfinished blocks the feedback takes off are put on top)
        pk_forest_cut_feedback(&f, candidates, (int64_t)(open_frac[i] * (double)candidates) + 32768, 32768);
    }
    return PK_OK;
}

int pk_launch_forest_q_walk(pk_device_ctx *ctx, pk_forest *f, const uint8_t *d_status, int64_t c0, int64_t cn,
                            double *d_prob, double prune_sum, double split_sum, void *scratch, size_t scratch_bytes)
{
    if (cn <= 0) return PK_OK;
    if (f->q_state != 1 || !f->q_layout) {
        pk_set_error("forest rank kernel launched without a rank image (internal error)");
        return PK_E_INVALID;
    }
    const pk_q_layout &L = *f->q_layout;
    pk_prof_scope prof(ctx, PK_K_FOREST);
    const int C = 64 * L.ch;
    unsigned grid = (unsigned)((cn + C - 1) / C);
    // persistent launch: forest_q_persist workgroups per CU, each looping over tiles (a negative
    // value = exactly that many workgroups in all: tests)
    const unsigned want = f->opt.forest_q_persist > 0 ? (unsigned)(ctx->cu_count * f->opt.forest_q_persist)
                                                     : (unsigned)(-f->opt.forest_q_persist);
    const bool persist = f->opt.forest_q_persist != 0 && grid > want;
    if (persist) grid = want;
    // two waves per tree when the groups leave half the waves without one
    const bool wpt2 = L.ch == 4 && f->q_slots <= 8 && f->opt.forest_q_wpt != 1;
    const bool early = wpt2 && f->q_slot_bytes > 0;
    q_slot_table slots_at;
    for (int i = 0; i < 16; i++) slots_at.off[i] = L.slot_off[i];
    // rows of 16 KiB of the largest group: forest_qr_kernel loads that many without looking
    const int rows = (f->q_max_group_bytes + 16383) / 16384;
    f->last_cut = 0;
    // the cut for the generic kernel (forest_q_kernel; decided here, used only if that kernel is launched)
    int gcut = 0;
    qr_split_args gsp{};
    const bool qr_shape = L.ch == 4 && wpt2 && !early && (f->opt.forest_q_rsv & 1) && !(f->opt.forest_dbg & (8 | 32)) &&
                          (L.half1 == 32768 || L.half1 == 49152) && rows <= 5;
    const bool q2_shape = L.ch == 1 && f->opt.forest_q_two && cn > 64 && f->q_F <= 639 && f->q_max_group_bytes <= 6 * 16384;
    if (!qr_shape && !q2_shape && !early && split_sum > -1e300 && f->opt.forest_split && scratch &&
        cn >= f->opt.forest_split_min && ctx->split_k < PK_SPLIT_SLOTS && (L.ch == 1 || L.ch == 2 || L.ch == 4))
        gcut = q_pick_cut(f, split_sum);
    if (gcut > 0) {
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t per_tile = L.ch == 1 ? 64 : 128;
        const size_t cap = (size_t)cn + 256u * (size_t)grid + 256u;
        const size_t tiles_b = up((cap + per_tile - 1) / per_tile * f->q_F * per_tile * sizeof(unsigned short) + PK_Q_PAD_BYTES);
        const size_t acc_b = up(cap * 8), idx_b = up(cap * 4), st_b = up(cap);
        if (tiles_b + acc_b + idx_b + st_b > scratch_bytes) {
            gcut = 0;
        } else {
            if (!ctx->split_cnt) {
                PK_HIP(hipMalloc((void **)&ctx->split_cnt, PK_SPLIT_SLOTS * sizeof(unsigned)));
                ctx->split_k = 0;
            }
            if (ctx->split_k == 0)
                PK_HIP(hipMemsetAsync(ctx->split_cnt, 0, PK_SPLIT_SLOTS * sizeof(unsigned), ctx->stream));
            char *b = static_cast<char *>(scratch);
            gsp.tiles = reinterpret_cast<unsigned short *>(b);
            gsp.acc = reinterpret_cast<double *>(b + tiles_b);
            gsp.idx = reinterpret_cast<int32_t *>(b + tiles_b + acc_b);
            gsp.st = reinterpret_cast<uint8_t *>(b + tiles_b + acc_b + idx_b);
            gsp.cnt = ctx->split_cnt + ctx->split_k++;
            ctx->split_n += cn;
            ctx->split_slack += 128 * (int64_t)grid;
            gsp.rem = (double)f->q_T - (double)f->q_gtab_h[4 * (size_t)gcut];
            f->last_cut = gcut;
        }
    }
    if (L.ch == 4 && wpt2 && !early && (f->opt.forest_q_rsv & 1) && !(f->opt.forest_dbg & (8 | 32)) &&
        (L.half1 == 32768 || L.half1 == 49152) && rows <= 5) {
        // the cut: allowed by the caller (split_sum = thre * T: the run may leave decided candidates at
        // probability 0), worth two launches (forest_split_min candidates), room for the parked candidates
        // (worst case: all of them) in the scratch the caller lends -- the chunk's dead float tiles
        int cut = 0;
        qr_split_args sp{};
        unsigned grid_tail = 1;
        if (split_sum > -1e300 && f->opt.forest_split && scratch && cn >= f->opt.forest_split_min &&
            ctx->split_k < PK_SPLIT_SLOTS)
            cut = q_pick_cut(f, split_sum);
        if (cut > 0) {
            auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
            // (slots are handed out in blocks of 256 per workgroup: at most one unfinished block each)
            const size_t cap = (size_t)cn + 256u * (size_t)grid + 256u;
            const size_t tiles_b = up((cap + 127) / 128 * f->q_F * 128 * sizeof(unsigned short) + PK_Q_PAD_BYTES);
            const size_t acc_b = up(cap * 8), idx_b = up(cap * 4), st_b = up(cap);
            if (tiles_b + acc_b + idx_b + st_b > scratch_bytes) {
                cut = 0;
            } else {
                if (!ctx->split_cnt) {
                    PK_HIP(hipMalloc((void **)&ctx->split_cnt, PK_SPLIT_SLOTS * sizeof(unsigned)));
                    ctx->split_k = 0;
                }
                if (ctx->split_k == 0)
                    PK_HIP(hipMemsetAsync(ctx->split_cnt, 0, PK_SPLIT_SLOTS * sizeof(unsigned), ctx->stream));
                char *b = static_cast<char *>(scratch);
                sp.tiles = reinterpret_cast<unsigned short *>(b);
                sp.acc = reinterpret_cast<double *>(b + tiles_b);
                sp.idx = reinterpret_cast<int32_t *>(b + tiles_b + acc_b);
                sp.st = reinterpret_cast<uint8_t *>(b + tiles_b + acc_b + idx_b);
                sp.cnt = ctx->split_cnt + ctx->split_k++;
                ctx->split_n += cn;
                ctx->split_slack += 128 * (int64_t)grid;
                sp.rem = (double)f->q_T - (double)f->q_gtab_h[4 * (size_t)cut];
                grid_tail = grid;
            }
        }
        // the default shape: staging registers outside the compiler's reach, loads from inside the walk
#define QR_LAUNCH(HALF1, PRUNE, NR)                                                                        \
    do {                                                                                                   \
        int rc__ = q_set_max_lds(forest_qr_kernel<HALF1, PRUNE, NR, 0>, 163840);                           \
        if (rc__) return rc__;                                                                             \
        hipLaunchKernelGGL((forest_qr_kernel<HALF1, PRUNE, NR, 0>), dim3(grid), dim3(Q_THREADS), 163840, ctx->stream, \
                           reinterpret_cast<const v4u *>(f->q_img), reinterpret_cast<const int4 *>(f->q_gtab), \
                           f->q_n_grp, reinterpret_cast<const int4 *>(f->q_ttab), f->q_T, f->T, f->q_F, L.dec_off,   \
                           L.val_off, L.img_off, ctx->q_tiles, d_status, c0, cn, d_prob, prune_sum,        \
                           (int)(f->opt.forest_q_rsv >> 1), (int)f->opt.forest_dbg, ctx->dbg_buf, qr_split_args{}); \
    } while (0)
#define QR_LAUNCH_NR(HALF1, PRUNE)                                  \
    do {                                                            \
        if (rows <= 4) QR_LAUNCH(HALF1, PRUNE, 4);                  \
        else QR_LAUNCH(HALF1, PRUNE, 5);                            \
    } while (0)
        // the forest cut in two (see q_pick_cut): head over every candidate, tail over the parked ones
#define QR_LAUNCH_CUT(HALF1, NR)                                                                           \
    do {                                                                                                   \
        int rc__ = q_set_max_lds(forest_qr_kernel<HALF1, false, NR, 1>, 163840);                           \
        if (!rc__) rc__ = q_set_max_lds(forest_qr_kernel<HALF1, false, NR, 2>, 163840);                    \
        if (rc__) return rc__;                                                                             \
        hipLaunchKernelGGL((forest_qr_kernel<HALF1, false, NR, 1>), dim3(grid), dim3(Q_THREADS), 163840, ctx->stream, \
                           reinterpret_cast<const v4u *>(f->q_img), reinterpret_cast<const int4 *>(f->q_gtab), \
                           cut, reinterpret_cast<const int4 *>(f->q_ttab), f->q_T, f->T, f->q_F, L.dec_off,  \
                           L.val_off, L.img_off, ctx->q_tiles, d_status, c0, cn, d_prob, split_sum,        \
                           (int)(f->opt.forest_q_rsv >> 1), (int)f->opt.forest_dbg, ctx->dbg_buf, sp);      \
        PK_HIP(hipGetLastError());                                                                         \
        {                                                                                                  \
            pk_prof_scope prof_tail(ctx, PK_K_FOREST_TAIL);                                                \
            hipLaunchKernelGGL((forest_qr_kernel<HALF1, false, NR, 2>), dim3(grid_tail), dim3(Q_THREADS), 163840, ctx->stream, \
                               reinterpret_cast<const v4u *>(f->q_img), reinterpret_cast<const int4 *>(f->q_gtab) + cut, \
                               f->q_n_grp - cut, reinterpret_cast<const int4 *>(f->q_ttab), f->q_T, f->T, f->q_F, L.dec_off, \
                               L.val_off, L.img_off, sp.tiles, sp.st, c0, (int64_t)0, d_prob, split_sum,    \
                               (int)(f->opt.forest_q_rsv >> 1), (int)f->opt.forest_dbg & ~16, ctx->dbg_buf, sp); /* (stamps: the head's) */ \
        }                                                                                                  \
    } while (0)
        // forest_q_rsv bits: 1 on, 2 PREF0 (the next tile's first group prefetched with the tile)
        const bool prune = prune_sum > -1e300;
        f->last_family = 1;
        f->last_cut = cut;
        if (cut > 0) {
            if (L.half1 == 32768) {
                if (rows <= 4) QR_LAUNCH_CUT(32768, 4);
                else QR_LAUNCH_CUT(32768, 5);
            } else {
                if (rows <= 4) QR_LAUNCH_CUT(49152, 4);
                else QR_LAUNCH_CUT(49152, 5);
            }
        } else if (L.half1 == 32768) {
            if (prune) QR_LAUNCH_NR(32768, true);
            else QR_LAUNCH_NR(32768, false);
        } else {
            if (prune) QR_LAUNCH_NR(49152, true);
            else QR_LAUNCH_NR(49152, false);
        }
#undef QR_LAUNCH_CUT
#undef QR_LAUNCH_NR
#undef QR_LAUNCH
    } else if (L.ch == 4 && L.half1 == 32768) {
        f->last_family = 2;
        if (early) Q_LAUNCH(4, 2, 32768, true);
        else if (gcut > 0 && wpt2) Q_LAUNCH_CUT(4, 2, 32768);
        else if (gcut > 0) Q_LAUNCH_CUT(4, 1, 32768);
        else if (wpt2) Q_LAUNCH(4, 2, 32768, false);
        else Q_LAUNCH(4, 1, 32768, false);
    } else if (L.ch == 4 && L.half1 == 49152) {
        f->last_family = 2;
        if (early) Q_LAUNCH(4, 2, 49152, true);
        else if (gcut > 0 && wpt2) Q_LAUNCH_CUT(4, 2, 49152);
        else if (gcut > 0) Q_LAUNCH_CUT(4, 1, 49152);
        else if (wpt2) Q_LAUNCH(4, 2, 49152, false);
        else Q_LAUNCH(4, 1, 49152, false);
    } else if (L.ch == 2) {
        f->last_family = 2;
        if (gcut > 0) Q_LAUNCH_CUT(2, 1, 32768);
        else Q_LAUNCH(2, 1, 32768, false);
    } else if (L.ch == 1 && f->opt.forest_q_two && cn > 64 && f->q_F <= 639 && f->q_max_group_bytes <= 6 * 16384) {
        // two rank tiles per trip (see forest_q2_kernel); the waves that walk load their share of
        // the next group behind the first walk (option forest_q_help, on).  The kernel has no early
        // exit: a run that ALLOWS pruning (Chromosome.score's, pk_cands_set_prune) gets every
        // candidate's full probability here -- the scored pixels are the same either way, and the
        // one-tile kernel with its early exit is the slower of the two on these forests (round 5)
        f->last_family = 3;
        const int late_below = f->opt.forest_q_help ? f->q_slots : 0;
        unsigned grid2 = (unsigned)((cn + 127) / 128);
        if (f->opt.forest_q_persist != 0 && grid2 > want) grid2 = want;
        // the cut (see the default shape above): the spare tile must fit over the tree images
        int cut = 0;
        qr_split_args sp{};
        if (split_sum > -1e300 && f->opt.forest_split && scratch && cn >= f->opt.forest_split_min &&
            ctx->split_k < PK_SPLIT_SLOTS && L.HB <= L.cap && L.slots * 64 * 8 >= 4 * 130)
            cut = q_pick_cut(f, split_sum);
        if (cut > 0) {
            auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
            const size_t cap = (size_t)cn + 128u * (size_t)grid2 + 256u;
            const size_t tiles_b = up((cap + 63) / 64 * f->q_F * 64 * sizeof(unsigned short) + PK_Q_PAD_BYTES);
            const size_t acc_b = up(cap * 8), idx_b = up(cap * 4), st_b = up(cap);
            if (tiles_b + acc_b + idx_b + st_b > scratch_bytes) {
                cut = 0;
            } else {
                if (!ctx->split_cnt) {
                    PK_HIP(hipMalloc((void **)&ctx->split_cnt, PK_SPLIT_SLOTS * sizeof(unsigned)));
                    ctx->split_k = 0;
                }
                if (ctx->split_k == 0)
                    PK_HIP(hipMemsetAsync(ctx->split_cnt, 0, PK_SPLIT_SLOTS * sizeof(unsigned), ctx->stream));
                char *b = static_cast<char *>(scratch);
                sp.tiles = reinterpret_cast<unsigned short *>(b);
                sp.acc = reinterpret_cast<double *>(b + tiles_b);
                sp.idx = reinterpret_cast<int32_t *>(b + tiles_b + acc_b);
                sp.st = reinterpret_cast<uint8_t *>(b + tiles_b + acc_b + idx_b);
                sp.cnt = ctx->split_cnt + ctx->split_k++;
                ctx->split_n += cn;
                ctx->split_slack += 64 * (int64_t)grid2;
                sp.rem = (double)f->q_T - (double)f->q_gtab_h[4 * (size_t)cut];
            }
        }
        f->last_cut = cut;
#define Q2_LAUNCH(SPLIT_, GRID_, GTAB_, NGRP_, TILES_, STATUS_, CN_)                                              \
    do {                                                                                                           \
        int rc2 = q_set_max_lds(forest_q2_kernel<SPLIT_>, 163840);                                                 \
        if (rc2) return rc2;                                                                                       \
        hipLaunchKernelGGL(forest_q2_kernel<SPLIT_>, dim3(GRID_), dim3(Q_THREADS), 163840, ctx->stream,            \
                           reinterpret_cast<const v4u *>(f->q_img), GTAB_, NGRP_,                                  \
                           reinterpret_cast<const int4 *>(f->q_ttab), f->q_T, f->T, f->q_F, L.val_off, L.img_off,  \
                           TILES_, STATUS_, c0, CN_, d_prob, ctx->dbg_buf, (int)f->opt.forest_dbg, late_below, sp,  \
                           split_sum);                                                                             \
    } while (0)
        if (cut > 0) {
            Q2_LAUNCH(1, grid2, reinterpret_cast<const int4 *>(f->q_gtab), cut, ctx->q_tiles, d_status, cn);
            PK_HIP(hipGetLastError());
            pk_prof_scope prof_tail(ctx, PK_K_FOREST_TAIL);
            Q2_LAUNCH(2, grid2, reinterpret_cast<const int4 *>(f->q_gtab) + cut, f->q_n_grp - cut, sp.tiles, sp.st, (int64_t)0);
        } else {
            Q2_LAUNCH(0, grid2, reinterpret_cast<const int4 *>(f->q_gtab), f->q_n_grp, ctx->q_tiles, d_status, cn);
        }
#undef Q2_LAUNCH
    } else if (L.ch == 1) {
        f->last_family = 2;
        if (gcut > 0) Q_LAUNCH_CUT(1, 1, 32768);
        else Q_LAUNCH(1, 1, 32768, false);
    } else {
        pk_set_error("forest rank kernel: layout not instantiated");
        return PK_E_INVALID;
    }
    PK_HIP(hipGetLastError());
    return PK_OK;
}
