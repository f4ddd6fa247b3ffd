// pk_extract.hip -- window gather + distance-normalise + Gaussian blur +
// min-max scale for gfx950 (CDNA4), bit-exact float64.
//
// Replaces Chromosome.getwindow (peakachu/scoreUtils.py:70-93), which calls
// utils.distance_normalize (peakachu/utils.py:211-237),
// utils.distance_normaize_core (:180-202), scipy.ndimage.gaussian_filter
// (sigma=1) and utils.image_normalize (:204-209) once per candidate.
//
// Design (MI355X-first):
//  * the contact matrix lives in HBM as a diagonal-major dense band, so for a
//    wave of 64 candidates that are consecutive along one diagonal -- the
//    reference's candidate order -- every one of the (2w+1)^2 window-cell
//    loads is a contiguous 512-byte read;
//  * the window lives in VGPRs, fully unrolled so every register index is static: two lanes
//    per candidate at w = 5, 6 (extract_pair_*: each lane half the window, DPP swaps for the
//    row blur), one lane per candidate in the first kernel (extract_reg_kernel);
//  * for lists of neighbours on clean matrices the wave's strip of the band is staged in LDS
//    by LDS-DMA and the window read from there (extract_pair_strip_kernel, round 6: the
//    default of the benchmark regime); scattered lists gather straight from the band;
//  * the arithmetic order is scipy's / numba's exactly and the file is built
//    with -ffp-contract=off: no FMA contraction, IEEE division;
//  * float32 features leave in [tile][F][BLK] order so the store of feature
//    f by a wave is one 256-byte line and the forest kernel can copy a tile
//    into LDS with unit stride.
//  * other window sizes (w=11 stress config) use a one-wave-per-candidate
//    kernel that keeps the window in LDS.
#include "pk_common.h"

// scoreUtils.py:75 tests two edges only; candidates come from the upper triangle
// (x <= y, scoreUtils.py:40-68), where the other two follow.  For x > y the reference's
// fancy index raises IndexError (x + w >= n) or wraps (y - w < 0); such a window is
// filtered here, so no kernel ever reads outside the band.
#define PK_OTHER_EDGES(xi, yi, W, n) ((xi) + (W) < (n) && (yi) - (W) >= 0)

#ifndef PK_CLEAN_OCC5
#define PK_CLEAN_OCC5 2  // waves per SIMD of the clean two-lane extractor at w = 5 (3 spills 30 registers: 1.81 vs 1.20 ms)
#endif
#ifndef PK_EXTRACT_OCC
#define PK_EXTRACT_OCC 2  // waves per SIMD the two-lane extractor (w=5) is compiled for
#endif

namespace {

// scipy.ndimage._filters._gaussian_kernel1d(sigma=1, order=0, radius=4) = exp(-x^2/2)/sum:
// centre tap and the four on one side.  The defaults are what numpy 2.2 / scipy 1.15 give
// (the environment of the golden fixtures); the last bit of two of them depends on the
// numpy that evaluates exp() -- numpy 1.26 differs by one ulp -- so the host passes the taps
// of ITS numpy at load time (pk_set_gauss_taps) and the kernels read them from constant
// memory (scalar loads; a 64-bit literal would be materialised in registers as well).
__constant__ double pk_gk[5] = {0x1.9884a307594fbp-2, 0x1.ef8eb9ad499bap-3, 0x1.ba4b99d1799abp-5,
                                0x1.22724cb7eb269p-8, 0x1.18a9c4fd536c6p-13};
#define GK0 pk_gk[0]
#define GK1 pk_gk[1]
#define GK2 pk_gk[2]
#define GK3 pk_gk[3]
#define GK4 pk_gk[4]

// scipy 'reflect' line extension: (d c b a | a b c d | d c b a)
__host__ __device__ constexpr int reflect_idx(int i, int n)
{
    while (i < 0 || i >= n) i = (i < 0) ? (-i - 1) : (2 * n - 1 - i);
    return i;
}

// NI_Correlate1D, symmetric kernel: centre tap first, then outermost pair
// inwards; one rounding per operation (no FMA).
#define PK_BLUR9(c, m4, p4, m3, p3, m2, p2, m1, p1) \
    ((((((c) * GK0 + ((m4) + (p4)) * GK4) + ((m3) + (p3)) * GK3) + ((m2) + (p2)) * GK2) + ((m1) + (p1)) * GK1))

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// ------------------------------------------------------------------------
// Register kernel: one candidate per lane.
// ------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(64) void extract_reg_kernel(
    const double *__restrict__ band, int64_t ld, int dlo, int dhi, int n,
    const double *__restrict__ exp_arr, int exp_len, const int32_t *__restrict__ xs,
    const int32_t *__restrict__ ys, int64_t c0, int64_t cn, float *__restrict__ tiles, int blk,
    uint8_t *__restrict__ status, double *__restrict__ fea64_rows)
{
    constexpr int S = 2 * W + 1;
    constexpr int F = S * S;
    const int64_t local = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (local >= cn) return;
    const int64_t c = c0 + local;
    const int xi = xs[c], yi = ys[c];

    // scoreUtils.py:75: the window must lie inside the matrix
    if (!(xi - W >= 0 && yi + W + 1 <= n && PK_OTHER_EDGES(xi, yi, W, n))) {
        status[c] = 0;
        return;
    }
    const int d = yi - xi;
    const int64_t r0 = (int64_t)(xi - W);

    // ---- gather (scoreUtils.py:77-82): win[i][j] = M[x-w+i, y-w+j];
    // cell (i,j) sits on diagonal k = d + j - i of the band.
    double win[S][S];
#pragma unroll
    for (int i = 0; i < S; i++) {
#pragma unroll
        for (int j = 0; j < S; j++) {
            const int k = d + j - i;
            double v = 0.0;
            if (k >= dlo && k <= dhi) v = band[(int64_t)(k - dlo) * ld + r0 + i];
            win[i][j] = v;
        }
    }

    // ---- utils.py:221-225: NaN -> 0, sparsity filter
    int nnz = 0;
#pragma unroll
    for (int i = 0; i < S; i++) {
#pragma unroll
        for (int j = 0; j < S; j++) {
            double v = win[i][j];
            v = (v != v) ? 0.0 : v;
            win[i][j] = v;
            nnz += (v != 0.0) ? 1 : 0;
        }
    }
    bool ok = !((double)nnz < (double)F * 0.1);

    // ---- utils.py:228-232: top-left w x w mean (numba: sequential C order)
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < W; i++) {
#pragma unroll
        for (int j = 0; j < W; j++) acc += win[i][j];
    }
    const double ll_mean = acc / (double)(W * W);
    ok = ok && (ll_mean > 0.0);
    const double p2ll = win[W][W] / ll_mean;
    ok = ok && (p2ll > 0.1);
    if (!ok) {
        status[c] = 0;
        return;
    }

    // ---- utils.py:180-202: divide by expected(|col-row|) unless the largest
    // distance in the window falls outside exp_arr
    const int dmax = max(iabs(d - 2 * W), iabs(d + 2 * W));
    if (dmax < exp_len) {
        double e[4 * W + 1];
#pragma unroll
        for (int q = 0; q <= 4 * W; q++) e[q] = exp_arr[iabs(d - 2 * W + q)];
#pragma unroll
        for (int i = 0; i < S; i++) {
#pragma unroll
            for (int j = 0; j < S; j++) win[i][j] = win[i][j] / e[j - i + 2 * W];
        }
    }

    // ---- scipy gaussian_filter(sigma=1): axis 0 (down each column) ...
#pragma unroll
    for (int j = 0; j < S; j++) {
        double col[S];
#pragma unroll
        for (int i = 0; i < S; i++) col[i] = win[i][j];
#pragma unroll
        for (int i = 0; i < S; i++) {
            win[i][j] = PK_BLUR9(col[i], col[reflect_idx(i - 4, S)], col[reflect_idx(i + 4, S)],
                                 col[reflect_idx(i - 3, S)], col[reflect_idx(i + 3, S)],
                                 col[reflect_idx(i - 2, S)], col[reflect_idx(i + 2, S)],
                                 col[reflect_idx(i - 1, S)], col[reflect_idx(i + 1, S)]);
        }
    }
    // ... then axis 1 (along each row)
#pragma unroll
    for (int i = 0; i < S; i++) {
        double row[S];
#pragma unroll
        for (int j = 0; j < S; j++) row[j] = win[i][j];
#pragma unroll
        for (int j = 0; j < S; j++) {
            win[i][j] = PK_BLUR9(row[j], row[reflect_idx(j - 4, S)], row[reflect_idx(j + 4, S)],
                                 row[reflect_idx(j - 3, S)], row[reflect_idx(j + 3, S)],
                                 row[reflect_idx(j - 2, S)], row[reflect_idx(j + 2, S)],
                                 row[reflect_idx(j - 1, S)], row[reflect_idx(j + 1, S)]);
        }
    }

    // ---- utils.py:204-209 image_normalize; numpy min/max propagate NaN
    double mn = win[0][0], mx = win[0][0];
    bool has_nan = false;
#pragma unroll
    for (int i = 0; i < S; i++) {
#pragma unroll
        for (int j = 0; j < S; j++) {
            const double v = win[i][j];
            has_nan = has_nan || (v != v);
            mn = (v < mn) ? v : mn;
            mx = (v > mx) ? v : mx;
        }
    }
    if (has_nan) {
        mn = __builtin_nan("");
        mx = mn;
    }
    const double den = mx - mn;

    const int64_t tile = local / blk;
    const int lane = (int)(local - tile * blk);
    float *tp = tiles + (size_t)tile * F * blk + lane;
    bool fea_nan = false;  // e.g. a constant window: 0/0
#pragma unroll
    for (int i = 0; i < S; i++) {
#pragma unroll
        for (int j = 0; j < S; j++) {
            const double v = (win[i][j] - mn) / den;
            win[i][j] = v;
            fea_nan = fea_nan || (v != v);
            tp[(size_t)(i * S + j) * blk] = (float)v;  // sklearn's float32 cast (RNE)
        }
    }
    if (fea64_rows) {
        double *fp = fea64_rows + (size_t)local * F;
#pragma unroll
        for (int i = 0; i < S; i++) {
#pragma unroll
            for (int j = 0; j < S; j++) fp[i * S + j] = win[i][j];
        }
    }
    status[c] = fea_nan ? 2 : 1;  // 2: NaN features (the forest kernel then honours missing_go_to_left)
}

// ------------------------------------------------------------------------
// Pair kernel: TWO lanes per candidate (default for w = 5, 6).
//
// The one-lane kernel above needs 346 (w=5) / 504 (w=6) registers, i.e. one
// wave per SIMD, and its dependent FP64 chains run ~2.5x off the issue rate
// with nothing to overlap.  Here lane A (even) owns window columns 0..w and
// lane B (odd) owns the SAME window rotated by 180 degrees (its local cell
// (i, q) is the global cell (2w-i, 2w-q)), so both lanes run identical code in
// local coordinates: local column 0 is a window edge (scipy's `reflect`),
// local column w is the centre column, and the "virtual" columns w+1..w+4
// needed by the row blur are the partner's local columns w-1..w-4 (of its
// local row 2w-i), fetched with one DPP lane swap each.  The Gaussian taps are
// symmetric and IEEE addition commutes, so every sum is the sum scipy
// computes; the column blur (axis 0) needs no exchange at all.  The half
// window takes 132 (w=5) / 182 (w=6) registers -> 2 waves per SIMD.
// ------------------------------------------------------------------------
__device__ __forceinline__ double lane_swap(double v)
{
    // quad_perm [1,0,3,2]: exchange with the neighbouring lane (the partner)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int lane_swap_i(int v)
{
    return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
}

// row blur of one local row: own columns 0..W (`own`) plus the 4 virtual columns
// W+1..W+4 (`recv`, the partner's columns W-1..W-4 of the matching row)
template <int W>
__device__ __forceinline__ void blur_row(const double (&own)[W + 1], const double (&recv)[4],
                                         double (&out)[W + 1])
{
#define RR(q_) ((q_) < 0 ? own[-(q_) - 1] : ((q_) <= W ? own[(q_)] : recv[(q_) - W - 1]))
#pragma unroll
    for (int q = 0; q <= W; q++)
        out[q] = PK_BLUR9(own[q], RR(q - 4), RR(q + 4), RR(q - 3), RR(q + 3), RR(q - 2), RR(q + 2),
                          RR(q - 1), RR(q + 1));
#undef RR
}

// (w=6 needs 273 registers: one wave per SIMD; forcing two spills and is slower)
template <int W>
__global__ __launch_bounds__(64, (W <= 5 ? PK_EXTRACT_OCC : 1)) void extract_pair_kernel(
    const double *__restrict__ band, int64_t ld, int dlo, int dhi, int n,
    const double *__restrict__ exp_arr, int exp_len, const int32_t *__restrict__ xs,
    const int32_t *__restrict__ ys, int64_t c0, int64_t cn, float *__restrict__ tiles, int blk,
    uint8_t *__restrict__ status, double *__restrict__ fea64_rows)
{
    constexpr int S = 2 * W + 1;
    constexpr int F = S * S;
    constexpr int H = W + 1;  // local columns 0..W
    static_assert(W >= 4, "the row blur borrows 4 partner columns");
    const int role = threadIdx.x & 1;  // 0: lane A, 1: lane B (rotated window)
    const int64_t local = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 1);
    // both lanes of a pair share `local`, xi, yi and therefore every decision.
    // No early return: the DPP swaps need both lanes of every pair alive; a pair
    // that is out of range or filtered keeps computing on safe dummy coordinates
    // and simply stores nothing.
    const bool in_range = local < cn;
    const int64_t c = c0 + (in_range ? local : 0);
    const int xi = xs[c], yi = ys[c];
    bool ok = in_range && (xi - W >= 0 && yi + W + 1 <= n) &&  // scoreUtils.py:75
              PK_OTHER_EDGES(xi, yi, W, n);
    const int xc = ok ? xi : 0, yc = ok ? yi : 0;
    const int d = yc - xc;
    const int sgn = role ? -1 : 1;
    const int64_t r0 = (int64_t)(xc - W);

    // ---- gather: local (i, q) = global (gi, gj), diagonal k = d + gj - gi = d + sgn (q - i)
    double win[S][H];
#pragma unroll
    for (int i = 0; i < S; i++) {
#pragma unroll
        for (int q = 0; q < H; q++) {
            const int gi = role ? 2 * W - i : i;
            const int k = d + sgn * (q - i);
            double v = 0.0;
            if (ok && k >= dlo && k <= dhi) v = band[(int64_t)(k - dlo) * ld + r0 + gi];
            win[i][q] = v;
        }
    }
    // ---- utils.py:221-225: NaN -> 0, sparsity filter (B skips the shared centre column)
    int nnz = 0;
#pragma unroll
    for (int i = 0; i < S; i++) {
#pragma unroll
        for (int q = 0; q < H; q++) {
            double v = win[i][q];
            v = (v != v) ? 0.0 : v;
            win[i][q] = v;
            const bool mine = (q < W) || (role == 0);
            nnz += (mine && v != 0.0) ? 1 : 0;
        }
    }
    nnz += lane_swap_i(nnz);
    ok = ok && !((double)nnz < (double)F * 0.1);
    // ---- utils.py:228-232: top-left w x w mean (numba: sequential C order) = lane A's
    // local rows / columns 0..w-1
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < W; i++) {
#pragma unroll
        for (int q = 0; q < W; q++) acc += win[i][q];
    }
    const double acc_partner = lane_swap(acc);
    acc = role ? acc_partner : acc;
    const double ll_mean = acc / (double)(W * W);
    ok = ok && (ll_mean > 0.0);
    const double p2ll = win[W][W] / ll_mean;  // the centre cell is local (W, W) in both lanes
    ok = ok && (p2ll > 0.1);

    // ---- utils.py:180-202: divide by expected(|col-row|), col-row = d + sgn (q - i)
    const int dmax = max(iabs(d - 2 * W), iabs(d + 2 * W));
    if (dmax < exp_len) {
        double e[3 * W + 1];  // m = q - i + 2W runs over 0 .. 3W
#pragma unroll
        for (int m = 0; m <= 3 * W; m++) e[m] = exp_arr[iabs(d + sgn * (m - 2 * W))];
#pragma unroll
        for (int i = 0; i < S; i++) {
#pragma unroll
            for (int q = 0; q < H; q++) win[i][q] = win[i][q] / e[q - i + 2 * W];
        }
    }
    // ---- scipy gaussian_filter(sigma=1), axis 0: down each local column
#pragma unroll
    for (int q = 0; q < H; q++) {
        double col[S];
#pragma unroll
        for (int i = 0; i < S; i++) col[i] = win[i][q];
#pragma unroll
        for (int i = 0; i < S; i++) {
            win[i][q] = PK_BLUR9(col[i], col[reflect_idx(i - 4, S)], col[reflect_idx(i + 4, S)],
                                 col[reflect_idx(i - 3, S)], col[reflect_idx(i + 3, S)],
                                 col[reflect_idx(i - 2, S)], col[reflect_idx(i + 2, S)],
                                 col[reflect_idx(i - 1, S)], col[reflect_idx(i + 1, S)]);
        }
    }
    // ---- axis 1: along each local row.  My row i continues into the partner's local
    // row 2W-i; rows i and 2W-i are processed together so that both still hold their
    // axis-0 values when they are exchanged.
#pragma unroll
    for (int ip = 0; ip <= W; ip++) {
        const int ia = ip, ib = S - 1 - ip;
        // exchange first (both rows still hold their axis-0 values) ...
        double ra[4], rb[4];
#pragma unroll
        for (int u = 1; u <= 4; u++) {
            ra[u - 1] = lane_swap(win[ib][W - u]);  // partner's row 2W-ia = ib
            rb[u - 1] = lane_swap(win[ia][W - u]);
        }
        // ... then blur and write back one row at a time (few live temporaries)
        {
            double out[H];
            blur_row<W>(win[ia], ra, out);
            if (ia != ib) {
                double outb[H];
                blur_row<W>(win[ib], rb, outb);
#pragma unroll
                for (int q = 0; q < H; q++) win[ib][q] = outb[q];
            }
#pragma unroll
            for (int q = 0; q < H; q++) win[ia][q] = out[q];
        }
    }
    // ---- utils.py:204-209 image_normalize; numpy min/max propagate NaN
    double mn = win[0][0], mx = win[0][0];
    int has_nan = 0;
#pragma unroll
    for (int i = 0; i < S; i++) {
#pragma unroll
        for (int q = 0; q < H; q++) {
            const double v = win[i][q];
            has_nan |= (v != v) ? 1 : 0;
            mn = (v < mn) ? v : mn;
            mx = (v > mx) ? v : mx;
        }
    }
    {
        const double omn = lane_swap(mn), omx = lane_swap(mx);
        mn = (omn < mn) ? omn : mn;
        mx = (omx > mx) ? omx : mx;
        has_nan |= lane_swap_i(has_nan);
    }
    if (has_nan) {
        mn = __builtin_nan("");
        mx = mn;
    }
    const double den = mx - mn;
    const int64_t tile = local / blk;
    const int tl = (int)(local - tile * blk);
    float *tp = tiles + (size_t)tile * F * blk + tl;
    int fea_nan = 0;
#pragma unroll
    for (int i = 0; i < S; i++) {
#pragma unroll
        for (int q = 0; q < H; q++) {
            const double v = (win[i][q] - mn) / den;
            fea_nan |= (v != v) ? 1 : 0;
            const bool mine = (q < W) || (role == 0);  // A stores the shared centre column
            const int gi = role ? 2 * W - i : i, gj = role ? 2 * W - q : q;
            if (ok && mine) {
                tp[(size_t)(gi * S + gj) * blk] = (float)v;  // sklearn's float32 cast (RNE)
                if (fea64_rows) fea64_rows[(size_t)local * F + gi * S + gj] = v;
            }
        }
    }
    fea_nan |= lane_swap_i(fea_nan);
    if (in_range && role == 0) status[c] = ok ? (fea_nan ? 2 : 1) : 0;
}

// ------------------------------------------------------------------------
// Two lanes per candidate, CLEAN matrices (the default scoring kernel).
//
// norm_band_kernel (below) has checked on the device that no count is NaN, negative, -0
// or >= 1e150, that non-zero counts and quotients are >= 1e-100, that every expected
// value is finite and positive and every quotient finite and < 1e150, and it has stored
// the quotients count / expected(|col-row|) -- the values distance_normalize computes
// for every window cell (utils.py:180-202) -- once per band cell, behind the raw band.
// Under those conditions
//  * the window is read from the quotient band (same IEEE division, same operands);
//    only the top-left mean and the centre test (utils.py:228-235) read raw counts;
//  * NaN -> 0, NaN tracking and the compare-select form of min / max are identities: no
//    NaN, infinity or -0 can reach them (sums and products of finite non-negative
//    doubles below 1e150); v_min_f64 / v_max_f64 are used;
//  * blurred values are 0 or in [1e-108, 1e152], so a = v - min is 0 or >= 1e-124 and
//    den = max - min is 0 or in [1e-124, 1e152].  For such operands v_div_scale_f64
//    leaves both unscaled and v_div_fixup_f64 passes the quotient through, i.e. the
//    IEEE division a / den the compiler emits is exactly
//        r = rcp(den); r = fma(r, fma(-den, r, 1), r) twice;
//        m = a * r;  q = fma(fma(-den, m, a), r, m)
//    (LLVM's f64 fdiv lowering); r depends on den only and is computed once per window
//    instead of once per feature.  den == 0 (constant window) takes the true division.
//  * addresses are 32-bit byte offsets from a scalar base (the launcher checks that the
//    two bands and the tile buffer stay below 4 GiB), one v_mad_i32_i24 per access; a
//    wave whose windows all lie inside the band (always, for candidates of the band)
//    gathers without per-cell range tests; lanes without a valid candidate shadow a
//    valid one of the wave and store nothing.  w = 6 is compiled for two waves per SIMD
//    like w = 5 (14 registers go to scratch: faster than one wave without spills).
// The arithmetic order is unchanged; tests compare this kernel bit for bit with the
// general one and with the CPU restatement of the reference.
// ------------------------------------------------------------------------
// Everything of a candidate pair behind the gather (shared by the register-gather kernel and the
// LDS-staged one below): the two tests of utils.py:228-235, the sparsity filter, both blur passes,
// the min-max scaling and the stores.  `win` = this lane's half window (quotients, or raw counts
// when the window is not normalised), `acc` = lane A's sequential top-left sum, `centre` = the raw
// centre cell; `ok` = the lane has a window (the filters below may still drop it), `st` = the lane of role A
// writes its candidate's status byte (a lane that only shadows another one does not).  STORE_DROPPED: a lane
// with `st` stores its features whether the filters drop it or not.
// A feature is written once and read once, by the quantizer: non-temporal stores keep the 1.5 GB of a chunk's
// float tiles from displacing the band in the L2 (and the quantizer behind runs faster: profiles/r06_ab_nontemporal.log)
#define PK_TILE_STORE(p_, v_) __builtin_nontemporal_store((v_), (p_))
template <int W, bool FEA64, bool STORE_DROPPED = false>
__device__ __forceinline__ void pair_clean_finish(
    double (&win)[2 * W + 1][W + 1], double acc, const double centre, bool ok, const bool st, const int role,
    const unsigned lane_id,
    const int64_t wave0, const int64_t local, const bool in_range, const int64_t c,
    float *__restrict__ tiles, const int blk, uint8_t *__restrict__ status, double *__restrict__ fea64_rows)
{
    constexpr int S = 2 * W + 1;
    constexpr int F = S * S;
    constexpr int H = W + 1;
    const int sgn = role ? -1 : 1;
    const double acc_partner = lane_swap(acc);
    acc = role ? acc_partner : acc;
    const double ll_mean = acc / (double)(W * W);
    ok = ok && (ll_mean > 0.0);
    const double p2ll = centre / ll_mean;
    ok = ok && (p2ll > 0.1);

    // ---- utils.py:221-225 sparsity filter (quotient != 0 <=> count != 0), counted column by
    // column on the way, and scipy gaussian_filter(sigma=1), axis 0: down each local column
    int nnz = 0;
#pragma unroll
    for (int q = 0; q < H; q++) {
        double col[S];
        const bool mine = (q < W) || (role == 0);
#pragma unroll
        for (int i = 0; i < S; i++) {
            col[i] = win[i][q];
            nnz += (mine && col[i] != 0.0) ? 1 : 0;
        }
#pragma unroll
        for (int i = 0; i < S; i++) {
            win[i][q] = PK_BLUR9(col[i], col[reflect_idx(i - 4, S)], col[reflect_idx(i + 4, S)],
                                 col[reflect_idx(i - 3, S)], col[reflect_idx(i + 3, S)],
                                 col[reflect_idx(i - 2, S)], col[reflect_idx(i + 2, S)],
                                 col[reflect_idx(i - 1, S)], col[reflect_idx(i + 1, S)]);
        }
    }
    nnz += lane_swap_i(nnz);
    ok = ok && !((double)nnz < (double)F * 0.1);
    // ---- axis 1 (see extract_pair_kernel)
#pragma unroll
    for (int ip = 0; ip <= W; ip++) {
        const int ia = ip, ib = S - 1 - ip;
        double ra[4], rb[4];
#pragma unroll
        for (int u = 1; u <= 4; u++) {
            ra[u - 1] = lane_swap(win[ib][W - u]);
            rb[u - 1] = lane_swap(win[ia][W - u]);
        }
        {
            double out[H];
            blur_row<W>(win[ia], ra, out);
            if (ia != ib) {
                double outb[H];
                blur_row<W>(win[ib], rb, outb);
#pragma unroll
                for (int q = 0; q < H; q++) win[ib][q] = outb[q];
            }
#pragma unroll
            for (int q = 0; q < H; q++) win[ia][q] = out[q];
        }
    }
    // ---- utils.py:204-209 image_normalize
    double mn = win[0][0], mx = win[0][0];
#pragma unroll
    for (int i = 0; i < S; i++) {
#pragma unroll
        for (int q = 0; q < H; q++) {
            mn = __builtin_fmin(mn, win[i][q]);
            mx = __builtin_fmax(mx, win[i][q]);
        }
    }
    mn = __builtin_fmin(mn, lane_swap(mn));
    mx = __builtin_fmax(mx, lane_swap(mx));
    const double den = mx - mn;
    const bool flat = !(den > 0.0);  // constant window: 0 / 0
    // tile cell (e, tl) with e = gi * S + gj; B's e is F-1 minus A's.  Every feature is
    // stored as soon as it is computed (its window register dies with it: keeping the
    // floats for a grouped store cost 20 registers, the difference between one and two
    // waves per SIMD at w = 6); A stores the shared centre column.
    const int64_t first = wave0 / blk;  // the 32 candidates of a wave share a tile
    char *tbase = reinterpret_cast<char *>(tiles + (size_t)first * F * blk);
    const int tl = (int)(wave0 - first * blk) + (lane_id >> 1);
    const int t0 = (tl + (role ? (F - 1) * blk : 0)) * 4;
    const int sblk4 = sgn * blk * 4;
    double *rp = FEA64 ? fea64_rows + (size_t)local * F : nullptr;
#define PK_PUT(i_, q_, v_)                                                                      \
    do {                                                                                        \
        if ((STORE_DROPPED ? st : ok) && ((q_) < W || role == 0)) {                             \
            PK_TILE_STORE(reinterpret_cast<float *>(tbase + (unsigned)(t0 + __mul24((i_) * S + (q_), sblk4))), \
                          (float)(v_)); /* sklearn's float32 cast (RNE) */                      \
            if (FEA64) rp[role ? F - 1 - ((i_) * S + (q_)) : (i_) * S + (q_)] = (v_);           \
        }                                                                                       \
    } while (0)
    if (flat) {
        const double qn = (mn - mn) / den;  // the true division
#pragma unroll
        for (int i = 0; i < S; i++) {
#pragma unroll
            for (int q = 0; q < H; q++) PK_PUT(i, q, qn);
        }
    } else {
        double r = __builtin_amdgcn_rcp(den);
        r = __builtin_fma(r, __builtin_fma(-den, r, 1.0), r);
        r = __builtin_fma(r, __builtin_fma(-den, r, 1.0), r);
#pragma unroll
        for (int i = 0; i < S; i++) {
#pragma unroll
            for (int q = 0; q < H; q++) {
                const double a = win[i][q] - mn;
                const double m = a * r;
                const double v = __builtin_fma(__builtin_fma(-den, m, a), r, m);
                PK_PUT(i, q, v);
            }
        }
    }
#undef PK_PUT
    if (st && role == 0) status[c] = ok ? (flat ? 2 : 1) : 0;
}

// (the body of the kernel, for wave `vblock` of `vgrid`)
// DIAG (round 5, scattered candidate lists -- the list get_candidate makes holds one band pixel in ~50): a
// lane's loads are issued DIAGONAL BY DIAGONAL of its window instead of column by column.  Neighbouring
// cells of a window diagonal are neighbouring doubles of the diagonal-major band, i.e. the same 64-byte
// line: requested back to back they hit the line the first of them fetched, twelve instructions apart
// (column order, 64 lanes x 12 other lines in between) the CU's vector cache has long dropped it.  With
// consecutive candidates in a wave (the benchmark lists) the lanes share their lines anyway and the column
// order -- the order the column blur consumes -- stays.
template <int W, bool FEA64, bool DIAG = false>
__device__ __forceinline__ void extract_pair_clean_body(
    const unsigned vblock, const unsigned vgrid, const unsigned lane_id,
    const double *__restrict__ band, unsigned norm_off, int ld, int dlo, int dhi, int n, int exp_len,
    const int32_t *__restrict__ xs, const int32_t *__restrict__ ys, int64_t c0, int64_t cn,
    float *__restrict__ tiles, int blk, uint8_t *__restrict__ status,
    double *__restrict__ fea64_rows)
{
    constexpr int S = 2 * W + 1;
    constexpr int H = W + 1;
    static_assert(W >= 4, "the row blur borrows 4 partner columns");
    const int role = lane_id & 1;
    // XCD-aware order: workgroup b runs on XCD b % 8 (round-robin dispatch), and each XCD
    // has its own L2.  Consecutive candidates share band rows, so XCD x takes the x-th
    // contiguous eighth of the chunk instead of every eighth block (the grid is a
    // multiple of 8): the band lines a window needs are then fetched into one L2, not
    // into all eight.
    const unsigned per_xcd = vgrid >> 3;
    const int64_t wave0 = (int64_t)((vblock & 7u) * per_xcd + (vblock >> 3)) * 32;
    const int64_t local = wave0 + (lane_id >> 1);
    const bool in_range = local < cn;
    const int64_t c = c0 + (in_range ? local : 0);
    const int xi = xs[c], yi = ys[c];
    bool ok = in_range && (xi - W >= 0 && yi + W + 1 <= n) && PK_OTHER_EDGES(xi, yi, W, n);
    // lanes without a window shadow the first valid candidate of the wave
    const unsigned long long okmask = __ballot(ok);
    if (okmask == 0ull) {  // wave-uniform
        if (in_range && role == 0) status[c] = 0;
        return;
    }
    const int lead = __builtin_ctzll(okmask);
    const int xc = ok ? xi : __builtin_amdgcn_readlane(xi, lead);
    const int yc = ok ? yi : __builtin_amdgcn_readlane(yi, lead);
    const int d = yc - xc;
    const int sgn = role ? -1 : 1;
    const bool normalise = max(iabs(d - 2 * W), iabs(d + 2 * W)) < exp_len;
    const bool inside = (d - 2 * W >= dlo) && (d + 2 * W <= dhi);

    // byte offset of local cell (i, q): row0 + sgn * (i * 8 + (q - i) * ld * 8)
    const char *bbase = reinterpret_cast<const char *>(band);
    const int ld8 = ld * 8;
    const unsigned raw0 = (unsigned)(((int64_t)(d - dlo) * ld + (xc - W) + (role ? 2 * W : 0)) * 8);
    const unsigned row0 = raw0 + (normalise ? norm_off : 0u);
    const int sld8 = sgn * ld8, s8 = sgn * 8;
#define PK_CELL(base_, i_, q_) \
    (*reinterpret_cast<const double *>(bbase + (unsigned)((base_) + (i_) * s8 + __mul24((q_) - (i_), sld8))))
    // ---- utils.py:228-235 on the raw counts: top-left w x w mean (lane A's local block,
    // sequential C order) and the centre cell.
    double acc = 0.0;
    double centre = 0.0;
    auto raw_block = [&]() {  // (general path: every cell tested against the band)
        if (role == 0) {
            double tl[W][W];
#pragma unroll
            for (int i = 0; i < W; i++) {
#pragma unroll
                for (int q = 0; q < W; q++) {
                    const int k = d + q - i;
                    tl[i][q] = (k >= dlo && k <= dhi) ? PK_CELL(raw0, i, q) : 0.0;
                }
            }
#pragma unroll
            for (int i = 0; i < W; i++) {
#pragma unroll
                for (int q = 0; q < W; q++) acc += tl[i][q];
            }
        }
        centre = (d >= dlo && d <= dhi)
                     ? *reinterpret_cast<const double *>(
                           bbase + (unsigned)(((int64_t)(d - dlo) * ld + xc) * 8))
                     : 0.0;
    };
    double win[S][H];
    const bool fast = __all(inside);  // every window of the wave lies inside the band
    if (fast) {
        // Overlapped memory phases (no cell needs a range test here): the raw top-left block,
        // the centre cell and the first half of the window are requested together; the block
        // is summed while that half is in flight (its registers are needed for the second
        // half: block + window = 254 registers at w = 6), then the second half is requested.
        // The window goes column by column, the order in which the column blur consumes it,
        // so the blur starts on column 0 while later columns are still in flight.  (Before:
        // window -> wait -> count -> block -> wait -> sum at w = 5, block -> wait -> sum ->
        // window -> wait at w = 6: two full latencies with nothing to do.)
        double tl[W][W];
        const bool want_tl = normalise && role == 0;
        if constexpr (DIAG) {
            // every load of a diagonal behind its neighbour on that diagonal (same line of the band)
            if (want_tl) {
#pragma unroll
                for (int t = -(W - 1); t <= W - 1; t++) {
#pragma unroll
                    for (int i = 0; i < W; i++)
                        if (i + t >= 0 && i + t < W) tl[i][i + t] = PK_CELL(raw0, i, i + t);
                }
            }
            if (normalise)
                centre = *reinterpret_cast<const double *>(bbase + (unsigned)(((int64_t)(d - dlo) * ld + xc) * 8));
#pragma unroll
            for (int t = -(S - 1); t <= H - 1; t++) {
#pragma unroll
                for (int i = 0; i < S; i++)
                    if (i + t >= 0 && i + t < H) win[i][i + t] = PK_CELL(row0, i, i + t);
            }
            if (want_tl) {
#pragma unroll
                for (int i = 0; i < W; i++) {
#pragma unroll
                    for (int q = 0; q < W; q++) acc += tl[i][q];
                }
            }
        } else {
        if (want_tl) {
#pragma unroll
            for (int i = 0; i < W; i++) {
#pragma unroll
                for (int q = 0; q < W; q++) tl[i][q] = PK_CELL(raw0, i, q);
            }
        }
        if (normalise)
            centre = *reinterpret_cast<const double *>(bbase + (unsigned)(((int64_t)(d - dlo) * ld + xc) * 8));
        constexpr int H1 = H / 2;  // the block is summed (its registers freed) before the second half
#pragma unroll
        for (int q = 0; q < H1; q++) {
#pragma unroll
            for (int i = 0; i < S; i++) win[i][q] = PK_CELL(row0, i, q);
        }
        if (want_tl) {
#pragma unroll
            for (int i = 0; i < W; i++) {
#pragma unroll
                for (int q = 0; q < W; q++) acc += tl[i][q];
            }
        }
#pragma unroll
        for (int q = H1; q < H; q++) {
#pragma unroll
            for (int i = 0; i < S; i++) win[i][q] = PK_CELL(row0, i, q);
        }
        }
    } else {
        if (normalise) raw_block();
#pragma unroll
        for (int i = 0; i < S; i++) {
#pragma unroll
            for (int q = 0; q < H; q++) {
                const int k = d + sgn * (q - i);
                double v = 0.0;
                if (k >= dlo && k <= dhi) v = PK_CELL(row0, i, q);
                win[i][q] = v;
            }
        }
    }
    if (!normalise) {  // the window holds the raw counts themselves
        centre = win[W][W];
#pragma unroll
        for (int i = 0; i < W; i++) {
#pragma unroll
            for (int q = 0; q < W; q++) acc += win[i][q];
        }
    }
#undef PK_CELL
    pair_clean_finish<W, FEA64>(win, acc, centre, ok, in_range, role, lane_id, wave0, local, in_range, c, tiles, blk, status,
                                fea64_rows);
}


template <int W, bool FEA64, bool DIAG = false>
__global__ __launch_bounds__(64, (!FEA64 ? (W <= 5 ? PK_CLEAN_OCC5 : 2) : 1)) void extract_pair_clean_kernel(
    const double *__restrict__ band, unsigned norm_off, int ld, int dlo, int dhi, int n, int exp_len,
    const int32_t *__restrict__ xs, const int32_t *__restrict__ ys, int64_t c0, int64_t cn,
    float *__restrict__ tiles, int blk, uint8_t *__restrict__ status,
    double *__restrict__ fea64_rows)
{
    extract_pair_clean_body<W, FEA64, DIAG>(blockIdx.x, gridDim.x, threadIdx.x, band, norm_off, ld, dlo, dhi, n,
                                            exp_len, xs, ys, c0, cn, tiles, blk, status, fea64_rows);
}

// ------------------------------------------------------------------------
// Two lanes per candidate, clean matrices, lists of NEIGHBOURS: the wave's diagonal strip staged in LDS.
//
// The register-gather kernel above asks for every window cell with a load of its own: 92 + 25 vector-memory
// instructions per wave, 60 KB through the CU's address path, a third of the wave's cycles waiting (PMC,
// round 5: SQ_WAIT_INST_ANY 35 %, VALU issue 0.68), the LDS unused.  The windows of candidates that lie on
// ONE diagonal d within ~50 rows of each other (the lists of the benchmark regime: every non-zero band
// pixel, diagonal by diagonal) are the 4w+1 diagonals d-2w .. d+2w of the band over 64 consecutive rows.
// A wave stages that strip in LDS by LDS-DMA (global_load_lds_dwordx4: whole lines, no VGPR, no ds_write
// -- 16 instructions at w = 5) and every lane reads its window from there at constant offsets (ds_read
// with immediates: no address arithmetic).
//   * staging image: rows of 64 doubles (half a DMA instruction: lane l of an instruction fetches the 16
//     bytes l & 31 of the instruction's row l >> 5), first double = the even row rs <= x0 - w (16-byte
//     source alignment; pulled back so that rs + 64 stays inside the band row): quotient rows
//     k = d-2w .. d+2w (raw counts when the window is not normalised), behind them the raw rows of the
//     top-left blocks, k = d-(w-1) .. d+(w-1), and the candidates' own diagonal.  A row outside the band
//     [dlo, dhi] is fetched from the nearest one inside and zeroed in LDS (scoreUtils.py:30-33: such
//     cells are not stored; the last candidate diagonal's far corner);
//   * a PASS handles the candidates of the wave that share the first open candidate's diagonal and fit
//     its strip; a wave that needs more than one (the end of a diagonal, a gap of > ~50 rows) repeats.
//     Lanes that are not members of a pass shadow its lead candidate and store nothing;
//   * what was tried and not kept (EXPERIMENTS.md, round 6): a wave taking several batches with the next
//     strip and the coordinates after next in flight behind the current batch (1.09-1.12 ms where one
//     batch per wave takes 1.06: the partner wave of the SIMD already hides the latency, the look-ahead
//     only adds instructions).
// Arithmetic: pair_clean_finish, the same instructions as the register-gather kernel.
// ------------------------------------------------------------------------
template <int W>
struct strip_layout {
    static constexpr int ROW = 64;            // doubles per staged row
    static constexpr int QD = 4 * W + 1;      // quotient rows
    static constexpr int QI = (QD + 1) / 2;   // DMA instructions: two rows each (the last one: one row)
    static constexpr int RI = W;              // raw rows: 2w-1 diagonals of the top-left blocks + the centre diagonal
    static constexpr int Q_DOUBLES = QI * 2 * ROW;
    static constexpr int DOUBLES = (QI + RI) * 2 * ROW;
    static constexpr int MAXPOS = ROW - 1 - 2 * W;  // largest row position (x - w - rs) of a member
};

// LDS-DMA: 64 lanes x 16 bytes from sbase + voff (per lane) to LDS lds_dst + lane * 16.  (M0 carries the LDS
// address; nothing else in these kernels uses it: LDS instructions need no M0 on gfx9 and later.)
__device__ __forceinline__ void strip_dma(unsigned voff, const char *sbase, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int W, bool FEA64>
__global__ __launch_bounds__(64, (!FEA64 ? 2 : 1)) void extract_pair_strip_kernel(
    const double *__restrict__ band, unsigned norm_off, int ld, int dlo, int dhi, int n, int exp_len,
    const int32_t *__restrict__ xs, const int32_t *__restrict__ ys, int64_t c0, int64_t cn,
    float *__restrict__ tiles, int blk, uint8_t *__restrict__ status, double *__restrict__ fea64_rows)
{
    using L = strip_layout<W>;
    constexpr int S = 2 * W + 1;
    constexpr int H = W + 1;
    __shared__ __attribute__((aligned(1024))) double strip[L::DOUBLES];
    const unsigned lds0 = (unsigned)(uintptr_t)strip;
    const unsigned lane_id = threadIdx.x;
    const int role = lane_id & 1;
    // XCD-aware order (see extract_pair_clean_body): XCD x takes the x-th contiguous eighth of the waves
    const unsigned per_xcd = gridDim.x >> 3;
    const int64_t wave0 = (int64_t)((blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3)) * 32;
    if (wave0 >= cn) return;
    const int64_t local = wave0 + (lane_id >> 1);
    const bool in_range = local < cn;
    const int64_t c = c0 + (in_range ? local : wave0);
    const int xi = __builtin_nontemporal_load(xs + c), yi = __builtin_nontemporal_load(ys + c);  // (read once)
    const bool valid = in_range && (xi - W >= 0 && yi + W + 1 <= n) && PK_OTHER_EDGES(xi, yi, W, n);
    if (in_range && !valid && role == 0) status[c] = 0;
    const char *bbase = reinterpret_cast<const char *>(band);
    const unsigned ld8 = (unsigned)ld * 8u;
    const unsigned dma_lo = (lane_id & 31u) * 16u;
    const unsigned dma_off = dma_lo + (lane_id >> 5) * ld8;
    auto clampk = [&](int k) { return k < dlo ? dlo : (k > dhi ? dhi : k); };

    unsigned long long open = __ballot(valid);  // candidates still to do (both lanes of each)
    while (open != 0ull) {
        const int lead = __builtin_ctzll(open);
        const int x0 = __builtin_amdgcn_readlane(xi, lead);
        const int d0 = __builtin_amdgcn_readlane(yi - xi, lead);
        // first staged row (even; the 64 rows stay inside the band row)
        const int rs = ((x0 - W) & ~1) + L::ROW <= ld ? ((x0 - W) & ~1) : ld - L::ROW;
        const bool normalise = max(iabs(d0 - 2 * W), iabs(d0 + 2 * W)) < exp_len;
        const unsigned noff = normalise ? norm_off : 0u;
        // ---- the strip around (x0, d0), on its way into LDS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (a previous pass's reads have returned)
#pragma unroll
        for (int j = 0; j < L::QI - 1; j++) {
            const int k0 = clampk(d0 - 2 * W + 2 * j), k1 = clampk(d0 - 2 * W + 2 * j + 1);
            const unsigned off = noff + (unsigned)(((int64_t)(k0 - dlo) * ld + rs) * 8);
            strip_dma(k1 != k0 ? dma_off : dma_lo, bbase + off, lds0 + (unsigned)j * 1024u);
        }
        if (lane_id < 32u) {  // the last quotient row: half an instruction
            const int k0 = clampk(d0 + 2 * W);
            strip_dma(dma_lo, bbase + (noff + (unsigned)(((int64_t)(k0 - dlo) * ld + rs) * 8)),
                      lds0 + (unsigned)(L::QI - 1) * 1024u);
        }
#pragma unroll
        for (int j = 0; j < L::RI - 1; j++) {
            const int k0 = clampk(d0 - (W - 1) + 2 * j), k1 = clampk(d0 - (W - 1) + 2 * j + 1);
            const unsigned off = (unsigned)(((int64_t)(k0 - dlo) * ld + rs) * 8);
            strip_dma(k1 != k0 ? dma_off : dma_lo, bbase + off, lds0 + (unsigned)(L::QI + j) * 1024u);
        }
        {   // last raw instruction: lower half = diagonal d + w - 1 of the blocks, upper half = the candidates' own
            // diagonal (the base; the lower half's offset from it is not negative)
            const int kc = clampk(d0), kt = clampk(d0 + W - 1);
            const unsigned coff = (unsigned)(((int64_t)(kc - dlo) * ld + rs) * 8);
            strip_dma(dma_lo + (lane_id < 32u ? (unsigned)(kt - kc) * ld8 : 0u), bbase + coff,
                      lds0 + (unsigned)(L::QI + L::RI - 1) * 1024u);
        }
        // members of this pass: same diagonal, row position inside the strip
        const int pos = xi - W - rs;
        const bool mem = ((open >> lane_id) & 1ull) && yi - xi == d0 && pos >= 0 && pos <= L::MAXPOS;
        open &= ~__ballot(mem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (d0 - 2 * W < dlo || d0 + 2 * W > dhi) {
            // rows outside the band hold a neighbour's values: they read as zero (absent cells)
            for (int r = 0; r < L::QD; r++) {
                const int k = d0 - 2 * W + r;
                if (k < dlo || k > dhi) strip[r * L::ROW + lane_id] = 0.0;
            }
            for (int r = 0; r < 2 * W; r++) {
                const int k = r < 2 * W - 1 ? d0 - (W - 1) + r : d0;
                if (k < dlo || k > dhi) strip[L::Q_DOUBLES + r * L::ROW + lane_id] = 0.0;
            }
        }
        // lanes that are not members shadow the lead candidate.  Every read below is base + constant: the
        // two roles (lane B holds the window rotated by 180 degrees) read in separate, masked groups
        const double *sp = strip + ((mem ? xi : x0) - W - rs);
        double win[S][H];
        double acc = 0.0, centre;
        centre = sp[L::Q_DOUBLES + (2 * W - 1) * L::ROW + W];
        if (role == 0) {
            if (normalise) {
#pragma unroll
                for (int i = 0; i < W; i++) {
#pragma unroll
                    for (int q = 0; q < W; q++) acc += sp[L::Q_DOUBLES + (q - i + W - 1) * L::ROW + i];
                }
            }
#pragma unroll
            for (int q = 0; q < H; q++) {
#pragma unroll
                for (int i = 0; i < S; i++) win[i][q] = sp[(q - i + 2 * W) * L::ROW + i];
            }
        } else {
#pragma unroll
            for (int q = 0; q < H; q++) {
#pragma unroll
                for (int i = 0; i < S; i++) win[i][q] = sp[(i - q + 2 * W) * L::ROW + 2 * W - i];
            }
        }
        if (!normalise) {  // the window holds the raw counts themselves
            centre = win[W][W];
#pragma unroll
            for (int i = 0; i < W; i++) {
#pragma unroll
                for (int q = 0; q < W; q++) acc += win[i][q];
            }
        }
        // (w = 5 also stores the features of a member the filters drop -- into its own cells of the tile, which
        // nobody reads: its status byte says so.  The stores then form one block the compiler schedules freely:
        // 1.075 ms against 1.150 with a branch around every store; at w = 6, where registers are scarce, the
        // branches win, 2.07 against 2.18: profiles/r06_strip_store_ab.log)
        pair_clean_finish<W, FEA64, (W <= 5 && !FEA64)>(win, acc, centre, mem, mem, role, lane_id, wave0, local, in_range, c,
                                                        tiles, blk, status, fea64_rows);
    }
}

// ------------------------------------------------------------------------
// Wide windows on clean matrices (w = 8 .. 15; instantiated for w = 11, the 23 x 23 / 529-
// feature stress configuration): FOUR windows per wave, one per 16-lane DPP row, the window
// register-blocked.
//
// The one-window-per-wave kernel below keeps the window in LDS and is bound by LDS
// instructions (~400 per window: 220 tap reads, the 121 reads of the sequential top-left sum,
// the window writes) and by its 529 divisions.  Here
//   * lane l of a row holds window COLUMNS l and l + 16 in registers (2 x S doubles): the
//     axis-0 blur (scipy's first pass, along the rows i of one column) is lane-local;
//   * the columns come from the pre-divided band (norm_band_kernel): no division per cell;
//   * one transpose through LDS (S writes + S reads of 8 bytes per lane and slot; window
//     buffers 16 (mod 32) doubles apart, so the 16 rows a read instruction touches in two
//     windows fall into disjoint bank pairs) turns columns into ROWS l and l + 16: the
//     axis-1 blur is lane-local again;
//   * numba's sequential top-left sum (utils.py:228: C order, one rounding per add) runs as
//     W*W dependent additions in every lane: the cell (i, j) is broadcast from lane j to its
//     row by DPP (row_newbcast:j), beside the chain -- FOUR windows at once;
//   * min / max are row reductions, the 2 x S divisions of the min-max scaling share one
//     refined reciprocal (exact under the clean matrix's bounds, see the w = 5 kernel).
// ------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_lane_f64(double v, int src_lane)  // value of lane `src_lane`
{
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// out[j] = the value lane j of the row holds, j = 0 .. N-1 (row_newbcast:j)
template <int N, int J = 0>
__device__ __forceinline__ void row_bcast_all(double v, double (&out)[N])
{
    if constexpr (J < N) {
        out[J] = dpp_f64<0x150 + J>(v);
        row_bcast_all<N, J + 1>(v, out);
    }
}

template <int W>
struct row16_geom {
    static constexpr int S = 2 * W + 1, F = S * S;
    static constexpr int NS = (S + 15) / 16;             // column (row) slots per lane
    static constexpr int TS = F + ((16 - F % 32) + 32) % 32;  // window stride in LDS, 16 (mod 32) doubles
};

// WAVES = 4 (round 6, the scoring path): four waves = sixteen consecutive candidates per workgroup.  A wave's
// feature stores are 16-byte pieces (its four candidates) of sixteen different lines per instruction -- a quarter
// of the kernel's time (ablation: profiles/r06_w11_ab.log); here every wave parks its float features in
// its own (by then dead) transpose buffer and, behind one barrier, the workgroup writes rows of sixteen
// candidates: 64 contiguous bytes per feature, four lines per store instruction: 3.07 -> 2.88 ms on configs[4].
// (Tried on top and not kept: the wave's strip of the band staged in LDS like extract_pair_strip_kernel's -- 18
// global_load_lds_dwordx4 for FOUR candidates cost more than the 58 per-lane loads they replace: 3.53 ms;
// profiles/r06_w11_ab.log.)
template <int W, bool FEA64, int WAVES = 1>
__global__ __launch_bounds__(64 * WAVES, 2) void extract_row16_clean_kernel(
    const double *__restrict__ band, unsigned norm_off, int ld, int dlo, int dhi, int n, int exp_len,
    const int32_t *__restrict__ xs, const int32_t *__restrict__ ys, int64_t c0, int64_t cn,
    float *__restrict__ tiles, int blk, uint8_t *__restrict__ status,
    double *__restrict__ fea64_rows)
{
    using G = row16_geom<W>;
    constexpr int S = G::S, F = G::F, NS = G::NS, TS = G::TS;
    static_assert(S > 16 && S <= 32 && NS == 2, "two column slots per lane");
    static_assert(W >= 4, "reflect folds once");
    static_assert(WAVES == 1 || (WAVES == 4 && !FEA64), "the workgroup form is the scoring path's");
    constexpr int REG = 4 * TS;  // a wave's LDS region: its four window buffers (transposes)
    __shared__ double T[WAVES * REG];
    __shared__ unsigned char okf[16];
    const unsigned lane = threadIdx.x & 63u, q = lane >> 4, l = lane & 15;
    const unsigned wv = WAVES == 1 ? 0u : (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // XCD-aware order (see the w = 5 kernel): XCD x takes the x-th contiguous eighth
    const unsigned per_xcd = gridDim.x >> 3;
    const int64_t wg0 = (int64_t)((blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3)) * (4 * WAVES);
    const int64_t wave0 = wg0 + 4 * wv;
    const int64_t local = wave0 + q;
    const bool in_range = local < cn;
    const int64_t c = c0 + (in_range ? local : 0);
    const int xi = xs[c], yi = ys[c];
    bool ok = in_range && (xi - W >= 0 && yi + W + 1 <= n) && PK_OTHER_EDGES(xi, yi, W, n);
    const unsigned long long okmask = __ballot(ok);
    // the workgroup's write-out (WAVES = 4): thread t takes candidate t & 15 of the sixteen and, per trip, feature
    // 16 * trip + (t >> 4) -- a wave stores four features x sixteen candidates, 64 contiguous bytes each
    auto write_out = [&]() {
        const unsigned t = threadIdx.x, c16 = t & 15u, fsub = t >> 4;
        const bool ok16 = okf[c16] != 0;
        const int64_t first = wg0 / blk;  // (the sixteen share a tile: blk is a multiple of 16)
        float *tp16 = tiles + (size_t)first * F * blk + (int)(wg0 - first * blk) + c16;
        const float *O = reinterpret_cast<const float *>(T + (c16 >> 2) * REG) + (c16 & 3u);
        if (ok16)
            for (int f = (int)fsub; f < F; f += 16) PK_TILE_STORE(tp16 + (size_t)f * blk, O[f * 4]);
    };
    if (okmask == 0ull) {  // wave-uniform
        if (in_range && l == 0) status[c] = 0;
        if constexpr (WAVES > 1) {
            if (l == 0) okf[wv * 4 + q] = 0;
            __syncthreads();
            write_out();
        }
        return;
    }
    // rows without a window shadow the first valid one of the wave (every load stays in range)
    const int lead = __builtin_ctzll(okmask);
    const int xc = ok ? xi : __builtin_amdgcn_readlane(xi, lead);
    const int yc = ok ? yi : __builtin_amdgcn_readlane(yi, lead);
    const int d = yc - xc;
    const bool normalise = max(iabs(d - 2 * W), iabs(d + 2 * W)) < exp_len;
    const bool inside = (d - 2 * W >= dlo) && (d + 2 * W <= dhi);
    const bool fast = __all(inside);

    const char *bbase = reinterpret_cast<const char *>(band);
    const int ld8 = ld * 8;
    const unsigned raw0 = (unsigned)(((int64_t)(d - dlo) * ld + (xc - W)) * 8);
    const unsigned row0 = raw0 + (normalise ? norm_off : 0u);
    // byte offset of window cell (i, j): base + i * 8 + (j - i) * ld * 8
#define PK_CELL(base_, i_, j_) \
    (*reinterpret_cast<const double *>(bbase + (unsigned)((base_) + (i_) * 8 + __mul24((int)(j_) - (i_), ld8))))
    int jj[NS];
    bool jv[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        jv[s] = (int)l + 16 * s < S;
        jj[s] = jv[s] ? (int)l + 16 * s : (int)l;  // a slot without a column shadows slot 0
    }
    // ---- the window, column-wise (scoreUtils.py:77-82; pre-divided: utils.py:180-202)
    double colv[NS][S];
    double rawc[W + 1];  // raw counts of rows 0..W of column l: the top-left block and the centre
    if (fast) {
        if (normalise) {
#pragma unroll
            for (int i = 0; i <= W; i++) rawc[i] = PK_CELL(raw0, i, l);
        }
#pragma unroll
        for (int s = 0; s < NS; s++) {
#pragma unroll
            for (int i = 0; i < S; i++) colv[s][i] = PK_CELL(row0, i, jj[s]);
        }
    } else {
        if (normalise) {
#pragma unroll
            for (int i = 0; i <= W; i++) {
                const int k = d + (int)l - i;
                rawc[i] = (k >= dlo && k <= dhi) ? PK_CELL(raw0, i, l) : 0.0;
            }
        }
#pragma unroll
        for (int s = 0; s < NS; s++) {
#pragma unroll
            for (int i = 0; i < S; i++) {
                const int k = d + jj[s] - i;
                colv[s][i] = (k >= dlo && k <= dhi) ? PK_CELL(row0, i, jj[s]) : 0.0;
            }
        }
    }
#undef PK_CELL
    if (!normalise) {
#pragma unroll
        for (int i = 0; i <= W; i++) rawc[i] = colv[0][i];
    }
    // ---- utils.py:228: window[:w, :w].mean() as numba computes it -- sequentially in C order.
    // Cell (i, j) lives in lane j: it is broadcast to the lanes of its row (row_newbcast:j, two
    // 32-bit DPP moves that do not depend on the sum), and every lane adds it: the dependent
    // chain is the W*W additions alone, the broadcasts of the next cells run beside them.
    // (First version: the partial sum itself travelled from lane to lane, two DPP moves
    // inside every link of the chain.)
    // (left alone the compiler hoists all W*W broadcasts -- 242 live registers, 118 spills --
    // and scheduling fences derail its register allocation elsewhere: the source of row i+1's
    // broadcasts is tied to the sum as it stands BEFORE row i by an empty asm, so at most two
    // rows of broadcasts are ever live)
    double acc = 0.0;
    double rowc[W], rown[W];
    row_bcast_all<W>(rawc[0], rowc);
#pragma unroll
    for (int i = 0; i < W; i++) {
        if (i + 1 < W) {
            double src = rawc[i + 1];
            asm volatile("" : "+v"(src) : "v"(acc));
            row_bcast_all<W>(src, rown);
        }
#pragma unroll
        for (int j = 0; j < W; j++) acc = acc + rowc[j];
#pragma unroll
        for (int j = 0; j < W; j++) rowc[j] = rown[j];
    }
    const double ll_sum = acc;
    const double centre = dpp_f64<0x150 + W>(rawc[W]);
    const double ll_mean = ll_sum / (double)(W * W);
    ok = ok && (ll_mean > 0.0);
    const double p2ll = centre / ll_mean;  // utils.py:230-232
    ok = ok && (p2ll > 0.1);
    // ---- utils.py:221-225 sparsity filter (quotient != 0 <=> count != 0 on a clean matrix)
    int nnz = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
#pragma unroll
        for (int i = 0; i < S; i++) nnz += (jv[s] && colv[s][i] != 0.0) ? 1 : 0;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) nnz += __shfl_xor(nnz, o, 16);
    ok = ok && !((double)nnz < (double)F * 0.1);
    // ---- gaussian_filter(sigma=1), axis 0: down each column, straight into the transpose buffer
    double *Tq = T + wv * REG + q * TS;
#pragma unroll
    for (int s = 0; s < NS; s++) {
#pragma unroll
        for (int i = 0; i < S; i++) {
            const double v = PK_BLUR9(colv[s][i], colv[s][reflect_idx(i - 4, S)], colv[s][reflect_idx(i + 4, S)],
                                      colv[s][reflect_idx(i - 3, S)], colv[s][reflect_idx(i + 3, S)],
                                      colv[s][reflect_idx(i - 2, S)], colv[s][reflect_idx(i + 2, S)],
                                      colv[s][reflect_idx(i - 1, S)], colv[s][reflect_idx(i + 1, S)]);
            if (jv[s]) Tq[i * S + jj[s]] = v;
        }
    }
    // (orders this wave's LDS writes before its reads: every window buffer is its wave's own)
    if constexpr (WAVES == 1) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- axis 1: lane l now owns window ROWS l and l + 16
    double outv[NS][S];
    double mn = __builtin_inf(), mx = -__builtin_inf();
#pragma unroll
    for (int s = 0; s < NS; s++) {
        double rowv[S];
        const double *Tr = Tq + jj[s] * S;  // (row index = the same slot arithmetic as the columns)
#pragma unroll
        for (int j = 0; j < S; j++) rowv[j] = Tr[j];
#pragma unroll
        for (int j = 0; j < S; j++) {
            outv[s][j] = PK_BLUR9(rowv[j], rowv[reflect_idx(j - 4, S)], rowv[reflect_idx(j + 4, S)],
                                  rowv[reflect_idx(j - 3, S)], rowv[reflect_idx(j + 3, S)],
                                  rowv[reflect_idx(j - 2, S)], rowv[reflect_idx(j + 2, S)],
                                  rowv[reflect_idx(j - 1, S)], rowv[reflect_idx(j + 1, S)]);
            if (jv[s]) {
                mn = __builtin_fmin(mn, outv[s][j]);
                mx = __builtin_fmax(mx, outv[s][j]);
            }
        }
    }
    // ---- utils.py:204-209 image_normalize
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        mn = __builtin_fmin(mn, __shfl_xor(mn, o, 16));
        mx = __builtin_fmax(mx, __shfl_xor(mx, o, 16));
    }
    const double den = mx - mn;
    const bool flat = !(den > 0.0);  // constant window: 0 / 0
    const int64_t first = local / blk;
    float *tp = tiles + (size_t)first * F * blk + (int)(local - first * blk);
    double *rp = FEA64 ? fea64_rows + (size_t)local * F : nullptr;
    double r = 0.0, qn = 0.0;
    if (flat) {
        qn = (mn - mn) / den;  // the true division
    } else {
        r = __builtin_amdgcn_rcp(den);
        r = __builtin_fma(r, __builtin_fma(-den, r, 1.0), r);
        r = __builtin_fma(r, __builtin_fma(-den, r, 1.0), r);
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {
#pragma unroll
        for (int j = 0; j < S; j++) {
            const double a = outv[s][j] - mn;
            const double m = a * r;
            const double v = flat ? qn : __builtin_fma(__builtin_fma(-den, m, a), r, m);
            if constexpr (WAVES > 1) {
                // parked for the workgroup's write-out: [feature][4 candidates of this wave] floats over the wave's
                // window buffers (every read of them is behind this wave: the rows were taken above)
                if (jv[s]) reinterpret_cast<float *>(T + wv * REG)[(jj[s] * S + j) * 4 + q] = (float)v;
            } else if (ok && jv[s]) {
                const int f = jj[s] * S + j;
                tp[(size_t)f * blk] = (float)v;  // sklearn's float32 cast (RNE)
                if (FEA64) rp[f] = v;
            }
        }
    }
    if (in_range && l == 0) status[c] = ok ? (flat ? 2 : 1) : 0;
    if constexpr (WAVES > 1) {
        if (l == 0) okf[wv * 4 + q] = ok ? 1 : 0;
        __syncthreads();
        write_out();
    }
}

// ------------------------------------------------------------------------
// Generic kernel (any w <= 15): one candidate per 64-lane wave, window in LDS.
// Used for w = 11 (23x23, 529 features) where a window no longer fits one
// lane's registers.  Same operation order as above.
// ------------------------------------------------------------------------
constexpr int GEN_WAVES = 4;

// scipy 'reflect' at run time; windows narrower than the 9-tap kernel (w < 4) fold
// more than once
__device__ __forceinline__ int reflect1(int q, int n)
{
    while (q < 0 || q >= n) q = q < 0 ? -q - 1 : 2 * n - 1 - q;
    return q;
}

// WT > 0: the half-width is a compile-time constant (w = 11, the 23 x 23 stress
// configuration): every loop unrolls, so the 12 gather loads of a lane are all in flight
// together and the 121 LDS reads of the sequential top-left sum are issued back to back
// instead of one round trip each (measured at w = 11: 6.9 -> see EXPERIMENTS.md 4.1).  WT = 0: any w.
// ANY: coordinates of any kind (pk_extract / getwindow with x > y or off-matrix entries).
// The reference's getwindow (scoreUtils.py:70-93) masks only `x-w >= 0 and y+w+1 <= n`; a
// lower-triangle coordinate that passes reads its window from the stored diagonals
// (-2w < col-row < upper+2w, scoreUtils.py:30-33; below -2w the cells are 0), divides by
// expected(|col-row|) of the UNWRAPPED coordinates, and a window column y-w+j < 0 is the
// column n + (y-w+j) (scipy's negative indices).  Rows beyond the matrix make the reference
// raise; the host refuses such calls before anything is launched.
template <int WT, bool ANY = false>
__global__ __launch_bounds__(64 * GEN_WAVES) void extract_lds_kernel(
    int Wrt, const double *__restrict__ band, int64_t ld, int dlo, int dhi, int n,
    const double *__restrict__ exp_arr, int exp_len, const int32_t *__restrict__ xs,
    const int32_t *__restrict__ ys, int64_t c0, int64_t cn, float *__restrict__ tiles, int blk,
    uint8_t *__restrict__ status, double *__restrict__ fea64_rows)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int W = WT > 0 ? WT : Wrt;
    const int S = 2 * W + 1, F = S * S;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *A = smem + (size_t)wave * 2 * F;
    double *B = A + F;
    const int64_t local = (int64_t)blockIdx.x * GEN_WAVES + wave;
    if (local >= cn) return;  // wave-uniform; no block barriers are used below
    const int64_t c = c0 + local;
    const int xi = xs[c], yi = ys[c];
    if (ANY ? !(xi >= W && yi <= n - W - 1 && xi <= n - W - 1 && yi >= W - n)
            : !(xi - W >= 0 && yi + W + 1 <= n && PK_OTHER_EDGES(xi, yi, W, n))) {
        if (lane == 0) status[c] = 0;
        return;
    }
    const int d = yi - xi;
    const int64_t r0 = (int64_t)(xi - W);
    // lane -> cell mapping without div/mod: two window rows per sweep step,
    // lanes 0..31 take row 2*it, lanes 32..63 row 2*it+1, column = lane & 31
    // (S <= 31).  `steps` sweep steps cover the S rows.
    const int j = lane & 31, ihalf = lane >> 5;
    const int steps = (S + 1) >> 1;
    const bool jok = j < S;

    int nnz = 0;
#pragma unroll
    for (int it = 0; it < steps; it++) {
        const int i = 2 * it + ihalf;
        if (jok && i < S) {
            int k = d + j - i;
            if (ANY && yi - W + j < 0) k += n;  // a negative column counts from the far end
            double v = 0.0;
            if (k >= dlo && k <= dhi) v = band[(int64_t)(k - dlo) * ld + r0 + i];
            v = (v != v) ? 0.0 : v;
            A[i * S + j] = v;
            nnz += (v != 0.0) ? 1 : 0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nnz += __shfl_xor(nnz, o);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): LDS writes of this wave landed
    bool ok = !((double)nnz < (double)F * 0.1);
    // sequential top-left sum, every lane computes the same value from LDS
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < W; i++) {
#pragma unroll
        for (int q = 0; q < W; q++) acc += A[i * S + q];
    }
    const double ll_mean = acc / (double)(W * W);
    ok = ok && (ll_mean > 0.0);
    const double p2ll = A[W * S + W] / ll_mean;
    ok = ok && (p2ll > 0.1);
    if (!ok) {
        if (lane == 0) status[c] = 0;
        return;
    }
    const int dmax = max(iabs(d - 2 * W), iabs(d + 2 * W));
    if (dmax < exp_len) {
    #pragma unroll
    for (int it = 0; it < steps; it++) {
            const int i = 2 * it + ihalf;
            if (jok && i < S) A[i * S + j] = A[i * S + j] / exp_arr[iabs(d + j - i)];
        }
    }
    __builtin_amdgcn_wave_barrier();
    // axis 0: A -> B
#pragma unroll
    for (int it = 0; it < steps; it++) {
        const int i = 2 * it + ihalf;
        if (jok && i < S) {
#define AT(q) A[reflect1((q), S) * S + j]
            B[i * S + j] = PK_BLUR9(A[i * S + j], AT(i - 4), AT(i + 4), AT(i - 3), AT(i + 3),
                                    AT(i - 2), AT(i + 2), AT(i - 1), AT(i + 1));
#undef AT
        }
    }
    __builtin_amdgcn_wave_barrier();
    // axis 1: B -> A
    double mn = __builtin_inf(), mx = -__builtin_inf();
    int has_nan = 0;
#pragma unroll
    for (int it = 0; it < steps; it++) {
        const int i = 2 * it + ihalf;
        if (jok && i < S) {
#define BT(q) B[i * S + reflect1((q), S)]
            const double v = PK_BLUR9(B[i * S + j], BT(j - 4), BT(j + 4), BT(j - 3), BT(j + 3),
                                      BT(j - 2), BT(j + 2), BT(j - 1), BT(j + 1));
#undef BT
            A[i * S + j] = v;
            has_nan |= (v != v) ? 1 : 0;
            mn = (v < mn) ? v : mn;
            mx = (v > mx) ? v : mx;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double omn = __shfl_xor(mn, o), omx = __shfl_xor(mx, o);
        mn = (omn < mn) ? omn : mn;
        mx = (omx > mx) ? omx : mx;
        has_nan |= __shfl_xor(has_nan, o);
    }
    if (has_nan) {
        mn = __builtin_nan("");
        mx = mn;
    }
    const double den = mx - mn;
    const int64_t tile = local / blk;
    const int tl = (int)(local - tile * blk);
    float *tp = tiles + (size_t)tile * F * blk + tl;
    bool fea_nan = false;
#pragma unroll
    for (int it = 0; it < steps; it++) {
        const int i = 2 * it + ihalf;
        if (jok && i < S) {
            const int f = i * S + j;
            const double v = (A[f] - mn) / den;  // each lane reads back only what it wrote
            fea_nan = fea_nan || (v != v);
            tp[(size_t)f * blk] = (float)v;
            if (fea64_rows) fea64_rows[(size_t)local * F + f] = v;
        }
    }
    fea_nan = __any(fea_nan);
    if (lane == 0) status[c] = fea_nan ? 2 : 1;
}

// ---- pre-divided band for the CLEAN extractor ---------------------------------
// norm[kk][r] = count / expected(|k|) (count NaN -> 0 first), the quotient
// distance_normalize computes per window cell (utils.py:180-202).  flags collects
// everything that would make the CLEAN kernel's shortcuts differ from the reference.
__global__ void norm_band_kernel(const double *__restrict__ band, double *__restrict__ norm,
                                 int64_t ld, int ndiag, int dlo, int n,
                                 const double *__restrict__ exp_arr, int exp_len,
                                 int *__restrict__ flags)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kk = blockIdx.y;
    if (blockIdx.x == 0 && kk == 0) {
        // EVERY expected value, not only those of the band's diagonals: a window cell beyond
        // the band (the corner of a candidate at the largest distance reaches |k| = dhi + 1)
        // is 0 / expected in the reference -- NaN if that expected value is 0 or NaN, -0 if it
        // is negative -- where the clean extractor takes a plain +0
        int bad_e = 0;
        for (int i = threadIdx.x; i < exp_len; i += blockDim.x) {
            const double e = exp_arr[i];
            if (!(e > 0.0) || !(e < 1e300)) bad_e = 1;
        }
        if (__any(bad_e) && (threadIdx.x & 63) == 0) atomicOr(flags, 1);
    }
    if (r >= ld || kk >= ndiag) return;
    const int k = kk + dlo;
    const double raw = band[(int64_t)kk * ld + r];
    int bad = 0;
    double q = raw;
    if (raw != raw) bad = 1;                                   // NaN count
    if (__double2hiint(raw) < 0) bad = 1;                      // negative or -0
    if (!(raw < 1e150)) bad = 1;                               // inf / huge
    if (raw != 0.0 && !(raw >= 1e-100)) bad = 1;               // tiny
    const int ak = iabs(k);
    if (ak < exp_len) {
        const double e = exp_arr[ak];
        if (!(e > 0.0) || !(e < 1e300)) bad = 1;
        q = raw / e;
        if (!(q < 1e150) || ((raw != 0.0) != (q != 0.0))) bad = 1;
        if (q != 0.0 && !(q >= 1e-100)) bad = 1;
    }
    norm[(int64_t)kk * ld + r] = q;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flags, 1);
}

}  // namespace

int pk_matrix_prepare_norm(pk_device_ctx *ctx, pk_matrix *m)
{
    if (m->norm_tried) return PK_OK;
    m->norm_tried = true;
    m->clean = false;
    if (!m->norm) return PK_OK;  // pk_matrix_create found the bands too large for 32-bit offsets
    const int ndiag = m->dhi - m->dlo + 1;
    int *d_flags = nullptr;
    PK_HIP(hipMalloc((void **)&d_flags, sizeof(int)));
    PK_HIP(hipMemsetAsync(d_flags, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(norm_band_kernel, dim3((unsigned)((m->ld + 255) / 256), (unsigned)ndiag),
                       dim3(256), 0, ctx->stream, m->band, m->norm, m->ld, ndiag, m->dlo, m->n,
                       m->exp_arr, m->exp_len, d_flags);
    int h_flags = 1;
    const bool ok = hipGetLastError() == hipSuccess &&
                    hipMemcpyAsync(&h_flags, d_flags, sizeof(int), hipMemcpyDeviceToHost,
                                   ctx->stream) == hipSuccess &&
                    hipStreamSynchronize(ctx->stream) == hipSuccess;
    hipFree(d_flags);
    if (!ok) {
        pk_set_error("pk_matrix_prepare_norm: HIP error");
        return PK_E_HIP;
    }
    m->clean = (h_flags == 0);
    return PK_OK;
}

// copies the host's Gaussian taps into the current device's constant memory
int pk_extract_upload_taps(const double *taps5)
{
    PK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pk_gk), taps5, 5 * sizeof(double), 0, hipMemcpyHostToDevice));
    return PK_OK;
}

int pk_launch_extract(pk_device_ctx *ctx, hipStream_t st, const pk_matrix *m, int w,
                      const int32_t *d_x, const int32_t *d_y, int64_t c0, int64_t cn, float *tiles,
                      int blk, uint8_t *d_status, double *fea64_rows, bool any_coords, bool scattered, bool dense)
{
    if (cn <= 0) return PK_OK;
    pk_prof_scope prof(ctx, PK_K_EXTRACT, st);
    if (any_coords) {
        // coordinates outside 0 <= x <= y < n (pk_extract only): the general kernel
        if (w < 1 || w > 15) {
            pk_set_error("pk_extract: w=%d unsupported (1..15)", w);
            return PK_E_UNSUPPORTED;
        }
        const int F = (2 * w + 1) * (2 * w + 1);
        const size_t lds = (size_t)GEN_WAVES * 2 * F * sizeof(double);
        const unsigned grid = (unsigned)((cn + GEN_WAVES - 1) / GEN_WAVES);
        hipLaunchKernelGGL((extract_lds_kernel<0, true>), dim3(grid), dim3(64 * GEN_WAVES), lds, st, w,
                           m->band, m->ld, m->dlo, m->dhi, m->n, m->exp_arr, m->exp_len, d_x, d_y, c0,
                           cn, tiles, blk, d_status, fea64_rows);
    } else if ((w == 5 || w == 6) && m->opt.extract_pair) {
        const unsigned grid = (unsigned)((cn + 31) / 32);
        const int F = (2 * w + 1) * (2 * w + 1);
        // the clean kernel addresses the bands and the tile buffer with 32-bit offsets
        const bool clean = m->norm != nullptr && m->clean && m->opt.extract_clean != 0 &&
                           ((size_t)((cn + blk - 1) / blk) * blk * F * sizeof(float) < (1ull << 31)) &&
                           (blk % 32 == 0) && m->ld < (1 << 20);
        (clean ? g_stat_extract_clean : g_stat_extract_general)++;
        if (clean) {
            const unsigned norm_off = (unsigned)((const char *)m->norm - (const char *)m->band);
#define PK_CLEAN(WW, FF)                                                                         \
    hipLaunchKernelGGL((extract_pair_clean_kernel<WW, FF>), dim3((grid + 7u) & ~7u), dim3(64), 0, st, m->band, \
                       norm_off, (int)m->ld, m->dlo, m->dhi, m->n, m->exp_len, d_x, d_y, c0, cn,   \
                       tiles, blk, d_status, fea64_rows)
            // scattered lists: every lane's loads in the order of its window's diagonals (see extract_pair_clean_body)
#define PK_CLEAN_D(WW)                                                                           \
    hipLaunchKernelGGL((extract_pair_clean_kernel<WW, false, true>), dim3((grid + 7u) & ~7u), dim3(64), 0, st, m->band, \
                       norm_off, (int)m->ld, m->dlo, m->dhi, m->n, m->exp_len, d_x, d_y, c0, cn,   \
                       tiles, blk, d_status, fea64_rows)
            const bool diag = (scattered || m->opt.extract_diag == 2) && !fea64_rows && m->opt.extract_diag != 0;  // (2: always)
            // lists of neighbours (the benchmark regime): the wave's diagonal strip staged in LDS
            if (m->opt.extract_strip == 2 || (m->opt.extract_strip == 1 && dense && !diag)) {
                const unsigned sgrid = (grid + 7u) & ~7u;
                g_stat_extract_strip++;
#define PK_STRIP(WW, FF)                                                                                          \
    hipLaunchKernelGGL((extract_pair_strip_kernel<WW, FF>), dim3(sgrid), dim3(64), 0, st, m->band, norm_off, (int)m->ld, \
                       m->dlo, m->dhi, m->n, m->exp_len, d_x, d_y, c0, cn, tiles, blk, d_status, fea64_rows)
                if (w == 5) {
                    if (fea64_rows) PK_STRIP(5, true);
                    else PK_STRIP(5, false);
                } else {
                    if (fea64_rows) PK_STRIP(6, true);
                    else PK_STRIP(6, false);
                }
#undef PK_STRIP
            } else if (w == 5) {
                if (diag) PK_CLEAN_D(5);
                else if (fea64_rows) PK_CLEAN(5, true);
                else PK_CLEAN(5, false);
            } else {
                if (diag) PK_CLEAN_D(6);
                else if (fea64_rows) PK_CLEAN(6, true);
                else PK_CLEAN(6, false);
            }
#undef PK_CLEAN_D
#undef PK_CLEAN
        } else if (w == 5) {
            hipLaunchKernelGGL(extract_pair_kernel<5>, dim3(grid), dim3(64), 0, st, m->band,
                               m->ld, m->dlo, m->dhi, m->n, m->exp_arr, m->exp_len, d_x, d_y, c0,
                               cn, tiles, blk, d_status, fea64_rows);
        } else {
            hipLaunchKernelGGL(extract_pair_kernel<6>, dim3(grid), dim3(64), 0, st, m->band,
                               m->ld, m->dlo, m->dhi, m->n, m->exp_arr, m->exp_len, d_x, d_y, c0,
                               cn, tiles, blk, d_status, fea64_rows);
        }
    } else if (w == 5 || w == 6) {
        const int threads = 64;
        const unsigned grid = (unsigned)((cn + threads - 1) / threads);
        if (w == 5)
            hipLaunchKernelGGL(extract_reg_kernel<5>, dim3(grid), dim3(threads), 0, st,
                               m->band, m->ld, m->dlo, m->dhi, m->n, m->exp_arr, m->exp_len, d_x,
                               d_y, c0, cn, tiles, blk, d_status, fea64_rows);
        else
            hipLaunchKernelGGL(extract_reg_kernel<6>, dim3(grid), dim3(threads), 0, st,
                               m->band, m->ld, m->dlo, m->dhi, m->n, m->exp_arr, m->exp_len, d_x,
                               d_y, c0, cn, tiles, blk, d_status, fea64_rows);
    } else {
        if (w < 1 || w > 15) {
            pk_set_error("pk_extract: w=%d unsupported (1..15)", w);
            return PK_E_UNSUPPORTED;
        }
        const int F = (2 * w + 1) * (2 * w + 1);
        // w = 11 on a clean matrix: four register-blocked windows per wave (32-bit offsets, as
        // the clean w = 5 / 6 kernel)
        const bool row16 = w == 11 && m->opt.extract_row16 != 0 && m->norm != nullptr && m->clean &&
                           m->opt.extract_clean != 0 && m->ld < (1 << 20) && blk % 4 == 0 &&
                           ((size_t)((cn + blk - 1) / blk) * blk * F * sizeof(float) < (1ull << 33));
        if (row16) {
            g_stat_extract_clean++;
            const unsigned norm_off = (unsigned)((const char *)m->norm - (const char *)m->band);
            const unsigned grid4 = (unsigned)(((cn + 3) / 4 + 7) & ~(int64_t)7);
            const unsigned grid16 = (unsigned)(((cn + 15) / 16 + 7) & ~(int64_t)7);
            if (fea64_rows)
                hipLaunchKernelGGL((extract_row16_clean_kernel<11, true>), dim3(grid4), dim3(64), 0, st, m->band,
                                   norm_off, (int)m->ld, m->dlo, m->dhi, m->n, m->exp_len, d_x, d_y, c0, cn,
                                   tiles, blk, d_status, fea64_rows);
            else if (blk % 16 == 0 && m->opt.extract_row16 != 2)   // (option value 2: the one-wave form)
                hipLaunchKernelGGL((extract_row16_clean_kernel<11, false, 4>), dim3(grid16), dim3(256), 0, st, m->band,
                                   norm_off, (int)m->ld, m->dlo, m->dhi, m->n, m->exp_len, d_x, d_y, c0, cn,
                                   tiles, blk, d_status, fea64_rows);
            else
                hipLaunchKernelGGL((extract_row16_clean_kernel<11, false>), dim3(grid4), dim3(64), 0, st, m->band,
                                   norm_off, (int)m->ld, m->dlo, m->dhi, m->n, m->exp_len, d_x, d_y, c0, cn,
                                   tiles, blk, d_status, fea64_rows);
            PK_HIP(hipGetLastError());
            return PK_OK;
        }
        g_stat_extract_general++;
        const size_t lds = (size_t)GEN_WAVES * 2 * F * sizeof(double);
        const unsigned grid = (unsigned)((cn + GEN_WAVES - 1) / GEN_WAVES);
        if (w == 11)
            hipLaunchKernelGGL(extract_lds_kernel<11>, dim3(grid), dim3(64 * GEN_WAVES), lds, st,
                               w, m->band, m->ld, m->dlo, m->dhi, m->n, m->exp_arr, m->exp_len, d_x,
                               d_y, c0, cn, tiles, blk, d_status, fea64_rows);
        else
            hipLaunchKernelGGL(extract_lds_kernel<0>, dim3(grid), dim3(64 * GEN_WAVES), lds, st,
                               w, m->band, m->ld, m->dlo, m->dhi, m->n, m->exp_arr, m->exp_len, d_x,
                               d_y, c0, cn, tiles, blk, d_status, fea64_rows);
    }
    PK_HIP(hipGetLastError());
    return PK_OK;
}
