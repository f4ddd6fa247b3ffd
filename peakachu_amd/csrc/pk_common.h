// pk_common.h -- internal declarations shared by the HIP sources of
// libpeakachu_hip.so (gfx950 only).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/peakachu_hip.h"

// ------------------------------------------------------------------ errors
void pk_set_error(const char *fmt, ...);

#define PK_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) {                                              \
            pk_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                         __FILE__, __LINE__);                                 \
            return PK_E_HIP;                                                  \
        }                                                                     \
    } while (0)

#define PK_HIP_NULL(call)                                                     \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) {                                              \
            pk_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                         __FILE__, __LINE__);                                 \
            return nullptr;                                                   \
        }                                                                     \
    } while (0)

// ----------------------------------------------------------- device context
// One per device: the stream every kernel of the library is launched on and
// the reusable scratch (feature tiles) sized by the "chunk" option.
struct pk_device_ctx {
    int device = -1;
    hipStream_t stream = nullptr;   // forest, compaction, copies (high priority)
    hipStream_t stream2 = nullptr;  // extractor of the NEXT chunk (low priority), see run_pipeline
    hipEvent_t ev_ext[2] = {nullptr, nullptr};  // extract(k) done, per tile buffer
    hipEvent_t ev_for[2] = {nullptr, nullptr};  // forest(k) done with its tile buffer
    float *fea_tiles = nullptr;   // [tile][F][BLK] float32 feature tiles (two chunk buffers)
    size_t fea_tiles_bytes = 0;
    double *rows64 = nullptr;        // pk_extract: float64 feature rows of one chunk (grow-only)
    size_t rows64_bytes = 0;
    uint16_t *q_tiles = nullptr;  // [tile][F][128] rank codes of the current chunk (forest_q_kernel)
    size_t q_tiles_bytes = 0;
    struct pk_cands *score_cands = nullptr;  // candidate list reused by pk_score (host-buffer calls)
    int64_t score_cands_cap = 0;
    int64_t *scan_scratch = nullptr;  // block counts for the compaction scan
    size_t scan_scratch_bytes = 0;
    int cu_count = 0;
    long long *dbg_buf = nullptr;  // diagnostic cycle stamps (pk_debug_read), 64 Ki entries
    // what a scoring call brings back in ONE copy into pinned memory: {n_out, three status words of
    // dbg_buf} and, for pk_score, the first PK_RET_INLINE scored pixels [x | y | p | signal]
    char *d_ret = nullptr, *h_ret = nullptr;
    // the forest cut in two (pk_forest_q.hip): one counter of parked candidates per launch of a call
    unsigned *split_cnt = nullptr;  // device [PK_SPLIT_SLOTS], zeroed at the first cut launch of a call
    int split_k = 0;                // counters handed out in this call
    int64_t split_n = 0;            // candidates of this call's cut launches
    int64_t split_slack = 0;  // slots the launches' workgroups may have reserved without filling them (half a block each)
};
#define PK_SPLIT_SLOTS 4096
#define PK_RET_INLINE 8192
#define PK_RET_HEAD 64  // {n_out, three status words, candidates the cut forest parked, -, -, -}
#define PK_RET_BYTES (PK_RET_HEAD + (size_t)PK_RET_INLINE * 24)
pk_device_ctx *pk_ctx(int device);  // lazily created; nullptr + error on failure
int pk_ctx_reserve_tiles(pk_device_ctx *, size_t bytes);
int pk_ctx_reserve_scan(pk_device_ctx *, size_t bytes);

// ------------------------------------------------------------------ options
struct pk_options {
    int64_t chunk = 3145728;    // candidates per extract / quantize / forest launch (1.5 GB of float tiles at w = 5).
                                // Measured, ms per step of config 2: round 2 -- 524 288: 8.82, 1 M: 8.66, 2 M: 8.58;
                                // round 5 (cut forest) -- 2 M: 4.27, 2.8 M: 4.21, 3 M: 4.18, 4 M: 4.20 (w = 6: 7.36,
                                // -, 7.29, 10.3: beyond 3.17 M candidates a launch's float tiles pass the 2 GiB of
                                // the clean extractor's 32-bit offsets)
    int64_t forest_ilp = 4;     // L2 kernel: trees walked concurrently per lane
    int64_t forest_slots = 0;   // LDS kernel: tree slots (wave pairs) per workgroup; 0 = as many as
                                // average trees fit beside the tile (8 at w=5, 7 at w=6)
    int64_t forest_lds = 160;   // KiB cap of the LDS tree buffer (0 = read nodes via L2)
    // rank kernels: extract + quantize in pieces of this many candidates, every piece through
    // the START of the float tile buffer (so that it stays in the 256 MB Infinity Cache between
    // the extractor's stores and the quantizer's loads); 0 = the whole chunk at once (default).
    // MEASURED (round 3, config 2, ms per step extract / quantizer): whole chunk 1.21 / 0.95,
    // 512 Ki 1.34 / 1.16, 256 Ki 1.49 / 1.31, 128 Ki 1.69 / 2.21 -- both kernels want launches of
    // >= 1 M candidates (the quantizer loads 24 KB of tables per block), which costs more than
    // the cache residency gives
    int64_t sub_chunk = 0;
    int64_t overlap = 0;        // 1: extract(k+1) on a second stream as soon as its tile buffer is free
                                // (measured: no gain -- kernels that share the chip slow each other down)
    int64_t extract_clean = 1;  // use the pre-divided band + shortcuts when the matrix qualifies
    int64_t extract_pair = 1;   // two lanes per candidate (w = 5, 6); 0 = one lane per candidate
    int64_t extract_diag = 1;   // scattered candidate lists (w = 5, 6, clean matrices): a lane's loads in the order of its
                                // window's diagonals (neighbours on a diagonal share a line of the band)
    int64_t extract_row16 = 1;  // w = 11 on clean matrices: four register-blocked windows per wave
    int64_t extract_strip = 1;  // w = 5, 6, clean matrices: the wave's diagonal strip staged in LDS by LDS-DMA
                                // (extract_pair_strip_kernel): 1 = for lists whose batches are runs on one diagonal
                                // (coords_dense), 2 = for every list, 0 = never (the register-gather kernel)
    int64_t forest_warm = 1;    // last tree group: pull the tile of workgroup id + N into this XCD's L2
                                // (0 = off, 1 = N = number of CUs: the workgroup that follows on this XCD)
    int64_t early_exit = 0;     // pk_score_run: stop walking candidates that provably end at p <= thre
                                // (identical output pixels; per-candidate probabilities of pruned pixels read 0)
    // diagnostics of the rank kernels (pk_forest_q.hip), bits: 2 no walk, 4 no commit stores, 8 every
    // lane takes the left child (2, 4, 8: timing ablations, WRONG results); 16 in-kernel cycle stamps
    // of workgroup 0 (tools/stamps.py); 32 no wave priorities (set by forest_q_prio = 0);
    // 128 forest_q2_kernel without its L2 warm-up
    int64_t forest_dbg = 0;
    int64_t forest_q = 1;       // rank-quantised tiles + 4-byte nodes (forest_q_kernel) when the forest fits
    int64_t forest_q_two = 1;   // 64-candidate shape: two rank tiles per workgroup trip (forest_q2_kernel)
    int64_t forest_q_help = 1;  // forest_q2_kernel: the waves that walk load their share of a group behind the first walk
    int64_t forest_q_rank12 = 1; // rank image: the 12-bit rank word (4 095 thresholds per rank-tile row) when it keeps a
                                 // forest in a larger workgroup shape than 11-bit ranks; 0 never, 2 whenever it saves rows
    int64_t forest_q_ch = 0;    // walks per lane of forest_q_kernel: 0 = auto (4 when F <= 128), 2, 4
    int64_t forest_q_wpt = 0;   // waves per tree with 4 walks per lane: 0 = auto (2), 1, 2
    int64_t forest_q_persist = 1; // rank kernel: persistent launch, this many workgroups per CU, each looping over
                                  // tiles with the next tile prefetched (3.97 -> 3.81 ms); 0 = one workgroup per tile
    int64_t forest_q_prio = 1;  // rank kernel: rotate the waves' issue priorities during the walk (4.05 -> 3.89 ms)
    int64_t forest_q_rsv = 3;   // default shape (256 candidates x 8 trees): forest_qr_kernel -- staging registers
                                // reserved from the compiler, staging loads issued from inside the walk, no
                                // scalar tests on the walk path.  bits: 1 on (0 = forest_q_kernel); 2 the last
                                // group of a tile also fetches the next tile's first group
    int64_t forest_q_early = 0; // two waves per tree: stage the slot's next tree as soon as both are done with it
                                // (measured SLOWER, 4.51 vs 4.09 ms: two waves storing alone get half the LDS
                                // store rate and hold up the walkers' reads; kept as an experiment)
    int64_t forest_img = 1;     // LDS-image forest kernel (fixed-depth walks, absolute LDS addresses)
                                // when every tree fits; 0 = the grouped preorder kernel
    // the forest cut in two at a group boundary with a per-candidate compaction in between (runs that allow
    // the exact early exit only; forest_qr_kernel only): 0 never, 1 where it pays (q_pick_cut)
    int64_t forest_split = 1;
    int64_t forest_split_at = 0;      // > 0: cut in front of this group whatever the threshold (tests, sweeps)
    int64_t forest_split_frac = 200;  // per mille: cut at the first boundary where a partial sum of this share of
                                      // the trees walked is already decided (config 2's background pixels sit at
                                      // p ~ 0.10, 99 % below 0.26)
    int64_t forest_split_min = 262144; // candidates per launch below which two launches cost more than they save
                                       // (config 2 at 0.5: 131 072 candidates +1 %, 262 144 -3 %, 400 000 -5 %, 5.6 M -13 %)
    int64_t compact_small = 1;  // lists of up to 2^14 candidates: batch rule, p > thre, ordered compaction and the
                                // reply in ONE single-workgroup launch (0: the four kernels of long lists)
};
// one recursive lock per device (pk_api.hip, "Locks"); every entry point takes the lock of its handle's device
#define PK_MAX_DEVICES 64
std::recursive_mutex &pk_device_mutex(int device);
#define PK_DEV_LOCK(dev) std::lock_guard<std::recursive_mutex> api_lock__(pk_device_mutex(dev))
pk_options pk_default_options();  // the DEFAULTS of new handles (pk_set_option), copied under their lock; no launch path reads them
// launches of the two-lane extractor since load, by kernel (read-only options
// "stat_extract_clean" / "stat_extract_general": lets tests see which one ran; "stat_extract_strip": launches of
// the LDS-staged variant of the clean kernel)
extern std::atomic<int64_t> g_stat_extract_clean, g_stat_extract_general, g_stat_extract_strip;

// ---------------------------------------------------------------- profiling
// Brackets a kernel launch with HIP events on the library's stream when
// profiling is enabled (pk_prof_enable); otherwise a no-op.
enum pk_kclass { PK_K_EXTRACT = 0, PK_K_FOREST, PK_K_COMPACT, PK_K_BAND, PK_K_QUANT, PK_K_FOREST_TAIL, PK_K_NCLASS };
struct pk_prof_scope {
    pk_prof_scope(pk_device_ctx *ctx, pk_kclass k, hipStream_t st = nullptr);
    ~pk_prof_scope();
    pk_device_ctx *ctx;
    pk_kclass k;
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

// ------------------------------------------------------------------ handles
// Forest node word (8 bytes).  Interior node: {thr32, packed}; stored leaf:
// the float64 class-1 fraction.  A tree's words are in preorder; leaves whose
// value is exactly 0.0 or 1.0 ("pure", >90 % of the leaves of a grown forest)
// get no word at all -- the parent's kind field says which.  So the left
// child, if it has a word, is the next word, and the right child's word is
// `roff` words further (roff = 1 + words of the left subtree).
//   packed bit   0      NaN goes left
//               1..2    kind of the left child
//               3..4    kind of the right child
//               9..18   feature index, i.e. the field read in place is
//                       feature*512 = the byte offset of that feature's row
//                       in a [F][128] float tile
//               19..31  roff (0..8190); 8191 = look roff up in the side
//                       table big_roff[] (trees of more than 8190 words)
//   kind: 0 interior word, 1 stored leaf word, 2 pure leaf 0.0, 3 pure leaf 1.0
// Every tree starts on an even word (16-byte aligned) so groups of trees can
// be copied into LDS with 16-byte accesses.
#define PK_NODE_FEAT_MAX 1024
#define PK_NODE_MISS_BIT 0
#define PK_NODE_LKIND_SHIFT 1
#define PK_NODE_RKIND_SHIFT 3
#define PK_NODE_FEAT_SHIFT 9
#define PK_NODE_FEAT_MASK 0x7fe00u
#define PK_NODE_ROFF_SHIFT 19
#define PK_NODE_ROFF_BIG 8191
#define PK_KIND_NODE 0u
#define PK_KIND_LEAF 1u
#define PK_KIND_ZERO 2u
#define PK_KIND_ONE 3u

struct pk_forest {
    pk_options opt = pk_default_options();  // this handle's options: the process defaults (pk_set_option) at its creation,
                             // then whatever pk_forest_set_option changed -- another handle never sees it
    int device;
    int T, F;
    int64_t n_nodes;       // slots in `nodes`
    int max_depth;         // edges on the longest root->leaf path
    int max_tree_words;
    uint2 *nodes;          // device, n_nodes x 8 B
    int32_t *root;         // device, T+1 offsets of each tree's root
    int32_t *big_roff;     // device side table [n_nodes] or nullptr
    std::vector<int32_t> h_root;
    std::vector<uint8_t> h_big;   // per tree: uses the side table (never staged in LDS)
    // LDS-kernel tree groups for one (tree_words, slots) launch shape, cached
    int32_t *grp;          // device, n_grp+1 first-tree indices, then n_grp staged flags
    int n_grp;
    int grp_words, grp_slots;
    // sklearn's arrays as handed to pk_forest_create (the LDS image is built from them)
    std::vector<int32_t> h_tree_off, h_left, h_right, h_feat;
    std::vector<double> h_thr, h_p1;
    std::vector<uint8_t> h_miss;
    // LDS image (forest_img_kernel): 0 = not tried, 1 = built, -1 = does not apply
    int img_state = 0;
    int img_slots = 0, img_n_grp = 0;
    int64_t img_opt_slots = -1;  // value of the forest_slots option the image was planned for
    struct pk_img_layout *img_layout = nullptr;
    uint4 *img = nullptr;          // device
    int32_t *img_gtab = nullptr;   // device
    uint2 *img_troot = nullptr;    // device
    int32_t *img_tdepth = nullptr; // device
    // how pk_forest_plan_blk decided to evaluate this forest: 0 preorder kernels, 1 LDS image, 2 rank image
    int plan_kind = 0;
    // the kernel family that walked this forest last (read-only option "stat_family"):
    // 1 forest_qr_kernel (rank image, 256 candidates, the default for <= 255 features), 2 forest_q_kernel
    // (rank image, generic), 3 forest_q2_kernel (rank image, two 64-candidate tiles: wide forests),
    // 4 forest_img_kernel (8-byte LDS image), 6 forest_lds_kernel, 7 forest_gmem_kernel (preorder 8-byte nodes);
    // 0 none yet (5 and 8 were the pipe and L2 kernels of rounds 1-4, removed)
    int last_family = 0;
    // rank image (forest_q_kernel): 0 = not tried, 1 = built, -1 = does not apply
    int q_state = 0;
    int q_slots = 0, q_ch = 0, q_n_grp = 0;
    int q_T = 0;           // trees of the rank image: T, or more when trees were cut into pieces (pk_qimage.hip)
    int q_mode = 0;        // node word format of the rank image (PK_Q_NARROW / _WIDE / _NARROW12): codes are r << 5,
                           // r << 5, r << 4
    int q_F = 0;           // rows of a rank tile: F + the virtual features (pk_q_tables)
    int32_t *q_src = nullptr;  // device [q_F]: the float feature a row is quantized from
    int q_slot_bytes = 0;  // > 0: fixed tree slots, early staging
    int64_t q_opt_slots = -1, q_opt_ch = -1, q_opt_wpt = -1, q_opt_early = -1, q_opt_rank12 = -1;
    int q_max_group_bytes = 0;     // the largest tree group of the rank image
    struct pk_q_layout *q_layout = nullptr;
    uint4 *q_img = nullptr;        // device: tree images
    int32_t *q_gtab = nullptr, *q_ttab = nullptr, *q_off = nullptr;  // device
    float *q_thr = nullptr, *q_par = nullptr;                        // device
    uint32_t *q_lut = nullptr;                                       // device
    std::vector<int32_t> q_gtab_h;  // host copy of q_gtab: where a cut may go (pk_forest_q.hip, q_pick_cut)
    int last_cut = 0;               // group the last launch of forest_qr_kernel was cut in front of (0: one launch)
    // what the cut has learnt for one threshold (q_pick_cut, pk_forest_cut_feedback)
    double cut_sum = -1.0;          // thre * T it belongs to
    int cut_shift = 0;              // groups later than the first plausible boundary
    bool cut_off = false;           // no cut pays at this threshold
};
// ---- LDS-image forest (pk_image.hip builds it, pk_forest_img.hip walks it) ----
struct pk_img_layout {
    int F, slots;
    int HB;        // bytes of half a feature tile: [F][64] float32
    int lenA;      // bytes of image region A = [HB, 65536) (0 if too small to use)
    int B0;        // LDS byte offset of image region B = [B0, 163840)
    int val_off;   // [slots][128] float64 leaf values parked for the ordered sum
    int dec_off;   // early-termination flags: 128 ints + 3 vote words
    int cap;       // bytes a group image may occupy
};
struct pk_img_out {
    std::vector<uint2> words;      // images of all groups, each a whole number of 16-byte units
    std::vector<int32_t> gtab;     // per group: first tree, trees, offset and size in 16-byte units
    std::vector<uint2> troot;      // per tree: the word a walk starts from
    std::vector<int32_t> tdepth;   // per tree: levels to descend (depth of the tree)
    int n_grp = 0;
};
// staging registers (uint4) per thread of forest_img_kernel<slots>, and what they move
inline int pk_img_stage_regs(int slots)
{
    return slots >= 8 ? 6 : slots == 7 ? 7 : slots == 6 ? 8 : slots == 5 ? 10 : 12;
}
inline int pk_img_stage_bytes(int slots) { return pk_img_stage_regs(slots) * 128 * slots * 16; }
bool pk_img_make_layout(int F, int slots, int max_image_bytes, pk_img_layout *L);
int pk_img_build(int T, int F, const int32_t *tree_off, const int32_t *left, const int32_t *right,
                 const int32_t *feat, const double *thr, const uint8_t *miss, const double *p1,
                 const pk_img_layout &L, pk_img_out *out);
// decides how `f` is evaluated for tile-shaped input: builds (once) the LDS image when the
// image kernel applies; returns the tile width the extractor must produce for it
int pk_forest_plan_blk(pk_forest *f);
void pk_forest_img_release(pk_forest *f);
int pk_launch_forest_img(pk_device_ctx *, pk_forest *f, const float *tiles, const uint8_t *d_status,
                         int64_t c0, int64_t cn, double *d_prob, double prune_sum);

// ---- rank image (pk_qimage.hip builds it, pk_forest_q.hip quantizes tiles and walks it) ----
#define PK_Q_CELLS 4096
#define PK_Q_MAX_RANK 2047  // thresholds per rank-tile row (11-bit rank field); a feature with more is split
#define PK_Q_MAX_RANK12 4095  // ... of the 12-bit rank field (PK_Q_NARROW12)
enum { PK_Q_NARROW = 0, PK_Q_WIDE = 1, PK_Q_NARROW12 = 2 };  // node word formats (pk_qimage.hip)
#define PK_Q_FTILE 8   // 128-candidate tiles per float32 tile handed to the quantizer
struct pk_q_layout {
    int F, slots, ch;   // ch = 64-candidate blocks per workgroup: 2 (128 candidates), 4 (256), or 1 (64
                        // candidates, 64-candidate rank tiles and the wide node word: up to 1023 features)
    int HB;             // bytes of a rank tile: [F][128] u16 (ch = 1: [F][64] u16)
    int half1;          // LDS offset of the second half tile (ch == 4), 0 otherwise
    int dec_off;        // early-termination flags
    int val_off;        // [slots][64*ch] float64 leaf values parked for the ordered sum
    int img_off, cap;   // the group's trees: [img_off, img_off + cap)
    int slot_bytes;     // > 0: fixed tree slots (early staging): tree s of a group lives at
    int slot_off[17];   //      img_off + slot_off[s]; slot_bytes = slot_off[slots] = their total size
};
struct pk_q_out {
    int Fq = 0;                   // rows of the rank tile: the features + the virtual ones (pk_q_tables)
    int max_rank = PK_Q_MAX_RANK; // thresholds per row the tables were made for
    int mode = PK_Q_NARROW;       // node word format of `pairs` (pk_q_trees)
    std::vector<int32_t> qsrc;    // [Fq] the float feature a row is quantized from
    std::vector<int32_t> qfirst;  // [F] first virtual row of a feature with more than PK_Q_MAX_RANK thresholds, or -1
    std::vector<float> qthr;      // per row: sorted distinct float32 thresholds, laid end to end
    std::vector<int32_t> qoff;    // Fq+1 offsets into qthr
    std::vector<uint32_t> qlut;   // [F][PK_Q_CELLS]: thresholds below the cell | thresholds in it << 16
    std::vector<float> qpar;      // [F][2]: lower end of the cells, cells per unit
    std::vector<uint2> pairs;     // tree images (8-byte child pairs), tree after tree
    std::vector<uint32_t> troot;  // per tree: the word a walk starts from
    std::vector<int32_t> tdepth;  // per tree: levels to descend
    std::vector<int32_t> gtab;    // per group: first tree, trees, offset and size in 16-byte units
    std::vector<int32_t> ttab;    // per tree: byte offset inside its group, depth, root word, 0
    std::vector<int32_t> toff;    // T+1 offsets of the trees in `pairs` (8-byte units)
    int n_grp = 0;
};
// lookup cell of a feature value: the SAME float operations on the host (tables) and on
// the device (quantizer); monotone in x, which is all the tables rely on
__host__ __device__ inline int pk_q_cell(float x, float lo, float inv)
{
    float cf = (x - lo) * inv;  // +-inf, or NaN from inf * 0: fmaxf / fminf return the other operand
    cf = fminf(fmaxf(cf, 0.f), (float)(PK_Q_CELLS - 1));
    return (int)cf;
}
// rebuilds qlut from the cell of every threshold (qcell, parallel to qthr)
void pk_q_fill_lut(pk_q_out *out, int F, const std::vector<int32_t> &qcell);
#define PK_Q_PAD_BYTES 131072  // readable bytes behind the rank image and behind the rank tiles (forest_qr_kernel)
inline int pk_q_stage_regs() { return 8; }  // uint4 staging registers per thread (1024) of forest_q_kernel
bool pk_q_make_layout(int F, int slots, int ch, pk_q_layout *L);
int pk_q_tables(int T, int F, const int32_t *tree_off, const int32_t *left, const int32_t *feat,
                const double *thr, int max_rank, pk_q_out *out);
int pk_q_trees(int T, int F, const int32_t *tree_off, const int32_t *left, const int32_t *right,
               const int32_t *feat, const double *thr, const uint8_t *miss, const double *p1, int mode,
               pk_q_out *out);
int pk_q_build(int T, int F, const int32_t *tree_off, const int32_t *left, const int32_t *right,
               const int32_t *feat, const double *thr, const uint8_t *miss, const double *p1,
               const pk_q_layout &L, pk_q_out *out);
int pk_q_group(pk_q_out *out, const pk_q_layout &L);
// fixed tree slots for groups of exactly `slots` consecutive trees: slot s is as large as the
// largest tree in position s of any group; fills L->slot_off / slot_bytes, returns their sum
int pk_q_fixed_slots(const pk_q_out &out, int slots, pk_q_layout *L);
int pk_q_max_tree_bytes(const pk_q_out &out);  // largest tree image, a multiple of 16  // (re)group the trees of `out` for a layout
int pk_forest_q_plan(pk_forest *f);   // PK_OK when the rank image applies (built and uploaded)
void pk_forest_cut_feedback(pk_forest *f, int64_t candidates, int64_t parked, int64_t slack);  // after a call whose launches were cut
void pk_forest_q_release(pk_forest *f);
// split_sum: -inf, or thre * T when the run allows decided candidates to end at probability 0 and the
// float tiles may be overwritten once they are quantized (the cut forest parks its candidates there)
int pk_launch_forest_q(pk_device_ctx *, pk_forest *f, const float *tiles, const uint8_t *d_status,
                       int64_t c0, int64_t cn, double *d_prob, double prune_sum, double split_sum = -INFINITY);
// the same in three steps (run_pipeline quantizes sub-chunk by sub-chunk from a cache-resident
// float buffer and walks the whole chunk at once)
int pk_forest_q_reserve(pk_device_ctx *, pk_forest *f, int64_t cn);
int pk_launch_quant_q(pk_device_ctx *, hipStream_t st, pk_forest *f, const float *tiles, int64_t t0, int64_t cn);
int pk_launch_forest_q_walk(pk_device_ctx *, pk_forest *f, const uint8_t *d_status, int64_t c0, int64_t cn,
                            double *d_prob, double prune_sum, double split_sum = -INFINITY, void *scratch = nullptr,
                            size_t scratch_bytes = 0);

// (re)build f->grp for this launch shape; returns PK_OK or an error code
int pk_forest_groups(pk_forest *f, int tree_words, int slots);
// per-tree flag: words <= region_words and no side-table offsets

// Diagonal-major dense band: cell (r, r+k), dlo <= k <= dhi, lives at
// band[(k - dlo) * ld + r]; everything else reads 0.
struct pk_matrix {
    pk_options opt = pk_default_options();  // (see pk_forest::opt) extractor options; pipeline options of the calls that
                             // have no candidate handle (pk_score, pk_extract)
    int device;
    int32_t n, dlo, dhi;
    int64_t ld;            // row pitch in doubles (n rounded up to 64)
    double *band;          // device
    double *exp_arr;       // device
    int32_t exp_len;
    // the band divided by the expected values, built on first use for the two-lane
    // extractor when the matrix passes its checks (see norm_band_kernel)
    double *norm = nullptr;
    bool norm_tried = false;
    bool clean = false;
};

// a contact matrix uploaded once (CSR); bands, validity flags and value facts are made
// from it on the device
// the stored entries of an uploaded matrix; shared by a pk_csr and the views made of it
struct pk_pixels {
    int32_t *indptr, *indices;   // device
    double *data;                // device
    int refs;                    // handles that point here (under the device lock)
};

struct pk_csr {
    int device;
    int32_t n;
    int64_t nnz;
    int32_t *indptr, *indices;   // device (= pix->...)
    double *data;                // device
    uint8_t *valid_raw, *valid_bal;  // device [n], see csr_info_kernel
    unsigned long long info[6];  // host copy: finite, non-finite, non-integer, negative, max bits, entries out of order
    pk_pixels *pix;              // owner of indptr / indices / data
    bool upper;                  // the arrays hold the upper triangle of a symmetric matrix (a .cool's pixel table)
    double *bias;                // device [n] or nullptr: entry values are (bias[row] * bias[col]) * data
};

struct pk_cands {
    pk_options opt = pk_default_options();  // (see pk_forest::opt) pipeline options of pk_score_run: chunk, overlap, sub_chunk,
                             // early_exit
    int device;
    int64_t N;
    int32_t *x, *y;        // device, candidate coordinates
    double *prob;          // device [N]
    uint8_t *status;       // device [N] 0 = filtered out, 1 = window survived, 2 = survived
                           // and its features are NaN (constant window)
    // compacted outputs of the last run (device), capacity N
    int32_t *ox, *oy;
    double *op, *osig;
    int64_t *n_out_dev;    // device scalar
    bool ret_inline;       // the scored pixels of the last run lie in the context's h_ret (pk_score)
    int64_t n_out;         // host copy after the run
    int32_t *batch_cnt;    // device [n_batches] survivors per reference batch
    int64_t n_batches_cap;
    int prune;             // pk_cands_set_prune: exact early termination for this list's runs
    int scattered;         // 1: consecutive candidates are rarely neighbours on a diagonal (get_candidate's lists)
    int dense;             // 1: the batches of 32 consecutive candidates are runs on one diagonal (coords_dense)
    // pk_score (host buffers): coordinates still on the host; run_pipeline uploads chunk k + 1 on the
    // second stream while chunk k is being scored (nullptr: everything is on the device already)
    const int32_t *h_x = nullptr, *h_y = nullptr;
};

// ------------------------------------------------------------ kernel entry
// (implemented in the .hip files; all launch on ctx->stream)
int pk_launch_band_build(pk_device_ctx *, pk_matrix *, const int32_t *d_indptr,
                         const int32_t *d_indices, const double *d_data, int64_t nnz, int filter,
                         const double *d_bias = nullptr, int upper = 0);
int pk_launch_csr_info(pk_device_ctx *, const int32_t *d_indptr, const int32_t *d_indices,
                       const double *d_data, int64_t nnz, int n, unsigned long long *d_info6,
                       uint8_t *d_valid_raw, uint8_t *d_valid_bal, const double *d_bias = nullptr, int upper = 0);
int pk_launch_counts_to_f64(pk_device_ctx *, const int32_t *d_src, double *d_dst, int64_t n);

// features of candidates [c0, c0+cn) -> tiles (tile width BLK) + status.
// If fea64_rows != nullptr also writes row-major float64 features [cn][F].
int pk_matrix_prepare_norm(pk_device_ctx *, pk_matrix *);
int pk_extract_upload_taps(const double *taps5);  // into the current device's constant memory
int pk_launch_extract(pk_device_ctx *, hipStream_t st, const pk_matrix *, int w,
                      const int32_t *d_x, const int32_t *d_y, int64_t c0, int64_t cn, float *tiles,
                      int blk, uint8_t *d_status, double *fea64_rows, bool any_coords = false,
                      bool scattered = false,   // scattered: consecutive candidates are rarely neighbours
                      bool dense = false);      // dense: a wave's 32 candidates are a run on one diagonal

// The early-exit / cut decision of every forest kernel is `fl(acc + rem) < B`: acc = the candidate's
// partial sum (bit for bit the reference's: same trees, same order), rem = the trees (or tree pieces)
// still to come, each of which adds a value in [0, 1], B = this bound.  Claim: whoever passes the test
// ends at fl(S / T) <= thre, i.e. is not reported by `p > thre` (peakachu/scoreUtils.py:110).  Proof, with
// u = 2^-53, n = `additions` >= rem, M = 1 + (n + 4) * 2^-52 (exact in binary64 for n < 2^51):
//   * round-to-nearest addition is monotone and fl(a + b) <= (a + b)(1 + u), so the final sequential
//     sum S satisfies S <= (acc + rem)(1 + u)^rem (the real number acc + rem, every later term <= 1);
//   * the test's own addition gives fl(acc + rem) >= (acc + rem)(1 - u);
//   * B <= fl(thre * T) / M <= thre * T * (1 + u) / M (the quotient is rounded DOWN below);
//   hence S < thre * T * (1 + u)^(rem + 1) / (M (1 - u)) <= thre * T, because (1 + u)^(n + 1) <= 1 + (n + 2) u
//   <= M (1 - u) for every n < 2^26 (larger: no bound, nobody is decided); so S / T < thre as real numbers and, thre being a binary64 number and
//   division being monotone, fl(S / T) <= thre.  tests/test_prune_bound.py checks the inequality in exact
//   rational arithmetic for the forest sizes the library accepts and drives adversarial sums through it.
// (A fixed 1e-12, as in rounds 1-5, states this only up to ~9 000 additions.)
static inline double pk_prune_bound(double thre, int T, int64_t additions)
{
    if (additions < T) additions = T;
    if (additions >= ((int64_t)1 << 26)) return -INFINITY;
    const double M = 1.0 + (double)(additions + 4) * 0x1p-52;
    return nextafter(thre * (double)T / M, -INFINITY);
}

// walk the forest over feature tiles of candidates [c0, c0+cn)
// prune_sum: -inf (full evaluation) or pk_prune_bound(thre, T, ..): candidates whose sum provably cannot reach it
// are dropped early (grouped LDS kernel only; their reported probability is 0)
int pk_launch_forest(pk_device_ctx *, pk_forest *, const float *tiles, int blk,
                     const uint8_t *d_status, int64_t c0, int64_t cn, double *d_prob,
                     double prune_sum, double split_sum = -INFINITY);
// row-major float32 features [N][F] -> tiles (for pk_predict); status[i] = 1,
// or 2 if row i holds a NaN
int pk_launch_tile_rows(pk_device_ctx *, const float *d_rows, int64_t N, int F, float *tiles,
                        int blk, uint8_t *d_status);
int pk_forest_tile_width(int F, const pk_options &o);  // candidates per feature tile for F features

// with_records / reply_packed: a short list's compaction also packs the call's reply (ctx->d_ret: header
// and, with_records, the inline pixels) and says so; nullptr = the caller packs it itself
int pk_launch_compact(pk_device_ctx *, const pk_matrix *, pk_cands *, double thre,
                      int64_t batch, int with_records = 0, bool *reply_packed = nullptr);
int pk_launch_expected_means(pk_device_ctx *, const pk_matrix *m, int first, int top,
                             const uint8_t *d_valid, double *d_scratch, double *d_means);
int pk_launch_candidates(pk_device_ctx *, const pk_matrix *raw, int lower, int upper,
                         const int64_t *d_kstar, const double *d_bg, const double *d_w,
                         const double *d_mustar, int64_t n_mustar, int64_t *d_total,
                         int64_t *d_amb, int32_t *ox, int32_t *oy);
