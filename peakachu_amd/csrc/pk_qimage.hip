// pk_qimage.hip -- host-side builder of the forest's RANK image (gfx950 only).
//
// Serves model.predict_proba(fea)[:, 1] at peakachu/scoreUtils.py:109 (sklearn
// Tree._apply_dense: `x[feature] <= threshold` goes left, NaN goes where
// missing_go_to_left says).  The decision `(double)x <= thr` of a float32
// feature x depends only on how many of the forest's distinct thresholds of
// that feature lie below x:
//     t32  = largest float32 <= thr                  ((double)x <= thr  <=>  x <= t32)
//     k    = index of t32 among the feature's sorted distinct t32 values
//     r(x) = number of those values that are < x      (0 .. n_f)
//     x <= t32  <=>  r(x) <= k
// so the forest kernel can work on 16-bit rank codes instead of float32
// features (half the LDS tile) and on 4-BYTE nodes (half the LDS image, and both
// children of a node arrive in one 8-byte LDS read): forest_q_kernel in
// pk_forest_q.hip.  The codes are made by quantize_tiles_kernel from the float32
// tiles with the tables built here; exact, not an approximation.
//
// Node word (32 bits):  [31:21] rank k   [20] NaN goes left   [19:8] pair   [7:0] feature
//   (more than 255 features: the WIDE word, [31:21] rank | [20:10] pair | [9:0] feature, where
//   "NaN goes left" is WHERE THE PAIR LIES: at or beyond the tree's `split` -- see q_emit_tree)
//   pair = index (8-byte units, relative to the tree's first byte) of the node's
//   child pair: the left child's word, then the right child's word.
//   A feature code is r(x) << 5 (NaN: 0xFFFF), so `code <= (word >> 16)` is
//   `r(x) <= k` whatever bits 20..16 hold.
//   (round 4, more than 2 047 thresholds on features of a <= 255-row forest: the RANK-12 narrow word
//   [31:20] rank k (12 bits) | [19:8] pair | [7:0] feature with codes r(x) << 4 and the wide word's NaN
//   rule -- a forest fitted on 90 000 windows has 2 154 thresholds per feature: 121 rows instead of
//   220, so it keeps the 256-candidate kernel)
// Leaf word: rank field 0x7FF (every code compares <=: a leaf always goes left),
//   bit 20 set (NaN goes left too), pair = the leaf's own block [leaf word][0][float64
//   value]: a walk that has reached a leaf stays on it however many more levels it is
//   asked to descend; the kernel walks every tree for a fixed number of levels (its depth)
//   and reads the value at pair + 1 at the end.
// Per tree: blocks of the two pure leaves (0.0 / 1.0), the four pairs of two pure
// leaves, the child pairs of the interior nodes, the blocks of the other leaf values.
// Trees are position independent (a multiple of 16 bytes each); a group of
// consecutive trees is a contiguous slice of the image.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <map>

#include "pk_common.h"

namespace {

constexpr uint32_t Q_LEAF = (0x7FFu << 21) | (1u << 20);
constexpr int Q_CONST_PAIRS = 8;  // block(0.0), block(1.0), pairs (0,1) (1,0) (0,0) (1,1)

float q_floor32(double t)
{
    float f = (float)t;
    if ((double)f > t) f = nextafterf(f, -INFINITY);
    return f;
}

struct q_tree {
    int nn;
    const int32_t *left, *right, *feat;
    const double *thr, *p1;
    const uint8_t *miss;
    int kind(int v) const  // 0 interior, 1 stored leaf, 2 pure 0.0, 3 pure 1.0
    {
        if (left[v] != -1) return 0;
        uint64_t b;
        memcpy(&b, &p1[v], 8);
        if (b == 0) return 2;
        if (p1[v] == 1.0) return 3;
        return 1;
    }
};

// lays one tree out; pairs are appended to `out` (nullptr: only count).  Returns the
// number of pairs, or -1 (malformed) / -2 (does not fit the pair field).
// Word formats (rank in bits 31..21 either way, so `code <= word >> 16` decides a split):
//   narrow (<= 255 features): [20] NaN goes left | [19:8] pair index | [7:0] feature
//   wide   (<= 1023 features): [20:10] pair index (trees of <= 2047 pairs) | [9:0] feature.
//          All 32 bits are taken, and forests fitted by scikit-learn >= 1.3 carry a
//          missing_go_to_left flag on every node: the flag is WHERE the node's child pair lies --
//          the pairs of the nodes that send NaN right come first (level by level), then, from pair
//          index `split` on, those of the nodes that send it left (level by level), then the leaf
//          blocks; `NaN goes left  <=>  pair index >= split`, a compare that only the walk of a wave
//          holding NaN codes makes (round 4; rounds 2-3 sent such forests to the float kernels: 44 ms
//          instead of 11 for the fitted 500-tree forest of configs[4]).  The four pairs of two pure
//          leaves exist on either side of the split.  *split_out receives it (0 for the narrow word).
// mode: PK_Q_NARROW (11-bit rank, NaN flag in bit 20), PK_Q_WIDE, PK_Q_NARROW12 (12-bit rank at [31:20], pair
// and feature where the narrow word has them, NaN by the pair's side of `split` like the wide word)
int q_emit_tree(const q_tree &t, int F, const pk_q_out &tab, std::vector<uint2> *out, uint32_t *root,
                int *depth, std::string *err, int mode, int *split_out = nullptr)
{
    const bool wide = mode == PK_Q_WIDE;
    const bool two_sided = mode != PK_Q_NARROW;   // "NaN goes left" = the pair lies at or beyond `split`
    const int pair_shift = wide ? 10 : 8;
    const int max_pairs = wide ? 2048 : 4096;
    const int rank_shift = mode == PK_Q_NARROW12 ? 20 : 21;
    const uint32_t rank_leaf = mode == PK_Q_NARROW12 ? (0xFFFu << 20) : (0x7FFu << 21);
    const int const_pairs = Q_CONST_PAIRS;
    const int nn = t.nn;
    std::vector<int> order, dep((size_t)nn, 0), stack;
    std::vector<uint8_t> seen((size_t)nn, 0);
    int maxd = 0;
    stack.push_back(0);
    while (!stack.empty()) {
        const int v = stack.back();
        stack.pop_back();
        if (v < 0 || v >= nn || seen[(size_t)v]) {
            if (err) *err = "malformed tree (cycle or child out of range)";
            return -1;
        }
        seen[(size_t)v] = 1;
        if (dep[(size_t)v] > maxd) maxd = dep[(size_t)v];
        if (t.left[v] == -1) continue;
        const int l = t.left[v], r = t.right[v];
        if (l < 0 || l >= nn || r < 0 || r >= nn || t.feat[v] < 0 || t.feat[v] >= F) {
            if (err) *err = "malformed tree (bad child or feature index)";
            return -1;
        }
        order.push_back(v);
        dep[(size_t)l] = dep[(size_t)r] = dep[(size_t)v] + 1;
        stack.push_back(r);
        stack.push_back(l);
    }
    *depth = maxd;
    // pair indices: 0,1 block(0.0); 2,3 block(1.0); 4..7 pure/pure pairs; then node pairs;
    // then the blocks of the stored leaves (one per distinct value).
    // Node pairs are numbered LEVEL BY LEVEL: all walks of a wave are at the same depth at
    // the same time, so the pairs they read together are the pairs of one level, and
    // consecutive pairs lie in different LDS banks (a ds_read_b64 of 32 lanes is
    // conflict-free over 32 consecutive pairs; measured: a third of the kernel time is
    // bank conflicts when pairs are numbered in preorder).
    int next = const_pairs;
    std::vector<int> pairi((size_t)nn, -1);
    int split = 0;  // wide: first pair index of the "NaN goes left" side
    std::vector<int> by_level(order);
    std::stable_sort(by_level.begin(), by_level.end(),
                     [&](int a, int b) { return dep[(size_t)a] < dep[(size_t)b]; });
    for (int side = 0; side < (two_sided ? 2 : 1); side++) {
        if (side == 1) {
            split = next;   // the second set of pure/pure pairs opens the "NaN goes left" side
            next += 4;
        }
        for (int n : by_level) {
            const int want = (two_sided && t.miss && t.miss[n]) ? 1 : 0;
            if (two_sided && want != side) continue;
            const bool pl = t.kind(t.left[n]) >= 2, pr = t.kind(t.right[n]) >= 2;
            if (pl && pr) {
                const int vl = t.kind(t.left[n]) - 2, vr = t.kind(t.right[n]) - 2;
                const int combo = vl == 0 && vr == 1 ? 0 : vl == 1 && vr == 0 ? 1 : vl == 0 ? 2 : 3;
                pairi[(size_t)n] = (side ? split : 4) + combo;
            } else {
                pairi[(size_t)n] = next++;
            }
        }
    }
    if (split_out) *split_out = split;
    std::map<uint64_t, int> vblock;  // stored leaf value bits -> pair index of its block
    std::vector<std::pair<int, double>> blocks;
    auto leaf_block = [&](int leaf) {
        const int k = t.kind(leaf);
        if (k == 2) return 0;
        if (k == 3) return 2;
        uint64_t b;
        memcpy(&b, &t.p1[leaf], 8);
        auto it = vblock.find(b);
        if (it != vblock.end()) return it->second;
        const int p = next;
        next += 2;
        vblock[b] = p;
        blocks.push_back({p, t.p1[leaf]});
        return p;
    };
    std::vector<int> lblock((size_t)nn, -1);
    for (int v = 0; v < nn; v++)
        if (seen[(size_t)v] && t.left[v] == -1) lblock[(size_t)v] = leaf_block(v);
    if (next & 1) next++;  // whole 16-byte units
    if (next > max_pairs) return -2;
    auto leaf_word = [&](int block) {
        return (two_sided ? rank_leaf : Q_LEAF) | ((uint32_t)block << pair_shift);
    };
    auto word_of = [&](int v) -> uint32_t {
        if (t.left[v] == -1) return leaf_word(lblock[(size_t)v]);
        int f = t.feat[v];
        const float t32 = q_floor32(t.thr[v]);
        // the part of the feature's threshold list that holds t32: the feature's own row, or one of
        // its virtual features (their rows follow each other from qfirst[f] on; lists are ascending)
        for (int part = tab.qfirst[(size_t)f]; part >= 0 && part < tab.Fq && tab.qsrc[(size_t)part] == t.feat[v] &&
                                               t32 >= tab.qthr[(size_t)tab.qoff[(size_t)part]]; part++)
            f = part;
        const float *b = tab.qthr.data() + tab.qoff[(size_t)f], *e = tab.qthr.data() + tab.qoff[(size_t)f + 1];
        const int k = (int)(std::lower_bound(b, e, t32) - b);  // t32 is in the table by construction
        uint32_t w = ((uint32_t)k << rank_shift) | ((uint32_t)pairi[(size_t)v] << pair_shift) | (uint32_t)f;
        if (!two_sided && t.miss && t.miss[v]) w |= 1u << 20;
        return w;
    };
    *root = word_of(0);
    if (!out) return next;
    const size_t base = out->size();
    out->resize(base + (size_t)next, make_uint2(0, 0));
    uint2 *P = out->data() + base;
    auto put_block = [&](int p, double v) {
        uint64_t b;
        memcpy(&b, &v, 8);
        // (wide word: a leaf's block index carries no flag, so both ways lead back to the leaf)
        P[p] = make_uint2(leaf_word(p), two_sided ? leaf_word(p) : 0);
        P[p + 1] = make_uint2((uint32_t)(b & 0xffffffffu), (uint32_t)(b >> 32));
    };
    put_block(0, 0.0);
    put_block(2, 1.0);
    {
        const uint2 combos[4] = {make_uint2(leaf_word(0), leaf_word(2)), make_uint2(leaf_word(2), leaf_word(0)),
                                 make_uint2(leaf_word(0), leaf_word(0)), make_uint2(leaf_word(2), leaf_word(2))};
        for (int c = 0; c < 4; c++) {
            P[4 + c] = combos[c];
            if (two_sided) P[split + c] = combos[c];
        }
    }
    for (auto &bk : blocks) put_block(bk.first, bk.second);
    for (int n : order)
        if (pairi[(size_t)n] >= const_pairs && !(two_sided && pairi[(size_t)n] >= split && pairi[(size_t)n] < split + 4))
            P[pairi[(size_t)n]] = make_uint2(word_of(t.left[n]), word_of(t.right[n]));
    return next;
}

}  // namespace

// LDS map of forest_q_kernel<slots, ch>:
//   [0, HB)                  rank tile of candidates 0..127   ([F][128] u16)
//   [half1, half1 + HB)      rank tile of candidates 128..255 (ch == 4 only): half1 = 32 KiB
//                            for F <= 128, 48 KiB for F <= 192 (an immediate ds_read offset)
//   dec_off                  early-termination flags: one int per candidate + 3 vote words
//   val_off                  [slots][64*ch] float64 leaf values parked for the ordered sum
//   [img_off, 163840)        the group's trees
bool pk_q_make_layout(int F, int slots, int ch, pk_q_layout *L)
{
    if (F < 1 || slots < 2 || slots > 16 || (ch != 1 && ch != 2 && ch != 4)) return false;
    if (F > (ch == 1 ? 1023 : 255)) return false;
    L->F = F;
    L->slots = slots;
    L->ch = ch;
    L->slot_bytes = 0;
    memset(L->slot_off, 0, sizeof(L->slot_off));
    // ch = 1: one 64-candidate tile [F][64] u16 (the wide node word, up to 1023 features)
    L->HB = ch == 1 ? F * 128 : F * 256;
    const int C = 64 * ch;
    int top;
    L->half1 = 0;
    if (ch == 4) {
        if (L->HB > 49152) return false;
        L->half1 = L->HB <= 32768 ? 32768 : 49152;
        top = L->half1 + L->HB;
    } else {
        top = L->HB;
    }
    const int dec_bytes = (C * 4 + 16 + 32 + 15) & ~15;  // flags, 3 vote words (+1), 8 slot counters
    // the flags go into the gap below the second half tile when they fit there
    if (ch == 4 && L->HB + dec_bytes <= L->half1) {
        L->dec_off = L->HB;
    } else {
        L->dec_off = top;
        top += dec_bytes;
    }
    L->val_off = top;
    top += slots * C * 8;
    L->img_off = (top + 15) & ~15;
    L->cap = 163840 - L->img_off;
    const int stage = pk_q_stage_regs() * 1024 * 16;  // what the register staging moves per group
    if (L->cap > stage) L->cap = stage;
    return L->cap >= 4096;
}

int pk_q_max_tree_bytes(const pk_q_out &out)
{
    int m = 0;
    for (size_t t = 0; t + 1 < out.toff.size(); t++) m = std::max(m, (out.toff[t + 1] - out.toff[t]) * 8);
    return (m + 15) & ~15;
}

int pk_q_fixed_slots(const pk_q_out &out, int slots, pk_q_layout *L)
{
    const int T = (int)out.toff.size() - 1;
    L->slot_off[0] = 0;
    for (int s = 0; s < 16; s++) {
        int m = 0;
        if (s < slots)
            for (int t = s; t < T; t += slots) m = std::max(m, (out.toff[(size_t)t + 1] - out.toff[(size_t)t]) * 8);
        L->slot_off[s + 1] = L->slot_off[s] + ((m + 15) & ~15);
    }
    L->slot_bytes = L->slot_off[slots];
    return L->slot_bytes;
}

void pk_q_fill_lut(pk_q_out *out, int F, const std::vector<int32_t> &qcell)
{
    out->qlut.assign((size_t)F * PK_Q_CELLS, 0);
    for (int f = 0; f < F; f++) {
        const int o = out->qoff[(size_t)f], n = out->qoff[(size_t)f + 1] - o;
        uint32_t *lut = out->qlut.data() + (size_t)f * PK_Q_CELLS;
        int i = 0;
        for (int c = 0; c < PK_Q_CELLS; c++) {
            const int below = i;
            while (i < n && qcell[(size_t)o + i] <= c) i++;  // (cells of sorted thresholds never decrease)
            lut[c] = (uint32_t)below | ((uint32_t)(i - below) << 16);
        }
    }
}

// Rank tables of a forest: per (virtual) feature the sorted distinct float32 thresholds, the lookup
// cells of the quantizer, and which float feature each row of the rank tile is made from.
// A feature with more than 2047 distinct thresholds (the rank field has 11 bits; the fitted 500-tree
// forest of configs[4] has one with 2 284) is split: its first 2 047 thresholds stay with the feature,
// every further 2 047 become a VIRTUAL feature appended behind the real ones -- one more row of the
// rank tile, quantized from the same float values against its own part of the list.  A node whose
// threshold is number k of the feature's list tests virtual feature k / 2047 with rank k % 2047:
// exact, because the code of part c is clamp(r(x) - 2047 c, 0, 2047) by construction.  (Round 4;
// rounds 2-3 sent such forests to the float kernels.)
int pk_q_tables(int T, int F, const int32_t *tree_off, const int32_t *left, const int32_t *feat,
                const double *thr, int max_rank, pk_q_out *out)
{
    *out = pk_q_out();
    out->max_rank = max_rank;
    if (T <= 0 || F < 1 || F > 1023) return PK_E_UNSUPPORTED;
    std::vector<std::vector<float>> per((size_t)F);
    for (int t = 0; t < T; t++)
        for (int32_t v = tree_off[t]; v < tree_off[t + 1]; v++)
            if (left[v] != -1) {
                if (feat[v] < 0 || feat[v] >= F) {
                    pk_set_error("forest rank image: tree %d: feature index out of range", t);
                    return PK_E_INVALID;
                }
                if (!(thr[v] == thr[v])) return PK_E_UNSUPPORTED;  // NaN threshold
                per[(size_t)feat[v]].push_back(q_floor32(thr[v]));
            }
    // rows 0 .. F-1: the features' first parts; then the further parts, feature by feature
    out->qsrc.resize((size_t)F);
    out->qfirst.assign((size_t)F, -1);
    std::vector<std::vector<float>> rows((size_t)F);
    for (int f = 0; f < F; f++) {
        auto &p = per[(size_t)f];
        std::sort(p.begin(), p.end());
        p.erase(std::unique(p.begin(), p.end()), p.end());  // -0.0 == 0.0: one entry
        out->qsrc[(size_t)f] = f;
        rows[(size_t)f].assign(p.begin(), p.begin() + (ptrdiff_t)std::min<size_t>(p.size(), (size_t)max_rank));
    }
    for (int f = 0; f < F; f++) {
        const auto &p = per[(size_t)f];
        for (size_t o = (size_t)max_rank; o < p.size(); o += (size_t)max_rank) {
            if (out->qfirst[(size_t)f] < 0) out->qfirst[(size_t)f] = (int32_t)rows.size();
            out->qsrc.push_back(f);
            rows.emplace_back(p.begin() + (ptrdiff_t)o, p.begin() + (ptrdiff_t)std::min(p.size(), o + (size_t)max_rank));
        }
    }
    const int Fq = (int)rows.size();
    if (Fq > 1023) return PK_E_UNSUPPORTED;
    out->Fq = Fq;
    out->qoff.assign((size_t)Fq + 1, 0);
    for (int f = 0; f < Fq; f++) {
        out->qoff[(size_t)f + 1] = out->qoff[(size_t)f] + (int32_t)rows[(size_t)f].size();
        out->qthr.insert(out->qthr.end(), rows[(size_t)f].begin(), rows[(size_t)f].end());
    }
    out->qthr.push_back(0.f);
    // lookup cells for the quantizer.  cell(x) = pk_q_cell(x) is monotone in x, so every
    // threshold in a lower cell is below x and every threshold in a higher cell is not:
    // r(x) = thresholds in lower cells + those of x's own cell that are below x.
    out->qpar.assign((size_t)Fq * 2, 0.f);
    std::vector<int32_t> qcell(out->qthr.size(), 0);
    for (int f = 0; f < Fq; f++) {
        const float *b = out->qthr.data() + out->qoff[(size_t)f];
        const int n = out->qoff[(size_t)f + 1] - out->qoff[(size_t)f];
        float lo = 0.f, inv = 0.f;
        if (n >= 2 && b[n - 1] > b[0]) {
            lo = b[0];
            const double w = ((double)b[n - 1] - (double)b[0]) / (double)PK_Q_CELLS;
            inv = (float)(1.0 / w);
            if (!(inv > 0.f) || !std::isfinite(inv)) inv = 0.f;
        }
        out->qpar[(size_t)f * 2] = lo;
        out->qpar[(size_t)f * 2 + 1] = inv;
        for (int i = 0; i < n; i++) qcell[(size_t)out->qoff[(size_t)f] + i] = pk_q_cell(b[i], lo, inv);
    }
    pk_q_fill_lut(out, Fq, qcell);
    return PK_OK;
}

// A tree that has more child pairs than the pair field counts is cut in two (round 4): the subtree S
// nearest to half its size is taken out.  Piece A is the tree with a pure 0.0 leaf in S's place; piece B
// is the path from the root to S -- the same tests, every way off the path ending in a pure 0.0 leaf --
// with S at its end.  A candidate reaches S in exactly one of the two and a 0.0 leaf in the other, and
// x + 0.0 == x exactly: the model's sequential float64 sum is unchanged, only its divisor must stay the
// model's tree count (the kernels' t_div).  Pieces are cut again while they do not fit.
struct q_owned_tree {
    std::vector<int32_t> left, right, feat;
    std::vector<double> thr, p1;
    std::vector<uint8_t> miss;
    q_tree view() const
    {
        return q_tree{(int)left.size(), left.data(), right.data(), feat.data(), thr.data(), p1.data(),
                      miss.empty() ? nullptr : miss.data()};
    }
    int add(int l, int r, int f, double t, double v, uint8_t m)
    {
        left.push_back(l); right.push_back(r); feat.push_back(f); thr.push_back(t); p1.push_back(v); miss.push_back(m);
        return (int)left.size() - 1;
    }
};

static bool q_cut_tree(const q_tree &t, q_owned_tree *A, q_owned_tree *B)
{
    const int nn = t.nn;
    // interior nodes per subtree and parents (iterative post-order; the tree was validated by q_emit_tree)
    std::vector<int> parent((size_t)nn, -1), size((size_t)nn, 0), order;
    order.reserve((size_t)nn);
    std::vector<int> stack{0};
    while (!stack.empty()) {
        const int v = stack.back();
        stack.pop_back();
        order.push_back(v);
        if (t.left[v] != -1) {
            parent[(size_t)t.left[v]] = parent[(size_t)t.right[v]] = v;
            stack.push_back(t.left[v]);
            stack.push_back(t.right[v]);
        }
    }
    for (size_t i = order.size(); i-- > 0;) {
        const int v = order[i];
        if (t.left[v] != -1) size[(size_t)v] = 1 + size[(size_t)t.left[v]] + size[(size_t)t.right[v]];
    }
    const int total = size[0];
    if (total < 4) return false;
    int s = -1;
    for (int v : order)
        if (v != 0 && t.left[v] != -1 && (s < 0 || abs(2 * size[(size_t)v] - total) < abs(2 * size[(size_t)s] - total))) s = v;
    if (s < 0) return false;
    auto m_of = [&](int v) { return (uint8_t)(t.miss ? t.miss[v] : 0); };
    // copies the subtree of `v` (all of it, or with S replaced by a 0.0 leaf) into `dst`; returns its root there
    auto copy_sub = [&](int v0, bool drop_s, q_owned_tree *dst) {
        std::vector<std::pair<int, int>> st;  // (source node, index in dst)
        const int root = dst->add(-1, -1, -2, -2.0, 0.0, 0);
        st.push_back({v0, root});
        while (!st.empty()) {
            const auto [v, d] = st.back();
            st.pop_back();
            if (t.left[v] == -1 || (drop_s && v == s)) {
                dst->p1[(size_t)d] = (drop_s && v == s) ? 0.0 : t.p1[v];
                continue;
            }
            dst->feat[(size_t)d] = t.feat[v];
            dst->thr[(size_t)d] = t.thr[v];
            dst->miss[(size_t)d] = m_of(v);
            const int l = dst->add(-1, -1, -2, -2.0, 0.0, 0), r = dst->add(-1, -1, -2, -2.0, 0.0, 0);
            dst->left[(size_t)d] = l;
            dst->right[(size_t)d] = r;
            st.push_back({t.right[v], r});
            st.push_back({t.left[v], l});
        }
        return root;
    };
    *A = q_owned_tree();
    copy_sub(0, true, A);
    *B = q_owned_tree();
    std::vector<int> path;  // root .. parent of s
    for (int v = parent[(size_t)s]; v >= 0; v = parent[(size_t)v]) path.push_back(v);
    std::reverse(path.begin(), path.end());
    int at = B->add(-1, -1, -2, -2.0, 0.0, 0);  // the root of B
    for (size_t i = 0; i < path.size(); i++) {
        const int u = path[i], next = i + 1 < path.size() ? path[i + 1] : s;
        B->feat[(size_t)at] = t.feat[u];
        B->thr[(size_t)at] = t.thr[u];
        B->miss[(size_t)at] = m_of(u);
        const int on = B->add(-1, -1, -2, -2.0, 0.0, 0), off = B->add(-1, -1, -2, -2.0, 0.0, 0);  // off: a pure 0.0 leaf
        if (t.left[u] == next) {
            B->left[(size_t)at] = on;
            B->right[(size_t)at] = off;
        } else {
            B->left[(size_t)at] = off;
            B->right[(size_t)at] = on;
        }
        at = on;
    }
    {   // S itself, rooted at `at`
        q_owned_tree S;
        copy_sub(s, false, &S);
        const int base = (int)B->left.size();
        // node 0 of S lands on `at`, the others behind the nodes B has so far (indices shift by base - 1)
        auto map = [&](int i) { return i < 0 ? i : (i == 0 ? at : base + i - 1); };
        B->left[(size_t)at] = map(S.left[0]); B->right[(size_t)at] = map(S.right[0]);
        B->feat[(size_t)at] = S.feat[0]; B->thr[(size_t)at] = S.thr[0]; B->p1[(size_t)at] = S.p1[0]; B->miss[(size_t)at] = S.miss[0];
        for (size_t i = 1; i < S.left.size(); i++) B->add(map(S.left[i]), map(S.right[i]), S.feat[i], S.thr[i], S.p1[i], S.miss[i]);
    }
    return true;
}

// The tree images (they depend on the word format, not on the layout).  A tree beyond the pair field is
// cut into pieces (q_cut_tree): the image then holds more trees than the model (out->troot.size()).
// PK_E_UNSUPPORTED when a piece still does not fit.
int pk_q_trees(int T, int F, const int32_t *tree_off, const int32_t *left, const int32_t *right,
               const int32_t *feat, const double *thr, const uint8_t *miss, const double *p1, int mode,
               pk_q_out *out)
{
    if (out->Fq > (mode == PK_Q_WIDE ? 1023 : 255)) return PK_E_UNSUPPORTED;
    if ((mode == PK_Q_NARROW12) != (out->max_rank == PK_Q_MAX_RANK12)) return PK_E_INVALID;  // (tables of the other width)
    out->mode = mode;
    out->pairs.clear();
    out->troot.clear();
    out->tdepth.clear();
    std::vector<int32_t> toff{0};  // pairs
    // emits one tree (or its pieces, depth first: A before B keeps them next to each other in the image)
    int budget = 64 * T + 4096;   // pieces in all (a guard against a pathological forest, not a tuning knob)
    std::function<int(const q_tree &, int, int)> emit = [&](const q_tree &tv, int t, int level) -> int {
        std::string err;
        uint32_t root = 0;
        int depth = 0, split = 0;
        const int np = q_emit_tree(tv, F, *out, &out->pairs, &root, &depth, &err, mode, &split);
        if (np == -1) {
            pk_set_error("forest rank image: tree %d: %s", t, err.c_str());
            return PK_E_INVALID;
        }
        if (np == -2) {  // more pairs than the field counts: two pieces
            q_owned_tree A, B;
            if (level > 12 || --budget < 0 || !q_cut_tree(tv, &A, &B)) return PK_E_UNSUPPORTED;
            int rc = emit(A.view(), t, level + 1);
            if (!rc) rc = emit(B.view(), t, level + 1);
            return rc;
        }
        if (np < 0 || depth > 0xFFFF) return PK_E_UNSUPPORTED;
        toff.push_back(toff.back() + np);
        out->troot.push_back(root);
        // (the walk's level count in the low half, the split in the high half of one table word)
        out->tdepth.push_back(depth | (split << 16));
        return PK_OK;
    };
    for (int t = 0; t < T; t++) {
        const int32_t b = tree_off[t];
        const q_tree tv{tree_off[t + 1] - b, left + b, right + b, feat + b, thr + b, p1 + b,
                        miss ? miss + b : nullptr};
        if (tv.nn <= 0) {
            pk_set_error("forest rank image: tree %d is empty", t);
            return PK_E_INVALID;
        }
        const int rc = emit(tv, t, 0);
        if (rc) return rc;
    }
    out->toff = toff;
    out->pairs.push_back(make_uint2(0, 0));  // the clamped staging loads stay inside
    out->pairs.push_back(make_uint2(0, 0));
    return PK_OK;
}

// Rank tables + tree images + groups for one layout (made for out->Fq rows: pk_q_tables first when
// the layout depends on it).  PK_E_UNSUPPORTED when the forest does not fit the format (more than
// 1023 rank-tile rows, a tree of more pairs than the pair field counts or larger than the LDS budget).
int pk_q_build(int T, int F, const int32_t *tree_off, const int32_t *left, const int32_t *right,
               const int32_t *feat, const double *thr, const uint8_t *miss, const double *p1,
               const pk_q_layout &L, pk_q_out *out)
{
    int rc = pk_q_tables(T, F, tree_off, left, feat, thr, PK_Q_MAX_RANK, out);
    if (rc) return rc;
    if (L.F != out->Fq) return PK_E_UNSUPPORTED;  // (the layout was made for another row count)
    rc = pk_q_trees(T, F, tree_off, left, right, feat, thr, miss, p1, L.ch == 1 ? PK_Q_WIDE : PK_Q_NARROW, out);
    if (rc) return rc;
    return pk_q_group(out, L);
}

// Groups of consecutive trees for one layout: at most `slots` trees and `cap` bytes each.
// (The trees themselves do not depend on the layout.)  PK_E_UNSUPPORTED if a tree is
// larger than the image area.
int pk_q_group(pk_q_out *out, const pk_q_layout &L)
{
    const int T = (int)out->troot.size();
    const std::vector<int32_t> &toff = out->toff;
    out->gtab.clear();
    for (int t = 0; t < T; t++)
        if ((toff[(size_t)t + 1] - toff[(size_t)t]) * 8 >
            (L.slot_bytes ? L.slot_off[t % L.slots + 1] - L.slot_off[t % L.slots] : L.cap))
            return PK_E_UNSUPPORTED;
    if (L.slot_bytes > L.cap) return PK_E_UNSUPPORTED;
    out->ttab.assign((size_t)T * 4, 0);
    int t = 0;
    while (t < T) {
        int t1 = t + 1;
        // (fixed tree slots: every tree fits its slot, so only the count limits a group)
        while (t1 < T && t1 - t < L.slots &&
               (L.slot_bytes || (toff[(size_t)t1 + 1] - toff[(size_t)t]) * 8 <= L.cap))
            t1++;
        out->gtab.push_back(t);
        out->gtab.push_back(t1 - t);
        out->gtab.push_back(toff[(size_t)t] / 2);                        // 16-byte units
        out->gtab.push_back((toff[(size_t)t1] - toff[(size_t)t]) / 2);
        for (int k = t; k < t1; k++) {
            out->ttab[(size_t)k * 4] = (toff[(size_t)k] - toff[(size_t)t]) * 8;  // bytes inside the group
            out->ttab[(size_t)k * 4 + 1] = out->tdepth[(size_t)k];
            out->ttab[(size_t)k * 4 + 2] = (int32_t)out->troot[(size_t)k];
            out->ttab[(size_t)k * 4 + 3] = (toff[(size_t)k + 1] - toff[(size_t)k]) / 2;  // 16-byte units
        }
        t = t1;
    }
    out->n_grp = (int)(out->gtab.size() / 4);
    for (int k = 0; k < 2; k++) {  // the kernel reads one entry past the last group
        out->gtab.push_back(T);
        out->gtab.push_back(0);
        out->gtab.push_back(toff[(size_t)T] / 2);
        out->gtab.push_back(0);
    }
    return PK_OK;
}

// Diagnostic / test entry (no device needed): the rank tables and the rank image of a
// forest, so that tests can quantize and walk on the CPU.
extern "C" int pk_debug_forest_qimage(int T, int F, const int32_t *tree_off, const int32_t *left,
                                      const int32_t *right, const int32_t *feat, const double *thr,
                                      const uint8_t *miss_left, const double *p1, int slots, int ch,
                                      int32_t *layout8 /* 32 entries */, int32_t *qoff, int64_t cap_thr, float *qthr,
                                      uint32_t *qlut, float *qpar, int64_t cap_pairs, uint64_t *pairs,
                                      int64_t *n_pairs, int64_t cap_groups, int32_t *gtab,
                                      int32_t *n_groups, int32_t *ttab, int32_t cap_rows, int32_t *qsrc)
{
    // (qoff / qlut / qpar / qsrc hold cap_rows rows: the rank tile has F rows plus one per further
    // 2 047 thresholds of a feature; layout8[26] receives the row count)
    if (T <= 0 || !tree_off || !left || !right || !feat || !thr || !p1 || !layout8 || !qoff || !qthr ||
        !qlut || !qpar || !pairs || !n_pairs || !gtab || !n_groups || !ttab) {
        pk_set_error("pk_debug_forest_qimage: bad arguments");
        return PK_E_INVALID;
    }
    const bool fixed_slots = (ch & 0x100) != 0;
    const bool rank12 = (ch & 0x200) != 0;  // the 12-bit rank field (PK_Q_NARROW12; ch = 2 or 4 only)
    ch &= 0xFF;
    if (rank12 && ch == 1) {
        pk_set_error("pk_debug_forest_qimage: the wide word has no 12-bit rank form");
        return PK_E_INVALID;
    }
    pk_q_layout L;
    pk_q_out out;
    int rc = pk_q_tables(T, F, tree_off, left, feat, thr, rank12 ? PK_Q_MAX_RANK12 : PK_Q_MAX_RANK, &out);
    if (rc == PK_OK && (out.Fq > cap_rows || !qsrc)) {
        pk_set_error("pk_debug_forest_qimage: %d rank-tile rows, room for %d", out.Fq, (int)cap_rows);
        return PK_E_NOMEM;
    }
    if (rc == PK_OK && !pk_q_make_layout(out.Fq, slots, ch, &L)) {
        pk_set_error("pk_debug_forest_qimage: no LDS layout for %d rows, %d slots, %d walks per lane", out.Fq,
                     slots, ch);
        return PK_E_UNSUPPORTED;
    }
    if (rc == PK_OK)
        rc = pk_q_trees(T, F, tree_off, left, right, feat, thr, miss_left, p1,
                        L.ch == 1 ? PK_Q_WIDE : rank12 ? PK_Q_NARROW12 : PK_Q_NARROW, &out);
    if (rc == PK_OK) rc = pk_q_group(&out, L);
    if (fixed_slots && (rc == PK_OK || (rc == PK_E_UNSUPPORTED && !out.toff.empty()))) {
        pk_q_fixed_slots(out, slots, &L);
        rc = pk_q_group(&out, L);
    }
    if (rc) {
        if (rc == PK_E_UNSUPPORTED) pk_set_error("pk_debug_forest_qimage: the forest does not fit the rank format");
        return rc;
    }
    // (a tree cut into pieces, q_cut_tree: the tree table has more rows than the model has trees; the
    // caller's tables hold cap_groups - 4 trees)
    if ((int64_t)out.qthr.size() > cap_thr || (int64_t)out.pairs.size() > cap_pairs ||
        out.n_grp + 2 > cap_groups || (int64_t)out.troot.size() > cap_groups - 4) {
        pk_set_error("pk_debug_forest_qimage: output buffers too small");
        return PK_E_NOMEM;
    }
    int32_t lay[32] = {L.HB, L.ch | (L.half1 << 8), L.dec_off, L.val_off, L.img_off, L.cap, L.slots, L.F,
                       L.slot_bytes};
    for (int i = 0; i < 17; i++) lay[9 + i] = L.slot_off[i];
    lay[26] = out.Fq;
    lay[27] = out.mode;
    lay[28] = (int32_t)out.troot.size();  // trees of the image: the model's, or more (pieces)
    memcpy(layout8, lay, sizeof(lay));
    memcpy(qsrc, out.qsrc.data(), out.qsrc.size() * sizeof(int32_t));
    memcpy(qoff, out.qoff.data(), out.qoff.size() * sizeof(int32_t));
    memcpy(qthr, out.qthr.data(), out.qthr.size() * sizeof(float));
    memcpy(qlut, out.qlut.data(), out.qlut.size() * sizeof(uint32_t));
    memcpy(qpar, out.qpar.data(), out.qpar.size() * sizeof(float));
    for (size_t i = 0; i < out.pairs.size(); i++)
        pairs[i] = ((uint64_t)out.pairs[i].y << 32) | out.pairs[i].x;
    *n_pairs = (int64_t)out.pairs.size();
    memcpy(gtab, out.gtab.data(), out.gtab.size() * sizeof(int32_t));
    *n_groups = out.n_grp;
    memcpy(ttab, out.ttab.data(), out.ttab.size() * sizeof(int32_t));
    return PK_OK;
}
