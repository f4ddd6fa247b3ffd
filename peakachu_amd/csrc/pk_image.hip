// pk_image.hip -- host-side builder of the forest's "LDS image" (gfx950 only).
//
// The image is what forest_img_kernel (pk_forest_img.hip) copies into LDS,
// one group of trees at a time, and walks there.  It serves
// model.predict_proba(fea)[:, 1] at peakachu/scoreUtils.py:109 (sklearn
// Tree._apply_dense: `x[feature] <= threshold` goes left, NaN goes where
// missing_go_to_left says, the leaf's class-1 fraction is the tree's value).
//
// Format (every word is 8 bytes, `uint2 {x, y}`; all addresses are ABSOLUTE
// LDS byte addresses, which is why an image belongs to one LDS layout):
//   interior node  x = float32 threshold (largest float32 <= sklearn's double)
//                  y = bits 3..17 address of the node's CHILD PAIR: the word of
//                                 the left child at that address, the word of
//                                 the right child 8 bytes further
//                      bits 24..31 feature index, bit 0 NaN-goes-left
//   leaf           x = 0x7FC00000 | address of the leaf's float64 value
//                      (a quiet NaN: `feature <= NaN` is false for every
//                      feature, so a leaf always "goes right")
//                  y = its own address - 8, i.e. the pair whose RIGHT word is
//                      the leaf itself: a walk that has reached a leaf stays
//                      on it however many more levels it is asked to descend
// so the kernel walks every tree for a fixed number of levels (the depth of
// the tree) with no per-level termination test and no exec masking, and reads
// the value through the leaf word at the end.  Pure leaves (class-1 fraction
// exactly 0 or 1, > 90 % of the leaves of a grown forest) share two value
// slots per group; a node whose children are BOTH pure points at one of four
// constant pairs; a pure leaf next to a sibling with a word of its own needs
// a slot in the pair, and two such pairs (pure on the right / pure on the
// left, same value) overlap on it: [L][pure][R].
//
// The virtual image of a group is copied 16 bytes at a time into the two LDS
// regions the feature tile leaves free (pk_img_layout); pairs never straddle
// the gap between them.
#include <math.h>
#include <string.h>

#include <map>

#include "pk_common.h"

namespace {

constexpr uint32_t LEAF_NAN = 0x7FC00000u;
constexpr int CONST_SLOTS = 10;  // pad X0 X1 X0 X0 X1 X1 V0 V1 pad

float floor32(double t)
{
    float f = (float)t;
    if ((double)f > t) f = nextafterf(f, -INFINITY);
    return f;
}

struct tree_view {
    int nn;
    const int32_t *left, *right, *feat;
    const double *thr, *p1;
    const uint8_t *miss;
    // kind: 0 interior, 1 stored leaf, 2 pure 0.0, 3 pure 1.0
    int kind(int v) const
    {
        if (left[v] != -1) return 0;
        uint64_t b;
        memcpy(&b, &p1[v], 8);
        if (b == 0) return 2;
        if (p1[v] == 1.0) return 3;
        return 1;
    }
};

struct emitter {
    const pk_img_layout &L;
    std::vector<uint2> *words;  // group image under construction (nullptr = dry run)
    int vo;                     // next free virtual byte offset
    emitter(const pk_img_layout &l, std::vector<uint2> *w, int start) : L(l), words(w), vo(start) {}
    int phys(int v) const { return v < L.lenA ? L.HB + v : L.B0 + (v - L.lenA); }
    int alloc(int k)
    {
        if (vo < L.lenA && vo + 8 * k > L.lenA) vo = L.lenA;  // a pair never straddles the gap
        const int p = vo;
        vo += 8 * k;
        return p;
    }
    void put(int v, uint32_t x, uint32_t y)
    {
        if (!words) return;
        const size_t i = (size_t)v / 8;
        if (words->size() <= i) words->resize(i + 1, make_uint2(0, 0));
        (*words)[i] = make_uint2(x, y);
    }
    void put_f64(int v, double d)
    {
        uint64_t b;
        memcpy(&b, &d, 8);
        put(v, (uint32_t)(b & 0xffffffffu), (uint32_t)(b >> 32));
    }
    // a leaf word at virtual position v whose value lives at virtual position val
    void put_leaf(int v, int val) { put(v, LEAF_NAN | (uint32_t)phys(val), (uint32_t)(phys(v) - 8)); }
};

// virtual positions of the per-group constants (the group image starts with them)
inline int const_x(int i) { return 8 * i; }          // slot i of the constant block
constexpr int CONST_V0 = 8 * 7, CONST_V1 = 8 * 8;
// pair of two pure leaves (vl, vr) -> virtual address of the constant pair
inline int const_pair(int vl, int vr)
{
    if (vl == 0 && vr == 1) return const_x(1);
    if (vl == 1 && vr == 0) return const_x(2);
    if (vl == 0 && vr == 0) return const_x(3);
    return const_x(5);
}

void emit_consts(emitter &e)
{
    static const int val[CONST_SLOTS] = {-1, 0, 1, 0, 0, 1, 1, -1, -1, -1};
    for (int i = 0; i < CONST_SLOTS; i++)
        if (val[i] >= 0) e.put_leaf(const_x(i), val[i] ? CONST_V1 : CONST_V0);
        else e.put(const_x(i), 0, 0);
    e.put_f64(CONST_V0, 0.0);
    e.put_f64(CONST_V1, 1.0);
}

// Lays one tree out at the emitter's position.  Returns false on a malformed tree.
// root: the word a walk starts from; depth: decisions on the longest path.
bool emit_tree(emitter &e, const tree_view &t, int F, uint2 *root, int *depth, std::string *err)
{
    const int nn = t.nn;
    std::vector<int> order, dep((size_t)nn, 0), stack;
    std::vector<uint8_t> seen((size_t)nn, 0);
    int maxd = 0;
    stack.push_back(0);
    while (!stack.empty()) {  // preorder over the reachable nodes
        const int v = stack.back();
        stack.pop_back();
        if (v < 0 || v >= nn || seen[(size_t)v]) {
            if (err) *err = "malformed tree (cycle or child out of range)";
            return false;
        }
        seen[(size_t)v] = 1;
        if (dep[(size_t)v] > maxd) maxd = dep[(size_t)v];
        if (t.left[v] == -1) continue;
        const int l = t.left[v], r = t.right[v];
        if (l < 0 || l >= nn || r < 0 || r >= nn || t.feat[v] < 0 || t.feat[v] >= F) {
            if (err) *err = "malformed tree (bad child or feature index)";
            return false;
        }
        order.push_back(v);
        dep[(size_t)l] = dep[(size_t)r] = dep[(size_t)v] + 1;
        stack.push_back(r);
        stack.push_back(l);
    }
    *depth = maxd;
    std::map<uint64_t, int> vpos;  // stored leaf value bits -> virtual position of the float64
    auto value_slot = [&](int leaf) {
        const int k = t.kind(leaf);
        if (k == 2) return CONST_V0;
        if (k == 3) return CONST_V1;
        uint64_t b;
        memcpy(&b, &t.p1[leaf], 8);
        auto it = vpos.find(b);
        if (it != vpos.end()) return it->second;
        const int p = e.alloc(1);
        e.put_f64(p, t.p1[leaf]);
        vpos[b] = p;
        return p;
    };
    if (t.left[0] == -1) {
        // a one-leaf tree: [pad][leaf word]; the walk starts on the leaf and stays there
        const int val = value_slot(0);
        const int p = e.alloc(2);
        e.put(p, 0, 0);
        e.put_leaf(p + 8, val);
        *root = make_uint2(LEAF_NAN | (uint32_t)e.phys(val), (uint32_t)(e.phys(p + 8) - 8));
        return true;
    }
    // positions: pairv[n] = virtual address of n's child pair (or -1-constant for pure/pure),
    // wpos[v] = virtual position of the word of v (children that have a word)
    std::vector<int> pairv((size_t)nn, -1), wpos((size_t)nn, -1);
    std::vector<int> both, ip[2], pi[2];
    std::vector<std::pair<int, int>> xs;  // (virtual position, value) of pure-leaf slots
    for (int n : order) {
        const int kl = t.kind(t.left[n]), kr = t.kind(t.right[n]);
        const bool pl = kl >= 2, pr = kr >= 2;
        if (pl && pr) pairv[(size_t)n] = const_pair(kl - 2, kr - 2);
        else if (!pl && !pr) both.push_back(n);
        else if (pr) ip[kr - 2].push_back(n);
        else pi[kl - 2].push_back(n);
    }
    for (int n : both) {
        const int p = e.alloc(2);
        pairv[(size_t)n] = p;
        wpos[(size_t)t.left[n]] = p;
        wpos[(size_t)t.right[n]] = p + 8;
    }
    for (int v = 0; v < 2; v++) {
        const size_t m = ip[v].size() < pi[v].size() ? ip[v].size() : pi[v].size();
        for (size_t i = 0; i < m; i++) {  // [L of a][pure v][R of b]
            const int a = ip[v][i], b = pi[v][i];
            const int p = e.alloc(3);
            pairv[(size_t)a] = p;
            pairv[(size_t)b] = p + 8;
            wpos[(size_t)t.left[a]] = p;
            xs.push_back({p + 8, v});
            wpos[(size_t)t.right[b]] = p + 16;
        }
        for (size_t i = m; i < ip[v].size(); i++) {
            const int a = ip[v][i];
            const int p = e.alloc(2);
            pairv[(size_t)a] = p;
            wpos[(size_t)t.left[a]] = p;
            xs.push_back({p + 8, v});
        }
        for (size_t i = m; i < pi[v].size(); i++) {
            const int b = pi[v][i];
            const int p = e.alloc(2);
            pairv[(size_t)b] = p;
            xs.push_back({p, v});
            wpos[(size_t)t.right[b]] = p + 8;
        }
    }
    auto node_word = [&](int n) {
        const float t32 = floor32(t.thr[n]);
        uint32_t tb;
        memcpy(&tb, &t32, 4);
        uint32_t pk = (uint32_t)e.phys(pairv[(size_t)n]) | ((uint32_t)t.feat[n] << 24);
        if (t.miss && t.miss[n]) pk |= 1u;
        return make_uint2(tb, pk);
    };
    for (auto &x : xs) e.put_leaf(x.first, x.second ? CONST_V1 : CONST_V0);
    for (int n : order) {
        for (int c : {t.left[n], t.right[n]}) {
            const int p = wpos[(size_t)c];
            if (p < 0) continue;  // pure leaf
            if (t.kind(c) == 0) {
                const uint2 w = node_word(c);
                e.put(p, w.x, w.y);
            } else {
                e.put_leaf(p, value_slot(c));
            }
        }
    }
    *root = node_word(0);
    return true;
}

}  // namespace

// LDS map of forest_img_kernel for F features and `slots` tree slots:
//   [0, HB)                 feature tile of candidates 0..63   ([F][64] float32)
//   [HB, 65536)             image region A
//   [65536, 65536 + HB)     feature tile of candidates 64..127
//   [val_off, +slots*1024)  leaf values parked for the ordered sum
//   [dec_off, +544)         early-termination flags
//   [B0, 163840)            image region B
// max_image_bytes: what the kernel's register staging can move per group.
bool pk_img_make_layout(int F, int slots, int max_image_bytes, pk_img_layout *L)
{
    if (F < 1 || F > 255 || slots < 1 || slots > 16) return false;
    L->F = F;
    L->slots = slots;
    L->HB = F * 256;
    L->val_off = 65536 + L->HB;
    L->dec_off = L->val_off + slots * 1024;
    L->B0 = L->dec_off + 544;
    L->lenA = 65536 - L->HB;
    if (L->lenA < 256) L->lenA = 0;
    const int lenB = 163840 - L->B0;
    if (lenB < 1024) return false;
    L->cap = L->lenA + lenB;
    if (L->cap > max_image_bytes) L->cap = max_image_bytes & ~15;
    return L->cap >= 4096;
}

// Builds the images of all tree groups.  Groups hold consecutive trees, at most
// `slots` of them.  Returns PK_OK, or PK_E_UNSUPPORTED when a tree does not fit.
int pk_img_build(int T, int F, const int32_t *tree_off, const int32_t *left, const int32_t *right,
                 const int32_t *feat, const double *thr, const uint8_t *miss, const double *p1,
                 const pk_img_layout &L, pk_img_out *out)
{
    out->words.clear();
    out->gtab.clear();
    out->troot.assign((size_t)T, make_uint2(0, 0));
    out->tdepth.assign((size_t)T, 0);
    auto view = [&](int t) {
        const int32_t b = tree_off[t];
        return tree_view{tree_off[t + 1] - b, left + b, right + b, feat + b, thr + b, p1 + b,
                         miss ? miss + b : nullptr};
    };
    // dry run: bytes each tree needs (laid out alone, no straddle padding)
    std::vector<int> need((size_t)T);
    pk_img_layout flat = L;
    flat.lenA = 0;
    for (int t = 0; t < T; t++) {
        const tree_view tv = view(t);
        if (tv.nn <= 0) {
            pk_set_error("forest image: tree %d is empty", t);
            return PK_E_INVALID;
        }
        emitter e(flat, nullptr, 0);
        uint2 r;
        int d;
        std::string err;
        if (!emit_tree(e, tv, F, &r, &d, &err)) {
            pk_set_error("forest image: tree %d: %s", t, err.c_str());
            return PK_E_INVALID;
        }
        need[(size_t)t] = e.vo + 24;  // + one possible straddle pad
        if (8 * CONST_SLOTS + need[(size_t)t] > L.cap) return PK_E_UNSUPPORTED;
    }
    int t = 0;
    while (t < T) {
        int t1 = t, bytes = 8 * CONST_SLOTS;
        while (t1 < T && t1 - t < L.slots && bytes + need[(size_t)t1] <= L.cap) bytes += need[(size_t)t1++];
        std::vector<uint2> img;
        emitter e(L, &img, 0);
        e.alloc(CONST_SLOTS);
        emit_consts(e);
        for (int k = t; k < t1; k++) {
            std::string err;
            if (!emit_tree(e, view(k), F, &out->troot[(size_t)k], &out->tdepth[(size_t)k], &err)) {
                pk_set_error("forest image: tree %d: %s", k, err.c_str());
                return PK_E_INVALID;
            }
        }
        if (e.vo > L.cap) {
            pk_set_error("forest image: internal error, group of trees %d..%d overflows", t, t1 - 1);
            return PK_E_INVALID;
        }
        const size_t nw = ((size_t)e.vo / 8 + 1) & ~(size_t)1;  // whole 16-byte units
        img.resize(nw, make_uint2(0, 0));
        out->gtab.push_back(t);
        out->gtab.push_back(t1 - t);
        out->gtab.push_back((int32_t)(out->words.size() / 2));
        out->gtab.push_back((int32_t)(nw / 2));
        out->words.insert(out->words.end(), img.begin(), img.end());
        t = t1;
    }
    out->n_grp = (int)(out->gtab.size() / 4);
    for (int k = 0; k < 2; k++) {  // the kernel reads one entry past the last group
        out->gtab.push_back(T);
        out->gtab.push_back(0);
        out->gtab.push_back((int32_t)(out->words.size() / 2));
        out->gtab.push_back(0);
    }
    // the kernel's clamped staging loads may touch up to one unit past the end
    out->words.push_back(make_uint2(0, 0));
    out->words.push_back(make_uint2(0, 0));
    return PK_OK;
}

// Diagnostic / test entry (no device needed): build the image of a forest for the
// kernel's LDS layout and hand it out, so that tests can walk it on the CPU.
extern "C" int pk_debug_forest_image(int T, int F, const int32_t *tree_off, const int32_t *left,
                                     const int32_t *right, const int32_t *feat, const double *thr,
                                     const uint8_t *miss_left, const double *p1, int slots,
                                     int32_t *layout8, int64_t cap_words, uint64_t *words,
                                     int64_t *n_words, int64_t cap_groups, int32_t *gtab,
                                     int32_t *n_groups, uint64_t *troot, int32_t *tdepth)
{
    if (T <= 0 || !tree_off || !left || !right || !feat || !thr || !p1 || !layout8 || !words ||
        !n_words || !gtab || !n_groups || !troot || !tdepth) {
        pk_set_error("pk_debug_forest_image: bad arguments");
        return PK_E_INVALID;
    }
    pk_img_layout L;
    if (!pk_img_make_layout(F, slots, pk_img_stage_bytes(slots), &L)) {
        pk_set_error("pk_debug_forest_image: no LDS layout for F=%d, %d slots", F, slots);
        return PK_E_UNSUPPORTED;
    }
    pk_img_out out;
    const int rc = pk_img_build(T, F, tree_off, left, right, feat, thr, miss_left, p1, L, &out);
    if (rc) {
        if (rc == PK_E_UNSUPPORTED) pk_set_error("pk_debug_forest_image: a tree does not fit the LDS");
        return rc;
    }
    if ((int64_t)out.words.size() > cap_words || out.n_grp + 2 > cap_groups) {
        pk_set_error("pk_debug_forest_image: output buffers too small");
        return PK_E_NOMEM;
    }
    const int32_t lay[8] = {L.HB, L.lenA, L.B0, L.val_off, L.dec_off, L.cap, L.slots, L.F};
    memcpy(layout8, lay, sizeof(lay));
    for (size_t i = 0; i < out.words.size(); i++)
        words[i] = ((uint64_t)out.words[i].y << 32) | out.words[i].x;
    *n_words = (int64_t)out.words.size();
    memcpy(gtab, out.gtab.data(), out.gtab.size() * sizeof(int32_t));
    *n_groups = out.n_grp;
    for (int t = 0; t < T; t++) {
        troot[t] = ((uint64_t)out.troot[(size_t)t].y << 32) | out.troot[(size_t)t].x;
        tdepth[t] = out.tdepth[(size_t)t];
    }
    return PK_OK;
}
