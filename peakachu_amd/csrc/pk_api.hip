// pk_api.hip -- host side of the C ABI declared in include/peakachu_hip.h:
// handle lifetime, forest packing, the chunked extract -> forest pipeline and
// the result compaction.  Everything runs on one HIP stream per device; no
// deep-learning framework underneath, no CPU fallback.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <thread>

#include "pk_common.h"

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void pk_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static pk_options g_opt;
std::atomic<int64_t> g_stat_extract_clean{0}, g_stat_extract_general{0}, g_stat_extract_strip{0};
// Locks (include/peakachu_hip.h, 'Threading'): one recursive lock PER DEVICE, taken by every entry
// point for the device of its handle -- a device has one stream pair, one tile scratch and handles
// with cached launch tables, so calls on one device are serialised; calls on different devices run
// side by side (round 4; rounds 1-3 had one process-wide lock).  What is shared by all devices is
// small and has its own plain mutexes: g_mu (the context map, the option defaults, the taps) and
// g_prof_mu (the kernel-timer lists).  Order: device lock first, then g_mu / g_prof_mu, never the
// other way round.
static std::mutex g_mu, g_prof_mu;
static std::recursive_mutex g_dev_mu[PK_MAX_DEVICES];
std::recursive_mutex &pk_device_mutex(int device)
{
    return g_dev_mu[device >= 0 && device < PK_MAX_DEVICES ? device : 0];
}
pk_options pk_default_options()
{
    std::lock_guard<std::mutex> lk(g_mu);
    return g_opt;
}
static std::map<int, pk_device_ctx *> g_ctx;
// Gaussian taps of the window blur (see pk_set_gauss_taps); defaults = numpy 2.2 / scipy 1.15
static double g_taps[5] = {0x1.9884a307594fbp-2, 0x1.ef8eb9ad499bap-3, 0x1.ba4b99d1799abp-5,
                           0x1.22724cb7eb269p-8, 0x1.18a9c4fd536c6p-13};

// --------------------------------------------------------------- profiling
struct prof_rec {
    hipEvent_t e0, e1;
    pk_kclass k;
    int device;
};
static std::atomic<bool> g_prof_on{false};
static std::vector<prof_rec> g_prof_pending;
static double g_prof_ms[PK_K_NCLASS] = {0, 0, 0, 0, 0, 0};
static int64_t g_prof_n[PK_K_NCLASS] = {0, 0, 0, 0, 0, 0};

pk_prof_scope::pk_prof_scope(pk_device_ctx *c, pk_kclass kk, hipStream_t s)
    : ctx(c), k(kk), st(s ? s : c->stream)
{
    if (!g_prof_on) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        e0 = e1 = nullptr;
        return;
    }
    hipEventRecord(e0, st);
}
pk_prof_scope::~pk_prof_scope()
{
    if (!e0) return;
    hipEventRecord(e1, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_pending.push_back({e0, e1, k, ctx->device});
}

static void prof_drain()  // (under g_prof_mu)
{
    for (auto &r : g_prof_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess &&
            hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            g_prof_ms[r.k] += ms;
            g_prof_n[r.k] += 1;
        }
        hipEventDestroy(r.e0);
        hipEventDestroy(r.e1);
    }
    g_prof_pending.clear();
}

// ------------------------------------------------------------------ context
pk_device_ctx *pk_ctx(int device)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(device);
    if (it != g_ctx.end()) {
        if (hipSetDevice(device) != hipSuccess) {
            pk_set_error("hipSetDevice(%d) failed", device);
            return nullptr;
        }
        return it->second;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        pk_set_error("no HIP device available (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        pk_set_error("device %d out of range (0..%d)", device, ndev - 1);
        return nullptr;
    }
    PK_HIP_NULL(hipSetDevice(device));
    hipDeviceProp_t prop;
    PK_HIP_NULL(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        pk_set_error("device %d is %s; this library is built for gfx950 only", device,
                     prop.gcnArchName);
        return nullptr;
    }
    pk_device_ctx *c = new pk_device_ctx();
    c->device = device;
    c->cu_count = prop.multiProcessorCount;
    {
        // the forest stream gets the highest priority, the extractor stream the lowest:
        // extractor waves fill the registers / issue slots the forest leaves idle
        int lo = 0, hi = 0;
        PK_HIP_NULL(hipDeviceGetStreamPriorityRange(&lo, &hi));
        PK_HIP_NULL(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
        PK_HIP_NULL(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, lo));
        for (int i = 0; i < 2; i++) {
            PK_HIP_NULL(hipEventCreateWithFlags(&c->ev_ext[i], hipEventDisableTiming));
            PK_HIP_NULL(hipEventCreateWithFlags(&c->ev_for[i], hipEventDisableTiming));
        }
    }
    PK_HIP_NULL(hipMalloc((void **)&c->dbg_buf, 65536 * sizeof(long long)));
    PK_HIP_NULL(hipMemset(c->dbg_buf, 0, 65536 * sizeof(long long)));
    PK_HIP_NULL(hipMalloc((void **)&c->d_ret, PK_RET_BYTES));
    PK_HIP_NULL(hipHostMalloc((void **)&c->h_ret, PK_RET_BYTES, hipHostMallocDefault));
    if (pk_extract_upload_taps(g_taps) != PK_OK) return nullptr;
    g_ctx[device] = c;
    return c;
}

int pk_ctx_reserve_tiles(pk_device_ctx *c, size_t bytes)
{
    if (bytes <= c->fea_tiles_bytes) return PK_OK;
    if (c->fea_tiles) {
        PK_HIP(hipStreamSynchronize(c->stream));
        PK_HIP(hipFree(c->fea_tiles));
        c->fea_tiles = nullptr;
        c->fea_tiles_bytes = 0;
    }
    PK_HIP(hipMalloc((void **)&c->fea_tiles, bytes));
    c->fea_tiles_bytes = bytes;
    return PK_OK;
}

int pk_ctx_reserve_scan(pk_device_ctx *c, size_t bytes)
{
    if (bytes <= c->scan_scratch_bytes) return PK_OK;
    if (c->scan_scratch) {
        PK_HIP(hipStreamSynchronize(c->stream));
        PK_HIP(hipFree(c->scan_scratch));
        c->scan_scratch = nullptr;
        c->scan_scratch_bytes = 0;
    }
    bytes = bytes * 2 + 4096;
    PK_HIP(hipMalloc((void **)&c->scan_scratch, bytes));
    c->scan_scratch_bytes = bytes;
    return PK_OK;
}

// -------------------------------------------------------------- public: misc
extern "C" int pk_abi_version(void) { return PK_ABI_VERSION; }
extern "C" const char *pk_last_error(void) { return g_err; }

extern "C" int pk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n < 0 ? 0 : n;
}

extern "C" int pk_device_name(int device, char *buf, int buflen)
{
    if (!buf || buflen <= 0) return PK_E_INVALID;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        pk_set_error("no HIP device %d", device);
        return PK_E_NODEVICE;
    }
    snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName,
             prop.multiProcessorCount);
    return PK_OK;
}

extern "C" double pk_debug_prune_bound(double thre, int T, int64_t additions) { return pk_prune_bound(thre, T, additions); }

extern "C" int pk_runtime_versions(int *hip_runtime, int *hip_driver)
{
    if (!hip_runtime || !hip_driver) return PK_E_INVALID;
    *hip_runtime = *hip_driver = -1;
    if (hipRuntimeGetVersion(hip_runtime) != hipSuccess) *hip_runtime = -1;
    if (hipDriverGetVersion(hip_driver) != hipSuccess) *hip_driver = -1;
    (void)hipGetLastError();
    return PK_OK;
}

extern "C" int pk_device_synchronize(int device)
{
    PK_DEV_LOCK(device);
    pk_device_ctx *c = pk_ctx(device);
    if (!c) return PK_E_NODEVICE;
    PK_HIP(hipStreamSynchronize(c->stream2));
    PK_HIP(hipStreamSynchronize(c->stream));
    PK_HIP(hipDeviceSynchronize());
    return PK_OK;
}

extern "C" int pk_set_gauss_taps(const double *taps5)
{
    if (!taps5) {
        pk_set_error("pk_set_gauss_taps: null pointer");
        return PK_E_INVALID;
    }
    for (int i = 0; i < 5; i++)
        if (!(taps5[i] > 0.0) || !(taps5[i] < 1.0) || (i && !(taps5[i] < taps5[i - 1]))) {
            pk_set_error("pk_set_gauss_taps: taps must be finite, in (0, 1) and decreasing from the centre");
            return PK_E_INVALID;
        }
    std::vector<std::pair<int, pk_device_ctx *>> in_use;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        memcpy(g_taps, taps5, sizeof(g_taps));  // a device that comes up from here on starts with these
        for (auto &kv : g_ctx) in_use.push_back(kv);
    }
    for (auto &kv : in_use) {  // devices that are already in use: one after the other, under their own lock
        PK_DEV_LOCK(kv.first);
        double t[5];
        {
            std::lock_guard<std::mutex> lk(g_mu);
            memcpy(t, g_taps, sizeof(t));
        }
        PK_HIP(hipSetDevice(kv.first));
        PK_HIP(hipStreamSynchronize(kv.second->stream));
        PK_HIP(hipStreamSynchronize(kv.second->stream2));
        const int rc = pk_extract_upload_taps(t);
        if (rc) return rc;
    }
    return PK_OK;
}

extern "C" int pk_get_gauss_taps(double *taps5)
{
    if (!taps5) return PK_E_INVALID;
    std::lock_guard<std::mutex> lk(g_mu);
    memcpy(taps5, g_taps, sizeof(g_taps));
    return PK_OK;
}

// Options live in every handle (pk_forest / pk_matrix / pk_cands ::opt); pk_set_option sets the
// DEFAULTS new handles start from, pk_<handle>_set_option one handle's own copy.  (Rounds 1-3 had one
// process-wide set: a knob left set by one caller changed every other caller of the process.)
static int opt_assign(pk_options &o, const char *name, int64_t value)
{
    if (!name) return PK_E_INVALID;
    if (!strcmp(name, "chunk")) {
        if (value < 64) return PK_E_INVALID;
        o.chunk = value;
    } else if (!strcmp(name, "forest_ilp")) {
        if (value != 1 && value != 2 && value != 4 && value != 8) return PK_E_INVALID;
        o.forest_ilp = value;
    } else if (!strcmp(name, "forest_lds")) {
        if (value < 0) return PK_E_INVALID;
        o.forest_lds = value;
    } else if (!strcmp(name, "overlap")) {
        o.overlap = value != 0;
    } else if (!strcmp(name, "sub_chunk")) {
        if (value < 0) return PK_E_INVALID;
        o.sub_chunk = value;
    } else if (!strcmp(name, "forest_q_rank12")) {
        if (value < 0 || value > 2) return PK_E_INVALID;
        o.forest_q_rank12 = value;
    } else if (!strcmp(name, "forest_warm")) {
        if (value < 0) return PK_E_INVALID;
        o.forest_warm = value;
    } else if (!strcmp(name, "extract_clean")) {
        o.extract_clean = value != 0;
    } else if (!strcmp(name, "extract_pair")) {
        o.extract_pair = value != 0;
    } else if (!strcmp(name, "extract_row16")) {
        if (value < 0 || value > 2) return PK_E_INVALID;
        o.extract_row16 = value;  // (2: one wave per workgroup, the form of rounds 3-5)
    } else if (!strcmp(name, "extract_strip")) {
        if (value < 0 || value > 2) return PK_E_INVALID;
        o.extract_strip = value;
    } else if (!strcmp(name, "extract_diag")) {
        if (value < 0 || value > 2) return PK_E_INVALID;
        o.extract_diag = value;
    } else if (!strcmp(name, "forest_slots")) {
        if (value < 0 || value == 1 || value > 16) return PK_E_INVALID;
        o.forest_slots = value;
    } else if (!strcmp(name, "early_exit")) {
        o.early_exit = value != 0;
    } else if (!strcmp(name, "forest_q_two")) {
        o.forest_q_two = value != 0;
    } else if (!strcmp(name, "forest_q_help")) {
        o.forest_q_help = value != 0;
    } else if (!strcmp(name, "forest_q_rsv")) {
        o.forest_q_rsv = value;
    } else if (!strcmp(name, "forest_dbg")) {
        o.forest_dbg = value;
    } else if (!strcmp(name, "forest_q")) {
        o.forest_q = value != 0;
    } else if (!strcmp(name, "forest_q_ch")) {
        if (value != 0 && value != 1 && value != 2 && value != 4) return PK_E_INVALID;
        o.forest_q_ch = value;
    } else if (!strcmp(name, "forest_q_persist")) {
        if (value < -4096 || value > 8) return PK_E_INVALID;
        o.forest_q_persist = value;
    } else if (!strcmp(name, "forest_q_prio")) {
        o.forest_q_prio = value != 0;
    } else if (!strcmp(name, "forest_q_early")) {
        o.forest_q_early = value != 0;
    } else if (!strcmp(name, "forest_q_wpt")) {
        if (value < 0 || value > 2) return PK_E_INVALID;
        o.forest_q_wpt = value;
    } else if (!strcmp(name, "forest_img")) {
        o.forest_img = value != 0;
    } else if (!strcmp(name, "compact_small")) {
        o.compact_small = value != 0;
    } else if (!strcmp(name, "forest_split")) {
        o.forest_split = value != 0;
    } else if (!strcmp(name, "forest_split_at")) {
        if (value < 0) return PK_E_INVALID;
        o.forest_split_at = value;
    } else if (!strcmp(name, "forest_split_frac")) {
        if (value < 0 || value > 1000) return PK_E_INVALID;
        o.forest_split_frac = value;
    } else if (!strcmp(name, "forest_split_min")) {
        if (value < 1) return PK_E_INVALID;
        o.forest_split_min = value;
    } else {
        pk_set_error("unknown option '%s'", name);
        return PK_E_INVALID;
    }
    return PK_OK;
}

static int64_t opt_read(const pk_options &o, const char *name)
{
    if (!name) return -1;
    if (!strcmp(name, "chunk")) return o.chunk;
    if (!strcmp(name, "forest_ilp")) return o.forest_ilp;
    if (!strcmp(name, "forest_lds")) return o.forest_lds;
    if (!strcmp(name, "forest_slots")) return o.forest_slots;
    if (!strcmp(name, "extract_pair")) return o.extract_pair;
    if (!strcmp(name, "extract_row16")) return o.extract_row16;
    if (!strcmp(name, "extract_strip")) return o.extract_strip;
    if (!strcmp(name, "extract_diag")) return o.extract_diag;
    if (!strcmp(name, "extract_clean")) return o.extract_clean;
    if (!strcmp(name, "forest_warm")) return o.forest_warm;
    if (!strcmp(name, "overlap")) return o.overlap;
    if (!strcmp(name, "sub_chunk")) return o.sub_chunk;
    if (!strcmp(name, "forest_q_rank12")) return o.forest_q_rank12;
    if (!strcmp(name, "early_exit")) return o.early_exit;
    if (!strcmp(name, "forest_img")) return o.forest_img;
    if (!strcmp(name, "forest_q")) return o.forest_q;
    if (!strcmp(name, "forest_q_ch")) return o.forest_q_ch;
    if (!strcmp(name, "forest_q_wpt")) return o.forest_q_wpt;
    if (!strcmp(name, "forest_q_persist")) return o.forest_q_persist;
    if (!strcmp(name, "forest_q_prio")) return o.forest_q_prio;
    if (!strcmp(name, "forest_q_two")) return o.forest_q_two;
    if (!strcmp(name, "forest_q_help")) return o.forest_q_help;
    if (!strcmp(name, "forest_q_rsv")) return o.forest_q_rsv;
    if (!strcmp(name, "forest_dbg")) return o.forest_dbg;
    if (!strcmp(name, "forest_q_early")) return o.forest_q_early;
    if (!strcmp(name, "compact_small")) return o.compact_small;
    if (!strcmp(name, "forest_split")) return o.forest_split;
    if (!strcmp(name, "forest_split_at")) return o.forest_split_at;
    if (!strcmp(name, "forest_split_frac")) return o.forest_split_frac;
    if (!strcmp(name, "forest_split_min")) return o.forest_split_min;
    return -1;
}

extern "C" int pk_set_option(const char *name, int64_t value)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return opt_assign(g_opt, name, value);
}

extern "C" int64_t pk_get_option(const char *name)
{
    if (!name) return -1;
    if (!strcmp(name, "stat_extract_clean")) return g_stat_extract_clean;
    if (!strcmp(name, "stat_extract_general")) return g_stat_extract_general;
    if (!strcmp(name, "stat_extract_strip")) return g_stat_extract_strip;
    std::lock_guard<std::mutex> lk(g_mu);
    return opt_read(g_opt, name);
}

extern "C" int pk_forest_set_option(pk_forest *f, const char *name, int64_t value)
{
    PK_DEV_LOCK(f ? f->device : 0);
    return f ? opt_assign(f->opt, name, value) : PK_E_INVALID;
}
extern "C" int64_t pk_forest_get_option(pk_forest *f, const char *name)
{
    if (!f || !name) return -1;
    // read-only: what the rank plan of this forest came to (0 before the first scoring call plans it)
    if (!strcmp(name, "stat_family")) return f->last_family;
    if (!strcmp(name, "stat_q_mode")) return f->q_state == 1 ? f->q_mode : -1;   // PK_Q_NARROW / _WIDE / _NARROW12
    if (!strcmp(name, "stat_q_rows")) return f->q_state == 1 ? f->q_F : -1;
    if (!strcmp(name, "stat_q_shape")) return f->q_state == 1 ? f->q_ch : -1;
    if (!strcmp(name, "stat_q_trees")) return f->q_state == 1 ? f->q_T : -1;   // trees of the image (pieces count)
    if (!strcmp(name, "stat_q_groups")) return f->q_state == 1 ? f->q_n_grp : -1;
    // the cut forest: the group the last launch was cut in front of (0: one launch), the trees in front
    // of it, and how many candidates the cut launches of the last scoring call on this device parked
    if (!strcmp(name, "stat_split_group")) return f->last_cut;
    if (!strcmp(name, "stat_split_trees"))
        return f->last_cut > 0 && (size_t)f->last_cut * 4 < f->q_gtab_h.size() ? f->q_gtab_h[(size_t)f->last_cut * 4] : 0;
    if (!strcmp(name, "stat_split_shift")) return f->cut_off ? -1 : f->cut_shift;  // what the cut has learnt (-1: given up)
    if (!strcmp(name, "stat_split_parked")) {
        PK_DEV_LOCK(f->device);
        pk_device_ctx *ctx = pk_ctx(f->device);
        if (!ctx || !ctx->split_cnt || ctx->split_k <= 0) return 0;
        std::vector<unsigned> h((size_t)ctx->split_k);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
            hipMemcpy(h.data(), ctx->split_cnt, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess)
            return -1;
        int64_t n = 0;
        for (unsigned v : h) n += v;
        return n;
    }
    return opt_read(f->opt, name);
}

extern "C" int pk_matrix_set_option(pk_matrix *m, const char *name, int64_t value)
{
    PK_DEV_LOCK(m ? m->device : 0);
    return m ? opt_assign(m->opt, name, value) : PK_E_INVALID;
}
extern "C" int64_t pk_matrix_get_option(pk_matrix *m, const char *name) { return m ? opt_read(m->opt, name) : -1; }

extern "C" int pk_cands_set_option(pk_cands *c, const char *name, int64_t value)
{
    PK_DEV_LOCK(c ? c->device : 0);
    return c ? opt_assign(c->opt, name, value) : PK_E_INVALID;
}
extern "C" int64_t pk_cands_get_option(pk_cands *c, const char *name) { return c ? opt_read(c->opt, name) : -1; }

extern "C" int pk_debug_read(int device, int64_t *out, int64_t n)
{
    PK_DEV_LOCK(device);
    pk_device_ctx *c = pk_ctx(device);
    if (!c) return PK_E_NODEVICE;
    if (!out || n < 0 || n > 65536) return PK_E_INVALID;
    PK_HIP(hipStreamSynchronize(c->stream));
    PK_HIP(hipMemcpy(out, c->dbg_buf, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
    return PK_OK;
}

extern "C" int pk_debug_lock_probe(int device_a, int device_b)
{
    if (device_a < 0 || device_a >= PK_MAX_DEVICES || device_b < 0 || device_b >= PK_MAX_DEVICES) return PK_E_INVALID;
    PK_DEV_LOCK(device_a);
    int got = 0;
    std::thread t([&]() {  // (a recursive lock: the probe has to come from another thread)
        if (pk_device_mutex(device_b).try_lock()) {
            got = 1;
            pk_device_mutex(device_b).unlock();
        }
    });
    t.join();
    return got;
}

extern "C" int pk_prof_enable(int on)
{
    g_prof_on = on != 0;
    return PK_OK;
}
extern "C" int pk_prof_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain();
    for (int i = 0; i < PK_K_NCLASS; i++) {
        g_prof_ms[i] = 0;
        g_prof_n[i] = 0;
    }
    return PK_OK;
}
extern "C" int pk_prof_get(const char *name, double *ms_total, int64_t *launches)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    static const char *names[PK_K_NCLASS] = {"extract", "forest", "compact", "band", "quant", "forest_tail"};
    prof_drain();
    for (int i = 0; i < PK_K_NCLASS; i++)
        if (name && !strcmp(name, names[i])) {
            if (ms_total) *ms_total = g_prof_ms[i];
            if (launches) *launches = g_prof_n[i];
            return PK_OK;
        }
    pk_set_error("unknown kernel class '%s'", name ? name : "(null)");
    return PK_E_INVALID;
}

// ------------------------------------------------------------------- forest
// largest float32 <= t: for float32 x, (double)x <= t  <=>  x <= floor32(t)
static float floor_to_f32(double t)
{
    float f = (float)t;
    if ((double)f > t) f = nextafterf(f, -INFINITY);
    return f;
}

extern "C" pk_forest *pk_forest_create(int device, int T, int F, const int32_t *tree_off,
                                       const int32_t *left, const int32_t *right,
                                       const int32_t *feat, const double *thr,
                                       const uint8_t *miss_left, const double *p1)
{
    PK_DEV_LOCK(device);
    if (T <= 0 || F <= 0 || !tree_off || !left || !right || !feat || !thr || !p1) {
        pk_set_error("pk_forest_create: bad arguments");
        return nullptr;
    }
    if (F > PK_NODE_FEAT_MAX) {
        pk_set_error("pk_forest_create: F=%d exceeds %d features", F, PK_NODE_FEAT_MAX);
        return nullptr;
    }
    pk_device_ctx *ctx = pk_ctx(device);
    if (!ctx) return nullptr;

    std::vector<uint2> nodes;
    std::vector<int32_t> big;  // side table of right offsets >= PK_NODE_ROFF_BIG
    std::vector<uint8_t> tree_big(T, 0);
    bool any_big = false;
    std::vector<int32_t> root(T + 1);
    std::vector<int32_t> pos, stack, depth;
    int max_depth = 0, max_tree = 0;
    auto leaf_word = [](double v) {
        uint64_t b;
        memcpy(&b, &v, 8);
        return make_uint2((unsigned)(b & 0xffffffffu), (unsigned)(b >> 32));
    };
    auto leaf_kind = [](double v) -> unsigned {
        uint64_t b;
        memcpy(&b, &v, 8);
        if (b == 0) return PK_KIND_ZERO;                     // +0.0
        if (v == 1.0) return PK_KIND_ONE;
        return PK_KIND_LEAF;
    };
    for (int t = 0; t < T; t++) {
        const int32_t base = tree_off[t], nn = tree_off[t + 1] - base;
        if (nn <= 0) {
            pk_set_error("pk_forest_create: tree %d is empty", t);
            return nullptr;
        }
        if (nodes.size() & 1) nodes.push_back(make_uint2(0, 0));  // even start
        root[t] = (int32_t)nodes.size();
        if (left[base] == -1) {
            // a one-leaf tree: a root that sends everything to that leaf
            const unsigned kind = leaf_kind(p1[base]);
            unsigned pk = (1u << PK_NODE_MISS_BIT) | (kind << PK_NODE_LKIND_SHIFT) |
                          (kind << PK_NODE_RKIND_SHIFT) | (1u << PK_NODE_ROFF_SHIFT);  // feature 0
            float inf = INFINITY;
            unsigned tb;
            memcpy(&tb, &inf, 4);
            nodes.push_back(make_uint2(tb, pk));
            if (kind == PK_KIND_LEAF) nodes.push_back(leaf_word(p1[base]));
            if (max_tree < 2) max_tree = 2;
            continue;
        }
        // preorder word positions (left subtree first); pure leaves get none
        pos.assign(nn, -1);
        depth.assign(nn, 0);
        stack.clear();
        stack.push_back(0);
        int32_t next = 0, visited = 0;
        while (!stack.empty()) {
            const int32_t v = stack.back();
            stack.pop_back();
            if (v < 0 || v >= nn || pos[v] != -1 || ++visited > nn) {
                pk_set_error("pk_forest_create: tree %d is malformed at node %d", t, v);
                return nullptr;
            }
            if (depth[v] > max_depth) max_depth = depth[v];
            const int32_t l = left[base + v], r = right[base + v];
            if (l == -1) {
                pos[v] = (leaf_kind(p1[base + v]) == PK_KIND_LEAF) ? next++ : -2;
                continue;
            }
            pos[v] = next++;
            if (l < 0 || l >= nn || r < 0 || r >= nn) {
                pk_set_error("pk_forest_create: tree %d node %d has bad children", t, v);
                return nullptr;
            }
            depth[l] = depth[r] = depth[v] + 1;
            stack.push_back(r);
            stack.push_back(l);
        }
        const size_t tree_base = nodes.size();
        nodes.resize(tree_base + (size_t)next);
        for (int32_t v = 0; v < nn; v++) {
            if (pos[v] < 0) continue;  // unreachable node or pure leaf
            const int32_t l = left[base + v], r = right[base + v];
            if (l == -1) {
                nodes[tree_base + pos[v]] = leaf_word(p1[base + v]);
                continue;
            }
            const int32_t f = feat[base + v];
            if (f < 0 || f >= F) {
                pk_set_error("pk_forest_create: tree %d node %d feature %d out of range", t, v, f);
                return nullptr;
            }
            const unsigned lk = left[base + l] == -1 ? leaf_kind(p1[base + l]) : PK_KIND_NODE;
            const unsigned rk = left[base + r] == -1 ? leaf_kind(p1[base + r]) : PK_KIND_NODE;
            int64_t roff = 0;
            if (lk <= PK_KIND_LEAF && pos[l] != pos[v] + 1) roff = -1;
            if (rk <= PK_KIND_LEAF) {
                if (roff == 0) roff = (int64_t)pos[r] - pos[v];
                if (roff <= 0) roff = -1;
            }
            if (roff < 0) {
                pk_set_error("pk_forest_create: tree %d too large or inconsistent at node %d", t, v);
                return nullptr;
            }
            unsigned pk = (unsigned)f << PK_NODE_FEAT_SHIFT;
            if (miss_left && miss_left[base + v]) pk |= 1u << PK_NODE_MISS_BIT;
            pk |= lk << PK_NODE_LKIND_SHIFT;
            pk |= rk << PK_NODE_RKIND_SHIFT;
            if (roff >= PK_NODE_ROFF_BIG) {
                // rare: a tree of more than 8190 words keeps this offset in the side table
                if (big.size() < nodes.size()) big.resize(nodes.size(), 0);
                big[tree_base + pos[v]] = (int32_t)roff;
                tree_big[t] = 1;
                any_big = true;
                pk |= (unsigned)PK_NODE_ROFF_BIG << PK_NODE_ROFF_SHIFT;
            } else {
                pk |= (unsigned)roff << PK_NODE_ROFF_SHIFT;
            }
            const float t32 = floor_to_f32(thr[base + v]);
            unsigned tb;
            memcpy(&tb, &t32, 4);
            nodes[tree_base + pos[v]] = make_uint2(tb, pk);
        }
        if (next > max_tree) max_tree = next;
    }
    if (nodes.size() & 1) nodes.push_back(make_uint2(0, 0));
    root[T] = (int32_t)nodes.size();

    pk_forest *fo = new pk_forest();
    fo->device = device;
    fo->T = T;
    fo->F = F;
    fo->n_nodes = (int64_t)nodes.size();
    fo->max_depth = max_depth;
    fo->max_tree_words = max_tree;
    fo->nodes = nullptr;
    fo->root = nullptr;
    fo->big_roff = nullptr;
    fo->grp = nullptr;
    fo->n_grp = 0;
    fo->grp_words = fo->grp_slots = -1;
    fo->h_root = root;
    fo->h_big = tree_big;
    {
        const size_t nn = (size_t)tree_off[T];
        fo->h_tree_off.assign(tree_off, tree_off + T + 1);
        fo->h_left.assign(left, left + nn);
        fo->h_right.assign(right, right + nn);
        fo->h_feat.assign(feat, feat + nn);
        fo->h_thr.assign(thr, thr + nn);
        fo->h_p1.assign(p1, p1 + nn);
        if (miss_left) fo->h_miss.assign(miss_left, miss_left + nn);
    }
    if (hipMalloc((void **)&fo->nodes, (nodes.size() + 2) * sizeof(uint2)) != hipSuccess ||
        hipMalloc((void **)&fo->root, root.size() * sizeof(int32_t)) != hipSuccess) {
        pk_set_error("pk_forest_create: device allocation failed");
        pk_forest_destroy(fo);
        return nullptr;
    }
    if (hipMemcpy(fo->nodes, nodes.data(), nodes.size() * sizeof(uint2), hipMemcpyHostToDevice) !=
            hipSuccess ||
        hipMemcpy(fo->root, root.data(), root.size() * sizeof(int32_t), hipMemcpyHostToDevice) !=
            hipSuccess) {
        pk_set_error("pk_forest_create: upload failed");
        pk_forest_destroy(fo);
        return nullptr;
    }
    if (any_big) {
        big.resize(nodes.size() + 2, 0);
        if (hipMalloc((void **)&fo->big_roff, big.size() * sizeof(int32_t)) != hipSuccess ||
            hipMemcpy(fo->big_roff, big.data(), big.size() * sizeof(int32_t),
                      hipMemcpyHostToDevice) != hipSuccess) {
            pk_set_error("pk_forest_create: side-table upload failed");
            pk_forest_destroy(fo);
            return nullptr;
        }
    }
    return fo;
}

// Tree groups of the LDS forest kernel: consecutive whole trees, at most
// `slots` of them and at most `tree_words` words; a tree that is too large
// (or uses the side table) forms a group of its own that is not staged.
int pk_forest_groups(pk_forest *f, int tree_words, int slots)
{
    if (f->grp && f->grp_words == tree_words && f->grp_slots == slots) return PK_OK;
    std::vector<int32_t> first, staged;
    int t = 0;
    while (t < f->T) {
        first.push_back(t);
        const int g0 = f->h_root[t];
        const bool big = f->h_big[t] || f->h_root[t + 1] - g0 > tree_words;
        int t1 = t + 1;
        if (!big)
            while (t1 < f->T && t1 - t < slots && !f->h_big[t1] &&
                   f->h_root[t1 + 1] - g0 <= tree_words)
                t1++;
        staged.push_back(big ? 0 : 1);
        t = t1;
    }
    const int G = (int)first.size();
    first.push_back(f->T);
    first.push_back(f->T);  // one spare entry: the kernel reads grp[g + 2]
    std::vector<int32_t> tab(first);
    tab.insert(tab.end(), staged.begin(), staged.end());
    tab.push_back(0);
    if (f->grp) PK_HIP(hipFree(f->grp));
    f->grp = nullptr;
    PK_HIP(hipMalloc((void **)&f->grp, tab.size() * sizeof(int32_t)));
    PK_HIP(hipMemcpy(f->grp, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    f->n_grp = G;
    f->grp_words = tree_words;
    f->grp_slots = slots;
    return PK_OK;
}

extern "C" void pk_forest_destroy(pk_forest *f)
{
    PK_DEV_LOCK(f ? f->device : 0);
    if (!f) return;
    hipSetDevice(f->device);
    if (f->nodes) hipFree(f->nodes);
    if (f->root) hipFree(f->root);
    if (f->big_roff) hipFree(f->big_roff);
    if (f->grp) hipFree(f->grp);
    pk_forest_img_release(f);
    pk_forest_q_release(f);
    delete f;
}

extern "C" int pk_forest_info(const pk_forest *f, int *T, int *F, int64_t *n_nodes, int *max_depth)
{
    if (!f) return PK_E_INVALID;
    if (T) *T = f->T;
    if (F) *F = f->F;
    if (n_nodes) *n_nodes = f->n_nodes;
    if (max_depth) *max_depth = f->max_depth;
    return PK_OK;
}

// ------------------------------------------------------------------- matrix
extern "C" pk_matrix *pk_matrix_create(int device, int32_t n, const int32_t *indptr,
                                       const int32_t *indices, const double *data,
                                       const double *exp_arr, int32_t exp_len, int32_t dlo,
                                       int32_t dhi)
{
    PK_DEV_LOCK(device);
    if (n <= 0 || !indptr || !exp_arr || exp_len <= 0 || dhi < dlo) {
        pk_set_error("pk_matrix_create: bad arguments");
        return nullptr;
    }
    const int64_t nnz = indptr[n];
    if (nnz < 0 || indptr[0] != 0 || (nnz > 0 && (!indices || !data))) {
        pk_set_error("pk_matrix_create: malformed CSR");
        return nullptr;
    }
    pk_device_ctx *ctx = pk_ctx(device);
    if (!ctx) return nullptr;
    pk_matrix *m = new pk_matrix();
    m->device = device;
    m->n = n;
    m->dlo = dlo;
    m->dhi = dhi;
    m->ld = ((int64_t)n + 63) / 64 * 64;
    m->band = nullptr;
    m->exp_arr = nullptr;
    m->exp_len = exp_len;
    int32_t *d_indptr = nullptr, *d_indices = nullptr;
    double *d_data = nullptr;
    const size_t band_bytes = (size_t)(dhi - dlo + 1) * m->ld * sizeof(double);
    // room for the quotient band of the clean extractor right behind the raw band, when
    // 32-bit byte offsets reach both
    const bool with_norm = 2 * band_bytes < (1ull << 32) - 4096;
    bool ok = hipMalloc((void **)&m->band, with_norm ? 2 * band_bytes : band_bytes) == hipSuccess &&
              hipMalloc((void **)&m->exp_arr, sizeof(double) * (size_t)exp_len) == hipSuccess &&
              hipMalloc((void **)&d_indptr, sizeof(int32_t) * (size_t)(n + 1)) == hipSuccess &&
              hipMalloc((void **)&d_indices, sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)) == hipSuccess &&
              hipMalloc((void **)&d_data, sizeof(double) * (size_t)(nnz > 0 ? nnz : 1)) == hipSuccess;
    if (ok) {
        ok = hipMemcpyAsync(m->exp_arr, exp_arr, sizeof(double) * (size_t)exp_len,
                            hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
             hipMemcpyAsync(d_indptr, indptr, sizeof(int32_t) * (size_t)(n + 1),
                            hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
        if (ok && nnz > 0)
            ok = hipMemcpyAsync(d_indices, indices, sizeof(int32_t) * (size_t)nnz,
                                hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                 hipMemcpyAsync(d_data, data, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice,
                                ctx->stream) == hipSuccess;
        if (ok && with_norm) m->norm = m->band + band_bytes / sizeof(double);
        if (ok) ok = pk_launch_band_build(ctx, m, d_indptr, d_indices, d_data, nnz, 0) == PK_OK;
        if (ok) ok = hipStreamSynchronize(ctx->stream) == hipSuccess;
        if (!ok && !g_err[0]) pk_set_error("pk_matrix_create: upload / band build failed");
    } else {
        pk_set_error("pk_matrix_create: device allocation of %zu band bytes failed", band_bytes);
    }
    if (d_indptr) hipFree(d_indptr);
    if (d_indices) hipFree(d_indices);
    if (d_data) hipFree(d_data);
    if (!ok) {
        pk_matrix_destroy(m);
        return nullptr;
    }
    return m;
}

// ------------------------------------------- uploaded CSR and what is made from it
// scan of a pk_csr's entries (with its bias, if any): facts and validity flags
static bool csr_scan(pk_device_ctx *ctx, pk_csr *c, const char *who)
{
    unsigned long long *d_info = nullptr;
    bool ok = hipMalloc((void **)&c->valid_raw, (size_t)c->n) == hipSuccess &&
              hipMalloc((void **)&c->valid_bal, (size_t)c->n) == hipSuccess &&
              hipMalloc((void **)&d_info, 6 * sizeof(unsigned long long)) == hipSuccess;
    if (!ok) pk_set_error("%s: device allocation failed (%d bins)", who, c->n);
    if (ok) {
        ok = pk_launch_csr_info(ctx, c->indptr, c->indices, c->data, c->nnz, c->n, d_info, c->valid_raw, c->valid_bal,
                                c->bias, c->upper ? 1 : 0) == PK_OK &&
             hipMemcpyAsync(c->info, d_info, sizeof(c->info), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
        if (!ok && !g_err[0]) pk_set_error("%s: upload / scan failed", who);
    }
    if (d_info) hipFree(d_info);
    if (ok && c->upper && c->info[5]) {
        pk_set_error("%s: %llu pixels are out of order (columns must ascend strictly inside a row and none may lie "
                     "left of the diagonal); mirror the table on the host instead", who, c->info[5]);
        ok = false;
    }
    return ok;
}

static bool upload_bias(pk_device_ctx *ctx, pk_csr *c, const double *bias, const char *who)
{
    c->bias = nullptr;
    if (!bias) return true;
    if (hipMalloc((void **)&c->bias, sizeof(double) * (size_t)c->n) != hipSuccess ||
        hipMemcpyAsync(c->bias, bias, sizeof(double) * (size_t)c->n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        pk_set_error("%s: upload of the bias vector failed", who);
        return false;
    }
    return true;
}

extern "C" pk_csr *pk_csr_upload(int device, int32_t n, const int32_t *indptr, const int32_t *indices,
                                 const double *data)
{
    PK_DEV_LOCK(device);
    if (n <= 0 || !indptr || indptr[0] != 0 || indptr[n] < 0 || (indptr[n] > 0 && (!indices || !data))) {
        pk_set_error("pk_csr_upload: bad arguments / malformed CSR");
        return nullptr;
    }
    pk_device_ctx *ctx = pk_ctx(device);
    if (!ctx) return nullptr;
    pk_csr *c = new pk_csr();
    memset(c, 0, sizeof(*c));
    c->device = device;
    c->n = n;
    c->nnz = indptr[n];
    c->pix = new pk_pixels();
    memset(c->pix, 0, sizeof(*c->pix));
    c->pix->refs = 1;
    const size_t z = (size_t)(c->nnz > 0 ? c->nnz : 1);
    bool ok = hipMalloc((void **)&c->pix->indptr, sizeof(int32_t) * (size_t)(n + 1)) == hipSuccess &&
              hipMalloc((void **)&c->pix->indices, sizeof(int32_t) * z) == hipSuccess &&
              hipMalloc((void **)&c->pix->data, sizeof(double) * z) == hipSuccess;
    c->indptr = c->pix->indptr, c->indices = c->pix->indices, c->data = c->pix->data;
    if (!ok) pk_set_error("pk_csr_upload: device allocation failed (%lld entries)", (long long)c->nnz);
    if (ok) {
        ok = hipMemcpyAsync(c->indptr, indptr, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyHostToDevice,
                            ctx->stream) == hipSuccess;
        if (ok && c->nnz > 0)
            ok = hipMemcpyAsync(c->indices, indices, sizeof(int32_t) * z, hipMemcpyHostToDevice,
                                ctx->stream) == hipSuccess &&
                 hipMemcpyAsync(c->data, data, sizeof(double) * z, hipMemcpyHostToDevice, ctx->stream) ==
                     hipSuccess;
        if (!ok) pk_set_error("pk_csr_upload: upload failed");
        if (ok) ok = csr_scan(ctx, c, "pk_csr_upload");
    }
    if (!ok) {
        pk_csr_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" pk_csr *pk_csr_upload_upper(int device, int32_t n, const int32_t *indptr, const int32_t *cols,
                                       const void *counts, int counts_are_f64, const double *bias)
{
    PK_DEV_LOCK(device);
    if (n <= 0 || !indptr || indptr[0] != 0 || indptr[n] < 0 || (indptr[n] > 0 && (!cols || !counts))) {
        pk_set_error("pk_csr_upload_upper: bad arguments / malformed pixel table");
        return nullptr;
    }
    pk_device_ctx *ctx = pk_ctx(device);
    if (!ctx) return nullptr;
    pk_csr *c = new pk_csr();
    memset(c, 0, sizeof(*c));
    c->device = device;
    c->n = n;
    c->nnz = indptr[n];
    c->upper = true;
    c->pix = new pk_pixels();
    memset(c->pix, 0, sizeof(*c->pix));
    c->pix->refs = 1;
    const size_t z = (size_t)(c->nnz > 0 ? c->nnz : 1);
    int32_t *d_cnt = nullptr;
    bool ok = hipMalloc((void **)&c->pix->indptr, sizeof(int32_t) * (size_t)(n + 1)) == hipSuccess &&
              hipMalloc((void **)&c->pix->indices, sizeof(int32_t) * z) == hipSuccess &&
              hipMalloc((void **)&c->pix->data, sizeof(double) * z) == hipSuccess &&
              (counts_are_f64 || hipMalloc((void **)&d_cnt, sizeof(int32_t) * z) == hipSuccess);
    c->indptr = c->pix->indptr, c->indices = c->pix->indices, c->data = c->pix->data;
    if (!ok) pk_set_error("pk_csr_upload_upper: device allocation failed (%lld pixels)", (long long)c->nnz);
    if (ok) {
        ok = hipMemcpyAsync(c->indptr, indptr, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyHostToDevice,
                            ctx->stream) == hipSuccess;
        if (ok && c->nnz > 0) {
            ok = hipMemcpyAsync(c->indices, cols, sizeof(int32_t) * z, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
            if (ok && counts_are_f64)
                ok = hipMemcpyAsync(c->data, counts, sizeof(double) * z, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
            else if (ok)
                ok = hipMemcpyAsync(d_cnt, counts, sizeof(int32_t) * z, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                     pk_launch_counts_to_f64(ctx, d_cnt, c->data, c->nnz) == PK_OK;
        }
        if (!ok && !g_err[0]) pk_set_error("pk_csr_upload_upper: upload failed");
        if (ok) ok = upload_bias(ctx, c, bias, "pk_csr_upload_upper");
        if (ok) ok = csr_scan(ctx, c, "pk_csr_upload_upper");
    }
    if (d_cnt) {
        hipStreamSynchronize(ctx->stream);
        hipFree(d_cnt);
    }
    if (!ok) {
        pk_csr_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" pk_csr *pk_csr_view(pk_csr *src, const double *bias)
{
    PK_DEV_LOCK(src ? src->device : 0);
    if (!src || !src->pix) {
        pk_set_error("pk_csr_view: bad arguments");
        return nullptr;
    }
    pk_device_ctx *ctx = pk_ctx(src->device);
    if (!ctx) return nullptr;
    pk_csr *c = new pk_csr();
    memset(c, 0, sizeof(*c));
    c->device = src->device;
    c->n = src->n;
    c->nnz = src->nnz;
    c->upper = src->upper;
    c->pix = src->pix;
    c->pix->refs++;
    c->indptr = c->pix->indptr, c->indices = c->pix->indices, c->data = c->pix->data;
    if (!upload_bias(ctx, c, bias, "pk_csr_view") || !csr_scan(ctx, c, "pk_csr_view")) {
        pk_csr_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" void pk_csr_destroy(pk_csr *c)
{
    PK_DEV_LOCK(c ? c->device : 0);
    if (!c) return;
    hipSetDevice(c->device);
    void *ptrs[] = {c->valid_raw, c->valid_bal, c->bias};
    for (void *p : ptrs)
        if (p) hipFree(p);
    if (c->pix && --c->pix->refs == 0) {
        void *shared[] = {c->pix->indptr, c->pix->indices, c->pix->data};
        for (void *p : shared)
            if (p) hipFree(p);
        delete c->pix;
    }
    delete c;
}

extern "C" int pk_csr_info(const pk_csr *c, int64_t info[4], double *vmax)
{
    if (!c || !info) return PK_E_INVALID;
    for (int i = 0; i < 4; i++) info[i] = (int64_t)c->info[i];
    if (vmax) {
        double v;
        memcpy(&v, &c->info[4], 8);
        *vmax = v;
    }
    return PK_OK;
}

extern "C" pk_matrix *pk_matrix_from_csr(pk_csr *c, int32_t dlo, int32_t dhi, int keep_nan)
{
    PK_DEV_LOCK(c ? c->device : 0);
    if (!c || dhi < dlo) {
        pk_set_error("pk_matrix_from_csr: bad arguments");
        return nullptr;
    }
    pk_device_ctx *ctx = pk_ctx(c->device);
    if (!ctx) return nullptr;
    pk_matrix *m = new pk_matrix();
    m->device = c->device;
    m->n = c->n;
    m->dlo = dlo;
    m->dhi = dhi;
    m->ld = ((int64_t)c->n + 63) / 64 * 64;
    m->band = nullptr;
    m->exp_arr = nullptr;
    m->exp_len = 0;
    const size_t band_bytes = (size_t)(dhi - dlo + 1) * m->ld * sizeof(double);
    const bool with_norm = !keep_nan && 2 * band_bytes < (1ull << 32) - 4096;
    bool ok = hipMalloc((void **)&m->band, with_norm ? 2 * band_bytes : band_bytes) == hipSuccess;
    if (!ok) pk_set_error("pk_matrix_from_csr: device allocation of %zu band bytes failed", band_bytes);
    if (ok && with_norm) m->norm = m->band + band_bytes / sizeof(double);
    if (ok)
        ok = pk_launch_band_build(ctx, m, c->indptr, c->indices, c->data, c->nnz, keep_nan ? 2 : 1, c->bias,
                                  c->upper ? 1 : 0) == PK_OK &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
    if (!ok) {
        if (!g_err[0]) pk_set_error("pk_matrix_from_csr: band build failed");
        pk_matrix_destroy(m);
        return nullptr;
    }
    return m;
}

extern "C" int pk_matrix_set_expected(pk_matrix *m, const double *exp_arr, int32_t exp_len)
{
    PK_DEV_LOCK(m ? m->device : 0);
    if (!m || !exp_arr || exp_len <= 0) {
        pk_set_error("pk_matrix_set_expected: bad arguments");
        return PK_E_INVALID;
    }
    pk_device_ctx *ctx = pk_ctx(m->device);
    if (!ctx) return PK_E_NODEVICE;
    if (m->exp_arr) PK_HIP(hipFree(m->exp_arr));
    m->exp_arr = nullptr;
    PK_HIP(hipMalloc((void **)&m->exp_arr, sizeof(double) * (size_t)exp_len));
    PK_HIP(hipMemcpy(m->exp_arr, exp_arr, sizeof(double) * (size_t)exp_len, hipMemcpyHostToDevice));
    m->exp_len = exp_len;
    m->norm_tried = false;  // the quotient band depends on the expected values
    m->clean = false;
    return PK_OK;
}

extern "C" int pk_csr_expected_means(pk_csr *c, pk_matrix *band, int first, int top, int mode,
                                     double *means)
{
    PK_DEV_LOCK(c ? c->device : 0);
    if (!c || !band || !means || first < 0 || top < first || band->device != c->device ||
        band->n != c->n || first < band->dlo || top > band->dhi || top >= band->n) {
        pk_set_error("pk_csr_expected_means: bad arguments (diagonals %d..%d must lie in the band)", first, top);
        return PK_E_INVALID;
    }
    if (mode == 0 && c->info[3] != 0) {
        // the reference tests the SIGN of a column sum; with negative entries "holds a
        // positive entry" is not the same thing
        pk_set_error("pk_csr_expected_means: negative counts, use the host path");
        return PK_E_UNSUPPORTED;
    }
    pk_device_ctx *ctx = pk_ctx(c->device);
    if (!ctx) return PK_E_NODEVICE;
    const int nd = top - first + 1;
    double *d_scr = nullptr, *d_means = nullptr;
    int rc = PK_OK;
    if (hipMalloc((void **)&d_scr, (size_t)nd * band->ld * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&d_means, (size_t)nd * 8) != hipSuccess) {
        pk_set_error("pk_csr_expected_means: device allocation failed");
        rc = PK_E_NOMEM;
    }
    if (!rc)
        rc = pk_launch_expected_means(ctx, band, first, top, mode == 0 ? c->valid_raw : c->valid_bal,
                                      d_scr, d_means);
    if (!rc && (hipMemcpyAsync(means, d_means, (size_t)nd * 8, hipMemcpyDeviceToHost, ctx->stream) !=
                    hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)) {
        pk_set_error("pk_csr_expected_means: kernel / download failed");
        rc = PK_E_HIP;
    }
    if (d_scr) hipFree(d_scr);
    if (d_means) hipFree(d_means);
    return rc;
}

extern "C" void pk_matrix_destroy(pk_matrix *m)
{
    PK_DEV_LOCK(m ? m->device : 0);
    if (!m) return;
    hipSetDevice(m->device);
    if (m->band) hipFree(m->band);  // the quotient band lives in the same allocation
    if (m->exp_arr) hipFree(m->exp_arr);
    delete m;
}

// --------------------------------------------------------------- candidates
// Are consecutive candidates of a host list rarely neighbours on a diagonal?  (get_candidate's lists hold one
// band pixel in ~50; "every non-zero pixel of the band" lists are runs of neighbours.)  Looks at up to 8 192
// pairs spread over the list.  Decides in which order the extractor's lanes issue their loads -- never a result.
static int coords_scattered(int64_t N, const int32_t *x, const int32_t *y)
{
    if (N < 64 || !x || !y) return 0;
    const int64_t step = N / 8192 > 0 ? N / 8192 : 1;
    int64_t pairs = 0, next = 0;
    for (int64_t i = 0; i + 1 < N; i += step) {
        pairs++;
        next += (x[i + 1] == x[i] + 1 && y[i + 1] == y[i] + 1) ? 1 : 0;
    }
    return next * 4 < pairs ? 1 : 0;
}

// Are the list's batches of 32 consecutive candidates -- what a wave of the extractor takes -- runs on ONE
// diagonal within ~50 rows ("every non-zero pixel of the band": yes; a band with many empty pixels, a shuffled
// or strided list: no)?  Up to 2 048 batches spread over the list; nine in ten must be.  Decides whether the
// extractor stages a wave's strip of the band in LDS (one pass per batch) -- never a result.
static int coords_dense(int64_t N, const int32_t *x, const int32_t *y)
{
    if (N < 1 || !x || !y) return 0;
    const int64_t nb = (N + 31) / 32;
    const int64_t step = nb / 2048 > 0 ? nb / 2048 : 1;
    int64_t seen = 0, good = 0;
    for (int64_t b = 0; b < nb; b += step) {
        const int64_t i0 = b * 32, i1 = i0 + 32 < N ? i0 + 32 : N;
        bool ok = true;
        for (int64_t i = i0 + 1; i < i1 && ok; i++)
            ok = y[i] - x[i] == y[i0] - x[i0] && x[i] > x[i - 1] && x[i] - x[i0] <= 50;
        seen++;
        good += ok ? 1 : 0;
    }
    return good * 10 >= seen * 9 ? 1 : 0;
}

// (include/peakachu_hip.h: what the two samplers above say about a list, without a device)
extern "C" int pk_debug_classify_coords(int64_t N, const int32_t *x, const int32_t *y)
{
    if (N < 0 || (N > 0 && (!x || !y))) return PK_E_INVALID;
    return (coords_scattered(N, x, y) ? 1 : 0) | (coords_dense(N, x, y) ? 2 : 0);
}

extern "C" pk_cands *pk_cands_create(int device, int64_t N, const int32_t *x, const int32_t *y)
{
    PK_DEV_LOCK(device);
    if (N < 0 || (N > 0 && (!x || !y))) {
        pk_set_error("pk_cands_create: bad arguments");
        return nullptr;
    }
    pk_device_ctx *ctx = pk_ctx(device);
    if (!ctx) return nullptr;
    pk_cands *c = new pk_cands();
    memset(c, 0, sizeof(*c));
    c->opt = pk_default_options();  // (the memset took the defaults with it)
    c->device = device;
    c->N = N;
    const size_t n1 = (size_t)(N > 0 ? N : 1);
    c->n_batches_cap = 0;
    bool ok = hipMalloc((void **)&c->x, n1 * 4) == hipSuccess &&
              hipMalloc((void **)&c->y, n1 * 4) == hipSuccess &&
              hipMalloc((void **)&c->prob, n1 * 8) == hipSuccess &&
              hipMalloc((void **)&c->status, n1) == hipSuccess &&
              hipMalloc((void **)&c->ox, n1 * 4) == hipSuccess &&
              hipMalloc((void **)&c->oy, n1 * 4) == hipSuccess &&
              hipMalloc((void **)&c->op, n1 * 8) == hipSuccess &&
              hipMalloc((void **)&c->osig, n1 * 8) == hipSuccess &&
              hipMalloc((void **)&c->n_out_dev, 8) == hipSuccess;
    if (ok && N > 0)
        ok = hipMemcpy(c->x, x, n1 * 4, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(c->y, y, n1 * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        pk_set_error("pk_cands_create: device allocation / upload failed (N=%lld)", (long long)N);
        pk_cands_destroy(c);
        return nullptr;
    }
    c->scattered = coords_scattered(N, x, y);
    c->dense = coords_dense(N, x, y);
    return c;
}

extern "C" int pk_cands_set_prune(pk_cands *c, int on)
{
    PK_DEV_LOCK(c ? c->device : 0);
    if (!c) return PK_E_INVALID;
    c->prune = on != 0;
    return PK_OK;
}

extern "C" void pk_cands_destroy(pk_cands *c)
{
    PK_DEV_LOCK(c ? c->device : 0);
    if (!c) return;
    hipSetDevice(c->device);
    void *ptrs[] = {c->x, c->y, c->prob, c->status, c->ox, c->oy, c->op, c->osig, c->n_out_dev,
                    c->batch_cnt};
    for (void *p : ptrs)
        if (p) hipFree(p);
    delete c;
}

extern "C" int pk_expected_means(pk_matrix *m, int top, const uint8_t *valid, double *means)
{
    PK_DEV_LOCK(m ? m->device : 0);
    if (!m || !valid || !means || top < 0 || m->dlo != 0 || top > m->dhi || top >= m->n) {
        pk_set_error("pk_expected_means: bad arguments (needs a band with dlo = 0, dhi >= top)");
        return PK_E_INVALID;
    }
    pk_device_ctx *ctx = pk_ctx(m->device);
    if (!ctx) return PK_E_NODEVICE;
    uint8_t *d_valid = nullptr;
    double *d_scr = nullptr, *d_means = nullptr;
    int rc = PK_OK;
    const size_t scr = (size_t)(top + 1) * m->ld * sizeof(double);
    if (hipMalloc((void **)&d_valid, (size_t)m->n) != hipSuccess ||
        hipMalloc((void **)&d_scr, scr) != hipSuccess ||
        hipMalloc((void **)&d_means, (size_t)(top + 1) * 8) != hipSuccess) {
        pk_set_error("pk_expected_means: device allocation failed");
        rc = PK_E_NOMEM;
    }
    if (!rc && hipMemcpyAsync(d_valid, valid, (size_t)m->n, hipMemcpyHostToDevice, ctx->stream) !=
                   hipSuccess)
        rc = PK_E_HIP;
    if (!rc) rc = pk_launch_expected_means(ctx, m, 0, top, d_valid, d_scr, d_means);
    if (!rc && (hipMemcpyAsync(means, d_means, (size_t)(top + 1) * 8, hipMemcpyDeviceToHost,
                               ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)) {
        pk_set_error("pk_expected_means: kernel / download failed");
        rc = PK_E_HIP;
    }
    if (d_valid) hipFree(d_valid);
    if (d_scr) hipFree(d_scr);
    if (d_means) hipFree(d_means);
    return rc;
}

static pk_cands *cands_alloc(int device, int64_t N)
{
    pk_cands *c = new pk_cands();
    memset(c, 0, sizeof(*c));
    c->opt = pk_default_options();  // (the memset took the defaults with it)
    c->device = device;
    c->N = N;
    const size_t n1 = (size_t)(N > 0 ? N : 1);
    bool ok = hipMalloc((void **)&c->x, n1 * 4) == hipSuccess &&
              hipMalloc((void **)&c->y, n1 * 4) == hipSuccess &&
              hipMalloc((void **)&c->prob, n1 * 8) == hipSuccess &&
              hipMalloc((void **)&c->status, n1) == hipSuccess &&
              hipMalloc((void **)&c->ox, n1 * 4) == hipSuccess &&
              hipMalloc((void **)&c->oy, n1 * 4) == hipSuccess &&
              hipMalloc((void **)&c->op, n1 * 8) == hipSuccess &&
              hipMalloc((void **)&c->osig, n1 * 8) == hipSuccess &&
              hipMalloc((void **)&c->n_out_dev, 8) == hipSuccess;
    if (!ok) {
        pk_set_error("candidate buffers: device allocation failed (N=%lld)", (long long)N);
        pk_cands_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" pk_cands *pk_candidates_create(pk_matrix *raw, int lower, int upper,
                                          const int64_t *kstar, const double *bg,
                                          const double *weights, const double *mustar,
                                          int64_t n_mustar, int64_t *n_cand, int64_t *n_ambiguous)
{
    PK_DEV_LOCK(raw ? raw->device : 0);
    if (!raw || !bg || !n_cand || !n_ambiguous || lower < 0 || upper < lower ||
        (!weights && !kstar) || (weights && (!mustar || n_mustar <= 0))) {
        pk_set_error("pk_candidates_create: bad arguments");
        return nullptr;
    }
    if (lower < raw->dlo || upper > raw->dhi || upper >= raw->n) {
        pk_set_error("pk_candidates_create: diagonals %d..%d not inside the band %d..%d / n=%d",
                     lower, upper, raw->dlo, raw->dhi, raw->n);
        return nullptr;
    }
    pk_device_ctx *ctx = pk_ctx(raw->device);
    if (!ctx) return nullptr;
    const size_t nk = (size_t)upper + 1;
    int64_t *d_kstar = nullptr, *d_tot = nullptr;
    double *d_bg = nullptr, *d_w = nullptr, *d_ms = nullptr;
    pk_cands *out = nullptr;
    int64_t h[2] = {0, 0};
    bool ok = hipMalloc((void **)&d_bg, nk * 8) == hipSuccess &&
              hipMalloc((void **)&d_tot, 16) == hipSuccess &&
              hipMemcpyAsync(d_bg, bg, nk * 8, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
              hipMemsetAsync(d_tot, 0, 16, ctx->stream) == hipSuccess;
    if (ok && !weights)
        ok = hipMalloc((void **)&d_kstar, nk * 8) == hipSuccess &&
             hipMemcpyAsync(d_kstar, kstar, nk * 8, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    if (ok && weights)
        ok = hipMalloc((void **)&d_w, (size_t)raw->n * 8) == hipSuccess &&
             hipMalloc((void **)&d_ms, (size_t)n_mustar * 8) == hipSuccess &&
             hipMemcpyAsync(d_w, weights, (size_t)raw->n * 8, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
             hipMemcpyAsync(d_ms, mustar, (size_t)n_mustar * 8, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    if (!ok) pk_set_error("pk_candidates_create: table upload failed");
    if (ok) ok = pk_launch_candidates(ctx, raw, lower, upper, d_kstar, d_bg, d_w, d_ms, n_mustar,
                                      d_tot, d_tot + 1, nullptr, nullptr) == PK_OK;
    if (ok)
        ok = hipMemcpyAsync(h, d_tot, 16, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
    if (ok) {
        out = cands_alloc(raw->device, h[0]);
        if (out) out->scattered = 1;  // (the Poisson-filtered pixels of a band: one in tens)
        ok = out != nullptr;
    }
    if (ok && h[0] > 0)
        ok = pk_launch_candidates(ctx, raw, lower, upper, d_kstar, d_bg, d_w, d_ms, n_mustar, d_tot,
                                  d_tot + 1, out->x, out->y) == PK_OK &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
    void *tmp[] = {d_kstar, d_tot, d_bg, d_w, d_ms};
    for (void *p : tmp)
        if (p) hipFree(p);
    if (!ok) {
        if (out) pk_cands_destroy(out);
        if (!g_err[0]) pk_set_error("pk_candidates_create failed");
        return nullptr;
    }
    *n_cand = h[0];
    *n_ambiguous = h[1];
    return out;
}

extern "C" int pk_cands_fetch(pk_cands *cd, int32_t *x, int32_t *y)
{
    PK_DEV_LOCK(cd ? cd->device : 0);
    if (!cd) return PK_E_INVALID;
    pk_device_ctx *ctx = pk_ctx(cd->device);
    if (!ctx) return PK_E_NODEVICE;
    if (cd->N == 0) return PK_OK;
    if (x) PK_HIP(hipMemcpyAsync(x, cd->x, (size_t)cd->N * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (y) PK_HIP(hipMemcpyAsync(y, cd->y, (size_t)cd->N * 4, hipMemcpyDeviceToHost, ctx->stream));
    PK_HIP(hipStreamSynchronize(ctx->stream));
    return PK_OK;
}

// pk_score: the build requires 0 <= x <= y < n (what get_candidate produces).  pk_extract
// (getwindow) takes any coordinates, see classify_coords below.
static int check_coords(const char *who, int32_t n, int64_t N, const int32_t *x, const int32_t *y)
{
    // (block-wise and branch-free so that the compiler vectorises it: this runs over millions of
    // coordinates on the host inside pk_score's timed path)
    for (int64_t b = 0; b < N; b += 8192) {
        const int64_t e = b + 8192 < N ? b + 8192 : N;
        int bad = 0;
        for (int64_t i = b; i < e; i++) bad |= (x[i] < 0) | (y[i] < x[i]) | (y[i] >= n);
        if (bad)
            for (int64_t i = b; i < e; i++)
                if (x[i] < 0 || y[i] < x[i] || y[i] >= n) {
                    pk_set_error("%s: coordinate %lld = (%d, %d) violates 0 <= x <= y < n=%d", who,
                                 (long long)i, x[i], y[i], n);
                    return PK_E_INVALID;
                }
    }
    return PK_OK;
}

// getwindow's coordinates (peakachu/scoreUtils.py:70-93) are whatever the caller passes: the
// reference drops those with x-w < 0 or y+w+1 > n and gathers the rest with scipy's fancy
// indexing, which RAISES IndexError for a row x+w >= n or a column y-w < -n (possible only
// with x > y) and reads a negative column from the far end.  -> 0: all coordinates satisfy
// 0 <= x <= y < n (the fast kernels apply), 1: some do not (general kernel), < 0: the
// reference would raise.
static int classify_coords(int32_t n, int w, int64_t N, const int32_t *x, const int32_t *y)
{
    int any = 0;
    for (int64_t b = 0; b < N; b += 8192) {
        const int64_t e = b + 8192 < N ? b + 8192 : N;
        int odd = 0;
        for (int64_t i = b; i < e; i++) odd |= (x[i] < 0) | (y[i] < x[i]) | (y[i] >= n);
        if (!odd) continue;
        any = 1;
        for (int64_t i = b; i < e; i++) {
            const int64_t xi = x[i], yi = y[i];
            if (!(xi - w >= 0 && yi + w + 1 <= n)) continue;  // dropped by the reference's mask
            if (xi + w >= n || yi - w < -(int64_t)n) {
                pk_set_error("pk_extract: coordinate %lld = (%d, %d): window index out of range for n=%d "
                             "(IndexError in the reference)", (long long)i, x[i], y[i], n);
                return PK_E_INVALID;
            }
        }
    }
    return any;
}

// pk_score's coordinates are checked where they arrive (the host-side pass over 2 x 22 MB cost
// more than a millisecond in front of every call: a fifth of the scoring itself).  A coordinate
// outside 0 <= x <= y < n -- the contract of the fast extractors -- is replaced by (0, 0) (a window
// off the matrix, dropped by getwindow's mask: nothing can read out of bounds) and the first
// offender is recorded in word 65533 of the context's diagnostic buffer as 2^62 - index; pk_score_run
// reads the word with the result and refuses the call.
__global__ void coords_sanitize_kernel(int32_t *__restrict__ x, int32_t *__restrict__ y, int64_t cn, int32_t n,
                                       int64_t base, unsigned long long *__restrict__ err)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cn; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t xi = x[i], yi = y[i];
        if (xi < 0 || yi < xi || yi >= n) {
            x[i] = 0;
            y[i] = 0;
            atomicMax(err, (1ull << 62) - (unsigned long long)(base + i));
        }
    }
}

#ifdef PK_SCORE_TRACE
#include <chrono>
static double tr_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double tr_t0 = 0;
#define TR(tag) fprintf(stderr, "TR %-14s %9.1f\n", tag, tr_now() - tr_t0)
#else
#define TR(tag)
#endif
#ifndef PK_FIRST_UPLOAD
#define PK_FIRST_UPLOAD 262144  // candidates of pk_score's first upload chunk (the only exposed one)
#endif
#ifndef PK_UPLOAD_GROWTH
#define PK_UPLOAD_GROWTH 5
#endif
static int run_pipeline(pk_device_ctx *ctx, pk_matrix *m, pk_forest *f, pk_cands *cd, int w,
                        double prune_sum, double split_sum)
{
    const int F = (2 * w + 1) * (2 * w + 1);
    const int blk = pk_forest_plan_blk(f);
    if (blk <= 0) {
        pk_set_error("w=%d: feature tile does not fit LDS", w);
        return PK_E_UNSUPPORTED;
    }
    if (cd->opt.chunk < 64) {
        pk_set_error("candidate list without options (chunk = %lld): internal error", (long long)cd->opt.chunk);
        return PK_E_INVALID;
    }
    int64_t chunk = (cd->opt.chunk + blk - 1) / blk * blk;
    if (chunk > cd->N) chunk = (cd->N + blk - 1) / blk * blk;
    const size_t tile_floats = (size_t)chunk * F;
    const bool overlap = cd->opt.overlap != 0;
    int rc = pk_ctx_reserve_tiles(ctx, (overlap ? 2 : 1) * tile_floats * sizeof(float));
    if (rc) return rc;
    ctx->split_k = 0;  // (the cut forest's counters: one per launch of this call)
    ctx->split_n = 0;
    ctx->split_slack = 0;
    // the cut forest parks its candidates in the chunk's float tiles once they are quantized: not while
    // the extractor of the next chunk may be writing a tile buffer beside the forest
    if (overlap) split_sum = -INFINITY;
    if ((((w == 5 || w == 6) && m->opt.extract_pair) || (w == 11 && m->opt.extract_row16)) && m->opt.extract_clean) {
        rc = pk_matrix_prepare_norm(ctx, m);
        if (rc) return rc;
    }
    // Two tile buffers: extract(k+1) runs on the low-priority stream beside forest(k).
    // The forest kernel is LDS / latency bound and leaves ~40 % of the VALU issue slots
    // and (at <= 72 VGPRs) room for one 216-register extractor wave per SIMD, which is
    // FP64-VALU bound: the two kernels are complementary on a CU.
    hipStream_t st_ext = overlap ? ctx->stream2 : ctx->stream;
    if (overlap && cd->h_x != nullptr && cd->N > 0) {
        // pk_score lent host coordinates (they are NOT on the device yet) and the chunk-by-chunk
        // upload below is switched off in this mode: everything goes up front on the main
        // stream, ahead of the event the extractor's stream waits for
        PK_HIP(hipMemcpyAsync(cd->x, cd->h_x, (size_t)cd->N * 4, hipMemcpyHostToDevice, ctx->stream));
        PK_HIP(hipMemcpyAsync(cd->y, cd->h_y, (size_t)cd->N * 4, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(coords_sanitize_kernel, dim3(1024), dim3(256), 0, ctx->stream, cd->x, cd->y, cd->N, m->n,
                           (int64_t)0, reinterpret_cast<unsigned long long *>(ctx->dbg_buf + 65533));
    }
    if (overlap) {
        // whatever precedes on the main stream (uploads) must be visible to the extractor
        PK_HIP(hipEventRecord(ctx->ev_for[0], ctx->stream));
        PK_HIP(hipStreamWaitEvent(st_ext, ctx->ev_for[0], 0));
    }
    // pk_score hands host coordinates over: the upload of chunk k + 1 (second stream) overlaps the
    // kernels of chunk k.
    const bool stream_coords = cd->h_x != nullptr && !overlap && cd->N > 0;
    // Chunk sizes of a streamed call: the first one small (its upload is the exposed one), the second
    // PK_UPLOAD_GROWTH times as large -- an upload moves a candidate in 0.18 ns (8 bytes at ~45 GB/s,
    // measured through the pageable staging path), the kernels score one in 0.9-1.1 ns, so a chunk
    // five times its predecessor still arrives in time -- and the rest in equal launches of at most
    // `chunk` candidates (round 3-4 doubled from 256 Ki: five launches where four do, 80 us of small-
    // launch inefficiency per call of the bench workload; profiles/r04_pcie_timeline.log).
    std::vector<int64_t> sizes;
    if (stream_coords) {
        auto up = [&](int64_t v) { return (v + blk - 1) / blk * blk; };
        int64_t left = cd->N;
        for (int64_t want : {(int64_t)PK_FIRST_UPLOAD, (int64_t)PK_FIRST_UPLOAD * PK_UPLOAD_GROWTH}) {
            if (left <= 0) break;
            int64_t sz = up(want) < chunk ? up(want) : chunk;
            if (left - sz < sz / 2 && up(left) <= chunk) sz = up(left);  // no crumb behind it
            sizes.push_back(sz);
            left -= sz;
        }
        if (left > 0) {
            const int64_t n = (left + chunk - 1) / chunk;
            const int64_t each = up((left + n - 1) / n);
            for (int64_t i = 0; i < n; i++) sizes.push_back(each < chunk ? each : chunk);
        }
    }
    auto span = [&](int64_t k_) -> int64_t {  // candidates of chunk k_
        return (stream_coords && k_ < (int64_t)sizes.size()) ? sizes[(size_t)k_] : chunk;
    };
    auto upload = [&](int64_t c0, int64_t k_) -> int {
        const int64_t cn = cd->N - c0 < span(k_) ? cd->N - c0 : span(k_);
        // chunk 0 is the exposed one: on the kernels' own stream (no event hop in front of the first
        // extractor); the others on the second stream, beside the kernels of their predecessor
        hipStream_t su = k_ == 0 ? ctx->stream : ctx->stream2;
        PK_HIP(hipMemcpyAsync(cd->x + c0, cd->h_x + c0, (size_t)cn * 4, hipMemcpyHostToDevice, su));
        PK_HIP(hipMemcpyAsync(cd->y + c0, cd->h_y + c0, (size_t)cn * 4, hipMemcpyHostToDevice, su));
        // the range check rides right behind the copy
        hipLaunchKernelGGL(coords_sanitize_kernel, dim3((unsigned)((cn + 2047) / 2048 < 1024 ? (cn + 2047) / 2048 : 1024)),
                           dim3(256), 0, su, cd->x + c0, cd->y + c0, cn, m->n, c0,
                           reinterpret_cast<unsigned long long *>(ctx->dbg_buf + 65533));
        if (k_ != 0) PK_HIP(hipEventRecord(ctx->ev_ext[k_ & 1], ctx->stream2));
        return PK_OK;
    };
    TR("pipe:pre");
    if (stream_coords) {
        rc = upload(0, 0);
        if (rc) return rc;
    }
    TR("pipe:up0");
    int64_t k = 0;
    for (int64_t c0 = 0; c0 < cd->N; c0 += span(k), k++) {
        const int64_t cn = cd->N - c0 < span(k) ? cd->N - c0 : span(k);
        const int buf = overlap ? (int)(k & 1) : 0;
        float *tiles = ctx->fea_tiles + (size_t)buf * tile_floats;
        if (overlap && k >= 2)  // forest(k-2) must be done with this buffer
            PK_HIP(hipStreamWaitEvent(st_ext, ctx->ev_for[buf], 0));
        if (stream_coords && k != 0) PK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_ext[k & 1], 0));
        // rank kernels: the float tiles are an intermediate of this chunk only (extractor ->
        // quantizer); made and consumed piece by piece through the start of the buffer they
        // never have to leave the Infinity Cache
        int64_t sub = (cd->opt.sub_chunk + blk - 1) / blk * blk;
        const bool pieces = !overlap && sub > 0 && sub < cn && f->plan_kind == 2 && f->q_state == 1 &&
                            blk == (f->q_ch == 1 ? 128 : 128 * PK_Q_FTILE);
        if (pieces) {
            rc = pk_forest_q_reserve(ctx, f, cn);
            for (int64_t s0 = 0; !rc && s0 < cn; s0 += sub) {
                const int64_t sn = cn - s0 < sub ? cn - s0 : sub;
                rc = pk_launch_extract(ctx, st_ext, m, w, cd->x, cd->y, c0 + s0, sn, tiles, blk, cd->status,
                                       nullptr, false, cd->scattered > 0, cd->dense > 0);
                if (!rc) rc = pk_launch_quant_q(ctx, ctx->stream, f, tiles, s0 / 128, sn);
            }
            if (!rc)
                rc = pk_launch_forest_q_walk(ctx, f, cd->status, c0, cn, cd->prob, prune_sum, split_sum, tiles,
                                             tile_floats * sizeof(float));
            if (rc) return rc;
            if (stream_coords && c0 + cn < cd->N) {
                rc = upload(c0 + cn, k + 1);
                if (rc) return rc;
            }
            continue;
        }
        rc = pk_launch_extract(ctx, st_ext, m, w, cd->x, cd->y, c0, cn, tiles, blk, cd->status,
                               nullptr, false, cd->scattered > 0, cd->dense > 0);
        if (rc) return rc;
        if (overlap) {
            PK_HIP(hipEventRecord(ctx->ev_ext[buf], st_ext));
            PK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_ext[buf], 0));
        }
        rc = pk_launch_forest(ctx, f, tiles, blk, cd->status, c0, cn, cd->prob, prune_sum, split_sum);
        if (rc) return rc;
        if (overlap) PK_HIP(hipEventRecord(ctx->ev_for[buf], ctx->stream));
        if (stream_coords && c0 + cn < cd->N) {
            rc = upload(c0 + cn, k + 1);
            if (rc) return rc;
        }
    }
    return PK_OK;
}

// {n_out, status words} and -- with_records, when they fit -- the scored pixels into one buffer
__global__ void __launch_bounds__(256) ret_pack_kernel(const int64_t *__restrict__ n_out, const long long *__restrict__ dbg3,
                                                       const int32_t *__restrict__ ox, const int32_t *__restrict__ oy,
                                                       const double *__restrict__ op, const double *__restrict__ os,
                                                       char *__restrict__ ret, int with_records,
                                                       const unsigned *__restrict__ split_cnt, int split_k)
{
    const int64_t n = *n_out;
    if (blockIdx.x == 0 && threadIdx.x < 4)
        reinterpret_cast<long long *>(ret)[threadIdx.x] = threadIdx.x == 0 ? (long long)n : dbg3[threadIdx.x - 1];
    if (blockIdx.x == 0 && threadIdx.x == 4 && split_k > 0) {
        // what the cut forest's launches of this call parked (fifth word of the header)
        long long parked = 0;
        for (int i = 0; i < split_k; i++) parked += split_cnt[i];
        reinterpret_cast<long long *>(ret)[4] = parked;
    }
    if (!with_records || n > PK_RET_INLINE) return;
    int32_t *rx = reinterpret_cast<int32_t *>(ret + PK_RET_HEAD), *ry = rx + PK_RET_INLINE;
    double *rp = reinterpret_cast<double *>(ry + PK_RET_INLINE), *rs = rp + PK_RET_INLINE;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        rx[i] = ox[i];
        ry[i] = oy[i];
        rp[i] = op[i];
        rs[i] = os[i];
    }
}

static int score_run_impl(pk_matrix *m, pk_forest *f, pk_cands *cd, int w, double thre, int64_t batch, int64_t *n_out,
                          bool with_records);
extern "C" int pk_score_run(pk_matrix *m, pk_forest *f, pk_cands *cd, int w, double thre,
                            int64_t batch, int64_t *n_out)
{
    return score_run_impl(m, f, cd, w, thre, batch, n_out, false);
}
static int score_run_impl(pk_matrix *m, pk_forest *f, pk_cands *cd, int w, double thre, int64_t batch, int64_t *n_out,
                          bool with_records)
{
    PK_DEV_LOCK(m ? m->device : 0);
    if (!m || !f || !cd || w < 1) {
        pk_set_error("pk_score_run: bad arguments");
        return PK_E_INVALID;
    }
    if (m->device != f->device || m->device != cd->device) {
        pk_set_error("pk_score_run: handles live on different devices");
        return PK_E_INVALID;
    }
    const int F = (2 * w + 1) * (2 * w + 1);
    if (f->F != F) {
        pk_set_error("pk_score_run: forest has %d features, w=%d needs %d", f->F, w, F);
        return PK_E_INVALID;
    }
    if (batch <= 0) batch = 100000;
    pk_device_ctx *ctx = pk_ctx(m->device);
    if (!ctx) return PK_E_NODEVICE;
    const int64_t nb = (cd->N + batch - 1) / batch + 1;
    if (nb > cd->n_batches_cap) {
        if (cd->batch_cnt) PK_HIP(hipFree(cd->batch_cnt));
        cd->batch_cnt = nullptr;
        PK_HIP(hipMalloc((void **)&cd->batch_cnt, sizeof(int32_t) * (size_t)nb));
        cd->n_batches_cap = nb;
    }
    // Exact early termination: a candidate stops walking once its sum can no longer exceed thre * T.
    // Option early_exit = 1 forces it; pk_cands_set_prune ALLOWS it and the library applies it where it
    // pays -- measured on config 2 (profiles/r05_prune_ab.log): at thre = 0.5 no wave can stop before
    // half the forest and the kernel's extra test costs 1.5-4 %; from thre = 0.6 on lists of at least
    // ~0.5 M candidates (two tiles per workgroup and more) it saves 7 % (0.6) to 35 % (0.9) of the
    // forest's time; a list of one or two tiles per workgroup never gains.  Only meaningful for thre >= 0.
    const bool prune_on = thre >= 0.0 && (cd->opt.early_exit || (cd->prune && thre >= 0.55 && cd->N >= (int64_t)1 << 19));
    // (the bound every kernel tests against: thre * T less a margin that is PROVEN to cover the rounding of
    // the additions still to come -- pk_prune_bound, pk_common.h; q_T: the rank image's trees, more than T
    // when trees were cut into pieces)
    const double exit_bound = pk_prune_bound(thre, f->T, f->q_T > f->T ? f->q_T : f->T);
    const double prune_sum = prune_on ? exit_bound : -INFINITY;
    // The same permission, per CANDIDATE: the default forest kernel can be cut in two at a tree-group
    // boundary -- the head over everybody, the tail over the candidates whose sum could still exceed
    // thre * T -- which pays from the default threshold on (pk_forest_q.hip, q_pick_cut; long launches only)
    const double split_sum = thre >= 0.0 && (cd->opt.early_exit || cd->prune) ? exit_bound : -INFINITY;
    TR("run:enter");
    // A call that fails half-way may have left the record of an out-of-contract coordinate (word 65533
    // of the diagnostic buffer, coords_sanitize_kernel) behind: it belongs to THAT call and must not
    // be read by the next one on this device
    auto forget_offender = [&](int code) {
        hipStreamSynchronize(ctx->stream2);
        hipStreamSynchronize(ctx->stream);
        hipMemset(ctx->dbg_buf + 65533, 0, sizeof(long long));
        return code;
    };
    int rc = run_pipeline(ctx, m, f, cd, w, prune_sum, split_sum);
    if (rc) return forget_offender(rc);
    TR("run:launched");
    bool reply_packed = false;
    rc = pk_launch_compact(ctx, m, cd, thre, batch, with_records ? 1 : 0, &reply_packed);
    if (rc) return forget_offender(rc);
    // One copy into pinned memory brings back the count, words 65533 (first coordinate outside
    // pk_score's contract, see coords_sanitize_kernel), 65534 (a sink that keeps the kernels' warm-up
    // loads alive) and 65535 (the forest kernels' error word) of the diagnostic buffer and -- for
    // pk_score -- the scored pixels themselves when there are at most PK_RET_INLINE of them (rounds
    // 1-4: two pageable copies here and four more in pk_score_fetch, 25 us each).
    cd->ret_inline = false;
    if (!reply_packed) {  // (a short list's compaction has packed the reply itself)
        hipLaunchKernelGGL(ret_pack_kernel, dim3(with_records ? 32 : 1), dim3(256), 0, ctx->stream, cd->n_out_dev,
                           ctx->dbg_buf + 65533, cd->ox, cd->oy, cd->op, cd->osig, ctx->d_ret, with_records ? 1 : 0,
                           ctx->split_cnt, ctx->split_cnt ? ctx->split_k : 0);
        PK_HIP(hipGetLastError());
    }
    PK_HIP(hipMemcpyAsync(ctx->h_ret, ctx->d_ret, with_records ? PK_RET_BYTES : PK_RET_HEAD, hipMemcpyDeviceToHost, ctx->stream));
    TR("run:enq");
    PK_HIP(hipStreamSynchronize(ctx->stream));
    TR("run:synced");
    long long dbg3[3];
    memcpy(&cd->n_out, ctx->h_ret, 8);
    memcpy(dbg3, ctx->h_ret + 8, 24);
    if (ctx->split_k > 0) {
        long long parked = 0;
        memcpy(&parked, ctx->h_ret + 32, 8);
        pk_forest_cut_feedback(f, ctx->split_n, parked, ctx->split_slack);
    }
    cd->ret_inline = with_records && cd->n_out <= PK_RET_INLINE;
    const long long err = dbg3[2];
    if (dbg3[0]) {
        PK_HIP(hipMemset(ctx->dbg_buf + 65533, 0, sizeof(long long)));
        // only a call that streamed host coordinates through the check can have an offender of its
        // own (a resident list was checked when it was made); anything else is a stale record
        if (cd->h_x) {
            const long long i = (long long)((1ull << 62) - (unsigned long long)dbg3[0]);
            if (i >= 0 && i < cd->N)
                pk_set_error("pk_score: coordinate %lld = (%d, %d) violates 0 <= x <= y < n=%d", i, cd->h_x[i], cd->h_y[i], m->n);
            else
                pk_set_error("pk_score: coordinate %lld violates 0 <= x <= y < n=%d", i, m->n);
            return PK_E_INVALID;
        }
    }
    if (err) {
        PK_HIP(hipMemset(ctx->dbg_buf + 65535, 0, sizeof(long long)));
        pk_set_error("forest kernel raised its error word (%lld): 1 = LDS ring wait timed out, "
                     "2 = feature tile not at LDS offset 0 (internal error)", err);
        return PK_E_HIP;
    }
    if (n_out) *n_out = cd->n_out;
    return PK_OK;
}

extern "C" int pk_score_fetch(pk_cands *cd, int32_t *ox, int32_t *oy, double *op, double *osignal)
{
    PK_DEV_LOCK(cd ? cd->device : 0);
    if (!cd) return PK_E_INVALID;
    pk_device_ctx *ctx = pk_ctx(cd->device);
    if (!ctx) return PK_E_NODEVICE;
    const size_t k = (size_t)cd->n_out;
    if (k == 0) return PK_OK;
    if (ox) PK_HIP(hipMemcpyAsync(ox, cd->ox, k * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (oy) PK_HIP(hipMemcpyAsync(oy, cd->oy, k * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (op) PK_HIP(hipMemcpyAsync(op, cd->op, k * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (osignal) PK_HIP(hipMemcpyAsync(osignal, cd->osig, k * 8, hipMemcpyDeviceToHost, ctx->stream));
    PK_HIP(hipStreamSynchronize(ctx->stream));
    return PK_OK;
}

extern "C" int pk_score_fetch_all(pk_cands *cd, uint8_t *status, double *prob)
{
    PK_DEV_LOCK(cd ? cd->device : 0);
    if (!cd) return PK_E_INVALID;
    pk_device_ctx *ctx = pk_ctx(cd->device);
    if (!ctx) return PK_E_NODEVICE;
    if (cd->N == 0) return PK_OK;
    if (status)
        PK_HIP(hipMemcpyAsync(status, cd->status, (size_t)cd->N, hipMemcpyDeviceToHost, ctx->stream));
    if (prob)
        PK_HIP(hipMemcpyAsync(prob, cd->prob, (size_t)cd->N * 8, hipMemcpyDeviceToHost, ctx->stream));
    PK_HIP(hipStreamSynchronize(ctx->stream));
    return PK_OK;
}

extern "C" int pk_score(pk_matrix *m, pk_forest *f, int w, double thre, int64_t batch, int64_t N,
                        const int32_t *x, const int32_t *y, int32_t *ox, int32_t *oy, double *op,
                        double *osignal, int64_t *n_out)
{
    PK_DEV_LOCK(m ? m->device : 0);
    if (!m || !f || !n_out) {
        pk_set_error("pk_score: bad arguments");
        return PK_E_INVALID;
    }
#ifdef PK_SCORE_TRACE
    { const double n_ = tr_now(); fprintf(stderr, "TR between-calls %9.1f\n", n_ - tr_t0); tr_t0 = n_; }
#endif
    int rc = PK_OK;
    // the device-side candidate list of the host-buffer convenience call is kept per device
    // and reused while it is large enough: nine allocations per call cost more than the
    // upload of the coordinates
    pk_device_ctx *ctx = pk_ctx(m->device);
    if (!ctx) return PK_E_NODEVICE;
    pk_cands *cd = ctx->score_cands;
    bool deferred = false;
    if (cd && ctx->score_cands_cap >= N) {
        cd->N = N;
        // the coordinates travel chunk by chunk inside run_pipeline, behind the kernels
        cd->h_x = x;
        cd->h_y = y;
        deferred = true;
    } else {
        // (a new list is uploaded whole by pk_cands_create: checked on the host, once)
        rc = check_coords("pk_score", m->n, N, x, y);
        if (rc) return rc;
        if (cd) pk_cands_destroy(cd);
        ctx->score_cands = nullptr;
        ctx->score_cands_cap = 0;
        cd = pk_cands_create(m->device, N, x, y);
        if (!cd) return PK_E_HIP;
        ctx->score_cands = cd;
        ctx->score_cands_cap = N;
    }
    cd->opt = m->opt;  // (a call without a candidate handle: the matrix handle's pipeline options)
    cd->scattered = coords_scattered(N, x, y);
    cd->dense = coords_dense(N, x, y);
    cd->prune = 1;     // pk_score hands back the scored pixels only: what a decided candidate's probability reads is invisible
    rc = score_run_impl(m, f, cd, w, thre, batch, n_out, true);
    if (deferred) {
        cd->h_x = cd->h_y = nullptr;  // borrowed for this call only
        if (rc) {  // no copy may still read the caller's buffers
            hipStreamSynchronize(ctx->stream2);
            hipStreamSynchronize(ctx->stream);
        }
    }
    if (!rc && cd->ret_inline) {  // the pixels came back with the count
        const size_t k = (size_t)cd->n_out;
        const char *r = ctx->h_ret + PK_RET_HEAD;
        if (ox) memcpy(ox, r, k * 4);
        if (oy) memcpy(oy, r + (size_t)PK_RET_INLINE * 4, k * 4);
        if (op) memcpy(op, r + (size_t)PK_RET_INLINE * 8, k * 8);
        if (osignal) memcpy(osignal, r + (size_t)PK_RET_INLINE * 16, k * 8);
    } else if (!rc)
        rc = pk_score_fetch(cd, ox, oy, op, osignal);
    TR("score:fetched");
    return rc;
}

// ------------------------------------------------------------ getwindow API
extern "C" int pk_extract(pk_matrix *m, int w, int64_t N, const int32_t *x, const int32_t *y,
                          double *fea64, float *fea32, int64_t *keep, int64_t *n_keep)
{
    PK_DEV_LOCK(m ? m->device : 0);
    if (!m || !keep || !n_keep || N < 0 || w < 1 || w > 15) {
        pk_set_error("pk_extract: bad arguments");
        return PK_E_INVALID;
    }
    *n_keep = 0;
    if (N == 0) return PK_OK;
    int rc = classify_coords(m->n, w, N, x, y);
    if (rc < 0) return rc;
    const bool any_coords = rc == 1;
    rc = PK_OK;
    pk_device_ctx *ctx = pk_ctx(m->device);
    if (!ctx) return PK_E_NODEVICE;
    const int F = (2 * w + 1) * (2 * w + 1);
    int blk = pk_forest_tile_width(F, m->opt);
    if (blk <= 0) blk = 64;
    pk_cands *cd = pk_cands_create(m->device, N, x, y);
    if (!cd) return PK_E_HIP;
    // bounded staging: rows of float64 features for one chunk at a time
    int64_t chunk = 65536 / blk * blk;
    if (chunk > N) chunk = (N + blk - 1) / blk * blk;
    double *d_rows = nullptr;
    std::vector<double> h_rows((size_t)chunk * F);
    std::vector<uint8_t> h_status((size_t)chunk);
    rc = pk_ctx_reserve_tiles(ctx, (size_t)chunk * F * sizeof(float));
    if (!rc && (((w == 5 || w == 6) && m->opt.extract_pair) || (w == 11 && m->opt.extract_row16)) &&
        m->opt.extract_clean)
        rc = pk_matrix_prepare_norm(ctx, m);
    if (!rc) {  // the staging rows live in the context and only ever grow: no allocation per call
        const size_t need = (size_t)chunk * F * sizeof(double);
        if (need > ctx->rows64_bytes) {
            if (ctx->rows64) hipFree(ctx->rows64);
            ctx->rows64 = nullptr;
            ctx->rows64_bytes = 0;
            if (hipMalloc((void **)&ctx->rows64, need) != hipSuccess) {
                pk_set_error("pk_extract: staging allocation failed");
                rc = PK_E_NOMEM;
            } else {
                ctx->rows64_bytes = need;
            }
        }
        d_rows = ctx->rows64;
    }
    int64_t nk = 0;
    for (int64_t c0 = 0; !rc && c0 < N; c0 += chunk) {
        const int64_t cn = N - c0 < chunk ? N - c0 : chunk;
        rc = pk_launch_extract(ctx, ctx->stream, m, w, cd->x, cd->y, c0, cn, ctx->fea_tiles, blk,
                               cd->status, d_rows, any_coords, false, cd->dense > 0);
        if (rc) break;
        if (hipMemcpyAsync(h_rows.data(), d_rows, (size_t)cn * F * 8, hipMemcpyDeviceToHost,
                           ctx->stream) != hipSuccess ||
            hipMemcpyAsync(h_status.data(), cd->status + c0, (size_t)cn, hipMemcpyDeviceToHost,
                           ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            pk_set_error("pk_extract: download failed");
            rc = PK_E_HIP;
            break;
        }
        for (int64_t i = 0; i < cn; i++) {
            if (!h_status[(size_t)i]) continue;
            const double *src = h_rows.data() + (size_t)i * F;
            if (fea64) memcpy(fea64 + (size_t)nk * F, src, sizeof(double) * F);
            if (fea32)
                for (int q = 0; q < F; q++) fea32[(size_t)nk * F + q] = (float)src[q];
            keep[nk++] = c0 + i;
        }
    }
    pk_cands_destroy(cd);
    if (!rc) *n_keep = nk;
    return rc;
}

// --------------------------------------------------------- predict_proba API
extern "C" int pk_predict(pk_forest *f, int64_t N, const float *fea32, double *p1)
{
    PK_DEV_LOCK(f ? f->device : 0);
    if (!f || N < 0 || (N > 0 && (!fea32 || !p1))) {
        pk_set_error("pk_predict: bad arguments");
        return PK_E_INVALID;
    }
    if (N == 0) return PK_OK;
    pk_device_ctx *ctx = pk_ctx(f->device);
    if (!ctx) return PK_E_NODEVICE;
    const int F = f->F;
    const int blk = pk_forest_plan_blk(f);
    if (blk <= 0) {
        pk_set_error("pk_predict: F=%d does not fit an LDS tile", F);
        return PK_E_UNSUPPORTED;
    }
    int64_t chunk = (f->opt.chunk + blk - 1) / blk * blk;
    if (chunk > N) chunk = (N + blk - 1) / blk * blk;
    int rc = pk_ctx_reserve_tiles(ctx, (size_t)chunk * F * sizeof(float));
    if (rc) return rc;
    float *d_rows = nullptr;
    double *d_prob = nullptr;
    uint8_t *d_status = nullptr;
    if (hipMalloc((void **)&d_rows, (size_t)chunk * F * 4) != hipSuccess ||
        hipMalloc((void **)&d_prob, (size_t)chunk * 8) != hipSuccess ||
        hipMalloc((void **)&d_status, (size_t)chunk) != hipSuccess) {
        pk_set_error("pk_predict: device allocation failed");
        rc = PK_E_NOMEM;
    }
    for (int64_t c0 = 0; !rc && c0 < N; c0 += chunk) {
        const int64_t cn = N - c0 < chunk ? N - c0 : chunk;
        if (hipMemcpyAsync(d_rows, fea32 + (size_t)c0 * F, (size_t)cn * F * 4,
                           hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            pk_set_error("pk_predict: upload failed");
            rc = PK_E_HIP;
            break;
        }
        rc = pk_launch_tile_rows(ctx, d_rows, cn, F, ctx->fea_tiles, blk, d_status);
        if (rc) break;
        rc = pk_launch_forest(ctx, f, ctx->fea_tiles, blk, d_status, 0, cn, d_prob, -INFINITY);
        if (rc) break;
        if (hipMemcpyAsync(p1 + c0, d_prob, (size_t)cn * 8, hipMemcpyDeviceToHost, ctx->stream) !=
                hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            pk_set_error("pk_predict: download failed");
            rc = PK_E_HIP;
        }
    }
    if (d_rows) hipFree(d_rows);
    if (d_prob) hipFree(d_prob);
    if (d_status) hipFree(d_status);
    return rc;
}
