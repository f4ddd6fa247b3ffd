// pk_hostio.hip -- host-side helper of the contact-map reader (no device code).
//
// The reference reads a .cool through cooler -> h5py -> libhdf5, whose C filter pipeline
// inflates and un-shuffles the chunks of the pixel table (peakachu/score_genome.py:55-57).
// peakachu_amd/h5lite.py parses the container in Python; what it cannot do at a useful rate
// is the byte un-shuffle of a chunk (a strided transpose: 40 ms per 8 MB chunk through numpy,
// three times its inflate, with the interpreter lock held), so the chunk pipeline of a ranged
// read runs here: inflate (zlib's `uncompress`, resolved at run time from the libz the
// interpreter itself is linked against -- no build-time dependency), un-shuffle, and the copy
// of the wanted slice to its place, chunks side by side on host threads.
//
// No device code and no HIP call: the file needs the public header and pk_set_error only, so the
// CPU test suite builds it with plain g++ under Address / UndefinedBehaviour / Thread sanitizers
// (tests/test_hostio_sanitized.py, tests/native/test_hostio.cpp).
#include <dlfcn.h>

#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/peakachu_hip.h"

void pk_set_error(const char *fmt, ...);  // pk_api.hip (the test harness brings its own)

namespace {

typedef int (*uncompress_fn)(unsigned char *dest, unsigned long *dest_len, const unsigned char *src,
                             unsigned long src_len);
std::once_flag g_z_once;
uncompress_fn g_uncompress = nullptr;

void find_zlib()
{
    for (const char *name : {"libz.so.1", "libz.so"}) {
        if (void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
            g_uncompress = reinterpret_cast<uncompress_fn>(dlsym(h, "uncompress"));
            if (g_uncompress) return;
        }
    }
}

// bytes [skip, skip + take) of the un-shuffled chunk: element e, byte k lies at src[k * n_el + e]
void unshuffle_slice(const unsigned char *src, int64_t n_el, int es, int64_t skip, int64_t take, unsigned char *dst)
{
    const int64_t e0 = skip / es, e1 = (skip + take) / es;
    if (es == 8) {
        const unsigned char *p0 = src, *p1 = src + n_el, *p2 = src + 2 * n_el, *p3 = src + 3 * n_el,
                            *p4 = src + 4 * n_el, *p5 = src + 5 * n_el, *p6 = src + 6 * n_el, *p7 = src + 7 * n_el;
        // (memcpy stores: dst[i] carries no alignment requirement -- a slice may start anywhere in the
        // caller's array; the compiler turns them into plain 8-byte moves)
        for (int64_t e = e0; e < e1; e++) {
            const uint64_t v = (uint64_t)p0[e] | (uint64_t)p1[e] << 8 | (uint64_t)p2[e] << 16 | (uint64_t)p3[e] << 24 |
                               (uint64_t)p4[e] << 32 | (uint64_t)p5[e] << 40 | (uint64_t)p6[e] << 48 | (uint64_t)p7[e] << 56;
            memcpy(dst + (e - e0) * 8, &v, 8);
        }
    } else if (es == 4) {
        const unsigned char *p0 = src, *p1 = src + n_el, *p2 = src + 2 * n_el, *p3 = src + 3 * n_el;
        for (int64_t e = e0; e < e1; e++) {
            const uint32_t v = (uint32_t)p0[e] | (uint32_t)p1[e] << 8 | (uint32_t)p2[e] << 16 | (uint32_t)p3[e] << 24;
            memcpy(dst + (e - e0) * 4, &v, 4);
        }
    } else {
        for (int64_t e = e0; e < e1; e++)
            for (int k = 0; k < es; k++) dst[(e - e0) * es + k] = src[(int64_t)k * n_el + e];
    }
}

}  // namespace

// Chunk i is src[i] (src_len[i] bytes as stored in the file).  Read pipeline: inflate to chunk_bytes
// when `deflate`; un-shuffle with element size shuffle_es when shuffle_es > 1; then bytes
// [skip[i], skip[i] + take[i]) of the result (multiples of the element size) go to dst[i].
// Returns 0, or PK_E_UNSUPPORTED when no zlib could be found, PK_E_INVALID for bad arguments or a
// chunk that does not inflate to chunk_bytes.
extern "C" int pk_host_unfilter_chunks(int n_chunks, const void *const *src, const int64_t *src_len, int deflate,
                                       int shuffle_es, int64_t chunk_bytes, const int64_t *skip, const int64_t *take,
                                       void *const *dst, int threads)
{
    if (n_chunks < 0 || chunk_bytes <= 0 || (n_chunks > 0 && (!src || !src_len || !skip || !take || !dst))) {
        pk_set_error("pk_host_unfilter_chunks: bad arguments");
        return PK_E_INVALID;
    }
    const int es = shuffle_es > 1 ? shuffle_es : 1;
    if (chunk_bytes % es) {
        pk_set_error("pk_host_unfilter_chunks: chunk of %lld bytes, element size %d", (long long)chunk_bytes, es);
        return PK_E_INVALID;
    }
    for (int i = 0; i < n_chunks; i++)
        if (skip[i] < 0 || take[i] < 0 || skip[i] + take[i] > chunk_bytes || skip[i] % es || take[i] % es ||
            src_len[i] < 0 || (take[i] > 0 && (!src[i] || !dst[i])) || (!deflate && src_len[i] < chunk_bytes)) {
            pk_set_error("pk_host_unfilter_chunks: slice of chunk %d out of range", i);
            return PK_E_INVALID;
        }
    if (deflate) {
        std::call_once(g_z_once, find_zlib);
        if (!g_uncompress) {
            pk_set_error("pk_host_unfilter_chunks: no libz.so.1 to be found");
            return PK_E_UNSUPPORTED;
        }
    }
    std::atomic<int> next{0}, bad{-1};
    auto work = [&]() {
        std::vector<unsigned char> tmp(deflate ? (size_t)chunk_bytes : 0);
        for (int i = next.fetch_add(1); i < n_chunks; i = next.fetch_add(1)) {
            if (take[i] == 0) continue;
            const unsigned char *body = static_cast<const unsigned char *>(src[i]);
            if (deflate) {
                unsigned long got = (unsigned long)chunk_bytes;
                if (g_uncompress(tmp.data(), &got, body, (unsigned long)src_len[i]) != 0 || (int64_t)got != chunk_bytes) {
                    bad.store(i);
                    continue;
                }
                body = tmp.data();
            }
            if (es > 1)
                unshuffle_slice(body, chunk_bytes / es, es, skip[i], take[i], static_cast<unsigned char *>(dst[i]));
            else
                memcpy(dst[i], body + skip[i], (size_t)take[i]);
        }
    };
    int nt = threads < 1 ? 1 : (threads > n_chunks ? n_chunks : threads);
    if (nt <= 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    if (bad.load() >= 0) {
        pk_set_error("pk_host_unfilter_chunks: chunk %d does not inflate to %lld bytes", bad.load(), (long long)chunk_bytes);
        return PK_E_INVALID;
    }
    return PK_OK;
}
