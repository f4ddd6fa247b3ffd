// CPU test of peakachu_amd/csrc/pk_hostio.hip (pk_host_unfilter_chunks: the chunk pipeline of the
// .cool reader -- inflate, byte un-shuffle, placement -- that replaces libhdf5's filter pipeline behind
// peakachu/score_genome.py:55-57).  The translation unit has no device code, so it is compiled here as
// plain C++ and driven under Address / UndefinedBehaviour / Thread sanitizers: truncated and corrupt
// deflate streams, slices at the chunk's edges, unaligned destinations, 1-8 threads.  Every destination
// sits between guard bytes that must survive.  Built and run by tests/test_hostio_sanitized.py.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <random>
#include <string>
#include <vector>

#include "../../peakachu_amd/csrc/pk_hostio.hip"

static std::string g_last_error;
void pk_set_error(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

static int failures = 0;
#define CHECK(cond, ...)                                                          \
    do {                                                                          \
        if (!(cond)) {                                                            \
            failures++;                                                           \
            if (failures < 20) {                                                  \
                printf("FAILED %s:%d %s -- ", __FILE__, __LINE__, #cond);         \
                printf(__VA_ARGS__);                                              \
                printf("\n");                                                     \
            }                                                                     \
        }                                                                         \
    } while (0)

typedef std::vector<unsigned char> bytes;

static bytes shuffle(const bytes &plain, int es)
{
    if (es <= 1) return plain;
    const size_t n_el = plain.size() / (size_t)es;
    bytes out(plain.size());
    for (size_t e = 0; e < n_el; e++)
        for (int k = 0; k < es; k++) out[(size_t)k * n_el + e] = plain[e * (size_t)es + (size_t)k];
    return out;
}

static bytes deflate_bytes(const bytes &in)
{
    uLongf cap = compressBound((uLong)in.size());
    bytes out(cap);
    if (compress2(out.data(), &cap, in.data(), (uLong)in.size(), 6) != Z_OK) abort();
    out.resize(cap);
    return out;
}

struct Case {
    int n, es, deflate, threads;
    int64_t chunk_bytes;
    std::vector<bytes> plain, stored;
    std::vector<int64_t> skip, take, src_len;
};

static const unsigned char GUARD = 0xA5;

// runs one call with every destination at byte offset `misalign` inside a guarded buffer; returns rc
static int run(const Case &c, int misalign, std::vector<bytes> *got, const std::vector<bytes> *override_src = nullptr)
{
    const std::vector<bytes> &st = override_src ? *override_src : c.stored;
    std::vector<bytes> buf((size_t)c.n);
    std::vector<const void *> src((size_t)c.n);
    std::vector<void *> dst((size_t)c.n);
    std::vector<int64_t> len((size_t)c.n);
    for (int i = 0; i < c.n; i++) {
        buf[(size_t)i].assign((size_t)c.take[(size_t)i] + 64, GUARD);
        // (an exactly sized heap copy of the stored bytes: a read past the end is the sanitizer's to see)
        src[(size_t)i] = st[(size_t)i].data();
        len[(size_t)i] = (int64_t)st[(size_t)i].size();
        dst[(size_t)i] = buf[(size_t)i].data() + 16 + misalign;
    }
    const int rc = pk_host_unfilter_chunks(c.n, src.data(), len.data(), c.deflate, c.es, c.chunk_bytes, c.skip.data(),
                                           c.take.data(), dst.data(), c.threads);
    for (int i = 0; i < c.n; i++) {
        const bytes &b = buf[(size_t)i];
        const size_t lo = 16 + (size_t)misalign, hi = lo + (size_t)c.take[(size_t)i];
        for (size_t k = 0; k < lo; k++) CHECK(b[k] == GUARD, "chunk %d: byte %zu in front of the slice was written", i, k);
        for (size_t k = hi; k < b.size(); k++) CHECK(b[k] == GUARD, "chunk %d: byte %zu behind the slice was written", i, k - hi);
    }
    if (got) {
        got->clear();
        for (int i = 0; i < c.n; i++)
            got->emplace_back(buf[(size_t)i].begin() + 16 + misalign, buf[(size_t)i].begin() + 16 + misalign + c.take[(size_t)i]);
    }
    return rc;
}

static Case make_case(std::mt19937_64 &rng, int n, int es, int deflate, int threads, int64_t chunk_bytes)
{
    Case c;
    c.n = n; c.es = es; c.deflate = deflate; c.threads = threads; c.chunk_bytes = chunk_bytes;
    const int e = es > 1 ? es : 1;
    const int64_t n_el = chunk_bytes / e;
    for (int i = 0; i < n; i++) {
        bytes p((size_t)chunk_bytes);
        // pixel-table-like content: slowly growing integers (compressible) with noise in the low byte
        for (int64_t k = 0; k < chunk_bytes; k++) p[(size_t)k] = (unsigned char)((k % e == 0) ? rng() : (k / e / 37));
        c.plain.push_back(p);
        bytes s = shuffle(p, es);
        c.stored.push_back(deflate ? deflate_bytes(s) : s);
        // slices: whole chunk, empty, the first element, the last element, skip + take ending AT the edge, random
        int64_t sk, tk;
        switch (i % 6) {
        case 0: sk = 0; tk = chunk_bytes; break;
        case 1: sk = (int64_t)(rng() % (uint64_t)(n_el + 1)) * e; tk = 0; break;
        case 2: sk = 0; tk = e; break;
        case 3: sk = chunk_bytes - e; tk = e; break;
        case 4: sk = (int64_t)(rng() % (uint64_t)n_el) * e; tk = chunk_bytes - sk; break;
        default: sk = (int64_t)(rng() % (uint64_t)n_el) * e; tk = (int64_t)(rng() % (uint64_t)((chunk_bytes - sk) / e + 1)) * e; break;
        }
        c.skip.push_back(sk);
        c.take.push_back(tk);
    }
    return c;
}

int main()
{
    std::mt19937_64 rng(20260106);
    // ---- what must work: every element size, stored / deflated, 1-8 threads, any destination alignment
    for (int es : {1, 2, 4, 8, 3}) {
        for (int deflate : {0, 1}) {
            for (int threads : {1, 2, 3, 8}) {
                const int64_t chunk_bytes = (int64_t)es * (es == 3 ? 1001 : 4096 + 8 * threads);
                Case c = make_case(rng, 13, es, deflate, threads, chunk_bytes);
                for (int mis : {0, 1, 3, 7}) {
                    std::vector<bytes> got;
                    const int rc = run(c, mis, &got);
                    CHECK(rc == PK_OK, "es %d deflate %d threads %d misalign %d: rc %d (%s)", es, deflate, threads, mis, rc, g_last_error.c_str());
                    for (int i = 0; i < c.n && rc == PK_OK; i++) {
                        const bytes want(c.plain[(size_t)i].begin() + c.skip[(size_t)i], c.plain[(size_t)i].begin() + c.skip[(size_t)i] + c.take[(size_t)i]);
                        CHECK(got[(size_t)i] == want, "es %d deflate %d threads %d misalign %d: chunk %d differs", es, deflate, threads, mis, i);
                    }
                }
            }
        }
    }
    printf("good cases done, failures so far %d\n", failures);
    // ---- hostile stored bytes: truncated streams, flipped bits, garbage, streams that inflate to another size
    for (int es : {4, 8}) {
        for (int threads : {1, 4, 8}) {
            Case c = make_case(rng, 9, es, 1, threads, (int64_t)es * 2048);
            for (int trial = 0; trial < 40; trial++) {
                std::vector<bytes> st = c.stored;
                const size_t victim = (size_t)(rng() % (uint64_t)c.n);
                bytes &v = st[victim];
                const int kind = trial % 5;
                if (kind == 0) v.resize((size_t)(rng() % (uint64_t)v.size()));                       // truncated (possibly to nothing)
                else if (kind == 1) v[(size_t)(rng() % (uint64_t)v.size())] ^= (unsigned char)(1u << (rng() % 8));  // one bit
                else if (kind == 2) for (auto &b : v) b = (unsigned char)rng();                    // garbage
                else if (kind == 3) { bytes p(c.plain[victim]); p.resize(p.size() / 2); v = deflate_bytes(p); }  // a SHORTER chunk
                else { bytes p(c.plain[victim]); p.resize(p.size() * 2, 7); v = deflate_bytes(p); }          // a LONGER chunk
                if (c.take[victim] == 0) c.take[victim] = es, c.skip[victim] = 0;
                std::vector<bytes> got;
                const int rc = run(c, (int)(trial % 5), &got, &st);
                // a flipped bit may land in a place zlib does not check before the end -- the checksum then
                // catches it; either way the call says PK_E_INVALID or (bit flip that changed nothing it reads) succeeds
                if (kind != 1) CHECK(rc == PK_E_INVALID, "hostile kind %d: rc %d", kind, rc);
                else CHECK(rc == PK_E_INVALID || rc == PK_OK, "bit flip: rc %d", rc);
                if (rc == PK_E_INVALID) CHECK(!g_last_error.empty(), "no message");
            }
        }
    }
    printf("hostile streams done, failures so far %d\n", failures);
    // ---- bad arguments are refused before anything is touched
    {
        Case c = make_case(rng, 3, 4, 0, 2, 4096);
        Case d = c; d.skip[1] = 2;                                  // not a multiple of the element size
        CHECK(run(d, 0, nullptr) == PK_E_INVALID, "unaligned skip accepted");
        d = c; d.skip[2] = 4096 - 8; d.take[2] = 16;               // runs over the chunk's end
        CHECK(run(d, 0, nullptr) == PK_E_INVALID, "slice over the edge accepted");
        d = c; d.take[0] = -4;
        { std::vector<bytes> st = d.stored; d.take[0] = 0; d.skip[0] = -4; CHECK(run(d, 0, nullptr, &st) == PK_E_INVALID, "negative skip accepted"); }
        d = c; d.stored[1].resize(100);                             // stored chunk shorter than chunk_bytes, no deflate
        CHECK(run(d, 0, nullptr) == PK_E_INVALID, "short stored chunk accepted");
        d = c; d.chunk_bytes = 4098;                                // not a multiple of the element size
        d.skip.assign(3, 0); d.take.assign(3, 0);
        CHECK(run(d, 0, nullptr) == PK_E_INVALID, "odd chunk size accepted");
        const void *nosrc[1] = {nullptr};
        void *nodst[1] = {nullptr};
        int64_t one_len[1] = {4096}, zero[1] = {0}, four[1] = {4};
        CHECK(pk_host_unfilter_chunks(1, nosrc, one_len, 0, 4, 4096, zero, four, nodst, 1) == PK_E_INVALID, "null pointers accepted");
        CHECK(pk_host_unfilter_chunks(0, nullptr, nullptr, 1, 4, 4096, nullptr, nullptr, nullptr, 4) == PK_OK, "no chunks");
        CHECK(pk_host_unfilter_chunks(-1, nullptr, nullptr, 1, 4, 4096, nullptr, nullptr, nullptr, 4) == PK_E_INVALID, "negative count");
    }
    printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
    return failures ? 1 : 0;
}
