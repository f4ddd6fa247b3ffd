// CPU test of peakachu_amd/csrc/pk_comm_protocol.h: the gather protocol of pk_comm.hip run by R
// THREADS over a fabric of mailboxes whose send / recv block like RCCL's (an unmatched one waits --
// here: until a timeout that the test reports as a deadlock), with local failures injected.
// The RCCL fabric differs only in what moves the bytes.  Built and run by tests/test_comm_protocol.py.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../peakachu_amd/csrc/pk_comm_protocol.h"

// how long a blocked operation waits: the protocol's own bound (PK_COMM_TIMEOUT, seconds), which
// the scenarios below set -- 4 s where nobody may ever be left waiting (a timeout there is counted
// as a deadlock), a fraction of a second where a peer is MADE to vanish
// (PK_TEST_WAIT_SCALE stretches the waits that count as deadlocks when they expire: a sanitizer
// build runs the same scenarios ten or more times slower)
static int wait_scale()
{
    const char *e = getenv("PK_TEST_WAIT_SCALE");
    const int v = e ? atoi(e) : 1;
    return v > 0 ? v : 1;
}
static int timeout_ms() { return (int)(pk_proto::comm_timeout_seconds() * 1000.0) * wait_scale(); }
#define TIMEOUT_MS timeout_ms()

// A timed condition-variable wait against the SYSTEM clock: that one is pthread_cond_timedwait, which
// ThreadSanitizer intercepts; wait_for / steady-clock waits are pthread_cond_clockwait, which the
// libtsan of GCC 11 does not know (it then reports double locks and races under a held mutex).
template <class Pred>
static bool timed_wait(std::condition_variable &cv, std::unique_lock<std::mutex> &lk, int ms, Pred pred)
{
    return cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(ms), pred);
}

struct World {
    int R;
    std::mutex mu;
    std::condition_variable cv;
    // all-gather: generation counter, contributions of the current generation
    long gen = 0;
    int arrived = 0;
    std::vector<std::vector<int64_t>> contrib;
    std::vector<int64_t> result;
    // mailboxes: (src, dst) -> queue of messages; consumed counts per (src, dst)
    std::map<std::pair<int, int>, std::deque<std::vector<char>>> box;
    std::map<std::pair<int, int>, long> posted, consumed;
    std::atomic<int> deadlocks{0};
    explicit World(int r) : R(r), contrib((size_t)r) {}
};

struct ThreadFabric {
    World *w;
    int me;
    // "device" memory of this rank
    std::vector<char> st[2];
    // injected failures
    bool fail_reserve[2] = {false, false}, fail_upload = false, fail_copy = false;
    bool vanish_before_send = false;  // this rank votes "ready" and is gone before its sends (a process that died)
    bool aborted = false;             // a wait ran out: like rccl_fabric after ncclCommAbort, every later call is refused
    int n_allgather = 0;
    std::string last_error;
    // grouped operations are started at group_end, like NCCL's
    struct Op { bool is_send; const void *sp; void *rp; size_t n; int peer; };
    std::vector<Op> pending;
    bool in_group = false;
    std::vector<std::pair<int, long>> my_sends;  // (dst, serial) still to be consumed

    int rank() const { return me; }
    int nranks() const { return w->R; }
    int refused() { error("the fabric was aborted after a timed-out wait"); return PK_E_COMM; }
    int allgather(const int64_t *mine, int words, int64_t *all)
    {
        if (aborted) return refused();
        n_allgather++;
        std::unique_lock<std::mutex> lk(w->mu);
        const long g = w->gen;
        w->contrib[(size_t)me].assign(mine, mine + words);
        if (++w->arrived == w->R) {
            w->result.clear();
            for (int r = 0; r < w->R; r++) {
                if ((int)w->contrib[(size_t)r].size() != words) { w->deadlocks++; }  // ranks out of step
                w->result.insert(w->result.end(), w->contrib[(size_t)r].begin(), w->contrib[(size_t)r].end());
            }
            w->arrived = 0;
            w->gen++;
            w->cv.notify_all();
        } else if (!timed_wait(w->cv, lk, TIMEOUT_MS, [&] { return w->gen != g; })) {
            w->deadlocks++;
            w->arrived--;   // (this rank leaves the round it was waiting in)
            aborted = true;
            error("all-gather timed out: a rank never arrived");
            return PK_E_COMM;
        }
        memcpy(all, w->result.data(), sizeof(int64_t) * (size_t)words * (size_t)w->R);
        return PK_OK;
    }
    size_t stage_cap(int i) const { return st[i].size(); }
    int reserve(int i, size_t bytes)
    {
        if (bytes <= st[i].size()) return PK_OK;
        if (fail_reserve[i]) { error("injected: staging allocation failed"); return PK_E_NOMEM; }
        st[i].assign(bytes + bytes / 2 + 64, 0);
        return PK_OK;
    }
    char *stage(int i) { return st[i].data(); }
    int copy_dd(void *d, const void *s, size_t n)
    {
        if (fail_copy) { error("injected: local copy refused"); return PK_E_HIP; }
        memcpy(d, s, n);
        return PK_OK;
    }
    int upload(void *d, const void *s, size_t n)
    {
        if (fail_upload) { error("injected: upload failed"); return PK_E_HIP; }
        memcpy(d, s, n);
        return PK_OK;
    }
    int download(void *d, const void *s, size_t n) { memcpy(d, s, n); return PK_OK; }
    int group_begin() { in_group = true; return PK_OK; }
    int run(const Op &op)
    {
        std::unique_lock<std::mutex> lk(w->mu);
        if (aborted) return refused();
        if (op.is_send) {
            if (vanish_before_send) { error("injected: this rank is gone"); return PK_E_HIP; }
            auto key = std::make_pair(me, op.peer);
            w->box[key].emplace_back((const char *)op.sp, (const char *)op.sp + op.n);
            my_sends.emplace_back(op.peer, ++w->posted[key]);
            w->cv.notify_all();
            return PK_OK;
        }
        auto key = std::make_pair(op.peer, me);
        if (!timed_wait(w->cv, lk, TIMEOUT_MS, [&] { return !w->box[key].empty(); })) {
            w->deadlocks++;
            aborted = true;
            error("recv timed out: rank %d never sent", op.peer);
            return PK_E_COMM;
        }
        std::vector<char> m = std::move(w->box[key].front());
        w->box[key].pop_front();
        w->consumed[key]++;
        w->cv.notify_all();
        if (m.size() != op.n) { w->deadlocks++; error("size mismatch %zu vs %zu", m.size(), op.n); return PK_E_COMM; }
        memcpy(op.rp, m.data(), op.n);
        return PK_OK;
    }
    int group_end()
    {
        in_group = false;
        int rc = PK_OK;
        for (const Op &op : pending)
            if (!rc) rc = run(op);
        pending.clear();
        return rc;
    }
    int send(const void *p, size_t n, int peer)
    {
        Op op{true, p, nullptr, n, peer};
        if (in_group) { pending.push_back(op); return PK_OK; }
        return run(op);
    }
    int recv(void *p, size_t n, int peer)
    {
        Op op{false, nullptr, p, n, peer};
        if (in_group) { pending.push_back(op); return PK_OK; }
        return run(op);
    }
    int sync()
    {   // a send is complete when its receiver has taken it: an unmatched one waits (RCCL: until PK_COMM_TIMEOUT)
        if (aborted) return refused();
        std::unique_lock<std::mutex> lk(w->mu);
        for (auto &s : my_sends) {
            auto key = std::make_pair(me, s.first);
            if (!timed_wait(w->cv, lk, TIMEOUT_MS, [&] { return w->consumed[key] >= s.second; })) {
                w->deadlocks++;
                aborted = true;
                error("send to rank %d was never received", s.first);
                my_sends.clear();
                return PK_E_COMM;
            }
        }
        my_sends.clear();
        return PK_OK;
    }
    void error(const char *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        last_error = buf;
    }
};

static int failures = 0;
#define CHECK(cond, ...)                                            \
    do {                                                            \
        if (!(cond)) {                                              \
            failures++;                                             \
            printf("FAILED %s:%d %s -- ", __FILE__, __LINE__, #cond); \
            printf(__VA_ARGS__);                                    \
            printf("\n");                                           \
        }                                                           \
    } while (0)

struct Data {  // one rank's scored pixels
    std::vector<int32_t> x, y;
    std::vector<double> p, s;
};
static Data make(int rank, int64_t n)
{
    Data d;
    for (int64_t i = 0; i < n; i++) {
        d.x.push_back(rank * 1000000 + (int)i);
        d.y.push_back(rank * 1000000 + (int)i + 7);
        d.p.push_back(0.5 + rank + i * 1e-6);
        d.s.push_back(3.0 * rank + i);
    }
    return d;
}

// one gather of the scored pixels by R threads; returns the ranks' return codes
static std::vector<int> run_scored(World &w, std::vector<ThreadFabric> &fb, const std::vector<int64_t> &n, int64_t cap,
                                   Data *out, std::vector<int64_t> *counts)
{
    const int R = w.R;
    std::vector<int> rcs((size_t)R, -999);
    std::vector<Data> in;
    for (int r = 0; r < R; r++) in.push_back(make(r, n[(size_t)r]));
    out->x.assign((size_t)cap + 1, -1); out->y.assign((size_t)cap + 1, -1);
    out->p.assign((size_t)cap + 1, -1); out->s.assign((size_t)cap + 1, -1);
    std::vector<std::vector<int64_t>> cnt((size_t)R, std::vector<int64_t>((size_t)R, -1));
    std::vector<std::thread> th;
    for (int r = 0; r < R; r++)
        th.emplace_back([&, r] {
            rcs[(size_t)r] = pk_proto::gather_scored(fb[(size_t)r], n[(size_t)r], in[(size_t)r].x.data(), in[(size_t)r].y.data(),
                                                     in[(size_t)r].p.data(), in[(size_t)r].s.data(), cnt[(size_t)r].data(),
                                                     r == 0 ? cap : 0, r == 0 ? out->x.data() : nullptr,
                                                     r == 0 ? out->y.data() : nullptr, r == 0 ? out->p.data() : nullptr,
                                                     r == 0 ? out->s.data() : nullptr);
        });
    for (auto &t : th) t.join();
    *counts = cnt[0];
    return rcs;
}

static void expect_merged(const Data &out, const std::vector<int64_t> &n)
{
    size_t o = 0;
    for (size_t r = 0; r < n.size(); r++) {
        Data d = make((int)r, n[r]);
        for (int64_t i = 0; i < n[r]; i++, o++)
            CHECK(out.x[o] == d.x[(size_t)i] && out.y[o] == d.y[(size_t)i] && out.p[o] == d.p[(size_t)i] && out.s[o] == d.s[(size_t)i],
                  "rank %zu record %lld", r, (long long)i);
    }
}

static void scenario_scored(int R)
{
    World w(R);
    std::vector<ThreadFabric> fb((size_t)R);
    for (int r = 0; r < R; r++) { fb[(size_t)r].w = &w; fb[(size_t)r].me = r; }
    Data out;
    std::vector<int64_t> counts;
    std::vector<int64_t> n((size_t)R);
    for (int r = 0; r < R; r++) n[(size_t)r] = r == 1 ? 0 : 1000 + 37 * r;   // (a rank with nothing)
    int64_t total = 0;
    for (auto v : n) total += v;
    auto all_ok = [&](const std::vector<int> &rcs) { for (int c : rcs) if (c != PK_OK) return false; return true; };

    // A. first call: the root's staging area has to grow -> two rounds; second call: one
    auto rcs = run_scored(w, fb, n, total + 5, &out, &counts);
    CHECK(all_ok(rcs), "first gather R=%d", R);
    expect_merged(out, n);
    for (int r = 0; r < R; r++) CHECK(counts[(size_t)r] == n[(size_t)r], "counts");
    for (int r = 0; r < R; r++) CHECK(fb[(size_t)r].n_allgather == 2, "first call: %d all-gathers", fb[(size_t)r].n_allgather);
    rcs = run_scored(w, fb, n, total, &out, &counts);
    CHECK(all_ok(rcs), "second gather");
    expect_merged(out, n);
    for (int r = 0; r < R; r++) CHECK(fb[(size_t)r].n_allgather == 3, "steady state is ONE all-gather (saw %d in all)", fb[(size_t)r].n_allgather);

    // B. the root's buffers are one record short: everybody is told, nobody sends, the next call works
    rcs = run_scored(w, fb, n, total - 1, &out, &counts);
    CHECK(rcs[0] == PK_E_INVALID, "root code %d", rcs[0]);
    for (int r = 1; r < R; r++) CHECK(rcs[(size_t)r] == PK_E_COMM, "peer code %d", rcs[(size_t)r]);
    rcs = run_scored(w, fb, n, total, &out, &counts);
    CHECK(all_ok(rcs), "gather after a refusal");
    expect_merged(out, n);

    // C. a larger result than ever: the staging area must grow and the allocation fails -> second
    //    vote, everybody leaves; afterwards (allocation allowed again) the same call succeeds
    std::vector<int64_t> big = n;
    big[(size_t)(R - 1)] += 50000;
    fb[0].fail_reserve[0] = true;
    rcs = run_scored(w, fb, big, total + 50000, &out, &counts);
    CHECK(rcs[0] == PK_E_NOMEM, "root code %d", rcs[0]);
    for (int r = 1; r < R; r++) CHECK(rcs[(size_t)r] == PK_E_COMM, "peer code %d", rcs[(size_t)r]);
    fb[0].fail_reserve[0] = false;
    rcs = run_scored(w, fb, big, total + 50000, &out, &counts);
    CHECK(all_ok(rcs), "gather after a failed growth");
    expect_merged(out, big);

    // D. no growth, and the root's own copy is refused: no vote is scheduled, so the root must
    //    still receive what the peers send (they finish) and fail afterwards
    fb[0].fail_copy = true;
    rcs = run_scored(w, fb, n, total, &out, &counts);
    CHECK(rcs[0] == PK_E_HIP, "root code %d", rcs[0]);
    for (int r = 1; r < R; r++) CHECK(rcs[(size_t)r] == PK_OK, "peer code %d", rcs[(size_t)r]);
    fb[0].fail_copy = false;
    rcs = run_scored(w, fb, n, total, &out, &counts);
    CHECK(all_ok(rcs), "gather after a refused copy");
    expect_merged(out, n);
    CHECK(w.deadlocks.load() == 0, "%d operations timed out (a rank was left waiting)", w.deadlocks.load());
}

static void scenario_bytes(int R)
{
    World w(R);
    std::vector<ThreadFabric> fb((size_t)R);
    for (int r = 0; r < R; r++) { fb[(size_t)r].w = &w; fb[(size_t)r].me = r; }
    auto run = [&](const std::vector<int64_t> &n, int64_t cap, std::vector<char> *out) {
        std::vector<int> rcs((size_t)R, -999);
        std::vector<std::vector<char>> in((size_t)R);
        for (int r = 0; r < R; r++)
            for (int64_t i = 0; i < n[(size_t)r]; i++) in[(size_t)r].push_back((char)(r * 31 + i));
        out->assign((size_t)cap + 1, 0x55);
        std::vector<std::thread> th;
        for (int r = 0; r < R; r++)
            th.emplace_back([&, r] {
                std::vector<int64_t> c((size_t)R);
                rcs[(size_t)r] = pk_proto::gatherv_bytes(fb[(size_t)r], in[(size_t)r].data(), n[(size_t)r], c.data(),
                                                         r == 0 ? out->data() : nullptr, r == 0 ? cap : 0);
            });
        for (auto &t : th) t.join();
        return rcs;
    };
    auto merged_ok = [&](const std::vector<char> &out, const std::vector<int64_t> &n) {
        size_t o = 0;
        for (int r = 0; r < R; r++)
            for (int64_t i = 0; i < n[(size_t)r]; i++, o++)
                if (out[o] != (char)(r * 31 + i)) return false;
        return true;
    };
    std::vector<int64_t> n((size_t)R);
    int64_t total = 0;
    for (int r = 0; r < R; r++) { n[(size_t)r] = r == 0 ? 0 : 5000 + 11 * r; total += n[(size_t)r]; }
    std::vector<char> out;
    auto rcs = run(n, total, &out);
    for (int c : rcs) CHECK(c == PK_OK, "bytes gather code %d", c);
    CHECK(merged_ok(out, n), "bytes merged");
    rcs = run(n, total - 1, &out);   // root capacity
    CHECK(rcs[0] == PK_E_INVALID, "root code %d", rcs[0]);
    for (int r = 1; r < R; r++) CHECK(rcs[(size_t)r] == PK_E_COMM, "peer code %d", rcs[(size_t)r]);
    if (R > 1) {   // a peer cannot stage its bytes: its status word travels with the counts
        fb[(size_t)(R - 1)].fail_upload = true;
        rcs = run(n, total, &out);
        CHECK(rcs[(size_t)(R - 1)] == PK_E_HIP, "failing peer code %d", rcs[(size_t)(R - 1)]);
        for (int r = 0; r < R - 1; r++) CHECK(rcs[(size_t)r] == PK_E_COMM, "other rank %d code %d", r, rcs[(size_t)r]);
        fb[(size_t)(R - 1)].fail_upload = false;
    }
    rcs = run(n, total, &out);
    for (int c : rcs) CHECK(c == PK_OK, "bytes gather after failures: code %d", c);
    CHECK(merged_ok(out, n), "bytes merged after failures");
    CHECK(w.deadlocks.load() == 0, "%d operations timed out (a rank was left waiting)", w.deadlocks.load());
}

// A peer votes "ready" and is gone before its sends (what a process that dies between the count
// all-gather and its ncclSend looks like to the others): nobody may wait longer than the bound, the
// ranks that were served finish, the others leave with PK_E_COMM and their side of the fabric
// refuses every later call.
static void scenario_vanishing_peer(int R)
{
    setenv("PK_COMM_TIMEOUT", "0.3", 1);
    World w(R);
    std::vector<ThreadFabric> fb((size_t)R);
    for (int r = 0; r < R; r++) { fb[(size_t)r].w = &w; fb[(size_t)r].me = r; }
    std::vector<int64_t> n((size_t)R, 500), counts;
    int64_t total = 500 * (int64_t)R;
    Data out;
    auto rcs = run_scored(w, fb, n, total, &out, &counts);   // healthy first call (staging areas exist afterwards)
    for (int c : rcs) CHECK(c == PK_OK, "healthy gather code %d", c);
    CHECK(w.deadlocks.load() == 0, "healthy gather timed out");
    const int gone = R - 1;
    fb[(size_t)gone].vanish_before_send = true;
    const auto t0 = std::chrono::steady_clock::now();
    rcs = run_scored(w, fb, n, total, &out, &counts);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    CHECK(rcs[(size_t)gone] == PK_E_HIP, "vanished peer code %d", rcs[(size_t)gone]);
    CHECK(rcs[0] == PK_E_COMM, "root code %d (must give up, not wait)", rcs[0]);
    for (int r = 1; r < gone; r++) CHECK(rcs[(size_t)r] == PK_OK, "served peer %d code %d", r, rcs[(size_t)r]);
    CHECK(dt < 1.5 * wait_scale(), "the gather took %.2f s with a 0.3 s bound", dt);
    CHECK(fb[0].aborted, "the root's fabric must refuse further calls");
    CHECK(fb[0].last_error.find("never sent") != std::string::npos, "root message: %s", fb[0].last_error.c_str());
    // the next call: the root refuses at once, the others run into their own bound -- everybody returns
    fb[(size_t)gone].vanish_before_send = false;
    const auto t1 = std::chrono::steady_clock::now();
    rcs = run_scored(w, fb, n, total, &out, &counts);
    const double dt2 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    for (int r = 0; r < R; r++) CHECK(rcs[(size_t)r] == PK_E_COMM, "call after an abort: rank %d code %d", r, rcs[(size_t)r]);
    CHECK(dt2 < 1.5 * wait_scale(), "the call after an abort took %.2f s", dt2);
    setenv("PK_COMM_TIMEOUT", "4", 1);
}

static void scenario_wait_until()
{
    int calls = 0;
    CHECK(pk_proto::wait_until([&] { return ++calls >= 3 ? 1 : 0; }, 1.0) == PK_OK && calls == 3, "done after three polls");
    CHECK(pk_proto::wait_until([] { return (int)PK_E_HIP; }, 1.0) == PK_E_HIP, "an error is handed on");
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(pk_proto::wait_until([] { return 0; }, 0.05) == pk_proto::WAIT_TIMED_OUT, "the bound");
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    CHECK(dt >= 0.05 && dt < 0.5, "waited %.3f s for a 0.05 s bound", dt);
    setenv("PK_COMM_TIMEOUT", "7.5", 1);
    CHECK(pk_proto::comm_timeout_seconds() == 7.5, "PK_COMM_TIMEOUT is read");
    setenv("PK_COMM_TIMEOUT", "nonsense", 1);
    CHECK(pk_proto::comm_timeout_seconds() == 120.0, "default bound");
}

int main()
{
    scenario_wait_until();
    setenv("PK_COMM_TIMEOUT", "4", 1);
    for (int R : {2, 3, 8}) scenario_vanishing_peer(R);
    printf("vanishing peer done, failures so far %d\n", failures);
    for (int R : {1, 2, 3, 8}) {
        scenario_scored(R);
        scenario_bytes(R);
        printf("R=%d done, failures so far %d\n", R, failures);
    }
    printf(failures ? "FAILED\n" : "OK\n");
    return failures ? 1 : 0;
}
