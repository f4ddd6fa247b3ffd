// pk_comm_protocol.h -- the control flow of the one multi-GPU exchange (gather-v of the scored
// pixels to rank 0), written against a `Fabric` so that the SAME code runs over RCCL
// (pk_comm.hip: ncclAllGather / grouped ncclSend / ncclRecv over xGMI) and, in
// tests/native/test_comm_protocol.cpp, over threads with failures injected -- no multi-GPU box was
// available to the builder, and a slip in this protocol would leave every rank waiting inside RCCL.
//
// The reference has no counterpart: it appends each chromosome's pixels to one file from one
// process (peakachu/score_genome.py:46-84, peakachu/scoreUtils.py:127-135).
//
// Rules the protocol keeps:
//   * a rank that cannot post its half of a send / recv pair says so BEFORE anybody sends: every
//     failure a rank can know of in advance (its bytes could not be staged, the root's buffers are
//     too small, the root's own records could not be copied) travels as a status word in the same
//     all-gather as the counts -- one round trip in the steady state (round 3 needed two);
//   * the only thing that needs a second round is a staging area that has to GROW on the root once
//     the total is known (first call, or a larger result than ever before): then, and only then,
//     all ranks vote again, and the root copies its own part before that vote;
//   * behind the last vote nothing can fail locally any more: the peers send, the root receives;
//   * no wait is unbounded: what a fabric waits for (a stream that holds a collective, a send, a
//     receive) is polled through wait_until() against PK_COMM_TIMEOUT seconds (default 120); a rank
//     whose peer went away between its "ready" and its send gives up with PK_E_COMM, tears its
//     side of the fabric down (RCCL: ncclCommAbort) and refuses further calls -- the process is
//     expected to report and exit non-zero, a retry is a fresh process.
//
// Fabric (all sizes in bytes, all calls return PK_OK or an error code):
//   int rank(), nranks();
//   int allgather(const int64_t *mine, int words, int64_t *all);   // [nranks][words] on the host
//   size_t stage_cap(int i);  int reserve(int i, size_t bytes);  char *stage(int i);  // device
//   int copy_dd(void *dst, const void *src, size_t n);   // device -> device, this rank
//   int upload(void *dst, const void *src, size_t n);    // host -> device
//   int download(void *dst, const void *src, size_t n);  // device -> host
//   int group_begin(), group_end();  int send(const void *p, size_t n, int peer);
//   int recv(void *p, size_t n, int peer);  int sync();
//   void error(const char *fmt, ...);
#ifndef PK_COMM_PROTOCOL_H
#define PK_COMM_PROTOCOL_H
#include <stddef.h>
#include <stdint.h>

#include <stdlib.h>

#include <chrono>
#include <thread>
#include <vector>

#include "../../include/peakachu_hip.h"

namespace pk_proto {

constexpr int WAIT_TIMED_OUT = -100;  // wait_until's own code: never leaves a fabric

// seconds a fabric waits for its peers before it gives up: PK_COMM_TIMEOUT, default 120
inline double comm_timeout_seconds()
{
    const char *e = getenv("PK_COMM_TIMEOUT");
    if (e && *e) {
        char *end = nullptr;
        const double v = strtod(e, &end);
        if (end != e && v > 0) return v;
    }
    return 120.0;
}

// Polls `poll()` -- 1: done, 0: not yet, < 0: an error code to hand on -- until it is done, fails or
// `seconds` have passed (WAIT_TIMED_OUT).  Spins for the first 200 us (a healthy gather of a few MB
// is over by then), then yields, then sleeps in 50-us steps: the step of a multi-GPU run waits
// here once, a hung one costs no core.
template <class Poll>
int wait_until(Poll poll, double seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const int r = poll();
        if (r != 0) return r < 0 ? r : PK_OK;
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt >= seconds) return WAIT_TIMED_OUT;
        if (dt > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(50));
        else if (dt > 2e-4) std::this_thread::yield();
    }
}

// the first rank with a non-zero word at `at` of its `words`-wide row, -1 if none
inline int first_bad(const std::vector<int64_t> &all, int R, int words, int at, int64_t *code)
{
    for (int r = 0; r < R; r++)
        if (all[(size_t)r * words + at] != 0) {
            *code = all[(size_t)r * words + at];
            return r;
        }
    return -1;
}

// A second vote: every rank passes its status, every rank learns the first failure.
template <class F>
int vote(F &fb, int local_rc, int *bad, int64_t *code)
{
    const int R = fb.nranks();
    std::vector<int64_t> all((size_t)R, 0);
    const int64_t mine = local_rc;
    const int rc = fb.allgather(&mine, 1, all.data());
    if (rc) return rc;
    *bad = first_bad(all, R, 1, 0, code);
    return PK_OK;
}

// Scored pixels (x, y, prob, signal) of every rank -> the root's host buffers, rank after rank.
// dx .. ds: this rank's compacted result on the device, `mine` records.  counts (may be null)
// receives every rank's count on every rank.
template <class F>
int gather_scored(F &fb, int64_t mine, const int32_t *dx, const int32_t *dy, const double *dp, const double *ds,
                  int64_t *counts, int64_t cap, int32_t *ox, int32_t *oy, double *op, double *os)
{
    const int R = fb.nranks(), me = fb.rank();
    const bool root = me == 0;
    // ---- round 1: {count, status, root's capacity (records), root's staging capacity (bytes)}
    int64_t w[4] = {mine, 0, 0, 0};
    if (root) {
        w[2] = (ox && oy && op && os) ? cap : 0;
        w[3] = (int64_t)fb.stage_cap(0);
    }
    std::vector<int64_t> all((size_t)R * 4, 0);
    int rc = fb.allgather(w, 4, all.data());
    if (rc) return rc;
    int64_t total = 0;
    for (int r = 0; r < R; r++) total += all[(size_t)r * 4];
    if (counts)
        for (int r = 0; r < R; r++) counts[r] = all[(size_t)r * 4];
    int64_t code = 0;
    int bad = first_bad(all, R, 4, 1, &code);
    if (bad >= 0) {
        if (bad != me) fb.error("gather of the scored pixels: rank %d cannot take part (code %lld); nothing was sent", bad, (long long)code);
        return bad == me ? (int)code : PK_E_COMM;
    }
    const int64_t root_cap = all[2], root_stage = all[3];
    if (total > root_cap) {  // every rank sees the same numbers and takes the same way out
        fb.error("gather of the scored pixels: %lld pixels exceed the root's capacity %lld; nothing was sent",
                 (long long)total, (long long)root_cap);
        return root ? PK_E_INVALID : PK_E_COMM;
    }
    // one staging area on the root: [x | y | p | signal], each part t1 records (8-byte aligned)
    const size_t t1 = ((size_t)(total > 0 ? total : 1) + 1) & ~(size_t)1;
    const size_t need = t1 * 24;
    int32_t *gx = nullptr, *gy = nullptr;
    double *gp = nullptr, *gs = nullptr;
    auto carve = [&]() {
        gx = reinterpret_cast<int32_t *>(fb.stage(0));
        gy = gx + t1;
        gp = reinterpret_cast<double *>(gy + t1);
        gs = gp + t1;
    };
    auto copy_own = [&]() -> int {
        const size_t k0 = (size_t)mine;
        int c = PK_OK;
        if (k0 > 0) {
            c = fb.copy_dd(gx, dx, k0 * 4);
            if (!c) c = fb.copy_dd(gy, dy, k0 * 4);
            if (!c) c = fb.copy_dd(gp, dp, k0 * 8);
            if (!c) c = fb.copy_dd(gs, ds, k0 * 8);
        }
        return c;
    };
    // ---- the root's own part.  A staging area that must GROW (first call, or a larger result than
    // ever before) is the one thing that can still fail for a reason the others cannot see: then --
    // the same decision on every rank, they all know the total and the root's capacity -- everybody
    // votes again.  Without growth the root's copies go into memory that exists; should one be
    // refused all the same, no vote is scheduled: the root still RECEIVES what the peers are about to
    // send (or they would wait for ever) and reports its failure afterwards.
    int local = PK_OK;
    const bool grow = (int64_t)need > root_stage;
    if (root) {
        if (grow) local = fb.reserve(0, need);
        if (!local) {
            carve();
            local = copy_own();
        }
    }
    if (grow) {
        rc = vote(fb, local, &bad, &code);
        if (rc) return rc;
        if (bad >= 0) {
            if (bad != me) fb.error("gather of the scored pixels: rank %d cannot take part (code %lld); nothing was sent", bad, (long long)code);
            return bad == me ? local : PK_E_COMM;
        }
    }
    if (!root) {
        const size_t k = (size_t)mine;
        if (k > 0) {
            rc = fb.group_begin();
            if (!rc) rc = fb.send(dx, k * 4, 0);
            if (!rc) rc = fb.send(dy, k * 4, 0);
            if (!rc) rc = fb.send(dp, k * 8, 0);
            if (!rc) rc = fb.send(ds, k * 8, 0);
            const int re = fb.group_end();
            if (!rc) rc = re;
            if (rc) return rc;
        }
        return fb.sync();
    }
    rc = fb.group_begin();
    size_t off = (size_t)mine;
    for (int r = 1; r < R && !rc; r++) {
        const size_t k = (size_t)all[(size_t)r * 4];
        if (k == 0) continue;
        rc = fb.recv(gx + off, k * 4, r);
        if (!rc) rc = fb.recv(gy + off, k * 4, r);
        if (!rc) rc = fb.recv(gp + off, k * 8, r);
        if (!rc) rc = fb.recv(gs + off, k * 8, r);
        off += k;
    }
    {
        const int re = fb.group_end();
        if (!rc) rc = re;
    }
    if (rc) return rc;
    if (local) {  // (the root's own copy had failed: the peers are served, the call is not)
        fb.sync();
        return local;
    }
    // (what the peers send has arrived -- or the bounded wait has run out -- BEFORE anything is
    // copied to the host: a copy to pageable memory may wait for the stream where no deadline reaches)
    rc = fb.sync();
    if (rc) return rc;
    if (total > 0) {
        const size_t t = (size_t)total;
        rc = fb.download(ox, gx, t * 4);
        if (!rc) rc = fb.download(oy, gy, t * 4);
        if (!rc) rc = fb.download(op, gp, t * 8);
        if (!rc) rc = fb.download(os, gs, t * 8);
        if (rc) return rc;
    }
    return fb.sync();
}

// Arbitrary bytes of every rank -> the root's host buffer, rank after rank (score_genome's packed
// records).  `send` is a HOST pointer.
template <class F>
int gatherv_bytes(F &fb, const void *send, int64_t nbytes, int64_t *counts, void *recv, int64_t cap)
{
    const int R = fb.nranks(), me = fb.rank();
    const bool root = me == 0;
    // this rank's bytes go up first; a rank that cannot stage them takes part with a count of zero
    // and says so in its status word
    int local = fb.reserve(0, (size_t)(nbytes > 0 ? nbytes : 1));
    if (!local && nbytes > 0) local = fb.upload(fb.stage(0), send, (size_t)nbytes);
    int64_t w[4] = {local ? 0 : nbytes, local, 0, 0};
    if (root) {
        w[2] = recv ? cap : 0;
        w[3] = (int64_t)fb.stage_cap(1);
    }
    std::vector<int64_t> all((size_t)R * 4, 0);
    int rc = fb.allgather(w, 4, all.data());
    if (rc) return rc;
    int64_t total = 0;
    for (int r = 0; r < R; r++) total += all[(size_t)r * 4];
    if (counts)
        for (int r = 0; r < R; r++) counts[r] = all[(size_t)r * 4];
    int64_t code = 0;
    int bad = first_bad(all, R, 4, 1, &code);
    if (bad >= 0) {
        if (bad != me) fb.error("gather: rank %d cannot take part (code %lld); nothing was sent", bad, (long long)code);
        return bad == me ? local : PK_E_COMM;
    }
    if (total > all[2]) {
        fb.error("gather: %lld bytes exceed the root's capacity %lld; nothing was sent", (long long)total, (long long)all[2]);
        return root ? PK_E_INVALID : PK_E_COMM;
    }
    const size_t need = (size_t)(total > 0 ? total : 1);
    const bool grow = (int64_t)need > all[3];
    local = PK_OK;
    if (root) {
        if (grow) local = fb.reserve(1, need);
        if (!local && nbytes > 0) local = fb.copy_dd(fb.stage(1), fb.stage(0), (size_t)nbytes);
    }
    if (grow) {
        rc = vote(fb, local, &bad, &code);
        if (rc) return rc;
        if (bad >= 0) {
            if (bad != me) fb.error("gather: rank %d cannot take part (code %lld); nothing was sent", bad, (long long)code);
            return bad == me ? local : PK_E_COMM;
        }
    }
    if (!root) {
        if (nbytes > 0) {
            rc = fb.send(fb.stage(0), (size_t)nbytes, 0);
            if (rc) return rc;
        }
        return fb.sync();
    }
    char *d_recv = fb.stage(1);
    rc = fb.group_begin();
    size_t off = (size_t)all[0];
    for (int r = 1; r < R && !rc; r++) {
        const size_t k = (size_t)all[(size_t)r * 4];
        if (k) rc = fb.recv(d_recv + off, k, r);
        off += k;
    }
    {
        const int re = fb.group_end();
        if (!rc) rc = re;
    }
    if (rc) return rc;
    if (local) {  // (the root's own copy was refused without a vote scheduled: the peers are served)
        fb.sync();
        return local;
    }
    rc = fb.sync();  // (as above: the receives first, under the deadline)
    if (rc) return rc;
    if (total > 0) {
        rc = fb.download(recv, d_recv, (size_t)total);
        if (rc) return rc;
    }
    return fb.sync();
}

}  // namespace pk_proto
#endif
