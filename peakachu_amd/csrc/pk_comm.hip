// pk_comm.hip -- the one multi-GPU exchange of the scoring path: a gather-v
// of the scored pixels (row, col, prob, signal) to rank 0 over RCCL.
//
// The reference scores chromosomes one after another in a single process
// (peakachu/score_genome.py:46-84) and appends each result to one file
// (peakachu/scoreUtils.py:127-135).  Here every rank (one process per GPU)
// scores its own chromosomes / candidate blocks; nothing is exchanged on the
// data path until the end, when rank 0 collects the pixels with p > thre.
// xGMI is point-to-point, so the gather is N-1 direct peer->root transfers
// (grouped ncclSend/ncclRecv), preceded by an all-gather of the counts.
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>

#include "pk_common.h"
#include "pk_comm_protocol.h"


struct pk_comm {
    int device, nranks, rank;
    ncclComm_t comm;
    int64_t *d_counts;  // device [4 * nranks]: what an all-gather of up to four words per rank returns
    int64_t *d_mine;    // device [4]: this rank's words
    // staging areas that live as long as the communicator and only ever grow: a gather is
    // part of the timed step of a multi-GPU run and must not allocate
    void *stage[2];     // [0] send bytes / root's gathered pixels, [1] root's gathered bytes
    size_t stage_cap[2];
    bool aborted;       // a wait ran out: the communicator was torn down (ncclCommAbort), every later call is refused
};

// makes stage[i] at least `bytes` large (grows by half again so that repeated gathers of
// slightly different sizes settle after a few calls)
static int comm_reserve(pk_comm *c, int i, size_t bytes)
{
    if (bytes <= c->stage_cap[i]) return PK_OK;
    size_t want = bytes + bytes / 2 + 4096;
    if (c->stage[i]) PK_HIP(hipFree(c->stage[i]));
    c->stage[i] = nullptr;
    c->stage_cap[i] = 0;
    PK_HIP(hipMalloc(&c->stage[i], want));
    c->stage_cap[i] = want;
    return PK_OK;
}

static_assert(sizeof(ncclUniqueId) == 128, "pk_comm_unique_id assumes a 128-byte id");

// The protocol (pk_comm_protocol.h: who tells whom what before anybody sends) over RCCL and HIP.
struct rccl_fabric {
    pk_comm *c;
    hipStream_t s;
    int rank() const { return c->rank; }
    int nranks() const { return c->nranks; }
    int nccl(ncclResult_t r, const char *what)
    {
        if (r == ncclSuccess) return PK_OK;
        pk_set_error("%s failed: %s", what, ncclGetErrorString(r));
        return PK_E_COMM;
    }
    int hip(hipError_t e, const char *what)
    {
        if (e == hipSuccess) return PK_OK;
        pk_set_error("%s failed: %s", what, hipGetErrorString(e));
        return PK_E_HIP;
    }
    // The one place a rank waits for its peers: the stream holds a collective / the sends / the
    // receives.  hipStreamSynchronize would wait for ever for a peer that died between its "ready"
    // and its send; this polls the stream (and RCCL's own asynchronous error state) against
    // PK_COMM_TIMEOUT and, when that runs out, aborts the communicator so that the stream drains.
    int wait(const char *what)
    {
        const double limit = pk_proto::comm_timeout_seconds();
        const int rc = pk_proto::wait_until(
            [&]() -> int {
                const hipError_t e = hipStreamQuery(s);
                if (e == hipSuccess) return 1;
                if (e != hipErrorNotReady) {
                    pk_set_error("%s failed: %s", what, hipGetErrorString(e));
                    return PK_E_HIP;
                }
                ncclResult_t ar = ncclSuccess;
                if (ncclCommGetAsyncError(c->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
                    pk_set_error("%s: RCCL reports %s", what, ncclGetErrorString(ar));
                    return PK_E_COMM;
                }
                return 0;
            },
            limit);
        (void)hipGetLastError();  // (hipErrorNotReady of the polls is not an error of the next launch)
        if (rc == pk_proto::WAIT_TIMED_OUT || rc == PK_E_COMM) {
            if (rc == pk_proto::WAIT_TIMED_OUT)
                pk_set_error("%s: rank %d of %d waited %.0f s (PK_COMM_TIMEOUT) for its peers -- one of them is gone or "
                             "stuck; the communicator was aborted, this process should report and exit",
                             what, c->rank, c->nranks, limit);
            ncclCommAbort(c->comm);
            c->comm = nullptr;
            c->aborted = true;
            return PK_E_COMM;
        }
        return rc;
    }
    int allgather(const int64_t *mine, int words, int64_t *all)
    {
        if (c->aborted) return refused();
        int rc = hip(hipMemcpyAsync(c->d_mine, mine, 8 * (size_t)words, hipMemcpyHostToDevice, s), "count upload");
        if (!rc) rc = nccl(ncclAllGather(c->d_mine, c->d_counts, (size_t)words, ncclInt64, c->comm, s), "ncclAllGather");
        // (the bounded wait comes BEFORE the copy to pageable host memory: such a copy may wait for the
        // stream inside the HIP call, where no deadline reaches)
        if (!rc) rc = wait("all-gather of the counts");
        if (!rc)
            rc = hip(hipMemcpyAsync(all, c->d_counts, 8 * (size_t)words * (size_t)c->nranks, hipMemcpyDeviceToHost, s),
                     "count download");
        if (!rc) rc = hip(hipStreamSynchronize(s), "count download");  // (this rank's own copy: no peer involved)
        return rc;
    }
    int refused()
    {
        pk_set_error("the communicator was aborted after a timed-out wait (rank %d of %d): no further exchange", c->rank,
                     c->nranks);
        return PK_E_COMM;
    }
    size_t stage_cap(int i) const { return c->stage_cap[i]; }
    int reserve(int i, size_t bytes) { return comm_reserve(c, i, bytes); }
    char *stage(int i) { return static_cast<char *>(c->stage[i]); }
    int copy_dd(void *dst, const void *src, size_t n) { return hip(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s), "local copy"); }
    int upload(void *dst, const void *src, size_t n) { return hip(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, s), "upload"); }
    int download(void *dst, const void *src, size_t n) { return hip(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, s), "download"); }
    int group_begin() { return c->aborted ? refused() : nccl(ncclGroupStart(), "ncclGroupStart"); }
    int group_end() { return c->aborted ? refused() : nccl(ncclGroupEnd(), "ncclGroupEnd"); }
    int send(const void *p, size_t n, int peer)
    {
        return c->aborted ? refused() : nccl(ncclSend(p, n, ncclUint8, peer, c->comm, s), "ncclSend");
    }
    int recv(void *p, size_t n, int peer)
    {
        return c->aborted ? refused() : nccl(ncclRecv(p, n, ncclUint8, peer, c->comm, s), "ncclRecv");
    }
    int sync() { return c->aborted ? refused() : wait("gather of the scored pixels"); }
    template <typename... A>
    void error(const char *fmt, A... a) { pk_set_error(fmt, a...); }
};

extern "C" int pk_comm_unique_id(uint8_t id[128])
{
    if (!id) return PK_E_INVALID;
    ncclUniqueId u;
    ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) {
        pk_set_error("ncclGetUniqueId failed: %s", ncclGetErrorString(r));
        return PK_E_COMM;
    }
    memcpy(id, &u, 128);
    return PK_OK;
}

extern "C" pk_comm *pk_comm_create(int device, int nranks, int rank, const uint8_t id[128])
{
    PK_DEV_LOCK(device);
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) {
        pk_set_error("pk_comm_create: bad arguments");
        return nullptr;
    }
    pk_device_ctx *ctx = pk_ctx(device);
    if (!ctx) return nullptr;
    pk_comm *c = new pk_comm();
    c->device = device;
    c->nranks = nranks;
    c->rank = rank;
    c->comm = nullptr;
    c->d_counts = nullptr;
    c->d_mine = nullptr;
    c->stage[0] = c->stage[1] = nullptr;
    c->stage_cap[0] = c->stage_cap[1] = 0;
    c->aborted = false;
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        pk_set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r));
        delete c;
        return nullptr;
    }
    if (hipMalloc((void **)&c->d_counts, sizeof(int64_t) * 4 * (size_t)nranks) != hipSuccess ||
        hipMalloc((void **)&c->d_mine, sizeof(int64_t) * 4) != hipSuccess) {
        pk_set_error("pk_comm_create: device allocation failed");
        ncclCommDestroy(c->comm);
        delete c;
        return nullptr;
    }
    return c;
}

extern "C" int pk_comm_ranks(pk_comm *c)
{
    PK_DEV_LOCK(c ? c->device : 0);
    if (!c || c->aborted || !c->comm) {
        pk_set_error("pk_comm_ranks: no live communicator");
        return PK_E_COMM;
    }
    int n = 0;
    const ncclResult_t r = ncclCommCount(c->comm, &n);
    if (r != ncclSuccess) {
        pk_set_error("ncclCommCount failed: %s", ncclGetErrorString(r));
        return PK_E_COMM;
    }
    return n;
}

extern "C" int pk_comm_version(int *rccl)
{
    if (!rccl) return PK_E_INVALID;
    *rccl = -1;
    const ncclResult_t r = ncclGetVersion(rccl);
    if (r != ncclSuccess) {
        pk_set_error("ncclGetVersion failed: %s", ncclGetErrorString(r));
        return PK_E_COMM;
    }
    return PK_OK;
}

extern "C" void pk_comm_destroy(pk_comm *c)
{
    PK_DEV_LOCK(c ? c->device : 0);
    if (!c) return;
    hipSetDevice(c->device);
    if (c->d_counts) hipFree(c->d_counts);
    if (c->d_mine) hipFree(c->d_mine);
    for (int i = 0; i < 2; i++)
        if (c->stage[i]) hipFree(c->stage[i]);
    if (c->comm) ncclCommDestroy(c->comm);
    delete c;
}

extern "C" int pk_comm_gather_scored(pk_comm *c, pk_cands *cd, int64_t *counts, int64_t cap,
                                     int32_t *ox, int32_t *oy, double *op, double *osignal)
{
    PK_DEV_LOCK(c ? c->device : 0);
    if (!c || !cd || cd->device != c->device) {
        pk_set_error("pk_comm_gather_scored: bad arguments");
        return PK_E_INVALID;
    }
    pk_device_ctx *ctx = pk_ctx(c->device);
    if (!ctx) return PK_E_NODEVICE;
    rccl_fabric fb{c, ctx->stream};
    // (cd->n_out: the host copy pk_score_run made of this rank's count)
    return pk_proto::gather_scored(fb, cd->n_out, cd->ox, cd->oy, cd->op, cd->osig, counts, cap, ox, oy, op,
                                   osignal);
}

extern "C" int pk_comm_gatherv_bytes(pk_comm *c, const void *send, int64_t nbytes, int64_t *counts,
                                     void *recv, int64_t cap)
{
    PK_DEV_LOCK(c ? c->device : 0);
    if (!c || nbytes < 0 || (nbytes > 0 && !send)) {
        pk_set_error("pk_comm_gatherv_bytes: bad arguments");
        return PK_E_INVALID;
    }
    pk_device_ctx *ctx = pk_ctx(c->device);
    if (!ctx) return PK_E_NODEVICE;
    rccl_fabric fb{c, ctx->stream};
    return pk_proto::gatherv_bytes(fb, send, nbytes, counts, recv, cap);
}
