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

extern std::recursive_mutex g_api_mu;
#define PK_API_LOCK std::lock_guard<std::recursive_mutex> api_lock__(g_api_mu)

struct pk_comm {
    int device, nranks, rank;
    ncclComm_t comm;
    int64_t *d_counts;  // device [2 * nranks]: the counts, then the ranks' status words (comm_agree)
    int64_t *d_mine;    // device scalar: this rank's byte count (pk_comm_gatherv_bytes)
    // staging areas that live as long as the communicator and only ever grow: a gather is
    // part of the timed step of a multi-GPU run and must not allocate
    void *stage[2];     // [0] send bytes / root's gathered pixels, [1] root's gathered bytes
    size_t stage_cap[2];
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

#define PK_NCCL(call)                                                                  \
    do {                                                                               \
        ncclResult_t r__ = (call);                                                     \
        if (r__ != ncclSuccess) {                                                      \
            pk_set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(r__),      \
                         __FILE__, __LINE__);                                          \
            return PK_E_COMM;                                                          \
        }                                                                              \
    } while (0)

static_assert(sizeof(ncclUniqueId) == 128, "pk_comm_unique_id assumes a 128-byte id");

// Every rank tells every other whether it can go through with the transfer that follows
// (0 = yes, else its error code): a rank that cannot post its half of a send/recv pair --
// the root without a staging area, a result larger than the caller's buffers -- must not
// simply return, or its peers sit in ncclSend / ncclRecv for ever.  All ranks call this at
// the same point and all take the same decision: *bad = the first rank with a non-zero
// word (-1 if none), *code = its word.
static int comm_agree(pk_comm *c, hipStream_t s, int local_rc, int *bad, int *code)
{
    const int R = c->nranks;
    const int64_t mine = local_rc;
    std::vector<int64_t> h((size_t)R, 0);
    PK_HIP(hipMemcpyAsync(c->d_mine, &mine, 8, hipMemcpyHostToDevice, s));
    PK_NCCL(ncclAllGather(c->d_mine, c->d_counts + R, 1, ncclInt64, c->comm, s));
    PK_HIP(hipMemcpyAsync(h.data(), c->d_counts + R, 8 * (size_t)R, hipMemcpyDeviceToHost, s));
    PK_HIP(hipStreamSynchronize(s));
    *bad = -1;
    *code = 0;
    for (int r = 0; r < R; r++)
        if (h[(size_t)r] != 0) {
            *bad = r;
            *code = (int)h[(size_t)r];
            break;
        }
    return PK_OK;
}

extern "C" int pk_comm_unique_id(uint8_t id[128])
{
    if (!id) return PK_E_INVALID;
    ncclUniqueId u;
    PK_NCCL(ncclGetUniqueId(&u));
    memcpy(id, &u, 128);
    return PK_OK;
}

extern "C" pk_comm *pk_comm_create(int device, int nranks, int rank, const uint8_t id[128])
{
    PK_API_LOCK;
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
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        pk_set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r));
        delete c;
        return nullptr;
    }
    if (hipMalloc((void **)&c->d_counts, sizeof(int64_t) * 2 * (size_t)nranks) != hipSuccess ||
        hipMalloc((void **)&c->d_mine, sizeof(int64_t)) != hipSuccess) {
        pk_set_error("pk_comm_create: device allocation failed");
        ncclCommDestroy(c->comm);
        delete c;
        return nullptr;
    }
    return c;
}

extern "C" void pk_comm_destroy(pk_comm *c)
{
    PK_API_LOCK;
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
    PK_API_LOCK;
    if (!c || !cd || cd->device != c->device) {
        pk_set_error("pk_comm_gather_scored: bad arguments");
        return PK_E_INVALID;
    }
    pk_device_ctx *ctx = pk_ctx(c->device);
    if (!ctx) return PK_E_NODEVICE;
    hipStream_t s = ctx->stream;
    const int R = c->nranks;
    // 1. counts: n_out_dev holds this rank's count after pk_score_run
    PK_NCCL(ncclAllGather(cd->n_out_dev, c->d_counts, 1, ncclInt64, c->comm, s));
    std::vector<int64_t> h_counts((size_t)R);
    PK_HIP(hipMemcpyAsync(h_counts.data(), c->d_counts, sizeof(int64_t) * (size_t)R,
                          hipMemcpyDeviceToHost, s));
    PK_HIP(hipStreamSynchronize(s));
    int64_t total = 0;
    for (int r = 0; r < R; r++) total += h_counts[(size_t)r];
    if (counts)
        for (int r = 0; r < R; r++) counts[r] = h_counts[(size_t)r];

    // 2. what only the root can know -- whether the result fits the caller's buffers and its
    // staging area could be made -- is agreed on by all ranks BEFORE anybody sends
    const size_t t1 = ((size_t)(total > 0 ? total : 1) + 1) & ~(size_t)1;
    int local_rc = PK_OK;
    if (c->rank == 0) {
        if (total > cap || (total > 0 && (!ox || !oy || !op || !osignal))) {
            pk_set_error("pk_comm_gather_scored: %lld pixels exceed the root capacity %lld",
                         (long long)total, (long long)cap);
            local_rc = PK_E_INVALID;
        } else {
            local_rc = comm_reserve(c, 0, t1 * 24);  // one staging area [x | y | p | signal]
        }
    }
    {
        int bad = -1, code = 0;
        const int rca = comm_agree(c, s, local_rc, &bad, &code);
        if (rca) return rca;
        if (bad >= 0) {
            if (bad != c->rank)
                pk_set_error("pk_comm_gather_scored: rank %d cannot take part in the gather (code %d); "
                             "nothing was sent", bad, code);
            return bad == c->rank ? local_rc : PK_E_COMM;
        }
    }

    if (c->rank != 0) {
        // 3. peer -> root: four typed sends in one group
        const size_t k = (size_t)h_counts[(size_t)c->rank];
        if (k > 0) {
            PK_NCCL(ncclGroupStart());
            PK_NCCL(ncclSend(cd->ox, k, ncclInt32, 0, c->comm, s));
            PK_NCCL(ncclSend(cd->oy, k, ncclInt32, 0, c->comm, s));
            PK_NCCL(ncclSend(cd->op, k, ncclFloat64, 0, c->comm, s));
            PK_NCCL(ncclSend(cd->osig, k, ncclFloat64, 0, c->comm, s));
            PK_NCCL(ncclGroupEnd());
        }
        PK_HIP(hipStreamSynchronize(s));
        return PK_OK;
    }

    // root: 8-byte aligned parts of the staging area
    int32_t *gx = static_cast<int32_t *>(c->stage[0]), *gy = gx + t1;
    double *gp = reinterpret_cast<double *>(gy + t1), *gs = gp + t1;
    int rc = PK_OK;
    do {
        const size_t k0 = (size_t)h_counts[0];
        if (k0 > 0) {
            if (hipMemcpyAsync(gx, cd->ox, k0 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess ||
                hipMemcpyAsync(gy, cd->oy, k0 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess ||
                hipMemcpyAsync(gp, cd->op, k0 * 8, hipMemcpyDeviceToDevice, s) != hipSuccess ||
                hipMemcpyAsync(gs, cd->osig, k0 * 8, hipMemcpyDeviceToDevice, s) != hipSuccess) {
                pk_set_error("pk_comm_gather_scored: local copy failed");
                rc = PK_E_HIP;
                break;
            }
        }
        ncclResult_t nr = ncclGroupStart();
        size_t off = k0;
        for (int r = 1; r < R && nr == ncclSuccess; r++) {
            const size_t k = (size_t)h_counts[(size_t)r];
            if (k == 0) continue;
            nr = ncclRecv(gx + off, k, ncclInt32, r, c->comm, s);
            if (nr == ncclSuccess) nr = ncclRecv(gy + off, k, ncclInt32, r, c->comm, s);
            if (nr == ncclSuccess) nr = ncclRecv(gp + off, k, ncclFloat64, r, c->comm, s);
            if (nr == ncclSuccess) nr = ncclRecv(gs + off, k, ncclFloat64, r, c->comm, s);
            off += k;
        }
        ncclResult_t ne = ncclGroupEnd();
        if (nr != ncclSuccess || ne != ncclSuccess) {
            pk_set_error("pk_comm_gather_scored: RCCL recv failed: %s",
                         ncclGetErrorString(nr != ncclSuccess ? nr : ne));
            rc = PK_E_COMM;
            break;
        }
        if (total > 0) {
            const size_t t = (size_t)total;
            if (hipMemcpyAsync(ox, gx, t * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipMemcpyAsync(oy, gy, t * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipMemcpyAsync(op, gp, t * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipMemcpyAsync(osignal, gs, t * 8, hipMemcpyDeviceToHost, s) != hipSuccess) {
                pk_set_error("pk_comm_gather_scored: download failed");
                rc = PK_E_HIP;
                break;
            }
        }
        if (hipStreamSynchronize(s) != hipSuccess) {
            pk_set_error("pk_comm_gather_scored: stream sync failed");
            rc = PK_E_HIP;
        }
    } while (0);
    return rc;
}

extern "C" int pk_comm_gatherv_bytes(pk_comm *c, const void *send, int64_t nbytes, int64_t *counts,
                                     void *recv, int64_t cap)
{
    PK_API_LOCK;
    if (!c || nbytes < 0 || (nbytes > 0 && !send)) {
        pk_set_error("pk_comm_gatherv_bytes: bad arguments");
        return PK_E_INVALID;
    }
    pk_device_ctx *ctx = pk_ctx(c->device);
    if (!ctx) return PK_E_NODEVICE;
    hipStream_t s = ctx->stream;
    const int R = c->nranks;
    int64_t *d_mine = c->d_mine;
    uint8_t *d_send = nullptr, *d_recv = nullptr;
    // a rank that cannot stage its bytes still takes part in the count exchange (with a count
    // of zero) and reports the failure in the agreement that follows
    int local_rc = comm_reserve(c, 0, (size_t)(nbytes > 0 ? nbytes : 1));
    if (local_rc) nbytes = 0;
    int rc = PK_OK;
    d_send = static_cast<uint8_t *>(c->stage[0]);
    std::vector<int64_t> h_counts((size_t)R);
    do {
        if (hipMemcpyAsync(d_mine, &nbytes, 8, hipMemcpyHostToDevice, s) != hipSuccess) {
            pk_set_error("pk_comm_gatherv_bytes: upload failed");
            rc = PK_E_HIP;  // (the stream itself is gone: the peers will see the communicator fail)
            break;
        }
        if (nbytes > 0 &&
            hipMemcpyAsync(d_send, send, (size_t)nbytes, hipMemcpyHostToDevice, s) != hipSuccess) {
            pk_set_error("pk_comm_gatherv_bytes: upload failed");
            local_rc = PK_E_HIP;
        }
        ncclResult_t nr = ncclAllGather(d_mine, c->d_counts, 1, ncclInt64, c->comm, s);
        if (nr != ncclSuccess ||
            hipMemcpyAsync(h_counts.data(), c->d_counts, 8 * (size_t)R, hipMemcpyDeviceToHost, s) !=
                hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            pk_set_error("pk_comm_gatherv_bytes: count exchange failed");
            rc = PK_E_COMM;
            break;
        }
        int64_t total = 0;
        for (int r = 0; r < R; r++) total += h_counts[(size_t)r];
        if (counts)
            for (int r = 0; r < R; r++) counts[r] = h_counts[(size_t)r];
        // the root's part (staging area, capacity) is settled before anybody sends
        if (c->rank == 0 && !local_rc) {
            if (total > cap || (total > 0 && !recv)) {
                pk_set_error("pk_comm_gatherv_bytes: %lld bytes exceed the root capacity %lld",
                             (long long)total, (long long)cap);
                local_rc = PK_E_INVALID;
            } else {
                local_rc = comm_reserve(c, 1, (size_t)(total > 0 ? total : 1));
            }
        }
        {
            int bad = -1, code = 0;
            rc = comm_agree(c, s, local_rc, &bad, &code);
            if (rc) break;
            if (bad >= 0) {
                if (bad != c->rank)
                    pk_set_error("pk_comm_gatherv_bytes: rank %d cannot take part in the gather (code %d); "
                                 "nothing was sent", bad, code);
                rc = bad == c->rank ? local_rc : PK_E_COMM;
                break;
            }
        }
        if (c->rank != 0) {
            if (nbytes > 0) {
                nr = ncclSend(d_send, (size_t)nbytes, ncclUint8, 0, c->comm, s);
                if (nr != ncclSuccess) {
                    pk_set_error("ncclSend failed: %s", ncclGetErrorString(nr));
                    rc = PK_E_COMM;
                    break;
                }
            }
            if (hipStreamSynchronize(s) != hipSuccess) rc = PK_E_HIP;
            break;
        }
        d_recv = static_cast<uint8_t *>(c->stage[1]);
        if (nbytes > 0 &&
            hipMemcpyAsync(d_recv, d_send, (size_t)nbytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            rc = PK_E_HIP;
            break;
        }
        nr = ncclGroupStart();
        size_t off = (size_t)h_counts[0];
        for (int r = 1; r < R && nr == ncclSuccess; r++) {
            const size_t k = (size_t)h_counts[(size_t)r];
            if (k) nr = ncclRecv(d_recv + off, k, ncclUint8, r, c->comm, s);
            off += k;
        }
        ncclResult_t ne = ncclGroupEnd();
        if (nr != ncclSuccess || ne != ncclSuccess) {
            pk_set_error("pk_comm_gatherv_bytes: RCCL recv failed");
            rc = PK_E_COMM;
            break;
        }
        if (total > 0 &&
            hipMemcpyAsync(recv, d_recv, (size_t)total, hipMemcpyDeviceToHost, s) != hipSuccess)
            rc = PK_E_HIP;
        if (hipStreamSynchronize(s) != hipSuccess) rc = PK_E_HIP;
    } while (0);
    return rc;
}
