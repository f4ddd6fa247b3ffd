"""Multi-GPU sharding of the scoring path: one process per GPU, independent
units, ONE gather of the scored pixels to rank 0.

The reference scores chromosomes sequentially in one process and appends each
result to the output file (peakachu/score_genome.py:46-84); iterations share
only the model.  Here
  * `lpt_assign` deals chromosomes to ranks (longest-processing-time first),
  * `block_ranges` splits one chromosome's candidate list into contiguous
    blocks aligned to the reference batch size, so the per-batch rule of
    peakachu/scoreUtils.py:104-108 sees the same batches on any rank count,
  * `gather_records` moves the packed records to rank 0 -- over RCCL
    (pk_comm_gatherv_bytes, xGMI peer->root sends) on GPUs, or over a
    torch.distributed/gloo group on CPUs (tests).
The host rendezvous (rank, world size, the RCCL unique id broadcast) uses
torch.distributed's env:// init exactly as torch.distributed.run provides it.
"""
import os

import numpy as np

RECORD = np.dtype([("chrom", "<i4"), ("x", "<i4"), ("y", "<i4"), ("prob", "<f8"),
                   ("signal", "<f8")], align=False)


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def lpt_assign(weights, nranks):
    """Longest-processing-time-first: returns, per rank, the sorted list of
    unit indices it owns.  Deterministic (ties broken by index)."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    load = [0] * nranks
    owned = [[] for _ in range(nranks)]
    for i in order:
        r = min(range(nranks), key=lambda q: (load[q], q))
        owned[r].append(i)
        load[r] += weights[i]
    return [sorted(o) for o in owned]


def block_ranges(N, nranks, batch=100000):
    """Contiguous candidate ranges [lo, hi) per rank, cut at multiples of the
    reference batch size."""
    nb = (N + batch - 1) // batch
    cuts = [min(N, ((nb * r) // nranks) * batch) for r in range(nranks)] + [N]
    return [(cuts[r], cuts[r + 1]) for r in range(nranks)]


def pack_records(chrom_id, x, y, prob, signal):
    rec = np.empty(len(x), RECORD)
    rec["chrom"] = chrom_id
    rec["x"], rec["y"], rec["prob"], rec["signal"] = x, y, prob, signal
    return rec


class GlooTransport:
    """CPU transport for tests: torch.distributed gather of byte tensors."""

    def __init__(self):
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo")
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def gatherv(self, payload: bytes):
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(payload, out, dst=0)
        return out

    def barrier(self):
        self.dist.barrier()

    def all_failures(self, failure):
        """Every rank passes None (fine) or a message; every rank gets the list of messages."""
        out = [None] * self.world
        self.dist.all_gather_object(out, failure)
        return [m for m in out if m]

    def close(self):
        pass


class RcclTransport:
    """GPU transport: RCCL gather-v through the C ABI (pk_comm_*).  The unique
    id travels over the torch.distributed (gloo) rendezvous."""

    def __init__(self, device):
        import torch.distributed as dist
        from . import _lib
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo")
        self.dist, self._lib = dist, _lib
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        L = _lib.require_device()
        ids = [None]
        if self.rank == 0:
            buf = np.zeros(128, np.uint8)
            _lib.check(L.pk_comm_unique_id(buf), "pk_comm_unique_id")
            ids = [buf.tobytes()]
        dist.broadcast_object_list(ids, src=0)
        self.h = L.pk_comm_create(device, self.world, self.rank,
                                  np.frombuffer(ids[0], np.uint8).copy())
        if not self.h:
            raise _lib.PeakachuHipError("pk_comm_create: " + _lib.last_error())
        self._L = L

    def gatherv(self, payload: bytes):
        counts = np.zeros(self.world, np.int64)
        send = np.frombuffer(payload, np.uint8) if payload else np.zeros(1, np.uint8)
        # first exchange sizes only (cap = 0 on non-root), then the data
        sizes = [None] * self.world
        self.dist.all_gather_object(sizes, len(payload))
        total = int(sum(sizes))
        recv = np.empty(max(total, 1), np.uint8) if self.rank == 0 else None
        self._lib.check(self._L.pk_comm_gatherv_bytes(
            self.h, send.ctypes.data, len(payload), counts,
            recv.ctypes.data if recv is not None else None, total if self.rank == 0 else 0),
            "pk_comm_gatherv_bytes")
        if self.rank != 0:
            return None
        out, off = [], 0
        for r in range(self.world):
            out.append(recv[off:off + int(counts[r])].tobytes())
            off += int(counts[r])
        return out

    def barrier(self):
        self.dist.barrier()

    def all_failures(self, failure):
        """Every rank passes None (fine) or a message; every rank gets the list of messages
        (over the host rendezvous: a rank whose GPU failed can still take part)."""
        out = [None] * self.world
        self.dist.all_gather_object(out, failure)
        return [m for m in out if m]

    def close(self):
        if getattr(self, "h", None):
            self._L.pk_comm_destroy(self.h)
            self.h = None


def gather_records(local_records, transport):
    """local_records: RECORD array of this rank.  Returns on rank 0 the
    concatenation over ranks (rank order), None elsewhere."""
    parts = transport.gatherv(np.ascontiguousarray(local_records).tobytes())
    if parts is None:
        return None
    return np.concatenate([np.frombuffer(p, RECORD) for p in parts]) if parts else \
        np.empty(0, RECORD)
