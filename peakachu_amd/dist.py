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
    (pk_comm_gatherv_bytes, xGMI peer->root sends) on GPUs, or over the host
    rendezvous itself on CPUs (tests, PK_TRANSPORT=tcp).
The host rendezvous (the RCCL unique id, sizes, failure messages, barriers) is
peakachu_amd.rendezvous: TCP on 127.0.0.1, standard library only; rank and world size
come from RANK / WORLD_SIZE as peakachu_amd.launch or an external launcher (the bench driver's, srun, mpirun) set them.
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


class TcpTransport:
    """Everything over the host rendezvous (peakachu_amd.rendezvous: TCP on 127.0.0.1, standard
    library only).  The transport of the CPU tests, and of PK_TRANSPORT=tcp."""

    def __init__(self, rdzv=None):
        from .rendezvous import Rendezvous
        self.r = rdzv or Rendezvous()
        self.rank, self.world = self.r.rank, self.r.world

    def gatherv(self, payload: bytes):
        return self.r.gather(payload)

    def barrier(self):
        self.r.barrier()

    def all_failures(self, failure):
        """Every rank passes None (fine) or a message; every rank gets the list of messages."""
        return [m for m in self.r.all_gather_obj(failure) if m]

    def close(self):
        self.r.close()


class RcclTransport(TcpTransport):
    """GPU transport: the payload travels by RCCL gather-v through the C ABI (pk_comm_*: xGMI
    peer -> root sends); the unique id, the sizes, failure messages and barriers over the host
    rendezvous (a rank whose GPU failed can still say so)."""

    def __init__(self, device, rdzv=None):
        from . import _lib
        super().__init__(rdzv)
        self._lib = _lib
        L = _lib.require_device()
        buf = np.zeros(128, np.uint8)
        if self.rank == 0:
            _lib.check(L.pk_comm_unique_id(buf), "pk_comm_unique_id")
        uid = self.r.broadcast(buf.tobytes())
        self.h = L.pk_comm_create(device, self.world, self.rank, np.frombuffer(uid, np.uint8).copy())
        # all ranks learn whether every communicator exists (a rank without one must not leave
        # the others waiting inside RCCL)
        errs = self.all_failures(None if self.h else "rank %d: pk_comm_create: %s" % (self.rank, _lib.last_error()))
        if errs:
            self.close()
            raise _lib.PeakachuHipError("; ".join(errs))
        self._L = L

    def gatherv(self, payload: bytes):
        counts = np.zeros(self.world, np.int64)
        send = np.frombuffer(payload, np.uint8) if payload else np.zeros(1, np.uint8)
        # the sizes first (host side), so that rank 0 can offer a buffer of the right size
        total = sum(self.r.all_gather_obj(len(payload)))
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

    def close(self):
        if getattr(self, "h", None):
            self._L.pk_comm_destroy(self.h)
            self.h = None
        super().close()


def gather_records(local_records, transport):
    """local_records: RECORD array of this rank.  Returns on rank 0 the
    concatenation over ranks (rank order), None elsewhere."""
    parts = transport.gatherv(np.ascontiguousarray(local_records).tobytes())
    if parts is None:
        return None
    return np.concatenate([np.frombuffer(p, RECORD) for p in parts]) if parts else \
        np.empty(0, RECORD)
