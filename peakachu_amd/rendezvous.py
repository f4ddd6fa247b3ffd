"""Host-side rendezvous of the ranks of one node -- standard library only.

The reference scores chromosomes one after another in one process
(peakachu/score_genome.py:46-84); the build runs one process per GPU.  What those
processes have to tell each other on the HOST is tiny: the 128-byte RCCL id, a few
integers (payload sizes), error strings, and "I am here" (barriers).  Rounds 1-3 borrowed a
gloo process group of the deep-learning framework in the image for that; this module replaces
it: a star over TCP on
127.0.0.1, rank 0 at the centre, every operation a collective that all ranks enter in the
same order (a sequence number guards that).

Finding rank 0:
  * `PK_RDZV_ENDPOINT=host:port` -- rank 0 listens exactly there (a launcher that owns a port);
  * otherwise rank 0 listens on an ephemeral port of 127.0.0.1 and PUBLISHES it in a file the
    peers poll: `PK_RDZV_FILE` (peakachu_amd.launch sets it), or a name derived from
    MASTER_ADDR / MASTER_PORT / TORCHELASTIC_RUN_ID / TORCHELASTIC_RESTART_COUNT under the
    temp directory -- under the elastic launcher the bench driver uses, MASTER_PORT itself
    belongs to the launcher's own store, so it only serves as a name here.  The file carries a random
    token; a peer that meets a stale file (refused connection, wrong token) reads it again.

Failure behaviour (tests/test_dist.py): a rank that dies closes its socket; rank 0 sees the
end of stream in the next operation, tells everybody else (`abort` frame) and raises; the
others raise RendezvousError on the abort frame -- or on the end of stream when rank 0
itself died.  Two bounds: the SETUP (every rank arrives, the handshake) takes at most
PK_RDZV_TIMEOUT seconds (default 1800); inside a collective a peer that is silent but whose
socket is open is a peer that is still scoring -- a rank may legitimately work for a long time
between two operations, and an unbalanced run must not be lost to a clock -- so the wait there
ends with the socket (a process that dies closes it at once on one host; TCP keep-alive covers
the rest) or after PK_RDZV_OP_TIMEOUT seconds (default 24 h).

The published file holds the token that admits a rank: it is created exclusively
(O_EXCL | O_NOFOLLOW, mode 0600: a planted symlink is refused, other users cannot read it) and a
peer only trusts a regular file of its own user that nobody else can read or write.
"""
import json
import os
import secrets
import socket
import struct
import tempfile
import time

MAGIC = b"PKRZ1\0"
_HDR = struct.Struct("<BxxxIQ")  # op, sequence number, payload bytes
OP_ALLGATHER, OP_GATHER, OP_BCAST, OP_ABORT = 1, 2, 3, 255


class RendezvousError(RuntimeError):
    pass


def default_file():
    """Where rank 0 publishes its port when nobody named a file or an endpoint."""
    name = "pk_rdzv_%d_%s_%s_%s_%s.json" % (
        os.getuid(), os.environ.get("MASTER_ADDR", "local"), os.environ.get("MASTER_PORT", "0"),
        os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
    return os.path.join(tempfile.gettempdir(), "".join(c if c.isalnum() or c in "._-" else "_" for c in name))


def _keepalive(sock):
    """A peer on another host that vanishes without closing its socket is noticed within minutes."""
    try:
        sock.setsockopt(socket.SOL_SOCKET, socket.SO_KEEPALIVE, 1)
        for name, val in (("TCP_KEEPIDLE", 60), ("TCP_KEEPINTVL", 20), ("TCP_KEEPCNT", 6)):
            if hasattr(socket, name):
                sock.setsockopt(socket.IPPROTO_TCP, getattr(socket, name), val)
    except OSError:
        pass


def _read_published(path):
    """The endpoint rank 0 published -- from a regular file of this user that no one else can read
    or write (it holds the token that admits a rank); anything else is treated like a missing file."""
    fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
    try:
        st = os.fstat(fd)
        import stat
        if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise ValueError("%s is not a private file of this user" % path)
        with os.fdopen(fd, "r") as fh:
            fd = -1
            return json.load(fh)
    finally:
        if fd >= 0:
            os.close(fd)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        part = sock.recv(min(1 << 20, n - len(buf)))
        if not part:
            raise ConnectionError("peer closed the connection")
        buf += part
    return bytes(buf)


def _send_frame(sock, op, seq, payload=b""):
    sock.sendall(_HDR.pack(op, seq, len(payload)) + payload)


def _recv_frame(sock):
    op, seq, n = _HDR.unpack(_recv_exact(sock, _HDR.size))
    return op, seq, (_recv_exact(sock, n) if n else b"")


def _pack_list(parts):
    return b"".join(struct.pack("<Q", len(p)) + p for p in parts)


def _unpack_list(blob, count):
    out, o = [], 0
    for _ in range(count):
        (n,) = struct.unpack_from("<Q", blob, o)
        out.append(blob[o + 8:o + 8 + n])
        o += 8 + n
    return out


class Rendezvous:
    def __init__(self, rank=None, world=None, timeout=None):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.timeout = float(os.environ.get("PK_RDZV_TIMEOUT", "1800")) if timeout is None else float(timeout)
        # (an explicit `timeout` bounds the collectives too: tests; otherwise they wait for the socket)
        self.op_timeout = float(os.environ.get("PK_RDZV_OP_TIMEOUT", "86400")) if timeout is None else float(timeout)
        self._seq = 0
        self._peers = {}      # rank 0: rank -> socket
        self._sock = None     # other ranks: the socket to rank 0
        self._listener = None
        self._file = None
        if self.world < 1 or not (0 <= self.rank < self.world):
            raise RendezvousError("bad rank %d / world size %d" % (self.rank, self.world))
        if self.world > 1:
            (self._serve if self.rank == 0 else self._connect)()

    # ------------------------------------------------------------------ setup
    def _serve(self):
        ep = os.environ.get("PK_RDZV_ENDPOINT")
        ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        token = secrets.token_hex(16)
        if ep:
            host, _, port = ep.rpartition(":")
            ls.bind((host or "127.0.0.1", int(port)))
            token = os.environ.get("PK_RDZV_TOKEN", "")
        else:
            ls.bind(("127.0.0.1", 0))
        ls.listen(self.world)
        self._listener = ls
        if not ep:
            self._file = os.environ.get("PK_RDZV_FILE") or default_file()
            tmp = "%s.%s.tmp" % (self._file, secrets.token_hex(8))
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
            with os.fdopen(fd, "w") as fh:
                json.dump({"host": "127.0.0.1", "port": ls.getsockname()[1], "token": token,
                           "world": self.world, "pid": os.getpid()}, fh)
            os.replace(tmp, self._file)  # atomically: a peer sees the old file or the new one
        deadline = time.monotonic() + self.timeout
        while len(self._peers) < self.world - 1:
            ls.settimeout(max(0.05, deadline - time.monotonic()))
            try:
                c, _ = ls.accept()
            except socket.timeout:
                missing = sorted(set(range(1, self.world)) - set(self._peers))
                self.close()
                raise RendezvousError("rendezvous: ranks %s did not arrive within %.0f s" % (missing, self.timeout))
            try:
                c.settimeout(10.0)
                hello = _recv_exact(c, len(MAGIC) + 8 + 32)
                r, wd = struct.unpack_from("<II", hello, len(MAGIC))
                tok = hello[len(MAGIC) + 8:].rstrip(b"\0").decode("ascii", "replace")
                good = hello.startswith(MAGIC) and wd == self.world and 0 < r < self.world and \
                    r not in self._peers and tok == token
                c.sendall(b"\1" if good else b"\0")
                if not good:
                    c.close()
                    continue
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                _keepalive(c)
                c.settimeout(self.op_timeout)
                self._peers[r] = c
            except (OSError, struct.error):
                c.close()

    def _connect(self):
        ep = os.environ.get("PK_RDZV_ENDPOINT")
        path = None if ep else (os.environ.get("PK_RDZV_FILE") or default_file())
        deadline = time.monotonic() + self.timeout
        last = "no attempt made"
        while time.monotonic() < deadline:
            try:
                if ep:
                    host, _, port = ep.rpartition(":")
                    info = {"host": host or "127.0.0.1", "port": int(port), "token": os.environ.get("PK_RDZV_TOKEN", "")}
                else:
                    info = _read_published(path)
                s = socket.create_connection((info["host"], int(info["port"])), timeout=5.0)
                s.sendall(MAGIC + struct.pack("<II", self.rank, self.world) +
                          info["token"].encode("ascii").ljust(32, b"\0"))
                if _recv_exact(s, 1) == b"\1":
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    _keepalive(s)
                    s.settimeout(self.op_timeout)
                    self._sock = s
                    return
                s.close()
                last = "rank 0 refused the handshake (stale file or another job?)"
            except (OSError, ValueError, KeyError) as e:  # file not there yet / stale / half-written
                last = "%s: %s" % (type(e).__name__, e)
            time.sleep(0.05)
        raise RendezvousError("rendezvous: rank %d could not reach rank 0 within %.0f s (%s)"
                              % (self.rank, self.timeout, last))

    # ------------------------------------------------------------ collectives
    def _abort(self, why):
        for r, c in list(self._peers.items()):
            try:
                _send_frame(c, OP_ABORT, self._seq, why.encode("utf-8", "replace"))
            except OSError:
                pass
        self.close()
        raise RendezvousError(why)

    def _exchange(self, op, payload, reply_of):
        """One collective.  Rank 0 collects a frame from every peer and answers each with
        reply_of(parts, rank); the peers send and wait for their answer."""
        self._seq += 1
        if self.world == 1:
            return reply_of([payload], 0)
        if self.rank != 0:
            try:
                _send_frame(self._sock, op, self._seq, payload)
                rop, rseq, blob = _recv_frame(self._sock)
            except (OSError, struct.error) as e:
                self.close()
                raise RendezvousError("rendezvous: rank %d lost rank 0 (%s: %s)" % (self.rank, type(e).__name__, e))
            if rop == OP_ABORT:
                self.close()
                raise RendezvousError("rendezvous aborted: " + blob.decode("utf-8", "replace"))
            if rop != op or rseq != self._seq:
                self.close()
                raise RendezvousError("rendezvous: rank %d is out of step (operation %d/%d, sequence %d/%d)"
                                      % (self.rank, rop, op, rseq, self._seq))
            return blob
        parts = [payload] + [b""] * (self.world - 1)
        for r in range(1, self.world):
            try:
                pop, pseq, blob = _recv_frame(self._peers[r])
            except (OSError, struct.error) as e:
                self._abort("rank %d left the job (%s: %s)" % (r, type(e).__name__, e))
            if pop != op or pseq != self._seq:
                self._abort("rank %d is out of step (operation %d/%d, sequence %d/%d)" % (r, pop, op, pseq, self._seq))
            parts[r] = blob
        for r in range(1, self.world):
            try:
                _send_frame(self._peers[r], op, self._seq, reply_of(parts, r))
            except OSError as e:
                self._abort("rank %d left the job (%s: %s)" % (r, type(e).__name__, e))
        return reply_of(parts, 0)

    def all_gather(self, payload: bytes):
        blob = self._exchange(OP_ALLGATHER, payload, lambda parts, r: _pack_list(parts))
        return _unpack_list(blob, self.world)

    def gather(self, payload: bytes):
        """Rank 0 gets the list of all ranks' payloads, the others None."""
        blob = self._exchange(OP_GATHER, payload, lambda parts, r: _pack_list(parts) if r == 0 else b"")
        return _unpack_list(blob, self.world) if self.rank == 0 else None

    def broadcast(self, payload=None):
        """Rank 0's payload, on every rank."""
        return self._exchange(OP_BCAST, payload if self.rank == 0 else b"", lambda parts, r: parts[0])

    def barrier(self):
        self._exchange(OP_ALLGATHER, b"", lambda parts, r: b"")

    def all_gather_obj(self, obj):
        return [json.loads(b.decode("utf-8")) for b in self.all_gather(json.dumps(obj).encode("utf-8"))]

    def close(self):
        for c in list(self._peers.values()) + [self._sock, self._listener]:
            if c is not None:
                try:
                    c.close()
                except OSError:
                    pass
        self._peers, self._sock, self._listener = {}, None, None
        if self._file:
            try:
                os.unlink(self._file)
            except OSError:
                pass
            self._file = None

    __del__ = close
