"""A small read-only HDF5 reader: just what a `.cool` / `.mcool` file needs.

The reference opens contact maps with `cooler.Cooler(path)` (peakachu/score_genome.py:26-35),
i.e. h5py + libhdf5.  Neither exists in the deployment image, and the contact map is the one
input of the scoring path that cannot be converted without them -- so this module parses the
container itself, in pure Python + numpy, after the HDF5 File Format Specification (v2/v3):

  * superblock versions 0-3; object headers version 1 and 2 ('OHDR'), continuation blocks;
  * groups stored as symbol tables (v1 B-tree 'TREE' + 'SNOD' nodes + local heap) -- what
    h5py writes by default and therefore what cooler's files are -- and as compact link
    messages (libver='latest'); dense link storage (fractal heaps) is refused;
  * datasets: compact, contiguous and chunked (v1 B-tree chunk index, layout message v3;
    the v4 layouts of libver='latest': single chunk, implicit, fixed array, extensible array);
    filters deflate and shuffle (cooler: gzip 6 + shuffle);
  * datatypes: fixed-point, IEEE float, fixed-length string, enum (returned as its integer
    base), variable-length string (attributes; global heap);
  * attributes stored in the object header (messages v1-v3).
Everything else raises `H5Unsupported` naming the feature.  Pinned by files the genuine
library wrote (tests/golden/cool_small*.cool, tools/make_cool_fixture.py).
"""
import os
import struct
import zlib

_POOL = None


def _pool_size():
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(8, n))


def _inflate_pool():
    """Worker threads for chunk inflation (made on first use; sized to the CPUs this process
    may use, at most 8)."""
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        try:
            n = len(os.sched_getaffinity(0))
        except AttributeError:
            n = os.cpu_count() or 1
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(8, n)), thread_name_prefix="h5lite-inflate")
    return _POOL

import numpy as np


_NATIVE = False  # False: not looked up yet; None: not available


def _native_unfilter():
    """pk_host_unfilter_chunks of the package's C library (inflate + un-shuffle of chunks on host
    threads), or None when the library has not been built: the Python pipeline then does it."""
    global _NATIVE
    if _NATIVE is False:
        try:
            from . import _lib
            _NATIVE = _lib.load().pk_host_unfilter_chunks
        except Exception:
            _NATIVE = None
    return _NATIVE

SIGNATURE = b"\x89HDF\r\n\x1a\n"


class H5Unsupported(Exception):
    pass


class H5FormatError(Exception):
    pass


class _Reader:
    def __init__(self, path):
        self.fh = open(path, "rb")
        self.base = 0
        self.O = self.L = 8

    def close(self):
        self.fh.close()

    def at(self, addr, n):
        # (positional reads: the file position is not shared state, so that chromosomes can be
        # read on several threads at once)
        b = os.pread(self.fh.fileno(), n, self.base + addr)
        while 0 < len(b) < n:  # (a short read is legal for pread)
            more = os.pread(self.fh.fileno(), n - len(b), self.base + addr + len(b))
            if not more:
                break
            b += more
        if len(b) != n:
            raise H5FormatError("read past the end of the file (address %d, %d bytes)" % (addr, n))
        return b

    def uint(self, buf, off, size):
        return int.from_bytes(buf[off:off + size], "little")

    def undefined(self, addr):
        return addr == (1 << (8 * self.O)) - 1


# ---------------------------------------------------------------- datatypes
class _Type:
    def __init__(self, dtype=None, kind="plain", size=0, base=None, enum=None, vlen_string=False):
        self.dtype, self.kind, self.size, self.base, self.enum = dtype, kind, size, base, enum
        self.vlen_string = vlen_string


def _parse_datatype(buf, off=0):
    """-> (_Type, bytes consumed)."""
    cv = buf[off]
    cls, ver = cv & 0x0F, cv >> 4
    bits = buf[off + 1] | (buf[off + 2] << 8) | (buf[off + 3] << 16)
    size = struct.unpack_from("<I", buf, off + 4)[0]
    p = off + 8
    if cls == 0:  # fixed point
        order = ">" if bits & 1 else "<"
        signed = bool(bits & 8)
        if size not in (1, 2, 4, 8):
            raise H5Unsupported("integer of %d bytes" % size)
        return _Type(np.dtype("%s%s%d" % (order, "i" if signed else "u", size)), size=size), p + 4 - off
    if cls == 1:  # floating point
        order = ">" if bits & 1 else "<"
        if size not in (2, 4, 8):
            raise H5Unsupported("float of %d bytes" % size)
        return _Type(np.dtype("%sf%d" % (order, size)), size=size), p + 12 - off
    if cls == 3:  # fixed-length string
        return _Type(np.dtype("S%d" % size), kind="string", size=size), p - off
    if cls == 8:  # enumeration: members = bits & 0xffff; base type, names, values
        n = bits & 0xFFFF
        base, used = _parse_datatype(buf, p)
        p += used
        names = []
        for _ in range(n):
            e = buf.index(b"\x00", p)
            names.append(buf[p:e].decode("utf-8", "replace"))
            ln = e - p + 1
            p += ln if ver >= 3 else (ln + 7) // 8 * 8
        vals = np.frombuffer(buf, base.dtype, n, p)
        p += n * base.size
        return _Type(base.dtype, kind="enum", size=size, base=base,
                     enum=dict(zip(names, (int(v) for v in vals)))), p - off
    if cls == 9:  # variable length
        vtype = bits & 0x0F
        base, used = _parse_datatype(buf, p)
        return _Type(None, kind="vlen", size=size, base=base, vlen_string=(vtype == 1)), p + used - off
    if cls == 6:
        raise H5Unsupported("compound datatype")
    raise H5Unsupported("datatype class %d" % cls)


def _parse_dataspace(buf, off=0):
    ver, rank, flags = buf[off], buf[off + 1], buf[off + 2]
    if ver == 1:
        p = off + 8
    elif ver == 2:
        if buf[off + 3] == 2:  # null dataspace
            return None
        p = off + 4
    else:
        raise H5Unsupported("dataspace message version %d" % ver)
    dims = struct.unpack_from("<%dQ" % rank, buf, p) if rank else ()
    return tuple(int(d) for d in dims)


# ---------------------------------------------------------------- objects
class _Object:
    """An object header: its messages [(type, flags, bytes)]."""

    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        self.msgs = []
        r = f._r
        head = r.at(addr, 16)
        if head[:4] == b"OHDR":
            self._read_v2(addr)
        elif head[0] == 1:
            nmsgs = struct.unpack_from("<H", head, 2)[0]
            size = struct.unpack_from("<I", head, 8)[0]
            self._read_v1_block(addr + 16, size, nmsgs)
        else:
            raise H5FormatError("no object header at address %d" % addr)

    def _read_v1_block(self, addr, size, budget):
        r = self.f._r
        blocks = [(addr, size)]
        while blocks:
            a, n = blocks.pop(0)
            buf = r.at(a, n)
            p = 0
            while p + 8 <= n and len(self.msgs) < budget + 64:
                mtype, msize = struct.unpack_from("<HH", buf, p)
                flags = buf[p + 4]
                body = buf[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x10:  # continuation
                    blocks.append((r.uint(body, 0, r.O), r.uint(body, r.O, r.L)))
                elif mtype != 0:
                    self.msgs.append((mtype, flags, body))

    def _read_v2(self, addr):
        r = self.f._r
        head = r.at(addr, 64)
        flags = head[5]
        p = 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        csz = 1 << (flags & 3)
        chunk0 = r.uint(head, p, csz)
        p += csz
        track = bool(flags & 0x04)
        blocks = [(addr + p, chunk0)]
        while blocks:
            a, n = blocks.pop(0)
            buf = r.at(a, n)
            q = 0
            while q + 4 <= n:
                mtype = buf[q]
                msize = struct.unpack_from("<H", buf, q + 1)[0]
                mflags = buf[q + 3]
                q += 4 + (2 if track else 0)
                body = buf[q:q + msize]
                q += msize
                if mtype == 0x10:
                    ca, cn = r.uint(body, 0, r.O), r.uint(body, r.O, r.L)
                    if r.at(ca, 4) != b"OCHK":
                        raise H5FormatError("object header continuation without OCHK")
                    blocks.append((ca + 4, cn - 8))  # signature in front, checksum behind
                elif mtype != 0:
                    self.msgs.append((mtype, mflags, body))

    def first(self, mtype):
        for t, fl, b in self.msgs:
            if t == mtype:
                return b
        return None

    # -- attributes
    def attrs(self):
        out = {}
        for t, fl, b in self.msgs:
            if t == 0x15:  # attribute info: dense storage?
                aflags = b[1]
                p = 2 + (2 if aflags & 1 else 0)
                if not self.f._r.undefined(self.f._r.uint(b, p, self.f._r.O)):
                    out["__dense_attributes_not_read__"] = True  # (fractal heap: libver='latest', > 8 attributes)
            if t != 0x0C:
                continue
            if fl & 2:
                raise H5Unsupported("shared attribute message")
            ver = b[0]
            nsz, tsz, ssz = struct.unpack_from("<HHH", b, 2)
            p = 8
            if ver == 3:
                p = 9
            pad = (lambda n: (n + 7) // 8 * 8) if ver == 1 else (lambda n: n)
            name = b[p:p + nsz].split(b"\x00")[0].decode("utf-8", "replace")
            p += pad(nsz)
            try:
                typ, _ = _parse_datatype(b, p)
            except H5Unsupported:
                continue  # an attribute of a type nobody here needs
            p += pad(tsz)
            shape = _parse_dataspace(b, p)
            p += pad(ssz)
            if shape is None:
                out[name] = None
                continue
            count = int(np.prod(shape)) if shape else 1
            out[name] = self.f._decode(typ, b[p:], count, shape)
        return out


class Dataset:
    def __init__(self, f, obj, name):
        self._f, self._obj, self.name = f, obj, name
        self.shape = _parse_dataspace(obj.first(0x01))
        self._type, _ = _parse_datatype(obj.first(0x03))
        self.dtype = self._type.dtype
        self.enum = self._type.enum
        self.attrs = obj.attrs()

    def __len__(self):
        return self.shape[0]

    def _filters(self):
        b = self._obj.first(0x0B)
        if b is None:
            return []
        ver, n = b[0], b[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = struct.unpack_from("<H", b, p)[0]
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = struct.unpack_from("<H", b, p)[0]
                p += 2
            flags, ncv = struct.unpack_from("<HH", b, p)
            p += 4
            if nlen:
                p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            cv = struct.unpack_from("<%dI" % ncv, b, p)
            p += 4 * ncv
            if ver == 1 and ncv % 2:
                p += 4
            out.append((fid, cv))
        return out

    def _unfilter(self, raw, filters, mask):
        for k in range(len(filters) - 1, -1, -1):
            if mask & (1 << k):
                continue
            fid, cv = filters[k]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = cv[0] if cv else self._type.size
                n = len(raw) // es
                body = np.frombuffer(raw, np.uint8, n * es).reshape(es, n).T.tobytes()
                raw = body + raw[n * es:]
            elif fid == 3:
                raw = raw[:-4]  # fletcher32 checksum behind the data
            else:
                raise H5Unsupported("filter %d (%s); repack the file with gzip: "
                                    "`h5repack -f GZIP=6 in out`" %
                                    (fid, {32000: "lzf", 32001: "blosc", 4: "szip"}.get(fid, "?")))
        return raw

    def __getitem__(self, key):
        """1-D datasets: d[lo:hi] reads (and inflates) only the chunks the range touches -- the
        pixel table of a genome-wide map is gigabytes, a chromosome a few per cent of it."""
        if key is Ellipsis or (isinstance(key, slice) and key == slice(None)):
            return self.read()
        if not isinstance(key, slice) or key.step not in (None, 1) or len(self.shape) != 1:
            raise H5Unsupported("only d[lo:hi] on 1-D datasets and d[...]")
        lo, hi, _ = key.indices(self.shape[0])
        return self.read(lo, max(lo, hi))

    def read(self, lo=None, hi=None):
        """The whole dataset as a numpy array (C order), or elements [lo, hi) of a 1-D one."""
        f, r = self._f, self._f._r
        if lo is not None:
            return self._read_range(lo, hi)
        shape = self.shape
        count = int(np.prod(shape)) if shape else 1
        es = self._type.size
        lay = self._obj.first(0x08)
        if lay is None:
            raise H5FormatError("dataset %s has no layout message" % self.name)
        ver = lay[0]
        if ver not in (3, 4):
            raise H5Unsupported("data layout message version %d (file written by HDF5 < 1.6.3?)" % ver)
        cls = lay[1]
        if cls == 0:  # compact
            n = struct.unpack_from("<H", lay, 2)[0]
            return f._decode(self._type, lay[4:4 + n], count, shape)
        if cls == 1:  # contiguous
            addr, n = r.uint(lay, 2, r.O), r.uint(lay, 2 + r.O, r.L)
            if r.undefined(addr):
                return f._decode(self._type, b"\x00" * (count * es), count, shape)
            return f._decode(self._type, r.at(addr, count * es), count, shape)
        if cls != 2:
            raise H5Unsupported("data layout class %d (virtual dataset)" % cls)
        filters = self._filters()
        rank = len(shape)
        out = np.zeros(count * es, np.uint8)
        out_nd = out.reshape(tuple(shape) + (es,))
        if ver == 3:
            nd = lay[2]
            btree = r.uint(lay, 3, r.O)
            cdims = struct.unpack_from("<%dI" % nd, lay, 3 + r.O)[:rank]
            chunks = [] if r.undefined(btree) else list(self._walk_chunk_btree(btree, rank))
        else:
            chunks, cdims = self._chunks_v4(lay, rank)
        cbytes = int(np.prod(cdims)) * es
        for offs, addr, nbytes, mask in chunks:
            raw = r.at(addr, nbytes)
            if filters:
                raw = self._unfilter(raw, filters, mask)
            if len(raw) < cbytes:
                raise H5FormatError("chunk of %s shorter than its dimensions" % self.name)
            block = np.frombuffer(raw, np.uint8, cbytes).reshape(tuple(cdims) + (es,))
            sl_out, sl_in = [], []
            for d in range(rank):
                lo = offs[d]
                hi = min(lo + cdims[d], shape[d])
                if hi <= lo:
                    break
                sl_out.append(slice(lo, hi))
                sl_in.append(slice(0, hi - lo))
            else:
                out_nd[tuple(sl_out)] = block[tuple(sl_in)]
        return f._decode(self._type, out.tobytes(), count, shape)

    def _chunk_index(self):
        """[(offsets, address, bytes, filter mask)] sorted by offset, and the chunk dimensions."""
        if getattr(self, "_cidx", None) is None:
            r = self._f._r
            lay = self._obj.first(0x08)
            rank = len(self.shape)
            if lay[0] == 3:
                nd = lay[2]
                btree = r.uint(lay, 3, r.O)
                cdims = struct.unpack_from("<%dI" % nd, lay, 3 + r.O)[:rank]
                chunks = [] if r.undefined(btree) else list(self._walk_chunk_btree(btree, rank))
            else:
                chunks, cdims = self._chunks_v4(lay, rank)
            chunks.sort(key=lambda c: c[0])
            self._cidx = (chunks, tuple(cdims))
        return self._cidx

    def _read_range(self, lo, hi):
        f, r = self._f, self._f._r
        es = self._type.size
        n = max(0, hi - lo)
        lay = self._obj.first(0x08)
        if lay is None or lay[0] not in (3, 4):
            raise H5Unsupported("data layout of %s" % self.name)
        if n == 0:
            return f._decode(self._type, b"", 0, (0,))
        cls = lay[1]
        if cls == 1:
            addr = r.uint(lay, 2, r.O)
            if r.undefined(addr):
                return f._decode(self._type, b"\x00" * (n * es), n, (n,))
            return f._decode(self._type, r.at(addr + lo * es, n * es), n, (n,))
        if cls != 2:
            return self.read()[lo:hi]
        chunks, cdims = self._chunk_index()
        c = cdims[0]
        filters = self._filters()
        out = np.zeros(n * es, np.uint8)
        import bisect
        starts = [ch[0][0] for ch in chunks]
        k = max(0, bisect.bisect_right(starts, lo) - 1)
        todo = []
        while k < len(chunks) and chunks[k][0][0] < hi:
            offs, addr, nbytes, mask = chunks[k]
            k += 1
            a = max(lo, offs[0])
            b = min(hi, offs[0] + c, self.shape[0])
            if b <= a:
                continue
            todo.append((a, b, offs[0], r.at(addr, nbytes), mask))  # (file reads stay on this thread)

        # The chunk pipeline in C when it is the usual one (shuffle + deflate, no chunk exempt from
        # a filter) and the type is a plain number: numpy's byte transpose of a chunk costs three
        # times its inflate and holds the interpreter lock
        ids = [fid for fid, _ in filters]
        native = _native_unfilter() if (todo and ids in ([2, 1], [1], [2]) and self._type.kind not in ("vlen", "string")
                                        and all(t[4] == 0 for t in todo)) else None
        if native is not None:
            import ctypes as C
            shuffle_es = 0
            if 2 in ids:
                cv = filters[ids.index(2)][1]
                shuffle_es = cv[0] if cv else es
            m = len(todo)
            src = (C.c_char_p * m)(*[t[3] for t in todo])
            src_len = np.array([len(t[3]) for t in todo], np.int64)
            skip = np.array([(t[0] - t[2]) * es for t in todo], np.int64)
            take = np.array([(t[1] - t[0]) * es for t in todo], np.int64)
            dst = (C.c_void_p * m)(*[out.ctypes.data + (t[0] - lo) * es for t in todo])
            rc = native(m, C.cast(src, C.c_void_p), src_len, 1 if 1 in ids else 0, shuffle_es, c * es, skip, take,
                        C.cast(dst, C.c_void_p), _pool_size())
            if rc == 0:
                return f._decode(self._type, out, n, (n,))
            if rc != -5:  # (PK_E_UNSUPPORTED = no zlib found: the Python pipeline below)
                raise H5FormatError("chunks of %s do not pass the filter pipeline (code %d)" % (self.name, rc))

        def place(item):
            a, b, o0, raw, mask = item
            if filters:
                raw = self._unfilter(raw, filters, mask)
            out[(a - lo) * es:(b - lo) * es] = np.frombuffer(raw, np.uint8, (b - o0) * es)[(a - o0) * es:]

        # a chromosome of a 10 kb map is tens of 1 M-element chunks: zlib and numpy's copies
        # release the GIL, so the chunks are inflated side by side (each writes its own range)
        if filters and len(todo) > 2:
            for _ in _inflate_pool().map(place, todo):
                pass
        else:
            for item in todo:
                place(item)
        return f._decode(self._type, out.tobytes(), n, (n,))

    def _walk_chunk_btree(self, addr, rank):
        r = self._f._r
        head = r.at(addr, 8 + 2 * r.O)
        if head[:4] != b"TREE" or head[4] != 1:
            raise H5FormatError("chunk index of %s is not a v1 B-tree" % self.name)
        level = head[5]
        used = struct.unpack_from("<H", head, 6)[0]
        ksz = 8 + 8 * (rank + 1)
        body = r.at(addr + 8 + 2 * r.O, used * (ksz + r.O) + ksz)
        for i in range(used):
            p = i * (ksz + r.O)
            nbytes, mask = struct.unpack_from("<II", body, p)
            offs = struct.unpack_from("<%dQ" % rank, body, p + 8)
            child = r.uint(body, p + ksz, r.O)
            if level == 0:
                yield offs, child, nbytes, mask
            else:
                yield from self._walk_chunk_btree(child, rank)

    def _chunks_v4(self, lay, rank):
        """Layout message version 4 (libver='latest'): single chunk, implicit, fixed array."""
        r = self._f._r
        flags, nd, enc = lay[2], lay[3], lay[4]
        cdims = [r.uint(lay, 5 + i * enc, enc) for i in range(nd)][:rank]
        p = 5 + nd * enc
        itype = lay[p]
        p += 1
        es = self._type.size
        cbytes = int(np.prod(cdims)) * es
        nchunks = [(-(-self.shape[d] // cdims[d])) for d in range(rank)]

        def offsets(k):
            o = []
            for d in range(rank - 1, -1, -1):
                o.append((k % nchunks[d]) * cdims[d])
                k //= nchunks[d]
            return tuple(reversed(o))

        if itype == 1:  # single chunk
            if flags & 2:
                size, mask = r.uint(lay, p, r.L), struct.unpack_from("<I", lay, p + r.L)[0]
                p += r.L + 4
            else:
                size, mask = cbytes, 0
            addr = r.uint(lay, p, r.O)
            return ([] if r.undefined(addr) else [((0,) * rank, addr, size, mask)]), cdims
        if itype == 2:  # implicit: chunks laid end to end, no filters
            addr = r.uint(lay, p, r.O)
            total = int(np.prod(nchunks))
            return ([] if r.undefined(addr) else
                    [(offsets(k), addr + k * cbytes, cbytes, 0) for k in range(total)]), cdims
        if itype == 3:  # fixed array
            page_bits = lay[p]
            addr = r.uint(lay, p + 1, r.O)
            if r.undefined(addr):
                return [], cdims
            hd = r.at(addr, 12 + r.L + r.O)
            if hd[:4] != b"FAHD":
                raise H5FormatError("fixed array header missing")
            client, esize = hd[5], hd[6]
            nent = r.uint(hd, 8, r.L)
            dblk = r.uint(hd, 8 + r.L, r.O)
            if r.undefined(dblk):
                return [], cdims
            def fa_elem(buf, q, k):
                a = r.uint(buf, q, r.O)
                if client == 1:  # filtered chunks: address, size (esize - O - 4 bytes), mask
                    ssz = esize - r.O - 4
                    size = r.uint(buf, q + r.O, ssz)
                    mask = struct.unpack_from("<I", buf, q + r.O + ssz)[0]
                else:
                    size, mask = cbytes, 0
                return None if r.undefined(a) else (offsets(k), a, size, mask)

            out = []
            if nent > (1 << page_bits):  # paged: bitmap of initialised pages, a checksum behind every page
                pg = 1 << page_bits
                npages = -(-nent // pg)
                bm = (npages + 7) // 8
                hdr = r.at(dblk, 6 + r.O + bm)
                if hdr[:4] != b"FADB":
                    raise H5FormatError("fixed array data block missing")
                bitmap = hdr[6 + r.O:]
                base = dblk + 6 + r.O + bm + 4
                for pi in range(npages):
                    if not (bitmap[pi >> 3] >> (7 - (pi & 7))) & 1:
                        continue
                    n = min(pg, nent - pi * pg)
                    buf = r.at(base + pi * (pg * esize + 4), n * esize)
                    out += [e for e in (fa_elem(buf, k * esize, pi * pg + k) for k in range(n)) if e]
                return out, cdims
            db = r.at(dblk, 6 + r.O + nent * esize)
            if db[:4] != b"FADB":
                raise H5FormatError("fixed array data block missing")
            out = [e for e in (fa_elem(db, 6 + r.O + k * esize, k) for k in range(nent)) if e]
            return out, cdims
        if itype == 4:  # extensible array (a dataset with one unlimited dimension)
            p += 5  # max bits, index elements, min pointers, min elements, page bits (repeated in the header)
            addr = r.uint(lay, p, r.O)
            if r.undefined(addr):
                return [], cdims
            elems = self._extensible_array(addr, cbytes)
            return [(offsets(k), a, size, mask) for k, (a, size, mask) in enumerate(elems)
                    if a is not None], cdims
        raise H5Unsupported("chunk index type %d (%s) of libver='latest'; rewrite the file with "
                            "`h5repack in out` or `cooler cp`" %
                            (itype, {5: "v2 B-tree"}.get(itype, "?")))

    def _extensible_array(self, addr, cbytes):
        """All elements of an extensible-array chunk index, in order: [(address | None, bytes, mask)]."""
        r = self._f._r
        hd = r.at(addr, 12 + 6 * r.L + r.O)
        if hd[:4] != b"EAHD":
            raise H5FormatError("extensible array header missing")
        client, esize, max_bits, idx_elmts, dblk_min, sblk_min_ptrs, page_bits = hd[5:12]
        max_idx_set = r.uint(hd, 12 + 4 * r.L, r.L)
        iblk = r.uint(hd, 12 + 6 * r.L, r.O)
        if r.undefined(iblk):
            return []
        off_size = (max_bits + 7) // 8

        def elem(buf, q):
            a = r.uint(buf, q, r.O)
            if client == 1:  # filtered chunks: address, chunk size, filter mask
                ssz = esize - r.O - 4
                size = r.uint(buf, q + r.O, ssz)
                mask = struct.unpack_from("<I", buf, q + r.O + ssz)[0]
            else:
                size, mask = cbytes, 0
            return (None if r.undefined(a) else a, size, mask)

        def log2(v):
            return v.bit_length() - 1

        # super block s: 2^(s//2) data blocks of dblk_min * 2^((s+1)//2) elements each
        nsblks = 1 + (max_bits - log2(dblk_min))
        iblock_nsblks = 2 * log2(sblk_min_ptrs)
        sb_ndblks = [1 << (s // 2) for s in range(nsblks)]
        sb_nelmts = [dblk_min << ((s + 1) // 2) for s in range(nsblks)]
        ndblk_addrs = 2 * (sblk_min_ptrs - 1)
        nsblk_addrs = nsblks - iblock_nsblks
        ib = r.at(iblk, 6 + r.O + idx_elmts * esize + (ndblk_addrs + nsblk_addrs) * r.O)
        if ib[:4] != b"EAIB":
            raise H5FormatError("extensible array index block missing")
        q = 6 + r.O
        out = [elem(ib, q + k * esize) for k in range(idx_elmts)]
        q += idx_elmts * esize
        dblk_addrs = [r.uint(ib, q + k * r.O, r.O) for k in range(ndblk_addrs)]
        q += ndblk_addrs * r.O
        sblk_addrs = [r.uint(ib, q + k * r.O, r.O) for k in range(nsblk_addrs)]

        def data_block(a, nel):
            if r.undefined(a):
                return [(None, 0, 0)] * nel
            if nel > (1 << page_bits):  # paged: a checksum behind every page
                pg = 1 << page_bits
                base = a + 6 + r.O + off_size
                res = []
                for p0 in range(0, nel, pg):
                    n = min(pg, nel - p0)
                    buf = r.at(base + (p0 // pg) * (pg * esize + 4), n * esize)
                    res += [elem(buf, k * esize) for k in range(n)]
                return res
            buf = r.at(a, 6 + r.O + off_size + nel * esize)
            if buf[:4] != b"EADB":
                raise H5FormatError("extensible array data block missing")
            return [elem(buf, 6 + r.O + off_size + k * esize) for k in range(nel)]

        k = 0
        for s in range(iblock_nsblks):  # data blocks addressed from the index block
            for _ in range(sb_ndblks[s]):
                if len(out) >= max_idx_set or k >= len(dblk_addrs):
                    return out[:max_idx_set]
                out += data_block(dblk_addrs[k], sb_nelmts[s])
                k += 1
        for j, sa in enumerate(sblk_addrs):  # the rest through super blocks
            s = iblock_nsblks + j
            if len(out) >= max_idx_set:
                break
            nd, nel = sb_ndblks[s], sb_nelmts[s]
            if r.undefined(sa):
                out += [(None, 0, 0)] * (nd * nel)
                continue
            paged = nel > (1 << page_bits)
            bitmask = 0
            if paged:
                npages = nel >> page_bits
                bitmask = nd * ((npages + 7) // 8)
            sb = r.at(sa, 6 + r.O + off_size + bitmask + nd * r.O)
            if sb[:4] != b"EASB":
                raise H5FormatError("extensible array super block missing")
            q = 6 + r.O + off_size + bitmask
            for d in range(nd):
                if len(out) >= max_idx_set:
                    break
                out += data_block(r.uint(sb, q + d * r.O, r.O), nel)
        return out[:max_idx_set]


class Group:
    def __init__(self, f, obj, name):
        self._f, self._obj, self.name = f, obj, name
        self._links = None
        self.attrs = obj.attrs()

    def _load(self):
        if self._links is not None:
            return
        f, r = self._f, self._f._r
        links = {}
        st = self._obj.first(0x11)
        if st is not None:  # symbol table: B-tree + local heap
            btree, heap = r.uint(st, 0, r.O), r.uint(st, r.O, r.O)
            hh = r.at(heap, 8 + 2 * r.L + r.O)
            if hh[:4] != b"HEAP":
                raise H5FormatError("local heap missing")
            hsize = r.uint(hh, 8, r.L)
            hdata = r.at(r.uint(hh, 8 + 2 * r.L, r.O), hsize)
            for noff, oaddr in self._walk_group_btree(btree):
                e = hdata.index(b"\x00", noff)
                links[hdata[noff:e].decode("utf-8", "replace")] = oaddr
        for t, fl, b in self._obj.msgs:
            if t == 0x02:  # link info: dense storage?
                lflags = b[1]
                p = 2 + (8 if lflags & 1 else 0)
                if not r.undefined(r.uint(b, p, r.O)):
                    raise H5Unsupported("group links in dense storage (fractal heap); rewrite the file "
                                        "with `h5repack in out` or `cooler cp`")
            if t != 0x06:
                continue
            flags = b[1]
            p = 2
            ltype = 0
            if flags & 0x08:
                ltype = b[p]
                p += 1
            if flags & 0x04:
                p += 8
            if flags & 0x10:
                p += 1
            lsz = 1 << (flags & 3)
            nlen = r.uint(b, p, lsz)
            p += lsz
            name = b[p:p + nlen].decode("utf-8", "replace")
            p += nlen
            if ltype == 0:
                links[name] = r.uint(b, p, r.O)
        self._links = links

    def _walk_group_btree(self, addr):
        r = self._f._r
        head = r.at(addr, 8 + 2 * r.O)
        if head[:4] != b"TREE" or head[4] != 0:
            raise H5FormatError("group B-tree missing")
        level = head[5]
        used = struct.unpack_from("<H", head, 6)[0]
        body = r.at(addr + 8 + 2 * r.O, used * (r.L + r.O) + r.L)
        for i in range(used):
            child = r.uint(body, r.L + i * (r.L + r.O), r.O)
            if level > 0:
                yield from self._walk_group_btree(child)
                continue
            sn = r.at(child, 8)
            if sn[:4] != b"SNOD":
                raise H5FormatError("symbol table node missing")
            n = struct.unpack_from("<H", sn, 6)[0]
            esz = 2 * r.O + 24
            ent = r.at(child + 8, n * esz)
            for k in range(n):
                yield r.uint(ent, k * esz, r.O), r.uint(ent, k * esz + r.O, r.O)

    def keys(self):
        self._load()
        return sorted(self._links)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            node._load()
            if part not in node._links:
                raise KeyError("%s: no object named %r in %s" % (self._f.path, part, node.name or "/"))
            node = node._f._open(node._links[part], (node.name.rstrip("/") + "/" + part))
        return node


class File(Group):
    def __init__(self, path):
        self.path = str(path)
        self._r = r = _Reader(path)
        self._cache = {}
        self._gheaps = {}
        try:
            root = self._superblock()
            Group.__init__(self, self, _Object(self, root), "/")
        except Exception:
            r.close()
            raise

    def close(self):
        self._r.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _superblock(self):
        r = self._r
        r.fh.seek(0, 2)
        size = r.fh.tell()
        off = 0
        while True:  # the signature sits at 0, 512, 1024, ... (user block)
            if off + 8 > size:
                raise H5FormatError("%s is not an HDF5 file (no signature)" % self.path)
            r.fh.seek(off)
            if r.fh.read(8) == SIGNATURE:
                break
            off = 512 if off == 0 else off * 2
        r.base = 0
        sb = r.at(off, 128 if off + 128 <= size else size - off)
        ver = sb[8]
        if ver in (0, 1):
            r.O, r.L = sb[13], sb[14]
            p = 24 + (4 if ver == 1 else 0)
            base = r.uint(sb, p, r.O)
            p += 4 * r.O  # base, free-space, end-of-file, driver-info addresses
            root = r.uint(sb, p + r.O, r.O)  # root symbol-table entry: name offset, header address
        elif ver in (2, 3):
            r.O, r.L = sb[9], sb[10]
            base = r.uint(sb, 12, r.O)
            root = r.uint(sb, 12 + 3 * r.O, r.O)
        else:
            raise H5Unsupported("superblock version %d" % ver)
        if r.O not in (4, 8) or r.L not in (4, 8):
            raise H5Unsupported("offsets of %d / lengths of %d bytes" % (r.O, r.L))
        r.base = base if base else 0
        if off and not base:
            r.base = off  # addresses are relative to the superblock when a user block precedes it
        return root

    def _open(self, addr, name):
        if addr in self._cache:
            return self._cache[addr]
        obj = _Object(self, addr)
        node = Dataset(self, obj, name) if obj.first(0x08) is not None or obj.first(0x03) is not None \
            else Group(self, obj, name)
        self._cache[addr] = node
        return node

    # -- values
    def _global_heap_object(self, addr, index):
        r = self._r
        if addr not in self._gheaps:
            hd = r.at(addr, 8 + r.L)
            if hd[:4] != b"GCOL":
                raise H5FormatError("global heap collection missing")
            size = r.uint(hd, 8, r.L)
            buf = r.at(addr, size)
            objs = {}
            p = 8 + r.L
            while p + 8 + r.L <= size:
                idx = struct.unpack_from("<H", buf, p)[0]
                osz = r.uint(buf, p + 8, r.L)
                if idx == 0:
                    break
                objs[idx] = buf[p + 8 + r.L:p + 8 + r.L + osz]
                p += 8 + r.L + (osz + 7) // 8 * 8
            self._gheaps[addr] = objs
        return self._gheaps[addr][index]

    def _decode(self, typ, buf, count, shape):
        r = self._r
        if typ.kind == "vlen":
            if not typ.vlen_string:
                raise H5Unsupported("variable-length sequence")
            vals = []
            step = 4 + r.O + 4
            for k in range(count):
                n = struct.unpack_from("<I", buf, k * step)[0]
                addr = r.uint(buf, k * step + 4, r.O)
                idx = struct.unpack_from("<I", buf, k * step + 4 + r.O)[0]
                vals.append("" if (n == 0 or addr == 0) else
                            self._global_heap_object(addr, idx)[:n].decode("utf-8", "replace"))
            return vals[0] if not shape else np.array(vals, dtype=object).reshape(shape)
        if isinstance(buf, np.ndarray):  # (a buffer of the reader's own: no copy)
            arr = buf.view(typ.dtype)[:count]
        else:
            arr = np.frombuffer(buf, typ.dtype, count).copy()
        if typ.kind == "string":
            if not shape:
                return arr[0].split(b"\x00")[0].decode("utf-8", "replace")
            return arr.reshape(shape)
        if not shape:
            v = arr[0]
            return v.item() if typ.dtype.byteorder in "=|<" or True else v
        return arr.reshape(shape)
