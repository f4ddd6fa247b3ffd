"""CPU-side checks of the drop-in boundary: the library loads and exports
every symbol include/peakachu_hip.h declares; no compute call is made."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "peakachu_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pk_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from peakachu_amd import _lib
    L = _lib.load()
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libpeakachu_hip.so lacks %s" % n
    assert sorted(_lib.SIGNATURES) == names
    assert L.pk_abi_version() == 1


def test_locks_are_per_device():
    """include/peakachu_hip.h 'threading': calls on one device are serialised, calls on different
    devices do not wait for each other (the probe takes the locks themselves; no device needed)."""
    from peakachu_amd import _lib
    L = _lib.load()
    assert L.pk_debug_lock_probe(0, 0) == 0
    assert L.pk_debug_lock_probe(3, 3) == 0
    assert L.pk_debug_lock_probe(0, 1) == 1
    assert L.pk_debug_lock_probe(7, 0) == 1
    assert L.pk_debug_lock_probe(0, 64) == _lib.PK_E_INVALID
    assert L.pk_debug_lock_probe(-1, 0) == _lib.PK_E_INVALID
    # the process-wide pieces have their own locks: defaults can be set and read from threads
    import threading
    errs = []

    def flip(k):
        try:
            for i in range(200):
                L.pk_set_option(b"chunk", 1 << (16 + (i + k) % 4))
                assert L.pk_get_option(b"chunk") in (1 << 16, 1 << 17, 1 << 18, 1 << 19)
        except Exception as e:  # pragma: no cover
            errs.append(e)
    old = L.pk_get_option(b"chunk")
    ts = [threading.Thread(target=flip, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    L.pk_set_option(b"chunk", old)
    assert not errs and L.pk_get_option(b"chunk") == old


def test_no_device_fails_loudly():
    from peakachu_amd import _lib
    L = _lib.load()
    if L.pk_device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(_lib.PeakachuHipError):
        _lib.HipCands(np.zeros(4, np.int32), np.zeros(4, np.int32))
    with pytest.raises(_lib.PeakachuHipError):
        _lib.require_device()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "peakachu_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f


def test_host_chunk_pipeline_inflates_and_unshuffles():
    """pk_host_unfilter_chunks (include/peakachu_hip.h; what h5lite hands a ranged read's chunks to):
    inflate + byte un-shuffle + the wanted slice of every chunk, for element sizes 1 .. 8, whole and
    partial slices, one and several threads -- against zlib + numpy; a chunk that does not inflate to
    its size and a slice outside the chunk are refused.  No device involved."""
    import ctypes as C
    import zlib
    from peakachu_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(9)
    for es in (8, 4, 2, 1):
        n_el, m = 1000, 7
        chunks = [rng.integers(0, 256, n_el * es, dtype=np.uint8).tobytes() for _ in range(m)]
        stored = []
        for c in chunks:  # the write pipeline: shuffle, then deflate
            sh = np.frombuffer(c, np.uint8).reshape(n_el, es).T.tobytes() if es > 1 else c
            stored.append(zlib.compress(sh, 1))
        skip = np.array([0, es * 3, 0, es * 999, es * 500, 0, es * 10], np.int64)
        take = np.array([n_el * es, es * 5, 0, es, es * 500, es * 1000, es * 17], np.int64)
        out = [np.zeros(max(int(t), 1), np.uint8) for t in take]
        src = (C.c_char_p * m)(*stored)
        dst = (C.c_void_p * m)(*[o.ctypes.data for o in out])
        lens = np.array([len(s) for s in stored], np.int64)
        for threads in (1, 4):
            for o in out:
                o[:] = 0
            rc = L.pk_host_unfilter_chunks(m, C.cast(src, C.c_void_p), lens, 1, es, n_el * es, skip, take,
                                           C.cast(dst, C.c_void_p), threads)
            assert rc == 0, _lib.last_error()
            for c, o, s0, t in zip(chunks, out, skip, take):
                assert o[:int(t)].tobytes() == c[int(s0):int(s0 + t)]
        # a chunk that does not inflate to chunk_bytes
        bad = (C.c_char_p * 1)(zlib.compress(b"short", 1))
        one = np.zeros(8, np.uint8)
        rc = L.pk_host_unfilter_chunks(1, C.cast(bad, C.c_void_p), np.array([len(bad[0])], np.int64), 1, es, n_el * es,
                                       np.zeros(1, np.int64), np.array([es], np.int64),
                                       C.cast((C.c_void_p * 1)(one.ctypes.data), C.c_void_p), 1)
        assert rc == _lib.PK_E_INVALID and "does not inflate" in _lib.last_error()
        # a slice beyond the chunk
        rc = L.pk_host_unfilter_chunks(1, C.cast(src, C.c_void_p), lens, 1, es, n_el * es, np.array([es * 999], np.int64),
                                       np.array([es * 2], np.int64), C.cast(dst, C.c_void_p), 1)
        assert rc == _lib.PK_E_INVALID


def test_candidate_lists_are_classified_by_shape():
    """pk_debug_classify_coords (no device): what decides the extractor's route when a list is made -- bit 1:
    batches of 32 consecutive candidates are runs on one diagonal (the strip of the band is staged in LDS),
    bit 0: consecutive candidates are rarely neighbours (loads in the order of the window's diagonals).
    Shapes only: get_candidate's order is diagonal by diagonal, row ascending (peakachu/scoreUtils.py:46-68)."""
    from peakachu_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(5)
    n = 4000
    xs, ys = [], []
    for d in range(6, 60):                      # every pixel of the band, diagonal by diagonal
        x = np.arange(0, n - d, dtype=np.int32)
        xs.append(x); ys.append(x + d)
    x, y = np.concatenate(xs), np.concatenate(ys)

    def kind(px, py):
        return L.pk_debug_classify_coords(px.size, np.ascontiguousarray(px, np.int32), np.ascontiguousarray(py, np.int32))

    assert kind(x, y) == 2
    keep = rng.random(x.size) < 0.93            # seven per cent of the pixels empty: still runs
    assert kind(x[keep], y[keep]) == 2
    keep = rng.random(x.size) < 0.5             # half empty: a batch spans ~64 rows -- neither
    assert kind(x[keep], y[keep]) == 0
    assert kind(x[::4], y[::4]) == 1            # strided: scattered, no runs
    keep = rng.random(x.size) < 0.02            # the Poisson-thinned kind
    assert kind(x[keep], y[keep]) == 1
    p = rng.permutation(x.size)
    assert kind(x[p], y[p]) == 1
    assert kind(x[:20], y[:20]) == 2            # a short run
    assert kind(x[:0], y[:0]) == 0
