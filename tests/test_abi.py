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
