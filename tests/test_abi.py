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
