import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The product C-ABI library with a live device; GPU tests only."""
    from peakachu_amd import _lib
    L = _lib.load()
    if L.pk_device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box")
    return L
