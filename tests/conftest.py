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
    """The product library.  GPU tests must run on the HIP path: no fallback, fail loudly."""
    from taichislam_amd import _lib
    L = _lib.lib()
    assert _lib.device_count() > 0, "no HIP device visible: GPU tests cannot run"
    return L
