import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return _have_gpu()


# MMT_PACKED_TEXT=1 runs every collection through the two-bit text (textref.hpp), which only the bucket-wise producer reads:
# assertions about which producer ran accept "guided" then
PACKED_TEXT = os.environ.get("MMT_PACKED_TEXT", "0") not in ("", "0")


def producer_is(engine, kind):
    return engine.producer_used() == ("guided" if PACKED_TEXT else kind)
