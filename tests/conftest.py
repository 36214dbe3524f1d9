import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle is test infrastructure: build it on demand (gcc, < 2 s)
    so = os.path.join(ROOT, "oracle", "liblqr_oracle.so")
    src = os.path.join(ROOT, "oracle", "lqr_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    import lqr_ctypes
    return lqr_ctypes.oracle_api()


@pytest.fixture(scope="session")
def engine():
    """The HIP engine through its C ABI.  Fails loudly (no fallback) if the library
    or the GPU is missing."""
    import lqr_ctypes
    api = lqr_ctypes.engine_api()
    return api
