import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; oracle/crt_oracle.c)."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def crt_built():
    """The product library, built in-tree if it is missing (hipcc cross-compiles on CPU)."""
    from cudaraytracer_amd import _lib
    if not os.path.exists(_lib.library_path()):
        _lib.build_library()
    return _lib.lib()


@pytest.fixture(scope="session")
def gpu(crt_built):
    """torch with a usable HIP device (GPU tests only).  Fails — never skips — without one: the
    product has no CPU fallback.  A fresh box can take a moment to expose the device to torch."""
    if crt_built.crt_device_count() < 1:
        pytest.fail("no HIP device visible: the product has no CPU fallback")
    import time
    import torch
    for _ in range(10):
        if torch.cuda.is_available():
            break
        time.sleep(2.0)
    assert torch.cuda.is_available()
    return torch
