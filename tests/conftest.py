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
