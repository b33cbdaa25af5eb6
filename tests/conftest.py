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
    """torch with a usable HIP device (GPU tests only).  Fails at once — never skips, never waits — without one: the
    product has no CPU fallback."""
    if crt_built.crt_device_count() < 1:
        pytest.fail("no HIP device visible: the product has no CPU fallback")
    import torch
    if not torch.cuda.is_available():
        pytest.fail("torch sees no device although libcrt_hip.so does: two HIP runtimes in this process — something loaded "
                    "libamdhip64.so.7 before torch did (import torch before libcrt_hip.so or any other ROCm library; "
                    "cudaraytracer_amd/_lib.py, INTEGRATION.md \"One HIP runtime per process\")")
    return torch


@pytest.fixture(scope="session")
def c2_oracle_frame(oracle):
    """BASELINE configs[1] (800x600, 16 spp, the 32-triangle Cornell box) through the CPU oracle, once per session: the frame
    and the RNG states after it (d, v0..v4 per pixel).  Ten seconds of CPU that seven GPU tests used to spend each."""
    import numpy as np
    from cudaraytracer_amd import scenes
    sc = scenes.cornell32()
    nx, ny, spp = 800, 600, 16
    c = sc["camera"]
    w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    ocam = oracle.make_camera(c["lookfrom"], c["lookat"], (0, 1, 0), c["vfov"], np.float32(nx) / np.float32(ny), c["aperture"], c["focus"])
    ref, states = oracle.render(w, ocam, nx, ny, spp)
    return sc, nx, ny, spp, ref, np.concatenate([states["d"][:, None], states["v"]], axis=1)
