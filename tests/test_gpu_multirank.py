"""Multi-process rehearsal of the multi-GPU path on ONE GPU: W processes (torch.distributed.run,
gloo, 127.0.0.1) each render their rows with the production BandRenderer on the shared GPU;
the gathered frame must equal the golden fixture bit for bit.  (RCCL itself is only exercised
by the driver's multi-GPU run.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,mode", [(2, "strips"), (3, "strips"), (2, "bands")])
def test_multirank_frame_equals_golden(world, mode, tmp_path, crt_built):
    if crt_built.crt_device_count() < 1:
        pytest.fail("no HIP device visible")
    out = str(tmp_path / "result.txt")
    port = 29600 + world * 7 + (0 if mode == "strips" else 3)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_multirank_worker.py"), out, mode]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(out).read() == "OK"
