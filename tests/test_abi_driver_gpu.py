"""The C ABI without Python: tools/abi_driver.cpp (hipMalloc'd buffers, no torch in the process) checks the reference's IoU / NMS known
answers, a convolution against a host loop and a world-1 communicator through bd_comm_*, and times a head convolution."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_driver_passes():
    exe = os.path.join(ROOT, "basedet_amd", "lib", "abi_driver")
    if not os.path.exists(exe):
        from basedet_amd import build
        exe = build.build_driver()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "all checks passed" in r.stdout and "bd_comm (world 1) ok" in r.stdout
    print(r.stdout)
