"""GPU test of the optional RCCL binding (include/c2ray_rccl.h, libc2ray_rccl.so): communicator set-up
from a unique id, the all-reduce callback on a device array (one rank: identity), state errors.  Runs in a
child process without torch (see tests/_rccl_worker.py).  Real multi-rank RCCL needs several GPUs; the
rank arithmetic around it is covered by the gloo tests."""
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_binding_single_rank():
    if not os.path.exists(os.path.join(ROOT, "c2-ray3dm_amd", "libc2ray_rccl.so")):
        pytest.skip("libc2ray_rccl.so not built")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_SOCKET_IFNAME=os.environ.get("NCCL_SOCKET_IFNAME", "lo"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_worker.py")], env=env, timeout=300,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    assert "RCCL_OK" in out, out[-2000:]
