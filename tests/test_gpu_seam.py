"""GPU: the C-ABI seam exchange (include/td_seam.h, libtd_seam.so) executed on hardware.  One GPU is what the boxes have, so the RCCL
communicator has world 1 and every message goes to the rank itself (tests/_seam_worker.py says what that does and does not show); the
two-GPU form of the same path rides in tests/_nccl_exchange_worker.py and runs whenever >= 2 GPUs are visible."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_seam_exchange_on_one_gpu_through_the_c_abi():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    # a child process with a deadline: an RCCL call that never returns is killed by PID, not waited for
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_seam_worker.py")], env=env, capture_output=True, text=True, timeout=420)
    tail = out.stdout[-3000:] + out.stderr[-3000:]
    assert out.returncode == 0, tail
    for stage in "ABCDE":
        assert f"SEAM_{stage}_OK" in out.stdout, tail


def test_capi_seam_exchange_two_gpus():
    """two ranks on two GPUs: exchange_windows and the sharded sampler with seam_comm= (td_seam_exchange_windows over xGMI) against the fabricated
    windows / the single-GPU canvas, bit for bit.  Needs >= 2 GPUs (the torch.distributed transport's twin is test_nccl_seam_exchange_two_gpus)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29733",
                          os.path.join(ROOT, "tests", "_nccl_exchange_worker.py"), "capi"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "CAPI_EXCHANGE_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
