"""Two ranks on the ONE GPU of the test box (gloo transport, ARCO_FORCE_DEVICE=0): the data-parallel step
keeps banks, pointers, student and teacher parameters bit-identical across ranks (-m gpu)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_stay_identical():
    env = dict(os.environ, ARCO_DIST_BACKEND="gloo", ARCO_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "tools", "ddp_check.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "DDP_OK" in out.stdout
