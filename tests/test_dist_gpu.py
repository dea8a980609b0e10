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


def test_two_ranks_f16_volume_step_with_an_overflow_on_one_rank():
    """the f16 loss-scale guard under data parallelism - tools/ddp_check3d.py: six 3-D steps with --act_dtype f16 on
    two ranks, an inf forced into ONE rank's V-Net gradient at step 3.  The guard writes only flat_g[:heads_start] (the heads' bucket is
    inside an asynchronous all-reduce), the flag is all-reduced with MIN: both ranks skip that V-Net update, both halve the loss scale,
    weights / teacher / heads / banks stay bit-identical across ranks."""
    env = dict(os.environ, ARCO_DIST_BACKEND="gloo", ARCO_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29537", os.path.join(ROOT, "tools", "ddp_check3d.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "DDP3D_OK" in out.stdout


def test_forced_one_rank_nccl_group_equals_the_plain_step_and_stays_reproducible():
    """ARCO_FORCE_DIST=1: a one-rank `nccl` group, every exchange of arco_amd/dist.py issued through RCCL /
    ProcessGroupNCCL inside the default two-stream, graph-replayed step at the headline size.  The run must (i) issue the step's
    collectives (two gradient buckets + counters + prototypes + percentile sums every step), (ii) give the loss terms and final weights
    of the plain single-process run (a one-rank sum is the identity; the rank mean divides by 1), (iii) reproduce one step's gradient
    over 30 executions to 1e-5 with RCCL's stream present."""
    import json
    res = {}
    for force in ("0", "1"):
        env = dict(os.environ, ARCO_FORCE_DIST=force, HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK"):
            env.pop(k, None)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "force_dist_check.py"), "30"], env=env, capture_output=True,
                             text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        res[force] = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("FORCE_DIST ")][-1][len("FORCE_DIST "):])
    plain, forced = res["0"], res["1"]
    assert forced["dist"] and not plain["dist"]
    c = forced["collectives_per_step"]
    assert c["all_reduce"] >= 4 and c["all_gather"] >= 1, c              # buckets (2) + prototypes + percentile sums; counter table
    # The exchanged quantities pass through other arithmetic (count-weighted prototype sums divided by the counts again, the phased
    # percentile selection): equal to fp32 rounding in the first step, and six chained steps of training amplify a 1e-7 difference
    # like any other (measured 5e-5 on the contrastive term at step 5) - first step strict, trajectory at north_star's 1e-3
    for it, (a, b) in enumerate(zip(plain["terms"], forced["terms"])):
        for k in a:
            # (measured: step 0 bit-identical; steps 1-5 drift from 2e-5 to 3e-4 on the contrastive term; the equivariance term
            #  depends on the TPS warp, i.e. on the CPU generator's position behind the samplers - a flipped decision moves it)
            tol = 2e-6 if it == 0 else (0.25 if k == "eqv" else 1e-3)
            assert abs(a[k] - b[k]) <= tol * max(1.0, abs(a[k])) or (k == "eqv" and abs(a[k] - b[k]) <= 0.25 * abs(a[k])), (it, k, a[k], b[k])
    assert abs(plain["checksum"] - forced["checksum"]) <= 1e-5 * plain["checksum"]
    assert forced["worst_repro"] <= 1e-5 and plain["worst_repro"] <= 1e-5, (forced["worst_repro"], plain["worst_repro"])
    print("ms per step plain / forced one-rank nccl:", plain["ms_per_step"], forced["ms_per_step"], "collectives per step:", c)
