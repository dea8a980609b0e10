"""Run by tests/test_dist_gpu.py under torchrun (2 ranks sharing one GPU over gloo): the 3-D step with f16 activation storage and a
FORCED overflow of the f16 backward on ONE rank.  The loss-scale guard must (i) leave the heads' gradient bucket
alone - it is inside an asynchronous all-reduce started from the backward -, (ii) make every rank take the same decision (the flag is
all-reduced with MIN), so that after the overflowed step and at the end of the run student, heads, teacher, banks and loss scale are
bit-identical across ranks, and (iii) halve the loss scale on BOTH ranks at the next step."""
import os, sys, hashlib
os.environ.setdefault("OMP_NUM_THREADS", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import random, numpy as np, torch
import torch.distributed as td
from arco_amd import dist as adist, ops, train_arco_3d as T3

rank, world = adist.init()
assert world == 2 and adist.is_dist()
dev = torch.device("cuda", adist.local_rank())
torch.cuda.set_device(dev)
random.seed(7); np.random.seed(7); torch.manual_seed(7)
patch = (32, 32, 32)
args = T3.build_parser().parse_args(["--batch_size", "1", "--queue_size", "200", "--synthetic", "1", "--num_classes", "2", "--k1", "1.0",
                                     "--act_dtype", "f16", "--num_queries", "32", "--num_negatives", "16"])
args.patch_size = list(patch)
st = T3.ArcoStep3D(args, dev)
assert ops.ACT_HALF and st.heads_start > 0
scale0 = ops.LOSS_SCALE
N_STEPS, OVF_STEP = 6, 3
real_guard = st._unscale_and_guard
state = {"i": -1, "heads_before": None, "heads_after": None}


def guard():
    g = st.optimizer.flat_g
    if state["i"] == OVF_STEP:
        if rank == 1:
            g[7] = float("inf")                      # an overflowed V-Net gradient on this rank only
        state["heads_before"] = g[st.heads_start:].clone()       # (snapshot: the bucket may still be reducing - read only)
    ok = real_guard()
    if state["i"] == OVF_STEP:
        state["vnet_after"] = g[:st.heads_start].clone()
    return ok


st._unscale_and_guard = guard
flags = []
for i in range(N_STEPS):
    state["i"] = i
    l, ll = T3.synthetic_volume_batch(1, patch, 2, 100 + 10 * i + rank, dev)      # different data per rank
    u, _ = T3.synthetic_volume_batch(1, patch, 2, 200 + 10 * i + rank, dev)
    st.step(l, ll, u)
    torch.cuda.synchronize()
    flags.append(bool(st._ovf_host[0]) if st._ovf_host is not None else True)
    if i == OVF_STEP:
        # both ranks zeroed the V-Net's gradient of this step (the rank that did NOT overflow too)
        assert float(state["vnet_after"].abs().max()) == 0.0, "the V-Net gradient of the overflowed step was applied on this rank"
    if i == OVF_STEP + 1:
        assert ops.LOSS_SCALE == scale0 / 2, (ops.LOSS_SCALE, scale0)            # consumed at the start of the next step, on both ranks
assert flags[OVF_STEP] is False and all(f for k, f in enumerate(flags) if k != OVF_STEP), flags
assert st.overflow_steps == 1


def digest(ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().float().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


mine = [digest(st.optimizer.params), digest(list(st.ema_model.parameters())), digest([b[0] for b in st.memobank]),
        digest(list(st.q_representation.parameters()) + list(st.q_feature_extractor.parameters())), str(ops.LOSS_SCALE), str(flags)]
both = [None, None]
td.all_gather_object(both, mine)
assert both[0] == both[1], (both[0], both[1])
td.barrier()
if rank == 0:
    print(f"DDP3D_OK {N_STEPS} f16 steps, overflow forced on rank 1 at step {OVF_STEP}: both ranks skipped the V-Net update, heads applied, "
          f"loss scale {scale0:g} -> {ops.LOSS_SCALE:g} on both, student / heads / teacher / banks identical across ranks")
td.destroy_process_group()
