"""Run by tests/test_dist_gpu.py under torchrun (2 ranks sharing one GPU over gloo): after a few data-parallel
steps every rank must hold bit-identical banks, queue pointers, student parameters and teacher parameters."""
import os, sys, hashlib
os.environ.setdefault("OMP_NUM_THREADS", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import random, numpy as np, torch
import torch.distributed as td
from arco_amd import dist as adist, train_arco_2d as T

rank, world = adist.init()
assert world == 2 and adist.is_dist()
dev = torch.device("cuda", adist.local_rank())
torch.cuda.set_device(dev)
# entropy percentiles of the GLOBAL batch: the phased radix select with summed histograms must give
# the masks that np.percentile over both ranks' valid entropies gives
from arco_amd import glue, _lib as L_
g = torch.Generator().manual_seed(100 + rank)
pred = (3 * torch.randn(3, 4, 24, 40, generator=g)).to(dev)
lab_l = torch.randint(0, 4, (3, 24, 40), generator=g).to(dev)
lab_u = torch.randint(-1, 4, (3, 24, 40), generator=g).to(dev)
assert glue.state_reduce_hook is adist.allreduce_sum
low, high = glue.entropy_masks(pred, lab_l, lab_u, 20.0)
r_, ld_, b_, C_n, P_ = glue._geom(pred)
ent = torch.empty(b_ * P_, dtype=torch.float32, device=dev)
L_.call("arco_softmax_rows", L_.ptr(r_), ld_, b_ * P_, C_n, P_, None, None, None, L_.ptr(ent))
ents, labs = [torch.empty_like(ent) for _ in range(world)], [torch.empty_like(lab_u) for _ in range(world)]
td.all_gather(ents, ent); td.all_gather(labs, lab_u.contiguous())
vals = torch.cat([e[l.reshape(-1) >= 0] for e, l in zip(ents, labs)]).cpu().numpy()
t_lo, t_hi = np.float32(np.percentile(vals, 20.0)), np.float32(np.percentile(vals, 80.0))
valid = (lab_u.reshape(-1) >= 0)
exp_low = ((ent <= float(t_lo)) & valid).float().view(3, 1, 24, 40)
exp_high = ((ent >= float(t_hi)) & valid).float().view(3, 1, 24, 40)
assert torch.equal(low[3:], exp_low) and torch.equal(high[3:], exp_high), "global entropy thresholds differ"
assert torch.equal(low[:3, 0], (lab_l >= 0).float())
local_lo = np.float32(np.percentile(ent[valid].cpu().numpy(), 20.0))
assert local_lo != t_lo                                           # the per-rank threshold would have been different

random.seed(7); np.random.seed(7); torch.manual_seed(7)          # same sampler sequence on every rank
args = T.build_parser().parse_args(["--batch_size", "2", "--queue_size", "200", "--synthetic", "1", "--num_queries", "64",
                                    "--num_negatives", "32", "--k1", "1.0", "--base_lr", "0.05"])
args.patch_size = [64, 64]
st = T.ArcoStep2D(args, dev)
for i in range(4):
    l_img, l_lab = T.synthetic_batch(2, args.patch_size, 4, 100 + 10 * i + rank, dev)      # different data per rank
    u_img, _ = T.synthetic_batch(2, args.patch_size, 4, 200 + 10 * i + rank, dev)
    loss, reco = st.step(l_img, l_lab, u_img)
torch.cuda.synchronize()


def digest(ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


mine = [digest([b[0] for b in st.memobank]), digest([p for p in st.queue_ptrlis]), digest(st.optimizer.params),
        digest(list(st.ema_model.parameters())), str([int(b[0].shape[0]) for b in st.memobank])]
both = [None, None]
td.all_gather_object(both, mine)
assert both[0] == both[1], (both[0], both[1])
assert sum(int(b[0].shape[0]) for b in st.memobank) > 4, "banks never grew"
if rank == 0:
    print("DDP_OK global entropy thresholds; banks/ptr/params/teacher identical on 2 ranks; bank lens", mine[4], "loss", float(reco))
