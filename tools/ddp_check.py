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
    print("DDP_OK banks/ptr/params/teacher identical on 2 ranks; bank lens", mine[4], "loss", float(reco))
