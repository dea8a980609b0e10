"""GPU time of the backward segment (events around loss.backward()) and of the whole step, eager vs --graph_train 1."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_2d as T
gt = os.environ.get("GT", "0")
args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--synthetic", "1", "--graphs", "1",
                                    "--graph_train", gt, "--k2", "0"])
st = T.ArcoStep2D(args, "cuda:0")
l, ll = T.synthetic_batch(8, args.patch_size, 4, 1, "cuda:0")
u, _ = T.synthetic_batch(8, args.patch_size, 4, 2, "cuda:0")
evs = []
orig = torch.Tensor.backward
def bw(self, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(self, *a, **k); e1.record(); evs.append((e0, e1)); return r
torch.Tensor.backward = bw
from arco_amd import ops
batches = [(l, ll, u)]
if os.environ.get("CYCLE"):
    batches = [(T.synthetic_batch(8, args.patch_size, 4, 100 + 2 * i, "cuda:0") + (T.synthetic_batch(8, args.patch_size, 4, 101 + 2 * i, "cuda:0")[0],)) for i in range(8)]
for i in range(8): st.step(*batches[i % len(batches)])
torch.cuda.synchronize(); evs.clear()
if os.environ.get("PROF"):
    ops.PROFILE, ops.PROFILE_EVERY = {}, 7
s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s0.record(); t0 = time.perf_counter()
n = 40
for i in range(n): st.step(*batches[i % len(batches)])
s1.record(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / n * 1e3
bwd = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
print(f"graph_train={gt} cycle={os.environ.get('CYCLE')} prof={os.environ.get('PROF')}: wall {wall:.2f} ms/step, gpu span {s0.elapsed_time(s1)/n:.2f} ms/step, backward segment on GPU {bwd:.2f} ms")
