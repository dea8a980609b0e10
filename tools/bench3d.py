"""Timing of the 3-D hot-path step (BASELINE.json configs[2]: LA V-Net 112x112x80, 4 volumes/step).
`python tools/bench3d.py B [X Y Z]`: B volumes per stream; X Y Z = patch size (configs[4] shape: 1 160 160 96, fp32)."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, random, numpy as np
from arco_amd import train_arco_3d as T3
random.seed(1337); np.random.seed(1337); torch.manual_seed(1337)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 2
eqv = int(os.environ.get("EQV_PASS", "0"))     # 1: with the reference's equivariance block (one more student forward per step)
args = T3.build_parser().parse_args(["--batch_size", str(b), "--queue_size", "4096", "--synthetic", "1", "--num_classes", "2",
                                     "--eqv_pass", str(eqv), "--conv_mma", os.environ.get("CONV_MMA", "f32"), "--graph_train", os.environ.get("GRAPH_TRAIN", "0"),
                                     "--act_dtype", os.environ.get("ACT_DTYPE", "f32")])
if len(sys.argv) > 4:
    args.patch_size = [int(v) for v in sys.argv[2:5]]
st = T3.ArcoStep3D(args, "cuda:0")
l, ll = T3.synthetic_volume_batch(b, args.patch_size, 2, 1, "cuda:0")
u, _ = T3.synthetic_volume_batch(b, args.patch_size, 2, 2, "cuda:0")
for _ in range(4): st.step(l, ll, u)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 8
for _ in range(n): loss, reco = st.step(l, ll, u)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"3D step b={b} patch={args.patch_size} eqv_pass={eqv} conv_mma={args.conv_mma} act_dtype={args.act_dtype}: {dt*1e3:.1f} ms/step  {1/dt:.2f} steps/s  loss {float(reco):.4f}  mem {torch.cuda.max_memory_allocated()/1e9:.1f} GB")
