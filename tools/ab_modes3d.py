"""Same-process A/B of train_arco_3d.PASS_SIDE modes (see tools/ab_modes.py).  python tools/ab_modes3d.py 2 3 [lits]"""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_3d as T3
modes = [int(x) for x in sys.argv[1:] if x.isdigit()] or [2, 3]
lits = "lits" in sys.argv
b = 1 if lits else 2
sts = {}
for m in modes:
    T3.PASS_SIDE = m
    args = T3.build_parser().parse_args(["--batch_size", str(b), "--queue_size", "4096", "--synthetic", "1", "--num_classes", "2",
                                         "--conv_mma", "f32x3", "--act_dtype", "f16" if lits else "f32"])
    args.patch_size = [160, 160, 96] if lits else [112, 112, 80]
    sts[m] = T3.ArcoStep3D(args, "cuda:0")
l, ll = T3.synthetic_volume_batch(b, args.patch_size, 2, 1, "cuda:0")
u, _ = T3.synthetic_volume_batch(b, args.patch_size, 2, 2, "cuda:0")
def run(m, n):
    T3.PASS_SIDE = m
    for i in range(n):
        sts[m].step(l, ll, u)
for m in modes:
    run(m, 12)
res = {m: [] for m in modes}
for r in range(6):
    for m in (modes if r % 2 == 0 else modes[::-1]):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(m, 20); torch.cuda.synchronize()
        res[m].append((time.perf_counter() - t0) / 20 * 1e3)
for m in modes:
    v = res[m]
    print(f"PASS_SIDE {m}: mean {sum(v) / len(v):.3f}  min {min(v):.3f}  max {max(v):.3f}   " + " ".join(f"{x:.2f}" for x in v))
