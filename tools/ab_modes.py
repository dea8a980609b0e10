"""Same-process A/B of train_arco_2d.TEACHER_SIDE modes: one trainer per mode (own graphs), alternating blocks of steps on one box.
python tools/ab_modes.py 3 4 -4 [--city]      (-4: mode 4 without train_arco_2d.SIDE_SYNC)"""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_2d as T
modes = [int(x) for x in sys.argv[1:] if x.lstrip("-").isdigit()] or [3, 4]
city = "--city" in sys.argv          # the Cityscapes-shaped shard (BASELINE.json configs[3]): 2 images 3x512x1024, 19 classes
reps, block = (6, 20) if city else (8, 100)
sts = {}
for m in modes:
    T.TEACHER_SIDE = abs(m)
    if city:
        args = T.build_parser().parse_args(["--batch_size", "1", "--queue_size", "4096", "--synthetic", "1", "--num_classes", "19", "--in_chns", "3"])
        args.patch_size = [512, 1024]
    else:
        args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--func", "smc", "--synthetic", "1"])
    sts[m] = T.ArcoStep2D(args, "cuda:0")
if city:
    bs = [(T.synthetic_batch(1, args.patch_size, 19, 1 + 2 * i, "cuda:0", in_chns=3), T.synthetic_batch(1, args.patch_size, 19, 2 + 2 * i, "cuda:0", in_chns=3)[0]) for i in range(4)]
else:
    bs = [(T.synthetic_batch(8, args.patch_size, 4, 100 + 2 * i, "cuda:0"), T.synthetic_batch(8, args.patch_size, 4, 101 + 2 * i, "cuda:0")[0]) for i in range(4)]
def run(m, n):
    key = m
    T.SIDE_SYNC = 0 if m < 0 else 1           # -m: mode m without the host-side wait in front of backward()
    m = abs(m)
    T.TEACHER_SIDE = m
    st = sts[key]
    for i in range(n):
        (l, ll), u = bs[i % 4]
        st.step(l, ll, u, 0, 100)
for m in modes:
    run(m, 12 if city else 60)
res = {m: [] for m in modes}
for r in range(reps):
    for m in (modes if r % 2 == 0 else modes[::-1]):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(m, block); torch.cuda.synchronize()
        res[m].append((time.perf_counter() - t0) / block * 1e3)
for m in modes:
    v = res[m]
    print(f"mode {m}: mean {sum(v) / len(v):.3f}  min {min(v):.3f}  max {max(v):.3f}   " + " ".join(f"{x:.2f}" for x in v))
