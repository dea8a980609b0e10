"""Shapes, launch counts and HIP-event times of every conv / GEMM launch of the default 2-D step (eager pass, every launch
timed): `python tools/gemm_shapes.py [taps]`."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_2d as T, ops, graphs

args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--func", "smc", "--synthetic", "1"])
st = T.ArcoStep2D(args, "cuda:0")
bs = [(T.synthetic_batch(8, args.patch_size, 4, 100 + 2 * i, "cuda:0"), T.synthetic_batch(8, args.patch_size, 4, 101 + 2 * i, "cuda:0")[0]) for i in range(4)]
def run(n):
    for i in range(n):
        (l, ll), u = bs[i % 4]
        st.step(l, ll, u, 0, 100)
run(16)
graphs.set_enabled(st, False)
run(2)
torch.cuda.synchronize()
ops.PROFILE, ops.PROFILE_EVERY = {}, 1
N = 3
run(N)
torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
want = int(sys.argv[1]) if len(sys.argv) > 1 else None
rows = collections.defaultdict(lambda: [0, 0.0, 0.0])
for cfg, rec in prof.items():
    for s_, e_, f, shp in rec["timed"]:
        r = rows[(cfg, shp)]
        r[0] += 1; r[1] += s_.elapsed_time(e_) * 1e3; r[2] = f
tot = 0.0
for (cfg, shp), (n, us, f) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    if want is not None and (isinstance(cfg, tuple) or shp[0] != want):
        continue
    tot += us / N
    print(f"{str(cfg):>28s} {str(shp):>30s}  {n / N:5.1f}/step  {us / n:7.1f} us  {f / (us / n) / 1e6:6.1f} TF  {us / N / 1e3:6.3f} ms/step")
print("total ms/step", tot / 1e3)
