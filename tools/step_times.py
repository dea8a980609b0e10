"""Per-step times of the default 2-D step.  SYNC=1: wall time with a sync after every step; default: GPU-timeline deltas
between events recorded after each step (no host syncs: the pipeline runs as in bench.py)."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_2d as T
args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--func", "smc", "--synthetic", "1", "--conv_mma", os.environ.get("MMA", "f32x3"), "--head_levels", os.environ.get("HEAD_LEVELS", "2")])
st = T.ArcoStep2D(args, "cuda:0")
bs = [(T.synthetic_batch(8, args.patch_size, 4, 100 + 2 * i, "cuda:0"), T.synthetic_batch(8, args.patch_size, 4, 101 + 2 * i, "cuda:0")[0]) for i in range(4)]
import gc
if os.environ.get("NOGC"): gc.disable()
N = int(os.environ.get("N", 80))
if os.environ.get("SYNC"):
    ts = []
    for i in range(N):
        (l, ll), u = bs[i % 4]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st.step(l, ll, u, 0, 100)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
else:
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    host = []
    evs[0].record()
    for i in range(N):
        (l, ll), u = bs[i % 4]
        t0 = time.perf_counter()
        st.step(l, ll, u, 0, 100)
        host.append((time.perf_counter() - t0) * 1e3)
        evs[i + 1].record()
    torch.cuda.synchronize()
    ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
    print("host ms:", " ".join(f"{t:.1f}" for t in host))
print("step ms:", " ".join(f"{t:.1f}" for t in ts))
