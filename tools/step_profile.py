"""Host-side stage timing of ArcoStep2D.step (diagnostic; synchronises between stages)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_2d as T, _contrast as C_, glue, head
import cProfile, pstats

args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--synthetic", "1", "--conv_mma", os.environ.get("MMA", "f32x3")])
st = T.ArcoStep2D(args, "cuda:0")
l, ll = T.synthetic_batch(8, args.patch_size, 4, 1, "cuda:0")
u, _ = T.synthetic_batch(8, args.patch_size, 4, 2, "cuda:0")
for _ in range(3):
    st.step(l, ll, u)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    st.step(l, ll, u)
t1 = time.perf_counter()      # host time to ENQUEUE 3 steps (no sync)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue per step {1e3*(t1-t0)/3:.1f} ms ; wall per step {1e3*(t2-t0)/3:.1f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    st.step(l, ll, u)
torch.cuda.synchronize()
pr.disable()
ps = pstats.Stats(pr).sort_stats("cumulative")
ps.print_stats(45)
