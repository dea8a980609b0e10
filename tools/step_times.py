import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_2d as T
g = sys.argv[1] if len(sys.argv) > 1 else "1"
args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--synthetic", "1", "--graphs", g])
st = T.ArcoStep2D(args, "cuda:0")
l, ll = T.synthetic_batch(8, args.patch_size, 4, 1, "cuda:0")
u, _ = T.synthetic_batch(8, args.patch_size, 4, 2, "cuda:0")
ts = []
for i in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.step(l, ll, u)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("graphs", g, " ".join(f"{t:.0f}" for t in ts))
print("mem GB alloc/reserved", torch.cuda.max_memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9)
