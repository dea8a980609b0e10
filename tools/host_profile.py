"""cProfile of the host side of the 2-D step (where do the ~14 ms of launch work per step go)."""
import os, sys, cProfile, pstats
os.environ.setdefault("OMP_NUM_THREADS", "4"); os.environ.setdefault("MKL_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_2d as T
args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--synthetic", "1", "--graphs", "1"])
st = T.ArcoStep2D(args, "cuda:0")
l, ll = T.synthetic_batch(8, args.patch_size, 4, 1, "cuda:0")
u, _ = T.synthetic_batch(8, args.patch_size, 4, 2, "cuda:0")
for _ in range(8):
    st.step(l, ll, u)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    st.step(l, ll, u)
torch.cuda.synchronize()
pr.disable()
ps = pstats.Stats(pr); ps.sort_stats("tottime").print_stats(28)
