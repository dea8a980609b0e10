"""Fixed-length run of the default 2-D step for `rocprofv3 --kernel-trace`: WARM eager/capture steps + N replayed steps of
BASELINE configs[1] (16 x 256^2, every flag at the trainer default), nothing else.  The kernel summary divided by
(WARM + N) is the per-step kernel time / launch count (eager and replayed steps launch the same kernels).
  cd /tmp && rocprofv3 --kernel-trace -d <out> -- python3 /root/repo/tools/prof_step.py [N] [extra trainer flags...]
  python tools/prof_summary.py <out> profiles/<name>_kernel_stats.csv <WARM + N>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import random
import numpy as np
import torch
from arco_amd import train_arco_2d as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
WARM = 6
random.seed(1337); np.random.seed(1337); torch.manual_seed(1337)
torch.set_num_threads(4)
args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--func", "smc", "--synthetic", "1"] + sys.argv[2:])
st = T.ArcoStep2D(args, "cuda:0")
bs = [(T.synthetic_batch(8, args.patch_size, 4, 100 + 2 * i, "cuda:0"), T.synthetic_batch(8, args.patch_size, 4, 101 + 2 * i, "cuda:0")[0]) for i in range(4)]
for i in range(WARM):
    (l, ll), u = bs[i % 4]
    st.step(l, ll, u, i, 100)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    (l, ll), u = bs[i % 4]
    st.step(l, ll, u, WARM + i, 100)
torch.cuda.synchronize()
print(f"prof_step: {WARM}+{N} steps, {1e3 * (time.perf_counter() - t0) / N:.3f} ms/step (under the profiler when profiled)")
