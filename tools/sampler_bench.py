"""Host sampler timing: one step's draws (4 classes: anchors + 131072 negatives) sequential vs threaded multi-call."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import samplers
jobs = []
for na in (60000, 25000, 9000, 30000):
    jobs += [(na, 256), (4096, 131072)]
def t(fn, n=30):
    torch.manual_seed(0); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3
print("single calls        %.2f ms" % t(lambda: [samplers.grid_monte_carlo_sample(h, s) for h, s in jobs]))
for thr in (0, 1, 2, 4, 8):
    print("many, threads=%d     %.2f ms" % (thr, t(lambda: samplers.grid_sample_many(jobs, False, max_threads=thr))))
pin = torch.empty(sum(s for _, s in jobs), dtype=torch.int64).pin_memory() if torch.cuda.is_available() else None
if pin is not None:
    print("many, pinned out    %.2f ms" % t(lambda: samplers.grid_sample_many(jobs, False, out=pin, max_threads=8)))
