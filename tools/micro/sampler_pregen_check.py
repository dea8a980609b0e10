import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from arco_amd import samplers
jobs = []
for k, na in enumerate((572607, 86, 217, 210)):
    jobs += [(na, 256), (4096, 256 * 512)]
tot = sum(s for _, s in jobs)
def run(pre, thr, seed):
    torch.manual_seed(seed)
    torch.rand(7)            # mid-block position
    if pre: samplers.pregen(pre, background=True)
    buf = torch.empty(tot, dtype=torch.int64)
    t0 = time.perf_counter(); samplers.grid_sample_many(jobs, False, out=buf, max_threads=thr); dt = (time.perf_counter() - t0) * 1e3
    return buf, torch.get_rng_state().clone(), dt
ref, st_ref, t_ref = run(0, 8, 3)
for pre in (3 << 20, 700000, 10000):          # enough / exhausted inside a negative job / exhausted inside the first anchor job
    for thr in (8, 1):
        got, st, dt = run(pre, thr, 3)
        print(pre, thr, "equal", torch.equal(ref, got), "state", torch.equal(st_ref, st), f"{dt:.2f} ms (plain {t_ref:.2f})")
# stale blocks: generator advanced after pregen
torch.manual_seed(3); torch.rand(7); samplers.pregen(3 << 20); torch.rand(1)
b1 = torch.empty(tot, dtype=torch.int64); samplers.grid_sample_many(jobs, False, out=b1); s1 = torch.get_rng_state().clone()
torch.manual_seed(3); torch.rand(7); torch.rand(1)
b2 = torch.empty(tot, dtype=torch.int64); samplers.grid_sample_many(jobs, False, out=b2); s2 = torch.get_rng_state().clone()
print("stale blocks ignored:", torch.equal(b1, b2), torch.equal(s1, s2))
ts = []
for i in range(10):
    torch.manual_seed(i); samplers.pregen(3 << 20); time.sleep(0.02)
    t0 = time.perf_counter(); samplers.grid_sample_many(jobs, False, out=b1, max_threads=8); ts.append((time.perf_counter() - t0) * 1e3)
print("pregen + draw ms:", " ".join(f"{t:.2f}" for t in ts))
ts = []
for i in range(10):
    torch.manual_seed(i)
    t0 = time.perf_counter(); samplers.grid_sample_many(jobs, False, out=b1, max_threads=8); ts.append((time.perf_counter() - t0) * 1e3)
print("plain draw ms:   ", " ".join(f"{t:.2f}" for t in ts))
