"""The default 2-D step at the headline size with / without ARCO_FORCE_DIST=1 (a one-rank `nccl` group: every exchange of
arco_amd/dist.py goes through RCCL and ProcessGroupNCCL's stream beside the step's two streams).  Prints one JSON line: loss terms
of six chained steps, a checksum of the final weights, the worst relative gradient difference over N executions of one step from
one snapshot, and ms per step.  tests/test_dist_gpu.py compares the two modes.   python tools/force_dist_check.py [executions]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import torch.distributed as td
from arco_amd import dist as adist, ops
rank, world = adist.init()
import test_configs_at_size_gpu as TC
n_exec = int(sys.argv[1]) if len(sys.argv) > 1 else 30
calls = {"all_reduce": 0, "all_gather": 0, "broadcast": 0}
if adist.is_dist():
    assert td.get_backend() == "nccl" and world == 1
    for name in ("all_reduce", "all_gather_into_tensor", "all_gather", "broadcast"):
        real = getattr(td, name)
        def wrap(*a, _real=real, _k=("all_gather" if "gather" in name else name), **kw):
            calls[_k] += 1
            return _real(*a, **kw)
        setattr(td, name, wrap)
st = TC._make_acdc([])
TC._drop_off(st)
terms = []
for it in range(6):
    TC.seed_all(800 + it)
    st.step(*TC._acdc_batch(20 + it))
    torch.cuda.synchronize()
    terms.append({k: float(v) for k, v in st.last_terms.items()})
per_step_calls = {k: v / 6.0 for k, v in calls.items()}
checksum = float(st.optimizer.flat_p.double().abs().sum())
snap = TC._snapshot(st)
batch = TC._acdc_batch(30)
ref, worst = None, 0.0
for t in range(n_exec):
    TC._restore(st, snap)
    TC.seed_all(900)
    st.step(*batch)
    torch.cuda.synchronize()
    g = st.optimizer.flat_g
    if ref is None:
        ref = g.clone()
    else:
        worst = max(worst, float((g - ref).abs().max()) / float(ref.abs().max()))
# ms per step (resident batches, host-driven like bench.py's timed region)
batches = [TC._acdc_batch(40 + i) for i in range(4)]
for i in range(5):
    st.step(*batches[i % 4])
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 40
for i in range(K):
    st.step(*batches[i % 4])
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
print("FORCE_DIST " + json.dumps(dict(dist=bool(adist.is_dist()), terms=terms, checksum=checksum, worst_repro=worst, ms_per_step=round(ms, 3),
                                      collectives_per_step=per_step_calls)))
if td.is_initialized():
    td.destroy_process_group()
