"""Race hunt: the at-size graph-vs-eager comparison of tests/test_configs_at_size_gpu.py with a SLOW HOST - every kernel-launching
call is followed by a busy-wait, so the GPU streams drain between launches and a stream never sits behind a backlog: work that is
missing a cross-stream dependency then runs too early every time instead of once in a few full-suite runs (the flake appeared on
cold boxes).  python tools/debug/slow_host.py [delay_us ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from arco_amd import _lib as L
import test_configs_at_size_gpu as TC

delay = [0.0]
real_call = L.call
def slow_call(name, *args):
    real_call(name, *args)
    if delay[0] > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < delay[0]:
            pass
L.call = slow_call
for mod in list(sys.modules.values()):           # modules that bound `L` keep the module object: patched attribute is seen
    pass
real_replay = torch.cuda.CUDAGraph.replay
def slow_replay(self):
    real_replay(self)
    if delay[0] > 0:
        time.sleep(delay[0] * 20)
torch.cuda.CUDAGraph.replay = slow_replay

for d_us in [float(x) for x in sys.argv[1:]] or [0.0, 50.0, 300.0]:
    delay[0] = 0.0
    st_e, st_g = TC._make_acdc(["--graphs", "0", "--graph_train", "0"]), TC._make_acdc([])
    TC._drop_off(st_e); TC._drop_off(st_g)
    worst = []
    for it in range(8):
        batch = TC._acdc_batch(20 + it)
        TC._sync_full(st_g, st_e)
        torch.cuda.synchronize()
        for st in (st_e, st_g):
            TC.seed_all(800 + it)
            delay[0] = d_us * 1e-6 if it >= 3 else 0.0        # graphs are captured at the third call
            st.step(*batch)
            delay[0] = 0.0
            torch.cuda.synchronize()
        dev_ = []
        for (k, ve), (_, vg) in zip(st_e.model.named_parameters(), st_g.model.named_parameters()):
            dev_.append((float((ve - vg).abs().max()) / max(1e-12, float(ve.abs().max())), k))
        d, k = max(dev_)
        worst.append(d)
        print(f"delay {d_us:6.0f} us step {it}: worst update deviation {d:.2e} ({k})  terms e/g reco {float(st_e.last_terms['reco']):.6f} / {float(st_g.last_terms['reco']):.6f}", flush=True)
    del st_e, st_g
    torch.cuda.empty_cache()
