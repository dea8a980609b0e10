"""Ablation timing of the split-bf16 3x3 forward kernels: ARCO_LIB=<path to a libarco_hip.so variant> python tools/micro/conv_abl.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import _lib as L
if os.environ.get("ARCO_LIB"):
    L.LIB_PATH = os.environ["ARCO_LIB"]
from arco_amd import ops

def timeit(fn, reps=40):
    for _ in range(8): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

nb = 16
out = []
for ci, co, s in [(64, 64, 64), (32, 32, 128), (128, 128, 32), (32, 64, 64), (16, 16, 256), (256, 256, 16)]:
    x = torch.randn(nb, s, s, ci, device="cuda").permute(0, 3, 1, 2)
    wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wp = ops.pack_weight(wt, 9, 0)
    xr, ldx = ops.rows_view(x)
    t = timeit(lambda: ops.conv_raw(xr, ldx, ci, wp, co, nb, s, s, 9, stats=True))
    fl = 2.0 * nb * s * s * ci * co * 9
    out.append(f"{ci}->{co}@{s}: {t:6.1f} us {fl / t / 1e6:6.1f} TF")
print(os.environ.get("ARCO_LIB", "product"), " | ".join(out))
