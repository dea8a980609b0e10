"""Per-kernel timing of the 3x3x3 conv family on the V-Net shapes (HIP events, 10 reps): forward (= data gradient
kernel) and weight gradient.  python tools/kernel_bench3d.py [nv]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import ops

def timeit(fn, reps=10):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

def run(nv, ci, co, sp):
    x = torch.randn(nv, *sp, ci, device="cuda").permute(0, 4, 1, 2, 3)
    wt = (torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    xg = x.clone().requires_grad_(True)
    fl = 2.0 * nv * sp[0] * sp[1] * sp[2] * ci * co * 27
    with torch.no_grad():
        ms = timeit(lambda: ops.conv(x, wt, None))
    ops.CONV_MMA = int(os.environ.get("MMA", ops.CONV_MMA))
    y = ops.conv(xg, wt, None)
    gy = torch.randn_like(y)
    def bwd():
        xg.grad = None; wt.grad = None
        y.backward(gy, retain_graph=True)
    msb = timeit(bwd)
    print(f"3x3x3 nv={nv} {ci:4d}->{co:4d} @{sp}: fwd {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF | dgrad+wgrad {msb*1e3:8.1f} us {2*fl/msb/1e9:6.1f} TF")

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for ci, co, sp in ((1, 16, (112, 112, 80)), (16, 16, (112, 112, 80)), (32, 32, (56, 56, 40)), (64, 64, (28, 28, 20)),
                   (128, 128, (14, 14, 10)), (256, 256, (7, 7, 5))):
    run(nv, ci, co, sp)
