"""1x1 weight-gradient kernel (wgrad_kernel, fp32 MFMA) in isolation: the step's shapes, HIP-event time per layer incl. its slab reduction."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import ops
for m, co, ci in [(524288, 496, 496), (131072, 480, 480), (65536, 448, 448), (40000, 252, 200), (16384, 448, 448), (1024, 496, 496), (4096, 480, 480), (16384, 384, 128), (262144, 16, 32), (4096, 128, 256), (65536, 32, 64), (4096, 256, 256)]:
    xs = [torch.randn(1, m, 1, ci, device="cuda").permute(0, 3, 1, 2) for _ in range(4)]
    gs = [torch.randn(1, m, 1, co, device="cuda").permute(0, 3, 1, 2) for _ in range(4)]
    wt = torch.zeros(co, ci, 1, 1, device="cuda")
    def run(i):
        xr, ldx = ops.rows_view(xs[i]); dr, ldz = ops.rows_view(gs[i])
        return ops.conv_wgrad(dr, ldz, co, xr, ldx, ci, 1, 1, m, 1, wt)
    ref = (gs[0].reshape(co, m).double() @ xs[0].reshape(ci, m).double().t())
    got = run(0).reshape(co, ci).double()
    err = float((got - ref).abs().max() / ref.abs().max())
    for i in range(4): run(i)
    torch.cuda.synchronize()
    n = 16
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): run(i % 4)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"M={m:7d} {co:4d}x{ci:4d}: {us:7.1f} us  {2.0 * m * co * ci / us / 1e6:6.1f} TF  rel err {err:.1e}")
