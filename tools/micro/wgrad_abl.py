"""3x3 split-bf16 weight-gradient kernel (wgrad_split_kernel) in isolation: HIP-event time per launch for the step's shapes,
rotating over 4 buffer pairs.  ARCO_WGRAD_ABL bits: 1 no MFMA phase, 2 no staging, 4 no global loads - only in a library built with
`make -C arco_amd/csrc -B EXTRA=-DARCO_WGRAD_ABLATION` (the branches are compiled out of the product library)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import ops
shapes = [(16, 64, 64, 64), (16, 32, 32, 128), (16, 128, 128, 32), (16, 16, 16, 256), (16, 16, 32, 256), (16, 128, 64, 64), (16, 256, 256, 16)]
for nb, ci, co, s in shapes:
    xs = [torch.randn(nb, s, s, ci, device="cuda").permute(0, 3, 1, 2) for _ in range(4)]
    gs = [torch.randn(nb, s, s, co, device="cuda").permute(0, 3, 1, 2) for _ in range(4)]
    wt = torch.zeros(co, ci, 3, 3, device="cuda")
    def run(i):
        xr, ldx = ops.rows_view(xs[i]); dr, ldz = ops.rows_view(gs[i])
        return ops.conv_wgrad(dr, ldz, co, xr, ldx, ci, 9, nb, s, s, wt)
    for i in range(4): run(i)
    torch.cuda.synchronize()
    n = 16
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): run(i % 4)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = 2.0 * 9 * nb * s * s * ci * co
    print(f"{ci:4d}->{co:4d} @{s:3d}^2 x{nb}: {us:7.1f} us (incl. reduce)  {fl / us / 1e6:6.1f} TF   in {4e-6 * nb * s * s * (ci + co):6.1f} MB")
