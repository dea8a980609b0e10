"""Per-kernel timing of the conv family on the hot-path shapes (HIP events, 20 reps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import _lib as L, ops

def timeit(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

def fwd(nb, ci, co, h, w, k):
    x = torch.randn(nb, h, w, ci, device="cuda").permute(0, 3, 1, 2)
    wt = torch.randn(co, ci, k, k, device="cuda") * 0.05
    wp = ops.pack_weight(wt, k * k, 0)
    xr, ld = ops.rows_view(x)
    ms = timeit(lambda: ops.conv_raw(xr, ld, ci, wp, co, nb, h, w, k * k, stats=(k == 3)))
    fl = 2.0 * nb * h * w * ci * co * k * k
    print(f"fwd  {k}x{k} nb={nb} {ci:4d}->{co:4d} @{h}x{w}: {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TFLOP/s")

def wg(nb, ci, co, h, w, k):
    x = torch.randn(nb, h, w, ci, device="cuda").permute(0, 3, 1, 2)
    dz = torch.randn(nb, h, w, co, device="cuda").permute(0, 3, 1, 2)
    wt = torch.randn(co, ci, k, k, device="cuda")
    xr, ldx = ops.rows_view(x); dr, ldz = ops.rows_view(dz)
    ms = timeit(lambda: ops.conv_wgrad(dr, ldz, co, xr, ldx, ci, k * k, nb, h, w, wt))
    fl = 2.0 * nb * h * w * ci * co * k * k
    print(f"wgrad {k}x{k} nb={nb} {ci:4d}->{co:4d} @{h}x{w}: {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TFLOP/s")

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "gemm"):
    fwd(16, 496, 496, 256, 256, 1); fwd(16, 480, 480, 128, 128, 1); fwd(16, 448, 448, 64, 64, 1)
    fwd(1, 496, 496, 32, 32, 1); fwd(8, 256, 128, 16, 16, 1); fwd(8, 32, 16, 128, 128, 1)
if which in ("all", "conv"):
    for nb in ((8, 16) if os.environ.get("NB16") else (8,)):
        fwd(nb, 1, 16, 256, 256, 3); fwd(nb, 16, 16, 256, 256, 3); fwd(nb, 32, 16, 256, 256, 3); fwd(nb, 16, 32, 256, 256, 3)
        fwd(nb, 16, 32, 128, 128, 3); fwd(nb, 32, 32, 128, 128, 3); fwd(nb, 64, 32, 128, 128, 3)
        fwd(nb, 64, 64, 64, 64, 3); fwd(nb, 128, 64, 64, 64, 3); fwd(nb, 128, 128, 32, 32, 3)
        fwd(nb, 256, 128, 32, 32, 3); fwd(nb, 256, 256, 16, 16, 3); fwd(nb, 16, 4, 256, 256, 3)
if which in ("all", "wgrad"):
    wg(16, 480, 480, 128, 128, 1); wg(1, 496, 496, 32, 32, 1)
    if os.environ.get("NB16"):
        wg(16, 16, 16, 256, 256, 3); wg(16, 32, 32, 128, 128, 3); wg(16, 64, 64, 64, 64, 3); wg(16, 128, 128, 32, 32, 3); wg(16, 256, 256, 16, 16, 3)
    wg(8, 16, 16, 256, 256, 3); wg(8, 32, 32, 128, 128, 3); wg(8, 64, 64, 64, 64, 3); wg(8, 128, 128, 32, 32, 3)
    wg(8, 256, 256, 16, 16, 3); wg(8, 32, 16, 256, 256, 3)
