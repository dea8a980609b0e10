"""1x1 split-bf16 GEMM shapes whose N prefers the 224-wide tile (N = 448 / 192 / 224): time per tile choice (ARCO_GEMM224)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import ops

def timeit(fn, reps=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

for (m, k, n, res) in ((65536, 64, 448, True), (16384, 384, 448, False), (65536, 448, 448, False), (16384, 448, 384, False), (65536, 448, 64, False),
                       (4 * 56 * 56 * 40, 192, 192, False), (4 * 56 * 56 * 40, 32, 224, True), (2 * 80 * 80 * 48, 192, 224, False), (2 * 80 * 80 * 48, 224, 224, False)):
    x = torch.randn(m, k, device="cuda")
    w = torch.randn(n, k, 1, 1, device="cuda") * 0.05
    r = torch.randn(m, n, device="cuda") if res else None
    wp = ops.pack_weight(w, 1, 0)
    t = timeit(lambda: ops.conv_raw(x, k, k, wp, n, 1, 1, m, 1, residual=r, ld_res=n if res else 0))
    print(f"M={m:7d} K={k:4d} N={n:4d} res={int(res)}: {t:7.1f} us  {2.0 * m * n * k / t / 1e6:6.1f} TF")
