"""1x1-conv GEMM launch times for the shapes of the 2-D step: python tools/micro/gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import _lib as L
if os.environ.get("ARCO_LIB"):
    L.LIB_PATH = os.environ["ARCO_LIB"]
from arco_amd import ops

def timeit(fn, reps=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

shapes = [(65536, 448, 64), (16384, 448, 384), (16384, 384, 128), (65536, 64, 448), (262144, 16, 32), (4096, 256, 256), (1024, 496, 496),
          (16384, 64, 128), (65536, 32, 64), (262144, 32, 16), (4096, 480, 480), (34000, 480, 480), (16384, 384, 448), (1048576, 496, 496)]
for M, N, K in shapes:
    s = int(M ** 0.5) if int(M ** 0.5) ** 2 == M else None
    nb, h, w = (1, s, s) if s else (1, 1, M)
    x = torch.randn(nb, h, w, K, device="cuda").permute(0, 3, 1, 2)
    wt = torch.randn(N, K, 1, 1, device="cuda") * 0.05
    wp = ops.pack_weight(wt, 1, 0)
    xr, ldx = ops.rows_view(x)
    cfg = L.query("arco_conv_config_mma", 1, nb, h, w, K, N, ldx, 3)
    t = timeit(lambda: ops.conv_raw(xr, ldx, K, wp, N, nb, h, w, 1))
    fl = 2.0 * M * N * K
    by = 4.0 * M * (N + K)
    print(f"M={M:8d} N={N:4d} K={K:4d} cfg {cfg}: {t:7.1f} us  {fl / t / 1e6:6.1f} TF   HBM floor {by / 6.3e6:6.1f} us ({by / t / 1e3:6.0f} GB/s)")
