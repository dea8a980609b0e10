"""gemm_sp_kernel (csrc/gemm_sp.hip) against igemm_kernel on the wide 1x1 GEMMs: bit equality and microseconds per launch.
python tools/gemm_sp_bench.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arco_amd import ops, _lib as L
PMC = len(sys.argv) > 1 and sys.argv[1] == "pmc"          # under rocprofv3 --pmc: the first shape only, 3 launches per kernel
reps = 3 if PMC else (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
ops.CONV_MMA = 3
dev = "cuda:0"
torch.manual_seed(0)
# (label, M, K, N, residual)
SHAPES = [("fea4 / q_rep dense, 8 images (model_2D.py:51-53)", 8 * 65536, 496, 496, False),
          ("q_rep dense, 16 images", 16 * 65536, 496, 496, False),
          ("fea3 dense 480 @128^2 x8 + residual", 8 * 16384, 480, 480, True),
          ("fea2 dense 448 @64^2 x16 + residual", 16 * 4096, 448, 448, True),
          ("FE_3d fea3 240 @LA full res x1 + residual", 112 * 112 * 80, 240, 240, True),
          ("FE_3d fea2 hi: 32 -> 224 @56x56x40 x4 + residual", 4 * 56 * 56 * 40, 32, 224, True),
          ("FE_3d fea2 lo: 192 -> 224 @28x28x20 x4", 4 * 28 * 28 * 20, 192, 224, False),
          ("ragged: M = 100001, K = 100, N = 252 + residual", 100001, 100, 252, True)]
def run(x, w, res, M, K, N):
    wp = ops.pack_weight(w, 1, 0)
    out, _ = ops.conv_raw(x, K, K, wp, N, 1, 1, M, 1, residual=res, ld_res=N if res is not None else 0)
    return out
for label, M, K, N, has_res in (SHAPES[:1] if PMC else SHAPES):
    x = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev))
    w = (torch.randn(N, K, 1, 1, device=dev) / K ** 0.5).requires_grad_(False)
    res = torch.randn(M, N, device=dev) if has_res else None
    outs, times = {}, {}
    for mode in (0, 1):
        L.query("arco_gemm_sp_set", mode, 1)          # min_tiles 1: the shape decides alone
        ops._cfg_cache.clear()
        o = run(x, w, res, M, K, N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            o = run(x, w, res, M, K, N)
        e1.record(); torch.cuda.synchronize()
        outs[mode], times[mode] = (o.permute(0, 2, 3, 1).reshape(M, N).clone() if o.dim() == 4 else o.clone()), e0.elapsed_time(e1) / reps * 1e3
    L.query("arco_gemm_sp_set", 1, 2048)
    ref = (x.double() @ w.view(N, K).double().t() + (res.double() if has_res else 0)).float()
    a, b = outs[0].reshape(-1), outs[1].reshape(-1)
    same = bool(torch.equal(a, b))
    err = float((b - ref.reshape(-1)).abs().max() / ref.abs().max())
    fl = 2.0 * M * N * K
    byt = 4.0 * (M * K + M * N * (2 if has_res else 1))
    print(f"{label}: igemm {times[0]:8.1f} us ({fl / times[0] / 1e6:6.1f} TFLOP/s)  gemm_sp {times[1]:8.1f} us ({fl / times[1] / 1e6:6.1f} TFLOP/s, {byt / times[1] / 1e6:5.2f} TB/s algorithmic)  "
          f"bit-identical {same}  max err vs fp64 {err:.1e}", flush=True)
