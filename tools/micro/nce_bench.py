"""Per-kernel timing of the fused InfoNCE stage (arco_nce_prep / _score / _finish) against the staged route's kernels at the headline
shape: E = 4 entries, Q = 256 queries, 512 negatives, 4096-key banks, D = 496.  python tools/micro/nce_bench.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import _lib as L
L.load()
dev = "cuda:0"
E, Q, Nn, Lb, D = 4, 256, 512, 4096, 496
Dp, Lp, n = D, Lb, E * Q
torch.manual_seed(0)
banks = [torch.randn(Lb, D, device=dev) for _ in range(E)]
A = torch.randn(n, D, device=dev); P = torch.randn(E, D, device=dev)
idx_all = torch.randint(0, Lb, (E, Q + Q * Nn), device=dev, dtype=torch.int64)
bank_ptrs = (ctypes.c_void_p * E)(*[b.data_ptr() for b in banks]); lens = (ctypes.c_int * E)(*[Lb] * E); prow = (ctypes.c_int * E)(*range(E))
An = torch.empty(n, Dp, device=dev); invA = torch.empty(n, device=dev); Pn = torch.empty(E, Dp, device=dev)
M = torch.empty(n, Lp, dtype=torch.int16, device=dev)
n_lt = int(L.query("arco_nce_score_ltiles", Lp))
Wu = torch.empty(E, Q, Lp, device=dev); Zp = torch.empty(n, n_lt, device=dev); Bt = torch.empty(E, Dp, Lp, device=dev); pos = torch.empty(n, device=dev)
gpos = torch.empty(n, device=dev); gsc = torch.empty(n, device=dev); lq = torch.empty(n, device=dev); ls = torch.empty(1, device=dev)
Bn = torch.empty(E, Lp, Dp, device=dev); Btn = torch.empty(E, Dp, Lp, device=dev); S = torch.empty(E, Q, Lp, device=dev); W = torch.empty(E, Q, Lp, device=dev)


def timeit(name, fn, it=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:34s} {e0.elapsed_time(e1) / it * 1e3:8.1f} us")


timeit("nce_prep", lambda: L.call("arco_nce_prep", L.ptr(A), n, L.ptr(P), E, D, Dp, 1e-8, L.ptr(An), L.ptr(invA), L.ptr(Pn), lens, E, L.ptr(idx_all), Q, Q + Q * Nn, Q, Nn, Lp, L.ptr(M)))
timeit("nce_score (with Bt)", lambda: L.call("arco_nce_score", L.ptr(An), Dp, D, bank_ptrs, lens, prow, E, Lp, Q, L.ptr(M), L.ptr(Pn), 0.5, 1e-8, L.ptr(Wu), L.ptr(Zp), L.ptr(pos), L.ptr(Bt)))
timeit("nce_score (no Bt)", lambda: L.call("arco_nce_score", L.ptr(An), Dp, D, bank_ptrs, lens, prow, E, Lp, Q, L.ptr(M), L.ptr(Pn), 0.5, 1e-8, L.ptr(Wu), L.ptr(Zp), L.ptr(pos), None))
timeit("nce_finish", lambda: L.call("arco_nce_finish", L.ptr(pos), n, L.ptr(Zp), Lp, 0.5, 1.0 / n, L.ptr(gpos), L.ptr(gsc), L.ptr(lq), L.ptr(ls)))
timeit("staged: normalize_banks", lambda: L.call("arco_nce_normalize_banks", bank_ptrs, lens, E, D, Dp, Lp, 1e-8, L.ptr(Bn), L.ptr(Btn)))
timeit("staged: score GEMM", lambda: L.call("arco_gemm_batched", L.ptr(An), Dp, Dp, L.ptr(Bn), Lp, L.ptr(S), Lp, Q, E, Q * Dp, Lp * Dp, Q * Lp, 1, None))
timeit("staged: nce_fused", lambda: L.call("arco_nce_fused", L.ptr(S), Lp, lens, prow, E, L.ptr(idx_all), Q, Q + Q * Nn, Q, Nn, L.ptr(An), L.ptr(Pn), Dp, 0.5, L.ptr(W), L.ptr(gpos), L.ptr(lq)))
splits = 16
ws = torch.empty(E, splits, Q, Dp, device=dev); G = torch.empty(n, Dp, device=dev)
timeit("grad GEMM + slab sum", lambda: L.call("arco_gemm_batched", L.ptr(Wu), Lp, Lp, L.ptr(Bt), Dp, L.ptr(G), Dp, Q, E, Q * Lp, Dp * Lp, Q * Dp, splits, L.ptr(ws)))
