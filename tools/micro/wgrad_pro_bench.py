"""wgrad_split_kernel with / without the consumer-side activation on its input operand: us per launch (incl. the slab reduction).
ARCO_LIB=<variant .so> python tools/micro/wgrad_pro_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import ops, _lib as L
from arco_amd._contrast import rows_view
big = torch.randn(64, 1024, 1024, device="cuda")
for nb, ci, co, s, p in [(16, 16, 16, 256, 0.0), (16, 16, 16, 256, 0.05), (16, 32, 32, 128, 0.0), (16, 64, 64, 64, 0.2)]:
    z = ops.new_act(nb, ci, s, s, "cuda"); z.normal_()
    dz = ops.new_act(nb, co, s, s, "cuda"); dz.normal_()
    mean, istd = torch.randn(2 * ci, device="cuda") * 0.1, torch.rand(2 * ci, device="cuda") + 0.5
    gamma, beta = torch.randn(ci, device="cuda"), torch.randn(ci, device="cuda") * 0.1
    like = torch.empty(co, ci, 3, 3, device="cuda")
    zr, ld = rows_view(z); dzr, ldz = rows_view(dz)
    res = []
    for use_pro in (False, True):
        def run():
            pro = L.act_pro(mean, istd, gamma, beta, 0.01, 2, 1 if p > 0 else 0, p, 12345, None) if use_pro else None
            ops.conv_wgrad(dzr, ldz, co, zr, ld, ci, 9, nb, s, s, like, pro=pro)
        for _ in range(3):
            run()
        ts = []
        for _ in range(10):
            big.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res.append(sorted(ts)[len(ts) // 2])
    print(f"{ci}->{co} @{s}^2 x{nb} p={p}: plain {res[0]:.1f} us, PRO {res[1]:.1f} us")
