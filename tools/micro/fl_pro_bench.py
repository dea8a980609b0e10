"""Stage -> stage link of a V-Net ConvBlock in a gradient-free pass: bn_act_fwd + conv3d_fc_kernel against conv3d_fc_kernel<PRO> (the
activation in the loaders), per level.  python tools/micro/fl_pro_bench.py [nv]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import ops, _lib as L
ops.CONV_MMA = 3


def timeit(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


nv = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for c, sp in ((32, (56, 56, 40)), (64, (28, 28, 20)), (128, (14, 14, 10)), (256, (7, 7, 5))):
    d3, h, w = sp
    z = torch.randn(nv, d3, h, w, c, device="cuda").permute(0, 4, 1, 2, 3)
    wt = torch.randn(c, c, 3, 3, 3, device="cuda") * 0.05
    wp = ops.pack_weight(wt, 27, 0)
    zr, ld = ops.rows_view(z)
    m = nv * d3 * h * w
    mean, istd = torch.randn(c, device="cuda") * 0.1, torch.rand(c, device="cuda") + 0.5
    gamma, beta = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.1
    a = torch.empty_like(z)
    ar, _ = ops.rows_view(a)
    pro = L.act_pro(mean, istd, gamma, beta, 0.0, 1, 0, 0.0, 0, None)
    t_apply = timeit(lambda: ops._bn_apply(zr, ld, m, c, mean, istd, gamma, beta, 0.0, 0, 0.0, 0, d3 * h * w, a, c, 1))
    t_conv = timeit(lambda: ops.conv_raw(ar, ld, c, wp, c, nv, h, w, 27, stats=True, d3=d3))
    t_pro = timeit(lambda: ops.conv_raw(zr, ld, c, wp, c, nv, h, w, 27, stats=True, d3=d3, pro=pro))
    y0 = ops.conv_raw(ar, ld, c, wp, c, nv, h, w, 27, stats=True, d3=d3)[0]
    y1 = ops.conv_raw(zr, ld, c, wp, c, nv, h, w, 27, stats=True, d3=d3, pro=pro)[0]
    print(f"nv={nv} {c:3d} ch @{sp}: apply {t_apply:6.1f} + conv {t_conv:6.1f} = {t_apply + t_conv:6.1f} us | conv<PRO> {t_pro:6.1f} us | {'identical' if torch.equal(y0, y1) else 'DIFFERENT'}", flush=True)
