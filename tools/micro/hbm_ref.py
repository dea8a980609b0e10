"""What does a streaming pass over the 16 x 256^2 x 16-channel plane cost on this chip?  Times (HIP events, 20 launches each,
rotating over 6 buffers so that nothing stays in the 256 MB infinity cache): torch copy, the BN/activation pass, the
resident-weights 3x3 kernel (16 -> 16) - all 67 MB in + 67 MB out."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import ops, _lib as L
nb, c, s = 16, 16, 256
xs = [torch.randn(nb, s, s, c, device="cuda").permute(0, 3, 1, 2) for _ in range(6)]
ys = [torch.empty(nb, s, s, c, device="cuda").permute(0, 3, 1, 2) for _ in range(6)]
w = torch.randn(c, c, 3, 3, device="cuda") * 0.05
wp = ops.pack_weight(w, 9, 0)
mean = torch.zeros(c, device="cuda"); istd = torch.ones(c, device="cuda"); gamma = torch.ones(c, device="cuda"); beta = torch.zeros(c, device="cuda")


def t(fn, n=24):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i % 6)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def copy(i): ys[i].copy_(xs[i])
def bn(i):
    xr, ld = ops.rows_view(xs[i]); yr, ldy = ops.rows_view(ys[i])
    L.call("arco_bn_act_fwd", L.ptr(xr), ld, nb * s * s, c, L.ptr(mean), L.ptr(istd), L.ptr(gamma), L.ptr(beta), 0.01, 0, 0.0, 0, s * s, L.ptr(yr), ldy, None, 1)
def conv(i):
    xr, ld = ops.rows_view(xs[i])
    ops.conv_raw(xr, ld, c, wp, c, nb, s, s, 9, stats=True, stat_groups=2)
for name, fn in (("copy", copy), ("bn_act_fwd", bn), ("conv3x3_rw<8,1> 16->16", conv)):
    us = t(fn)
    print(f"{name:28s} {us:7.1f} us  {134.2 / us * 1e3 / 1e3:5.2f} TB/s")
