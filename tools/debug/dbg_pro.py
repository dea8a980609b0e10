import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from arco_amd import ops, _lib as L
from arco_amd._contrast import rows_view
def _rand(shape, seed, scale=1.0):
    rs = np.random.RandomState(seed); return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32)).cuda()
k, n, nb, h, w, groups, p = 16, 16, 8, 128, 128, 1, 0.0
z = ops.new_act(nb, k, h, w, "cuda"); z.copy_(_rand((nb, k, h, w), 1, 2.0))
mean, istd = _rand((groups * k,), 2, 0.3), _rand((groups * k,), 3).abs() + 0.5
gamma, beta = _rand((k,), 4), _rand((k,), 5, 0.2)
wt = _rand((n, k, 3, 3), 6, 0.1); bias = _rand((n,), 7)
zr, ld = rows_view(z); m = nb * h * w
a = ops.new_act(nb, k, h, w, "cuda")
L.call("arco_bn_act_fwd", L.ptr(zr), ld, m, k, L.ptr(mean), L.ptr(istd), L.ptr(gamma), L.ptr(beta), 0.01, 0, 0.0, 0, h * w, L.ptr(a), k, None, groups)
ar, lda = rows_view(a)
wp = ops.pack_weight(wt, 9, 0)
y0, _ = ops.conv_raw(ar, lda, k, wp, n, nb, h, w, 9, bias=bias)
pro = L.act_pro(mean, istd, gamma, beta, 0.01, groups, 0, 0.0, 0, None)
y1, _ = ops.conv_raw(zr, ld, k, wp, n, nb, h, w, 9, bias=bias, pro=pro)
y2, _ = ops.conv_raw(zr, ld, k, wp, n, nb, h, w, 9, bias=bias)
zz = torch.zeros_like(z); zzr, _ = rows_view(zz)
y3, _ = ops.conv_raw(zzr, ld, k, wp, n, nb, h, w, 9, bias=bias)
torch.cuda.synchronize()
print("y1 vs y0 (expected equal):", float((y1 - y0).abs().max()))
print("y1 vs conv(raw z):", float((y1 - y2).abs().max()))
print("y1 vs bias only:", float((y1 - y3).abs().max()))
d = (y1 - y0).abs().amax(dim=1)[0]
print("diff map rows (first image):", [float(d[i].max()) for i in (0, 1, 2, 31, 32, 33, 64, 127)])
print("diff map cols:", [float(d[:, i].max()) for i in (0, 1, 2, 15, 16, 17, 127)])
