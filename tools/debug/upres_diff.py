import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import ops, _lib as L
ops.CONV_MMA = 3
torch.manual_seed(5)
for (nv, ci, co, lo_sp, hi_sp) in ((2, 32, 224, (14, 14, 10), (28, 28, 20)), (1, 20, 100, (5, 7, 6), (10, 13, 11))):
    x = torch.randn(nv, ci, *hi_sp, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    lo = torch.randn(nv, co, *lo_sp, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(co, ci, 1, 1, 1, device="cuda") / ci ** 0.5
    ys = []
    for fuse in (False, True):
        L.query("arco_gemm_sp_set", 1, 1); ops._cfg_cache.clear(); ops.UPRES_FUSE = fuse
        with torch.no_grad():
            ys.append(ops.conv_upres(x, w, lo).clone())
    d = (ys[0] - ys[1]).abs()
    ref = torch.nn.functional.conv3d(x.double(), w.double()) + torch.nn.functional.interpolate(lo.double(), size=hi_sp, mode="trilinear", align_corners=True)
    print(hi_sp, "max |fused - unfused|", float(d.max()), "differing elements", int((d > 0).sum()), "of", d.numel(),
          "| err vs fp64: unfused", float((ys[0].double() - ref).abs().max()), "fused", float((ys[1].double() - ref).abs().max()))
# zero weights: the output IS the blend
nv, ci, co, lo_sp, hi_sp = 2, 32, 224, (14, 14, 10), (28, 28, 20)
x = torch.randn(nv, ci, *hi_sp, device="cuda").contiguous(memory_format=torch.channels_last_3d)
lo = torch.randn(nv, co, *lo_sp, device="cuda").contiguous(memory_format=torch.channels_last_3d)
w0 = torch.zeros(co, ci, 1, 1, 1, device="cuda")
L.query("arco_gemm_sp_set", 1, 1); ops._cfg_cache.clear(); ops.UPRES_FUSE = True
with torch.no_grad():
    yf = ops.conv_upres(x, w0, lo).clone()
    up = ops.trilinear(lo, hi_sp).clone()
    ops.UPRES_FUSE = False
    yu = ops.conv_upres(x, w0, lo).clone()
print("zero weights: fused vs trilinear kernel", float((yf - up).abs().max()), int(((yf - up) != 0).sum()), "| unfused vs trilinear kernel", float((yu - up).abs().max()))
w = torch.randn(co, ci, 1, 1, 1, device="cuda") / ci ** 0.5
lo0 = torch.zeros_like(lo)
with torch.no_grad():
    ops.UPRES_FUSE = True; a = ops.conv_upres(x, w, lo0).clone()
    ops.UPRES_FUSE = False; b = ops.conv_upres(x, w, lo0).clone()
    c = ops.conv(x, w).clone()
print("zero lo: fused vs unfused", float((a - b).abs().max()), "fused vs plain conv", float((a - c).abs().max()), "unfused vs plain conv", float((b - c).abs().max()))
