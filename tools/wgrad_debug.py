import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from arco_amd import ops
def run(mode, x, gy, co, ci, nb, h, w):
    ops.CONV_MMA = mode
    wt = torch.zeros(co, ci, 3, 3, device="cuda")
    xr, ldx = ops.rows_view(x); dr, ldz = ops.rows_view(gy)
    return ops.conv_wgrad(dr, ldz, co, xr, ldx, ci, 9, nb, h, w, wt).cpu()
nb, ci, co, h, w = 1, 16, 16, 8, 16
x = torch.ones(nb, h, w, ci, device="cuda").permute(0, 3, 1, 2)
gy = torch.ones(nb, h, w, co, device="cuda").permute(0, 3, 1, 2)
d3, d0 = run(3, x, gy, co, ci, nb, h, w), run(0, x, gy, co, ci, nb, h, w)
print("ones: d0[0,0]\n", d0[0, 0].numpy(), "\nd3[0,0]\n", d3[0, 0].numpy(), "\nd3 unique", torch.unique(d3)[:20])
# delta in gy at pixel (3, 5), channel 2 ; x = pixel index ramp in channel 1
gy = torch.zeros(nb, h, w, co, device="cuda"); gy[0, 3, 5, 2] = 1.0
x = torch.zeros(nb, h, w, ci, device="cuda"); x[0, :, :, 1] = torch.arange(h * w, device="cuda").view(h, w).float()
d3, d0 = run(3, x.permute(0, 3, 1, 2), gy.permute(0, 3, 1, 2), co, ci, nb, h, w), run(0, x.permute(0, 3, 1, 2), gy.permute(0, 3, 1, 2), co, ci, nb, h, w)
print("delta: expected dW[2,1]\n", d0[2, 1].numpy(), "\ngot\n", d3[2, 1].numpy())
nz = torch.nonzero(d3.abs() > 1e-6)
print("nonzero entries of d3 (co, ci, ky, kx):", nz[:30].tolist())
