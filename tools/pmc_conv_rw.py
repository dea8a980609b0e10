"""Launch classes of conv3x3_rw_kernel<8,1> (one 16-wide output block from <= 32 inputs; conv_sp.hip) for PMC passes, BN-statistics
epilogue as in the step, a 256 MB tensor touched between launches so that inputs come from HBM; 5 launches per class in this order:
  16->16 @256^2 (16 images) | 32->16 @256^2 (16 images) | 16->4 @256^2 (16 images, the logits layer) | 4->16 @256^2 (its data gradient)
  | 16->16 @256^2 (8 images)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import ops, _lib as L
big = torch.randn(64, 1024, 1024, device="cuda")
for nb, ci, co, s in [(16, 16, 16, 256), (16, 32, 16, 256), (16, 16, 4, 256), (16, 4, 16, 256), (8, 16, 16, 256)]:
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wp = ops.pack_weight(w, 9, 0)
    xs = [torch.randn(nb, s, s, ci, device="cuda").permute(0, 3, 1, 2) for _ in range(5)]
    assert L.query("arco_conv_config_mma", 9, nb, s, s, ci, co, ci, 3) == 9358016
    for x in xs:
        big.add_(1.0)
        xr, ld = ops.rows_view(x)
        ops.conv_raw(xr, ld, ci, wp, co, nb, s, s, 9, stats=True, stat_groups=2 if nb == 16 else 1)
torch.cuda.synchronize()
# round 6: the PRO instantiation (consumer-side BatchNorm + LeakyReLU + dropout of the producing layer in the loader): same launch
# classes, input = the producing layer's pre-activation; 16->16 with dropout 0.05 (encoder in_conv), 32->16 without (decoder up4)
for nb, ci, co, s, p in [(16, 16, 16, 256, 0.05), (16, 32, 16, 256, 0.0)]:
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wp = ops.pack_weight(w, 9, 0)
    mean, istd = torch.randn(2 * ci, device="cuda") * 0.1, torch.rand(2 * ci, device="cuda") + 0.5
    gamma, beta = torch.randn(ci, device="cuda"), torch.randn(ci, device="cuda") * 0.1
    xs = [torch.randn(nb, s, s, ci, device="cuda").permute(0, 3, 1, 2) for _ in range(5)]
    assert ops.pro_ok(9, nb, 1, s, s, ci, co, ci, 2)
    for x in xs:
        big.add_(1.0)
        xr, ld = ops.rows_view(x)
        pro = L.act_pro(mean, istd, gamma, beta, 0.01, 2, 1 if p > 0 else 0, p, 12345, None)
        ops.conv_raw(xr, ld, ci, wp, co, nb, s, s, 9, stats=True, stat_groups=2, pro=pro)
torch.cuda.synchronize()
