"""Launch classes of the dominant 3x3 backbone kernel (igemm_kernel<9,64,64,...>) for PMC passes: the 64-channel
level of the U-Net (M = nb*64*64 pixels, nb = 16 and 8): 64->64 and 128->64, with the BN-statistics epilogue as in the step.
5 launches each (inputs differ per launch so the Infinity Cache does not hold them)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import ops
big = torch.randn(64, 1024, 1024, device="cuda")                # 256 MB: evict the caches between launches
for nb in (16, 8):                                              # 16 = the grouped (labelled + unlabelled) passes; 8 = teacher u0 / stats passes
    for ci in (64, 128):
        w = torch.randn(64, ci, 3, 3, device="cuda") * 0.05
        wp = ops.pack_weight(w, 9, 0)
        xs = [torch.randn(nb, 64, 64, ci, device="cuda").permute(0, 3, 1, 2) for _ in range(5)]
        for x in xs:
            big.add_(1.0)
            xr, ld = ops.rows_view(x)
            ops.conv_raw(xr, ld, ci, wp, 64, nb, 64, 64, 9, stats=True, stat_groups=2 if nb == 16 else 1)
torch.cuda.synchronize()
