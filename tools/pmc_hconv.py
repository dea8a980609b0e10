"""f16-storage convolution launches for PMC passes (HBM bytes, LDS bank conflicts): 5 launches per class, a 256 MB tensor touched
between launches.  Classes at the LiTS shard's shapes (2 volumes): 16->16 @160x160x96 (hconv_rw_kernel, or hconv_kernel<9,128,16>
with ARCO_HCONV_RW=0) | 32->32 @80x80x48 (hconv_kernel<9,128,32>) | 64->64 @40x40x24 (flat tiles) | weight gradient 32->32 @80x80x48"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import ops
big = torch.randn(64, 1024, 1024, device="cuda")
nv = 2
for c, sp in ((16, (160, 160, 96)), (32, (80, 80, 48)), (64, (40, 40, 24))):
    w = (torch.randn(c, c, 3, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    wp = ops.pack_weight(w, 27, 0, half=True)
    xs = [torch.randn((nv, *sp, c), device="cuda").half().movedim(-1, 1) for _ in range(5)]
    for x in xs:
        big.add_(1.0)
        xr, ld = ops.rows_view(x)
        ops.conv_raw(xr, ld, c, wp, c, nv, sp[1], sp[2], 27, d3=sp[0], sp=sp, stats=True, half=True)
    if c == 32:
        for x in xs:
            big.add_(1.0)
            xr, ld = ops.rows_view(x)
            ops.conv_wgrad(xr, ld, c, xr, ld, c, 27, nv, sp[1], sp[2], w, d3=sp[0])
torch.cuda.synchronize()
