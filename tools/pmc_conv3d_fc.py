"""Launch classes of the pipelined 3x3x3 kernels (conv3d_fl.hip: conv3d_fc_kernel, fp32-split; HALF=1: hconv_fc_kernel, f16 storage) for
PMC passes: the V-Net levels below full resolution at the LA patch (4 volumes) - LiTS patch (2 volumes) with HALF=1 -, BN-statistics
epilogue on, a 256 MB tensor touched between launches so that inputs come from HBM; 5 launches per class in the order printed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import ops, _lib as L
HALF = bool(int(os.environ.get("HALF", "0")))
ops.CONV_MMA = 3
nv = 2 if HALF else 4
big = torch.randn(64, 1024, 1024, device="cuda")
levels = ((32, (80, 80, 48)), (64, (40, 40, 24)), (128, (20, 20, 12)), (256, (10, 10, 6))) if HALF else \
         ((32, (56, 56, 40)), (64, (28, 28, 20)), (128, (14, 14, 10)), (256, (7, 7, 5)))
for c, (d3, h, w) in levels:
    wt = torch.randn(c, c, 3, 3, 3, device="cuda") * 0.05
    wp = ops.pack_weight(wt, 27, 0, half=HALF)
    xs = [torch.randn(nv, d3, h, w, c, device="cuda").permute(0, 4, 1, 2, 3) for _ in range(5)]
    if HALF:
        xs = [x.half() for x in xs]
    cfg = L.query("arco_conv_config_mma", 27, nv * d3, h, w, c, c, c, 4 if HALF else 3)
    m = nv * d3 * h * w
    eb = 2 if HALF else 4
    wbytes = 27 * c * c * (2 if HALF else 6)
    print(f"{c}->{c} @{d3}x{h}x{w} x{nv}: config {cfg}  algorithmic {m * c * eb * 2 + wbytes} B (+ BN slabs)", flush=True)
    for x in xs:
        big.add_(1.0)
        xr, ld = ops.rows_view(x)
        ops.conv_raw(xr, ld, c, wp, c, nv, h, w, 27, stats=True, d3=d3, half=HALF)
torch.cuda.synchronize()
