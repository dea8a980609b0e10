"""One dominant-kernel launch class for PMC passes: the dense teacher fea4 GEMM (M=16*65536, N=K=496) and the
fea3 GEMM (M=16*16384, N=K=480).  5 launches each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import ops
for (nb, c, hw) in ((16, 496, 256), (16, 480, 128)):
    x = torch.randn(nb, hw, hw, c, device="cuda").permute(0, 3, 1, 2)
    w = torch.randn(c, c, 1, 1, device="cuda") * 0.05
    xr, ld = ops.rows_view(x)
    wp = ops.pack_weight(w, 1, 0)
    for _ in range(5):
        ops.conv_raw(xr, ld, c, wp, c, nb, hw, hw, 1)
torch.cuda.synchronize()
