import os, sys
sys.path.insert(0, "/root/repo")
import torch
from arco_amd import _lib as L, ops
torch.manual_seed(0)
for nb, ci, co, h, w in [(8, 32, 32, 32, 48), (8, 16, 32, 64, 96), (8, 64, 64, 16, 48), (8, 32, 64, 32, 48), (16, 64, 64, 64, 64)]:
    x = torch.randn(nb, h, w, ci, device="cuda").permute(0, 3, 1, 2)
    wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wp = ops.pack_weight(wt, 9, 0)
    xr, ldx = ops.rows_view(x)
    res = {}
    for on in (0, 1):
        ops.conv_sp_set(on)
        cfg = L.query("arco_conv_config_mma", 9, nb, h, w, ci, co, ldx, 3)
        out, (ssum, ssq, nmb) = ops.conv_raw(xr, ldx, ci, wp, co, nb, h, w, 9, stats=True, stat_groups=2)
        g = [ssum[:, :nmb // 2].double().sum(1), ssum[:, nmb // 2:].double().sum(1), ssq[:, :nmb // 2].double().sum(1), ssq[:, nmb // 2:].double().sum(1)]
        res[on] = (cfg, out.clone(), g, nmb)
    ops.conv_sp_set(1)
    ref = [res[1][1][:nb // 2].double().sum((0, 2, 3)), res[1][1][nb // 2:].double().sum((0, 2, 3)),
           (res[1][1][:nb // 2].double() ** 2).sum((0, 2, 3)), (res[1][1][nb // 2:].double() ** 2).sum((0, 2, 3))]
    print(nb, ci, co, h, w, "cfg", res[0][0], res[1][0], "nmb", res[0][3], res[1][3], "out equal", torch.equal(res[0][1], res[1][1]),
          "group sums rel err old", [float(((a - b).abs().max() / b.abs().max())) for a, b in zip(res[0][2], ref)],
          "sp", [float(((a - b).abs().max() / b.abs().max())) for a, b in zip(res[1][2], ref)])
