import sys; sys.path.insert(0, '.')
import numpy as np, torch
from arco_amd import ops
from arco_amd.networks.vnetWithArgs import VNet
dev = "cuda:0"
torch.manual_seed(3)
net = VNet(n_channels=1, n_classes=2, normalization='batchnorm', has_dropout=False).to(dev).train()
rs = np.random.RandomState(1)
x = torch.from_numpy(rs.uniform(size=(2, 1, 64, 64, 32)).astype(np.float32)).to(dev)
tgt = torch.from_numpy(rs.standard_normal((2, 2, 64, 64, 32)).astype(np.float32)).to(dev)
sd = {k: v.clone() for k, v in net.state_dict().items()}
res = {}
for half, scale_ in ((False, 1.0), (True, 16384.0), (True, 1024.0), (True, 65536.0 * 16)):
    ops.ACT_HALF = half; ops.LOSS_SCALE = scale_
    ops.bump_weight_epoch(); net.load_state_dict(sd); net.zero_grad()
    out, f0, fm = net(x)
    loss = ((out - tgt) ** 2).mean() + sum((f ** 2).mean() for f in fm) * 0.1
    loss.backward()
    s = scale_ if half else 1.0
    res[(half, scale_)] = {n: p.grad.detach().clone() / s for n, p in net.named_parameters() if p.grad is not None}
ref = res[(False, 1.0)]
for key in list(res)[1:]:
    print("==== scale", key[1])
    for n, g in ref.items():
        gh = res[key][n]
        gn = float(g.norm())
        if gn < 1e-12: continue
        print(f"{n:40s} |g| {gn:.3e} max {float(g.abs().max()):.3e} relL2 {float((gh-g).norm())/gn:.4f} finite {bool(torch.isfinite(gh).all())}")
