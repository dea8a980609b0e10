import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests", ROOT + "/oracle"): sys.path.insert(0, p)
import numpy as np, torch
import arco_oracle as orc, fixture_inputs as fx
from arco_amd.networks.vnetWithArgs import VNet
def probe_like(t, seed):
    rs = np.random.RandomState(seed); return torch.from_numpy(rs.standard_normal(tuple(t.shape)).astype(np.float32))
for size, b in (((32, 32, 32), 2), ((48, 48, 32), 2)):
    sd = fx.vnet_state(52); x = fx.image_batch(9, b, 1, size)
    for dt in (torch.float32, torch.float64):
        sdo = {k: (v.to(dt) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
        xo = x.to(dt).clone().requires_grad_(True)
        out_o, _, fm_o = orc.vnet_forward(xo, sdo)
        lo = (out_o * probe_like(out_o, 4).to(dt)).sum() + sum((f * probe_like(f, 20 + i).to(dt)).sum() for i, f in enumerate(fm_o))
        lo.backward()
        if dt == torch.float32: g32 = {k: v.grad.clone() for k, v in sdo.items() if v.grad is not None}; dx32 = xo.grad.clone()
        else: g64 = {k: v.grad.clone() for k, v in sdo.items() if v.grad is not None}; dx64 = xo.grad.clone()
    net = VNet(1, 2, normalization='batchnorm', has_dropout=True).cuda(); net.load_state_dict(sd, strict=True); net.train()
    xg = x.cuda().requires_grad_(True)
    out_g, _, fm_g = net(xg, turnoff_drop=True)
    lg = (out_g * probe_like(out_g, 4).cuda()).sum() + sum((f * probe_like(f, 20 + i).cuda()).sum() for i, f in enumerate(fm_g))
    lg.backward()
    def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
    print(size, "dx: gpu-vs-f64", rel(xg.grad.cpu(), dx64), " cpu32-vs-f64", rel(dx32, dx64))
    worst = []
    for n, p in net.named_parameters():
        if n.endswith(".bias") and ".conv." in n and int(n.split(".")[-2]) % 3 == 0: continue
        worst.append((rel(p.grad.cpu(), g64[n]), rel(g32[n], g64[n]), n))
    worst.sort(reverse=True)
    for w in worst[:6]: print("   gpu %.2e  cpu32 %.2e  %s" % w)

print("---- forward error per feature map (32^3), relative to max")
sd = fx.vnet_state(52); x = fx.image_batch(9, 2, 1, (32, 32, 32))
outs = {}
for dt in (torch.float32, torch.float64):
    sdo = {k: (v.to(dt) if v.is_floating_point() else v).clone() for k, v in sd.items()}
    with torch.no_grad(): o, _, fm = orc.vnet_forward(x.to(dt), sdo)
    outs[dt] = [o] + fm
net = VNet(1, 2, normalization='batchnorm', has_dropout=True).cuda(); net.load_state_dict(sd, strict=True); net.train()
with torch.no_grad(): o, _, fm = net(x.cuda(), turnoff_drop=True)
g = [o] + fm
for i in range(len(g)):
    r64 = outs[torch.float64][i]
    print(i, "gpu %.2e  cpu32 %.2e" % (float((g[i].cpu().double() - r64).abs().max() / r64.abs().max()), float((outs[torch.float32][i].double() - r64).abs().max() / r64.abs().max())))
# single conv+BN stage precision at a large-mean input
from arco_amd import ops
import torch.nn.functional as F
rs = np.random.RandomState(0)
xx = torch.from_numpy((rs.standard_normal((2, 64, 8, 8, 8)) + 3.0).astype(np.float32))
ww = torch.from_numpy((rs.standard_normal((64, 64, 3, 3, 3)) / 40).astype(np.float32)); bb = torch.zeros(64)
z64 = F.conv3d(xx.double(), ww.double(), bb.double(), padding=1)
y64 = F.batch_norm(z64, None, None, torch.ones(64).double(), torch.zeros(64).double(), True, 0.1, 1e-5)
y32 = F.batch_norm(F.conv3d(xx, ww, bb, padding=1), None, None, torch.ones(64), torch.zeros(64), True, 0.1, 1e-5)
yg = ops.conv_bn_act(xx.cuda().contiguous(memory_format=torch.channels_last_3d), ww.cuda(), bb.cuda(), torch.ones(64).cuda(), torch.zeros(64).cuda(), torch.zeros(64).cuda(), torch.ones(64).cuda(), slope=1.0)
print("conv+BN (mean/std of z = %.1f): gpu %.2e cpu32 %.2e" % (float(z64.mean() / z64.std()), float((yg.cpu().double() - y64).abs().max()), float((y32.double() - y64).abs().max())))
