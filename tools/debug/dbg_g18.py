"""tests/test_nets3d_gpu.py::test_vnet_gradients_strict_vs_float64_reference - the per-tensor deviations from the float64 reference next to the fp32 reference's own."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import fixture_inputs as fx
from test_nets3d_gpu import probe_like
from arco_amd.networks.vnetWithArgs import VNet
g = np.load(os.path.join(os.path.dirname(fx.__file__), "golden", "g18_vnet_strict.npz"), allow_pickle=False)
net = VNet(n_channels=1, n_classes=2, normalization='batchnorm', has_dropout=True).cuda()
net.load_state_dict(fx.vnet_state(52), strict=True); net.train()
x = fx.image_batch(int(g["seed"]), 2, 1, (48, 48, 32)).cuda().requires_grad_(True)
out, _, fmap = net(x, turnoff_drop=True)
loss = (out * probe_like(out, 4)).sum()
for i, f in enumerate(fmap): loss = loss + (f * probe_like(f, 20 + i)).sum()
loss.backward()
stride = int(g["stride"]); names = [str(s) for s in g["grad_names"]]; ref32 = dict(zip(names, g["ref32_dev"]))
params = dict(net.named_parameters()); worst = {}
for n in names:
    if n.endswith(".bias") and ".conv." in n and int(n.split(".")[-2]) % 3 == 0: continue
    flat = params[n].grad.detach().reshape(-1).cpu().numpy(); ref = g["grad::" + n]
    got = flat if flat.size <= 120000 else flat[::stride]
    worst[n] = float(np.abs(got - ref).max()) / float(np.abs(ref).max())
w = np.array(list(worst.values())); r = np.array([ref32[n] for n in worst])
print("hip: worst %.4f median %.4f | fp32 reference: worst %.4f median %.4f" % (w.max(), np.median(w), r.max(), np.median(r)))
for n in sorted(worst, key=worst.get, reverse=True)[:6]: print("  ", n, round(worst[n], 4), "ref32", round(float(ref32[n]), 4))
