"""tests/test_nets3d_gpu.py::test_bn_groups_3d_equal_separate_passes with the pipelined 3x3x3 kernel on / off: worst parameter-gradient
difference between the grouped pass and the two separate passes, per setting."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import ops
from arco_amd.networks.vnetWithArgs import VNet

SP = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 32, 16)
for on in (0, 1):
    ops.conv3d_fl_set(on)
    torch.manual_seed(5)
    m = VNet(n_channels=1, n_classes=2, normalization='batchnorm', has_dropout=False).cuda().train()
    xa, xb = torch.rand(1, 1, *SP, device="cuda"), torch.rand(1, 1, *SP, device="cuda")
    state0 = {k: v.clone() for k, v in m.state_dict().items()}
    params = list(m.parameters())
    wa, wb = torch.randn(1, 2, *SP, device="cuda"), torch.randn(1, 2, *SP, device="cuda")
    loss_of = lambda p, fm, w: (p * w).sum() + sum((f * f).mean() for f in fm)
    pa, _, fa = m(xa); pb, _, fb = m(xb)
    g_sep = torch.autograd.grad(loss_of(pa, fa, wa) + loss_of(pb, fb, wb), params, allow_unused=True)
    m.load_state_dict(state0)
    with ops.bn_groups(2):
        p, _, fm = m(torch.cat((xa, xb)))
        g_grp = torch.autograd.grad(loss_of(p[:1], [f[:1] for f in fm], wa) + loss_of(p[1:], [f[1:] for f in fm], wb), params, allow_unused=True)
    print("fl", on, "out diff", float((p[:1] - pa).abs().max()), float((p[1:] - pb).abs().max()))
    gmax = max(float(g.abs().max()) for g in g_sep if g is not None)
    worst = []
    for (n, _), gs, gg in zip(m.named_parameters(), g_sep, g_grp):
        if gs is None: continue
        scale = float(gs.abs().max()) + 1e-12
        if scale < 1e-4 * gmax: continue
        worst.append((float((gs - gg).abs().max()) / scale, n))
    worst.sort(reverse=True)
    print("   worst rel grad diffs:", worst[:6])
    import numpy as np
    r = np.array([w_[0] for w_ in worst]); print("   n", len(r), "median %.2e p75 %.2e p90 %.2e p95 %.2e frac>2e-2 %.3f" % (np.median(r), np.percentile(r, 75), np.percentile(r, 90), np.percentile(r, 95), (r > 2e-2).mean()))
    num = sum(float(((gs - gg) ** 2).sum()) for gs, gg in zip(g_sep, g_grp) if gs is not None); den = sum(float((gs ** 2).sum()) for gs in g_sep if gs is not None); print("   global rel L2 %.3e" % (num / den) ** 0.5)
    for i, (fs, fg) in enumerate(zip(list(fa) , fm)):
        print("   feature", i, tuple(fs.shape), float((fs - fg[:1]).abs().max() / fs.abs().max()))
