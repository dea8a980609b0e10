"""The V-Net blocks with `normalization='groupnorm' | 'instancenorm' | 'none'` (vnetWithArgs.py:5-118,145-252) on the HIP path
against golden vectors from the reference's own modules (tests/golden/g17_vnet_norms.npz, oracle/gen_golden.py g17):
ConvBlock / DownsamplingConvBlock / UpsamplingDeconvBlock outputs, input and parameter gradients; the whole V-Net at 32^3;
GnActFn against torch.nn.functional.group_norm in fp64."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import fixture_inputs as fx

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g17_vnet_norms.npz"), allow_pickle=False)
NORMS = ("groupnorm", "instancenorm", "none")


def probe_like(t, seed):
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.standard_normal(tuple(t.shape)).astype(np.float32)).cuda()


def close(got, ref, tol, name):
    err = float(np.abs(got.detach().cpu().numpy() - ref).max())
    assert err <= tol * max(1e-6, float(np.abs(ref).max())), (name, err, float(np.abs(ref).max()))


@pytest.mark.parametrize("norm", NORMS)
@pytest.mark.parametrize("tag", ["cb", "dw", "up"])
def test_vnet_block_variants_vs_reference(norm, tag):
    from arco_amd.networks import vnetWithArgs as V
    mod, shape = {"cb": (lambda: V.ConvBlock(2, 16, 32, normalization=norm), (2, 16, 8, 8, 8)),
                  "dw": (lambda: V.DownsamplingConvBlock(16, 32, normalization=norm), (2, 16, 8, 8, 8)),
                  "up": (lambda: V.UpsamplingDeconvBlock(32, 16, normalization=norm), (2, 32, 4, 4, 4))}[tag]
    mod = mod()
    fx.fill_state(mod, 170 + len(tag) + len(norm))
    mod = mod.cuda().train()
    x = fx.image_batch(171, shape[0], shape[1], shape[2:]).sub(0.5).mul(2.0).cuda().requires_grad_(True)
    y = mod(x)
    (y * probe_like(y, 6)).sum().backward()
    t = f"{norm}_{tag}_"
    close(y, G[t + "y"], 1e-4, t + "y")
    close(x.grad, G[t + "dx"], 1e-3, t + "dx")
    grads = {n: p.grad for n, p in mod.named_parameters()}
    for n, g in grads.items():
        ref = G[t + "g::" + n]
        wname = n[:-4] + "weight"
        if n.endswith(".bias") and wname in grads and float(np.abs(ref).max()) < 1e-3 * float(np.abs(G[t + "g::" + wname]).max()):
            # a conv bias in front of a normalisation that removes the per-(sample, channel) mean (InstanceNorm; GroupNorm with
            # one channel per group) has an analytically ZERO gradient: the reference holds fp32 rounding noise there, this
            # build exact zeros or its own noise - compare at the noise scale
            assert float(g.abs().max()) <= 1e-3 * float(np.abs(G[t + "g::" + wname]).max()), n
            continue
        close(g, ref, 1e-3, t + n)


@pytest.mark.parametrize("norm", NORMS)
def test_vnet_variants_whole_net_vs_reference(norm):
    from arco_amd.networks.vnetWithArgs import VNet
    net = VNet(n_channels=1, n_classes=2, normalization=norm, has_dropout=True)
    assert len(net.state_dict()) == int(G[f"{norm}_vnet_n_state_keys"])
    fx.fill_state(net, 175)
    net = net.cuda().train()
    xv = fx.image_batch(8, 2, 1, (32, 32, 32)).cuda().requires_grad_(True)
    vo, v0, vf = net(xv, turnoff_drop=True)
    t = f"{norm}_vnet_"
    close(vo[..., ::2, ::2, ::2], G[t + "out_sub"], 1e-3, t + "out")
    np.testing.assert_allclose(float(vo.detach().double().pow(2).sum().sqrt()), float(G[t + "out_l2"]), rtol=1e-4)
    for i, f in enumerate(vf):
        close(f[:, ::3, ::3, ::3, ::3], G[t + f"fmap{i}_sub"], 1e-3, t + f"fmap{i}")
        np.testing.assert_allclose(float(f.detach().double().pow(2).sum().sqrt()), float(G[t + f"fmap{i}_l2"]), rtol=1e-4)
    lossv = (vo * probe_like(vo, 4)).sum()
    for i, f in enumerate(vf):
        lossv = lossv + (f * probe_like(f, 20 + i)).sum()
    lossv.backward()
    # gradient norms (element-wise gradients through ~20 ReLU layers are kink-limited: tests/test_nets3d_gpu.py)
    np.testing.assert_allclose(float(xv.grad.double().pow(2).sum().sqrt()), float(G[t + "dx_l2"]), rtol=2e-2)
    params = dict(net.named_parameters())
    for n, ref in zip([str(s) for s in G[t + "grad_names"]], G[t + "grad_l2"]):
        got = float(params[n].grad.double().pow(2).sum().sqrt())
        assert abs(got - ref) <= 3e-2 * max(ref, 1e-3 * float(G[t + "grad_l2"].max())), (n, got, ref)


@pytest.mark.parametrize("C,groups,shape", [(32, 16, (3, 6, 5, 4)), (64, 16, (2, 7, 3, 5)), (16, 16, (2, 8, 8, 8)), (48, 4, (2, 5, 6))])
def test_gn_act_vs_torch_fp64(C, groups, shape):
    """relu(group_norm(z)) and its gradients (z, gamma, beta) against torch in fp64; odd voxel counts, 2-D and 3-D."""
    from arco_amd import ops
    g = torch.Generator().manual_seed(C + groups)
    z = (torch.randn(shape[0], C, *shape[1:], generator=g) * 1.5 + 0.3).cuda()
    z = z.contiguous(memory_format=torch.channels_last_3d if z.dim() == 5 else torch.channels_last).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda().requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).cuda().requires_grad_(True)
    w = torch.randn(shape[0], C, *shape[1:], generator=g).cuda()
    a = ops.gn_act(z, gamma, beta, groups)
    (a * w).sum().backward()
    zd, gd, bd = (t.detach().double().cpu().requires_grad_(True) for t in (z, gamma, beta))
    ref = torch.relu(F.group_norm(zd, groups, gd, bd, 1e-5))
    (ref * w.double().cpu()).sum().backward()
    np.testing.assert_allclose(a.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=2e-5)
    for got, r, nm in ((z.grad, zd.grad, "dz"), (gamma.grad, gd.grad, "dgamma"), (beta.grad, bd.grad, "dbeta")):
        err = float((got.cpu().double() - r).abs().max())
        assert err <= 2e-4 * float(r.abs().max()), (nm, err)
