"""Equivariance loss on the HIP path vs golden vectors produced by the reference's tps modules
(oracle/gen_golden.py g5): RandTPS grid for a fixed seed (and how much of the torch / numpy / python generators a
reset consumes), grid_sample, masked-KL loss and its gradient."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g5():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "g5_eqv.npz"))


def seed_all(s):
    random.seed(s); np.random.seed(s); torch.manual_seed(s)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_rand_tps_and_eqv_loss_vs_reference(g5, tag):
    from arco_amd import glue
    from arco_amd.tps import RandTPS
    B, W, H, sigma, seed = g5[f"{tag}_cfg"]
    B, W, H, seed = int(B), int(W), int(H), int(seed)
    seed_all(seed)
    tps = RandTPS(W, H, batch_size=B, sigma=float(sigma), border_padding=False, random_mirror=True,
                  random_scale=(0.8, 1.2), mode='affine')
    np.testing.assert_allclose(tps.grid.cpu().numpy(), g5[f"{tag}_grid_init"], rtol=1e-4, atol=3e-5)   # 28-term sums with cancellation: ~0.004 px at 256 px
    probe = (int(torch.randint(1 << 30, (1,))), float(np.random.uniform()), random.random())
    np.testing.assert_allclose(np.array(probe), g5[f"{tag}_probe_init"], rtol=0, atol=0)     # same generator consumption
    seed_all(seed + 100)
    tps.reset_control_points()
    np.testing.assert_allclose(tps.grid.cpu().numpy(), g5[f"{tag}_grid"], rtol=1e-4, atol=3e-5)   # 28-term sums with cancellation: ~0.004 px at 256 px
    probe = (int(torch.randint(1 << 30, (1,))), float(np.random.uniform()), random.random())
    np.testing.assert_allclose(np.array(probe), g5[f"{tag}_probe"], rtol=0, atol=0)
    cu = lambda k: torch.from_numpy(g5[f"{tag}_{k}"]).cuda()
    # end to end on the product's own grid: white-noise images are the worst case (|d img / d px| ~ 1, grid error ~1e-5 * W)
    np.testing.assert_allclose(tps(cu("img")).cpu().numpy(), g5[f"{tag}_images_tps"], atol=2e-3)
    # the sampling kernel itself, on the reference's grid
    tps.grid.copy_(cu("grid"))
    np.testing.assert_allclose(tps(cu("img")).cpu().numpy(), g5[f"{tag}_images_tps"], atol=2e-5)
    mask = glue.eqv_mask(cu("labels"), cu("logits"), 0.7)
    mask_tps = tps(mask, padding_mode='zeros')
    np.testing.assert_allclose(mask_tps.cpu().numpy(), g5[f"{tag}_mask_tps"], atol=2e-5)
    org = tps(cu("pred_all"), padding_mode='zeros')
    np.testing.assert_allclose(org.cpu().numpy(), g5[f"{tag}_pred_tps_org"], atol=5e-5)
    # the loss on the reference's own warped tensors (isolates the loss kernels from grid rounding)
    p = cu("pred_tps").requires_grad_(True)
    loss = glue.eqv_loss(p, cu("pred_tps_org"), cu("mask_tps"))
    (3.0 * loss).backward()
    np.testing.assert_allclose(loss.item(), float(g5[f"{tag}_loss"]), rtol=1e-5)
    np.testing.assert_allclose(p.grad.cpu().numpy() / 3.0, g5[f"{tag}_grad"], rtol=1e-4, atol=1e-9)
    # and end to end on the product's own warps
    p2 = cu("pred_tps").requires_grad_(True)
    loss2 = glue.eqv_loss(p2, org, mask_tps)
    np.testing.assert_allclose(loss2.item(), float(g5[f"{tag}_loss"]), rtol=1e-3)


def test_grid_sample_border_and_identity():
    """identity grid reproduces the input; border padding clamps (F.grid_sample semantics, align_corners=True)."""
    import torch.nn.functional as F
    from arco_amd.tps import RandTPS
    seed_all(0)
    tps = RandTPS(12, 10, batch_size=2, sigma=0.0, random_mirror=False, random_scale=(1.0, 1.0))
    x = torch.rand(2, 3, 10, 12, device="cuda")
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, 10), torch.linspace(-1, 1, 12), indexing="ij")
    ident = torch.stack((xs, ys), -1).unsqueeze(0).repeat(2, 1, 1, 1).cuda()
    tps.grid.copy_(ident)
    np.testing.assert_allclose(tps(x).cpu().numpy(), x.cpu().numpy(), atol=1e-5)
    tps.grid.copy_(ident * 1.3 + 0.1)
    for pm in ("zeros", "border"):
        ref = F.grid_sample(x.cpu(), tps.grid.cpu(), mode="bilinear", padding_mode=pm, align_corners=True)
        np.testing.assert_allclose(tps(x, padding_mode=pm).cpu().numpy(), ref.numpy(), atol=2e-5)


def test_rand_tps_3d_slicewise_vs_reference(g5):
    """Volume variant (tps/rand_tps_3d.py): same grid for the seed, the warp applied to every slice x[..., z]."""
    from arco_amd.tps.rand_tps_3d import RandTPS
    seed_all(21)
    tps = RandTPS(12, 12, 6, batch_size=2, sigma=0.02, border_padding=False, random_mirror=True, random_scale=(0.8, 1.2),
                  mode='affine')
    seed_all(121)
    tps.reset_control_points()
    np.testing.assert_allclose(tps.grid.cpu().numpy(), g5["v_grid"], rtol=1e-4, atol=3e-5)
    probe = (int(torch.randint(1 << 30, (1,))), float(np.random.uniform()), random.random())
    np.testing.assert_allclose(np.array(probe), g5["v_probe"], rtol=0, atol=0)
    tps.grid.copy_(torch.from_numpy(g5["v_grid"]).cuda())
    got = tps(torch.from_numpy(g5["v_vol"]).cuda(), padding_mode='zeros')
    assert got.shape == (2, 3, 12, 12, 6)
    np.testing.assert_allclose(got.cpu().numpy(), g5["v_vol_tps"], atol=2e-5)


def test_eqv_loss_edge_cases():
    """all-zero mask -> loss 0 and zero gradient (denominator 1e-7, train_arco_2d.py:422); identical predictions ->
    zero KL; 19 classes; the value against a plain torch evaluation."""
    import torch.nn.functional as F
    from arco_amd import glue
    torch.manual_seed(0)
    B, C, H, W = 3, 19, 12, 20
    p = torch.randn(B, C, H, W, device="cuda").requires_grad_(True)
    q = torch.randn(B, C, H, W, device="cuda")
    z = torch.zeros(B, 1, H, W, device="cuda")
    loss = glue.eqv_loss(p, q, z)
    loss.backward()
    assert float(loss) == 0.0 and float(p.grad.abs().max()) == 0.0
    m = (torch.rand(B, 1, H, W, device="cuda") > 0.4).float()
    m[1] = 0                                                   # one image without any valid pixel
    p2 = torch.randn(B, C, H, W, device="cuda").requires_grad_(True)
    got = glue.eqv_loss(p2, q, m)
    got.backward()
    pr = p2.detach().cpu().requires_grad_(True)
    kl = F.kl_div(F.log_softmax(pr, 1), F.softmax(q.cpu(), 1), reduction='none')
    ref = ((kl * m.cpu()).flatten(1).sum(1) / (m.cpu().flatten(1).sum(1) + 1e-7)).mean()
    ref.backward()
    np.testing.assert_allclose(float(got), float(ref), rtol=1e-5)
    np.testing.assert_allclose(p2.grad.cpu().numpy(), pr.grad.numpy(), rtol=1e-4, atol=1e-9)
    same = glue.eqv_loss(q.clone().requires_grad_(True), q, m)
    assert abs(float(same)) < 1e-6
