"""AdvMorph on the HIP path (arco_amd/adv_morph.py) vs outputs of the reference class (tests/golden/g10_morph.npz,
oracle/gen_golden.py g10) and its invariants (-m gpu)."""
import os

import numpy as np
import pytest
import torch

import fixture_inputs as fx

pytestmark = pytest.mark.gpu
G10 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g10_morph.npz"))


def _make(B, C, H, W):
    from arco_amd.adv_morph import AdvMorph
    return AdvMorph(config_dict={'epsilon': 1.5, 'xi': 0.5, 'data_size': [B, C, H, W], 'vector_size': [W // 8, W // 8],
                                 'interpolator_mode': 'bilinear'}, debug=False, use_gpu=True)


@pytest.mark.parametrize("case", fx.MORPH_CASES)
def test_adv_morph_vs_reference_golden(case):
    tag, B, C, H, W, seed = case
    data, _ = fx.morph_inputs(seed, B, C, H, W)
    aug = _make(B, C, H, W)
    aug.set_parameters(torch.from_numpy(G10[f"{tag}_param"]).cuda())
    grid, disp = aug.get_deformation_displacement_field(duv=aug.epsilon * aug.param)
    warped = aug.forward(data.cuda())
    st = fx.MORPH_STRIDE if H * W > 10000 else 1
    np.testing.assert_allclose(grid.cpu().numpy()[:, :, ::st, ::st], G10[f"{tag}_grid"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(warped.cpu().numpy()[:, :, ::st, ::st], G10[f"{tag}_warped"], rtol=0, atol=2e-4)
    assert abs(float(disp.abs().max()) - float(G10[f"{tag}_maxdisp"][0])) < 1e-4
    assert float(grid.abs().max()) <= 1.0                            # clamped (adv_morph.py:530-531)


def test_adv_morph_draws_on_the_device_generator_and_zero_velocity_is_the_identity():
    B, C, H, W = 2, 1, 64, 64
    torch.manual_seed(5)
    state = torch.get_rng_state()
    aug = _make(B, C, H, W)
    p = aug.init_parameters()                                        # torch.rand(device=cuda): device generator
    assert torch.equal(torch.get_rng_state(), state)                 # the CPU generator (sampler sequence) is untouched
    np.testing.assert_allclose(p.reshape(B, -1).norm(dim=1).cpu().numpy(), 1.0, rtol=1e-5)      # unit_normalize
    x = torch.rand(B, C, H, W, device="cuda")
    y = aug.forward(x)
    assert y.shape == x.shape and float((y - x).abs().max()) > 1e-3  # a real warp
    aug.set_parameters(torch.zeros_like(p))
    # zero velocity: the identity up to the float error of the eight self-compositions (the reference's own output differs
    # from its input by ~1e-3 on a noise image: grid error 3e-5)
    np.testing.assert_allclose(aug.forward(x).cpu().numpy(), x.cpu().numpy(), atol=5e-3)
