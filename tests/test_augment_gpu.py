"""Mixing strategies of the unlabeled stream: arco_amd.augment on the HIP path vs the reference
functions' outputs (tests/golden/g7_mix.npz, oracle/gen_golden.py g7) - bit-exact tensors, same RNG consumption."""
import os
import random

import numpy as np
import pytest
import torch

import arco_oracle as orc
import fixture_inputs as fx

pytestmark = pytest.mark.gpu
G7 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_mix.npz"))


def _seed(s):
    random.seed(s); np.random.seed(s); torch.manual_seed(s)


@pytest.mark.parametrize("tag", sorted(fx.MIX_CASES))
def test_generate_unsup_data_matches_reference(tag):
    from arco_amd import augment
    mode, b, c, spatial, n_cls, seed = fx.MIX_CASES[tag]
    data, target, logits = (t.cuda() for t in fx.mix_inputs(seed, b, c, spatial, n_cls))
    _seed(seed + 1)
    fn = augment.generate_unsup_data if len(spatial) == 2 else augment.generate_unsup_data_3d
    nd, nt, nl = fn(data, target, logits, mode=mode)
    assert nt.dtype == torch.int64 and nd.shape == data.shape and nt.shape == target.shape
    np.testing.assert_array_equal(nd.cpu().numpy(), G7[f"{tag}_data"])
    np.testing.assert_array_equal(nt.cpu().numpy(), G7[f"{tag}_target"].astype(np.int64))
    np.testing.assert_array_equal(nl.cpu().numpy(), G7[f"{tag}_logits"])
    np.testing.assert_array_equal(target.cpu().numpy(), G7[f"{tag}_target_after"].astype(np.int64))
    probe = (int(torch.randint(1 << 30, (1,))), float(np.random.uniform()), random.random())
    np.testing.assert_array_equal(np.array(probe, dtype=np.float64), G7[f"{tag}_probe"])


def test_masks_match_oracle():
    from arco_amd import augment
    for size in ([256, 256], [37, 53], [112, 112, 80]):
        _seed(5)
        exp = orc.cutout_mask(list(size))
        _seed(5)
        got = augment.generate_cutout_mask(size) if len(size) == 2 else augment.generate_cutout_mask_3d(size)
        np.testing.assert_array_equal(got.numpy(), exp)
        assert abs(float((got == 0).sum()) - (size[0] * size[1] / 2) * (10 if len(size) == 3 else 1)) <= size[1] * (10 if len(size) == 3 else 1)
    rs = np.random.RandomState(1)
    lab = torch.from_numpy(rs.randint(0, 7, size=(40, 50)).astype(np.int64))
    _seed(9)
    exp = orc.class_mask(lab.numpy())
    _seed(9)
    got = augment.generate_class_mask(lab.cuda())
    np.testing.assert_array_equal(got.cpu().numpy(), exp)
    assert 0 < exp.mean() < 1


def test_full_size_cutmix_properties():
    """BASELINE configs[1] size: every output pixel is the pixel of image i or of image i+1, the replaced region is one
    rectangle of half the image, and labels / confidences move with the pixels."""
    from arco_amd import augment
    b, H, W = 8, 256, 256
    data, target, logits = (t.cuda() for t in fx.mix_inputs(3, b, 1, (H, W), 4))
    _seed(11)
    nd, nt, nl = augment.generate_unsup_data(data, target, logits, mode="cutmix")
    own = nd == data
    other = nd == torch.roll(data, -1, 0)
    assert bool((own | other).all())
    for i in range(b):
        repl = (~own[i, 0]).cpu().numpy()
        ys, xs = np.where(repl)
        assert repl[ys.min():ys.max() + 1, xs.min():xs.max() + 1].mean() > 0.999      # one solid rectangle
        assert abs(repl.sum() - H * W / 2) <= W
        r = torch.from_numpy(repl).cuda()
        assert bool((nt[i][r] == target[(i + 1) % b][r]).all()) and bool((nl[i][~r] == logits[i][~r]).all())


@pytest.mark.parametrize("tag", list(fx.JITTER_CASES))
def test_jitter_blur_kernels_vs_pillow(tag):
    """arco_jitter_blur (Pillow's 8-bit integer arithmetic on the GPU) vs Pillow's own output (g12): bit exact."""
    import os
    from arco_amd import augment
    g12 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g12_jitter.npz"))
    C, H, W, seed, order, factors, sigma = fx.JITTER_CASES[tag]
    x = torch.from_numpy(fx.jitter_image(seed, C, H, W)).unsqueeze(0)
    # a batch of three images with different parameters: the case, an untouched image, the case again
    data = torch.cat((x, x.flip(-1), x)).cuda()
    p = dict(order=list(order) if order is not None else None, factors=factors, sigma=sigma)
    out = augment.jitter_blur(data, [p, dict(order=None, factors=None, sigma=None), p])
    got = torch.round(out * 255).to(torch.uint8).cpu().numpy()
    np.testing.assert_array_equal(got[0], g12[tag])
    np.testing.assert_array_equal(got[2], g12[tag])
    np.testing.assert_array_equal(got[1], (x.flip(-1)[0] * 255).to(torch.uint8).numpy())     # plain 8-bit round trip
    assert torch.equal(out * 255, torch.round(out * 255))                                     # exact k / 255 values


def test_batch_transform_draws_and_outputs_match_oracle():
    """augment.batch_transform vs the oracle's restatement (cpu_step.batch_transform): same consumption of the python /
    torch CPU generators, identical 8-bit images and confidences, AdvMorph on the same velocity field."""
    import random
    import cpu_step
    from arco_amd import augment
    from arco_amd.adv_morph import AdvMorph
    rs = np.random.RandomState(3)
    data = torch.from_numpy(rs.uniform(size=(6, 1, 64, 64)).astype(np.float32))
    label = torch.from_numpy(rs.randint(-1, 4, size=(6, 64, 64)).astype(np.int64))
    logits = torch.from_numpy(rs.uniform(size=(6, 64, 64)).astype(np.float32))
    vel = lambda B, h, w: torch.from_numpy(np.random.RandomState(11).uniform(-1, 1, size=(B, 2, h, w)).astype(np.float32))
    real = AdvMorph.init_velocity
    AdvMorph.init_velocity = lambda self, batch_size, height, width, use_zero=False: self.unit_normalize(vel(batch_size, height, width).cuda())
    try:
        n_morph = 0
        for seed in range(6):                         # several seeds: jitter / blur / morph on and off
            for aug in (True, False):
                random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
                d_o, l_o, g_o = cpu_step.batch_transform(data, label, logits, aug, vel)
                probe_o = (random.random(), float(torch.rand(1)))
                random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
                d_g, l_g, g_g = augment.batch_transform(data.cuda(), label.cuda(), logits.cuda(), [64, 64], (1.0, 1.0), aug)
                probe_g = (random.random(), float(torch.rand(1)))
                assert probe_o == probe_g, (seed, aug)                       # same generator consumption
                assert torch.equal(l_g.cpu(), l_o) and torch.equal(g_g.cpu(), g_o)
                morphed = not torch.equal(d_o * 255, torch.round(d_o * 255))
                n_morph += int(morphed)
                if morphed:
                    np.testing.assert_allclose(d_g.cpu().numpy(), d_o.numpy(), atol=2e-4)
                else:
                    assert torch.equal(d_g.cpu(), d_o), (seed, aug)
        assert n_morph >= 1
    finally:
        AdvMorph.init_velocity = real
