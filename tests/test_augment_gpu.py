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
