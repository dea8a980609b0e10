"""Evaluation path: eval-mode inference, zoom round trip and Dice / Jaccard on the HIP path vs
the CPU oracle (oracle/arco_oracle.py: test_single_volume, dice_jaccard) - test_2D.py:52-92 of the reference."""
import numpy as np
import pytest
import torch

import arco_oracle as orc
import fixture_inputs as fx

pytestmark = pytest.mark.gpu


def _volume(rs, S, x, y, C):
    img = rs.uniform(size=(S, x, y)).astype(np.float32)
    lab = np.stack([fx.blob_labels(rs, 1, (x, y), C)[0] for _ in range(S)]).astype(np.int64)
    return img, lab


def test_overlap_counts_match_numpy():
    from arco_amd import test_2D
    rs = np.random.RandomState(0)
    for n, C in ((1000, 4), (70001, 19), (5, 2)):
        p = rs.randint(0, C, size=n); g = rs.randint(0, C, size=n)
        got = test_2D.overlap_counts(torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda(), C).cpu().numpy()
        exp = np.array([[(p == c).sum(), (g == c).sum(), ((p == c) & (g == c)).sum()] for c in range(C)])
        np.testing.assert_array_equal(got, exp)


def test_single_volume_matches_oracle():
    """Non-square slices (zoom both ways), eval-mode BatchNorm with non-trivial running statistics."""
    from arco_amd import test_2D
    from arco_amd.networks.unetWithArgs import UNet
    C = 4
    sd = fx.unet_state(21, 1, C)
    rs = np.random.RandomState(4)
    for k in list(sd):                      # make the running statistics differ from (0, 1)
        if k.endswith("running_mean"):
            sd[k] = torch.from_numpy(rs.normal(scale=0.1, size=tuple(sd[k].shape)).astype(np.float32))
        if k.endswith("running_var"):
            sd[k] = torch.from_numpy(rs.uniform(0.5, 1.5, size=tuple(sd[k].shape)).astype(np.float32))
    net = UNet(1, C).cuda()
    net.load_state_dict(sd, strict=True)
    img, lab = _volume(rs, 5, 48, 80, C)
    patch = (64, 64)
    exp_metrics, exp_pred = orc.test_single_volume(img, lab, sd, C, patch)
    got_pred = test_2D.predict_volume(img, net, patch)
    agree = float((got_pred == exp_pred).mean())
    assert agree > 0.999, agree             # argmax flips only where two logits tie to fp32 rounding
    got = test_2D.test_single_volume(img, lab, net, C, patch)
    assert len(got) == C - 1
    for (d, j, _, _), (de, je) in zip(got, exp_metrics):
        assert abs(d - de) < 5e-3 and abs(j - je) < 5e-3, ((d, j), (de, je))
    # the reference's empty-set conventions
    assert test_2D.calculate_metric_percase(np.zeros((4, 4)), np.zeros((4, 4)))[:2] == (0.0, 0.0)
    assert test_2D.calculate_metric_percase(np.ones((4, 4)), np.zeros((4, 4)))[:2] == (1.0, 1.0)
    m = np.zeros((4, 4)); m[:2] = 1; g = np.zeros((4, 4)); g[1:3] = 1
    d, j, _, _ = test_2D.calculate_metric_percase(m, g)
    assert abs(d - 0.5) < 1e-12 and abs(j - 1.0 / 3.0) < 1e-12
    assert net.training                      # predict_volume restores the mode it found


@pytest.mark.parametrize("tag", sorted(fx.EVAL3D_CASES))
def test_single_case_3d_matches_reference(tag):
    """3-D sliding-window inference on the HIP path vs the reference's outputs (g6) and the oracle."""
    import os
    from arco_amd import test_util
    from arco_amd.networks.vnetWithArgs import VNet
    g6 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g6_eval3d.npz"))
    shape, patch, sxy, sz, C, nf, seed = fx.EVAL3D_CASES[tag]
    sd = fx.randomize_running_stats(fx.vnet_state(seed, 1, C, nf), seed + 1)
    net = VNet(n_channels=1, n_classes=C, n_filters=nf, normalization='batchnorm', has_dropout=False).cuda()
    net.load_state_dict(sd, strict=True)
    image = fx.eval3d_volume(seed + 2, shape)
    exp_score, exp_label = g6[f"{tag}_score"], g6[f"{tag}_label"].astype(np.int64)
    label, score = test_util.test_single_case(net, image, sxy, sz, patch, num_classes=C)
    assert label.shape == tuple(shape) and score.shape == (C, *shape) and label.dtype == np.int64
    np.testing.assert_allclose(score, exp_score, rtol=0, atol=1e-4)
    top2 = np.sort(exp_score, axis=0)
    decided = (top2[-1] - top2[-2]) > 2e-4                       # arg-max may flip only where two scores tie to rounding
    np.testing.assert_array_equal(label[decided], exp_label[decided])
    assert decided.mean() > 0.99
    np.testing.assert_array_equal(label, np.argmax(score, axis=0))   # the finalize kernel's arg-max is numpy's (first maximum)
    # windows sharing a forward vs one forward per window: same sums in the same order
    label1, score1 = test_util.test_single_case(net, image, sxy, sz, patch, num_classes=C, batch=1)
    np.testing.assert_allclose(score1, score, rtol=0, atol=2e-6)
    assert net.training


def test_all_case_3d_metrics():
    from arco_amd import test_util
    from arco_amd.networks.vnetWithArgs import VNet
    shape, patch, sxy, sz, C, nf, seed = fx.EVAL3D_CASES["ragged"]
    net = VNet(n_channels=1, n_classes=C, n_filters=nf, normalization='batchnorm', has_dropout=False).cuda()
    net.load_state_dict(fx.randomize_running_stats(fx.vnet_state(seed, 1, C, nf), seed + 1), strict=True)
    image = fx.eval3d_volume(seed + 2, shape)
    pred, _ = test_util.test_single_case(net, image, sxy, sz, patch, num_classes=C)
    avg = test_util.test_all_case(net, [(image, pred), (image, pred)], C, patch_size=patch, stride_xy=sxy, stride_z=sz)
    np.testing.assert_allclose(avg, [1.0, 1.0, 0.0, 0.0])        # a prediction scored against itself
    other = np.roll(pred, 2, axis=0)
    d, j, hd, asd = test_util.calculate_metric_percase(pred, other)
    hd_o, asd_o = orc.surface_metrics(pred, other)
    assert 0 < d < 1 and 0 < j < d and abs(hd - hd_o) < 1e-9 and abs(asd - asd_o) < 1e-9
    cc = test_util.getLargestCC(pred)
    assert cc.dtype == bool and 0 < cc.sum() <= (pred > 0).sum()


@pytest.mark.parametrize("tag", list(fx.EVAL2D_CASES))
def test_single_volume_matches_reference_function(tag):
    """arco_amd.test_2D.test_single_volume vs the prediction volume of the reference's own test_single_volume
    (tests/golden/g11_eval2d.npz: pulled out of test_2D.py:67-103 and run on the reference U-Net in eval mode)."""
    import os
    from arco_amd import test_2D
    from arco_amd.networks.unetWithArgs import UNet
    g11 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_eval2d.npz"))
    shape, C, seed = fx.EVAL2D_CASES[tag]
    image, label = fx.eval2d_volume(seed, shape, C)
    net = UNet(in_chns=1, class_num=C).cuda()
    net.load_state_dict(fx.randomize_running_stats(fx.unet_state(seed, 1, C), seed + 1))
    pred = test_2D.predict_volume(image, net)
    exp = g11[f"{tag}_pred"].astype(np.int64)
    assert pred.shape == exp.shape and float((pred != exp).mean()) < 1e-3, float((pred != exp).mean())
    got = test_2D.test_single_volume(image, label, net, C)
    for c in range(1, C):                          # Dice / Jaccard of the reference's prediction vs the product's
        p, g = exp == c, label == c
        inter = float(np.logical_and(p, g).sum())
        if p.sum() > 0 and g.sum() > 0:
            np.testing.assert_allclose(got[c - 1][0], 2 * inter / (p.sum() + g.sum()), atol=5e-3)
            np.testing.assert_allclose(got[c - 1][1], inter / np.logical_or(p, g).sum(), atol=5e-3)
