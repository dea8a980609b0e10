"""The rest of the trainers' import surface against the reference's own code (tests/golden/g16_boundary.npz, made by
oracle/gen_golden.py g16): utils.ramps, LocalConLoss / SupConLoss, randomGeneratorWithLogits on the CPU; the HIP-backed
utils.losses.DiceLoss and the loss helpers' label_onehot under -m gpu."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g16_boundary.npz"), allow_pickle=False)


def test_ramps_match_reference():
    from arco_amd.utils import ramps
    cur = G["ramp_cur"]
    np.testing.assert_allclose([ramps.sigmoid_rampup(c, 200.0) for c in cur], G["ramp_sigmoid_200"], rtol=1e-15)
    np.testing.assert_allclose([ramps.sigmoid_rampup(c, 0) for c in cur], G["ramp_sigmoid_0"], rtol=0)
    np.testing.assert_allclose([ramps.linear_rampup(c, 200.0) for c in cur[1:]], G["ramp_linear_200"], rtol=1e-15)
    np.testing.assert_allclose([ramps.cosine_rampdown(c, 250.0) for c in cur[1:]], G["ramp_cosine_250"], rtol=1e-15, atol=1e-17)
    np.testing.assert_allclose([ramps.exp_rampup(100.0)(c) for c in cur], G["ramp_exp_100"], rtol=1e-15)
    assert isinstance(ramps.sigmoid_rampup(3, 10), float)


@pytest.mark.parametrize("tag,kw,use_lab", [("s4_lab", dict(temperature=0.7, stride=4), True), ("s4_nolab", dict(temperature=0.7, stride=4), False),
                                            ("s8_lab", dict(temperature=0.5, stride=8), True)])
def test_local_con_loss_matches_reference(tag, kw, use_lab):
    from arco_amd.loss_helper_3d import LocalConLoss
    f = torch.from_numpy(G["lcl_f"]).requires_grad_(True)
    lab = torch.from_numpy(G["lcl_lab"])
    crit = LocalConLoss(**kw)
    loss = crit(f, lab) if use_lab else crit(f)
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(G[f"lcl_{tag}"]), rtol=1e-5)
    ref = G[f"lcl_{tag}_grad"]
    np.testing.assert_allclose(f.grad.numpy(), ref, rtol=1e-4, atol=1e-6 * float(np.abs(ref).max()))


def test_local_con_loss_zero_labels_and_signature():
    from arco_amd.loss_helper_3d import LocalConLoss, SupConLoss
    crit = LocalConLoss()
    assert (crit.temp, crit.stride) == (0.7, 4) and isinstance(crit.supconloss, SupConLoss)
    z = crit(torch.from_numpy(G["lcl_f"]), torch.zeros_like(torch.from_numpy(G["lcl_lab"])))
    assert float(z) == float(G["lcl_zero_labels"]) == 0.0
    with pytest.raises(ValueError):
        SupConLoss()(torch.zeros(4, 4))


@pytest.mark.parametrize("tag,size", [("same", [32, 32]), ("zoom", [48, 40])])
def test_random_generator_with_logits_matches_reference(tag, size):
    from arco_amd.augment import randomGeneratorWithLogits
    a, b, c = randomGeneratorWithLogits(torch.from_numpy(G["rg_img"]), torch.from_numpy(G["rg_lab"]), torch.from_numpy(G["rg_logit"]),
                                        output_size=size)
    assert a.dtype == torch.float32 and b.dtype == torch.int64
    assert np.array_equal(a.numpy(), G[f"rg_{tag}_img"]) and np.array_equal(b.numpy(), G[f"rg_{tag}_lab"])
    assert np.array_equal(c.numpy(), G[f"rg_{tag}_logit"])


def test_augment_3d_batch_transform_is_the_identity():
    from arco_amd.augment_3d import batch_transform, transform
    d, l, g = torch.rand(2, 1, 4, 5, 6), torch.zeros(2, 4, 5, 6, dtype=torch.long), torch.rand(2, 4, 5, 6)
    for aug in (False, True):
        out = batch_transform(d, l, logits=g, scale_size=(1.0, 1.0), apply_augmentation=aug)
        assert all(torch.equal(x, y) for x, y in zip(out, (d, l, g)))
    assert len(transform(d[0], l[0])) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("tag,C", [("2d", 4), ("3d", 2), ("c19", 19)])
@pytest.mark.parametrize("mode", ["plain", "w", "sm"])
def test_dice_loss_class_matches_reference(tag, C, mode):
    """utils.losses.DiceLoss(C)(scores, target[, weight][, softmax]) - value and gradient w.r.t. the scores' pre-softmax
    input - against the reference class; also on this package's channels-last network-output layout."""
    from arco_amd.utils.losses import DiceLoss
    kw = {"plain": {}, "w": dict(weight=[0.5 + 0.25 * i for i in range(C)]), "sm": dict(softmax=True)}[mode]
    lab = torch.from_numpy(G[f"dice_{tag}_lab"]).cuda()
    for channels_last in (False, True):
        x = torch.from_numpy(G[f"dice_{tag}_x"]).cuda()
        if channels_last:
            x = x.movedim(1, -1).contiguous().movedim(-1, 1)
        x.requires_grad_(True)
        inp = x if mode == "sm" else torch.softmax(x, dim=1)
        loss = DiceLoss(C)(inp, lab, **kw)
        loss.backward()
        np.testing.assert_allclose(float(loss.detach()), float(G[f"dice_{tag}_{mode}"]), rtol=1e-5)
        ref = G[f"dice_{tag}_{mode}_grad"]
        np.testing.assert_allclose(x.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * float(np.abs(ref).max()))


@pytest.mark.gpu
def test_dice_loss_rejects_cpu_tensors_and_shape_mismatch():
    from arco_amd.utils.losses import DiceLoss
    with pytest.raises(RuntimeError):
        DiceLoss(4)(torch.rand(1, 4, 8, 8), torch.zeros(1, 1, 8, 8, dtype=torch.long))
    with pytest.raises(AssertionError):
        DiceLoss(4)(torch.rand(1, 4, 8, 8).cuda(), torch.zeros(1, 1, 8, 7, dtype=torch.long).cuda())


@pytest.mark.gpu
def test_loss_helper_label_onehot_matches_reference():
    from arco_amd.loss_helper import label_onehot as lo5
    from arco_amd.loss_helper_3d import label_onehot
    out = label_onehot(torch.from_numpy(G["lh_onehot_in"]).cuda(), 4)
    assert out.dtype == torch.float32 and np.array_equal(out.cpu().numpy(), G["lh_onehot"]) and lo5 is label_onehot
