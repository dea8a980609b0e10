"""Host-side bookkeeping of the step schedule (no GPU work): the BatchNorm running-statistics deferral slots of ops.bn_defer
(train_arco_2d mode 4 postpones the statistics pass's updates into slot 1 and the u half's into slot 0, so that they land in the
reference's order l, cj2_l, u - train_arco_2d.py:310-312) and the workspace sizing of the per-image losses."""
import torch

from arco_amd import _lib as L, ops


def test_bn_defer_slots_are_separate_and_nest():
    assert ops.BN_DEFER is None
    rm, rv = torch.zeros(8), torch.ones(8)
    assert ops._defer_args(rm, rv, 8, 2, 0.1) == (0, None)                 # outside any context: immediate update
    with ops.bn_defer(1):                                                  # slot 0: groups >= 1 deferred
        g0, buf0 = ops._defer_args(rm, rv, 8, 2, 0.1)
        assert g0 == 1 and buf0.numel() == 1 * 2 * 8 + 1
        assert ops._defer_args(rm, rv, 8, 1, 0.1) == (0, None)             # a one-group pass has no group >= 1
        with ops.bn_defer(0, 1):                                           # slot 1: every group deferred
            g1, buf1 = ops._defer_args(rm, rv, 8, 1, 0.1)
            assert g1 == 0 and buf1.numel() == 1 * 2 * 8 + 1 and buf1.data_ptr() != buf0.data_ptr()
            assert ops._defer_args(rm, rv, 8, 1, 0.1)[1].data_ptr() == buf1.data_ptr()     # same layer, same slot: same buffer
        assert ops.BN_DEFER == (1, 0)
        assert ops._defer_args(rm, rv, 8, 2, 0.1)[1].data_ptr() == buf0.data_ptr()
    assert ops.BN_DEFER is None
    key = rm.data_ptr()
    assert key in ops._DEFERRED[0] and key in ops._DEFERRED[1]
    assert ops._DEFERRED[0][key]["n"] == 1 and ops._DEFERRED[1][key]["n"] == 1
    # a different group count for the same layer re-registers the slot's entry (and invalidates its descriptor table)
    with ops.bn_defer(0):
        _, buf2 = ops._defer_args(rm, rv, 8, 2, 0.1)
    assert buf2.numel() == 2 * 2 * 8 + 1 and ops._DEFERRED[0][key]["n"] == 2
    for slot in (0, 1):
        ops._DEFERRED[slot].pop(key)
        ops._DEFER_TABLE.pop(slot, None)


def test_loss_slab_counts():
    """arco_loss_slabs(B): slabs per image of the unsupervised-CE / equivariance partial sums - 64 from 8 images on, more with fewer
    (one 2.5 M-voxel volume on 64 blocks left the chip idle); the Python callers size their workspaces from it."""
    got = {b: int(L.query("arco_loss_slabs", b)) for b in (1, 2, 3, 4, 5, 7, 8, 9, 16, 64)}
    assert got == {1: 512, 2: 256, 3: 256, 4: 128, 5: 128, 7: 128, 8: 64, 9: 64, 16: 64, 64: 64}
    for b, n in got.items():
        assert n * b >= 512 or b >= 8


def test_3x3x3_launch_shapes_of_the_la_levels():
    """Which kernel and tile shape the library picks for the V-Net's 3x3x3 convolutions at the LA patch (vnetWithArgs.py:5-31;
    csrc/conv3d_fl.hip: the fitted launch cost model on a 256-CU device - the default without a GPU), and how many BatchNorm
    partial-sum slabs per channel the caller has to provide.  ids: 9.45e6 conv3d_rw16_kernel, 9.27e6 + A_T*1e3 + BN conv3d_fl_kernel
    (per-step rendezvous), 9.29e6 + ... conv3d_fc_kernel (per-chunk rendezvous); slabs = 4 per flat tile of 64 A_T positions.
    (The query describes a launch by its plane count only: the choice cannot depend on the depth.)"""
    levels = ((16, (112, 112, 80)), (32, (56, 56, 40)), (64, (28, 28, 20)), (128, (14, 14, 10)), (256, (7, 7, 5)))
    want = {2: (9450016, 9293032, 9295032, 9292032, 9291032), 4: (9450016, 9295032, 9295032, 9293032, 9291032)}
    for nv, ids in want.items():
        for (c, (d, h, w)), cfg in zip(levels, ids):
            assert int(L.query("arco_conv_config_mma", 27, nv * d, h, w, c, c, c, 3)) == cfg, (nv, c)
            nmb = int(L.query("arco_conv_mblocks_mma", 27, nv * d, h, w, c, c, c, 1, 3))
            if 9270000 <= cfg < 9300000:
                a_t = cfg % 10000 // 1000
                assert nmb == 4 * nv * d * -(-(h * (w + 2)) // (64 * a_t)), (nv, c, nmb)
    # shapes the pipelined kernels do not take stay on igemm_kernel: planes wider than 61 pixels, N not a multiple of 32
    assert int(L.query("arco_conv_config_mma", 27, 64, 80, 80, 32, 32, 32, 3)) < 9270000
    assert not 9270000 <= int(L.query("arco_conv_config_mma", 27, 64, 28, 20, 32, 48, 32, 3)) < 9300000


def test_draw_boxes_consumes_the_generator_like_the_inline_mixing_loop():
    """augment.draw_boxes (round 6: the boxes of cutout / cutmix drawn apart from the mixing, so that the image side can run before the
    teacher's pseudo-labels exist) must leave numpy's generator where the reference's per-image generate_cutout_mask(_3d) calls leave it
    (augment.py:229-245, 284-313) and return those boxes - 2-D and 3-D."""
    import numpy as np
    from arco_amd import augment
    for sp in ((256, 256), (112, 112, 80), (64, 48)):
        np.random.seed(17)
        desc = augment.draw_boxes(5, sp)
        after = np.random.randint(0, 1 << 30)
        np.random.seed(17)
        ref = [augment._cutout_box(list(sp), ratio=2) for _ in range(5)]
        assert np.random.randint(0, 1 << 30) == after
        assert desc.shape == (5, 8) and desc.dtype == np.int32
        for i, box in enumerate(ref):
            assert list(desc[i, :len(box)]) == box and desc[i, 5] == (box[5] if len(box) == 6 else 1)
