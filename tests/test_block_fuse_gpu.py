"""Consumer-side activation (round 6): a ConvBlock's first BatchNorm + LeakyReLU + Dropout applied in the loader of the second
convolution (arco_conv3d_fwd_pro) and of its weight gradient (arco_conv3d_wgrad_pro) - unetWithArgs.py:31-47.

The loaders evaluate arco_bn_act_fwd's arithmetic operation for operation, so every comparison here is BIT-EXACT against the
two-pass route (BN-apply pass writing the activation, plain convolution reading it): kernel by kernel on every pipelined kernel
instantiation the U-Net uses, block by block, and over the whole U-Net forward + backward (-m gpu)."""
import numpy as np
import pytest
import torch

import fixture_inputs as fx

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32)).cuda()


# (Cin of the consumer = channels of z1, Cout, images, H, W): one shape per kernel instantiation of conv_sp.hip's dispatch
SHAPES = [
    (16, 16, 8, 128, 128),      # conv3x3_rw_kernel<8,1>
    (32, 16, 8, 128, 128),      # conv3x3_rw_kernel<8,1>, two chunks
    (32, 32, 4, 128, 128),      # conv3x3_rw_kernel<4,2>
    (64, 64, 4, 128, 128),      # conv3x3_sp_kernel<4,4>
    (128, 128, 16, 32, 32),     # conv3x3_sp_kernel<2,4>
    (256, 256, 16, 16, 16),     # conv3x3_sp_kernel<1,4>
    (64, 32, 8, 64, 64),        # conv3x3_sp_kernel<*,2>
]


@pytest.mark.parametrize("k,n,nb,h,w", SHAPES)
@pytest.mark.parametrize("groups,p,salted", [(1, 0.0, False), (2, 0.3, False), (2, 0.1, True)])
def test_loader_activation_is_bit_identical_to_the_apply_pass(k, n, nb, h, w, groups, p, salted):
    from arco_amd import ops, _lib as L
    from arco_amd._contrast import rows_view
    assert ops.CONV_MMA == 3
    if not ops.pro_ok(9, nb, 1, h, w, k, n, k, groups):
        pytest.skip("shape not taken by the pipelined kernels on this device")
    z = ops.new_act(nb, k, h, w, "cuda")
    z.copy_(_rand((nb, k, h, w), 1, 2.0))
    mean, istd = _rand((groups * k,), 2, 0.3), _rand((groups * k,), 3).abs() + 0.5
    gamma, beta = _rand((k,), 4), _rand((k,), 5, 0.2)
    wt = _rand((n, k, 3, 3), 6, 0.1)
    bias = _rand((n,), 7)
    seed = 0x1234567887654321
    salt = torch.tensor([5], dtype=torch.int64, device="cuda") if salted else None
    slope = 0.01
    zr, ld = rows_view(z)
    m = nb * h * w
    # two-pass route
    a = ops.new_act(nb, k, h, w, "cuda")
    L.call("arco_bn_act_fwd", L.ptr(zr), ld, m, k, L.ptr(mean), L.ptr(istd), L.ptr(gamma), L.ptr(beta), slope,
           1 if p > 0 else 0, p, seed, h * w, L.ptr(a), k, L.ptr(salt), groups)
    ar, lda = rows_view(a)
    wp = ops.pack_weight(wt, 9, 0)
    y0, (s0, q0, nmb0) = ops.conv_raw(ar, lda, k, wp, n, nb, h, w, 9, bias=bias, stats=True, stat_groups=groups)
    # loader route
    pro = L.act_pro(mean, istd, gamma, beta, slope, groups, 1 if p > 0 else 0, p, seed, salt)
    y1, (s1, q1, nmb1) = ops.conv_raw(zr, ld, k, wp, n, nb, h, w, 9, bias=bias, stats=True, stat_groups=groups, pro=pro)
    torch.cuda.synchronize()
    assert nmb0 == nmb1
    assert torch.equal(y0, y1)
    assert torch.equal(s0, s1) and torch.equal(q0, q1)
    if p > 0:      # the mask really dropped something (and not everything)
        frac = float((a == 0).float().mean())
        assert abs(frac - p) < 0.02, frac
    # weight gradient: dW = dZ^T . act(z)
    dz = ops.new_act(nb, n, h, w, "cuda")
    dz.copy_(_rand((nb, n, h, w), 8))
    dzr, lddz = rows_view(dz)
    like = torch.empty_like(wt)
    dw0 = ops.conv_wgrad(dzr, lddz, n, ar, lda, k, 9, nb, h, w, like)
    dw1 = ops.conv_wgrad(dzr, lddz, n, zr, ld, k, 9, nb, h, w, like, pro=pro)
    torch.cuda.synchronize()
    assert torch.equal(dw0, dw1)
    assert float(dw0.abs().max()) > 0


def _block_run(fuse, cin, cout, nb, hw, p, groups, pool, cat_room, seed):
    from arco_amd import ops
    from arco_amd.networks.unetWithArgs import ConvBlock
    torch.manual_seed(3)
    blk = ConvBlock(cin, cout, p).cuda().train()
    blk.cat_room = cat_room
    with torch.no_grad():
        for prm in blk.parameters():
            prm.copy_(_rand(tuple(prm.shape), 11 + prm.numel() % 7, 0.2))
    x = ops.to_channels_last(_rand((nb, cin, hw, hw), 21)).requires_grad_(True)
    prev = ops.BLOCK_FUSE
    ops.BLOCK_FUSE = fuse
    ops.reseed_dropout(seed)
    before = dict(ops.block_fuse_stats)
    try:
        with ops.bn_groups(groups):
            out = blk(x, pool=pool)
    finally:
        ops.BLOCK_FUSE = prev
    outs = out if pool else (out,)
    loss = sum((o * _rand(tuple(o.shape), 31 + i)).sum() for i, o in enumerate(outs))
    loss.backward()
    torch.cuda.synchronize()
    took = ops.block_fuse_stats["fused"] - before["fused"]
    res = [o.detach().clone() for o in outs] + [x.grad.clone()] + [prm.grad.clone() for prm in blk.parameters()]
    res += [b.clone() for b in blk.buffers()]
    return res, took


@pytest.mark.parametrize("cin,cout,nb,hw", [(16, 16, 8, 128), (16, 32, 8, 128), (32, 64, 8, 64), (256, 128, 16, 32), (128, 256, 16, 16)])
@pytest.mark.parametrize("p,groups,pool,cat_room", [(0.0, 1, False, 0), (0.2, 2, True, 0), (0.1, 2, True, 8), (0.3, 1, False, 16)])
def test_fused_block_equals_two_stages(cin, cout, nb, hw, p, groups, pool, cat_room):
    """ConvBlock forward + backward: ops.ConvBlockFn (first activation never written) vs two ConvBnActFn stages - outputs, input
    gradient, every parameter gradient and the BatchNorm buffers, bit for bit, with dropout, BN groups, pooled output, concat room."""
    cr = cout if cat_room else 0
    a, took_a = _block_run(1, cin, cout, nb, hw, p, groups, pool, cr, 77)
    b, took_b = _block_run(0, cin, cout, nb, hw, p, groups, pool, cr, 77)
    assert took_b == 0
    if took_a == 0:
        pytest.skip("block not taken by the fused route at this shape")
    assert len(a) == len(b)
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), (i, float((u - v).abs().max()))


def test_whole_unet_fused_equals_unfused():
    """U-Net forward + backward at 8 x 128^2 (two BN groups, dropout on): logits, feature maps, every gradient and buffer bit-identical
    with and without the fused blocks; the fused run takes the fused route for the levels the pipelined kernels cover."""
    from arco_amd import ops
    from arco_amd.networks.unetWithArgs import UNet

    def run(fuse):
        net = UNet(1, 4).cuda()
        net.load_state_dict(fx.unet_state(21), strict=True)
        net.train()
        x = fx.image_batch(5, 8, 1, (128, 128)).cuda()
        prev = ops.BLOCK_FUSE
        ops.BLOCK_FUSE = fuse
        ops.reseed_dropout(5)
        before = dict(ops.block_fuse_stats)
        try:
            with ops.bn_groups(2):
                logits, latent, fmap = net(x)
            loss = (logits * _rand(tuple(logits.shape), 1)).sum() + sum((f * _rand(tuple(f.shape), 2 + i)).sum() for i, f in enumerate(fmap))
            loss.backward()
        finally:
            ops.BLOCK_FUSE = prev
        torch.cuda.synchronize()
        took = ops.block_fuse_stats["fused"] - before["fused"]
        return [logits.detach()] + [f.detach() for f in fmap] + [p.grad for p in net.parameters()] + list(net.buffers()), took

    a, na = run(1)
    b, nb_ = run(0)
    assert nb_ == 0 and na >= 3, (na, nb_)
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), (i, float((u.float() - v.float()).abs().max()))


@pytest.mark.parametrize("cfg", [dict(n=2, c=32, sp=(9, 28, 20), nv=2, groups=1), dict(n=3, c=64, sp=(7, 14, 10), nv=4, groups=2),
                                 dict(n=3, c=128, sp=(5, 7, 5), nv=2, groups=2), dict(n=2, c=32, sp=(12, 56, 40), nv=2, groups=2)])
def test_vnet_convblock_gradient_free_pass_fuses_its_stage_links(cfg):
    """vnetWithArgs.py:5-31 in a gradient-free pass (the teacher's forwards, the warped student pass): stage i + 1 reads stage i's
    pre-activation through conv3d_fc_kernel's loaders (ops.conv_block3d_nograd) - output and running statistics bit-identical to the
    staged route; with gradients enabled the block takes the staged route.  (Opt-in, ops.BLOCK_FUSE3D: measured level with the staged route.)"""
    from arco_amd import ops
    from arco_amd.networks.vnetWithArgs import ConvBlock
    torch.manual_seed(cfg["c"] + cfg["n"])
    blk = ConvBlock(cfg["n"], cfg["c"], cfg["c"], normalization='batchnorm').cuda().train()
    x = torch.randn(cfg["nv"], cfg["c"], *cfg["sp"], device="cuda").contiguous(memory_format=torch.channels_last_3d)
    state0 = {k: v.clone() for k, v in blk.state_dict().items()}
    prev, prev_mma = ops.BLOCK_FUSE3D, ops.CONV_MMA
    ops.CONV_MMA = 3
    out, states, fused = {}, {}, {}
    try:
        for on in (0, 1):
            ops.BLOCK_FUSE3D = on
            blk.load_state_dict(state0)
            n0 = ops.block_fuse_stats.get("fused3d", 0)
            with torch.no_grad(), ops.bn_groups(cfg["groups"]):
                out[on] = blk(x).clone()
            fused[on] = ops.block_fuse_stats.get("fused3d", 0) - n0
            states[on] = {k: v.clone() for k, v in blk.state_dict().items()}
        ops.BLOCK_FUSE3D = 1
        n0 = ops.block_fuse_stats.get("fused3d", 0)
        y = blk(x.clone().requires_grad_(True))              # gradients enabled: staged route
        assert ops.block_fuse_stats.get("fused3d", 0) == n0 and y.requires_grad
    finally:
        ops.BLOCK_FUSE3D, ops.CONV_MMA = prev, prev_mma
    assert fused == {0: 0, 1: cfg["n"] - 1}
    assert torch.equal(out[0], out[1])
    for k in states[0]:
        assert torch.equal(states[0][k], states[1][k]), k
