"""f16 ACTIVATION STORAGE of the volume path (BASELINE.json configs[4], "fp16 MFMA conv"; csrc/conv_h.hip, the *_h entry points)
against the fp32 kernels of the same operators on the SAME f16-representable inputs: what differs is only the rounding of the
stored outputs (2^-11 relative) - products are exact in both (f16 x f16 fits fp32), accumulation is fp32 in both.  Tolerances are
written per test.  Reference operators: nn.Conv3d / BatchNorm3d / ReLU / ConvTranspose3d of vnetWithArgs.py:5-31,67-118."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last_3d)


def _rand_act(rs, shape, dev, scale=1.0):
    x = torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32)).to(dev)
    return _cl(x.half())            # f16-representable values, channels-last


def _maxrel(a, b):
    return float((a.float() - b.float()).abs().max()) / max(1e-12, float(b.float().abs().max()))


def _torch_conv_ref(x16, w, b, dy16, pad):
    """Plain PyTorch on the CPU, float64, same f16-representable operands: (y, dx, dw, db) of nn.Conv3d - the f16 kernels are
    held to it DIRECTLY (they used to be pinned only through their fp32 HIP siblings)."""
    import torch.nn.functional as F
    x = x16.detach().cpu().double().contiguous().requires_grad_(True)
    wr = w.detach().cpu().double().requires_grad_(True)
    br = b.detach().cpu().double().requires_grad_(True)
    y = F.conv3d(x, wr, br, padding=pad)
    y.backward(dy16.detach().cpu().double().contiguous())
    return y.detach(), x.grad, wr.grad, br.grad


def _l2rel(a, b):
    return float((a.float() - b.float()).norm()) / max(1e-20, float(b.float().norm()))


# (ci, co, D, H, W): rectangular tiles (W % 16 == 0), flat tiles (narrow planes), ragged planes, several N-tile shapes
# (16, 16, ...) with H % 8 == 0 and (32, 32, ...) with H % 16 == 0, W % 16 == 0: the resident-weights kernel (units of several
# planes, ragged last unit, a single plane); the others: hconv_kernel
CONV3_SHAPES = [(16, 16, 6, 16, 32), (32, 32, 5, 24, 48), (16, 32, 4, 20, 12), (64, 64, 6, 12, 6), (128, 128, 5, 10, 6),
                (32, 64, 3, 9, 7), (256, 256, 3, 5, 3), (32, 32, 6, 32, 32), (16, 16, 11, 24, 16), (16, 16, 1, 8, 16),
                (32, 32, 9, 16, 48)]


@pytest.mark.parametrize("ci,co,D,H,W", CONV3_SHAPES)
def test_conv3x3x3_f16_storage_forward_backward(ci, co, D, H, W):
    from arco_amd import ops
    dev = "cuda:0"
    rs = np.random.RandomState(ci + co + W)
    x16 = _rand_act(rs, (2, ci, D, H, W), dev)
    w = torch.from_numpy((rs.standard_normal((co, ci, 3, 3, 3)) / np.sqrt(27 * ci)).astype(np.float32)).to(dev)
    w = w.half().float().requires_grad_(True)              # f16-representable weights: the f16 pack is then exact
    b = torch.from_numpy(rs.standard_normal(co).astype(np.float32)).to(dev).requires_grad_(True)
    dy16 = _rand_act(rs, (2, co, D, H, W), dev)
    outs = {}
    for mode in ("h", "f"):
        x = (x16 if mode == "h" else x16.float()).clone().requires_grad_(True)
        w.grad = b.grad = None
        y = ops.conv(x, w, b)
        assert y.dtype == (torch.float16 if mode == "h" else torch.float32)
        y.backward(dy16 if mode == "h" else dy16.float())
        outs[mode] = (y.detach().float(), x.grad.float(), w.grad.clone(), b.grad.clone())
    yh, dxh, dwh, dbh = outs["h"]
    yf, dxf, dwf, dbf = outs["f"]
    assert _maxrel(yh, yf) < 1e-3                # one f16 rounding of the output
    assert _maxrel(dxh, dxf) < 1e-3
    assert _maxrel(dwh, dwf) < 2e-5              # exact products, fp32 accumulation in another order
    assert _maxrel(dbh, dbf) < 2e-5
    yt, dxt, dwt, dbt = _torch_conv_ref(x16, w, b, dy16, 1)
    assert _maxrel(yh.cpu().double(), yt) < 1e-3 and _maxrel(dxh.cpu().double(), dxt) < 1e-3      # one f16 rounding of a stored result
    assert _maxrel(dwh.cpu().double(), dwt) < 2e-5 and _maxrel(dbh.cpu().double(), dbt) < 2e-5    # fp32 results of exact products


@pytest.mark.parametrize("ci,co,M", [(128, 32, (6, 10, 12)), (256, 64, (5, 6, 6)), (32, 128, (8, 12, 16)), (16, 2, (8, 16, 16)),
                                     (1024, 256, (3, 5, 3)), (256, 1024, (3, 5, 3)),
                                     # >= 65 536 rows with <= 4 channels on one side: the streaming kernels (hconv1x1_narrow_out / _in)
                                     (16, 2, (48, 40, 36)), (16, 4, (40, 40, 44)), (32, 2, (36, 40, 24)), (8, 4, (48, 48, 32))])
def test_conv1x1x1_f16_storage_forward_backward(ci, co, M):
    from arco_amd import ops
    dev = "cuda:0"
    rs = np.random.RandomState(ci + co)
    x16 = _rand_act(rs, (2, ci, *M), dev)
    w = torch.from_numpy((rs.standard_normal((co, ci, 1, 1, 1)) / np.sqrt(ci)).astype(np.float32)).to(dev).half().float().requires_grad_(True)
    b = torch.from_numpy(rs.standard_normal(co).astype(np.float32)).to(dev).requires_grad_(True)
    dy16 = _rand_act(rs, (2, co, *M), dev)
    outs = {}
    for mode in ("h", "f"):
        x = (x16 if mode == "h" else x16.float()).clone().requires_grad_(True)
        w.grad = b.grad = None
        y = ops.conv(x, w, b)
        y.backward(dy16 if mode == "h" else dy16.float())
        outs[mode] = (y.detach().float(), x.grad.float(), w.grad.clone(), b.grad.clone())
    ref = _torch_conv_ref(x16, w, b, dy16, 0)
    for k, tol in enumerate((1e-3, 1e-3, 2e-5, 2e-5)):
        assert _maxrel(outs["h"][k], outs["f"][k]) < tol, k
        assert _maxrel(outs["h"][k].cpu().double(), ref[k]) < tol, ("vs torch", k)


def test_first_layer_f16_output_and_weight_gradient():
    """The one-channel fp32 volume enters the f16 region through the first 3x3x3 layer (vnetWithArgs.py:182 block_one)."""
    from arco_amd import ops
    dev = "cuda:0"
    rs = np.random.RandomState(5)
    x = _cl(torch.from_numpy(rs.uniform(size=(2, 1, 6, 20, 24)).astype(np.float32)).to(dev))
    w = torch.from_numpy((rs.standard_normal((16, 1, 3, 3, 3)) / 5).astype(np.float32)).to(dev).requires_grad_(True)
    b = torch.from_numpy(rs.standard_normal(16).astype(np.float32)).to(dev).requires_grad_(True)
    dy16 = _rand_act(rs, (2, 16, 6, 20, 24), dev)
    outs = {}
    try:
        for half in (True, False):
            ops.ACT_HALF = half
            w.grad = b.grad = None
            y = ops.conv(x, w, b)
            assert y.dtype == (torch.float16 if half else torch.float32)
            y.backward(dy16 if half else dy16.float())
            outs[half] = (y.detach().float(), w.grad.clone(), b.grad.clone())
    finally:
        ops.ACT_HALF = False; ops.HEAD_MMA = 0
    assert _maxrel(outs[True][0], outs[False][0]) < 1e-3
    assert _maxrel(outs[True][1], outs[False][1]) < 2e-5
    assert _maxrel(outs[True][2], outs[False][2]) < 2e-5


@pytest.mark.parametrize("groups", [1, 2])
def test_conv_bn_relu_stage_f16_storage(groups):
    """conv -> train-mode BatchNorm (statistics from the conv epilogue, of the ROUNDED outputs) -> ReLU, forward and backward;
    running statistics updated alike."""
    from arco_amd import ops
    dev = "cuda:0"
    rs = np.random.RandomState(11)
    ci, co = 32, 32
    x16 = _rand_act(rs, (2, ci, 6, 16, 16), dev)
    w = torch.from_numpy((rs.standard_normal((co, ci, 3, 3, 3)) / np.sqrt(27 * ci)).astype(np.float32)).to(dev).half().float().requires_grad_(True)
    b = torch.zeros(co, device=dev, requires_grad=True)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, co).astype(np.float32)).to(dev).requires_grad_(True)
    beta = torch.from_numpy(rs.standard_normal(co).astype(np.float32) * 0.2).to(dev).requires_grad_(True)
    da16 = _rand_act(rs, (2, co, 6, 16, 16), dev)
    outs = {}
    for mode in ("h", "f"):
        rm, rv = torch.zeros(co, device=dev), torch.ones(co, device=dev)
        nbt = torch.zeros((), dtype=torch.long, device=dev)
        x = (x16 if mode == "h" else x16.float()).clone().requires_grad_(True)
        for t in (w, b, gamma, beta):
            t.grad = None
        with ops.bn_groups(groups):
            a = ops.conv_bn_act(x, w, b, gamma, beta, rm, rv, slope=0.0, p=0.0, num_batches_tracked=nbt)
        assert a.dtype == (torch.float16 if mode == "h" else torch.float32)
        a.backward(da16 if mode == "h" else da16.float())
        outs[mode] = (a.detach().float(), x.grad.float(), w.grad.clone(), gamma.grad.clone(), beta.grad.clone(), rm.clone(), rv.clone())
    # the f16 path normalises z rounded to f16 (2^-11 of |z| ~ 1) against statistics of the same rounded values.  A rounding can
    # move a pre-activation across the ReLU kink: the handful of elements it flips carry their whole gradient, so the gradients
    # are compared in the L2 norm (the flips are a 1e-3 fraction of the elements), the activation element-wise
    assert _maxrel(outs["h"][0], outs["f"][0]) < 3e-3
    for k in range(1, 5):
        assert _l2rel(outs["h"][k], outs["f"][k]) < 1e-2, k
    for k in (5, 6):
        assert _maxrel(outs["h"][k], outs["f"][k]) < 1e-3, k


def test_bn_act_and_dropout3d_on_f16_tensors():
    from arco_amd import ops
    dev = "cuda:0"
    rs = np.random.RandomState(3)
    z16 = _rand_act(rs, (2, 16, 4, 8, 8), dev)
    da16 = _rand_act(rs, (2, 16, 4, 8, 8), dev)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, 16).astype(np.float32)).to(dev).requires_grad_(True)
    beta = torch.zeros(16, device=dev, requires_grad=True)
    outs = {}
    for mode in ("h", "f"):
        z = (z16 if mode == "h" else z16.float()).clone().requires_grad_(True)
        gamma.grad = beta.grad = None
        a = ops.bn_act(z, gamma, beta, torch.zeros(16, device=dev), torch.ones(16, device=dev), slope=0.0)
        a.backward(da16 if mode == "h" else da16.float())
        outs[mode] = (a.detach().float(), z.grad.float(), gamma.grad.clone(), beta.grad.clone())
    assert _maxrel(outs["h"][0], outs["f"][0]) < 2e-3       # same stored z in both: the kink sits at the same elements
    assert _maxrel(outs["h"][1], outs["f"][1]) < 3e-3
    assert _maxrel(outs["h"][2], outs["f"][2]) < 1e-4 and _maxrel(outs["h"][3], outs["f"][3]) < 1e-4
    # ... and directly against nn.BatchNorm3d(train) + ReLU in float64 on the CPU (same f16-representable z and dA)
    zt = z16.detach().cpu().double().contiguous().requires_grad_(True)
    gt, bt = gamma.detach().cpu().double().requires_grad_(True), beta.detach().cpu().double().requires_grad_(True)
    at = torch.relu(torch.nn.functional.batch_norm(zt, None, None, gt, bt, True, 0.1, 1e-5))
    at.backward(da16.detach().cpu().double().contiguous())
    assert _maxrel(outs["h"][0].cpu().double(), at.detach()) < 2e-3
    assert _maxrel(outs["h"][1].cpu().double(), zt.grad) < 3e-3
    assert _maxrel(outs["h"][2].cpu().double(), gt.grad) < 1e-4 and _maxrel(outs["h"][3].cpu().double(), bt.grad) < 1e-4
    # Dropout3d: whole (sample, channel) volumes dropped, survivors scaled by 1 / (1 - p); same mask for both storage types
    ops.reseed_dropout(7)
    dh = ops.dropout3d(z16, 0.5)
    ops.reseed_dropout(7)
    df = ops.dropout3d(z16.float(), 0.5)
    assert dh.dtype == torch.float16 and _maxrel(dh, df) < 1e-3
    kept = (dh.float().abs().sum(dim=(2, 3, 4)) > 0)
    assert 0 < int(kept.sum()) < kept.numel()


def test_space_to_depth_and_boundary_casts_on_f16_tensors():
    from arco_amd import ops
    dev = "cuda:0"
    rs = np.random.RandomState(9)
    x16 = _rand_act(rs, (2, 16, 4, 6, 8), dev)
    s = ops.space_to_depth3(x16)
    assert s.dtype == torch.float16 and tuple(s.shape) == (2, 128, 2, 3, 4)
    assert torch.equal(s.float(), ops.space_to_depth3(x16.float()))
    assert torch.equal(ops.depth_to_space3(s), x16)
    x = x16.clone().requires_grad_(True)
    y = ops.from_half(x)
    assert y.dtype == torch.float32 and torch.equal(y, x16.float())
    g = torch.from_numpy((rs.standard_normal(tuple(y.shape)) * 1e-7).astype(np.float32)).to(dev)
    y.backward(_cl(g))
    ref = (g * ops.LOSS_SCALE).half()
    assert x.grad.dtype == torch.float16 and torch.equal(x.grad, _cl(ref))
    assert float(x.grad.float().abs().max()) > 1e-4          # 1e-7 survives thanks to the loss scale


def test_vnet_f16_storage_tracks_fp32_forward_and_gradients():
    """The whole V-Net (batchnorm, dropout off), 2 volumes of 64 x 64 x 32: logits and every parameter gradient of the f16-storage
    body against the fp32 body (split-bf16 kernels) from the same weights - the budget of BASELINE.json configs[4] is 1e-2."""
    from arco_amd import ops
    from arco_amd.networks.vnetWithArgs import VNet
    dev = "cuda:0"
    torch.manual_seed(3)
    net = VNet(n_channels=1, n_classes=2, normalization='batchnorm', has_dropout=False).to(dev).train()
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.uniform(size=(2, 1, 64, 64, 32)).astype(np.float32)).to(dev)
    tgt = torch.from_numpy(rs.standard_normal((2, 2, 64, 64, 32)).astype(np.float32)).to(dev)
    res = {}
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    try:
        for half in (False, True, "operands"):       # "operands": fp32 tensors, f16-rounded MFMA operands (--conv_mma f16), for scale
            ops.ACT_HALF = half is True
            ops.CONV_MMA = 1 if half == "operands" else 3
            ops.bump_weight_epoch()
            net.load_state_dict(sd)
            net.zero_grad()
            out, f0, fm = net(x)
            assert out.dtype == torch.float32 and all(f.dtype == torch.float32 for f in fm)
            loss = ((out - tgt) ** 2).mean() + sum((f ** 2).mean() for f in fm) * 0.1
            loss.backward()
            scale = ops.LOSS_SCALE if half is True else 1.0
            res[half] = (out.detach().clone(), [f.detach().clone() for f in fm],
                         {n: p.grad.detach().clone() / scale for n, p in net.named_parameters() if p.grad is not None})
    finally:
        ops.ACT_HALF = False; ops.HEAD_MMA = 0
        ops.CONV_MMA = 3
        ops.bump_weight_epoch()
    # Random weights, ~25 rounded layers: a perturbation of 2^-11 per stored value grows through the untrained network (every
    # conv + BatchNorm + ReLU stage amplifies it a little), so this comparison is looser than the per-operator ones above and
    # than the loss terms at LiTS size (test_cfg5_... in test_configs_at_size_gpu.py: 1e-2).  The operand-rounding mode on
    # fp32 tensors is measured beside it: storage rounding must stay in that mode's league.
    e2, em = _l2rel(res[True][0], res[False][0]), _maxrel(res[True][0], res[False][0])
    o2 = _l2rel(res["operands"][0], res[False][0])
    print("logits: relative L2 / max error", e2, em, "| f16-operand mode on fp32 tensors: L2", o2)
    assert e2 < 2e-2 and em < 3e-2 and e2 < 3 * max(o2, 3e-3), (e2, em, o2)
    for i, (a, b) in enumerate(zip(res[True][1], res[False][1])):
        e2, em = _l2rel(a, b), _maxrel(a, b)
        assert e2 < 2e-2 and em < 5e-2, (i, e2, em)
    assert set(res[True][2]) == set(res[False][2])
    # Gradients: a forward perturbation of 1e-3 moves ~4e-4 of a layer's pre-activations across the ReLU kink, and every flipped
    # element carries its whole gradient - an L2 gradient difference of ~2 % per layer that accumulates towards the input, for ANY
    # two forwards that differ by a rounding (the same effect made tests/golden/g15 search a kink-free input for the U-Net).  The
    # gradients of the last layers (no flips below them yet) are held to 2e-2; all of them must stay within 1.5 x the distance
    # of the operand-rounding mode on fp32 tensors (+ 2e-2), which rounds the same activations at every conv input.
    worst = 0.0
    for n, g in res[False][2].items():
        gh, go = res[True][2][n], res["operands"][2][n]
        if float(g.abs().max()) < 1e-9:           # conv biases under train-mode BN: exact zeros in every mode
            assert float(gh.abs().max()) < 1e-6
            continue
        e, eo = float((gh - g).norm()) / float(g.norm()), float((go - g).norm()) / float(g.norm())
        worst = max(worst, e)
        assert e < 1.5 * eo + 2e-2, (n, e, eo)
        if n.startswith(("out_conv", "block_nine")):
            assert e < 2e-2, (n, e)
    print("worst relative L2 gradient error", worst)


def _make3d_small(extra):
    import random
    from arco_amd import train_arco_3d as T3
    args = T3.build_parser().parse_args(["--batch_size", "1", "--queue_size", "256", "--synthetic", "1", "--num_classes", "2",
                                         "--num_queries", "32", "--num_negatives", "16", "--k1", "1.0", "--act_dtype", "f16"] + list(extra))
    args.patch_size = [32, 32, 32]
    random.seed(5); np.random.seed(5); torch.manual_seed(5)
    return T3.ArcoStep3D(args, "cuda:0")


def test_f16_storage_step_graph_replay_equals_eager():
    """The whole 3-D step with f16 activation storage (boundary casts, loss scale, un-scaling of the flat gradient, f16 packs from
    the PackPlan): five steps with the passes replayed as HIP graphs (trainer default; captured at the third call) against five
    eager steps from the same state - same kernels in the same order, so the loss terms and the updated weights must agree."""
    import random
    from arco_amd import ops, train_arco_3d as T3
    try:
        st_g, st_e = _make3d_small([]), _make3d_small(["--graphs", "0", "--graph_train", "0"])
        assert ops.ACT_HALF and st_g.args.graph_train == 1
        st_e.isd.load_state_dict(st_g.isd.state_dict())
        st_e.q_representation.load_state_dict(st_g.q_representation.state_dict())
        st_e.q_feature_extractor.load_state_dict(st_g.q_feature_extractor.state_dict())
        st_e.k_feature_extractor.load_state_dict(st_g.k_feature_extractor.state_dict())
        ops.bump_weight_epoch()
        for st in (st_g, st_e):
            for m in (st.model, st.ema_model):
                m.has_dropout = False
        terms = {}
        for name, st in (("g", st_g), ("e", st_e)):
            out = []
            for it in range(5):
                l, ll = T3.synthetic_volume_batch(1, (32, 32, 32), 2, 10 + it, "cuda:0")
                u, _ = T3.synthetic_volume_batch(1, (32, 32, 32), 2, 20 + it, "cuda:0")
                random.seed(100 + it); np.random.seed(100 + it); torch.manual_seed(100 + it)
                st.step(l, ll, u)
                out.append([float(st.last_terms[k]) for k in ("ce", "dice", "unsup", "reco", "eqv")])
            terms[name] = np.array(out)
        assert np.all(np.isfinite(terms["g"]))
        np.testing.assert_array_equal(terms["g"][:2], terms["e"][:2])             # both eager: bit-identical
        # replayed steps: the captured backward accumulates the flat gradient in another order (1e-7), which the f16 roundings of
        # the following forwards amplify like any other perturbation of that size (see the V-Net test above)
        np.testing.assert_allclose(terms["g"], terms["e"], rtol=1e-2, atol=1e-5)
        sd_g, sd_e = st_g.model.state_dict(), st_e.model.state_dict()
        for k, v in sd_e.items():
            if v.is_floating_point() and v.dim() > 1:          # conv weights (BatchNorm shifts start at 0: five tiny updates of noise scale)
                assert float((sd_g[k] - v).abs().max()) <= 5e-2 * max(1e-6, float(v.abs().max())), k
        # the f16 region really ran: the student's first activation is f16, its gradients were un-scaled (finite, small)
        assert st_g.model.block_one.conv[0].weight.grad is None or torch.isfinite(st_g.model.block_one.conv[0].weight.grad).all()
    finally:
        ops.ACT_HALF = False; ops.HEAD_MMA = 0
        ops.bump_weight_epoch()


def test_f16_backward_overflow_is_contained_and_the_loss_scale_adapts():
    """an overflow of the f16 backward must not reach the optimiser.  A loss scale of 2^40 drives the boundary
    cast into its saturation (+-65504, arco_cast_f2h) and the f16 backward into inf / NaN: the guarded step must leave every
    parameter, the EMA teacher, the momentum buffers and the banks finite, count the overflow, halve the scale (re-capturing the
    backward graphs that carry it as a kernel argument) - and with a sane scale no step is flagged."""
    import random
    from arco_amd import ops, train_arco_3d as T3
    try:
        st = _make3d_small(["--loss_scale", str(float(2 ** 40))])
        assert ops.ACT_HALF and ops.LOSS_SCALE == float(2 ** 40)
        scales = []
        for it in range(6):
            l, ll = T3.synthetic_volume_batch(1, (32, 32, 32), 2, 10 + it, "cuda:0")
            u, _ = T3.synthetic_volume_batch(1, (32, 32, 32), 2, 20 + it, "cuda:0")
            random.seed(100 + it); np.random.seed(100 + it); torch.manual_seed(100 + it)
            loss, reco = st.step(l, ll, u)
            scales.append(ops.LOSS_SCALE)
            assert bool(torch.isfinite(st.optimizer.flat_p).all()) and bool(torch.isfinite(st.optimizer.flat_buf).all()), it
            assert all(bool(torch.isfinite(p).all()) for p in st.ema_model.parameters()), it
            assert all(bool(torch.isfinite(m[0]).all()) for m in st.memobank), it
        assert st.overflow_steps >= 2, (st.overflow_steps, scales)
        assert scales[-1] <= float(2 ** 40) / 4 and scales == sorted(scales, reverse=True), scales
        # the cast saturates instead of producing inf
        big = torch.full((2, 16, 4, 4, 4), 1e30, device="cuda").contiguous(memory_format=torch.channels_last_3d)
        x = _rand_act(np.random.RandomState(0), (2, 16, 4, 4, 4), "cuda:0").requires_grad_(True)
        ops.from_half(x).backward(big)
        assert float(x.grad.float().abs().max()) == 65504.0
        st2 = _make3d_small([])
        for it in range(3):
            l, ll = T3.synthetic_volume_batch(1, (32, 32, 32), 2, 10 + it, "cuda:0")
            u, _ = T3.synthetic_volume_batch(1, (32, 32, 32), 2, 20 + it, "cuda:0")
            st2.step(l, ll, u)
        st2._loss_scale_update()
        assert st2.overflow_steps == 0 and ops.LOSS_SCALE == 16384.0
    finally:
        ops.ACT_HALF = False; ops.HEAD_MMA = 0
        ops.LOSS_SCALE = 16384.0
        ops.bump_weight_epoch()


def test_row_sparse_heads_on_f16_feature_maps_equal_the_cast_path():
    """ops.fm_rows_half (round 4): the row-sparse student head (head.lazy_head3d; FeatureExtractor_3d + q_representation rows,
    model_3D.py:46-58, train_arco_3d.py:289-296) and the lazy teacher (prototypes = class-weighted row sums, key rows) read the two
    full-resolution feature maps as stored f16 and return a row-sparse f16 gradient carrying the loss scale.  Against the path they
    replace - dense cast to fp32 (ops.from_half), fp32 heads, dense cast of the fp32 gradient back: identical rows, identical
    prototypes (same fp32 sums of the same f16-representable values), identical weight / low-resolution gradients, and the f16
    feature-map gradients bit-equal (rows summed in fp32, rounded once, in both)."""
    from arco_amd import head, ops, _contrast as C_
    import fixture_inputs as fx
    dev = "cuda:0"
    rs = np.random.RandomState(3)
    nb, c2, c3, c4, sp, lo_sp = 2, 224, 16, 16, (16, 16, 16), (8, 8, 8)
    k3 = c2 + c3
    x2p = _cl(torch.from_numpy(rs.standard_normal((nb, c2) + lo_sp).astype(np.float32)).to(dev))
    f3h, f4h = _rand_act(rs, (nb, c3) + sp, dev), _rand_act(rs, (nb, c4) + sp, dev)
    w3 = torch.from_numpy((rs.standard_normal((k3, k3, 1, 1, 1)) / np.sqrt(k3)).astype(np.float32)).to(dev)
    w4 = torch.from_numpy((rs.standard_normal((16, k3 + c4, 1, 1, 1)) / np.sqrt(k3 + c4)).astype(np.float32)).to(dev)
    w1 = torch.from_numpy((rs.standard_normal((16, 16, 1, 1, 1)) / 4).astype(np.float32)).to(dev)
    w2 = torch.from_numpy((rs.standard_normal((16, 16, 1, 1, 1)) / 4).astype(np.float32)).to(dev)
    n_vox = nb * sp[0] * sp[1] * sp[2]
    pix = torch.from_numpy(rs.randint(0, n_vox, size=300)).to(dev)
    pix[7] = pix[3]; pix[100] = pix[3]                      # repeated voxels: their gradient rows add up
    da = torch.from_numpy(rs.standard_normal((300, 16)).astype(np.float32) * 1e-3).to(dev)
    prev_scale = ops.LOSS_SCALE
    try:
        ops.LOSS_SCALE = 1024.0
        res = {}
        for mode in ("half", "cast"):
            leaves = [t.clone().requires_grad_(True) for t in (x2p, w3, w4, w1, w2)]
            f3l, f4l = f3h.clone().requires_grad_(True), f4h.clone().requires_grad_(True)
            f3, f4 = (f3l, f4l) if mode == "half" else (ops.from_half(f3l), ops.from_half(f4l))
            a = head.lazy_head3d(leaves[0], f3, f4, leaves[1], leaves[2], leaves[3], leaves[4], pix)
            a.backward(da)
            res[mode] = [a.detach()] + [t.grad for t in leaves] + [f3l.grad, f4l.grad]
            assert f3l.grad.dtype == torch.float16 and f4l.grad.dtype == torch.float16
        for i, (h, c) in enumerate(zip(res["half"], res["cast"])):
            if i in (1, 6, 7):      # scattered with fp32 atomics (trilinear adjoint / repeated voxels): summation order is free
                tol = 1e-6 if i == 1 else 2.0 ** -10
                assert float((h.float() - c.float()).abs().max()) <= tol * float(c.float().abs().max()), i
                if i > 1:           # every row that was touched once is bit-equal
                    assert float((h != c).float().mean()) < 1e-4, i
            else:
                assert torch.equal(h, c), (i, float((h.float() - c.float()).abs().max()))
        g3 = res["half"][-2].float()
        assert float(g3.abs().max()) > 0 and int((g3.movedim(1, -1).reshape(n_vox, c3).abs().sum(1) > 0).sum()) <= 298
        # teacher: prototypes and key rows
        inp = {k: v.to(dev) for k, v in fx.loss_inputs(9, b=1, n_cls=3, feat=16, spatial=sp).items()}
        pl = C_.contrast_masks(inp["label_l"], inp["label_u"], inp["prob_l"], inp["prob_u"], inp["low_mask"], inp["high_mask"], 0.97)
        t_h = head.LazyTeacher3D(x2p, f3h, f4h, w3, w4)
        t_c = head.LazyTeacher3D(x2p, ops.from_half(f3h), ops.from_half(f4h), w3, w4)
        assert torch.equal(t_h.prototypes(pl), t_c.prototypes(pl))
        assert torch.equal(t_h.rows(pix), t_c.rows(pix))
    finally:
        ops.LOSS_SCALE = prev_scale


def test_f16_step_with_f16_feature_map_rows_equals_the_dense_cast_step():
    """Whole 3-D steps in f16 mode with train_arco_3d.FM_ROWS_HALF on (default) and off: the same numbers reach the same kernels
    (the casts of the touched rows replace the casts of the whole maps), so loss terms and weights agree exactly, eager and replayed."""
    import random
    from arco_amd import ops, train_arco_3d as T3
    prev = T3.FM_ROWS_HALF
    try:
        sts = {}
        for flag in (1, 0):
            T3.FM_ROWS_HALF = flag
            sts[flag] = _make3d_small([])
        for k in ("isd", "q_representation", "q_feature_extractor", "k_feature_extractor"):
            getattr(sts[0], k).load_state_dict(getattr(sts[1], k).state_dict())
        ops.bump_weight_epoch()
        for st in sts.values():
            for m in (st.model, st.ema_model):
                m.has_dropout = False
        def sync(dst, src):       # every step starts from EQUAL state: what is compared is one step, not a trajectory
            with torch.no_grad():
                dst.optimizer.flat_p.copy_(src.optimizer.flat_p)
                dst.optimizer.flat_buf.copy_(src.optimizer.flat_buf)
                dst.optimizer._started = list(src.optimizer._started)
                for md, ms in ((dst.model, src.model), (dst.ema_model, src.ema_model), (dst.k_feature_extractor, src.k_feature_extractor)):
                    for (kd, vd), (ks, vs) in zip(md.state_dict().items(), ms.state_dict().items()):
                        vd.copy_(vs)
                dst.memobank = [[m[0].clone()] for m in src.memobank]
                dst.queue_ptrlis = [q.clone() if torch.is_tensor(q) else q for q in src.queue_ptrlis]
            ops.bump_weight_epoch()

        for it in range(5):
            l, ll = T3.synthetic_volume_batch(1, (32, 32, 32), 2, 10 + it, "cuda:0")
            u, _ = T3.synthetic_volume_batch(1, (32, 32, 32), 2, 20 + it, "cuda:0")
            sync(sts[0], sts[1])
            terms = {}
            for flag, st in sts.items():
                T3.FM_ROWS_HALF = flag
                random.seed(100 + it); np.random.seed(100 + it); torch.manual_seed(100 + it)
                st.step(l, ll, u)
                terms[flag] = [float(st.last_terms[k]) for k in ("ce", "dice", "unsup", "reco", "eqv")]
            # the forward sees identical numbers (exact); the gradients differ by the order of fp32 atomic adds in the scatters
            assert terms[1] == terms[0], (it, terms)
            pa, pb = sts[1].optimizer.flat_p, sts[0].optimizer.flat_p
            assert float((pa - pb).abs().max()) <= 1e-6 * float(pa.abs().max()), (it, float((pa - pb).abs().max()))
        assert sts[1].s_train_lu.captured and sts[1].s_train_lu.flat_outs[-1].dtype == torch.float16      # the last feature map left the graph as f16
        assert sts[0].s_train_lu.flat_outs[-1].dtype == torch.float32
    finally:
        T3.FM_ROWS_HALF = prev
        ops.ACT_HALF = False; ops.HEAD_MMA = 0
        ops.bump_weight_epoch()


def test_head_gemms_on_f16_operands():
    """ops.HEAD_MMA = 1 (train_arco_3d --act_dtype f16, --head_mma auto): a 1x1x1 convolution over an fp32 map - FeatureExtractor_3d /
    q_representation (model_3D.py:37-63, train_arco_3d.py:206-209) - rounds its operands to f16 in registers (bf16 for the data
    gradient's operands), fp32 accumulate: output and input gradient within 2e-3 of the fp32-accurate GEMM (K = 448: 1e-2 budget of
    BASELINE configs[4] with room), not bit-equal to it (the reduced-precision kernel really ran), weight gradient fp32."""
    from arco_amd import ops
    dev = "cuda:0"
    torch.manual_seed(0)
    x = ops.to_channels_last(torch.randn(2, 448, 8, 12, 10, device=dev))
    w = (torch.randn(448, 448, 1, 1, 1, device=dev) * 0.05)
    gy = ops.to_channels_last(torch.randn(2, 448, 8, 12, 10, device=dev))
    res = {}
    try:
        for mode in (0, 1):
            ops.HEAD_MMA = mode
            xi, wi = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            y = ops.conv(xi, wi, None, residual=True)
            y.backward(gy)
            res[mode] = (y.detach().clone(), xi.grad.clone(), wi.grad.clone())
    finally:
        ops.HEAD_MMA = 0
    for k, tol in ((0, 2e-3), (1, 1e-2)):          # forward f16 operands (2^-11), data gradient bf16 operands (2^-8)
        a, b = res[1][k], res[0][k]
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()), k
        assert not torch.equal(a, b), k
    assert float((res[1][2] - res[0][2]).abs().max()) <= 1e-5 * float(res[0][2].abs().max())      # weight gradient: fp32 either way
