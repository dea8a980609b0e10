"""HIP operators vs a plain PyTorch-CPU fp32 reference of the same op (-m gpu)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rnd(rs, *shape, scale=1.0):
    return torch.from_numpy((scale * rs.standard_normal(shape)).astype(np.float32))


def cl(x):
    return x.cuda().contiguous(memory_format=torch.channels_last)


def close(a, b, rtol=2e-4, atol=2e-5):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("nb,ci,co,h,w,k,bias,res", [
    (2, 16, 16, 32, 32, 3, True, False), (1, 1, 16, 48, 32, 3, True, False), (2, 3, 16, 20, 24, 3, True, False),
    (2, 32, 64, 16, 16, 3, True, False), (1, 64, 128, 32, 32, 3, True, False), (2, 256, 256, 16, 16, 3, True, False),
    (1, 16, 4, 32, 32, 3, True, False), (1, 16, 19, 16, 32, 3, True, False), (2, 128, 64, 8, 8, 1, True, False),
    (1, 496, 496, 24, 24, 1, False, False), (2, 48, 48, 16, 16, 1, False, True), (1, 384, 384, 32, 32, 1, False, True),
    (1, 32, 16, 64, 64, 1, True, False), (3, 20, 36, 10, 14, 3, False, False),
    # persistent halo-tile kernel (Cin in {16,32}, Cout <= 32): all instantiations, ragged edges, residual
    (2, 16, 32, 13, 37, 3, True, False), (2, 32, 16, 9, 70, 3, False, False), (3, 32, 32, 21, 19, 3, True, False),
    (1, 16, 16, 7, 5, 3, True, False), (2, 32, 32, 24, 40, 3, False, True), (2, 16, 16, 40, 64, 3, True, True),
    (1, 32, 4, 17, 33, 3, True, False), (5, 16, 16, 256, 256, 3, True, False), (3, 32, 32, 200, 136, 3, True, False),
])
def test_conv_fwd_bwd(nb, ci, co, h, w, k, bias, res):
    from arco_amd import ops
    rs = np.random.RandomState(nb * 1000 + ci + co + h)
    x = rnd(rs, nb, ci, h, w)
    wt = rnd(rs, co, ci, k, k, scale=1.0 / np.sqrt(ci * k * k))
    b = rnd(rs, co, scale=0.1) if bias else None
    gy = rnd(rs, nb, co, h, w)
    xr = x.clone().requires_grad_(True); wr = wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, padding=k // 2)
    if res:
        yr = yr + xr
    yr.backward(gy)
    xg = cl(x).requires_grad_(True); wg = wt.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True) if bias else None
    yg = ops.conv(xg, wg, bg, residual=res)
    assert yg.shape == yr.shape
    yg.backward(cl(gy))
    close(yg, yr)
    close(xg.grad, xr.grad)
    close(wg.grad, wr.grad, rtol=5e-4, atol=5e-4 * float(wr.grad.abs().max()))
    if bias:
        close(bg.grad, br.grad, rtol=5e-4, atol=5e-4 * float(br.grad.abs().max()))


@pytest.mark.parametrize("nb,ci,co,h,w,slope", [(2, 16, 16, 32, 32, 0.01), (2, 1, 16, 32, 32, 0.01),
                                                 (2, 64, 128, 16, 16, 0.01), (1, 32, 32, 48, 64, 0.0)])
def test_conv_bn_act(nb, ci, co, h, w, slope):
    from arco_amd import ops
    rs = np.random.RandomState(ci + co + h)
    x = rnd(rs, nb, ci, h, w); wt = rnd(rs, co, ci, 3, 3, scale=1 / np.sqrt(9 * ci)); b = rnd(rs, co, scale=0.1)
    gam = 1 + rnd(rs, co, scale=0.1); bet = rnd(rs, co, scale=0.1); gy = rnd(rs, nb, co, h, w)
    rm, rv = torch.zeros(co), torch.ones(co)
    leaves = [t.clone().requires_grad_(True) for t in (x, wt, b, gam, bet)]
    z = F.conv2d(leaves[0], leaves[1], leaves[2], padding=1)
    y = F.leaky_relu(F.batch_norm(z, rm, rv, leaves[3], leaves[4], True, 0.1, 1e-5), slope)
    y.backward(gy)
    gl = [cl(x).requires_grad_(True)] + [t.cuda().requires_grad_(True) for t in (wt, b, gam, bet)]
    rmg, rvg = torch.zeros(co).cuda(), torch.ones(co).cuda()
    yg = ops.conv_bn_act(gl[0], gl[1], gl[2], gl[3], gl[4], rmg, rvg, slope=slope, p=0.0)
    yg.backward(cl(gy))
    close(yg, y, rtol=5e-4, atol=5e-5)
    close(rmg, rm, rtol=1e-4, atol=1e-6); close(rvg, rv, rtol=1e-4, atol=1e-6)
    for i, (a, r) in enumerate(zip(gl, leaves)):
        if i == 2:
            # conv bias under train-mode BN: analytically zero gradient, only rounding noise on both sides
            assert float(a.grad.abs().max()) < 1e-3 and float(r.grad.abs().max()) < 1e-3
            continue
        scale = float(r.grad.abs().max())
        close(a.grad, r.grad, rtol=2e-3, atol=2e-4 * max(scale, 1e-3))


def test_dropout_statistics_and_backward_mask():
    from arco_amd import ops
    rs = np.random.RandomState(0)
    x = cl(rnd(rs, 2, 16, 64, 64)).requires_grad_(True)
    wt = rnd(rs, 16, 16, 3, 3, scale=0.1).cuda(); b = torch.zeros(16).cuda()
    y = ops.conv_bn_act(x, wt, b, torch.ones(16).cuda(), torch.zeros(16).cuda(), torch.zeros(16).cuda(),
                        torch.ones(16).cuda(), slope=1.0, p=0.3)
    y0 = ops.conv_bn_act(x, wt, b, torch.ones(16).cuda(), torch.zeros(16).cuda(), torch.zeros(16).cuda(),
                         torch.ones(16).cuda(), slope=1.0, p=0.0)
    dropped = (y == 0) & (y0 != 0)
    frac = dropped.float().mean().item()
    assert abs(frac - 0.3) < 0.01
    kept = ~dropped
    close(y[kept], y0[kept] / 0.7, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("c,h,w", [(16, 32, 32), (64, 16, 24)])
def test_maxpool(c, h, w):
    from arco_amd import ops
    rs = np.random.RandomState(1)
    x = rnd(rs, 2, c, h, w); gy = rnd(rs, 2, c, h // 2, w // 2)
    xr = x.clone().requires_grad_(True); yr = F.max_pool2d(xr, 2); yr.backward(gy)
    xg = cl(x).requires_grad_(True); yg = ops.maxpool2(xg); yg.backward(cl(gy))
    close(yg, yr, 0, 0); close(xg.grad, xr.grad, 0, 0)


@pytest.mark.parametrize("c,hi,wi,ho,wo", [(16, 16, 16, 32, 32), (32, 2, 2, 4, 4), (48, 8, 12, 16, 24), (16, 5, 7, 13, 9)])
def test_bilinear(c, hi, wi, ho, wo):
    from arco_amd import ops
    rs = np.random.RandomState(2)
    x = rnd(rs, 2, c, hi, wi); gy = rnd(rs, 2, c, ho, wo)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, size=(ho, wo), mode='bilinear', align_corners=True); yr.backward(gy)
    xg = cl(x).requires_grad_(True); yg = ops.bilinear(xg, (ho, wo)); yg.backward(cl(gy))
    close(yg, yr, 1e-4, 5e-6); close(xg.grad, xr.grad, 1e-4, 2e-5)


def test_sgd_and_ema_vs_golden(golden):
    from arco_amd import optim
    g = golden["g4_glue"]
    p = torch.nn.Parameter(torch.from_numpy(g["ema_q"]).cuda())
    opt = optim.SGDNesterov([p], lr=0.01, momentum=0.9, weight_decay=0.0001, nesterov=True)
    for it in range(3):
        opt.zero_grad()
        (p * torch.from_numpy(g["sgd_g"][it]).cuda()).sum().backward()
        opt.step()
        opt.param_groups[0]['lr'] = 0.01 * (1.0 - it / 30000) ** 0.9
        np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"sgd_p{it}"], rtol=1e-6, atol=1e-7)
    q = [torch.nn.Parameter(torch.from_numpy(g["ema_q"]).cuda())]
    k = [torch.nn.Parameter(torch.from_numpy(g["ema_k"]).cuda())]
    optim.EmaPair(q, k).update(0.99)
    np.testing.assert_allclose(k[0].detach().cpu().numpy(), g["ema_out"], rtol=1e-6, atol=1e-7)


def test_glue_functions_match_torch_expressions():
    """ops.split_batch / fold_residual / combine_terms (one launch each way) against the tensor expressions they replace:
    values and gradients."""
    from arco_amd import ops
    g = torch.Generator().manual_seed(11)
    # split_batch: views + one-buffer backward; a missing half's gradient is zero
    x = torch.randn(6, 4, 8, 8, generator=g).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    a, b = ops.split_batch(x, 2)
    assert a.data_ptr() == x.data_ptr() and tuple(a.shape) == (2, 4, 8, 8) and tuple(b.shape) == (4, 4, 8, 8)
    wa, wb = torch.randn(2, 4, 8, 8, generator=g).cuda(), torch.randn(4, 4, 8, 8, generator=g).cuda()
    ((a * wa).sum() + (b * wb).sum()).backward()
    assert torch.equal(x.grad, torch.cat((wa, wb)))
    x.grad = None
    a, b = ops.split_batch(x, 2)
    (b * wb).sum().backward()
    assert torch.equal(x.grad, torch.cat((torch.zeros_like(wa), wb)))
    # fold_residual: (W + I) split by columns, 2-D and 3-D 1x1 weights
    for ones in ((1, 1), (1, 1, 1)):
        n, c = 48, 32
        w = torch.randn(n, n, *ones, generator=g).cuda().requires_grad_(True)
        lo, hi = ops.fold_residual(w, c)
        ref = w.detach().view(n, n) + torch.eye(n, device="cuda")
        assert tuple(lo.shape) == (n, c) + ones and torch.equal(lo.view(n, c), ref[:, :c]) and torch.equal(hi.view(n, n - c), ref[:, c:])
        glo, ghi = torch.randn_like(lo), torch.randn_like(hi)
        ((lo * glo).sum() + (hi * ghi).sum()).backward()
        assert torch.equal(w.grad.view(n, n), torch.cat((glo.view(n, c), ghi.view(n, n - c)), dim=1))
    # combine_terms
    ts = [torch.randn((), generator=g).cuda().requires_grad_(True) for _ in range(5)]
    ws = [0.01, 1.0, 1.0, 1.0, 0.5]
    out = ops.combine_terms(ws, ts)
    ref = sum(w * t.detach().double() for w, t in zip(ws, ts))
    np.testing.assert_allclose(float(out.detach()), float(ref), rtol=1e-6)
    (out * 3.0).backward()
    for w, t in zip(ws, ts):
        np.testing.assert_allclose(float(t.grad), 3.0 * w, rtol=1e-6)


def test_zero_rows():
    from arco_amd import _lib as L
    buf = torch.ones(100, 16, device="cuda")
    idx = torch.tensor([3, 7, 7, 99], dtype=torch.int64, device="cuda")
    L.call("arco_zero_rows", L.ptr(buf), 16, 16, L.ptr(idx), 4)
    ref = torch.ones(100, 16); ref[[3, 7, 99]] = 0
    assert torch.equal(buf.cpu(), ref)
