"""The default matrix-core mode (--conv_mma f32x3: every fp32 operand split exactly into three bf16 terms, six
v_mfma_f32_16x16x32_bf16 per 32 k, fp32 accumulate) against the native fp32 MFMA (--conv_mma f32) and an fp64 reference:
the split mode must be as accurate as fp32 arithmetic itself - its error against fp64 within a small factor of the native
fp32 kernel's error and far inside the 1e-3 budget of BASELINE.json's north_star (-m gpu)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cl(x):
    return x.cuda().contiguous(memory_format=torch.channels_last if x.dim() == 4 else torch.channels_last_3d)


def _run(mode, x, wt, gy):
    from arco_amd import ops
    prev = ops.CONV_MMA
    ops.CONV_MMA = mode
    try:
        xg, wg = _cl(x).requires_grad_(True), wt.cuda().requires_grad_(True)
        y = ops.conv(xg, wg, None)
        y.backward(_cl(gy))
        return y.detach().cpu().double(), xg.grad.detach().cpu().double(), wg.grad.detach().cpu().double()
    finally:
        ops.CONV_MMA = prev


@pytest.mark.parametrize("shape", [dict(nb=2, ci=64, co=64, sp=(32, 32), k=3), dict(nb=1, ci=128, co=256, sp=(16, 16), k=3),
                                   dict(nb=1, ci=496, co=496, sp=(32, 64), k=1), dict(nb=2, ci=48, co=80, sp=(24, 40), k=1),
                                   dict(nb=1, ci=16, co=32, sp=(6, 20, 24), k=3), dict(nb=1, ci=240, co=240, sp=(4, 12, 16), k=1),
                                   dict(nb=1, ci=64, co=64, sp=(5, 14, 10), k=3)])
@pytest.mark.parametrize("data", ["normal", "wide_range"])
def test_split_bf16_is_fp32_accurate(shape, data):
    from arco_amd import _lib as L
    rs = np.random.RandomState(shape["ci"] + shape["co"])
    nd = len(shape["sp"])
    kk = (shape["k"],) * nd
    x = rs.standard_normal((shape["nb"], shape["ci"], *shape["sp"])).astype(np.float32)
    wt = (rs.standard_normal((shape["co"], shape["ci"], *kk)) / np.sqrt(shape["ci"] * shape["k"] ** nd)).astype(np.float32)
    gy = rs.standard_normal((shape["nb"], shape["co"], *shape["sp"])).astype(np.float32)
    if data == "wide_range":          # six decades of magnitudes and exact zeros: nothing may depend on operand scale
        x = x * (10.0 ** rs.uniform(-3, 3, size=x.shape)).astype(np.float32)
        x[rs.uniform(size=x.shape) < 0.2] = 0.0
        wt = wt * (10.0 ** rs.uniform(-2, 2, size=wt.shape)).astype(np.float32)
    x, wt, gy = torch.from_numpy(x), torch.from_numpy(wt), torch.from_numpy(gy)
    conv = F.conv2d if nd == 2 else F.conv3d
    x64, w64 = x.double().requires_grad_(True), wt.double()
    y64 = conv(x64, w64, None, padding=shape["k"] // 2)
    y64.backward(gy.double())
    ld = shape["ci"]
    taps = shape["k"] ** nd
    nbd = shape["nb"] * (shape["sp"][0] if nd == 3 else 1)
    assert L.query("arco_conv_split_ok", taps, nbd, shape["sp"][-2], shape["sp"][-1], shape["ci"], shape["co"], ld) == 1
    y3, dx3, dw3 = _run(3, x, wt, gy)
    y0, dx0, dw0 = _run(0, x, wt, gy)
    # weight gradient (3x3 / 3x3x3: the split-bf16 kernel with pixel-major operands; 1x1: fp32 MFMA in both modes)
    w64g = w64.clone().requires_grad_(True)
    conv(x64.detach(), w64g, None, padding=shape["k"] // 2).backward(gy.double())
    scale_dw = torch.autograd.grad(conv(x64.detach().abs(), w64g, None, padding=shape["k"] // 2), w64g, gy.double().abs())[0]
    ew3 = float(((dw3 - w64g.grad).abs() / (scale_dw + 1e-30)).max())
    ew0 = float(((dw0 - w64g.grad).abs() / (scale_dw + 1e-30)).max())
    assert ew3 < 3e-6 and ew3 <= 1.5 * ew0 + 6e-8, (ew3, ew0)
    # error measured against the magnitude a term-by-term fp32 evaluation can resolve: sum |x||w| per output
    scale_y = conv(x64.detach().abs(), w64.abs(), None, padding=shape["k"] // 2)
    scale_dx = torch.autograd.grad(conv(x64, w64.abs(), None, padding=shape["k"] // 2), x64, gy.double().abs())[0]
    for got3, got0, ref, sc in ((y3, y0, y64.detach(), scale_y), (dx3, dx0, x64.grad, scale_dx)):
        e3 = float(((got3 - ref).abs() / (sc + 1e-30)).max())
        e0 = float(((got0 - ref).abs() / (sc + 1e-30)).max())
        assert e3 < 3e-6, (e3, e0)                      # fp32-level: a few ulp relative to sum |x||w| (grows ~sqrt(K) like any fp32 sum)
        assert e3 <= 1.5 * e0 + 6e-8, (e3, e0)          # never meaningfully worse than the native fp32 MFMA (measured: equal or better)
    assert not torch.equal(y3, y0)                      # the two modes really are different kernels


def test_split_weights_pack_is_an_exact_decomposition():
    """The split pack format holds x0 + x1 + x2 == x bit for bit (three bf16 terms of every weight)."""
    from arco_amd import _lib as L
    rs = np.random.RandomState(0)
    co, ci, taps = 40, 24, 9
    w = torch.from_numpy((rs.standard_normal((co, ci, 3, 3)) * 10.0 ** rs.uniform(-4, 4, size=(co, ci, 3, 3))).astype(np.float32)).cuda()
    npad, kp32 = 48, 32
    sp = torch.zeros((taps, npad, kp32 * 3 // 2), dtype=torch.float32, device="cuda")
    L.call("arco_pack_conv_weight", L.ptr(w), co, ci, taps, 2, L.ptr(sp))
    torch.cuda.synchronize()
    planes = sp.view(torch.int16).view(taps, npad, kp32 // 16, 3, 16).cpu()
    f = (planes.to(torch.int32) << 16).view(torch.float32).double()              # bf16 bits -> fp32 value
    rec = f.sum(dim=3).permute(0, 1, 2, 3).reshape(taps, npad, kp32)             # x0 + x1 + x2
    exp = torch.zeros((taps, npad, kp32), dtype=torch.float64)
    exp[:, :co, :ci] = w.cpu().double().reshape(co, ci, taps).permute(2, 0, 1)
    assert torch.equal(rec, exp)


@pytest.mark.parametrize("shape", [dict(nb=16, ci=64, co=64, s=64), dict(nb=12, ci=32, co=128, s=64), dict(nb=3, ci=16, co=64, s=128),
                                   dict(nb=17, ci=128, co=64, s=64), dict(nb=16, ci=32, co=32, s=128), dict(nb=5, ci=16, co=32, s=256),
                                   dict(nb=16, ci=128, co=128, s=32), dict(nb=16, ci=256, co=256, s=16), dict(nb=8, ci=48, co=96, s=64),
                                   dict(nb=12, ci=64, co=64, s=32), dict(nb=16, ci=32, co=16, s=256), dict(nb=8, ci=32, co=16, s=128),
                                   dict(nb=16, ci=32, co=32, s=256), dict(nb=7, ci=16, co=32, s=128)])
def test_pipelined_3x3_kernel_equals_igemm_kernel(shape):
    """conv_sp.hip (persistent workgroups of 4 MFMA + 4 loader waves, LDS-DMA weight ring, software-pipelined fragment
    reads) computes the same products in the same order as igemm_kernel<9,..,MMA=3>: outputs bit-identical, BN partial
    sums equal to fp32 rounding of a different slab partition, both within fp32 rounding of an fp64 convolution.  Covered:
    several tiles per workgroup (nb=17: 272 > 256 tiles, ragged; odd and even chunk counts per workgroup), one chunk
    (K = 16), 32- and 64-channel blocks, two to four N-blocks, the 8- and 4-row tiles of the deep levels, N = 96 (32-blocks),
    and the resident-weights kernel of the shallow levels (conv3x3_rw_kernel: K <= 32, N = 16 / 32; one and two chunks, odd
    and even chunk counts, 16+ tiles per workgroup)."""
    from arco_amd import _lib as L, ops
    nb, ci, co, s = shape["nb"], shape["ci"], shape["co"], shape["s"]
    g = torch.Generator().manual_seed(ci * 1000 + co)
    x = _cl(torch.randn(nb, ci, s, s, generator=g))
    wt = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).cuda()
    bias = torch.randn(co, generator=g).cuda()
    wp = ops.pack_weight(wt, 9, 0)
    xr, ldx = ops.rows_view(x)
    res = {}
    prev = ops.conv_sp_set(1)
    try:
        for on in (0, 1):
            ops.conv_sp_set(on)
            cfg = L.query("arco_conv_config_mma", 9, nb, s, s, ci, co, ldx, 3)
            assert (9300000 <= cfg < 9400000) == bool(on), cfg
            out, (ssum, ssq, nmb) = ops.conv_raw(xr, ldx, ci, wp, co, nb, s, s, 9, bias=bias, stats=True)
            assert ssum.shape == (co, nmb)
            res[on] = (out.clone(), ssum.double().sum(1), ssq.double().sum(1))
    finally:
        ops.conv_sp_set(prev)
    assert torch.equal(res[0][0], res[1][0])
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    assert float((res[1][0].double() - ref).abs().max() / ref.abs().max()) < 3e-6
    for k in (1, 2):
        assert torch.allclose(res[0][k], res[1][k], rtol=1e-5, atol=1e-6)
    assert torch.allclose(res[1][1], ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("shape", [dict(nb=16, s=256), dict(nb=8, s=128), dict(nb=3, s=256), dict(nb=16, s=256, co=4), dict(nb=7, s=128, co=8),
                                   dict(nb=16, s=256, ci=4), dict(nb=6, s=128, ci=4, co=4)])
def test_narrow_planes_run_on_the_resident_weights_kernel(shape):
    """At most 16 outputs from at most 16 inputs at 256^2 / 128^2 (16 -> 16, the 16 -> 4 logits layer, its 4 -> 16 data
    gradient): one K chunk, one N block on conv3x3_rw_kernel<8,1> (32-row tiles; missing input channels staged as zeros,
    lane groups beyond N idle in the epilogue); with the pipelined kernels off the fp32 halo / few-channel kernels.  Both
    within fp32 rounding of an fp64 convolution, BN partial sums equal up to the slab partition."""
    from arco_amd import _lib as L, ops
    nb, s, ci, co = shape["nb"], shape["s"], shape.get("ci", 16), shape.get("co", 16)
    g = torch.Generator().manual_seed(1616 + nb + ci * 7 + co)
    x = _cl(torch.randn(nb, ci, s, s, generator=g))
    wt = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).cuda()
    bias = torch.randn(co, generator=g).cuda()
    wp = ops.pack_weight(wt, 9, 0)
    xr, ldx = ops.rows_view(x)
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    res = {}
    prev = ops.conv_sp_set(1)
    try:
        for on in (0, 1):
            ops.conv_sp_set(on)
            ops._cfg_cache.clear()
            split = ops._split_ok(9, nb, s, s, ci, co, ldx)
            assert split == bool(on)
            cfg = L.query("arco_conv_config_mma", 9, nb, s, s, ci, co, ldx, 3 if split else 0)
            assert (cfg == 9358016) == bool(on), cfg
            out, (ssum, ssq, nmb) = ops.conv_raw(xr, ldx, ci, wp, co, nb, s, s, 9, bias=bias, stats=True)
            assert out.shape[1] == co and ssum.shape == (co, nmb)
            res[on] = (out.clone(), ssum.double().sum(1), ssq.double().sum(1))
            assert float((res[on][0].double() - ref).abs().max() / ref.abs().max()) < 3e-6
    finally:
        ops.conv_sp_set(prev)
        ops._cfg_cache.clear()
    for k in (1, 2):
        assert torch.allclose(res[0][k], res[1][k], rtol=1e-5, atol=1e-5 * nb * s * s / 1024)
    assert torch.allclose(res[1][1], ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)


def test_pipelined_gemm_is_bit_identical_to_the_implicit_gemm_kernel():
    """gemm_sp_kernel (csrc/gemm_sp.hip: persistent workgroups, loader / MFMA wave roles, 64 x 256 tiles) computes every product of
    igemm_kernel<1,...,MMA=3> in the same order: outputs equal BIT FOR BIT on interior, ragged (M, N, K edges) and residual / bias
    shapes; both agree with a float64 GEMM to split-bf16 accuracy.  (model_2D.py:20-55, train_arco_2d.py:231-234 at production widths.)"""
    from arco_amd import ops, _lib as L
    prev_mma = ops.CONV_MMA
    ops.CONV_MMA = 3
    torch.manual_seed(3)
    try:
        for (M, K, N, res, bias) in ((8192, 496, 496, False, False), (4100, 480, 480, True, False), (3001, 100, 252, True, True),
                                     (16384, 32, 224, True, False), (6000, 240, 16, False, False), (5000, 448, 448, False, True)):
            x = torch.randn(M, K, device="cuda") * torch.exp(torch.randn(M, 1, device="cuda"))
            w = torch.randn(N, K, 1, 1, device="cuda") / K ** 0.5
            r = torch.randn(M, N, device="cuda") if res else None
            b = torch.randn(N, device="cuda") if bias else None
            outs = []
            for on in (0, 1):
                L.query("arco_gemm_sp_set", on, 1)
                ops._cfg_cache.clear()
                wp = ops.pack_weight(w, 1, 0)
                o, _ = ops.conv_raw(x, K, K, wp, N, 1, 1, M, 1, bias=b, residual=r, ld_res=N if res else 0)
                outs.append(o.permute(0, 2, 3, 1).reshape(M, N).clone())
            assert torch.equal(outs[0], outs[1]), (M, K, N)
            ref = x.double() @ w.view(N, K).double().t() + (r.double() if res else 0) + (b.double() if bias else 0)
            scale = (x.double().abs() @ w.view(N, K).double().abs().t()).max()
            assert float((outs[1].double() - ref).abs().max() / scale) < 2e-6, (M, K, N)
    finally:
        L.query("arco_gemm_sp_set", 1, 2048)
        ops._cfg_cache.clear()
        ops.CONV_MMA = prev_mma


@pytest.mark.parametrize("shape", [dict(nb=16, c=16, sp=(256, 256)), dict(nb=16, c=64, sp=(64, 64)), dict(nb=4, c=32, sp=(128, 128)),
                                   dict(nb=2, c=32, sp=(12, 56, 40)), dict(nb=2, c=64, sp=(9, 28, 20)), dict(nb=1, c=16, sp=(8, 32, 32)),
                                   dict(nb=1, c=496, sp=(64, 64), k=1)])
def test_activation_split_is_an_exact_decomposition_through_the_kernels(shape):
    """The loaders split every activation into three bf16 terms (round 6: the first by round-to-nearest, the second by truncation, the
    third the exact rest - igemm_args.h).  x = t0 + t1 + t2 must hold bit for bit: a convolution whose weights are the identity on the
    centre tap (1, 0, 0 in the weight terms) accumulates exactly (t2 + t1) + t0 per output and has to hand the input back unchanged -
    on six decades of magnitudes, exact zeros and negative values, through the pipelined 2-D kernels (resident weights, LDS-DMA ring),
    the 3x3x3 flat-tile and plane-ring kernels and the wide 1x1 GEMM."""
    from arco_amd import ops
    rs = np.random.RandomState(shape["c"])
    nd, k = len(shape["sp"]), shape.get("k", 3)
    x = rs.standard_normal((shape["nb"], shape["c"], *shape["sp"])).astype(np.float32)
    x = x * (10.0 ** rs.uniform(-3, 3, size=x.shape)).astype(np.float32)
    x[rs.uniform(size=x.shape) < 0.1] = 0.0
    wt = np.zeros((shape["c"], shape["c"]) + (k,) * nd, dtype=np.float32)
    idx = (np.arange(shape["c"]), np.arange(shape["c"])) + (k // 2,) * nd
    wt[idx] = 1.0
    prev = ops.CONV_MMA
    ops.CONV_MMA = 3
    try:
        xg = _cl(torch.from_numpy(x))
        with torch.no_grad():
            y = ops.conv(xg, torch.from_numpy(wt).cuda(), None)
    finally:
        ops.CONV_MMA = prev
    assert torch.equal(y, xg)
