"""Order-independent row scatter (csrc/det_scatter.hip) through the C ABI: against a float64 restatement of the adjoint it replaces
(the backward of `F.interpolate(lo, mode='trilinear', align_corners=True)[rows]` and of `hi[rows]`: model_3D.py:52-55 at the sampled
voxels), against the fp32-atomics kernel it stands in for, and bit-identical over repeated executions with many collisions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(seed, n, lo_sp, hi_sp, c_lo, c_hi, nb=2, dup=8):
    g = torch.Generator().manual_seed(seed)
    vol = hi_sp[0] * hi_sp[1] * hi_sp[2]
    pix = torch.randint(0, nb * vol, (n // dup,), generator=g)
    pix = pix.repeat(dup)[torch.randperm(n // dup * dup, generator=g)]          # every voxel drawn `dup` times: collisions everywhere
    dX = torch.randn(pix.numel(), c_lo + c_hi, generator=g) * torch.logspace(-6, 2, pix.numel()).unsqueeze(1)   # 8 decades of row norms
    return pix.cuda(), dX.cuda().contiguous()


def _run(pix, dX, lo_sp, hi_sp, c_lo, c_hi, nb, det):
    from arco_amd import _lib as L, head
    n = int(pix.shape[0])
    k = c_lo + c_hi
    dlo = torch.zeros((nb, *lo_sp, c_lo), dtype=torch.float32, device="cuda")
    dhi = torch.zeros((nb, *hi_sp, c_hi), dtype=torch.float32, device="cuda")
    if det:
        idx8 = torch.empty(8 * n, dtype=torch.int64, device="cuda")
        w8 = torch.empty(8 * n, dtype=torch.float32, device="cuda")
        L.call("arco_corner_rows3d", L.ptr(pix), n, *lo_sp, *hi_sp, L.ptr(idx8), L.ptr(w8))
        head._det_scatter_rows(dX, k, c_lo, 8, idx8, w8, 8 * n, dlo, c_lo)
        head._det_scatter_rows(dX[:, c_lo:], k, c_hi, 1, pix, None, n, dhi, c_hi)
    else:
        L.call("arco_scatter_upcat_rows3d", L.ptr(dX), k, L.ptr(pix), n, L.ptr(dlo), c_lo, c_lo, *lo_sp, L.ptr(dhi), c_hi, c_hi, *hi_sp)
    torch.cuda.synchronize()
    return dlo, dhi


def _f64(pix, dX, lo_sp, hi_sp, c_lo, c_hi, nb):
    """the adjoint by autograd in float64 on the CPU"""
    lo = torch.zeros((nb, c_lo, *lo_sp), dtype=torch.float64, requires_grad=True)
    hi = torch.zeros((nb, c_hi, *hi_sp), dtype=torch.float64, requires_grad=True)
    up = torch.nn.functional.interpolate(lo, size=hi_sp, mode="trilinear", align_corners=True)
    rows = torch.cat([up, hi], 1).movedim(1, -1).reshape(-1, c_lo + c_hi)[pix.cpu()]
    (rows * dX.cpu().double()).sum().backward()
    return lo.grad.movedim(1, -1).contiguous(), hi.grad.movedim(1, -1).contiguous()


@pytest.mark.parametrize("shape", [((5, 6, 4), (10, 12, 8), 32, 16), ((7, 5, 3), (13, 9, 5), 224, 16), ((4, 4, 4), (4, 4, 4), 8, 4)])
def test_det_scatter_is_the_adjoint_and_bit_reproducible(shape):
    lo_sp, hi_sp, c_lo, c_hi = shape
    nb = 2
    pix, dX = _case(1, 4096, lo_sp, hi_sp, c_lo, c_hi, nb)
    glo, ghi = _f64(pix, dX, lo_sp, hi_sp, c_lo, c_hi, nb)
    dlo, dhi = _run(pix, dX, lo_sp, hi_sp, c_lo, c_hi, nb, det=True)
    alo, ahi = _run(pix, dX, lo_sp, hi_sp, c_lo, c_hi, nb, det=False)
    s_lo, s_hi = float(glo.abs().max()), float(ghi.abs().max())
    # fixed point: 2^-44 of the largest source element per contribution; fp32 sums: 2^-24 of the partial sums
    e_det = max(float((dlo.cpu().double() - glo).abs().max()) / s_lo, float((dhi.cpu().double() - ghi).abs().max()) / s_hi)
    e_atm = max(float((alo.cpu().double() - glo).abs().max()) / s_lo, float((ahi.cpu().double() - ghi).abs().max()) / s_hi)
    print("error against float64 / largest element: fixed-point", e_det, " fp32 atomics", e_atm)
    assert e_det <= 4e-7 and e_atm <= 2e-6
    # elementwise (the interpolation weights are fp32 in the kernel, float64 in the restatement: 1e-7 of the largest TERM of a sum)
    np.testing.assert_allclose(dlo.cpu().double().numpy(), glo.numpy(), rtol=1e-5, atol=2e-7 * s_lo)
    np.testing.assert_allclose(dhi.cpu().double().numpy(), ghi.numpy(), rtol=1e-6, atol=1e-9 * s_hi)
    for _ in range(20):
        d2, h2 = _run(pix, dX, lo_sp, hi_sp, c_lo, c_hi, nb, det=True)
        assert torch.equal(d2, dlo) and torch.equal(h2, dhi)
    from arco_amd import head
    for acc in head._ACC64.values():                  # the accumulators are left zero
        assert int(acc.abs().max()) == 0


def test_det_scatter_nonfinite_source_reaches_the_overflow_guard_and_zero_source_is_zero():
    """An inf / nan in the source rows (an overflowed loss-scaled gradient) must arrive at the destination: train_arco_3d's
    _unscale_and_guard skips the step on it.  An all-zero source leaves the destination zero."""
    lo_sp, hi_sp, c_lo, c_hi, nb = (4, 4, 4), (8, 8, 8), 16, 8, 1
    pix, dX = _case(3, 512, lo_sp, hi_sp, c_lo, c_hi, nb)
    bad = dX.clone()
    bad[17, 3] = float("inf")
    bad[99, c_lo + 2] = float("nan")
    dlo, dhi = _run(pix, bad, lo_sp, hi_sp, c_lo, c_hi, nb, det=True)
    assert not bool(torch.isfinite(dlo).all()) and not bool(torch.isfinite(dhi).all())
    z = torch.zeros_like(dX)
    dlo, dhi = _run(pix, z, lo_sp, hi_sp, c_lo, c_hi, nb, det=True)
    assert float(dlo.abs().max()) == 0.0 and float(dhi.abs().max()) == 0.0
    dlo, dhi = _run(pix, dX, lo_sp, hi_sp, c_lo, c_hi, nb, det=True)      # and the accumulators were not poisoned
    assert bool(torch.isfinite(dlo).all()) and float(dlo.abs().max()) > 0


@pytest.mark.parametrize("shape", [((8, 8), (16, 16), 64, 16), ((5, 7), (11, 9), 384, 64)])
def test_det_scatter_2d_is_the_bilinear_adjoint_and_bit_reproducible(shape, monkeypatch):
    """The 2-D heads' adjoints (head._scatter_upcat2d / _lerp4_cat_rows_bwd with ARCO_DET_SCATTER=2) against float64 autograd of
    `F.interpolate(lo, mode='bilinear', align_corners=True)[rows]` and `hi[rows]` (model_2D.py:43-50), and against the fp32 atomics."""
    from arco_amd import head, _lib as L
    lo_sp, hi_sp, c_lo, c_hi = shape
    nb, n = 2, 2048
    g = torch.Generator().manual_seed(5)
    pix = torch.randint(0, nb * hi_sp[0] * hi_sp[1], (n // 8,), generator=g).repeat(8)[torch.randperm(n, generator=g)].cuda()
    dX = (torch.randn(n, c_lo + c_hi, generator=g) * torch.logspace(-5, 1, n).unsqueeze(1)).cuda().contiguous()
    lo = torch.zeros((nb, c_lo, *lo_sp), dtype=torch.float64, requires_grad=True)
    hi = torch.zeros((nb, c_hi, *hi_sp), dtype=torch.float64, requires_grad=True)
    up = torch.nn.functional.interpolate(lo, size=hi_sp, mode="bilinear", align_corners=True)
    rows = torch.cat([up, hi], 1).movedim(1, -1).reshape(-1, c_lo + c_hi)[pix.cpu()]
    (rows * dX.cpu().double()).sum().backward()
    glo, ghi = lo.grad.movedim(1, -1), hi.grad.movedim(1, -1)
    res = {}
    for det in (0, 2):
        monkeypatch.setattr(head, "DET_SCATTER", det)
        outs = []
        for _ in range(10 if det else 1):
            dlo = torch.zeros((nb, *lo_sp, c_lo), dtype=torch.float32, device="cuda")
            dhi = torch.zeros((nb, *hi_sp, c_hi), dtype=torch.float32, device="cuda")
            head._scatter_upcat2d(dX, c_lo + c_hi, pix, n, dlo, c_lo, *lo_sp, dhi, c_hi, *hi_sp)
            torch.cuda.synchronize()
            outs.append((dlo, dhi))
        assert all(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) for o in outs)
        res[det] = outs[0]
        e = max(float((outs[0][0].cpu().double() - glo).abs().max() / glo.abs().max()), float((outs[0][1].cpu().double() - ghi).abs().max() / ghi.abs().max()))
        print("det" if det else "atomics", "error against float64 / largest element:", e)
        assert e <= (4e-7 if det else 2e-6)          # (the interpolation weights themselves are fp32 products: 1e-7 relative)
    # the hi part of the second-level adjoint alone (lerp4 form: Chi = 0 in the kernel, rows through the fixed-point path)
    monkeypatch.setattr(head, "DET_SCATTER", 2)
    lylx = torch.rand(n, 2, generator=g).cuda()
    dV = torch.empty((4 * n, c_lo), dtype=torch.float32, device="cuda")
    dhi = torch.zeros((nb, *hi_sp, c_hi), dtype=torch.float32, device="cuda")
    head._lerp4_cat_rows_bwd(dX, c_lo + c_hi, c_lo, lylx, pix, n, dV, dhi, c_hi)
    monkeypatch.setattr(head, "DET_SCATTER", 0)
    dV0 = torch.empty_like(dV)
    dhi0 = torch.zeros_like(dhi)
    head._lerp4_cat_rows_bwd(dX, c_lo + c_hi, c_lo, lylx, pix, n, dV0, dhi0, c_hi)
    torch.cuda.synchronize()
    assert torch.equal(dV, dV0) and torch.equal(dhi, res[2][1])
    np.testing.assert_allclose(dhi0.cpu().numpy(), dhi.cpu().numpy(), rtol=1e-5, atol=1e-6 * float(dhi.abs().max()))
