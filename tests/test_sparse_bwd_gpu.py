"""Row-sparse backward of the dense per-pixel layers (ops.ConvFn with 1x1 kernels on whole maps, the reference's dataflow:
model_2D.py:51-53, train_arco_2d.py:231-234): when the incoming gradient holds a few non-zero rows - the loss reads `rep` at sampled
rows only, loss_helper_3d.py:455-457 - the backward runs on those rows.  Against the dense backward of the same call."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(x, w, b, residual, dy, sparse):
    from arco_amd import ops
    ops.SPARSE_BWD = int(sparse)
    try:
        xx = x.clone().requires_grad_(True)
        ww = w.clone().requires_grad_(True)
        bb = b.clone().requires_grad_(True) if b is not None else None
        y = ops.conv(xx, ww, bb, residual=residual)
        y.backward(dy)
        torch.cuda.synchronize()
        return y.detach(), xx.grad, ww.grad, (bb.grad if bb is not None else None)
    finally:
        ops.SPARSE_BWD = 1


@pytest.mark.parametrize("case", [dict(nb=2, hw=(256, 256), ci=160, co=160, residual=True, bias=False, rows=300),
                                  dict(nb=2, hw=(256, 256), ci=192, co=128, residual=False, bias=True, rows=1000),
                                  dict(nb=1, hw=(40, 48, 40), ci=128, co=128, residual=True, bias=False, rows=257),      # 1x1x1 on a volume
                                  dict(nb=2, hw=(256, 256), ci=128, co=128, residual=True, bias=False, rows=0)])
def test_conv1x1_backward_on_the_nonzero_rows_equals_the_dense_backward(case):
    from arco_amd import ops
    g = torch.Generator().manual_seed(4)
    nb, hw, ci, co = case["nb"], case["hw"], case["ci"], case["co"]
    M = nb * int(np.prod(hw))
    x = ops.to_channels_last(torch.randn((nb, ci, *hw), generator=g).cuda())
    w = (torch.randn((co, ci) + (1,) * len(hw), generator=g) / ci ** 0.5).cuda()
    b = torch.randn(co, generator=g).cuda() if case["bias"] else None
    dy_rows = torch.zeros((M, co))
    pick = torch.randperm(M, generator=g)[:case["rows"]]
    dy_rows[pick] = torch.randn((case["rows"], co), generator=g) * torch.logspace(-4, 2, max(case["rows"], 1))[:case["rows"]].unsqueeze(1)
    dy = dy_rows.view(nb, *hw, co).movedim(-1, 1).cuda()
    before = dict(ops.sparse_bwd_stats)
    y1, dx1, dw1, db1 = _run(x, w, b, case["residual"], dy, sparse=1)
    assert ops.sparse_bwd_stats["sparse"] == before["sparse"] + 1
    y0, dx0, dw0, db0 = _run(x, w, b, case["residual"], dy, sparse=0)
    assert torch.equal(y0, y1)
    assert dx1.shape == dx0.shape and torch.equal(dx1, dx0)                      # a row's data gradient does not depend on the other rows
    s = float(dw0.abs().max())
    np.testing.assert_allclose(dw1.cpu().numpy(), dw0.cpu().numpy(), rtol=1e-4, atol=2e-6 * s if s else 0)      # fp32 summation order
    if b is not None:
        np.testing.assert_allclose(db1.cpu().numpy(), db0.cpu().numpy(), rtol=1e-5, atol=1e-6 * float(db0.abs().max()))
    # and against float64 on the CPU
    ref_dw = dy_rows.double().t() @ x.movedim(1, -1).reshape(M, ci).cpu().double()
    np.testing.assert_allclose(dw1.view(co, ci).cpu().double().numpy(), ref_dw.numpy(), rtol=1e-4, atol=1e-5 * max(float(ref_dw.abs().max()), 1e-30))


def test_dense_gradient_keeps_the_dense_backward_and_nonfinite_rows_survive():
    from arco_amd import ops
    g = torch.Generator().manual_seed(5)
    x = ops.to_channels_last(torch.randn((2, 128, 256, 256), generator=g).cuda())
    w = (torch.randn((128, 128, 1, 1), generator=g) / 11).cuda()
    dy = ops.to_channels_last(torch.randn((2, 128, 256, 256), generator=g).cuda())
    before = dict(ops.sparse_bwd_stats)
    _, dx1, dw1, _ = _run(x, w, None, False, dy, sparse=1)
    assert ops.sparse_bwd_stats["dense"] == before["dense"] + 1 and ops.sparse_bwd_stats["sparse"] == before["sparse"]
    _, dx0, dw0, _ = _run(x, w, None, False, dy, sparse=0)
    assert torch.equal(dx1, dx0) and torch.equal(dw1, dw0)
    # a row of nan / inf is a non-zero row: it reaches dx and dw as it does in the dense backward
    dys = torch.zeros_like(dy)
    dys[1, :, 7, 9] = float("nan")
    dys[0, 3, 100, 50] = float("inf")
    _, dx1, dw1, _ = _run(x, w, None, False, dys, sparse=1)
    assert bool(torch.isnan(dx1[1, :, 7, 9]).all()) and not bool(torch.isfinite(dx1[0, :, 100, 50]).all()) and not bool(torch.isfinite(dw1).all())
    assert float(dx1[0, :, 0, 0].abs().max()) == 0.0
