"""Row-sparse student head (trainer-internal fast path; results identical to the dense modules).

In the reference step the student representation
    rep = q_representation(FeatureExtractor(feature_map))          (train_arco_2d.py:317-330)
is a dense [B,496,256,256] tensor, but the loss only ever reads <= num_queries rows of it per
class (loss_helper_3d.py:455-457) and d loss / d rep is non-zero only on those rows.  Every op
between the 128x128 level of the FeatureExtractor and `rep` is per-pixel (1x1 convs) or a
4-neighbour bilinear lerp, so the sampled rows can be computed exactly from
    x3p = fea3(x)+x   [B,480,128,128]   (dense, cheap)      and    f4 [B,16,256,256]:
    X4[j] = cat(bilinear(x3p)[pix_j], f4[pix_j]);   A[j] = W2 . W1 . W4 . X4[j]
Row values are bit-identical to the dense path (same fp32 lerp order, same K-order MFMA
accumulation); the backward scatters dX4 rows back into x3p / f4 and the weight gradients are
GEMMs over the anchor rows only.  This removes ~3.3 TFLOP/step of dense fp32 GEMM work at
BASELINE config 2 without changing any result.
"""
import os

import torch

from . import _lib as L
from . import ops
from ._contrast import _ceil, rows_view


def _gemm(x, w):
    """x [n, K] @ w[N, K]^T on the MFMA conv kernel (1x1)."""
    n, k = int(x.shape[0]), int(x.shape[1])
    y, _ = ops.conv_raw(x, x.stride(0), k, ops.pack_weight(w, 1, 0), int(w.shape[0]), 1, 1, n, 1)
    return y.permute(0, 2, 3, 1).reshape(n, int(w.shape[0]))


def _gemm_t(dy, w):
    """dy [n, N] @ w[N, K]  (data gradient of _gemm)."""
    n, nn_ = int(dy.shape[0]), int(dy.shape[1])
    y, _ = ops.conv_raw(dy, dy.stride(0), nn_, ops.pack_weight(w, 1, 1), int(w.shape[1]), 1, 1, n, 1)
    return y.permute(0, 2, 3, 1).reshape(n, int(w.shape[1]))


def _wgrad(dy, x, like):
    n = int(dy.shape[0])
    return ops.conv_wgrad(dy, dy.stride(0), int(dy.shape[1]), x, x.stride(0), int(x.shape[1]), 1, 1, 1, n, like)


class LazyHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x3p, f4, w4, w1, w2, pix):
        lo, ldlo = rows_view(x3p)
        hi, ldhi = rows_view(f4)
        nb, clo, hi_h, hi_w = int(x3p.shape[0]), int(x3p.shape[1]), int(x3p.shape[2]), int(x3p.shape[3])
        chi, ho, wo = int(f4.shape[1]), int(f4.shape[2]), int(f4.shape[3])
        n = int(pix.shape[0])
        X = torch.empty((n, clo + chi), dtype=torch.float32, device=x3p.device)
        L.call("arco_gather_upcat_rows", L.ptr(lo), ldlo, clo, hi_h, hi_w, L.ptr(hi), ldhi, chi, ho, wo, L.ptr(pix), n,
               L.ptr(X), clo + chi)
        h0 = _gemm(X, w4)
        h1 = _gemm(h0, w1)
        a = _gemm(h1, w2)
        ctx.save_for_backward(X, h0, h1, w4, w1, w2, pix)
        ctx.geom = (nb, clo, hi_h, hi_w, chi, ho, wo)
        return a

    @staticmethod
    def backward(ctx, da):
        X, h0, h1, w4, w1, w2, pix = ctx.saved_tensors
        nb, clo, hi_h, hi_w, chi, ho, wo = ctx.geom
        da = da.contiguous()
        dw2 = _wgrad(da, h1, w2)
        dh1 = _gemm_t(da, w2)
        dw1 = _wgrad(dh1, h0, w1)
        dh0 = _gemm_t(dh1, w1)
        dw4 = _wgrad(dh0, X, w4)
        dX = _gemm_t(dh0, w4)
        dlo = torch.zeros((nb, hi_h, hi_w, clo), dtype=torch.float32, device=da.device)
        dhi = torch.zeros((nb, ho, wo, chi), dtype=torch.float32, device=da.device)
        _scatter_upcat2d(dX, clo + chi, pix, int(pix.shape[0]), dlo, clo, hi_h, hi_w, dhi, chi, ho, wo)
        return dlo.permute(0, 3, 1, 2), dhi.permute(0, 3, 1, 2), dw4, dw1, dw2, None


def lazy_head(x3p, f4, fea4_weight, q1_weight, q2_weight, pix):
    """Anchor rows of q_representation(FeatureExtractor(...)) at high-res pixel ids `pix`."""
    return LazyHeadFn.apply(x3p, f4, fea4_weight, q1_weight, q2_weight, pix)


def _rows2d_forward(x2p, f3, f4, w3, w4, pix):
    """Two-level row path: X3 = cat(bilinear(x2p), f3) on the 4 low-res neighbours of each anchor, X3p = fea3(X3)+X3,
    X4 = cat(4-way lerp of X3p, f4[pix]); returns (X3, X4, nb4, lylx, fea4(X4))."""
    dev = x2p.device
    n = int(pix.shape[0])
    nb, c2, h2, w2_ = (int(v) for v in x2p.shape)
    c3, h3, w3_ = int(f3.shape[1]), int(f3.shape[2]), int(f3.shape[3])
    c4, h4, w4_ = int(f4.shape[1]), int(f4.shape[2]), int(f4.shape[3])
    lo2, ld2 = rows_view(x2p)
    r3, ld3 = rows_view(f3)
    r4, ld4 = rows_view(f4)
    nb4 = torch.empty(4 * n, dtype=torch.int64, device=dev)
    lylx = torch.empty(2 * n, dtype=torch.float32, device=dev)
    L.call("arco_up_neighbors", L.ptr(pix), n, h3, w3_, h4, w4_, L.ptr(nb4), L.ptr(lylx))
    k3 = c2 + c3
    X3 = torch.empty((4 * n, k3), dtype=torch.float32, device=dev)
    L.call("arco_gather_upcat_rows", L.ptr(lo2), ld2, c2, h2, w2_, L.ptr(r3), ld3, c3, h3, w3_, L.ptr(nb4), 4 * n,
           L.ptr(X3), k3)
    y3, _ = ops.conv_raw(X3, k3, k3, ops.pack_weight(w3, 1, 0), k3, 1, 1, 4 * n, 1, residual=X3, ld_res=k3)
    X3p = y3.permute(0, 2, 3, 1).reshape(4 * n, k3)                     # fea3(x)+x rows
    k4 = k3 + c4
    X4 = torch.empty((n, k4), dtype=torch.float32, device=dev)
    L.call("arco_lerp4_cat_rows", L.ptr(X3p), k3, k3, L.ptr(lylx), L.ptr(r4), ld4, c4, L.ptr(pix), n, L.ptr(X4), k4)
    return X3, X4, nb4, lylx, _gemm(X4, w4)


class LazyHead2Fn(torch.autograd.Function):
    """Two-level row-sparse head: also fea3 (the 128x128 level) is evaluated only on the <= 4 low-res
    neighbours of each anchor.  Inputs: x2p = fea2(x)+x [B,448,64,64] (dense), f3 [B,32,128,128],
    f4 [B,16,256,256].  Rows are bit-identical to the dense modules (same lerp order, same K-order MFMA
    accumulation); d loss/d x2p becomes dense again from the 64x64 level down."""

    @staticmethod
    def forward(ctx, x2p, f3, f4, w3, w4, w1, w2, pix):
        nb, c2, h2, w2_ = (int(v) for v in x2p.shape)
        c3, h3, w3_ = int(f3.shape[1]), int(f3.shape[2]), int(f3.shape[3])
        c4, h4, w4_ = int(f4.shape[1]), int(f4.shape[2]), int(f4.shape[3])
        X3, X4, nb4, lylx, h0 = _rows2d_forward(x2p, f3, f4, w3, w4, pix)
        h1 = _gemm(h0, w1)
        a = _gemm(h1, w2)
        ctx.save_for_backward(X3, X4, h0, h1, w3, w4, w1, w2, pix, nb4, lylx)
        ctx.geom = (nb, c2, h2, w2_, c3, h3, w3_, c4, h4, w4_)
        return a

    @staticmethod
    def backward(ctx, da):
        X3, X4, h0, h1, w3, w4, w1, w2, pix, nb4, lylx = ctx.saved_tensors
        nb, c2, h2, w2_, c3, h3, w3_, c4, h4, w4_ = ctx.geom
        dev = da.device
        n = int(pix.shape[0])
        k3 = c2 + c3
        da = da.contiguous()
        dw2 = _wgrad(da, h1, w2)
        dh1 = _gemm_t(da, w2)
        dw1 = _wgrad(dh1, h0, w1)
        dh0 = _gemm_t(dh1, w1)
        dw4 = _wgrad(dh0, X4, w4)
        dX4 = _gemm_t(dh0, w4)
        dX3p = torch.empty((4 * n, k3), dtype=torch.float32, device=dev)
        df4 = torch.zeros((nb, h4, w4_, c4), dtype=torch.float32, device=dev)
        _lerp4_cat_rows_bwd(dX4, k3 + c4, k3, lylx, pix, n, dX3p, df4, c4)
        dw3 = _wgrad(dX3p, X3, w3)
        # d(fea3(x)+x)/dx: W3^T dy + dy  (residual fused in the dgrad GEMM epilogue)
        y, _ = ops.conv_raw(dX3p, k3, k3, ops.pack_weight(w3, 1, 1), k3, 1, 1, 4 * n, 1, residual=dX3p, ld_res=k3)
        dX3 = y.permute(0, 2, 3, 1).reshape(4 * n, k3)
        dx2p = torch.zeros((nb, h2, w2_, c2), dtype=torch.float32, device=dev)
        df3 = torch.zeros((nb, h3, w3_, c3), dtype=torch.float32, device=dev)
        _scatter_upcat2d(dX3, k3, nb4, 4 * n, dx2p, c2, h2, w2_, df3, c3, h3, w3_)
        return (dx2p.permute(0, 3, 1, 2), df3.permute(0, 3, 1, 2), df4.permute(0, 3, 1, 2), dw3, dw4, dw1, dw2, None)


def lazy_head2(x2p, f3, f4, fea3_weight, fea4_weight, q1_weight, q2_weight, pix):
    return LazyHead2Fn.apply(x2p, f3, f4, fea3_weight, fea4_weight, q1_weight, q2_weight, pix)


def _fea_rows(X, w, mode):
    """fea_i(X) + X on rows (mode 0) / its data gradient W^T dY + dY (mode 1): the 1x1 conv with the residual in the GEMM epilogue."""
    k = int(X.shape[1])
    y, _ = ops.conv_raw(X, k, k, ops.pack_weight(w, 1, mode), k, 1, 1, int(X.shape[0]), 1, residual=X, ld_res=k)
    return y.permute(0, 2, 3, 1).reshape(int(X.shape[0]), k)


def _rows3lvl_forward(x1p, f2, f3, f4, w2, w3, w4, pix):
    """Three-level row path: the 4 neighbours at f3's resolution of every anchor, the 4 neighbours at f2's resolution of
    each of those; X2 = cat(bilinear(x1p), f2) on the 16 n rows, X2p = fea2(X2)+X2, X3 = cat(4-way lerp of X2p, f3 rows),
    X3p = fea3(X3)+X3, X4 = cat(4-way lerp of X3p, f4[pix]).  Returns (X2, X3, X4, nb4, nb16, lylx3, lylx4, fea4(X4))."""
    dev = x1p.device
    n = int(pix.shape[0])
    nb, c1, h1, w1_ = (int(v) for v in x1p.shape)
    c2, h2, w2_ = int(f2.shape[1]), int(f2.shape[2]), int(f2.shape[3])
    c3, h3, w3_ = int(f3.shape[1]), int(f3.shape[2]), int(f3.shape[3])
    c4, h4, w4_ = int(f4.shape[1]), int(f4.shape[2]), int(f4.shape[3])
    lo1, ld1 = rows_view(x1p)
    r2, ld2 = rows_view(f2)
    r3, ld3 = rows_view(f3)
    r4, ld4 = rows_view(f4)
    nb4 = torch.empty(4 * n, dtype=torch.int64, device=dev)
    lylx4 = torch.empty(2 * n, dtype=torch.float32, device=dev)
    L.call("arco_up_neighbors", L.ptr(pix), n, h3, w3_, h4, w4_, L.ptr(nb4), L.ptr(lylx4))
    nb16 = torch.empty(16 * n, dtype=torch.int64, device=dev)
    lylx3 = torch.empty(8 * n, dtype=torch.float32, device=dev)
    L.call("arco_up_neighbors", L.ptr(nb4), 4 * n, h2, w2_, h3, w3_, L.ptr(nb16), L.ptr(lylx3))
    k2, k3, k4 = c1 + c2, c1 + c2 + c3, c1 + c2 + c3 + c4
    X2 = torch.empty((16 * n, k2), dtype=torch.float32, device=dev)
    L.call("arco_gather_upcat_rows", L.ptr(lo1), ld1, c1, h1, w1_, L.ptr(r2), ld2, c2, h2, w2_, L.ptr(nb16), 16 * n,
           L.ptr(X2), k2)
    X2p = _fea_rows(X2, w2, 0)
    X3 = torch.empty((4 * n, k3), dtype=torch.float32, device=dev)
    L.call("arco_lerp4_cat_rows", L.ptr(X2p), k2, k2, L.ptr(lylx3), L.ptr(r3), ld3, c3, L.ptr(nb4), 4 * n, L.ptr(X3), k3)
    X3p = _fea_rows(X3, w3, 0)
    X4 = torch.empty((n, k4), dtype=torch.float32, device=dev)
    L.call("arco_lerp4_cat_rows", L.ptr(X3p), k3, k3, L.ptr(lylx4), L.ptr(r4), ld4, c4, L.ptr(pix), n, L.ptr(X4), k4)
    return X2, X3, X4, nb4, nb16, lylx3, lylx4, _gemm(X4, w4)


def _row_grad_buffer(ptr, shape, dev):
    """Zero channels-last buffer [nb, *spatial, c] for a sparse row-scatter gradient of the feature map [nb, c, *spatial] at
    `ptr`, and a finaliser fin(idx, n).  When the map is an output of a graph-replayed pass the buffer IS that pass's
    gradient-input buffer (ops.grad_sink: zero by invariant) - no dense fill, no copy into the graph - and fin registers
    the re-zeroing of the n touched rows idx after the backward graph has replayed; otherwise a fresh zeros tensor and
    a no-op."""
    nb, c, spatial = int(shape[0]), int(shape[1]), tuple(int(v) for v in shape[2:])
    gt, k = ops.grad_sink(ptr, (nb, c) + spatial)
    if gt is not None:
        buf = gt.static_grads[k].movedim(1, -1)
        if buf.is_contiguous() and c % 4 == 0:
            def fin(idx, n):
                gt.cleanup.append(lambda: L.call("arco_zero_rows", L.ptr(buf), c, c, L.ptr(idx), n))
            return buf, fin
        gt.sink_busy[k] = False
    return torch.zeros((nb,) + spatial + (c,), dtype=torch.float32, device=dev), (lambda idx, n: None)


def _row_grad_buffer_h(ptr, shape, dev):
    """_row_grad_buffer for a feature map stored as f16 (ops.fm_rows_half): (fp32 scatter target [nb, *spatial, c], done(idx, n) ->
    the f16 gradient [nb, *spatial, c] carrying ops.LOSS_SCALE).  Graph-replayed producer: the scatter target is a persistent fp32
    scratch (zero by invariant), done() casts the n touched rows into the producer's f16 gradient-input buffer (ops.grad_sink),
    re-zeroes them in the scratch and registers their re-zeroing in the f16 buffer for after the backward graph - the rows are summed
    in fp32 and rounded once, exactly as the dense cast of the dense fp32 gradient did.  Otherwise: a fresh fp32 tensor and a dense cast."""
    nb, c, spatial = int(shape[0]), int(shape[1]), tuple(int(v) for v in shape[2:])
    gt, k = ops.grad_sink(ptr, (nb, c) + spatial)
    if gt is not None:
        buf = gt.static_grads[k].movedim(1, -1)
        if buf.is_contiguous() and c % 4 == 0 and buf.dtype == torch.float16:
            scratch = gt.__dict__.setdefault("_row_scratch", {}).get(k)
            if scratch is None:
                scratch = gt._row_scratch[k] = torch.zeros(buf.shape, dtype=torch.float32, device=dev)

            def done(idx, n):
                L.call("arco_cast_rows_f2h", L.ptr(scratch), c, c, L.ptr(idx), n, float(ops.LOSS_SCALE), L.ptr(buf), c)
                L.call("arco_zero_rows", L.ptr(scratch), c, c, L.ptr(idx), n)
                gt.cleanup.append(lambda: L.call("arco_zero_rows_h", L.ptr(buf), c, c, L.ptr(idx), n))
                return buf
            return scratch, done
        gt.sink_busy[k] = False
    dense = torch.zeros((nb,) + spatial + (c,), dtype=torch.float32, device=dev)

    def done_dense(idx, n):
        out = torch.empty(dense.shape, dtype=torch.float16, device=dev)
        L.call("arco_cast_f2h", L.ptr(dense), dense.numel(), float(ops.LOSS_SCALE), L.ptr(out))
        return out
    return dense, done_dense


# Order-independent row scatter (csrc/det_scatter.hip): fixed-point int64 accumulation instead of fp32 atomics - the gradient of the
# row-sparse 3-D head is bit-reproducible, which f16 activation storage needs to be reproducible at all (the ulp of an fp32 atomic's
# arrival order decides f16 roundings downstream; profiles/r05_notes.md section 7).  ARCO_DET_SCATTER = 0: fp32 atomics everywhere;
# 1 (default): the 3-D head order-independent; 2: the 2-D heads too (17 more small launches on the 2-D step's critical path).
DET_SCATTER = int(os.environ.get("ARCO_DET_SCATTER", "1"))
# The fixed-point accumulators of the order-independent scatter: one int64 buffer of the destination's size per (device, rows, channels),
# kept between steps (zero between uses: every call clears exactly the rows it touched) - 315 MB for the LiTS-shaped full-resolution map,
# 80 MB for its half-resolution one (a trainer uses two).  Nothing evicts them behind the caller's back - captured HIP graphs hold their
# addresses; release_det_buffers() frees them once no graph that used them will be replayed.  A buffer first created inside a capture
# lives in that graph's pool, which is why the trainers run their first steps eagerly.  Resolution: values are scaled to the launch-wide largest magnitude, so a contribution below 2^-45 of that
# maximum is dropped - far below one fp32 ulp of any sum the maximum takes part in, not of a row that only holds tiny values.
_ACC64 = {}


def _det_acc(dev, rows, C):
    key = (dev.index, rows, C)
    acc = _ACC64.get(key)
    if acc is None:
        acc = _ACC64[key] = torch.zeros((rows, C), dtype=torch.int64, device=dev)
    return acc


def release_det_buffers():
    """Free the int64 accumulators of the order-independent scatter (they are re-created, zeroed, on the next use)."""
    _ACC64.clear()


def _det_scatter_rows(src, ld_src, C, div, idx, w, n_e, dst, ld_dst):
    """dst[idx[e]] = sum over e of w[e] * src[e // div] (C channels), dst rows zero before; src: [n_e // div, >= C] fp32 view."""
    dev = src.device
    rows = dst.numel() // ld_dst
    acc = _det_acc(dev, rows, C)
    mb = torch.empty(1, dtype=torch.int32, device=dev)
    try:
        L.call("arco_det_absmax", L.ptr(src), ld_src, C, n_e // div, L.ptr(mb))
        L.call("arco_det_scatter_rows", L.ptr(src), ld_src, C, div, None, L.ptr(idx), None if w is None else L.ptr(w), n_e,
               L.ptr(acc), C, L.ptr(mb))
        L.call("arco_det_finish_rows", None, L.ptr(idx), n_e, L.ptr(acc), C, C, L.ptr(mb), 1.0, L.ptr(dst), ld_dst)
        L.call("arco_det_clear_rows", None, L.ptr(idx), n_e, L.ptr(acc), C, C)
    except Exception:
        _ACC64.pop((dev.index, rows, C), None)      # "zero between uses" may no longer hold: the next call starts from a fresh buffer
        raise


def _scatter_upcat2d(dX, ldx, pix, n, dlo, clo, hi_h, hi_w, dhi, chi, ho, wo):
    """adjoint of arco_gather_upcat_rows: dlo += bilinear corners of dX[:, :clo], dhi[pix] += dX[:, clo:]"""
    if DET_SCATTER >= 2:
        dev = dX.device
        idx4 = torch.empty(4 * n, dtype=torch.int64, device=dev)
        w4 = torch.empty(4 * n, dtype=torch.float32, device=dev)
        L.call("arco_corner_rows2d", L.ptr(pix), n, hi_h, hi_w, ho, wo, L.ptr(idx4), L.ptr(w4))
        _det_scatter_rows(dX, ldx, clo, 4, idx4, w4, 4 * n, dlo, clo)
        _det_scatter_rows(dX[:, clo:], ldx, chi, 1, pix, None, n, dhi, chi)
    else:
        L.call("arco_scatter_upcat_rows", L.ptr(dX), ldx, L.ptr(pix), n, L.ptr(dlo), clo, clo, hi_h, hi_w, L.ptr(dhi), chi, chi, ho, wo)


def _lerp4_cat_rows_bwd(dX, ldx, clo, lylx, pix, n, dV, dhi, chi):
    """adjoint of arco_lerp4_cat_rows: the four weighted copies of dX[:, :clo] (plain stores) and dhi[pix] += dX[:, clo:]"""
    if DET_SCATTER >= 2:
        L.call("arco_lerp4_cat_rows_bwd", L.ptr(dX), ldx, clo, L.ptr(lylx), L.ptr(pix), n, L.ptr(dV), clo, L.ptr(dhi), chi, 0)
        _det_scatter_rows(dX[:, clo:], ldx, chi, 1, pix, None, n, dhi, chi)
    else:
        L.call("arco_lerp4_cat_rows_bwd", L.ptr(dX), ldx, clo, L.ptr(lylx), L.ptr(pix), n, L.ptr(dV), clo, L.ptr(dhi), chi, chi)


class LazyHead3Fn(torch.autograd.Function):
    """Three-level row-sparse head: fea2 (the 64 x 64 level), fea3 and fea4 are all evaluated only where the anchors need
    them - the 4 neighbours at 128 x 128 of every anchor and the 4 neighbours at 64 x 64 of each of those (16 n rows of
    448 channels instead of the 65 536 rows of the dense 64 x 64 map: ~1000 anchors per step).  Inputs: x1p =
    fea1(x)+x [B,384,32,32] (dense), f2 [B,64,64,64], f3 [B,32,128,128], f4 [B,16,256,256].  Same kernels as the two-level
    head, applied once more; d loss / d x1p becomes dense again from the 32 x 32 level down.  The dense fea2 GEMM
    (65 536 x 448 x 64 + the upsampled 117 MB residual), its data and weight gradients and the 64 x 64 bilinear backward
    leave the student path."""

    @staticmethod
    def forward(ctx, x1p, f2, f3, f4, w2, w3, w4, w1, wq2, pix):
        X2, X3, X4, nb4, nb16, lylx3, lylx4, h0 = _rows3lvl_forward(x1p, f2, f3, f4, w2, w3, w4, pix)
        hh = _gemm(h0, w1)
        a = _gemm(hh, wq2)
        ctx.save_for_backward(X2, X3, X4, h0, hh, w2, w3, w4, w1, wq2, pix, nb4, nb16, lylx3, lylx4)
        ctx.fptrs = (f2.data_ptr(), f3.data_ptr(), f4.data_ptr())
        ctx.geom = (int(x1p.shape[0]), int(x1p.shape[1]), int(x1p.shape[2]), int(x1p.shape[3]),
                    int(f2.shape[1]), int(f2.shape[2]), int(f2.shape[3]), int(f3.shape[1]), int(f3.shape[2]), int(f3.shape[3]),
                    int(f4.shape[1]), int(f4.shape[2]), int(f4.shape[3]))
        return a

    @staticmethod
    def backward(ctx, da):
        X2, X3, X4, h0, hh, w2, w3, w4, w1, wq2, pix, nb4, nb16, lylx3, lylx4 = ctx.saved_tensors
        nb, c1, h1, w1_, c2, h2, w2_, c3, h3, w3_, c4, h4, w4_ = ctx.geom
        dev = da.device
        n = int(pix.shape[0])
        k2, k3 = c1 + c2, c1 + c2 + c3
        da = da.contiguous()
        dwq2 = _wgrad(da, hh, wq2)
        dhh = _gemm_t(da, wq2)
        dw1 = _wgrad(dhh, h0, w1)
        dh0 = _gemm_t(dhh, w1)
        dw4 = _wgrad(dh0, X4, w4)
        dX4 = _gemm_t(dh0, w4)
        dX3p = torch.empty((4 * n, k3), dtype=torch.float32, device=dev)
        df4, fin4 = _row_grad_buffer(ctx.fptrs[2], (nb, c4, h4, w4_), dev)
        _lerp4_cat_rows_bwd(dX4, k3 + c4, k3, lylx4, pix, n, dX3p, df4, c4)
        fin4(pix, n)
        dw3 = _wgrad(dX3p, X3, w3)
        dX3 = _fea_rows(dX3p, w3, 1)
        dX2p = torch.empty((16 * n, k2), dtype=torch.float32, device=dev)
        df3, fin3 = _row_grad_buffer(ctx.fptrs[1], (nb, c3, h3, w3_), dev)
        _lerp4_cat_rows_bwd(dX3, k3, k2, lylx3, nb4, 4 * n, dX2p, df3, c3)
        fin3(nb4, 4 * n)
        dw2 = _wgrad(dX2p, X2, w2)
        dX2 = _fea_rows(dX2p, w2, 1)
        dx1p = torch.zeros((nb, h1, w1_, c1), dtype=torch.float32, device=dev)
        df2, fin2 = _row_grad_buffer(ctx.fptrs[0], (nb, c2, h2, w2_), dev)
        _scatter_upcat2d(dX2, k2, nb16, 16 * n, dx1p, c1, h1, w1_, df2, c2, h2, w2_)
        fin2(nb16, 16 * n)
        return (dx1p.permute(0, 3, 1, 2), df2.permute(0, 3, 1, 2), df3.permute(0, 3, 1, 2), df4.permute(0, 3, 1, 2),
                dw2, dw3, dw4, dw1, dwq2, None)


def lazy_head3(x1p, f2, f3, f4, fea2_weight, fea3_weight, fea4_weight, q1_weight, q2_weight, pix):
    return LazyHead3Fn.apply(x1p, f2, f3, f4, fea2_weight, fea3_weight, fea4_weight, q1_weight, q2_weight, pix)


def _class_weights(pl):
    """low-valid bits of every pixel as float rows [n_pix, Cp] (Cp = C padded to 4)."""
    Cp = _ceil(pl.C, 4)
    wm = torch.empty((pl.n_pix, Cp), dtype=torch.float32, device=pl.dev)
    L.call("arco_lv_weights", L.ptr(pl.codes), pl.n_pix, pl.C, Cp, L.ptr(wm))
    return wm, Cp


def _wsum(rows, ld, wt, ldw, n_rows, C, D, totals, out, ldo):
    ws = torch.empty(L.query("arco_proto_ws_floats", n_rows, C, D), dtype=torch.float32, device=rows.device)
    fn = "arco_weighted_row_sum_h" if rows.dtype == torch.float16 else "arco_weighted_row_sum"      # (f16 activation storage)
    L.call(fn, L.ptr(rows), ld, L.ptr(wt), ldw, n_rows, C, D, L.ptr(totals), L.ptr(ws), L.ptr(out), ldo)


class LazyTeacher2D:
    """Teacher side of the 2-D step without the dense 496-channel tensor: x3p = fea3(x)+x [B,480,128,128],
    f4 [B,16,256,256], w4 = fea4 weight (model_2D.py:51-53)."""

    def __init__(self, x3p, f4, w4):
        self.x3p, self.f4, self.w4 = x3p, f4, w4

    @torch.no_grad()
    def prototypes(self, pl):
        lo, ldlo = rows_view(self.x3p)
        hi, ldhi = rows_view(self.f4)
        nb, clo, hi_h, hi_w = (int(v) for v in self.x3p.shape)
        chi, ho, wo = int(self.f4.shape[1]), int(self.f4.shape[2]), int(self.f4.shape[3])
        C, K, D = pl.C, clo + chi, int(self.w4.shape[0])
        wm, Cp = _class_weights(pl)
        n_lo = nb * hi_h * hi_w
        wlo = torch.empty((n_lo, Cp), dtype=torch.float32, device=pl.dev)
        L.call("arco_bilinear_bwd", L.ptr(wm), Cp, nb, hi_h, hi_w, Cp, ho, wo, L.ptr(wlo), Cp, 0)
        S = torch.zeros((_ceil(C, 16), K), dtype=torch.float32, device=pl.dev)
        _wsum(lo, ldlo, wlo, Cp, n_lo, C, clo, pl.totals, S, K)
        _wsum(hi, ldhi, wm, Cp, pl.n_pix, C, chi, pl.totals, S[:, clo:], K)
        return _gemm(S, self.w4)[:C].contiguous()

    @torch.no_grad()
    def rows(self, pix):
        lo, ldlo = rows_view(self.x3p)
        hi, ldhi = rows_view(self.f4)
        nb, clo, hi_h, hi_w = (int(v) for v in self.x3p.shape)
        chi, ho, wo = int(self.f4.shape[1]), int(self.f4.shape[2]), int(self.f4.shape[3])
        n = int(pix.shape[0])
        X = torch.empty((n, clo + chi), dtype=torch.float32, device=pix.device)
        L.call("arco_gather_upcat_rows", L.ptr(lo), ldlo, clo, hi_h, hi_w, L.ptr(hi), ldhi, chi, ho, wo, L.ptr(pix), n,
               L.ptr(X), clo + chi)
        return _gemm(X, self.w4)


def _rows3d_forward(x2p, f3, f4, w3, w4, pix):
    """X3 = cat(trilinear(x2p)[pix], f3[pix]); X3p = fea3(X3)+X3; X4 = cat(X3p, f4[pix]); returns (X3, X4, fea4(X4))."""
    lo, ldlo = rows_view(x2p)
    r3, ld3 = rows_view(f3)
    r4, ld4 = rows_view(f4)
    nb, c2, d2, h2, w2_ = (int(v) for v in x2p.shape)
    c3, d3, h3, w3_ = (int(v) for v in f3.shape[1:])
    c4 = int(f4.shape[1])
    n = int(pix.shape[0])
    k3 = c2 + c3
    X3 = torch.empty((n, k3), dtype=torch.float32, device=pix.device)
    L.call("arco_gather_upcat_rows3d_h" if r3.dtype == torch.float16 else "arco_gather_upcat_rows3d",
           L.ptr(lo), ldlo, c2, d2, h2, w2_, L.ptr(r3), ld3, c3, d3, h3, w3_, L.ptr(pix), n, L.ptr(X3), k3)
    y3, _ = ops.conv_raw(X3, k3, k3, ops.pack_weight(w3, 1, 0), k3, 1, 1, n, 1, residual=X3, ld_res=k3)
    X4 = torch.empty((n, k3 + c4), dtype=torch.float32, device=pix.device)
    X4[:, :k3] = y3.permute(0, 2, 3, 1).reshape(n, k3)
    L.call("arco_gather_rows_h" if r4.dtype == torch.float16 else "arco_gather_rows",
           L.ptr(r4), ld4, c4, None, L.ptr(pix), None, 0, n, L.ptr(X4[:, k3:]), k3 + c4)
    return X3, X4, _gemm(X4, w4)


class LazyHead3dFn(torch.autograd.Function):
    """Row-sparse student head of the 3-D step: q_representation(FeatureExtractor_3d(...)) rows at the sampled
    voxels only.  Everything above the 56x56x40 level is per-voxel (fea3, the identity resize, fea4, q_rep),
    so one trilinear gather from x2p = fea2(x)+x suffices (model_3D.py:46-58, train_arco_3d.py:289-296)."""

    @staticmethod
    def forward(ctx, x2p, f3, f4, w3, w4, w1, w2, pix):
        X3, X4, h0 = _rows3d_forward(x2p, f3, f4, w3, w4, pix)
        h1 = _gemm(h0, w1)
        a = _gemm(h1, w2)
        ctx.save_for_backward(X3, X4, h0, h1, w3, w4, w1, w2, pix)
        ctx.shapes = (tuple(x2p.shape), tuple(f3.shape), tuple(f4.shape))
        ctx.fptrs = (f3.data_ptr(), f4.data_ptr())
        ctx.fhalf = (f3.dtype == torch.float16, f4.dtype == torch.float16)      # ops.fm_rows_half: maps consumed as stored
        return a

    @staticmethod
    def backward(ctx, da):
        X3, X4, h0, h1, w3, w4, w1, w2, pix = ctx.saved_tensors
        s2, s3, s4 = ctx.shapes
        dev = da.device
        n = int(pix.shape[0])
        k3, c4 = int(X3.shape[1]), int(s4[1])
        da = da.contiguous()
        dw2 = _wgrad(da, h1, w2)
        dh1 = _gemm_t(da, w2)
        dw1 = _wgrad(dh1, h0, w1)
        dh0 = _gemm_t(dh1, w1)
        dw4 = _wgrad(dh0, X4, w4)
        dX4 = _gemm_t(dh0, w4)
        dX3p = dX4[:, :k3].contiguous()
        df4, fin4 = (_row_grad_buffer_h if ctx.fhalf[1] else _row_grad_buffer)(ctx.fptrs[1], s4, dev)
        if DET_SCATTER:
            _det_scatter_rows(dX4[:, k3:], k3 + c4, c4, 1, pix, None, n, df4, c4)
        else:
            L.call("arco_scatter_add_rows", L.ptr(dX4[:, k3:]), k3 + c4, c4, None, L.ptr(pix), n, None, 1.0, L.ptr(df4), c4)
        r = fin4(pix, n)
        if ctx.fhalf[1]:
            df4 = r
        dw3 = _wgrad(dX3p, X3, w3)
        y, _ = ops.conv_raw(dX3p, k3, k3, ops.pack_weight(w3, 1, 1), k3, 1, 1, n, 1, residual=dX3p, ld_res=k3)
        dX3 = y.permute(0, 2, 3, 1).reshape(n, k3)
        c2, c3 = int(s2[1]), int(s3[1])
        dx2p = torch.zeros((s2[0], *s2[2:], c2), dtype=torch.float32, device=dev)
        df3, fin3 = (_row_grad_buffer_h if ctx.fhalf[0] else _row_grad_buffer)(ctx.fptrs[0], s3, dev)
        if DET_SCATTER:
            idx8 = torch.empty(8 * n, dtype=torch.int64, device=dev)
            w8 = torch.empty(8 * n, dtype=torch.float32, device=dev)
            L.call("arco_corner_rows3d", L.ptr(pix), n, s2[2], s2[3], s2[4], s3[2], s3[3], s3[4], L.ptr(idx8), L.ptr(w8))
            _det_scatter_rows(dX3, k3, c2, 8, idx8, w8, 8 * n, dx2p, c2)
            _det_scatter_rows(dX3[:, c2:], k3, c3, 1, pix, None, n, df3, c3)
        else:
            L.call("arco_scatter_upcat_rows3d", L.ptr(dX3), k3, L.ptr(pix), n, L.ptr(dx2p), c2, c2, s2[2], s2[3], s2[4],
                   L.ptr(df3), c3, c3, s3[2], s3[3], s3[4])
        r = fin3(pix, n)
        if ctx.fhalf[0]:
            df3 = r
        return (dx2p.movedim(-1, 1), df3.movedim(-1, 1), df4.movedim(-1, 1), dw3, dw4, dw1, dw2, None)


def lazy_head3d(x2p, f3, f4, fea3_weight, fea4_weight, q1_weight, q2_weight, pix):
    return LazyHead3dFn.apply(x2p, f3, f4, fea3_weight, fea4_weight, q1_weight, q2_weight, pix)


class LazyTeacher3D:
    """Teacher side of the 3-D step: prototype_c = W4 . cat((W3+I) . mean_c(cat(up(x2p), f3)), mean_c(f4)) with the
    class mask pushed through the trilinear adjoint; key rows evaluated at the key voxels only."""

    def __init__(self, x2p, f3, f4, w3, w4):
        self.x2p, self.f3, self.f4, self.w3, self.w4 = x2p, f3, f4, w3, w4

    @torch.no_grad()
    def prototypes(self, pl):
        lo, ldlo = rows_view(self.x2p)
        r3, ld3 = rows_view(self.f3)
        r4, ld4 = rows_view(self.f4)
        nb, c2, d2, h2, w2_ = (int(v) for v in self.x2p.shape)
        c3, d3, h3, w3_ = (int(v) for v in self.f3.shape[1:])
        c4 = int(self.f4.shape[1])
        C, k3 = pl.C, c2 + c3
        wm, Cp = _class_weights(pl)
        n_lo = nb * d2 * h2 * w2_
        wlo = torch.empty((n_lo, Cp), dtype=torch.float32, device=pl.dev)
        L.call("arco_trilinear_bwd", L.ptr(wm), Cp, nb, d2, h2, w2_, Cp, d3, h3, w3_, L.ptr(wlo), Cp)
        R = _ceil(C, 16)
        S3 = torch.zeros((R, k3), dtype=torch.float32, device=pl.dev)
        _wsum(lo, ldlo, wlo, Cp, n_lo, C, c2, pl.totals, S3, k3)
        _wsum(r3, ld3, wm, Cp, pl.n_pix, C, c3, pl.totals, S3[:, c2:], k3)
        y3, _ = ops.conv_raw(S3, k3, k3, ops.pack_weight(self.w3, 1, 0), k3, 1, 1, R, 1, residual=S3, ld_res=k3)
        S4 = torch.zeros((R, k3 + c4), dtype=torch.float32, device=pl.dev)
        S4[:, :k3] = y3.permute(0, 2, 3, 1).reshape(R, k3)
        _wsum(r4, ld4, wm, Cp, pl.n_pix, C, c4, pl.totals, S4[:, k3:], k3 + c4)
        return _gemm(S4, self.w4)[:C].contiguous()

    @torch.no_grad()
    def rows(self, pix):
        return _rows3d_forward(self.x2p, self.f3, self.f4, self.w3, self.w4, pix)[2]


# ---------------------------------------------------------------------------------------------------------------------------
# The 3-D head with the 56x56x40 level lazy too (round 6).  Above that level everything is per-voxel, so a sampled voxel needs
# exactly its EIGHT trilinear corners of x2p = fea2(cat(up(x1p), f2)) + cat(...): 8 n rows of 224 channels instead of the dense
# 501 760-row map (450 MB at the LA size: the step's longest GEMM launch writes it, its prototype sums, its resize adjoint, its
# data and weight gradients read it - for ~1 % of its rows).  x1p = fea1(...)+... [B,192,28,28,20] stays dense.
# ---------------------------------------------------------------------------------------------------------------------------
def _rows3d_l3_forward(x1p, f2, f3, f4, w2, w3, w4, pix):
    """corner rows idx8 (at f2's resolution) of every sampled voxel; X2 = cat(trilinear(x1p), f2) on them; X2p = fea2(X2)+X2;
    X3 = cat(trilinear blend of X2p's eight rows, f3[pix]); then as _rows3d_forward.  Returns (X2, X3, X4, idx8, w8, fea4(X4))."""
    dev = pix.device
    lo, ldlo = rows_view(x1p)
    r2, ld2 = rows_view(f2)
    r3, ld3 = rows_view(f3)
    r4, ld4 = rows_view(f4)
    nb, c1, d1, h1, w1_ = (int(v) for v in x1p.shape)
    c2, d2, h2, w2_ = (int(v) for v in f2.shape[1:])
    c3, d3, h3, w3_ = (int(v) for v in f3.shape[1:])
    c4 = int(f4.shape[1])
    n = int(pix.shape[0])
    k2 = c1 + c2
    k3 = k2 + c3
    idx8 = torch.empty(8 * n, dtype=torch.int64, device=dev)
    w8 = torch.empty(8 * n, dtype=torch.float32, device=dev)
    L.call("arco_corner_rows3d", L.ptr(pix), n, d2, h2, w2_, d3, h3, w3_, L.ptr(idx8), L.ptr(w8))
    X2 = torch.empty((8 * n, k2), dtype=torch.float32, device=dev)
    L.call("arco_gather_upcat_rows3d_h" if r2.dtype == torch.float16 else "arco_gather_upcat_rows3d",
           L.ptr(lo), ldlo, c1, d1, h1, w1_, L.ptr(r2), ld2, c2, d2, h2, w2_, L.ptr(idx8), 8 * n, L.ptr(X2), k2)
    X2p = _fea_rows(X2, w2, 0)
    X3 = torch.empty((n, k3), dtype=torch.float32, device=dev)
    L.call("arco_lerp8_cat_rows3d_h" if r3.dtype == torch.float16 else "arco_lerp8_cat_rows3d",
           L.ptr(X2p), k2, k2, d2, h2, w2_, L.ptr(r3), ld3, c3, d3, h3, w3_, L.ptr(pix), n, L.ptr(X3), k3)
    y3, _ = ops.conv_raw(X3, k3, k3, ops.pack_weight(w3, 1, 0), k3, 1, 1, n, 1, residual=X3, ld_res=k3)
    X4 = torch.empty((n, k3 + c4), dtype=torch.float32, device=dev)
    X4[:, :k3] = y3.permute(0, 2, 3, 1).reshape(n, k3)
    L.call("arco_gather_rows_h" if r4.dtype == torch.float16 else "arco_gather_rows",
           L.ptr(r4), ld4, c4, None, L.ptr(pix), None, 0, n, L.ptr(X4[:, k3:]), k3 + c4)
    return X2, X3, X4, idx8, w8, _gemm(X4, w4)


class LazyHead3dL3Fn(torch.autograd.Function):
    """Row-sparse student head of the 3-D step with fea2 evaluated on rows too (model_3D.py:46-58, train_arco_3d.py:289-296): inputs
    x1p = fea1(.)+. [B,192,28,28,20] (dense), f2 [B,32,56,56,40], f3, f4 [B,16,112,112,80].  The gradient becomes dense again from the
    28x28x20 level down (48 MB); f2's, f3's and f4's are row scatters (order-independent by default, DET_SCATTER)."""

    @staticmethod
    def forward(ctx, x1p, f2, f3, f4, w2, w3, w4, w1, wq2, pix):
        X2, X3, X4, idx8, w8, h0 = _rows3d_l3_forward(x1p, f2, f3, f4, w2, w3, w4, pix)
        h1 = _gemm(h0, w1)
        a = _gemm(h1, wq2)
        ctx.save_for_backward(X2, X3, X4, h0, h1, w2, w3, w4, w1, wq2, pix, idx8, w8)
        ctx.shapes = (tuple(x1p.shape), tuple(f2.shape), tuple(f3.shape), tuple(f4.shape))
        ctx.fptrs = (f2.data_ptr(), f3.data_ptr(), f4.data_ptr())
        ctx.fhalf = (f2.dtype == torch.float16, f3.dtype == torch.float16, f4.dtype == torch.float16)
        return a

    @staticmethod
    def backward(ctx, da):
        X2, X3, X4, h0, h1, w2, w3, w4, w1, wq2, pix, idx8, w8 = ctx.saved_tensors
        s1, s2, s3, s4 = ctx.shapes
        dev = da.device
        n = int(pix.shape[0])
        k2, k3 = int(X2.shape[1]), int(X3.shape[1])
        c1, c2, c3, c4 = int(s1[1]), int(s2[1]), int(s3[1]), int(s4[1])
        da = da.contiguous()
        dwq2 = _wgrad(da, h1, wq2)
        dh1 = _gemm_t(da, wq2)
        dw1 = _wgrad(dh1, h0, w1)
        dh0 = _gemm_t(dh1, w1)
        dw4 = _wgrad(dh0, X4, w4)
        dX4 = _gemm_t(dh0, w4)
        dX3p = dX4[:, :k3].contiguous()

        def scatter(src, ld_src, C, idx, n_e, ptr, shape, half):
            buf, fin = (_row_grad_buffer_h if half else _row_grad_buffer)(ptr, shape, dev)
            if DET_SCATTER:
                _det_scatter_rows(src, ld_src, C, 1, idx, None, n_e, buf, C)
            else:
                L.call("arco_scatter_add_rows", L.ptr(src), ld_src, C, None, L.ptr(idx), n_e, None, 1.0, L.ptr(buf), C)
            r = fin(idx, n_e)
            return r if half else buf
        df4 = scatter(dX4[:, k3:], k3 + c4, c4, pix, n, ctx.fptrs[2], s4, ctx.fhalf[2])
        dw3 = _wgrad(dX3p, X3, w3)
        y, _ = ops.conv_raw(dX3p, k3, k3, ops.pack_weight(w3, 1, 1), k3, 1, 1, n, 1, residual=dX3p, ld_res=k3)
        dX3 = y.permute(0, 2, 3, 1).reshape(n, k3)
        df3 = scatter(dX3[:, k2:], k3, c3, pix, n, ctx.fptrs[1], s3, ctx.fhalf[1])
        dX2p = torch.empty((8 * n, k2), dtype=torch.float32, device=dev)
        L.call("arco_lerp8_rows3d_bwd", L.ptr(dX3), k3, k2, L.ptr(w8), n, L.ptr(dX2p), k2)
        dw2 = _wgrad(dX2p, X2, w2)
        dX2 = _fea_rows(dX2p, w2, 1)
        df2 = scatter(dX2[:, c1:], k2, c2, idx8, 8 * n, ctx.fptrs[0], s2, ctx.fhalf[0])
        dx1p = torch.zeros((s1[0], *s1[2:], c1), dtype=torch.float32, device=dev)
        if DET_SCATTER:
            idx64 = torch.empty(64 * n, dtype=torch.int64, device=dev)
            w64 = torch.empty(64 * n, dtype=torch.float32, device=dev)
            L.call("arco_corner_rows3d", L.ptr(idx8), 8 * n, s1[2], s1[3], s1[4], s2[2], s2[3], s2[4], L.ptr(idx64), L.ptr(w64))
            _det_scatter_rows(dX2, k2, c1, 8, idx64, w64, 64 * n, dx1p, c1)
        else:       # (the fp32-atomic adjoint of the gather: lo part only - dhi = a scratch row sink of the right shape)
            sink = torch.zeros((s2[0], *s2[2:], c2), dtype=torch.float32, device=dev)
            L.call("arco_scatter_upcat_rows3d", L.ptr(dX2), k2, L.ptr(idx8), 8 * n, L.ptr(dx1p), c1, c1, s1[2], s1[3], s1[4],
                   L.ptr(sink), c2, c2, s2[2], s2[3], s2[4])
        return (dx1p.movedim(-1, 1), df2.movedim(-1, 1), df3.movedim(-1, 1), df4.movedim(-1, 1), dw2, dw3, dw4, dw1, dwq2, None)


def lazy_head3d_l3(x1p, f2, f3, f4, fea2_weight, fea3_weight, fea4_weight, q1_weight, q2_weight, pix):
    return LazyHead3dL3Fn.apply(x1p, f2, f3, f4, fea2_weight, fea3_weight, fea4_weight, q1_weight, q2_weight, pix)


class LazyTeacher3DL3:
    """Teacher side with the 56x56x40 level lazy too: prototype_c = W4 . cat((W3+I) . cat((W2+I) . cat(S(x1p; w''), S(f2; w')), S(f3; w)),
    S(f4; w)) with S(t; w) = the class-weighted row sums of t and the class mask w pushed through the trilinear adjoint once (w', the
    56x56x40 level) and twice (w'', the 28x28x20 level); key rows evaluated at the key voxels only."""

    def __init__(self, x1p, f2, f3, f4, w2, w3, w4):
        self.x1p, self.f2, self.f3, self.f4, self.w2, self.w3, self.w4 = x1p, f2, f3, f4, w2, w3, w4

    @torch.no_grad()
    def prototypes(self, pl):
        r1, ld1 = rows_view(self.x1p)
        r2, ld2 = rows_view(self.f2)
        r3, ld3 = rows_view(self.f3)
        r4, ld4 = rows_view(self.f4)
        nb, c1, d1, h1, w1_ = (int(v) for v in self.x1p.shape)
        c2, d2, h2, w2_ = (int(v) for v in self.f2.shape[1:])
        c3, d3, h3, w3_ = (int(v) for v in self.f3.shape[1:])
        c4 = int(self.f4.shape[1])
        C, k2 = pl.C, c1 + c2
        k3 = k2 + c3
        wm, Cp = _class_weights(pl)
        n2, n1 = nb * d2 * h2 * w2_, nb * d1 * h1 * w1_
        wl2 = torch.empty((n2, Cp), dtype=torch.float32, device=pl.dev)
        L.call("arco_trilinear_bwd", L.ptr(wm), Cp, nb, d2, h2, w2_, Cp, d3, h3, w3_, L.ptr(wl2), Cp)
        wl1 = torch.empty((n1, Cp), dtype=torch.float32, device=pl.dev)
        L.call("arco_trilinear_bwd", L.ptr(wl2), Cp, nb, d1, h1, w1_, Cp, d2, h2, w2_, L.ptr(wl1), Cp)
        R = _ceil(C, 16)
        S2 = torch.zeros((R, k2), dtype=torch.float32, device=pl.dev)
        _wsum(r1, ld1, wl1, Cp, n1, C, c1, pl.totals, S2, k2)
        _wsum(r2, ld2, wl2, Cp, n2, C, c2, pl.totals, S2[:, c1:], k2)
        S3 = torch.zeros((R, k3), dtype=torch.float32, device=pl.dev)
        S3[:, :k2] = _fea_rows(S2, self.w2, 0)
        _wsum(r3, ld3, wm, Cp, pl.n_pix, C, c3, pl.totals, S3[:, k2:], k3)
        y3, _ = ops.conv_raw(S3, k3, k3, ops.pack_weight(self.w3, 1, 0), k3, 1, 1, R, 1, residual=S3, ld_res=k3)
        S4 = torch.zeros((R, k3 + c4), dtype=torch.float32, device=pl.dev)
        S4[:, :k3] = y3.permute(0, 2, 3, 1).reshape(R, k3)
        _wsum(r4, ld4, wm, Cp, pl.n_pix, C, c4, pl.totals, S4[:, k3:], k3 + c4)
        return _gemm(S4, self.w4)[:C].contiguous()

    @torch.no_grad()
    def rows(self, pix):
        return _rows3d_l3_forward(self.x1p, self.f2, self.f3, self.f4, self.w2, self.w3, self.w4, pix)[5]


class LazyTeacher2DL2:
    """Two-level lazy teacher (default of the 2-D step): also fea3 is never evaluated densely.
    prototype_c = W4 . cat((W3+I) . mean_c(cat(up(x2p), f3)), mean_c(f4)), the class mask being pushed through
    TWO bilinear adjoints (256^2 -> 128^2 -> 64^2); key rows through the two-level row path."""

    def __init__(self, x2p, f3, f4, w3, w4):
        self.x2p, self.f3, self.f4, self.w3, self.w4 = x2p, f3, f4, w3, w4

    @torch.no_grad()
    def prototypes(self, pl):
        lo, ldlo = rows_view(self.x2p)
        r3, ld3 = rows_view(self.f3)
        r4, ld4 = rows_view(self.f4)
        nb, c2, h2, w2_ = (int(v) for v in self.x2p.shape)
        c3, h3, w3_ = (int(v) for v in self.f3.shape[1:])
        c4, h4, w4_ = (int(v) for v in self.f4.shape[1:])
        C, k3 = pl.C, c2 + c3
        wm, Cp = _class_weights(pl)
        w3l = torch.empty((nb * h3 * w3_, Cp), dtype=torch.float32, device=pl.dev)
        L.call("arco_bilinear_bwd", L.ptr(wm), Cp, nb, h3, w3_, Cp, h4, w4_, L.ptr(w3l), Cp, 0)
        w2l = torch.empty((nb * h2 * w2_, Cp), dtype=torch.float32, device=pl.dev)
        L.call("arco_bilinear_bwd", L.ptr(w3l), Cp, nb, h2, w2_, Cp, h3, w3_, L.ptr(w2l), Cp, 0)
        R = _ceil(C, 16)
        S3 = torch.zeros((R, k3), dtype=torch.float32, device=pl.dev)
        _wsum(lo, ldlo, w2l, Cp, nb * h2 * w2_, C, c2, pl.totals, S3, k3)
        _wsum(r3, ld3, w3l, Cp, nb * h3 * w3_, C, c3, pl.totals, S3[:, c2:], k3)
        y3, _ = ops.conv_raw(S3, k3, k3, ops.pack_weight(self.w3, 1, 0), k3, 1, 1, R, 1, residual=S3, ld_res=k3)
        S4 = torch.zeros((R, k3 + c4), dtype=torch.float32, device=pl.dev)
        S4[:, :k3] = y3.permute(0, 2, 3, 1).reshape(R, k3)
        _wsum(r4, ld4, wm, Cp, pl.n_pix, C, c4, pl.totals, S4[:, k3:], k3 + c4)
        return _gemm(S4, self.w4)[:C].contiguous()

    @torch.no_grad()
    def rows(self, pix):
        return _rows2d_forward(self.x2p, self.f3, self.f4, self.w3, self.w4, pix)[4]


class LazyTeacher2DL3:
    """Three-level lazy teacher (with --head_levels 3): fea2 is never evaluated densely either.  prototype_c =
    W4 . cat((W3+I) . cat((W2+I) . mean_c(cat(up(x1p), f2)), mean_c(f3)), mean_c(f4)): every map is linear in the feature
    maps, so the class mask is pushed through THREE bilinear adjoints (256^2 -> 128^2 -> 64^2 -> 32^2) and the weighted row
    sums run on x1p, f2, f3, f4 (66 MB of reads instead of the dense 117 MB fea2 output, which is not built at all); key
    rows through the three-level row path."""

    def __init__(self, x1p, f2, f3, f4, w2, w3, w4):
        self.x1p, self.f2, self.f3, self.f4, self.w2, self.w3, self.w4 = x1p, f2, f3, f4, w2, w3, w4

    @torch.no_grad()
    def prototypes(self, pl):
        lo, ldlo = rows_view(self.x1p)
        r2, ld2 = rows_view(self.f2)
        r3, ld3 = rows_view(self.f3)
        r4, ld4 = rows_view(self.f4)
        nb, c1, h1, w1_ = (int(v) for v in self.x1p.shape)
        c2, h2, w2_ = (int(v) for v in self.f2.shape[1:])
        c3, h3, w3_ = (int(v) for v in self.f3.shape[1:])
        c4, h4, w4_ = (int(v) for v in self.f4.shape[1:])
        C, k2 = pl.C, c1 + c2
        k3 = k2 + c3
        wm, Cp = _class_weights(pl)
        w3l = torch.empty((nb * h3 * w3_, Cp), dtype=torch.float32, device=pl.dev)
        L.call("arco_bilinear_bwd", L.ptr(wm), Cp, nb, h3, w3_, Cp, h4, w4_, L.ptr(w3l), Cp, 0)
        w2l = torch.empty((nb * h2 * w2_, Cp), dtype=torch.float32, device=pl.dev)
        L.call("arco_bilinear_bwd", L.ptr(w3l), Cp, nb, h2, w2_, Cp, h3, w3_, L.ptr(w2l), Cp, 0)
        w1l = torch.empty((nb * h1 * w1_, Cp), dtype=torch.float32, device=pl.dev)
        L.call("arco_bilinear_bwd", L.ptr(w2l), Cp, nb, h1, w1_, Cp, h2, w2_, L.ptr(w1l), Cp, 0)
        R = _ceil(C, 16)
        S2 = torch.zeros((R, k2), dtype=torch.float32, device=pl.dev)
        _wsum(lo, ldlo, w1l, Cp, nb * h1 * w1_, C, c1, pl.totals, S2, k2)
        _wsum(r2, ld2, w2l, Cp, nb * h2 * w2_, C, c2, pl.totals, S2[:, c1:], k2)
        S3 = torch.zeros((R, k3), dtype=torch.float32, device=pl.dev)
        S3[:, :k2] = _fea_rows(S2, self.w2, 0)
        _wsum(r3, ld3, w3l, Cp, nb * h3 * w3_, C, c3, pl.totals, S3[:, k2:], k3)
        S4 = torch.zeros((R, k3 + c4), dtype=torch.float32, device=pl.dev)
        S4[:, :k3] = _fea_rows(S3, self.w3, 0)
        _wsum(r4, ld4, wm, Cp, pl.n_pix, C, c4, pl.totals, S4[:, k3:], k3 + c4)
        return _gemm(S4, self.w4)[:C].contiguous()

    @torch.no_grad()
    def rows(self, pix):
        return _rows3lvl_forward(self.x1p, self.f2, self.f3, self.f4, self.w2, self.w3, self.w4, pix)[7]
