"""Which operation of LazyHead3Fn.backward changes dX3p / dX2p after the adjoint kernel wrote them?  The backward body is restated
here with a snapshot right after each producing kernel and a comparison after every later operation (device-synchronised).
python tools/debug/head_canary.py [trials] [mode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import test_configs_at_size_gpu as TC
from arco_amd import train_arco_2d as T, ops, head as H_, _lib as L
n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
T.TEACHER_SIDE = int(sys.argv[2]) if len(sys.argv) > 2 else 3
st = TC._make_acdc([]); TC._drop_off(st)
_row_grad_buffer, _wgrad, _gemm_t, _fea_rows = H_._row_grad_buffer, H_._wgrad, H_._gemm_t, H_._fea_rows
events = []
first = {}
def dbg_backward(ctx, da):
    X2, X3, X4, h0, hh, w2, w3, w4, w1, wq2, pix, nb4, nb16, lylx3, lylx4 = ctx.saved_tensors
    nb, c1, h1, w1_, c2, h2, w2_, c3, h3, w3_, c4, h4, w4_ = ctx.geom
    dev = da.device
    n = int(pix.shape[0])
    k2, k3 = c1 + c2, c1 + c2 + c3
    watch = {}
    def snap(name, t):
        torch.cuda.synchronize(); watch[name] = (t, t.clone()); torch.cuda.synchronize()
    def chk(label):
        torch.cuda.synchronize()
        for name, (t, s) in watch.items():
            if not torch.equal(t, s):
                bad = (t != s)
                events.append(f"{name} changed after {label}: {int(bad.sum())} elements, zeros {int((t[bad] == 0).sum())}, ptr {t.data_ptr():#x} bytes {t.numel() * 4}")
                watch[name] = (t, t.clone())
    da = da.contiguous()
    dwq2 = _wgrad(da, hh, wq2)
    dhh = _gemm_t(da, wq2)
    dw1 = _wgrad(dhh, h0, w1)
    dh0 = _gemm_t(dhh, w1)
    dw4 = _wgrad(dh0, X4, w4)
    dX4 = _gemm_t(dh0, w4)
    dX3p = torch.empty((4 * n, k3), dtype=torch.float32, device=dev)
    df4, fin4 = _row_grad_buffer(ctx.fptrs[2], (nb, c4, h4, w4_), dev)
    L.call("arco_lerp4_cat_rows_bwd", L.ptr(dX4), k3 + c4, k3, L.ptr(lylx4), L.ptr(pix), n, L.ptr(dX3p), k3, L.ptr(df4), c4, c4)
    snap("dX3p", dX3p)
    # expected adjoint from the same inputs, with torch ops
    wts = torch.stack(((1 - lylx4[0::2]) * (1 - lylx4[1::2]), (1 - lylx4[0::2]) * lylx4[1::2], lylx4[0::2] * (1 - lylx4[1::2]), lylx4[0::2] * lylx4[1::2]), 1)
    exp = (dX4[:, None, :k3] * wts[:, :, None]).reshape(4 * n, k3)
    if not torch.allclose(exp, dX3p, rtol=1e-6, atol=0):
        bad = ~torch.isclose(exp, dX3p, rtol=1e-6, atol=0)
        events.append(f"lerp4 bwd output != expected: {int(bad.sum())} elements")
    for nm, tt in (("dX4", dX4), ("lylx4", lylx4), ("dX3p_out", dX3p), ("dh0", dh0), ("da", da), ("X4", X4), ("h0", h0), ("hh", hh)):
        if nm not in first:
            first[nm] = tt.clone()
        elif not torch.equal(first[nm], tt):
            bad = first[nm] != tt
            rows = bad.reshape(bad.shape[0], -1).any(1).nonzero().flatten() if bad.dim() > 1 else bad.nonzero().flatten()
            cols = bad.any(0).nonzero().flatten() if bad.dim() > 1 else rows
            events.append(f"{nm} differs from trial 0: {int(bad.sum())} el, rows {int(rows.numel())} [{int(rows.min())}..{int(rows.max())}] cols {int(cols.numel())} [{int(cols.min())}..{int(cols.max())}] zeros now {int((tt[bad] == 0).sum())} zeros then {int((first[nm][bad] == 0).sum())}")
    fin4(pix, n); chk("fin4")
    dw3 = _wgrad(dX3p, X3, w3); chk("wgrad w3")
    wp = ops.pack_weight(w3, 1, 1); chk("pack_weight w3 mode 1")
    dX3 = _fea_rows(dX3p, w3, 1); chk("fea_rows w3")
    dX2p = torch.empty((16 * n, k2), dtype=torch.float32, device=dev); chk("alloc dX2p")
    df3, fin3 = _row_grad_buffer(ctx.fptrs[1], (nb, c3, h3, w3_), dev)
    L.call("arco_lerp4_cat_rows_bwd", L.ptr(dX3), k3, k2, L.ptr(lylx3), L.ptr(nb4), 4 * n, L.ptr(dX2p), k2, L.ptr(df3), c3, c3)
    snap("dX2p", dX2p); chk("lerp4 bwd 2")
    fin3(nb4, 4 * n); chk("fin3")
    dw2 = _wgrad(dX2p, X2, w2); chk("wgrad w2")
    wp2 = ops.pack_weight(w2, 1, 1); chk("pack_weight w2 mode 1")
    dX2 = _fea_rows(dX2p, w2, 1); chk("fea_rows w2")
    dx1p = torch.zeros((nb, h1, w1_, c1), dtype=torch.float32, device=dev); chk("zeros dx1p")
    df2, fin2 = _row_grad_buffer(ctx.fptrs[0], (nb, c2, h2, w2_), dev)
    L.call("arco_scatter_upcat_rows", L.ptr(dX2), k2, L.ptr(nb16), 16 * n, L.ptr(dx1p), c1, c1, h1, w1_, L.ptr(df2), c2, c2, h2, w2_)
    chk("scatter_upcat_rows")
    fin2(nb16, 16 * n)
    return (dx1p.permute(0, 3, 1, 2), df2.permute(0, 3, 1, 2), df3.permute(0, 3, 1, 2), df4.permute(0, 3, 1, 2),
            dw2, dw3, dw4, dw1, dwq2, None)
H_.LazyHead3Fn.backward = staticmethod(dbg_backward)
for it in range(4):
    TC.seed_all(800 + it); st.step(*TC._acdc_batch(20 + it))
torch.cuda.synchronize()
batch = TC._acdc_batch(24)
for t in range(n_trials):
    events.clear()
    TC.seed_all(804)
    st.step(*batch)
    torch.cuda.synchronize()
    print(f"trial {t}: " + (" | ".join(events) if events else "clean"), flush=True)
