"""Which trainer is not reproducible?  From one snapshot of the state, the same step (same batch, same seeds) is executed again and
again by the eager trainer and by the graph-replay trainer; each run's flat gradient is compared with that trainer's FIRST run.
Head-backward atomics give ~1e-7; anything larger is a hazard.  python tools/debug/self_consistency.py [trials] [mode]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import test_configs_at_size_gpu as TC
from arco_amd import train_arco_2d as T, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
if len(sys.argv) > 2:
    T.TEACHER_SIDE = int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "eg"
sts = {}
if "e" in which:
    sts["eager"] = TC._make_acdc(["--graphs", "0", "--graph_train", "0"])
if "g" in which:
    sts["graph"] = TC._make_acdc([])
for st in sts.values():
    TC._drop_off(st)
def snapshot(st):
    return dict(p=st.optimizer.flat_p.clone(), b=st.optimizer.flat_buf.clone(), started=list(st.optimizer._started),
                lr=[g['lr'] for g in st.optimizer.param_groups],
                sd=[{k: v.clone() for k, v in m.state_dict().items()} for m in (st.model, st.ema_model, st.k_feature_extractor)],
                bank=[[t.clone() for t in m] for m in st.memobank], ptr=[q.clone() if torch.is_tensor(q) else q for q in st.queue_ptrlis],
                it=st.iter_num)
def restore(st, s):
    with torch.no_grad():
        st.optimizer.flat_p.copy_(s["p"]); st.optimizer.flat_buf.copy_(s["b"]); st.optimizer._started = list(s["started"])
        for g, lr in zip(st.optimizer.param_groups, s["lr"]):
            g['lr'] = lr
        for m, sd in zip((st.model, st.ema_model, st.k_feature_extractor), s["sd"]):
            for k, v in m.state_dict().items():
                v.copy_(sd[k])
        st.memobank = [[t.clone() for t in m] for m in s["bank"]]
        st.queue_ptrlis = [q.clone() if torch.is_tensor(q) else q for q in s["ptr"]]
    st.iter_num = s["it"]
    ops.bump_weight_epoch()
MAIN = torch.cuda.Stream() if os.environ.get("SC_MAIN_STREAM") else None      # the whole step on an explicit stream instead of the legacy default stream
import contextlib
def main_ctx():
    if MAIN is None:
        return contextlib.nullcontext()
    MAIN.wait_stream(torch.cuda.default_stream())
    return torch.cuda.stream(MAIN)
for name, st in sts.items():                 # warm: graphs captured at the third call
    for it in range(4):
        TC.seed_all(800 + it)
        with main_ctx():
            st.step(*TC._acdc_batch(20 + it))
    torch.cuda.synchronize()
snaps = {name: snapshot(st) for name, st in sts.items()}
# probes: the InfoNCE's precomputed anchor gradient and its inputs (head forward rows), the heads' incoming gradient
from arco_amd import _contrast as C_, head as H_
probe = {}
real_cg = C_._CompactGrad.apply
def cg(A_all, loss, dA_all):
    probe["A_all"] = A_all.detach().clone(); probe["dA_all"] = dA_all.detach().clone(); probe["reco"] = loss.detach().clone()
    return real_cg(A_all, loss, dA_all)
C_._CompactGrad.apply = cg
_row_grad_buffer, _wgrad, _gemm_t, _fea_rows = H_._row_grad_buffer, H_._wgrad, H_._gemm_t, H_._fea_rows
from arco_amd import _lib as L
def dbg_backward(ctx, da):
    X2, X3, X4, h0, hh, w2, w3, w4, w1, wq2, pix, nb4, nb16, lylx3, lylx4 = ctx.saved_tensors
    nb, c1, h1, w1_, c2, h2, w2_, c3, h3, w3_, c4, h4, w4_ = ctx.geom
    dev = da.device
    n = int(pix.shape[0])
    k2, k3 = c1 + c2, c1 + c2 + c3
    def rec(name, t):
        probe[name] = t.detach().clone().float()
    for nm, t in (("s_X2", X2), ("s_X3", X3), ("s_X4", X4), ("s_h0", h0), ("s_hh", hh), ("s_pix", pix), ("s_nb4", nb4), ("s_nb16", nb16),
                  ("s_lylx3", lylx3), ("s_lylx4", lylx4), ("w2", w2), ("w3", w3), ("w4", w4)):
        rec(nm, t)
    da = da.contiguous(); rec("b0_da", da)
    dwq2 = _wgrad(da, hh, wq2)
    dhh = _gemm_t(da, wq2); rec("b1_dhh", dhh)
    dw1 = _wgrad(dhh, h0, w1)
    dh0 = _gemm_t(dhh, w1); rec("b2_dh0", dh0)
    dw4 = _wgrad(dh0, X4, w4)
    dX4 = _gemm_t(dh0, w4); rec("b3_dX4", dX4)
    dX3p = torch.empty((4 * n, k3), dtype=torch.float32, device=dev)
    if seg_log[0] < 6:
        seg_log[0] += 1
        ptr = dX3p.data_ptr()
        for seg in torch.cuda.memory_snapshot():
            if seg["address"] <= ptr < seg["address"] + seg["total_size"]:
                blocks = [(b_["state"], b_["size"]) for b_ in seg["blocks"]]
                print(f"  SEGMENT of dX3p: pool_id {seg.get('segment_pool_id')} stream {seg.get('stream')} type {seg.get('segment_type')} size {seg['total_size']} "
                      f"blocks {len(blocks)} active {sum(1 for st_, _ in blocks if st_.startswith('active'))}; current stream {torch.cuda.current_stream().cuda_stream:#x} "
                      f"default {torch.cuda.default_stream().cuda_stream:#x}", flush=True)
    df4, fin4 = _row_grad_buffer(ctx.fptrs[2], (nb, c4, h4, w4_), dev)
    L.call("arco_lerp4_cat_rows_bwd", L.ptr(dX4), k3 + c4, k3, L.ptr(lylx4), L.ptr(pix), n, L.ptr(dX3p), k3, L.ptr(df4), c4, c4)
    rec("b4_dX3p", dX3p)
    fin4(pix, n)
    dw3 = _wgrad(dX3p, X3, w3)
    dX3 = _fea_rows(dX3p, w3, 1); rec("b5_dX3", dX3)
    dX2p = torch.empty((16 * n, k2), dtype=torch.float32, device=dev)
    df3, fin3 = _row_grad_buffer(ctx.fptrs[1], (nb, c3, h3, w3_), dev)
    L.call("arco_lerp4_cat_rows_bwd", L.ptr(dX3), k3, k2, L.ptr(lylx3), L.ptr(nb4), 4 * n, L.ptr(dX2p), k2, L.ptr(df3), c3, c3)
    rec("b6_dX2p", dX2p)
    fin3(nb4, 4 * n)
    dw2 = _wgrad(dX2p, X2, w2)
    dX2 = _fea_rows(dX2p, w2, 1); rec("b7_dX2", dX2)
    dx1p = torch.zeros((nb, h1, w1_, c1), dtype=torch.float32, device=dev)
    df2, fin2 = _row_grad_buffer(ctx.fptrs[0], (nb, c2, h2, w2_), dev)
    L.call("arco_scatter_upcat_rows", L.ptr(dX2), k2, L.ptr(nb16), 16 * n, L.ptr(dx1p), c1, c1, h1, w1_, L.ptr(df2), c2, c2, h2, w2_)
    rec("b8_dx1p", dx1p)
    fin2(nb16, 16 * n)
    return (dx1p.permute(0, 3, 1, 2), df2.permute(0, 3, 1, 2), df3.permute(0, 3, 1, 2), df4.permute(0, 3, 1, 2),
            dw2, dw3, dw4, dw1, dwq2, None)
H_.LazyHead3Fn.backward = staticmethod(dbg_backward)
ref_probe = {}
seg_log = [0]
dumped = [False]
def cmp_probe(name):
    out = []
    for k, v in probe.items():
        vs = v if isinstance(v, list) else [v]
        if (name, k) not in ref_probe:
            ref_probe[(name, k)] = [t.clone() for t in vs]
        for i, (t, r) in enumerate(zip(vs, ref_probe[(name, k)])):
            d = float((t - r).abs().max()) / max(1e-30, float(r.abs().max()))
            if d > 1e-5:
                out.append(f"{k}[{i}] {d:.1e}")
                if k.startswith("b") and not dumped[0]:
                    dumped[0] = True
                    bad = (t != r)
                    rows = bad.any(1).nonzero().flatten()
                    cols = bad.any(0).nonzero().flatten()
                    print(f"  FINGERPRINT {k}: {int(bad.sum())} of {bad.numel()} elements differ; rows {int(rows.numel())} (min {int(rows.min())} max {int(rows.max())}), "
                          f"cols {int(cols.numel())} (min {int(cols.min())} max {int(cols.max())}); ref |max| {float(r.abs().max()):.3e}, bad values |max| {float(t[bad].abs().max()):.3e} "
                          f"mean|bad| {float(t[bad].abs().mean()):.3e} mean|ref at bad| {float(r[bad].abs().mean()):.3e}; zeros among bad {int((t[bad] == 0).sum())}; "
                          f"ratio sample {(t[bad][:8] / r[bad][:8]).tolist()}", flush=True)
                    rr = rows.tolist()
                    runs, start = [], rr[0]
                    for a_, b_ in zip(rr, rr[1:] + [None]):
                        if b_ != a_ + 1:
                            runs.append((start, a_)); start = b_
                    print(f"  row runs (first 12 of {len(runs)}): {runs[:12]}", flush=True)
    return " ".join(out)
batch = TC._acdc_batch(24)
names = {name: [k for k, _ in st.model.named_parameters()] for name, st in sts.items()}
ref = {}
for t in range(n):
    line = []
    for name, st in sts.items():
        restore(st, snaps[name])
        TC.seed_all(804)
        torch.cuda.synchronize()
        with main_ctx():
            st.step(*batch)
        torch.cuda.synchronize()
        g = st.optimizer.flat_g.clone()
        if name not in ref:
            ref[name] = g
        d = (g - ref[name]).abs()
        worst = float(d.max()) / float(ref[name].abs().max())
        line.append(f"{name} {worst:.1e}")
        pr = cmp_probe(name)
        if pr:
            line.append(" PROBE " + pr)
        if worst > 1e-5:
            # where: per-parameter
            dev_ = []
            for (off, k), p in zip(st.optimizer.offsets, st.optimizer.params):
                r = ref[name][off:off + k]
                dev_.append((float(d[off:off + k].max()) / max(1e-20, float(r.abs().max())), off))
            dev_.sort(reverse=True)
            line.append("  [" + ", ".join(f"@{o} {v:.1e}" for v, o in dev_[:5]) + f"; {sum(v > 1e-5 for v, _ in dev_)} of {len(dev_)} params off]")
    print(f"trial {t}: " + "   ".join(line), flush=True)
