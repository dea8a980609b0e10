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
    lerp4_bwd("dX3p", dX4, k3 + c4, k3, lylx4, pix, n, dX3p, df4, c4)
    rec("b4_dX3p", dX3p)
    fin4(pix, n)
    dw3 = _wgrad(dX3p, X3, w3)
    dX3 = _fea_rows(dX3p, w3, 1); rec("b5_dX3", dX3)
    dX2p = torch.empty((16 * n, k2), dtype=torch.float32, device=dev)
    df3, fin3 = _row_grad_buffer(ctx.fptrs[1], (nb, c3, h3, w3_), dev)
    lerp4_bwd("dX2p", dX3, k3, k2, lylx3, nb4, 4 * n, dX2p, df3, c3)
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

# ---- round 5 probes (profiles/r05_notes.md section 1): canary pre-fill, expected-value check and an instrumented kernel twin
import ctypes
SC_CANARY = int(os.environ.get("SC_CANARY", "0"))
SC_DBG_LERP = int(os.environ.get("SC_DBG_LERP", "-1"))          # -1: the product kernel; 0 / 1 / 2: tools/debug/lerp4_dbg.hip variants
SC_CHECK = int(os.environ.get("SC_CHECK", "0"))                  # compare each adjoint output with its torch restatement
_dbglib = None
SC_ISA = os.environ.get("SC_ISA", "")                             # a hand-edited ISA variant of the product kernel: tools/debug/isa/<name>.co
if SC_DBG_LERP >= 0 or SC_ISA:
    _dbglib = ctypes.CDLL(os.path.join(ROOT, "tools", "debug", "liblerp4dbg.so"))
    _P, _I, _Lg = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    _dbglib.dbg_lerp4_cat_rows_bwd.argtypes = [_I, _P, _Lg, _I, _P, _P, _Lg, _P, _Lg, _P, _Lg, _I, _P, _P]
    _dbglib.dbg_lerp4_cat_rows_bwd.restype = _I
    _dbglib.dbg_lerp4_module_launch.argtypes = [_P, _Lg, _I, _P, _P, _Lg, _P, _Lg, _P, _Lg, _I, _P, _P]
    _dbglib.dbg_lerp4_module_launch.restype = _I
    if SC_ISA:
        torch.cuda.init(); torch.zeros(1, device="cuda")
        rc = _dbglib.dbg_lerp4_module_load(os.path.join(ROOT, "tools", "debug", "isa", SC_ISA + ".co").encode())
        assert rc == 0, rc
CANARY_BITS = 0x7fc0dead
lerp_events = [0]
def lerp4_bwd(tag, dXin, ldx, Clo, lylx, pix, n, dV, df, Chi):
    dbg = None
    if SC_CANARY:
        dV.view(torch.int32).fill_(CANARY_BITS)
    if SC_ISA:
        if "record" in SC_ISA:
            dbg = torch.zeros((n * 64, 64), dtype=torch.int32, device=dV.device)
        rc = _dbglib.dbg_lerp4_module_launch(L.ptr(dXin), ldx, Clo, L.ptr(lylx), L.ptr(pix), n, L.ptr(dV), Clo, L.ptr(df), Chi, Chi, L.ptr(dbg), L.stream())
        assert rc == 0
    elif _dbglib is not None:
        dbg = torch.zeros((n * 64, 16), dtype=torch.int32, device=dV.device)
        rc = _dbglib.dbg_lerp4_cat_rows_bwd(SC_DBG_LERP, L.ptr(dXin), ldx, Clo, L.ptr(lylx), L.ptr(pix), n, L.ptr(dV), Clo, L.ptr(df), Chi, Chi,
                                            L.ptr(dbg), L.stream())
        assert rc == 0
    else:
        L.call("arco_lerp4_cat_rows_bwd", L.ptr(dXin), ldx, Clo, L.ptr(lylx), L.ptr(pix), n, L.ptr(dV), Clo, L.ptr(df), Chi, Chi)
    if not SC_CHECK:
        return
    if SC_CHECK == 2:       # no host synchronisation inside backward(): snapshot now (stream-ordered copies), compare after the step
        pending_checks.append((tag, dXin.clone(), Clo, lylx, n, dV.clone(), dbg))
        return
    lerp4_check(tag, dXin, Clo, lylx, n, dV, dbg)
pending_checks = []
def lerp4_check(tag, dXin, Clo, lylx, n, dV, dbg):
    ly, lx = lylx[0:2 * n:2], lylx[1:2 * n:2]
    wts = torch.stack(((1 - ly) * (1 - lx), (1 - ly) * lx, ly * (1 - lx), ly * lx), 1)
    exp = (dXin[:n, None, :Clo] * wts[:, :, None]).reshape(4 * n, Clo)
    bad = ~((dV == exp) | ((dV == 0) & (exp == 0)))
    nbad = int(bad.sum())
    if nbad == 0:
        return
    lerp_events[0] += 1
    if lerp_events[0] > 6:
        return
    rows, cols = bad.nonzero(as_tuple=True)
    vb = dV[bad]
    bits = vb.view(torch.int32)
    lanes = (cols % 256) // 4
    passes = cols // 256
    urows = torch.unique(rows)
    print(f"  LERP4 {tag}: {nbad} bad elements in {int(urows.numel())} rows of {4 * n}; canary left {int((bits == CANARY_BITS).sum())}, zeros {int((vb == 0).sum())}, "
          f"other {int(((bits != CANARY_BITS) & (vb != 0)).sum())}; row%4 hist {torch.bincount(urows % 4, minlength=4).tolist()}; "
          f"lane hist (16-lane groups) {torch.bincount(lanes // 16, minlength=4).tolist()}; pass hist {torch.bincount(passes, minlength=2).tolist()}; "
          f"per-row counts {torch.unique(torch.bincount(rows)[urows]).tolist()}", flush=True)
    oth = (bits != CANARY_BITS) & (vb != 0)
    if int(oth.sum()):
        print(f"    other values: got {vb[oth][:6].tolist()} expected {exp[bad][oth][:6].tolist()} ratio {(vb[oth][:6] / exp[bad][oth][:6]).tolist()}", flush=True)
    if dbg is not None and dbg.shape[1] == 64:        # the ISA variant that records its own registers (isa/v5_record.s)
        d = dbg.view(n, 64, 64)
        j, t, e, p_ = rows // 4, rows % 4, cols % 4, cols // 256
        recd = d[j, lanes]                                                         # [nbad, 64]
        widx = torch.tensor([[2, 3, 10, 11], [4, 5, 12, 13], [6, 7, 14, 15], [0, 1, 8, 9]], device=dV.device)[t, e]      # weight register of (t, e)
        wrec = recd.gather(1, widx[:, None])[:, 0].contiguous().view(torch.float32)
        wexp = wts[j, t]
        didx = 16 + p_ * 16 + t * 4 + e
        drec = recd.gather(1, didx[:, None])[:, 0].contiguous().view(torch.float32)
        print(f"    recorded WEIGHT register of the bad element: zero at {int((wrec == 0).sum())} of {nbad}, equal to the expected weight at {int((wrec == wexp).sum())}; "
              f"recorded PRODUCT register (store data): zero at {int((drec == 0).sum())}, equal to expected at {int((drec == exp[bad]).sum())}", flush=True)
        # the same registers in the GOOD lanes of the bad waves, and ALL 16 weight registers of the bad lanes
        bj = torch.unique(j)
        wall = d[bj][:, :, 0:16].contiguous().view(torch.float32)               # [waves, 64 lanes, 16 regs v12..v27]
        zero_map = (wall == 0)
        print(f"    zero weight registers by 16-lane group (waves x regs v12..v27), bad waves {int(bj.numel())}: "
              f"{[zero_map[:, g * 16:(g + 1) * 16, :].any(1).sum(0).tolist() for g in range(4)]}", flush=True)
        uniform = (wall == wall[:, :1, :]).all(1)                                # register wave-uniform?
        print(f"    weight registers wave-uniform (count of bad waves per register): {uniform.sum(0).tolist()}", flush=True)
        hw = d[bj, 0, 48].to(torch.int64) & 0xffffffff; xcc = d[bj, 0, 49].to(torch.int64) & 0xf
        hw_all = d[:, 0, 48].to(torch.int64) & 0xffffffff; xcc_all = d[:, 0, 49].to(torch.int64) & 0xf
        print(f"    bad waves: xcc hist {torch.bincount(xcc, minlength=8).tolist()} (all {torch.bincount(xcc_all, minlength=8).tolist()}); simd {torch.bincount((hw >> 4) & 3, minlength=4).tolist()}; "
              f"wave slot {torch.bincount(hw & 15, minlength=16).tolist()}; se {torch.bincount((hw >> 13) & 7, minlength=8).tolist()}; cu {torch.bincount((hw >> 8) & 15, minlength=16).tolist()}; "
              f"queue bad {torch.unique((hw >> 24) & 7).tolist()} all {torch.unique((hw_all >> 24) & 7).tolist()}; pipe bad {torch.unique((hw >> 6) & 3).tolist()} all {torch.unique((hw_all >> 6) & 3).tolist()}; "
              f"vmid all {torch.unique((hw_all >> 20) & 15).tolist()}; me all {torch.unique((hw_all >> 30) & 3).tolist()}", flush=True)
        k = int(j[0]); ln = int(lanes[0])
        print(f"    sample: wave j={k} lane {ln} weights v12..v27 {d[k, ln, 0:16].contiguous().view(torch.float32).tolist()}; lane 0 of the same wave {d[k, 0, 0:16].contiguous().view(torch.float32).tolist()}; "
              f"expected (w0,w1,w2,w3) {wts[k].tolist()} ly,lx {float(ly[k])},{float(lx[k])}", flush=True)
    elif dbg is not None:
        d = dbg.view(n, 64, 16)
        j, t = rows // 4, rows % 4
        rec = d[j, lanes]                                   # [nbad, 16]
        wrec = rec[:, 0:4].contiguous().view(torch.float32)
        wexp = wts[j]
        w_wrong = (wrec != wexp).any(1)
        w_t = wrec.gather(1, t[:, None])[:, 0]
        print(f"    recorded weights differ from expected at {int(w_wrong.sum())} of {nbad} bad elements; recorded weight of the bad row is 0 at {int((w_t == 0).sum())}; "
              f"expected weight of the bad row is 0 at {int((wexp.gather(1, t[:, None])[:, 0] == 0).sum())}", flush=True)
        aexp = ((4 * j + t) * Clo + lanes * 4) * 4
        arec = rec.gather(1, (4 + t)[:, None])[:, 0].to(torch.int64) & 0xffffffff
        print(f"    recorded row address differs from expected at {int((arec != (aexp & 0xffffffff)).sum())}; recorded lane id wrong at {int(((rec[:, 13] & 63) != lanes).sum())}; "
              f"recorded ly/lx wrong at {int((rec[:, 14:16].contiguous().view(torch.float32) != torch.stack((ly[j], lx[j]), 1)).any(1).sum())}", flush=True)
        allw = d[:, 0, :]
        hw_all, xcc_all = allw[:, 8].to(torch.int64) & 0xffffffff, allw[:, 9].to(torch.int64) & 0xf
        bj = torch.unique(j)
        hb = hw_all[bj]
        print(f"    bad waves {int(bj.numel())}: xcc hist {torch.bincount(xcc_all[bj], minlength=8).tolist()} (all waves {torch.bincount(xcc_all, minlength=8).tolist()}); "
              f"simd hist {torch.bincount((hb >> 4) & 3, minlength=4).tolist()}; se hist {torch.bincount((hb >> 13) & 7, minlength=8).tolist()}; cu hist {torch.bincount((hb >> 8) & 15, minlength=16).tolist()}; "
              f"queue ids bad {torch.unique((hb >> 24) & 7).tolist()} all {torch.unique((hw_all >> 24) & 7).tolist()}; pipe ids bad {torch.unique((hb >> 6) & 3).tolist()} all {torch.unique((hw_all >> 6) & 3).tolist()}; "
              f"me bad {torch.unique((hb >> 30) & 3).tolist()}; vmid {torch.unique((hw_all >> 20) & 15).tolist()}", flush=True)
        tt0, tt1 = allw[:, 10].to(torch.int64) & 0xffffffff, allw[:, 11].to(torch.int64) & 0xffffffff
        dur = (tt1 - tt0) & 0xffffffff
        print(f"    wave duration (clock ticks): bad median {int(dur[bj].median())} max {int(dur[bj].max())}; all median {int(dur.median())} max {int(dur.max())}; "
              f"kernel span {int(((tt1.max() - tt0.min()) & 0xffffffff))}; bad waves start offsets (first 8) {((tt0[bj] - tt0.min()) & 0xffffffff)[:8].tolist()}", flush=True)
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
        for chk in pending_checks:
            lerp4_check(*chk)
        pending_checks.clear()
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
