"""Debug aid: capture the inputs of the equivariance loss inside the real 3-D step and compare with the oracle's."""
import random, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import cpu_step3d, fixture_inputs as fx, arco_oracle as orc
from arco_amd import ops, glue, train_arco_3d as T3
import test_step3d_parity_gpu as TT
C = 4
b, patch, Q, Nn, qs, lr = 2, (32, 32, 32), 48, 16, 200, 0.01
vnet_sd = TT._state(C); fe_sd = fx.fe_state(61, TT.FEA, 16, nd=3); qrep_w = [TT._qrep_w(71), TT._qrep_w(72)]
args = T3.build_parser().parse_args(["--batch_size", str(b), "--queue_size", str(qs), "--synthetic", "1", "--num_classes", str(C),
    "--num_queries", str(Q), "--num_negatives", str(Nn), "--k1", "1.0", "--base_lr", str(lr), "--graphs", "0"])
args.patch_size = list(patch)
random.seed(3); np.random.seed(3); torch.manual_seed(3)
st = T3.ArcoStep3D(args, "cuda:0")
st.model.load_state_dict(vnet_sd); st.ema_model.load_state_dict(vnet_sd)
st.q_feature_extractor.load_state_dict(fe_sd); st.k_feature_extractor.load_state_dict(fe_sd)
with torch.no_grad():
    st.q_representation[0].weight.copy_(qrep_w[0]); st.q_representation[1].weight.copy_(qrep_w[1])
for m in (st.model, st.ema_model): TT._drop_off(m)
ops.bump_weight_epoch()
so = cpu_step3d.make_state(vnet_sd, fe_sd, qrep_w, base_lr=lr)
bank_o = [[m[0].detach().cpu().clone()] for m in st.memobank]; ptr_o = [torch.zeros(1, dtype=torch.long) for _ in range(C)]
rs = np.random.RandomState(13)
l, lab, u = TT._volumes(rs, b, patch, C)
import hashlib
def rng_sig():
    return (hashlib.md5(torch.get_rng_state().numpy().tobytes()).hexdigest()[:8], hashlib.md5(repr(random.getstate()).encode()).hexdigest()[:8],
            hashlib.md5(repr(np.random.get_state()[1].tolist()).encode()).hexdigest()[:8], int(np.random.get_state()[2]))
real_src = orc.rand_tps_source_points
def spy_src(*a, **k):
    print("oracle rng before warp", rng_sig()); return real_src(*a, **k)
orc.rand_tps_source_points = spy_src
real_o_loss = orc.compute_contra_memobank_loss
def spy_o_loss(*a, **k):
    print("oracle rng before loss", rng_sig()); r = real_o_loss(*a, **k); print("oracle rng after loss", rng_sig()); return r
orc.compute_contra_memobank_loss = spy_o_loss
random.seed(10); np.random.seed(10); torch.manual_seed(10)
tr = {}
cpu_step3d.step(so, l, lab, u, bank_o, ptr_o, list(st.queue_size), n_cls=C, k1=1.0, nq=Q, nn_=Nn, trace=tr)
cap = {}
real = glue.eqv_loss
def spy(pred_tps, org, mask):
    cap.update(pred_tps=pred_tps.detach().cpu(), org=org.detach().cpu(), mask=mask.detach().cpu())
    return real(pred_tps, org, mask)
glue.eqv_loss = spy
real_mask = glue.eqv_mask
def spy_mask(labels, logits, thr):
    m = real_mask(labels, logits, thr)
    cap.update(eq_mask=m.detach().cpu(), labels=labels.cpu(), logits=logits.cpu())
    return m
glue.eqv_mask = spy_mask
real_reset = st.tps.reset_control_points
def spy_reset():
    print("product rng before warp", rng_sig()); return real_reset()
st.tps.reset_control_points = spy_reset
from arco_amd import _contrast as C_
real_draw = C_.contrast_draw
def spy_draw(pl, *a, **k):
    print("product rng before draw", rng_sig(), "n_anchor", pl.n_anchor, "bank_len", pl.bank_len, "valid", pl.valid_classes)
    r = real_draw(pl, *a, **k); print("product rng after draw", rng_sig()); return r
C_.contrast_draw = spy_draw
random.seed(10); np.random.seed(10); torch.manual_seed(10)
st.step(l.cuda(), lab.cuda(), u.cuda())
def rel(a, b): return float((a.float() - b.float()).abs().max() / b.float().abs().max())
print("oracle anchors", [int(a.numel()) for a in tr["anchor_rows"]], "banks", [int(x[0].shape[0]) for x in bank_o])
print("eqv", float(st.last_terms["eqv"]), so["last_terms"]["eqv"])
print("grid", rel(st.tps.grid.data.cpu(), tr["grid"]))
print("eq_mask", rel(cap["eq_mask"], tr["eqv_mask"]), float(cap["eq_mask"].sum()), float(tr["eqv_mask"].sum()))
print("mask_tps", rel(cap["mask"], tr["mask_tps"]))
print("org", rel(cap["org"], tr["org"]))
print("pred_tps", rel(cap["pred_tps"], tr["pred_tps"]))
d = (cap["pred_tps"] - tr["pred_tps"]).abs()
print(" pred_tps diff: max at", np.unravel_index(int(d.argmax()), d.shape), "count > 1e-3:", int((d > 1e-3 * tr["pred_tps"].abs().max()).sum()))
d = (cap["org"] - tr["org"]).abs()
print(" org diff: count > 1e-3:", int((d > 1e-3 * tr["org"].abs().max()).sum()))
