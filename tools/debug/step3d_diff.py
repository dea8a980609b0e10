"""Debug aid: one 3-D step, HIP path vs the CPU oracle, with the intermediate tensors side by side."""
import random, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import cpu_step3d, fixture_inputs as fx
from arco_amd import ops, train_arco_3d as T3
import test_step3d_parity_gpu as TT

C = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dense = int(sys.argv[2]) if len(sys.argv) > 2 else 1
b, patch, Q, Nn, qs, lr = 2, (32, 32, 32), 48, 16, 200, 0.01
vnet_sd = TT._state(C); fe_sd = fx.fe_state(61, TT.FEA, 16, nd=3); qrep_w = [TT._qrep_w(71), TT._qrep_w(72)]
args = T3.build_parser().parse_args(["--batch_size", str(b), "--queue_size", str(qs), "--synthetic", "1", "--num_classes", str(C),
    "--num_queries", str(Q), "--num_negatives", str(Nn), "--k1", "1.0", "--base_lr", str(lr), "--dense_head", str(dense), "--graphs", "0"])
args.patch_size = list(patch)
random.seed(3); np.random.seed(3); torch.manual_seed(3)
st = T3.ArcoStep3D(args, "cuda:0")
st.model.load_state_dict(vnet_sd); st.ema_model.load_state_dict(vnet_sd)
st.q_feature_extractor.load_state_dict(fe_sd); st.k_feature_extractor.load_state_dict(fe_sd)
with torch.no_grad():
    st.q_representation[0].weight.copy_(qrep_w[0]); st.q_representation[1].weight.copy_(qrep_w[1])
for m in (st.model, st.ema_model): TT._drop_off(m)
ops.bump_weight_epoch()
st.keep_debug = True
so = cpu_step3d.make_state(vnet_sd, fe_sd, qrep_w, base_lr=lr)
bank_o = [[m[0].detach().cpu().clone()] for m in st.memobank]; ptr_o = [torch.zeros(1, dtype=torch.long) for _ in range(C)]
rs = np.random.RandomState(13)
for it in range(2):
    l, lab, u = TT._volumes(rs, b, patch, C)
    random.seed(10 + it); np.random.seed(10 + it); torch.manual_seed(10 + it)
    tr = {}
    cpu_step3d.step(so, l, lab, u, bank_o, ptr_o, list(st.queue_size), n_cls=C, k1=1.0, nq=Q, nn_=Nn, trace=tr)
    random.seed(10 + it); np.random.seed(10 + it); torch.manual_seed(10 + it)
    st.step(l.cuda(), lab.cuda(), u.cuda())
    pl = st.debug["plan"]
    print("it", it, "terms", {k: float(v) for k, v in st.last_terms.items()}, so["last_terms"])
    print(" gpu n_lv", pl.n_lv, "n_anchor", pl.n_anchor, "n_neg", pl.n_neg)
    print(" cpu anchor", [int(a.numel()) for a in tr["anchor_rows"]], "neg", [int(a.numel()) for a in tr["neg_rows"]])
    print(" low sum", float(tr["low"].sum()), "high sum", float(tr["high"].sum()))
    for k, (a, n) in enumerate(zip(tr.get("anchor_idx", []), tr.get("neg_idx", []))):
        e = pl.entries[k]
        print("  entry", k, "anchor idx equal", torch.equal(e[2].cpu(), a), "neg idx equal", torch.equal(e[3].cpu(), n))
    A = st.debug["A_all"].cpu()
    # the oracle's anchors: rep rows at the anchor candidates
    rep = tr["rep_all"].permute(0, 2, 3, 4, 1).reshape(-1, 16)
    for k, a in enumerate(tr.get("anchor_idx", [])):
        rows = tr["anchor_rows"][pl.entries[k][0]][a]
        ref = rep[rows]
        got = A[k * Q:(k + 1) * Q]
        print("  anchors class", k, "max rel err", float((got - ref).abs().max() / ref.abs().max()))
    print("  proto gpu", pl.proto.cpu()[:, :4])
    rt = tr["rep_t"].permute(0, 2, 3, 4, 1).reshape(-1, 16)
