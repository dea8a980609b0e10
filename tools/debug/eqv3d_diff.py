"""Debug aid: the equivariance block of the 3-D step, piece by piece, HIP vs oracle."""
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
for m in (st.model, st.ema_model): TT._drop_off(m)
ops.bump_weight_epoch()
so = cpu_step3d.make_state(vnet_sd, fe_sd, qrep_w, base_lr=lr)
bank_o = [[m[0].detach().cpu().clone()] for m in st.memobank]; ptr_o = [torch.zeros(1, dtype=torch.long) for _ in range(C)]
rs = np.random.RandomState(13)
l, lab, u = TT._volumes(rs, b, patch, C)
random.seed(10); np.random.seed(10); torch.manual_seed(10)
tr = {}
cpu_step3d.step(so, l, lab, u, bank_o, ptr_o, list(st.queue_size), n_cls=C, k1=1.0, nq=Q, nn_=Nn, trace=tr)
print("oracle eqv", so["last_terms"]["eqv"])
def rel(a, b): return float((a.cpu() - b).abs().max() / b.abs().max())
# 1. the loss kernel on the oracle's tensors
e = glue.eqv_loss(tr["pred_tps"].cuda(), tr["org"].cuda(), tr["mask_tps"].cuda())
print("kernel on oracle tensors", float(e))
# 2. the warp on the oracle's grid
st.tps.grid.data.copy_(tr["grid"].cuda()) if hasattr(st.tps, "grid") else None
print("tps attrs", [k for k in vars(st.tps) if not k.startswith('_')])
with torch.no_grad():
    g_imgs = st.tps(torch.cat((l, tr["images_tps"][:0].new_zeros(0, 1, *patch))).cuda()) if False else None
    x_all = torch.cat((l, u))   # NB the oracle warps (l, u_aug); compare the warp op itself on the oracle's own inputs below
    w_mask = st.tps(tr["eqv_mask"].cuda(), padding_mode='zeros')
    print("warp(mask) vs oracle", rel(w_mask, tr["mask_tps"]), "sum", float(w_mask.sum()), float(tr["mask_tps"].sum()))
    w_org_in = torch.randn(4, C, *patch)
    w1 = st.tps(w_org_in.cuda(), padding_mode='zeros'); w2 = cpu_step3d.warp_volume(w_org_in, tr["grid"])
    print("warp(random 4ch) vs oracle", rel(w1, w2))
    w1 = st.tps(w_org_in[:, :1].contiguous().cuda()); w2 = cpu_step3d.warp_volume(w_org_in[:, :1].contiguous(), tr["grid"])
    print("warp(random 1ch) vs oracle", rel(w1, w2))
st.model.train()
with ops.logits_only():
    pg = st.model(tr["images_tps"].cuda())[0]
print("student(images_tps) vs oracle pred_tps", rel(pg.detach(), tr["pred_tps"]))
e2 = glue.eqv_loss(pg.detach(), tr["org"].cuda(), tr["mask_tps"].cuda())
print("eqv with product pred_tps", float(e2))
pg2 = st.model(tr["images_tps"].cuda())[0]
print("student(images_tps) full outputs vs oracle pred_tps", rel(pg2.detach(), tr["pred_tps"]))
