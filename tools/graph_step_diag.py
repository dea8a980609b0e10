"""Eager vs graphed trainer, step by step (which term diverges first, under which graph subset)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
import fixture_inputs as fx
from arco_amd import train_arco_2d as T, ops

def drop_off(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout): mod.p = 0.0

def trainer(graphs):
    b, patch, C = 2, (64, 64), 4
    argv = ["--batch_size", str(b), "--queue_size", "300", "--synthetic", "1", "--num_queries", "64", "--num_negatives", "32",
            "--k1", "1.0", "--base_lr", "0.01", "--graphs", str(graphs), "--graph_train", str(graphs)]
    args = T.build_parser().parse_args(argv); args.patch_size = list(patch)
    st = T.ArcoStep2D(args, "cuda:0")
    sd, fe = fx.unet_state(21, 1, C), fx.fe_state(31)
    st.model.load_state_dict(sd); st.ema_model.load_state_dict(sd)
    st.q_feature_extractor.load_state_dict(fe); st.k_feature_extractor.load_state_dict(fe)
    with torch.no_grad():
        st.q_representation[0].weight.copy_(fx.fe_state(32)["fea4.weight"]); st.q_representation[1].weight.copy_(fx.fe_state(33)["fea4.weight"])
    drop_off(st.model); drop_off(st.ema_model)
    ops.bump_weight_epoch()
    return st

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
st_e, st_g = trainer(0), trainer(1)
if mode == "train_only":
    for g in (st_g.t_fwd_u0, st_g.t_fwd_l, st_g.t_fwd_u, st_g.s_fwd_stats): g.enabled = False
if mode == "nograd_only":
    st_g.s_train_u.enabled = st_g.s_train_l.enabled = False
if mode == "train_u_only":
    st_g.s_train_l.enabled = False
    for g in (st_g.t_fwd_u0, st_g.t_fwd_l, st_g.t_fwd_u, st_g.s_fwd_stats): g.enabled = False
if mode == "train_l_only":
    st_g.s_train_u.enabled = False
    for g in (st_g.t_fwd_u0, st_g.t_fwd_l, st_g.t_fwd_u, st_g.s_fwd_stats): g.enabled = False
rs = np.random.RandomState(5)
for it in range(6):
    l = torch.from_numpy(rs.uniform(size=(2, 1, 64, 64)).astype(np.float32)).cuda()
    u = torch.from_numpy(rs.uniform(size=(2, 1, 64, 64)).astype(np.float32)).cuda()
    lab = torch.from_numpy(fx.blob_labels(rs, 2, (64, 64), 4)).cuda()
    out = []
    for st in (st_e, st_g):
        random.seed(10 + it); np.random.seed(10 + it); torch.manual_seed(10 + it)
        st.step(l, lab, u)
        out.append({k: float(v) for k, v in st.last_terms.items()})
    wd = max(float((a - b).abs().max()) for a, b in zip(st_e.optimizer.flat_p.split(100000), st_g.optimizer.flat_p.split(100000)))
    rm = max(float((a.float() - b.float()).abs().max()) for a, b in zip(st_e.model.buffers(), st_g.model.buffers()))
    print(mode, it, {k: f"{out[1][k] - out[0][k]:+.2e}" for k in out[0]}, "flat_p maxdiff", wd, "buffers maxdiff", rm)
