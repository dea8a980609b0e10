"""3-D twin of self_consistency.py: the same step from one snapshot, repeated; each run's flat gradient against the first run's.
python tools/debug/self_consistency3d.py [trials] [PASS_SIDE] [lits]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import random, numpy as np, torch
from arco_amd import train_arco_3d as T3, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
if len(sys.argv) > 2:
    T3.PASS_SIDE = int(sys.argv[2])
lits = "lits" in sys.argv
b = 1 if lits else 2
def seed_all(s):
    random.seed(s); np.random.seed(s); torch.manual_seed(s)
seed_all(7)
args = T3.build_parser().parse_args(["--batch_size", str(b), "--queue_size", "4096", "--synthetic", "1", "--num_classes", "4" if not lits else "2",
                                     "--conv_mma", "f32x3", "--act_dtype", "f16" if lits else "f32", "--k1", "1.0"])
args.patch_size = [160, 160, 96] if lits else [112, 112, 80]
st = T3.ArcoStep3D(args, "cuda:0")
for m in (st.model, st.ema_model):
    for mod in m.modules():
        if isinstance(mod, (torch.nn.Dropout, torch.nn.Dropout3d)):
            mod.p = 0.0
    if hasattr(m, "has_dropout"):
        m.has_dropout = False
C = args.num_classes
def batch(i):
    l, ll = T3.synthetic_volume_batch(b, args.patch_size, C, 10 + i, "cuda:0")
    u, _ = T3.synthetic_volume_batch(b, args.patch_size, C, 20 + i, "cuda:0")
    return l, ll, u
def snapshot():
    return dict(p=st.optimizer.flat_p.clone(), b=st.optimizer.flat_buf.clone(), started=list(st.optimizer._started),
                lr=[g['lr'] for g in st.optimizer.param_groups],
                sd=[{k: v.clone() for k, v in m.state_dict().items()} for m in (st.model, st.ema_model, st.k_feature_extractor)],
                bank=[[t.clone() for t in m] for m in st.memobank], ptr=[q.clone() if torch.is_tensor(q) else q for q in st.queue_ptrlis],
                it=st.iter_num, scale=ops.LOSS_SCALE)
def restore(s):
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
for it in range(4):
    seed_all(800 + it); st.step(*batch(it))
torch.cuda.synchronize()
snap = snapshot()
bt = batch(4)
ref = None
nbad = 0
for t in range(n):
    restore(snap)
    seed_all(804)
    st.step(*bt)
    torch.cuda.synchronize()
    g = st.optimizer.flat_g.clone()
    if ref is None:
        ref = g
    worst = float((g - ref).abs().max()) / float(ref.abs().max())
    if worst > 1e-5:
        nbad += 1
        print(f"trial {t}: {worst:.1e}", flush=True)
print(f"PASS_SIDE {T3.PASS_SIDE} {'lits f16' if lits else 'LA'}: {nbad} of {n} trials off (> 1e-5 of the largest gradient)")
