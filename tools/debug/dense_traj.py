"""Loss terms of N steps of the dense reference dataflow from one seed (ARCO_SPARSE_BWD=0/1 must give the same trajectory up to rounding).
python tools/debug/dense_traj.py [steps]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import random, numpy as np, torch
from arco_amd import train_arco_2d as T, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
random.seed(1337); np.random.seed(1337); torch.manual_seed(1337)
args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--synthetic", "1", "--num_classes", "4", "--in_chns", "1",
                                    "--dense_head", "1", "--dense_teacher", "1", "--graphs", "0", "--func", "smc"])
args.patch_size = [256, 256]
st = T.ArcoStep2D(args, "cuda:0")
for m in (st.model, st.ema_model):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
out = []
for it in range(n):
    l, ll = T.synthetic_batch(8, args.patch_size, 4, 1 + 2 * it, "cuda:0")
    u, _ = T.synthetic_batch(8, args.patch_size, 4, 2 + 2 * it, "cuda:0")
    random.seed(10 + it); np.random.seed(10 + it); torch.manual_seed(10 + it)
    st.step(l, ll, u, 0, 100)
    out.append({k: round(float(v), 6) for k, v in st.last_terms.items()})
    print(it, out[-1], flush=True)
print("stats", ops.sparse_bwd_stats, "wsum", float(sum(p.double().abs().sum() for p in st.q_representation.parameters())))
