"""3-D twin of self_consistency.py: the default volume step executed again and again from one snapshot; for every execution whose
flat gradient differs from the first one's, WHERE it differs (parameter name, number of elements, largest difference, positions).
python tools/debug/repro3d.py [trials] [lits|la] [extra trainer flags ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import test_configs_at_size_gpu as TC
from arco_amd import train_arco_3d as T3, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
lits = (sys.argv[2] if len(sys.argv) > 2 else "lits") == "lits"
extra = sys.argv[3:]
if os.environ.get("R3_PASS_SIDE"):
    T3.PASS_SIDE = int(os.environ["R3_PASS_SIDE"])
sp, b, C = ((160, 160, 96), 1, 2) if lits else ((112, 112, 80), 2, 4)
st = TC._make3d((["--act_dtype", "f16"] if lits else []) + extra, patch=sp, b=b, n_cls=C)
TC._drop_off(st)
def batch(i):
    l, ll = T3.synthetic_volume_batch(b, sp, C, 10 + i, "cuda:0")
    u, _ = T3.synthetic_volume_batch(b, sp, C, 20 + i, "cuda:0")
    return l, ll, u
for it in range(4):
    TC.seed_all(800 + it)
    st.step(*batch(it))
torch.cuda.synchronize()
snap = TC._snapshot(st)
bt = batch(4)
# flat-gradient layout: name every slice
names = []
opt = st.optimizer
off = 0
named = {}
for mod_name, m in (("model", st.model), ("fe", getattr(st, "q_feature_extractor", None)), ("q", getattr(st, "q_representation", None))):
    if m is None:
        continue
    for k, p in m.named_parameters():
        named[p.data_ptr()] = (mod_name + "." + k, p.numel())
layout = []
for g in opt.param_groups:
    for p in g["params"]:
        layout.append((named.get(p.data_ptr(), ("?", p.numel()))[0], p.numel(), tuple(p.shape)))
offs = []
o = 0
base = opt.flat_p.data_ptr()
for g in opt.param_groups:
    for p in g["params"]:
        offs.append(((p.data_ptr() - base) // 4, p.numel(), named.get(p.data_ptr(), ("?",))[0], tuple(p.shape)))
offs.sort()
ref = None
bad = 0
for trial in range(n):
    TC._restore(st, snap)
    TC.seed_all(804)
    st.step(*bt)
    torch.cuda.synchronize()
    g = st.optimizer.flat_g
    if ref is None:
        ref = g.clone()
        print("ref max", float(ref.abs().max()), "numel", ref.numel(), flush=True)
        continue
    d = (g - ref).abs()
    if float(d.max()) == 0.0:
        continue
    bad += 1
    print(f"trial {trial}: max diff {float(d.max()):.3e} rel {float(d.max()) / float(ref.abs().max()):.3e} n_diff {int((d > 0).sum())}", flush=True)
    for (o, ne, nm, shp) in offs:
        dd = d[o:o + ne]
        k = int((dd > 0).sum())
        if k:
            idx = torch.nonzero(dd > 0).flatten()[:6].tolist()
            print(f"   {nm} {shp}: {k}/{ne} differ, max {float(dd.max()):.3e} (|ref| max {float(ref[o:o + ne].abs().max()):.3e}) first idx {idx}", flush=True)
print(f"{bad} of {n - 1} executions differ from the first", flush=True)
