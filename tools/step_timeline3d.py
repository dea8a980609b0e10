"""GPU timeline of the 3-D step without a profiler (see tools/step_timeline.py).  python tools/step_timeline3d.py [lits]"""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_3d as T3, _contrast as C_, glue, ops, augment
lits = "lits" in sys.argv
b = 1 if lits else 2
args = T3.build_parser().parse_args(["--batch_size", str(b), "--queue_size", "4096", "--synthetic", "1", "--num_classes", "2",
                                     "--conv_mma", "f32x3", "--act_dtype", "f16" if lits else "f32"])
args.patch_size = [160, 160, 96] if lits else [112, 112, 80]
st = T3.ArcoStep3D(args, "cuda:0")
l, ll = T3.synthetic_volume_batch(b, args.patch_size, 2, 1, "cuda:0")
u, _ = T3.synthetic_volume_batch(b, args.patch_size, 2, 2, "cuda:0")
log, on = [], [False]
def E():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
def wrap_obj(obj, name, label=None):
    f = getattr(obj, name)
    def w(*a, **k):
        if not on[0]:
            return f(*a, **k)
        h0 = time.perf_counter(); e0 = E(); r = f(*a, **k); e1 = E()
        log.append((label or name, e0, e1, h0, time.perf_counter())); return r
    setattr(obj, name, w)
class Proxy:
    def __init__(self, inner, label): self.__dict__["_i"] = inner; self.__dict__["_l"] = label
    def __getattr__(self, k): return getattr(self._i, k)
    def __setattr__(self, k, v): setattr(self._i, k, v)
    def __call__(self, *a, **k):
        if not on[0]:
            return self._i(*a, **k)
        h0 = time.perf_counter(); e0 = E(); r = self._i(*a, **k); e1 = E()
        log.append((self._l, e0, e1, h0, time.perf_counter())); return r
for n in ("t_fwd_u0", "t_fwd_lu", "s_train_lu", "s_fwd_tps"):
    setattr(st, n, Proxy(getattr(st, n), n))
for n in ("contrast_masks", "contrast_lists_protos", "contrast_counts", "contrast_enqueue", "contrast_draw", "contrast_anchor_pix", "contrast_infonce"):
    wrap_obj(C_, n)
wrap_obj(glue, "eqv_loss"); wrap_obj(glue, "supervised_loss"); wrap_obj(glue, "compute_unsupervised_loss"); wrap_obj(glue, "entropy_masks")
wrap_obj(augment, "generate_unsup_data_3d")
wrap_obj(st.optimizer, "step", "optimizer.step")
wrap_obj(st.isd, "_momentum_update_key_encoder", "ema")
bw = torch.Tensor.backward
def backward(self, *a, **k):
    if not on[0]:
        return bw(self, *a, **k)
    h0 = time.perf_counter(); e0 = E(); r = bw(self, *a, **k); e1 = E()
    log.append(("backward(main)", e0, e1, h0, time.perf_counter()))
    return r
torch.Tensor.backward = backward
def run(n, collect):
    out = []
    for i in range(n):
        log.clear(); on[0] = collect
        h0 = time.perf_counter()
        st.step(l, ll, u)
        on[0] = False
        if collect:
            out.append((h0, list(log), time.perf_counter()))
    return out
run(20, False)
torch.cuda.synchronize(); t0 = time.perf_counter(); run(20, False); torch.cuda.synchronize()
print(f"uninstrumented {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step")
torch.cuda.synchronize(); t0 = time.perf_counter(); recs = run(16, True); torch.cuda.synchronize()
print(f"instrumented   {(time.perf_counter() - t0) / 16 * 1e3:.3f} ms/step")
recs = recs[4:]
agg, order = {}, []
for h0, lg, h1 in recs:
    base = lg[0][1]
    for name, e0, e1, a, b_ in lg:
        if name not in agg:
            agg[name] = []; order.append(name)
        agg[name].append((base.elapsed_time(e0), base.elapsed_time(e1), (a - h0) * 1e3, (b_ - h0) * 1e3))
print(f"{'piece':28s} {'GPU start':>9s} {'GPU end':>9s} {'dur':>7s} | {'host in':>8s} {'host out':>8s}")
for k in order:
    v = agg[k]; n = len(v)
    m = [sum(x[i] for x in v) / n for i in range(4)]
    print(f"{k:28s} {m[0]:9.3f} {m[1]:9.3f} {m[1] - m[0]:7.3f} | {m[2]:8.3f} {m[3]:8.3f}")
