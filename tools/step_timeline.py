"""GPU timeline of the default 2-D step without a profiler: HIP events recorded around the step's major pieces on the stream each
runs on (main / side), printed as start-end ms relative to the step's first kernel, mean over the last steps."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_2d as T, _contrast as C_, glue, ops, augment
args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--func", "smc", "--synthetic", "1"] + sys.argv[1:])
st = T.ArcoStep2D(args, "cuda:0")
bs = [(T.synthetic_batch(8, args.patch_size, 4, 100 + 2 * i, "cuda:0"), T.synthetic_batch(8, args.patch_size, 4, 101 + 2 * i, "cuda:0")[0]) for i in range(4)]
log = []          # (name, ev0, ev1, host0, host1)
on = [False]
def E():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
def wrap_obj(obj, name, label=None):
    f = getattr(obj, name)
    def w(*a, **k):
        if not on[0]:
            return f(*a, **k)
        h0 = time.perf_counter(); e0 = E(); r = f(*a, **k); e1 = E()
        log.append((label or name, e0, e1, h0, time.perf_counter())); return r
    if hasattr(f, "__dict__") and not isinstance(f, type(wrap_obj)) and not hasattr(f, "__func__"):
        w.__dict__.update({k: v for k, v in f.__dict__.items()})
        for attr in ("will_replay", "captured", "calls"):
            if hasattr(f, attr):
                pass
    setattr(obj, name, w)
    return f
class Proxy:
    """Callable stand-in that forwards attributes (will_replay, captured ...) to the wrapped pass."""
    def __init__(self, inner, label): self.__dict__["_i"] = inner; self.__dict__["_l"] = label
    def __getattr__(self, k): return getattr(self._i, k)
    def __setattr__(self, k, v): setattr(self._i, k, v)
    def __call__(self, *a, **k):
        if not on[0]:
            return self._i(*a, **k)
        h0 = time.perf_counter(); e0 = E(); r = self._i(*a, **k); e1 = E()
        log.append((self._l, e0, e1, h0, time.perf_counter())); return r
for n in ("t_fwd_u0", "t_fwd_lu", "s_train_lu", "s_fwd_stats", "s_train_tps"):
    setattr(st, n, Proxy(getattr(st, n), n))
for n in ("contrast_masks", "contrast_lists_protos", "contrast_counts", "contrast_enqueue", "contrast_draw", "contrast_anchor_pix", "contrast_infonce"):
    wrap_obj(C_, n)
wrap_obj(glue, "eqv_loss"); wrap_obj(glue, "supervised_loss"); wrap_obj(glue, "compute_unsupervised_loss")
wrap_obj(augment, "generate_unsup_data"); wrap_obj(augment, "batch_transform")
wrap_obj(st.optimizer, "step", "optimizer.step"); wrap_obj(st.optimizer, "merge_second")
wrap_obj(st.isd, "_momentum_update_key_encoder", "ema")
bw = torch.Tensor.backward
def backward(self, *a, **k):
    if not on[0]:
        return bw(self, *a, **k)
    h0 = time.perf_counter(); e0 = E(); r = bw(self, *a, **k); e1 = E()
    log.append(("backward(main)", e0, e1, h0, time.perf_counter()))
    if st._t_stream is not None:
        with torch.cuda.stream(st._t_stream):
            log.append(("backward(side) end", E(), E(), h0, h0))
    return r
torch.Tensor.backward = backward
def run(n, collect):
    out = []
    for i in range(n):
        (l, ll), u = bs[i % 4]
        log.clear(); on[0] = collect
        h0 = time.perf_counter()
        st.step(l, ll, u, 0, 100)
        on[0] = False
        if collect:
            out.append((h0, list(log), time.perf_counter()))
    return out
run(60, False)
torch.cuda.synchronize(); t0 = time.perf_counter(); run(60, False); torch.cuda.synchronize()
print(f"uninstrumented {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms/step")
torch.cuda.synchronize(); t0 = time.perf_counter(); recs = run(40, True); torch.cuda.synchronize()
print(f"instrumented   {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step")
recs = recs[8:]
agg = {}
order = []
for h0, lg, h1 in recs:
    base = lg[0][1]
    seen = {}
    for name, e0, e1, a, b in lg:
        seen[name] = seen.get(name, 0) + 1
        key = f"{name}#{seen[name]}" if seen[name] > 1 or name == "batch_transform" else name
        if key not in agg:
            agg[key] = []; order.append(key)
        agg[key].append((base.elapsed_time(e0), base.elapsed_time(e1), (a - h0) * 1e3, (b - h0) * 1e3))
print(f"{'piece':28s} {'GPU start':>9s} {'GPU end':>9s} {'dur':>7s} | {'host in':>8s} {'host out':>8s}   (ms after the step's first event / step entry)")
for k in order:
    v = agg[k]; n = len(v)
    m = [sum(x[i] for x in v) / n for i in range(4)]
    print(f"{k:28s} {m[0]:9.3f} {m[1]:9.3f} {m[1] - m[0]:7.3f} | {m[2]:8.3f} {m[3]:8.3f}")
