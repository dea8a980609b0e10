import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_2d as T, _lib as L
slow = collections.Counter(); slow_t = collections.defaultdict(float)
def wrap_fn(obj, name, label=None):
    f = getattr(obj, name); label = label or name
    def w(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); dt = (time.perf_counter() - t0) * 1e3
        if dt > 3.0:
            key = label if label != "L.call" else "L.call:" + a[0]
            slow[key] += 1; slow_t[key] += dt
        return r
    setattr(obj, name, w)
wrap_fn(L, "call", "L.call")
for n in ("empty", "zeros", "cat", "empty_like", "zeros_like"):
    wrap_fn(torch, n, "torch." + n)
wrap_fn(torch.Tensor, "pin_memory", "pin_memory")
wrap_fn(torch.Tensor, "to", "Tensor.to")
wrap_fn(torch.Tensor, "copy_", "copy_")
wrap_fn(torch.cuda.Event, "synchronize", "Event.synchronize")
wrap_fn(torch.cuda.Event, "record", "Event.record")
wrap_fn(torch.cuda.CUDAGraph, "replay", "graph.replay")
args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--synthetic", "1", "--graphs", sys.argv[1] if len(sys.argv) > 1 else "1"])
st = T.ArcoStep2D(args, "cuda:0")
l, ll = T.synthetic_batch(8, args.patch_size, 4, 1, "cuda:0")
u, _ = T.synthetic_batch(8, args.patch_size, 4, 2, "cuda:0")
for i in range(6): st.step(l, ll, u)
slow.clear(); slow_t.clear()
ts = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter(); st.step(l, ll, u); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("steps:", " ".join(f"{t:.0f}" for t in ts))
for k, v in slow.most_common(20): print(f"{k:40s} n={v:4d} total={slow_t[k]:8.1f} ms")
