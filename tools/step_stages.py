import sys, os, time, gc
os.environ.setdefault('OMP_NUM_THREADS', '4'); os.environ.setdefault('MKL_NUM_THREADS', '4')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_2d as T, _contrast as C_
args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--synthetic", "1", "--graphs", "1", "--graph_train", os.environ.get("GT", "0")])
st = T.ArcoStep2D(args, "cuda:0")
l, ll = T.synthetic_batch(8, args.patch_size, 4, 1, "cuda:0")
u, _ = T.synthetic_batch(8, args.patch_size, 4, 2, "cuda:0")
marks = {}
def wrap(mod, name):
    f = getattr(mod, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); marks.setdefault(name, []).append((time.perf_counter() - t0) * 1e3); return r
    setattr(mod, name, w)
for n in ("contrast_masks", "contrast_sample", "contrast_enqueue", "contrast_infonce"):
    wrap(C_, n)
import arco_amd.head as H
wrap(H, "lazy_head")
_es = torch.cuda.Event.synchronize
def es(self):
    t0 = time.perf_counter(); r = _es(self); marks.setdefault("event_sync", []).append((time.perf_counter() - t0) * 1e3); return r
torch.cuda.Event.synchronize = es
orig_bw = torch.Tensor.backward
def bw(self, *a, **k):
    t0 = time.perf_counter(); r = orig_bw(self, *a, **k); marks.setdefault("backward", []).append((time.perf_counter() - t0) * 1e3); return r
torch.Tensor.backward = bw
gc_t = []
def cb(phase, info):
    if phase == "start": cb.t = time.perf_counter()
    else: gc_t.append(((time.perf_counter() - cb.t) * 1e3, info["generation"]))
gc.callbacks.append(cb)
tot = []
for i in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.step(l, ll, u)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    tot.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
print("host/wall:", " ".join(f"{a:.0f}/{b:.0f}" for a, b in tot))
for k, v in marks.items():
    print(f"{k:18s}", " ".join(f"{x:.1f}" for x in v[-12:]))
print("gc:", [(round(a, 1), g) for a, g in gc_t if a > 2][:40])
