"""Host time of the two halves of the sampler stage in the product step: contrast_counts (waits for the counters) and
contrast_draw (sampler replay + index upload) - the GPU has only the sample-independent loss forwards queued meanwhile."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_2d as T, _contrast as C_
args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--func", "smc", "--synthetic", "1"])
st = T.ArcoStep2D(args, "cuda:0")
bs = [(T.synthetic_batch(8, args.patch_size, 4, 100 + 2 * i, "cuda:0"), T.synthetic_batch(8, args.patch_size, 4, 101 + 2 * i, "cuda:0")[0]) for i in range(4)]
marks = {}
def wrap(mod, name):
    f = getattr(mod, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); marks.setdefault(name, []).append((time.perf_counter() - t0) * 1e3); return r
    setattr(mod, name, w)
for n in ("contrast_counts", "contrast_draw", "contrast_enqueue", "contrast_masks"):
    wrap(C_, n)
for i in range(40):
    (l, ll), u = bs[i % 4]
    st.step(l, ll, u, 0, 100)
torch.cuda.synchronize()
for k, v in marks.items():
    print(f"{k:18s}", " ".join(f"{t:.2f}" for t in v[-12:]))
pl = None
orig_draw = C_.contrast_draw
def spy(p, *a, **k):
    global pl
    pl = p
    return orig_draw(p, *a, **k)
C_.contrast_draw = spy
for i in range(3):
    (l, ll), u = bs[i % 4]
    st.step(l, ll, u, 0, 100)
    print("n_anchor", [int(x) for x in pl.n_anchor], "bank_len", [int(x) for x in pl.bank_len], "valid", pl.valid_classes, "Q", pl.Q, "Nn", pl.Nn)
