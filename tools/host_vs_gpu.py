"""Host enqueue time of a step against its steady-state wall time (default 2-D step): if the host needs as long to queue a step
as the GPU to run it, the step is host-bound.  Also the host time of each stretch between the step's host synchronisations."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_2d as T, _contrast as C_
args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--func", "smc", "--synthetic", "1"] + sys.argv[1:])
st = T.ArcoStep2D(args, "cuda:0")
bs = [(T.synthetic_batch(8, args.patch_size, 4, 100 + 2 * i, "cuda:0"), T.synthetic_batch(8, args.patch_size, 4, 101 + 2 * i, "cuda:0")[0]) for i in range(4)]
marks = {}
def wrap(mod, name):
    f = getattr(mod, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); marks.setdefault(name, []).append((t0, time.perf_counter())); return r
    setattr(mod, name, w)
for n in ("contrast_masks", "contrast_lists_protos", "contrast_counts", "contrast_enqueue", "contrast_draw", "contrast_anchor_pix", "contrast_infonce"):
    wrap(C_, n)
def run(n, rec=None):
    for i in range(n):
        (l, ll), u = bs[i % 4]
        t0 = time.perf_counter()
        st.step(l, ll, u, 0, 100)
        if rec is not None:
            rec.append((t0, time.perf_counter()))
run(60)
torch.cuda.synchronize()
for k in marks: marks[k].clear()
rec = []
t0 = time.perf_counter(); run(100, rec); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 100 * 1e3
host = sum(b - a for a, b in rec) / len(rec) * 1e3
print(f"wall {wall:.3f} ms/step, host time inside step() {host:.3f} ms")
names = list(marks)
for k in names:
    v = marks[k]
    print(f"  {k:22s} host {sum(b - a for a, b in v) / len(v) * 1e3:6.3f} ms, starts {sum(a - r[0] for (a, b), r in zip(v, rec)) / len(v) * 1e3:6.3f} ms after step entry")
print(f"  step() returns {sum(b - a for a, b in rec) / len(rec) * 1e3:6.3f} ms after entry")
