"""3-D step (LA V-Net 112x112x80, 4 volumes): is the host on the critical path?  Steady-state ms/step, host time of the sampler
stage, and the step's sensitivity to an extra host delay in contrast_draw / at the start of the step."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_3d as T3, _contrast as C_
args = T3.build_parser().parse_args(["--batch_size", "2", "--queue_size", "4096", "--synthetic", "1", "--num_classes", "2"])
args.patch_size = [112, 112, 80]
st = T3.ArcoStep3D(args, "cuda:0")
l, ll = T3.synthetic_volume_batch(2, args.patch_size, 2, 1, "cuda:0")
u, _ = T3.synthetic_volume_batch(2, args.patch_size, 2, 2, "cuda:0")
marks = {}
delay = {"draw": 0.0, "start": 0.0}
def wrap(mod, name, key=None):
    f = getattr(mod, name)
    def w(*a, **k):
        if key and delay[key] > 0:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < delay[key]: pass
        t0 = time.perf_counter(); r = f(*a, **k); marks.setdefault(name, []).append((time.perf_counter() - t0) * 1e3); return r
    setattr(mod, name, w)
wrap(C_, "contrast_counts"); wrap(C_, "contrast_draw", "draw"); wrap(C_, "contrast_masks", "start")
def run(n):
    for _ in range(n):
        st.step(l, ll, u, 0, 100)
run(30)
torch.cuda.synchronize()
for k, v in marks.items():
    print(f"{k:18s}", " ".join(f"{t:.2f}" for t in v[-8:]))
for rep in range(2):
    for key, d in (("draw", 0.0), ("draw", 0.002), ("start", 0.002)):
        delay["draw"] = delay["start"] = 0.0
        delay[key] = d
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(20); torch.cuda.synchronize()
        print(f"delay {d * 1e3:.0f} ms at {key:5s}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step")
pl = None
orig_draw = C_.contrast_draw
def spy(p, *a, **k):
    global pl
    pl = p
    return orig_draw(p, *a, **k)
C_.contrast_draw = spy
run(1)
print("n_anchor", [int(x) for x in pl.n_anchor], "bank_len", [int(x) for x in pl.bank_len], "n_neg", [int(x) for x in pl.n_neg], "Q", pl.Q, "Nn", pl.Nn, "iter", st.iter_num)
