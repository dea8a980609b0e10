"""Is the host-side sampler replay (contrast_draw) on the step's critical path?  Steady-state ms/step with an extra host delay
of DELAY_MS inside contrast_draw (0 and 1): if the step grows by the delay, every ms saved there is a ms off the step."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import train_arco_2d as T, _contrast as C_
args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--func", "smc", "--synthetic", "1"])
st = T.ArcoStep2D(args, "cuda:0")
bs = [(T.synthetic_batch(8, args.patch_size, 4, 100 + 2 * i, "cuda:0"), T.synthetic_batch(8, args.patch_size, 4, 101 + 2 * i, "cuda:0")[0]) for i in range(4)]
delay = [0.0]
orig = C_.contrast_draw
def slow(*a, **k):
    if delay[0] > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < delay[0]: pass
    return orig(*a, **k)
C_.contrast_draw = slow
def run(n):
    for i in range(n):
        (l, ll), u = bs[i % 4]
        st.step(l, ll, u, 0, 100)
run(150)
for rep in range(3):
    for d in (0.0, 0.0005, 0.001, -1):
        delay[0] = max(d, 0.0)
        if d < 0:
            C_.contrast_draw = orig
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(60); torch.cuda.synchronize()
        print(f"delay {d * 1e3:4.1f} ms: {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms/step")
        C_.contrast_draw = slow
