"""Which python lines launch the remaining aten kernels of the default 2-D step?  Runs the step un-graphed under
torch.profiler with stacks and prints, per aten op that launched a GPU kernel, the innermost arco_amd frame."""
import os, sys, collections
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from arco_amd import train_arco_2d as T

args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--func", "smc", "--synthetic", "1",
                                    "--graphs", "0", "--graph_train", "0"])
st = T.ArcoStep2D(args, "cuda:0")
bs = [(T.synthetic_batch(8, args.patch_size, 4, 100 + 2 * i, "cuda:0"), T.synthetic_batch(8, args.patch_size, 4, 101 + 2 * i, "cuda:0")[0]) for i in range(4)]
for i in range(6):
    (l, ll), u = bs[i % 4]
    st.step(l, ll, u, i, 100)
torch.cuda.synchronize()
NSTEP = 4
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
agg = collections.defaultdict(int)
WATCH = ("copy_", "fill_", "zero_", "add", "add_", "cat", "mul", "mul_", "clone", "contiguous", "zeros", "zeros_like", "sum", "index", "_to_copy")


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        out = func(*args, **(kwargs or {}))
        if name in WATCH:
            t = out if isinstance(out, torch.Tensor) else (args[0] if args and isinstance(args[0], torch.Tensor) else None)
            if t is not None and t.is_cuda:
                frames = [f for f in traceback.extract_stack() if "arco_amd" in f.filename]
                fr = frames[-1] if frames else None
                where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}" if fr else "?"
                if fr and len(frames) > 1:
                    f2 = frames[-2]
                    where += f" <- {os.path.basename(f2.filename)}:{f2.lineno}"
                agg[(name, where, tuple(t.shape))] += 1
        return out


with Log():
    for i in range(NSTEP):
        (l, ll), u = bs[i % 4]
        st.step(l, ll, u, 6 + i, 100)
torch.cuda.synchronize()
for (name, where, shape), n in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"{n / NSTEP:6.1f}/step  {name:12s} {str(shape):28s} {where}")
