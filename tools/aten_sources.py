"""Which Python call sites of arco_amd still launch ATen kernels (fill / add / copy / cat / index ...) in a step?
A TorchDispatchMode over 2 eager steps (graphs off so that every op is dispatched) records every aten op on GPU tensors
with the innermost arco_amd frame of the Python stack (ops issued by the autograd engine show the backward() call site)."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from arco_amd import train_arco_2d as T

SKIP = ("aten::view", "aten::as_strided", "aten::permute", "aten::detach", "aten::slice.", "aten::select", "aten::expand", "aten::t.",
        "aten::transpose", "aten::unsqueeze", "aten::squeeze", "aten::alias", "aten::_unsafe_view", "aten::reshape", "aten::movedim",
        "aten::empty", "aten::_local_scalar_dense", "aten::is_pinned", "aten::_pin_memory", "aten::lift_fresh", "aten::record_stream",
        "aten::set_", "aten::resize_", "aten::is_same_size", "aten::sym_", "aten::stride", "aten::size", "aten::unbind", "aten::split")
cnt = collections.Counter()
byt = collections.Counter()      # bytes of the op's output(s): which of these launches are large

class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._schema.name + "." + func._overloadname
        if not name.startswith(SKIP):
            t = [a for a in list(args) + [out] if isinstance(a, torch.Tensor)]
            if any(x.is_cuda for x in t):
                site = "?"
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if "arco_amd/" in fr.filename:
                        site = f"{fr.filename.split('arco_amd/')[-1]}:{fr.lineno} {fr.name}"
                        break
                cnt[(name, site)] += 1
                outs = out if isinstance(out, (tuple, list)) else [out]
                byt[(name, site)] += sum(o.numel() * o.element_size() for o in outs if isinstance(o, torch.Tensor))
        return out

k2 = os.environ.get("K2", "1")
args = T.build_parser().parse_args(["--batch_size", "8", "--queue_size", "4096", "--synthetic", "1", "--graphs", "0", "--k2", k2])
st = T.ArcoStep2D(args, "cuda:0")
l, ll = T.synthetic_batch(8, args.patch_size, 4, 1, "cuda:0")
u, _ = T.synthetic_batch(8, args.patch_size, 4, 2, "cuda:0")
for _ in range(3):
    st.step(l, ll, u)
torch.cuda.synchronize()
N = 2
with Rec():
    for _ in range(N):
        st.step(l, ll, u)
torch.cuda.synchronize()
for (name, site), n in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"{n / N:6.1f} calls/step  {byt[(name, site)] / N / 1e6:9.2f} MB/step  {name:34s} {site}")
print("total", sum(cnt.values()) / N)
