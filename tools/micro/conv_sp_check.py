"""A/B of the software-pipelined split-bf16 3x3 kernel (conv_sp.hip) against igemm_kernel<9,..,MMA=3> on the same inputs:
max |difference| of outputs and BN partial sums, and the launch times of both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import _lib as L
if os.environ.get("ARCO_LIB"):
    L.LIB_PATH = os.environ["ARCO_LIB"]
from arco_amd import ops

def timeit(fn, reps=40):
    for _ in range(8): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

torch.manual_seed(0)
shapes = [(16, 64, 64, 64), (16, 32, 32, 128), (16, 128, 64, 64), (16, 32, 64, 64), (16, 64, 32, 128), (16, 64, 128, 64), (8, 64, 64, 64),
          (16, 16, 32, 128), (16, 16, 32, 256), (16, 32, 32, 256), (16, 128, 128, 32), (3, 64, 64, 64), (16, 32, 64, 128),
          (16, 16, 16, 256), (16, 32, 16, 256), (8, 16, 16, 256), (16, 32, 16, 128), (8, 32, 32, 128), (16, 256, 128, 32), (16, 128, 256, 32),
          (16, 256, 256, 16), (8, 128, 128, 32), (8, 256, 256, 16), (16, 128, 256, 16), (16, 256, 128, 16), (8, 64, 64, 64), (8, 128, 64, 64),
          (16, 64, 128, 32), (8, 64, 128, 32)]
if len(sys.argv) > 1:
    shapes = shapes[int(sys.argv[2]) if len(sys.argv) > 2 else 0:int(sys.argv[1])]
bad = 0
for nb, ci, co, s in shapes:
    x = torch.randn(nb, s, s, ci, device="cuda").permute(0, 3, 1, 2)
    wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    bias = torch.randn(co, device="cuda")
    wp = ops.pack_weight(wt, 9, 0)
    xr, ldx = ops.rows_view(x)
    res = {}
    for on in (0, 1):
        ops.conv_sp_set(on)
        ops._cfg_cache.clear()
        cfg = L.query("arco_conv_config_mma", 9, nb, s, s, ci, co, ldx, 3)
        out, (ssum, ssq, nmb) = ops.conv_raw(xr, ldx, ci, wp, co, nb, s, s, 9, bias=bias, stats=True)
        torch.cuda.synchronize()
        t = timeit(lambda: ops.conv_raw(xr, ldx, ci, wp, co, nb, s, s, 9, bias=bias, stats=True))
        res[on] = (out.clone(), ssum.sum(1).clone(), ssq.sum(1).clone(), t, cfg)
    ops.conv_sp_set(1)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    o0, o1 = res[0][0], res[1][0]
    scale = float(ref.abs().max())
    e0 = float((o0.double() - ref).abs().max()) / scale; e1 = float((o1.double() - ref).abs().max()) / scale
    d = float((o0 - o1).abs().max()) / scale
    ds = float((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max()); dq = float((res[0][2] - res[1][2]).abs().max() / res[0][2].abs().max())
    fl = 2.0 * nb * s * s * ci * co * 9
    ok = e1 < 3e-6 and d < 3e-6 and ds < 1e-4 and dq < 1e-4
    bad += not ok
    print(f"{'OK ' if ok else 'BAD'} nb={nb} {ci}->{co}@{s}: cfg {res[0][4]} -> {res[1][4]}  err64 old {e0:.2e} sp {e1:.2e}  |old-sp| {d:.2e}  stats {ds:.1e} {dq:.1e}"
          f"  time old {res[0][3]:.1f} us ({fl / res[0][3] / 1e6:.0f} TF)  sp {res[1][3]:.1f} us ({fl / res[1][3] / 1e6:.0f} TF)")
print("FAILED" if bad else "ALL OK")
