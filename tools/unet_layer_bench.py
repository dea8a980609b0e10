"""Per-layer timing of the U-Net's 3x3 convolutions at the benchmark's launch shapes (nb images per launch):
forward (+BN partial statistics), data gradient and weight gradient, with the MFMA (157.3 TFLOP/s) and HBM (6.3 TB/s
achievable) floors of each launch.  `python tools/unet_layer_bench.py [nb]`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import _lib as L, ops

def timeit(fn, reps=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3     # us

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
LAYERS = [(16, 16, 256), (16, 32, 128), (32, 32, 128), (32, 64, 64), (64, 64, 64), (64, 128, 32), (128, 128, 32), (128, 256, 16), (256, 256, 16),
          (256, 128, 32), (128, 64, 64), (64, 32, 128), (32, 16, 256)]
COUNT = {(16, 16, 256): 3, (32, 32, 128): 3, (64, 64, 64): 3, (128, 128, 32): 3}      # encoder conv2 + decoder conv2 (+ conv2 of ...)
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
floor = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
print(f"nb={nb}   (us: measured | mfma floor | hbm floor)")
for ci, co, s in LAYERS:
    x = torch.randn(nb, s, s, ci, device="cuda").permute(0, 3, 1, 2)
    dz = torch.randn(nb, s, s, co, device="cuda").permute(0, 3, 1, 2)
    wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wp, wd = ops.pack_weight(wt, 9, 0), ops.pack_weight(wt, 9, 1)
    xr, ldx = ops.rows_view(x); dr, ldz = ops.rows_view(dz)
    fl = 2.0 * nb * s * s * ci * co * 9
    by = 4.0 * nb * s * s * (ci + co)
    mf, hb = fl / 157.3e12 * 1e6, by / 6.3e12 * 1e6
    t_f = timeit(lambda: ops.conv_raw(xr, ldx, ci, wp, co, nb, s, s, 9, stats=True))
    t_d = timeit(lambda: ops.conv_raw(dr, ldz, co, wd, ci, nb, s, s, 9))
    t_w = timeit(lambda: ops.conv_wgrad(dr, ldz, co, xr, ldx, ci, 9, nb, s, s, wt))
    n = 2 if ci == co else 1
    for k, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)):
        tot[k] += n * t; floor[k] += n * max(mf, hb)
    print(f"{ci:4d}->{co:4d} @{s:3d}^2 x{n}: fwd {t_f:7.1f} ({fl/t_f/1e6:6.1f} TF)  dgrad {t_d:7.1f} ({fl/t_d/1e6:6.1f} TF)  wgrad {t_w:7.1f} ({fl/t_w/1e6:6.1f} TF) | mfma {mf:6.1f} hbm {hb:6.1f}")
print("per pass (us): " + "  ".join(f"{k} {tot[k]:.0f} (floor {floor[k]:.0f})" for k in tot))
