"""3x3x3 forward kernel per V-Net level (LA patch, nv volumes): igemm_kernel<9,..,FLAT,DEPTH=3> against conv3d_fl_kernel (HIP events,
20 reps), with the bit-identity check.  python tools/micro/fl_bench.py [nv]      (ARCO_CONV3D_FL_CFG=<A_T><C_T> forces a tile shape)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import ops, _lib as L

ops.CONV_MMA = 3
HALF = bool(int(os.environ.get("HALF", "0")))      # f16 activation storage: hconv_kernel against hconv_fc_kernel


def timeit(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


nv = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if HALF:
    SHAPES0 = ((32, 32, (80, 80, 48)), (64, 64, (40, 40, 24)), (128, 128, (20, 20, 12)), (256, 256, (10, 10, 6)), (32, 32, (56, 56, 40)), (64, 64, (28, 28, 20)))
SHAPES = SHAPES0 if HALF else ((32, 32, (56, 56, 40)), (64, 64, (28, 28, 20)), (128, 128, (14, 14, 10)), (256, 256, (7, 7, 5)), (32, 32, (80, 80, 48)), (64, 64, (40, 40, 24)))
if os.environ.get('FL_SHAPES'):
    SHAPES = SHAPES[:int(os.environ['FL_SHAPES'])]
for ci, co, sp in SHAPES:
    d3, h, w = sp
    x = torch.randn(nv, d3, h, w, ci, device="cuda").permute(0, 4, 1, 2, 3)
    if HALF:
        x = x.half()
    wt = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
    wp = ops.pack_weight(wt, 27, 0, half=HALF)
    xr, ld = ops.rows_view(x)
    fl = 2.0 * nv * d3 * h * w * ci * co * 27
    out = {}
    line = f"3x3x3 nv={nv} {ci:4d}->{co:4d} @{sp}:"
    for on in (0, 1):
        ops.conv3d_fl_set(on)
        cfg = L.query("arco_conv_config_mma", 27, nv * d3, h, w, ci, co, ld, 4 if HALF else 3)
        f = lambda: ops.conv_raw(xr, ld, ci, wp, co, nv, h, w, 27, stats=True, d3=d3, half=HALF)
        out[on] = f()[0].clone()
        us = timeit(f)
        line += f"  [{cfg}] {us:7.1f} us {fl / us / 1e6:6.1f} TF"
    ops.conv3d_fl_set(1)
    print(line, " identical" if torch.equal(out[0], out[1]) else "  DIFFERENT", flush=True)
