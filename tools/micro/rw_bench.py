"""conv3x3_rw_kernel<8,1> / <4,2> launch classes of the 2-D step, microseconds per launch with a 256 MB tensor touched between launches
(inputs from HBM): the `roofline` kernel of bench.py in isolation.  python tools/micro/rw_bench.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import ops, _lib as L
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
big = torch.randn(64, 1024, 1024, device="cuda")
for nb, ci, co, s in [(16, 16, 16, 256), (16, 32, 16, 256), (16, 16, 4, 256), (16, 4, 16, 256), (8, 16, 16, 256), (16, 32, 32, 128), (16, 32, 64, 128)]:
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wp = ops.pack_weight(w, 9, 0)
    xs = [torch.randn(nb, s, s, ci, device="cuda").permute(0, 3, 1, 2) for _ in range(4)]
    cfg = L.query("arco_conv_config_mma", 9, nb, s, s, ci, co, ci, 3)
    ts = []
    for r in range(reps):
        x = xs[r % 4]
        big.add_(1.0)
        xr, ld = ops.rows_view(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv_raw(xr, ld, ci, wp, co, nb, s, s, 9, stats=True, stat_groups=2 if nb == 16 else 1)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    byt = 4.0 * nb * s * s * (ci + co)
    print(f"{ci:3d}->{co:3d} @{s}^2 x{nb} cfg {cfg}: median {ts[len(ts)//2]:7.1f} us  min {ts[0]:7.1f}  {byt / ts[len(ts)//2] / 1e6:5.2f} TB/s algorithmic", flush=True)
