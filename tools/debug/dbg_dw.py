"""conv3d_dw_kernel against igemm_kernel on one small 3x3x3 shape: where do the outputs differ (plane, row, column, channel)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import ops, _lib as L
ops.CONV_MMA = 3
nv, ci, co, (d3, h, w) = 1, 32, 32, (6, 16, 10)
if len(sys.argv) > 5:
    nv, ci, co, d3, h, w = [int(v) for v in sys.argv[1:7]]
g = torch.Generator().manual_seed(1)
x = torch.randn(nv, d3, h, w, ci, generator=g).cuda().permute(0, 4, 1, 2, 3)
wt = (torch.randn(co, ci, 3, 3, 3, generator=g) / (5.2 * ci ** 0.5)).cuda()
wp = ops.pack_weight(wt, 27, 0)
xr, ld = ops.rows_view(x)
out = {}
for on in (0, 1):
    ops.conv3d_fl_set(on)
    print("cfg", L.query("arco_conv_config_mma", 27, nv * d3, h, w, ci, co, ld, 3))
    out[on] = ops.conv_raw(xr, ld, ci, wp, co, nv, h, w, 27, stats=False, d3=d3)[0].clone()
ref = torch.nn.functional.conv3d(x.double(), wt.double(), None, padding=1)
for on in (0, 1):
    print("on", on, "vs fp64:", float((out[on].double() - ref).abs().max() / ref.abs().max()))
d = (out[0] - out[1]).abs()          # [nv, co, d3, h, w]
print("max diff", float(d.max()), "ref max", float(ref.abs().max()))
print("per plane   ", [round(float(v), 4) for v in d.amax((0, 1, 3, 4))])
print("per row     ", [round(float(v), 4) for v in d.amax((0, 1, 2, 4))])
print("per column  ", [round(float(v), 4) for v in d.amax((0, 1, 2, 3))])
print("per channel ", [round(float(v), 4) for v in d.amax((0, 2, 3, 4))])
if os.environ.get("TAPS"):
    for tap in range(27):
        w1 = torch.zeros_like(wt)
        w1.view(co, ci, 27)[:, :, tap] = wt.view(co, ci, 27)[:, :, tap]
        ops.bump_weight_epoch()
        wp1 = ops.pack_weight(w1, 27, 0)
        o = {}
        for on in (0, 1):
            ops.conv3d_fl_set(on)
            o[on] = ops.conv_raw(xr, ld, ci, wp1, co, nv, h, w, 27, stats=False, d3=d3)[0].clone()
        print("tap", tap, "dz", tap // 9, "t9", tap % 9, "max diff", float((o[0] - o[1]).abs().max()), "of", float(o[0].abs().max()))
if os.environ.get("TAP8"):
    for tap in (8, 17):
        w1 = torch.zeros_like(wt)
        w1.view(co, ci, 27)[:, :, tap] = wt.view(co, ci, 27)[:, :, tap]
        ops.bump_weight_epoch()
        wp1 = ops.pack_weight(w1, 27, 0)
        o = {}
        for on in (0, 1):
            ops.conv3d_fl_set(on)
            o[on] = ops.conv_raw(xr, ld, ci, wp1, co, nv, h, w, 27, stats=False, d3=d3)[0].clone()
        d = (o[0] - o[1]).abs()
        print("tap", tap)
        print(" per plane   ", [round(float(v), 3) for v in d.amax((0, 1, 3, 4))])
        print(" per row     ", [round(float(v), 3) for v in d.amax((0, 1, 2, 4))])
        print(" per column  ", [round(float(v), 3) for v in d.amax((0, 1, 2, 3))])
        print(" per channel ", [round(float(v), 3) for v in d.amax((0, 2, 3, 4))])
        # is the wrong output the contribution of a subset of input channels?
        for lo, hi in ((0, 16), (16, 32), (0, 8), (8, 16)):
            w2 = torch.zeros_like(wt); w2.view(co, ci, 27)[:, lo:hi, tap] = wt.view(co, ci, 27)[:, lo:hi, tap]
            ref2 = torch.nn.functional.conv3d(x, w2, None, padding=1)
            print("  dw == contribution of input channels", lo, hi, "?", float((o[1] - ref2).abs().max()))
if os.environ.get("TAP8K"):
    for tap in (8, 26):
        for lo, hi in ((0, 16), (16, 32), (0, 4), (4, 8), (8, 12), (12, 16)):
            for nlo, nhi in ((0, 32), (0, 16), (16, 32)):
                w1 = torch.zeros_like(wt)
                w1.view(co, ci, 27)[nlo:nhi, lo:hi, tap] = wt.view(co, ci, 27)[nlo:nhi, lo:hi, tap]
                ops.bump_weight_epoch()
                wp1 = ops.pack_weight(w1, 27, 0)
                o = {}
                for on in (0, 1):
                    ops.conv3d_fl_set(on)
                    o[on] = ops.conv_raw(xr, ld, ci, wp1, co, nv, h, w, 27, stats=False, d3=d3)[0].clone()
                d = (o[0] - o[1]).abs()
                print("tap", tap, "k", lo, hi, "n", nlo, nhi, "diff", round(float(d.max()), 4), "of", round(float(o[0].abs().max()), 4),
                      "| dw max", round(float(o[1].abs().max()), 4), " nonzero dw channels", [int(c) for c in torch.nonzero(o[1].abs().amax((0, 2, 3, 4)) > 0).flatten()][:40])
