"""Per-layer timing of the f16-storage kernels (csrc/conv_h.hip) against the fp32-storage kernels on the V-Net's shapes.
usage: python tools/half_layer_bench.py [lits|la]   - prints one line per layer: forward / data gradient / weight gradient in us,
TFLOP/s and GB/s (algorithmic bytes: input + output of the layer at its storage type)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from arco_amd import ops  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "lits"
    sp0 = (160, 160, 96) if cfg == "lits" else (112, 112, 80)
    nv = 2 if cfg == "lits" else 4
    dev = "cuda:0"
    rs = np.random.RandomState(0)
    levels = [(16, 1), (32, 2), (64, 4), (128, 8), (256, 16)]
    print(f"# {cfg}: {nv} volumes of {sp0}")
    for c, div in levels:
        sp = tuple(s // div for s in sp0)
        vox = nv * sp[0] * sp[1] * sp[2]
        w = torch.from_numpy((rs.standard_normal((c, c, 3, 3, 3)) / np.sqrt(27 * c)).astype(np.float32)).to(dev).requires_grad_(True)
        for half in (True, False):
            dt = torch.float16 if half else torch.float32
            x = torch.randn((nv, *sp, c), device=dev).to(dt).movedim(-1, 1).requires_grad_(True)
            dy = torch.randn((nv, *sp, c), device=dev).to(dt).movedim(-1, 1)
            xr, ld = ops.rows_view(x.detach())
            dyr, _ = ops.rows_view(dy)
            wp = ops.pack_weight(w, 27, 0, half=half)
            wd = ops.pack_weight(w, 27, 1, half=half)
            f = timed(lambda: ops.conv_raw(xr, ld, c, wp, c, nv, sp[1], sp[2], 27, d3=sp[0], sp=sp, stats=True, half=half))
            d = timed(lambda: ops.conv_raw(dyr, ld, c, wd, c, nv, sp[1], sp[2], 27, d3=sp[0], sp=sp, grad=True, half=half))
            g = timed(lambda: ops.conv_wgrad(dyr, ld, c, xr, ld, c, 27, nv, sp[1], sp[2], w, d3=sp[0]))
            fl = 2.0 * 27 * c * c * vox
            by = vox * c * 2 * (2 if half else 4)
            print(f"3x3x3 {c:4d}ch {sp} {'f16' if half else 'f32'}: fwd {f:8.1f} us {fl / f / 1e6:7.1f} TF {by / f / 1e3:7.1f} GB/s | "
                  f"dgrad {d:8.1f} us {fl / d / 1e6:7.1f} TF | wgrad {g:8.1f} us {fl / g / 1e6:7.1f} TF")
        # BatchNorm + ReLU apply / backward on this level's tensor
        for half in (True, False):
            dt = torch.float16 if half else torch.float32
            z = torch.randn((nv, *sp, c), device=dev).to(dt).movedim(-1, 1).requires_grad_(True)
            da = torch.randn((nv, *sp, c), device=dev).to(dt).movedim(-1, 1)
            gm = torch.ones(c, device=dev, requires_grad=True)
            bt = torch.zeros(c, device=dev, requires_grad=True)
            rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
            a = ops.bn_act(z, gm, bt, rm, rv)
            f = timed(lambda: ops.bn_act(z, gm, bt, rm, rv))

            def bwd():
                a.backward(da, retain_graph=True)
            b = timed(bwd)
            by = vox * c * (2 if half else 4)
            print(f"  bn+relu {c:4d}ch {'f16' if half else 'f32'}: stats+apply {f:8.1f} us ({3 * by / f / 1e3:7.1f} GB/s)  backward {b:8.1f} us ({5 * by / b / 1e3:7.1f} GB/s)")
    # the GEMM-form layers: down conv (8C -> 2C after space-to-depth), up conv (2C -> 8C before depth-to-space)
    for c, div in levels[:-1]:
        sp = tuple(s // (2 * div) for s in sp0)
        vox = nv * sp[0] * sp[1] * sp[2]
        for (ci, co) in ((8 * c, 2 * c), (2 * c, 8 * c)):
            w = torch.from_numpy((rs.standard_normal((co, ci, 1, 1, 1)) / np.sqrt(ci)).astype(np.float32)).to(dev).requires_grad_(True)
            for half in (True, False):
                dt = torch.float16 if half else torch.float32
                x = torch.randn((nv, *sp, ci), device=dev).to(dt).movedim(-1, 1)
                dy = torch.randn((nv, *sp, co), device=dev).to(dt).movedim(-1, 1)
                xr, ldx = ops.rows_view(x)
                dyr, ldy = ops.rows_view(dy)
                wp = ops.pack_weight(w, 1, 0, half=half)
                wd = ops.pack_weight(w, 1, 1, half=half)
                f = timed(lambda: ops.conv_raw(xr, ldx, ci, wp, co, nv, sp[1], sp[2], 1, d3=sp[0], sp=sp, half=half))
                d = timed(lambda: ops.conv_raw(dyr, ldy, co, wd, ci, nv, sp[1], sp[2], 1, d3=sp[0], sp=sp, grad=True, half=half))
                g = timed(lambda: ops.conv_wgrad(dyr, ldy, co, xr, ldx, ci, 1, nv, sp[1], sp[2], w, d3=sp[0]))
                fl = 2.0 * ci * co * vox
                by = vox * (ci + co) * (2 if half else 4)
                print(f"1x1x1 {ci:4d}->{co:4d} {sp} {'f16' if half else 'f32'}: fwd {f:8.1f} us {fl / f / 1e6:7.1f} TF {by / f / 1e3:7.1f} GB/s | "
                      f"dgrad {d:8.1f} us | wgrad {g:8.1f} us {fl / g / 1e6:7.1f} TF")


if __name__ == "__main__":
    main()
