"""Effective HBM rate of the BatchNorm apply / backward passes at the full-resolution and second levels of the 3-D steps
(LA: 112x112x80 x 2 volumes fp32; LiTS: 160x160x96 x 2 volumes f16 storage).  Bytes = every operand once."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import ops


def timeit(fn, reps=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


big = torch.randn(64, 1024, 1024, device="cuda")
for name, nv, sp, c, dt in (("LA L1 f32", 2, (112, 112, 80), 16, torch.float32), ("LA L2 f32", 2, (56, 56, 40), 32, torch.float32),
                            ("LiTS L1 f16", 2, (160, 160, 96), 16, torch.float16), ("LiTS L2 f16", 2, (80, 80, 48), 32, torch.float16),
                            ("LiTS L1 f16 x1", 1, (160, 160, 96), 16, torch.float16)):
    z = ops.new_act_nd(nv, c, sp, "cuda", dt); z.copy_(torch.randn_like(z))
    da = ops.new_act_nd(nv, c, sp, "cuda", dt); da.copy_(torch.randn_like(da))
    g, b = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.1
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    el = z.numel() * z.element_size()
    zz = z.clone().requires_grad_(True)
    def fwd():
        return ops.bn_act(zz, g, b, rm, rv, slope=0.0)
    y = fwd()
    t_f = timeit(lambda: fwd())
    def bwd():
        y = fwd()
        y.backward(da)
    t_fb = timeit(bwd)
    with torch.no_grad():
        t_ng = timeit(lambda: ops.bn_act(z, g, b, rm, rv, slope=0.0))
    print(f"{name:16s} tensor {el / 1e6:7.1f} MB: stats+finalize+apply (grad mode) {t_f:7.1f} us, no_grad {t_ng:7.1f} us, fwd+bwd {t_fb:7.1f} us  "
          f"-> apply-only lower bound 2 x tensor / 4 TB/s = {2 * el / 4e6:6.1f} us", flush=True)
