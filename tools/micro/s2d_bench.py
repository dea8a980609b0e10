"""space-to-depth / depth-to-space passes of the V-Net levels: us and GB/s (read + write once)."""
import sys
sys.path.insert(0, ".")
import torch
from arco_amd import ops


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, nv, sp0 in (("la", 4, (112, 112, 80)), ("lits", 2, (160, 160, 96))):
    for c, div in ((16, 1), (32, 2), (64, 4), (128, 8)):
        sp = tuple(s // div for s in sp0)
        for dt in (torch.float32, torch.float16):
            x = torch.randn((nv, *sp, c), device="cuda").to(dt).movedim(-1, 1)
            y = ops.space_to_depth3(x)
            f = timed(lambda: ops.space_to_depth3(x))
            b = timed(lambda: ops.depth_to_space3(y))
            by = 2 * x.numel() * x.element_size()
            print(f"{name} {c:4d}ch {sp} {str(dt)[6:]}: s2d {f:7.1f} us {by / f / 1e3:7.1f} GB/s   d2s {b:7.1f} us {by / b / 1e3:7.1f} GB/s")
