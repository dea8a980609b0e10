import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import _lib as L, ops, samplers as S

def t(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6

a = torch.randn(8, 256, 256, 16, device="cuda").permute(0, 3, 1, 2)
b = torch.randn(8, 256, 256, 16, device="cuda").permute(0, 3, 1, 2)
print("cat dim1 cl  us", t(lambda: torch.cat([a, b], 1)))
print("cat dim0 cl  us", t(lambda: torch.cat([a, b], 0)))
ac, bc = a.contiguous(), b.contiguous()
print("cat dim1 nchw us", t(lambda: torch.cat([ac, bc], 1)))
print("empty us", t(lambda: torch.empty((8, 256, 256, 16), device="cuda")))
print("current_stream us", t(lambda: torch.cuda.current_stream().cuda_stream))
x = torch.zeros(1, dtype=torch.int64, device="cuda")
def inc():
    global x
    x += 1
print("tensor += 1 us", t(inc))
y = torch.randn(1000, device="cuda"); z = torch.empty(1000, device="cuda")
print("L.call ema us", t(lambda: L.call("arco_ema", L.ptr(y), L.ptr(z), 1000, 0.9)))
w = torch.randn(16, 16, 3, 3, device="cuda")
print("pack_weight us", t(lambda: ops.pack_weight(w, 9, 0)))
ev = lambda: torch.cuda.Event(enable_timing=True).record()
print("event record us", t(ev))
torch.manual_seed(0)
t0 = time.perf_counter()
for _ in range(5): S.grid_monte_carlo_sample(4096, 131072)
print("sampler neg ms", (time.perf_counter() - t0) / 5 * 1e3)
t0 = time.perf_counter()
for _ in range(5): S.grid_monte_carlo_sample(260000, 256)
print("sampler anchor ms", (time.perf_counter() - t0) / 5 * 1e3)
t0 = time.perf_counter()
for _ in range(20): torch.empty(131072, dtype=torch.int64).pin_memory()
print("pin ms", (time.perf_counter() - t0) / 20 * 1e3)
print("threads", torch.get_num_threads(), os.cpu_count())
