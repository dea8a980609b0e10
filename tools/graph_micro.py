import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import graphs
from arco_amd.networks.unetWithArgs import UNet
net = UNet(1, 4).cuda().train()
x = torch.rand(8, 1, 256, 256, device="cuda")
def wall(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
with torch.no_grad():
    print("eager  host/wall ms", wall(lambda: net(x)))
    g = graphs.GraphedForward(net, warmup=1)
    g(x); g(x); g(x)
    print("graph  host/wall ms", wall(lambda: g(x)))
