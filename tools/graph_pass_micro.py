"""Student U-Net forward+backward (batch 8, 256x256): eager vs graphs.GraphedTrain, wall per pass and kernel count."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arco_amd import graphs, ops, optim
from arco_amd.networks import unetWithArgs as U
torch.manual_seed(0)
m = U.UNet(1, 4).cuda().train()
opt = optim.SGDNesterov(list(m.parameters()), lr=0.01)
plan = ops.PackPlan([m], True); plan.refresh()
x = torch.rand(8, 1, 256, 256, device="cuda")
def run(fn, n):
    for _ in range(n):
        out, _, fm = fn(x)
        opt.zero_grad()
        loss = out.sum() + sum(f.sum() for f in fm)
        loss.backward()
        del out, fm, loss
gt = graphs.GraphedTrain(m, warmup=2)
for name, fn in (("eager", m), ("graphed", gt), ("eager", m), ("graphed", gt)):
    run(fn, 5); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(fn, 30); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{name:8s} {1e3 * (t1 - t0) / 30:.3f} ms per fwd+bwd pass")
