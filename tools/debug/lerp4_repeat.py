"""arco_lerp4_cat_rows_bwd on fixed inputs, repeated: is the output reproducible?  (the head backward's first non-reproducible tensor)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arco_amd import _lib as L
torch.manual_seed(0)
dev = "cuda:0"
for (n, Clo, Chi, hw) in ((1024, 480, 16, 256), (4096, 448, 32, 128)):
    dX = (torch.randn(n, Clo + Chi, device=dev) * 1e-6)
    lylx = torch.rand(2 * n, device=dev)
    lylx[::7] = 0.0
    pix = torch.randint(0, 16 * hw * hw, (n,), device=dev)
    ref = None
    nbad = 0
    side = torch.cuda.Stream()
    for it in range(3000):
        dV = torch.empty((4 * n, Clo), device=dev)
        if it % 3 == 0:
            dV.fill_(float("nan"))
        dhi = torch.zeros((16 * hw * hw, Chi), device=dev)
        L.call("arco_lerp4_cat_rows_bwd", L.ptr(dX), Clo + Chi, Clo, L.ptr(lylx), L.ptr(pix), n, L.ptr(dV), Clo, L.ptr(dhi), Chi, Chi)
        if it % 5 == 0:          # some unrelated traffic on another stream
            with torch.cuda.stream(side):
                junk = torch.zeros(8 << 20, device=dev)
        if ref is None:
            ref = dV.clone()
        elif not torch.equal(ref, dV):
            nbad += 1
            if nbad <= 3:
                bad = (ref != dV)
                print(f"  it {it}: {int(bad.sum())} elements differ, nan {int(torch.isnan(dV).sum())}, zeros among bad {int((dV[bad] == 0).sum())}")
    print(f"n={n} Clo={Clo} Chi={Chi}: {nbad} of 3000 repeats differ")
