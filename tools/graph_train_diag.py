"""Bisect HIP-graph capture of forward+backward per op (each case in a child process: a crash is a result)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = ["cmp_unet", "k_wgrad1", "k_wgrad9", "k_colsum", "k_pack1", "k_dgrad", "k_convfn_nograd_capture", "convbn", "conv1x1", "maxpool", "bilinear", "drop", "block", "unet"]

def run(case):
    import torch
    from arco_amd import graphs, ops
    from arco_amd.networks import unetWithArgs as U
    dev = "cuda:0"
    torch.manual_seed(0)
    if case == "cmp_unet":
        m = U.UNet(1, 4).to(dev).train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
        gt = graphs.GraphedTrain(m, warmup=0)
        for it in range(3):
            x = torch.rand(2, 1, 64, 64, device=dev)
            bufs = {k: v.clone() for k, v in m.state_dict().items()}
            pe, be, fe = m(x)
            ge = torch.autograd.grad([pe.sum() + sum(f.sum() for f in fe)], [p for p in m.parameters()], allow_unused=True)
            pe, fe = pe.detach().clone(), [f.detach().clone() for f in fe]
            del be
            m.load_state_dict(bufs)           # undo the running-stat update
            pg, bg, fg = gt(x)
            for p_ in m.parameters(): p_.grad = None
            (pg.sum() + sum(f.sum() for f in fg)).backward()
            torch.cuda.synchronize()
            print(it, "pred maxdiff", float((pg.detach() - pe).abs().max()), "fm", [float((a.detach() - b).abs().max()) for a, b in zip(fg, fe)])
            d = [float((p_.grad - g).abs().max() / (g.abs().max() + 1e-12)) for p_, g in zip(m.parameters(), ge) if g is not None and p_.grad is not None]
            print(it, "grad rel maxdiff", max(d), "n", len(d))
            del pg, bg, fg
        print(case, "OK"); return
    if case.startswith("k_"):
        x = torch.rand(2, 32, 32, 16, device=dev).permute(0, 3, 1, 2)
        dy = torch.rand(2, 32, 32, 16, device=dev).permute(0, 3, 1, 2)
        w1 = torch.rand(16, 16, 1, 1, device=dev); w9 = torch.rand(16, 16, 3, 3, device=dev)
        xr, ldx = ops.rows_view(x); dr, ldd = ops.rows_view(dy)
        def body():
            if case == "k_wgrad1": return ops.conv_wgrad(dr, ldd, 16, xr, ldx, 16, 1, 2, 32, 32, w1)
            if case == "k_wgrad9": return ops.conv_wgrad(dr, ldd, 16, xr, ldx, 16, 9, 2, 32, 32, w9)
            if case == "k_colsum": return ops.colsum(dr, ldd, 2048, 16)
            if case == "k_pack1": return ops.pack_weight(w9, 9, 1)
            if case == "k_dgrad": return ops.conv_raw(dr, ldd, 16, ops.pack_weight(w9, 9, 1), 16, 2, 32, 32, 9)[0]
            if case == "k_convfn_nograd_capture":
                xx = x.detach().requires_grad_(True); ww = w9.clone().requires_grad_(True)
                y = ops.conv(xx, ww)
                return torch.autograd.grad([y], [xx, ww], [dy])
        body(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            r = body()
        g.replay(); torch.cuda.synchronize()
        print(case, "OK"); return
    if case.startswith("torch"):
        m = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8)).to(dev)
        x = torch.rand(32, 64, device=dev)
        if case == "torch_mgc":
            f = torch.cuda.make_graphed_callables(m, (x,))
            for it in range(3):
                f(x).sum().backward()
            torch.cuda.synchronize(); print(case, "OK"); return
        gt = graphs.GraphedTrain(m, warmup=1)
        for it in range(4):
            gt(x).sum().backward(); torch.cuda.synchronize()
        print(case, "OK captured", gt.captured); return
    if case == "unet":
        m = U.UNet(1, 4).to(dev).train(); x = torch.rand(2, 1, 64, 64, device=dev)
    elif case == "block":
        m = U.ConvBlock(16, 16, 0.1).to(dev).train(); x = torch.rand(2, 16, 32, 32, device=dev)
    else:
        class M(torch.nn.Module):
            def __init__(s):
                super().__init__()
                s.c = torch.nn.Conv2d(16, 16, 3 if case == "convbn" else 1, padding=1 if case == "convbn" else 0)
                s.bn = torch.nn.BatchNorm2d(16)
            def forward(s, x):
                if case == "convbn":
                    return ops.conv_bn_act(x, s.c.weight, s.c.bias, s.bn.weight, s.bn.bias, s.bn.running_mean, s.bn.running_var)
                y = ops.conv(x, s.c.weight, s.c.bias)
                if case == "maxpool": return ops.maxpool2(y)
                if case == "bilinear": return ops.bilinear(y, (64, 64))
                if case == "drop": return ops.bn_act(y, None, None, None, None, slope=1.0, p=0.5, drop_mode=1)
                return y
        m = M().to(dev).train(); x = torch.rand(2, 16, 32, 32, device=dev)
    x = ops.to_channels_last(x) if hasattr(ops, "to_channels_last") else x
    gt = graphs.GraphedTrain(m, warmup=1)
    for it in range(4):
        out = gt(x)
        flat = [o for o in (out if isinstance(out, (tuple, list)) else [out])]
        loss = sum((o.float().sum() if torch.is_tensor(o) else sum(t.sum() for t in o)) for o in flat)
        loss.backward()
        torch.cuda.synchronize()
        lv = float(loss)
        del loss, out, flat
    print(case, "OK captured", gt.captured, lv)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True)
            tail = (r.stdout + r.stderr).strip().splitlines()
            msg = [l for l in tail if l.startswith(c) or "Error" in l or "error" in l][-3:]
            print(f"{c:10s} rc={r.returncode} {msg}")
