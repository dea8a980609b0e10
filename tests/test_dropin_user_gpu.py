"""The advertised boundary driven the way a reference-style trainer drives it: tests/dropin_user.py -
a user of `dropin/` written in this repository's own form, with torch's own nn.Conv2d q_representation and torch.optim.SGD - against
tests/golden/g19_trainer_loop.npz: the reference's loop body executed from the reference's text over the reference's modules on CPU
(oracle/gen_golden.py g19, build container only).  Two chained iterations: every loss term to 1e-3, bank lengths / pointers bit-exact,
bank contents, the state of the three host generators after each iteration, updated weights (-m gpu)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case", ["a", "b"])          # a: k2 = 1, cutmix;  b: k2 = 0, cutout (-1 labels through one-hot / masks / CE)
def test_dropin_user_matches_the_reference_loop(case):
    g = np.load(os.path.join(ROOT, "tests", "golden", "g19_trainer_loop.npz"))
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_user.py"), case], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    got = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("DROPIN_USER ")][-1][len("DROPIN_USER "):])
    assert got["model_file"].startswith(os.path.join(ROOT, "dropin")) and got["arco_modules"]            # the drop-in was bound
    for it, s in enumerate(got["steps"]):
        # Iteration 0 starts from the fixture state: everything is held, incl. the three host generators (bit-exact sampler replay).
        # Iteration 1 starts from weights the GPU and the CPU updated in fp32 in different summation orders (1e-5 apart): a pixel
        # within rounding of a pseudo-label / entropy threshold may flip, which moves a key between banks, with it a sampler
        # argument and every later draw of the CPU generator - incl. the TPS warp of the equivariance term (measured on one box:
        # one key of 317 moved, loss_eqv 5 % off, the other terms 1e-5 .. 4e-4).  The well-conditioned terms stay at 1e-3; the
        # generator state, the bank bookkeeping and the warp-dependent term are compared exactly when no decision flipped.
        same_decisions = s["bank_len"] == g[f"{case}_{it}_bank_len"].tolist() \
            and np.array_equal(np.asarray(s["probe"], dtype=np.float64), np.asarray(g[f"{case}_{it}_probe"], dtype=np.float64))
        assert it > 0 or same_decisions
        for k in ("loss_ce", "loss_dice", "unsup_loss", "reco_loss", "loss_q") + (("loss_eqv", "loss") if same_decisions else ()):
            np.testing.assert_allclose(s[k], float(g[f"{case}_{it}_{k}"]), rtol=1e-3, atol=1e-5, err_msg=f"step {it} {k}")     # north_star: 1e-3
        assert max(abs(a - b) for a, b in zip(s["bank_len"], g[f"{case}_{it}_bank_len"].tolist())) <= 2, (s["bank_len"], it)
        assert s["pool_ptr"] == int(g[f"{case}_{it}_pool_ptr"]) and s["banks_on_gpu"]
        if same_decisions:
            assert s["ptr"] == g[f"{case}_{it}_ptr"].tolist(), it
            np.testing.assert_allclose(s["bank_sum"], g[f"{case}_{it}_bank_sum"], rtol=2e-4)
            np.testing.assert_allclose(s["probe"], g[f"{case}_{it}_probe"], rtol=0, atol=0)
        else:
            np.testing.assert_allclose(s["loss_eqv"], float(g[f"{case}_{it}_loss_eqv"]), rtol=0.25)       # another warp of the same batch
    for k, v in got["end"].items():
        np.testing.assert_allclose(v, float(g[f"{case}_end_{k}"]), rtol=1e-3, err_msg=k)


def test_dropin_user_3d_matches_the_reference_volume_loop():
    """tests/dropin_user3d.py (torch nn.Conv3d q_representation, torch.optim.SGD, CPU banks with the reference's randn first row) against
    g19 'v': train_arco_3d.py's loop body run from the reference's text - three iterations, C = 4: the 5-D banks fill and truncate,
    iteration 0 optimises the equivariance objective, later ones the contrastive one."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "g19_trainer_loop.npz"))
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_user3d.py")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    got = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("DROPIN_USER ")][-1][len("DROPIN_USER "):])
    assert got["model_file"].startswith(os.path.join(ROOT, "dropin")) and got["arco_modules"]
    for it, s in enumerate(got["steps"]):
        # the generator probes (python / numpy / torch draws after the step) say whether every data-dependent draw count agreed
        same_decisions = s["bank_len"] == g[f"v_{it}_bank_len"].tolist() and s["ptr"] == g[f"v_{it}_ptr"].tolist() \
            and np.array_equal(np.asarray(s["probe"], dtype=np.float64), np.asarray(g[f"v_{it}_probe"], dtype=np.float64))
        assert it > 0 or same_decisions, (s["bank_len"], g[f"v_{it}_bank_len"].tolist(), s["probe"], g[f"v_{it}_probe"].tolist())
        for k in ("loss_ce", "loss_dice", "unsup_loss", "reco_loss", "loss_q") + (("loss_eqv", "loss") if same_decisions else ()):
            # (iterations >= 1 start from V-Net weights updated by two implementations: fp32 V-Net gradients differ by 0.3-1.8 % per
            #  parameter - DESIGN.md section 2 - and the contrastive term, a mean over 48 x 16 sampled pairs, sees it first: measured
            #  1.1e-3 .. 1.7e-3,
            #  the warp-consistency term of iteration 2 6.9e-3 - both held to 1e-2 there; iteration 0, from the fixture state, is held
            #  to 1e-3 on every term, the supervised terms to 1e-3 throughout)
            loose = it > 0 and k in ("reco_loss", "loss_eqv", "loss")
            np.testing.assert_allclose(s[k], float(g[f"v_{it}_{k}"]), rtol=1e-2 if loose else 1e-3, atol=1e-5,
                                       err_msg=f"step {it} {k}")
        assert max(abs(a - b) for a, b in zip(s["bank_len"], g[f"v_{it}_bank_len"].tolist())) <= 3 and s["banks_on_gpu"]
        if same_decisions:
            # (the keys appended at iterations >= 1 come from a teacher that has followed two differently rounded students: 6e-3 measured)
            np.testing.assert_allclose(s["bank_sum"], g[f"v_{it}_bank_sum"], rtol=2e-4 if it == 0 else 2e-2)
    assert max(got["steps"][-1]["bank_len"]) == 200                      # a bank met the truncation
    for k, v in got["end"].items():
        np.testing.assert_allclose(v, float(g[f"v_end_{k}"]), rtol=2e-3, err_msg=k)


def test_star_imported_nn_builds_q_representation_on_the_hip_path():
    """The reference trainers get `nn` from their star imports and build q_representation with it (train_arco_2d.py:231-234,
    train_arco_3d.py:206-209).  dropin/ exports arco_amd.nn_dropin.nn there: torch.nn with 1x1 Conv2d / Conv3d through ops.conv.
    Same module surface; values and gradients equal torch's own convolution up to fp32 summation order; anything that is not a plain
    1x1 convolution of an fp32 GPU tensor takes torch's forward."""
    code = r'''
import sys, json
sys.path.insert(0, sys.argv[1])
import torch
from model_2D import *
from loss_helper_3d import *
tnn = torch.nn
out = {}
out["cls"] = [nn.Conv2d.__name__, nn.Conv2d.__module__, issubclass(nn.Conv2d, tnn.Conv2d), nn.Sequential is tnn.Sequential,
              nn.functional is tnn.functional, nn.KLDivLoss is tnn.KLDivLoss]
torch.manual_seed(0)
q = nn.Sequential(nn.Conv2d(496, 496, kernel_size=1, bias=False), nn.Conv2d(496, 496, kernel_size=1, bias=False)).cuda()
r = tnn.Sequential(tnn.Conv2d(496, 496, kernel_size=1, bias=False), tnn.Conv2d(496, 496, kernel_size=1, bias=False)).cuda()
r.load_state_dict(q.state_dict())                       # same keys
out["keys"] = sorted(q.state_dict().keys())
x = torch.randn(2, 496, 96, 80, device="cuda")
dy = torch.zeros(2, 496, 96, 80, device="cuda")
dy[:, :, ::7, ::9] = torch.randn(2, 496, 14, 9, device="cuda")          # few rows: the backward takes the row route or the dense one
res = []
for m in (q, r):
    xx = x.clone().requires_grad_(True)
    y = m(xx)
    y.backward(dy)
    res.append((y.detach(), xx.grad, [p.grad for p in m.parameters()]))
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
out["fwd"] = rel(res[0][0], res[1][0]); out["dx"] = rel(res[0][1], res[1][1])
out["dw"] = max(rel(a, b) for a, b in zip(res[0][2], res[1][2]))
# not a plain 1x1 convolution: torch's own forward, bit for bit
c3 = nn.Conv2d(16, 16, 3, padding=1).cuda(); t3 = tnn.Conv2d(16, 16, 3, padding=1).cuda(); t3.load_state_dict(c3.state_dict())
xi = torch.randn(1, 16, 20, 20, device="cuda")
out["c3_same"] = bool(torch.equal(c3(xi), t3(xi)))
c1 = nn.Conv2d(32, 32, 1); out["cpu_same"] = bool(torch.equal(c1(torch.ones(1, 32, 4, 4)), tnn.functional.conv2d(torch.ones(1, 32, 4, 4), c1.weight, c1.bias)))
# ragged sizes, a bias, NCHW-contiguous input, no_grad, an input that needs no gradient
cb = nn.Conv2d(48, 32, kernel_size=1).cuda(); tb = tnn.Conv2d(48, 32, 1).cuda(); tb.load_state_dict(cb.state_dict())
xo = torch.randn(3, 48, 7, 5, device="cuda")
with torch.no_grad():
    out["odd_nograd"] = rel(cb(xo), tb(xo))
yb, yt = cb(xo), tb(xo)
go = torch.randn_like(yt)
yb.backward(go); yt.backward(go)
out["odd"] = max(rel(yb.detach(), yt.detach()), rel(cb.weight.grad, tb.weight.grad), rel(cb.bias.grad, tb.bias.grad))
out["odd_shape"] = [list(yb.shape), list(yt.shape)]
v = nn.Sequential(nn.Conv3d(16, 16, kernel_size=1, bias=False)).cuda(); tv = tnn.Conv3d(16, 16, 1, bias=False).cuda(); tv.load_state_dict(v[0].state_dict())
xv = torch.randn(1, 16, 12, 10, 8, device="cuda")
out["v"] = rel(v(xv), tv(xv))
print("NNDROP " + json.dumps(out))
'''
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "dropin")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("NNDROP ")][-1][len("NNDROP "):])
    assert out["cls"] == ["Conv2d", "arco_amd.nn_dropin", True, True, True, True] and out["keys"] == ["0.weight", "1.weight"]
    assert out["fwd"] <= 2e-5 and out["dx"] <= 2e-5 and out["dw"] <= 1e-4 and out["v"] <= 1e-5, out
    assert out["c3_same"] and out["cpu_same"]
    assert out["odd_nograd"] <= 2e-5 and out["odd"] <= 1e-4 and out["odd_shape"][0] == out["odd_shape"][1], out


def test_literal_reference_statements_on_the_dropin_nn():
    """The statements the reference trainers apply to q_representation and its output, literally: the module built from the
    star-imported `nn` (train_arco_2d.py:231-234), wrapped in nn.DataParallel (:240), its output sliced and flattened with `.view`
    (:127, 129, 400 - a RuntimeError on a plain channels-last tensor), normalised, and differentiated.  The accelerated 1x1 path must
    be the one that ran, and the flattening order must be torch's (c, h, w)."""
    import torch.nn.functional as F
    from arco_amd import ops
    from arco_amd.nn_dropin import nn
    torch.manual_seed(0)
    q_representation = nn.Sequential(nn.Conv2d(496, 496, kernel_size=1, bias=False), nn.Conv2d(496, 496, kernel_size=1, bias=False))
    q_representation = torch.nn.DataParallel(q_representation.cuda())
    x = torch.randn(4, 496, 32, 32, device="cuda")
    rep_all = q_representation(x)
    assert isinstance(rep_all, ops.BoundaryTensor) and rep_all.stride(1) == 1          # the HIP GEMM path, channels-last memory
    rep_u = rep_all[2:]
    rep_u = rep_u.view(rep_u.shape[0], -1)                                            # train_arco_2d.py:127
    rep_u = torch.nn.functional.normalize(rep_u, dim=-1)                              # :128
    with torch.no_grad():
        w0, w1 = [m.weight for m in q_representation.module]
        ref = F.conv2d(F.conv2d(x, w0), w1)[2:]
        ref = F.normalize(ref.reshape(ref.shape[0], -1), dim=-1)
    assert rep_u.shape == ref.shape
    np.testing.assert_allclose(rep_u.detach().cpu().numpy(), ref.cpu().numpy(), rtol=1e-3, atol=1e-6)
    random_pool = F.normalize(torch.randn(6, rep_u.shape[1], device="cuda"), dim=1)
    dist_t = 2 - 2 * torch.einsum('bc,kc->bk', [rep_u, random_pool])                  # :130
    dist_t.sum().backward()
    g = [m.weight.grad for m in q_representation.module]
    assert all(t is not None and torch.isfinite(t).all() and float(t.abs().max()) > 0 for t in g)
    # 3-D twin (train_arco_3d.py:206-209, 124)
    q3 = nn.Sequential(nn.Conv3d(16, 16, 1), nn.Conv3d(16, 16, 1)).cuda()
    y3 = q3(torch.randn(2, 16, 8, 16, 16, device="cuda"))
    assert isinstance(y3, ops.BoundaryTensor)
    assert y3[1:].view(1, -1).shape == (1, 16 * 8 * 16 * 16)
