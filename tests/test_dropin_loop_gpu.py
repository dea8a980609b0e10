"""The advertised boundary, run the way the reference runs it: tools/dropin_loop.py executes the
loop body of the reference's train_arco_2d.py:284-435 over the names its own import statements bind through `dropin/` - with a
TORCH `nn.Sequential(nn.Conv2d, nn.Conv2d)` as q_representation, `torch.optim.SGD`, the reference's `param_k.data = ...` EMA
statement and CPU banks, i.e. none of arco_amd's trainer-internal machinery (flat buffers, PackPlans, graphs, row-sparse head).
Two chained steps against the CPU oracle step: every loss term to 1e-3, bank lengths / pointers / contents, updated weights (-m gpu)."""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

import cpu_step
import fixture_inputs as fx

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("k2,aug", [(1.0, "cutmix"), (0.0, "none")])
def test_reference_style_loop_over_the_dropin_modules_matches_the_cpu_oracle(k2, aug):
    env = dict(os.environ, K2=str(k2), APPLY_AUG=aug)
    env.pop("PYTHONPATH", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dropin_loop.py")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("DROPIN_LOOP ")][-1]
    got = json.loads(line[len("DROPIN_LOOP "):])
    assert got["tail"]["model_file"].startswith(os.path.join(ROOT, "dropin")) and got["tail"]["arco_modules"]      # the drop-in was bound
    # ---- the CPU oracle on the same weights, data and generator seeds
    C, b, patch, Q, Nn, qs = 4, 2, (64, 64), 64, 32, 300
    unet_sd, fe_sd = fx.unet_state(21, 1, C), fx.fe_state(31)
    qrep_w = [fx.fe_state(32)["fea4.weight"], fx.fe_state(33)["fea4.weight"]]
    st = cpu_step.make_state(unet_sd, fe_sd, qrep_w)
    bank, ptr, qsz = fx.fresh_bank(C, 496, qs, 'zeros')
    rs = np.random.RandomState(3)
    for it in range(2):
        l = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32))
        u = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32))
        lab = torch.from_numpy(fx.blob_labels(rs, b, patch, C))
        random.seed(10 + it); np.random.seed(10 + it); torch.manual_seed(10 + it)
        cpu_step.step(st, l, lab, u, bank, ptr, qsz, C, k1=1.0, lr=0.01, nq=Q, nn_=Nn, k2=k2, apply_aug=aug)
        g, o = got["steps"][it], st["last_terms"]
        for k in ("ce", "dice", "unsup", "reco") + (("eqv",) if k2 else ()):
            np.testing.assert_allclose(g[k], o[k], rtol=1e-3, atol=1e-5, err_msg=f"step {it} {k}")          # north_star: 1e-3
        assert g["bank_len"] == [int(x[0].shape[0]) for x in bank] and g["ptr"] == [int(p) for p in ptr], it
        np.testing.assert_allclose(g["bank_sum"], [float(x[0].double().abs().sum()) for x in bank], rtol=1e-4)
    t = got["tail"]
    ref = dict(w_first=st["student"]["encoder.in_conv.conv_conv.0.weight"], w_last=st["student"]["decoder.out_conv.weight"],
               w_deep=st["student"]["encoder.down4.maxpool_conv.1.conv_conv.4.weight"], qrep0=st["q_rep"][0], qrep1=st["q_rep"][1],
               qfe4=st["q_fe"]["fea4.weight"], kfe4=st["k_fe"]["fea4.weight"], t_first=st["teacher"]["encoder.in_conv.conv_conv.0.weight"],
               rm=st["student"]["encoder.in_conv.conv_conv.1.running_mean"])
    for k, v in ref.items():
        np.testing.assert_allclose(t[k], float(v.detach().double().abs().sum()), rtol=2e-4, err_msg=k)


def test_reference_style_3d_loop_over_the_dropin_modules_runs_and_matches_the_oracle_where_it_can():
    """tools/dropin_loop3d.py: the loop body of train_arco_3d.py:255-400 over `dropin/` (torch `nn.Conv3d` q_representation,
    torch.optim.SGD, C = 4 so that the 5-D banks fill).  Step 0's supervised terms do not depend on any sampled index or threshold
    decision: they are compared with the CPU oracle of the 3-D step at 1e-3; the rest of the run (three steps, iteration 0 with the
    equivariance objective) must stay finite, fill the banks and move the weights - the strict whole-step comparison of the 3-D step,
    with forced decisions, is tests/test_step3d_parity_gpu.py."""
    import cpu_step3d
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dropin_loop3d.py")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    got = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("DROPIN_LOOP3D ")][-1][len("DROPIN_LOOP3D "):])
    assert got["tail"]["model_file"].startswith(os.path.join(ROOT, "dropin"))
    for s_ in got["steps"]:
        assert all(np.isfinite(s_[k]) for k in ("ce", "dice", "unsup", "reco", "eqv", "loss")) and s_["banks_on_gpu"]
    assert got["tail"]["finite"] and got["tail"]["qrep_finite"] and got["tail"]["moved"] > 0
    assert max(got["steps"][-1]["bank_len"]) > 16                       # C = 4: the 5-D enqueue ran
    C, b, patch = 4, 2, (32, 32, 32)
    torch.manual_seed(3)
    st = cpu_step3d.make_state(fx.vnet_state(52, 1, C), fx.fe_state(61, (128, 64, 32, 16, 16), 16, nd=3),
                               [torch.randn(16, 16, 1, 1, 1) / 4, torch.randn(16, 16, 1, 1, 1) / 4])
    bank, ptr, qsz = fx.fresh_bank(C, 16, 200, 'randn')
    rs = np.random.RandomState(13)
    l = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32))
    u = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32))
    lab = torch.from_numpy(fx.blob_labels(rs, b, patch, C))
    l = l + 0.5 * (lab > 0).unsqueeze(1).float()
    random.seed(10); np.random.seed(10); torch.manual_seed(10)
    cpu_step3d.step(st, l, lab, u, bank, ptr, qsz, n_cls=C, k1=1.0, nq=48, nn_=16, strong_threshold=0.3)
    for k in ("ce", "dice"):
        np.testing.assert_allclose(got["steps"][0][k], st["last_terms"][k], rtol=1e-3, err_msg=k)
