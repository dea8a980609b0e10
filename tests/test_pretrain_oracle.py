"""Stage-1 pre-training: the oracle's restatement of ISD.forward (model_2D.py:215-305) and of the
trainer's losses / optimizer (pretrain_2D.py:235-252) against the golden vectors written by oracle/gen_golden.py g13
from the imported reference - two chained iterations, so the second one sees the updated queues, teacher, momentum
buffers and BN running statistics.  CPU only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import arco_oracle as orc          # noqa: E402
import fixture_inputs as fx        # noqa: E402

GOLD = {2: np.load(os.path.join(ROOT, "tests", "golden", "g13_pretrain.npz")),
        3: np.load(os.path.join(ROOT, "tests", "golden", "g14_pretrain3d.npz"))}
HEADS = ("q_latent_head", "k_latent_head", "latent_predictor", "q_outputs_head", "k_outputs_head", "outputs_predictor")


def oracle_state(nd=2):
    heads = fx.isd_head_state(33) if nd == 2 else fx.isd3d_head_state(33)
    net_state = fx.unet_state if nd == 2 else fx.vnet_state
    st = {"model": {k: v.clone() for k, v in net_state(31).items()}, "ema_model": {k: v.clone() for k, v in net_state(32).items()}}
    for h in HEADS:
        st[h] = {k[len(h) + 1:]: v.clone() for k, v in heads.items() if k.startswith(h + ".")}
    for k in ("queue", "queue_mask", "queue_ptr", "mask_queue_ptr"):
        st[k] = heads[k].clone()
    return st


def check_forward(G, it, out, rtol=2e-4):
    outputs, ema_output, ema_ll, ll, ema_ol, ol = (t.detach() for t in out)
    sub = (slice(None), slice(None)) + (slice(None, None, 2),) * (outputs.dim() - 2)
    np.testing.assert_allclose(outputs[sub].numpy(), G[f"s{it}_outputs"], rtol=rtol, atol=2e-5)
    np.testing.assert_allclose(ema_output[sub].numpy(), G[f"s{it}_ema_output"], rtol=rtol, atol=2e-5)
    np.testing.assert_allclose(ema_ll.numpy(), G[f"s{it}_ema_latent_logits"], rtol=rtol, atol=2e-4)
    np.testing.assert_allclose(ll.numpy(), G[f"s{it}_latent_logits"], rtol=rtol, atol=2e-4)
    for tag, t in (("ema_output_logits", ema_ol), ("output_logits", ol)):
        assert tuple(t.shape) == tuple(G[f"s{it}_{tag}_shape"])
        np.testing.assert_allclose(t[::7, ::13].numpy(), G[f"s{it}_{tag}_sample"], rtol=rtol, atol=5e-4)
        np.testing.assert_allclose(float(t.double().abs().sum()), G[f"s{it}_{tag}_sums"][1], rtol=1e-4)


import pytest          # noqa: E402


@pytest.mark.parametrize("nd", [2, 3])
def test_stage1_two_iterations_vs_reference_golden(nd):
    cfg = fx.STAGE1_CFG if nd == 2 else fx.STAGE1_CFG_3D
    G = GOLD[nd]
    st, bufs = oracle_state(nd), {}
    net = None if nd == 2 else (lambda x, sd: orc.vnet_forward(x, sd, train=True))
    for it in range(cfg["steps"]):
        im_q, im_k, lab = (fx.stage1_batch if nd == 2 else fx.stage1_batch_3d)(40, it)
        torch.manual_seed(100 + it)
        terms, out, grads = orc.isd_stage1_step(st, bufs, im_q, im_k, lab, cfg["labeled_bs"], cfg["num_classes"], cfg["lr"], cfg["Ts"],
                                                cfg["Tt"], cfg["patch_size"], cfg["output_pooling_size"], cfg["K"], net=net,
                                                sup_scale=1.0 if nd == 2 else 0.5)
        np.testing.assert_allclose([terms[k] for k in ("loss", "ce", "dice", "latent", "output")], G[f"s{it}_terms"], rtol=2e-4)
        check_forward(G, it, out)
        if it == 0:
            names = list(G["grad_names"])
            for (g, k), gr in grads.items():
                full = f"{g}.{k}"
                if gr is None:
                    assert full not in names
                    continue
                i = names.index(full)
                np.testing.assert_allclose(float(gr.double().abs().sum()), G["grad_abs"][i], rtol=2e-3, atol=2e-6 * gr.numel(), err_msg=full)   # (a conv bias in front of BatchNorm has a zero gradient up to rounding)
                if "grad::" + full in G.files and gr.numel() <= 20000:
                    np.testing.assert_allclose(gr.numpy(), G["grad::" + full], rtol=5e-3, atol=2e-6, err_msg=full)
    assert int(st["queue_ptr"]) == int(G["final::queue_ptr"][0]) == (cfg["b"] * cfg["steps"]) % cfg["K"]
    assert int(st["mask_queue_ptr"]) == int(G["final::mask_queue_ptr"][0])
    np.testing.assert_allclose(st["queue"].numpy(), G["final::queue"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(st["queue_mask"].numpy(), G["final::queue_mask"], rtol=2e-4, atol=2e-5)
    for pre in ("model", "ema_model"):
        names = [str(n) for n in G[f"final_{pre}.names"]]
        for n, ref_abs in zip(names, G[f"final_{pre}.abs"]):
            if "running_" in n and nd == 3:
                continue
            v = st[pre][n[len(pre) + 1:]]
            np.testing.assert_allclose(float(v.detach().double().abs().sum()), ref_abs, rtol=2e-4, atol=1e-6, err_msg=n)
    for k in G.files:
        if not k.startswith("final::") or k.split("::")[1].startswith(("queue", "mask_queue")):
            continue
        name = k.split("::")[1]
        grp, key = name.split(".", 1)
        v = st[grp][key].detach()
        if "running_" in key and nd == 3:      # the functional V-Net oracle keeps no running statistics
            continue
        if v.dtype == torch.long:          # num_batches_tracked: the functional oracle does not count (the product test does)
            continue
        v = v if v.numel() <= 20000 else v[::4, ::4]
        np.testing.assert_allclose(v.numpy(), G[k], rtol=5e-4, atol=2e-6, err_msg=name)
