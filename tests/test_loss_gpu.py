"""GPU parity of the HIP contrastive loss: against the golden vectors captured from the
reference, and against the CPU oracle on fresh seeded inputs.  Bit-exact: new_keys, ptr,
bank contents (pure row moves), sampled indices.  fp32 tolerance 1e-3 on loss/grad
(north_star), tested much tighter here."""
import random

import numpy as np
import pytest
import torch

import arco_oracle as orc
import fixture_inputs as fx

pytestmark = pytest.mark.gpu


def seed_all(s):
    random.seed(s); np.random.seed(s); torch.manual_seed(s)


def to_dev(inp):
    return {k: v.cuda() for k, v in inp.items()}


@pytest.mark.parametrize("case", list(fx.LOSS_CASES))
def test_loss_chain_vs_golden(golden, case):
    from arco_amd.loss_helper_3d import compute_contra_memobank_loss
    g = golden["g2_loss"]
    ikw, lkw, qsize, binit = fx.LOSS_CASES[case]
    bank, ptr, qs = fx.fresh_bank(ikw["n_cls"], ikw["feat"], qsize, binit)
    mom = torch.zeros(ikw["n_cls"], lkw["num_queries"], 1, ikw["feat"]).cuda() if case == "proto_momentum" else None
    seed_all(1337)
    for step in range(fx.LOSS_STEPS):
        inp = to_dev(fx.loss_inputs(100 * step + 11, **ikw))
        rep = inp["rep"].clone().requires_grad_(True)
        trace = {}
        res = compute_contra_memobank_loss(
            rep, inp["label_l"], inp["label_u"], inp["prob_l"], inp["prob_u"], inp["low_mask"], inp["high_mask"],
            bank, ptr, qs, inp["rep_teacher"], momentum_prototype=mom, i_iter=step + 1, _trace=trace, **lkw)
        p = f"{case}_s{step}_"
        if mom is not None:
            mom, new_keys, loss = res
            np.testing.assert_allclose(mom.cpu().numpy(), g[p + "prototype"], rtol=1e-4, atol=1e-5)
        else:
            new_keys, loss = res
        loss.backward()
        assert new_keys == g[p + "new_keys"].tolist()
        assert [int(q) for q in ptr] == g[p + "ptr"].tolist()
        assert [b[0].shape[0] for b in bank] == g[p + "bank_len"].tolist()
        for c, b in enumerate(bank):
            assert b[0].is_cuda or step == 0
            np.testing.assert_array_equal(b[0].cpu().numpy(), g[p + f"bank{c}"])      # bit-exact rows
        if lkw["func"] in ("smc", "asmc"):
            draws = []
            for a, n in zip(trace.get("anchor_idx", []), trace.get("neg_idx", [])):
                draws += [a, n]
            assert len(draws) == int(g[p + "n_draws"])
            for k, d in enumerate(draws):
                np.testing.assert_array_equal(d.numpy(), g[p + f"draw{k}"].astype(np.int64))
        np.testing.assert_allclose(loss.item(), float(g[p + "loss"]), rtol=1e-5, atol=1e-6)
        grad = rep.grad if rep.grad is not None else torch.zeros_like(rep)
        np.testing.assert_allclose(grad.cpu().numpy(), g[p + "grad"], rtol=1e-4, atol=1e-6)
        assert int(torch.randint(1 << 30, (1,))) == int(g[p + "probe"][0])


@pytest.mark.parametrize("cfg", [
    dict(b=2, n_cls=4, feat=496, spatial=(32, 32), func='smc', Q=256, Nn=512, qsize=1024),
    dict(b=1, n_cls=2, feat=16, spatial=(16, 16, 12), func='asmc', Q=256, Nn=512, qsize=700),
    dict(b=1, n_cls=19, feat=32, spatial=(24, 40), func='smc', Q=64, Nn=32, qsize=300),
])
def test_loss_vs_oracle_fresh(cfg):
    """Production feature width / 3-D / 19 classes against the CPU oracle, 2 chained steps."""
    from arco_amd.loss_helper import compute_contra_memobank_loss
    ikw = dict(b=cfg["b"], n_cls=cfg["n_cls"], feat=cfg["feat"], spatial=cfg["spatial"])
    lkw = dict(func=cfg["func"], num_queries=cfg["Q"], num_negatives=cfg["Nn"], delta_n=0.97)
    bank_g, ptr_g, qs = fx.fresh_bank(cfg["n_cls"], cfg["feat"], cfg["qsize"], 'zeros')
    bank_o, ptr_o, _ = fx.fresh_bank(cfg["n_cls"], cfg["feat"], cfg["qsize"], 'zeros')
    for step in range(2):
        inp = fx.loss_inputs(500 + step, **ikw)
        seed_all(42 + step)
        rep_o = inp["rep"].clone().requires_grad_(True)
        tr_o = {}
        nk_o, loss_o = orc.compute_contra_memobank_loss(
            rep_o, inp["label_l"], inp["label_u"], inp["prob_l"], inp["prob_u"], inp["low_mask"], inp["high_mask"],
            bank_o, ptr_o, qs, inp["rep_teacher"], trace=tr_o, **lkw)
        loss_o.backward()
        seed_all(42 + step)
        d = to_dev(inp)
        rep_g = d["rep"].clone().requires_grad_(True)
        tr_g = {}
        nk_g, loss_g = compute_contra_memobank_loss(
            rep_g, d["label_l"], d["label_u"], d["prob_l"], d["prob_u"], d["low_mask"], d["high_mask"],
            bank_g, ptr_g, qs, d["rep_teacher"], _trace=tr_g, **lkw)
        loss_g.backward()
        assert nk_g == nk_o
        assert [int(q) for q in ptr_g] == [int(q) for q in ptr_o]
        for bo, bg in zip(bank_o, bank_g):
            assert torch.equal(bo[0], bg[0].cpu())
        for a, b in zip(tr_o.get("anchor_idx", []), tr_g.get("anchor_idx", [])):
            assert torch.equal(a, b)
        for a, b in zip(tr_o.get("neg_idx", []), tr_g.get("neg_idx", [])):
            assert torch.equal(a, b)
        # compaction lists == torch boolean-mask order
        C, lists, tot = cfg["n_cls"], tr_g["lists"].cpu(), tr_g["totals"]
        for c in range(C):
            assert torch.equal(lists[c][:tot[C + c]].long(), tr_o["anchor_rows"][c])
            assert torch.equal(lists[C + c][:tot[2 * C + c]].long(), tr_o["neg_rows"][c])
        assert abs(loss_g.item() - loss_o.item()) < 1e-4 * max(1.0, abs(loss_o.item()))
        go = rep_o.grad if rep_o.grad is not None else torch.zeros_like(rep_o)
        gg = rep_g.grad if rep_g.grad is not None else torch.zeros_like(rep_g)
        np.testing.assert_allclose(gg.cpu().numpy(), go.numpy(), rtol=1e-3, atol=1e-6)


def test_no_grad_and_cpu_input_rejected():
    from arco_amd.loss_helper_3d import compute_contra_memobank_loss
    ikw, lkw, qsize, binit = fx.LOSS_CASES["d16_smc"]
    inp = fx.loss_inputs(11, **ikw)
    bank, ptr, qs = fx.fresh_bank(ikw["n_cls"], ikw["feat"], qsize, binit)
    with pytest.raises(RuntimeError):
        compute_contra_memobank_loss(inp["rep"], inp["label_l"], inp["label_u"], inp["prob_l"], inp["prob_u"],
                                     inp["low_mask"], inp["high_mask"], bank, ptr, qs, inp["rep_teacher"], **lkw)
    d = to_dev(inp)
    with torch.no_grad():
        nk, loss = compute_contra_memobank_loss(d["rep"], d["label_l"], d["label_u"], d["prob_l"], d["prob_u"],
                                                d["low_mask"], d["high_mask"], bank, ptr, qs, d["rep_teacher"], **lkw)
    assert not loss.requires_grad
