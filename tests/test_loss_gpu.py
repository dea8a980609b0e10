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
    g = golden["g2_loss"]
    ikw, lkw, qsize, binit = fx.LOSS_CASES[case]
    if len(ikw["spatial"]) == 3:      # 5-D tensors go through the module the 3-D trainer imports (train_arco_3d.py:22)
        from arco_amd.loss_helper import compute_contra_memobank_loss
    else:
        from arco_amd.loss_helper_3d import compute_contra_memobank_loss
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


def test_loss_full_size_properties():
    """BASELINE.json configs[1] size (16 images 256x256, D = 496, 4 classes, 4096-key queues, 256 queries, 512
    negatives, smc): the oracle needs minutes there, so the HIP path is held to size-independent properties -
    bit-exact sampled indices against the host sampler replay, FIFO banks made of teacher rows, cosine scale
    invariance, a gradient that lives on the sampled anchor pixels only and is orthogonal to each anchor row."""
    from arco_amd import samplers
    from arco_amd.loss_helper_3d import compute_contra_memobank_loss
    b, C, D, sp, Q, Nn, qsize = 8, 4, 496, (256, 256), 256, 512, 4096
    small = fx.loss_inputs(77, b=b, n_cls=C, feat=1, spatial=sp)           # labels / probabilities / masks (CPU, small)
    g = torch.Generator(device="cuda").manual_seed(5)
    rep = torch.randn((2 * b, *sp, D), device="cuda", generator=g).permute(0, 3, 1, 2)          # channels-last, 2 GB
    rep_t = torch.randn((2 * b, *sp, D), device="cuda", generator=g).permute(0, 3, 1, 2)
    d = {k: v.cuda() for k, v in small.items() if k not in ("rep", "rep_teacher")}
    lkw = dict(func='smc', num_queries=Q, num_negatives=Nn, delta_n=0.97)

    def run(scale, steps=3):
        bank, ptr, qs = fx.fresh_bank(C, D, qsize, 'zeros')
        out = []
        for s in range(steps):
            seed_all(90 + s)
            r = (rep * scale).detach().requires_grad_(True)
            tr = {}
            nk, loss = compute_contra_memobank_loss(r, d["label_l"], d["label_u"], d["prob_l"], d["prob_u"], d["low_mask"],
                                                    d["high_mask"], bank, ptr, qs, rep_t, _trace=tr, **lkw)
            loss.backward()
            out.append((nk, float(loss.detach()), r.grad, tr))
        return out, bank, ptr

    out1, bank, ptr = run(1.0)
    # banks: FIFO of at most queue_size teacher rows, pointer = queue_size once full
    tr = out1[-1][3]
    lists, tot = tr["lists"], tr["totals"]
    flat_t = rep_t.permute(0, 2, 3, 1).reshape(-1, D)
    for c in range(C):          # every step enqueues the same key pixels: bank = teacher rows of the last qsize of them
        n_keys = int(tot[2 * C + c])
        assert out1[-1][0][c] == n_keys and 3 * n_keys >= qsize, (c, n_keys)
        assert bank[c][0].shape == (qsize, D) and int(ptr[c]) == qsize and bank[c][0].is_cuda
        rows = lists[C + c][:n_keys].long()
        assert torch.equal(bank[c][0], flat_t[torch.cat((rows, rows, rows))[-qsize:]])
    # sampled indices: bit-exact replay of the host samplers from the same generator state (last step: full banks)
    seed_all(92)
    exp_anchor, exp_neg = [], []
    for c in range(C):
        n_anchor = int(tot[C + c])
        exp_anchor.append(samplers.grid_monte_carlo_sample(n_anchor, Q))
        exp_neg.append(samplers.grid_monte_carlo_sample(qsize, Q * Nn))
    for c in range(C):
        assert torch.equal(tr["anchor_idx"][c].cpu(), exp_anchor[c]) and torch.equal(tr["neg_idx"][c].cpu(), exp_neg[c])
        assert int(tr["anchor_idx"][c].max()) < int(tot[C + c]) and int(tr["neg_idx"][c].max()) < qsize
    # gradient: non-zero exactly on the sampled anchor pixels, orthogonal to the anchor rows (cosine similarity)
    grad = out1[-1][2].permute(0, 2, 3, 1).reshape(-1, D)
    nz = (grad != 0).any(dim=1)
    pix = torch.cat([lists[c][:tot[C + c]].long()[tr["anchor_idx"][c].cuda()] for c in range(C)])
    assert set(torch.nonzero(nz).flatten().tolist()) <= set(pix.tolist()) and int(nz.sum()) >= 0.9 * len(set(pix.tolist()))
    flat_r = rep.permute(0, 2, 3, 1).reshape(-1, D)
    upix = torch.unique(pix)
    dots = (grad[upix] * flat_r[upix]).sum(1).abs() / (grad[upix].norm(dim=1) * flat_r[upix].norm(dim=1) + 1e-30)
    assert float(dots.max()) < 1e-4
    # cosine similarity: scaling the student representation changes nothing but the gradient's scale
    out2, _, _ = run(4.0)
    for (nk1, l1, g1, _), (nk2, l2, g2, _) in zip(out1, out2):
        assert nk1 == nk2 and abs(l1 - l2) < 2e-5 * max(1.0, abs(l1))
        np.testing.assert_allclose((4.0 * g2[:, :, ::16, ::16]).cpu().numpy(), g1[:, :, ::16, ::16].cpu().numpy(), rtol=2e-3, atol=1e-9)
    assert 0.5 < out1[-1][1] < 20.0


@pytest.mark.parametrize("cfg", [dict(b=2, n_cls=4, feat=64, spatial=(32, 32), Q=64, Nn=32, qs=300),
                                 dict(b=1, n_cls=19, feat=496, spatial=(32, 48), Q=32, Nn=64, qs=100),
                                 dict(b=1, n_cls=2, feat=16, spatial=(8, 12, 10), Q=48, Nn=16, qs=40)])
def test_grouped_infonce_equals_per_class_launches(cfg):
    """The grouped InfoNCE (all classes per launch: batched MFMA GEMMs, LDS multiplicities) against the per-class launch
    sequence on chained steps with ragged bank lengths: same loss (1e-6), same gradient, identical banks."""
    from arco_amd import _contrast as C_
    from arco_amd.loss_helper_3d import compute_contra_memobank_loss
    inp = to_dev(fx.loss_inputs(11, b=cfg["b"], n_cls=cfg["n_cls"], feat=cfg["feat"], spatial=cfg["spatial"]))
    out = {}
    try:
        for grouped in (True, False):
            C_.GROUPED = grouped
            bank, ptr, qs = fx.fresh_bank(cfg["n_cls"], cfg["feat"], cfg["qs"], 'zeros')
            res = []
            for s in range(3):
                seed_all(40 + s)
                r = inp["rep"].clone().requires_grad_(True)
                nk, loss = compute_contra_memobank_loss(r, inp["label_l"], inp["label_u"], inp["prob_l"], inp["prob_u"], inp["low_mask"],
                                                        inp["high_mask"], bank, ptr, qs, inp["rep_teacher"], func='smc',
                                                        num_queries=cfg["Q"], num_negatives=cfg["Nn"], delta_n=0.97)
                loss.backward()
                res.append((nk, float(loss.detach()), r.grad.clone()))
            out[grouped] = (res, [b[0].clone() for b in bank])
    finally:
        C_.GROUPED = True
    for (nk1, l1, g1), (nk2, l2, g2) in zip(out[True][0], out[False][0]):
        assert nk1 == nk2 and abs(l1 - l2) <= 2e-6 * max(1.0, abs(l2)), (l1, l2)
        np.testing.assert_allclose(g1.cpu().numpy(), g2.cpu().numpy(), rtol=1e-4, atol=1e-9)
    for b1, b2 in zip(out[True][1], out[False][1]):
        assert torch.equal(b1, b2)
    assert abs(out[True][0][-1][1]) > 1e-3


@pytest.mark.parametrize("cfg", [dict(b=2, n_cls=4, feat=64, spatial=(32, 32), Q=64, Nn=32, qs=300),
                                 dict(b=2, n_cls=4, feat=496, spatial=(64, 64), Q=256, Nn=512, qs=4096),
                                 dict(b=1, n_cls=19, feat=496, spatial=(32, 48), Q=32, Nn=64, qs=100),
                                 dict(b=1, n_cls=2, feat=16, spatial=(8, 12, 10), Q=48, Nn=16, qs=40)])
def test_score_gemm_with_softmax_ce_epilogue_equals_the_staged_route(cfg):
    """Round 6 (north_star: "contrastive score as an MFMA GEMM with fused temperature-scaled softmax-CE"; loss_helper_3d.py:503-509):
    arco_nce_prep / arco_nce_score / arco_nce_finish (no score matrix, no normalised bank copy, five launches) against the staged
    grouped route (normalised banks, score GEMM, arco_nce_fused) on chained steps with growing, ragged banks - incl. the headline's
    D = 496, 256 x 512 samples, 4096-key queues: loss 2e-6, gradient 1e-4 of its scale, identical banks and sampled indices."""
    from arco_amd import _contrast as C_
    from arco_amd.loss_helper_3d import compute_contra_memobank_loss
    inp = to_dev(fx.loss_inputs(13, b=cfg["b"], n_cls=cfg["n_cls"], feat=cfg["feat"], spatial=cfg["spatial"]))
    out = {}
    prev = C_.NCE_FUSED
    try:
        for fused in (1, 0):
            C_.NCE_FUSED = fused
            bank, ptr, qs = fx.fresh_bank(cfg["n_cls"], cfg["feat"], cfg["qs"], 'zeros')
            res = []
            for s in range(4):
                seed_all(60 + s)
                r = inp["rep"].clone().requires_grad_(True)
                nk, loss = compute_contra_memobank_loss(r, inp["label_l"], inp["label_u"], inp["prob_l"], inp["prob_u"], inp["low_mask"],
                                                        inp["high_mask"], bank, ptr, qs, inp["rep_teacher"], func='smc',
                                                        num_queries=cfg["Q"], num_negatives=cfg["Nn"], delta_n=0.97)
                loss.backward()
                res.append((nk, float(loss.detach()), r.grad.clone()))
            out[fused] = (res, [b[0].clone() for b in bank])
    finally:
        C_.NCE_FUSED = prev
    for (nk1, l1, g1), (nk2, l2, g2) in zip(out[1][0], out[0][0]):
        assert nk1 == nk2 and abs(l1 - l2) <= 2e-6 * max(1.0, abs(l2)), (l1, l2)
        scale = float(g2.abs().max())
        assert float((g1 - g2).abs().max()) <= 1e-4 * scale, (float((g1 - g2).abs().max()), scale)
    for b1, b2 in zip(out[1][1], out[0][1]):
        assert torch.equal(b1, b2)
    assert abs(out[1][0][-1][1]) > 1e-3
