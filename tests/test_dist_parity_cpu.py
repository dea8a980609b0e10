"""SURVEY §8e parity definition (ii), world size 2 over gloo on CPU:

  with num_queries / world anchors per rank (dist.anchors_for_rank, --anchors_per_rank split), prototypes of the global
  batch (dist.reduce_prototypes) and banks built from the rank-ordered key gather (dist.gather_keys - the reference's
  commented-out gather_together, loss_helper_3d.py:16-17), the RANK-AVERAGED contrastive loss - what the gradient
  all-reduce (dist.allreduce_grads) optimises - equals the single-process loss of the oracle on the CONCATENATED batch
  when the single process replays the ranks' sampled indices (rank-seeded replay); banks and pointers are identical;
  the rank-averaged gradient w.r.t. each rank's rep equals that rank's slice of the single-process gradient.

The product's exchange steps (arco_amd.dist) run for real under gloo; the per-rank arithmetic between them is the CPU
oracle's (the HIP kernels need a GPU - tools/ddp_check.py repeats this check on the product path under -m gpu).
"""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C, D, Q, NN, QS, DELTA_N, B, SP = 4, 16, 64, 16, 160, 0.97, 2, (24, 24)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _pieces(orc, inp):
    """Per-class row sets of one batch (oracle arithmetic of loss_helper_3d.py:341-401)."""
    R, T = orc._rows(inp["rep"]), orc._rows(inp["rep_teacher"]).detach()
    lab = orc._rows(torch.cat((inp["label_l"], inp["label_u"]), 0))
    prob = orc._rows(torch.cat((inp["prob_l"], inp["prob_u"]), 0))
    low_valid = lab * inp["low_mask"].reshape(-1, 1)
    high_valid = lab * inp["high_mask"].reshape(-1, 1)
    rank = orc.class_rank(prob)
    n_lab_rows = inp["label_l"].shape[0] * SP[0] * SP[1]
    labeled_row = torch.arange(lab.shape[0]) < n_lab_rows
    out = []
    for c in range(C):
        lv = low_valid[:, c].bool()
        anchor_m = (prob[:, c] > 0.3) & lv
        hard_m = (prob[:, c] < DELTA_N) & high_valid[:, c].bool()
        cls_u = (rank[:, c] >= 3) & (rank[:, c] < 20)
        cls_l = (rank[:, c] < 3) & (lab[:, c] == 0)
        neg_m = hard_m & torch.where(labeled_row, cls_l, cls_u)
        out.append(dict(lv=lv, anchor_rows=torch.nonzero(anchor_m).flatten(), neg_rows=torch.nonzero(neg_m).flatten()))
    return R, T, out


def _infonce(A, pos, bank, n_idx, q, temp=0.5):
    import torch.nn.functional as F
    eps = 1e-8
    An = A / A.norm(dim=1, keepdim=True).clamp_min(eps)
    Bn = bank / bank.norm(dim=1, keepdim=True).clamp_min(eps)
    Pn = pos / pos.norm().clamp_min(eps)
    logits = torch.cat(((An * Pn).sum(1, keepdim=True), (An @ Bn.t()).gather(1, n_idx.view(q, NN))), 1)
    return F.cross_entropy(logits / temp, torch.zeros(q).long())


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as td
    import arco_oracle as orc
    import fixture_inputs as fx
    from arco_amd import dist as adist
    adist.init(backend="gloo")
    q_rank = adist.anchors_for_rank(Q, "split")
    assert q_rank == Q // world and adist.anchors_for_rank(Q, "full") == Q
    assert [adist.anchors_for_rank(7, "split")] == [4 if rank == 0 else 3]         # remainder goes to the first ranks
    inp = fx.loss_inputs(50 + rank, b=B, n_cls=C, feat=D, spatial=SP)
    inp["rep"].requires_grad_(True)
    bank, ptr, qs = fx.fresh_bank(C, D, QS, 'zeros')
    losses = []
    traces = []
    for step in range(2):                                   # step 2 samples from banks that hold step 1's gathered keys
        R, T, pcs = _pieces(orc, inp)
        counts = torch.tensor([int(p_["lv"].sum()) for p_ in pcs])
        proto = torch.stack([T[p_["lv"]].mean(0) for p_ in pcs])
        proto = adist.reduce_prototypes(proto, counts)      # class means of the GLOBAL batch
        for c in range(C):                                  # rank-ordered key gather, then the reference's enqueue
            orc.dequeue_and_enqueue(adist.gather_keys(T[pcs[c]["neg_rows"]]), bank[c], ptr[c], qs[c])
        valid = [c for c in range(C) if counts[c] > 0]
        assert len(valid) == C                              # every class present on every rank (fixture property)
        torch.manual_seed(1000 * step + rank)               # rank-seeded sampler sequence
        loss = torch.zeros(())
        tr = []
        for k in range(len(valid)):
            a_idx = orc.grid_monte_carlo_sample(int(pcs[k]["anchor_rows"].shape[0]), q_rank)
            n_idx = orc.grid_monte_carlo_sample(int(bank[valid[k]][0].shape[0]), q_rank * NN)
            A = R[pcs[k]["anchor_rows"][a_idx]]
            loss = loss + _infonce(A, proto[k], bank[valid[k]][0], n_idx, q_rank)
            tr.append((pcs[k]["anchor_rows"][a_idx].clone(), n_idx.clone()))
        loss = loss / len(valid)
        losses.append(loss)
        traces.append(tr)
    inp["rep"].grad = None
    losses[-1].backward()
    g_local = inp["rep"].grad.clone()
    # what the data-parallel run optimises: the mean over ranks (gradient all-reduce / world)
    l_dp = losses[-1].detach().clone()
    td.all_reduce(l_dp); l_dp /= world

    # ---- single process on the concatenated batch, replaying the ranks' indices --------------------------------
    everything = [None] * world
    td.all_gather_object(everything, dict(inp={k: v.detach() for k, v in inp.items()}, traces=traces, grad=g_local))
    cat = lambda key: torch.cat([e["inp"][key] for e in everything], 0)
    nb = B
    rep_all = torch.cat([e["inp"]["rep"][:nb] for e in everything] + [e["inp"]["rep"][nb:] for e in everything], 0).requires_grad_(True)
    rept_all = torch.cat([e["inp"]["rep_teacher"][:nb] for e in everything] + [e["inp"]["rep_teacher"][nb:] for e in everything], 0)
    low_all = torch.cat([e["inp"]["low_mask"][:nb] for e in everything] + [e["inp"]["low_mask"][nb:] for e in everything], 0)
    high_all = torch.cat([e["inp"]["high_mask"][:nb] for e in everything] + [e["inp"]["high_mask"][nb:] for e in everything], 0)
    big = dict(rep=rep_all, rep_teacher=rept_all, label_l=cat("label_l"), label_u=cat("label_u"), prob_l=cat("prob_l"),
               prob_u=cat("prob_u"), low_mask=low_all, high_mask=high_all)
    P = SP[0] * SP[1]

    def to_global(rows, r):                                 # rank-local pixel row -> row of the concatenated batch
        lab_part = rows < nb * P
        return torch.where(lab_part, rows + r * nb * P, rows - nb * P + world * nb * P + r * nb * P)

    bank1, ptr1, qs1 = fx.fresh_bank(C, D, QS, 'zeros')
    real = orc.grid_monte_carlo_sample
    try:
        for step in range(2):
            _, _, gp = _pieces(orc, big)
            replay = []
            for k in range(C):
                g_rows = torch.cat([to_global(everything[r]["traces"][step][k][0], r) for r in range(world)])
                pos_in_list = torch.searchsorted(gp[k]["anchor_rows"], g_rows)
                assert torch.equal(gp[k]["anchor_rows"][pos_in_list], g_rows)         # every rank anchor is a global candidate
                replay += [pos_in_list, torch.cat([everything[r]["traces"][step][k][1] for r in range(world)])]
            it = iter(replay)
            orc.grid_monte_carlo_sample = lambda high, shape, cut_count=4: next(it)
            _, l_single = orc.compute_contra_memobank_loss(big["rep"], big["label_l"], big["label_u"], big["prob_l"], big["prob_u"],
                                                           big["low_mask"], big["high_mask"], bank1, ptr1, qs1, big["rep_teacher"],
                                                           delta_n=DELTA_N, func='smc', num_queries=Q, num_negatives=NN)
    finally:
        orc.grid_monte_carlo_sample = real
    assert abs(float(l_single) - float(l_dp)) < 2e-6 * max(1.0, abs(float(l_single))), (float(l_single), float(l_dp))
    for c in range(C):                                      # banks / pointers: bit-identical to the single process
        assert torch.equal(bank[c][0], bank1[c][0]) and int(ptr[c]) == int(ptr1[c]), c
    l_single.backward()
    g_all = rep_all.grad
    mine = torch.cat((g_all[rank * nb:(rank + 1) * nb], g_all[world * nb + rank * nb: world * nb + (rank + 1) * nb]), 0)
    assert torch.allclose(g_local / world, mine, rtol=1e-4, atol=1e-9)
    td.barrier()
    q.put((rank, float(l_dp)))


def test_rank_averaged_loss_equals_single_process_on_concatenated_batch():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got[0][1] == got[1][1] and got[0][1] > 0.1


def test_data_pipeline_seeds_differ_per_rank(tmp_path):
    """with --synthetic 0 every rank must draw its own samples and augmentations.  seed_data_pipeline(seed, rank)
    gives rank r the generators of seed + r: the loaders' index streams and the numpy / python / torch draws differ
    between ranks, rank 0 keeps the single-process sequence."""
    import random
    import numpy as np
    from torch.utils.data.sampler import RandomSampler
    sys.path.insert(0, ROOT)
    from arco_amd import dist as adist
    streams, probes = [], []
    for rank in (0, 1):
        random.seed(1337); np.random.seed(1337); torch.manual_seed(1337)
        g = adist.seed_data_pipeline(1337, rank)
        streams.append(list(RandomSampler(range(1000), replacement=True, num_samples=64, generator=g)))
        probes.append((random.random(), float(np.random.uniform()), float(torch.rand(1))))
    assert streams[0] != streams[1] and all(a != b for a, b in zip(probes[0], probes[1]))
    random.seed(1337); np.random.seed(1337); torch.manual_seed(1337)
    assert probes[0] == (random.random(), float(np.random.uniform()), float(torch.rand(1)))     # rank 0 = reference sequence
