"""world_size-2 gloo tests of the data-parallel exchange steps on CPU:
flat gradient all-reduce, key-count all-reduce, key all-gather-v in rank order."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as td
    from arco_amd import dist as adist
    from arco_amd import _contrast
    r, w = adist.init(backend="gloo")
    assert (r, w) == (rank, world) and adist.is_dist()
    assert _contrast.key_gather_hook is adist.gather_keys and _contrast.count_gather_hook is adist.gather_counts
    # 1. key all-gather-v: ragged counts incl. an empty rank, rank order, exact values
    for n_by_rank in ([3, 5], [0, 4], [0, 0], [7, 1]):
        n = n_by_rank[rank]
        keys = torch.arange(n * 4, dtype=torch.float32).view(n, 4) + 1000 * rank
        out = adist.gather_keys(keys)
        exp = torch.cat([torch.arange(m * 4, dtype=torch.float32).view(m, 4) + 1000 * rr for rr, m in enumerate(n_by_rank)])
        assert torch.equal(out, exp), (rank, n_by_rank)
    # 2. key counters summed identically on every rank
    assert adist.gather_counts([rank + 1, 10 * rank, 0, 7]) == [3, 10, 0, 14]
    # 3. flat gradient all-reduce = mean over ranks, every parameter marked as touched
    class Opt:
        pass
    opt = Opt()
    opt.flat_g = torch.full((17,), float(rank + 1))
    opt.params = [0, 1, 2]
    opt._touched = set()
    adist.allreduce_grads(opt)
    assert torch.allclose(opt.flat_g, torch.full((17,), 1.5)) and opt._touched == {0, 1, 2}
    # 4. broadcast of module state from rank 0
    m = torch.nn.Linear(3, 2)
    with torch.no_grad():
        m.weight.fill_(float(rank + 5))
    adist.broadcast_module_states([m])
    assert float(m.weight[0, 0]) == 5.0
    # 5. banks built from gathered keys are identical on all ranks (CPU emulation of the FIFO truncation)
    old = torch.zeros(1, 4)
    keys = torch.randn(6 + rank, 4, generator=torch.Generator().manual_seed(rank))
    allk = adist.gather_keys(keys)
    bank = torch.cat((old, allk))[-8:]
    gathered = [torch.empty_like(bank) for _ in range(world)]
    td.all_gather(gathered, bank)
    assert all(torch.equal(g, bank) for g in gathered)
    # 5b. prototypes: count-weighted mean over ranks == mean over the concatenated rows; absent classes handled
    rows = [torch.randn(5 + 3 * rr, 6, generator=torch.Generator().manual_seed(40 + rr)) for rr in range(world)]
    cls = [torch.tensor([0, 1, 1, 0, 1]), torch.tensor([1, 1, 3, 1, 3, 1, 1, 3])]          # class 0 only on rank 0, 3 only on rank 1, 2 nowhere
    mine, lab = rows[rank], cls[rank]
    counts = torch.tensor([int((lab == c).sum()) for c in range(4)])
    proto = torch.stack([mine[lab == c].mean(0) if counts[c] > 0 else torch.full((6,), float("nan")) for c in range(4)])
    got = adist.reduce_prototypes(proto, counts)
    allr, alll = torch.cat(rows), torch.cat(cls)
    for c in (0, 1, 3):
        assert torch.allclose(got[c], allr[alll == c].mean(0), atol=1e-6), (rank, c)
    assert bool(torch.isnan(got[2]).all())
    assert _contrast.proto_reduce_hook is adist.reduce_prototypes
    # 6. tail gather: only the rows that survive the truncation travel; the bank equals the all-gather-v one
    D = 4
    for qs, n_by_rank in ((8, [20, 3]), (8, [5, 20]), (8, [2, 3]), (8, [0, 5]), (8, [6, 0]), (8, [0, 0]), (5, [5, 5])):
        n = n_by_rank[rank]
        full = torch.arange(n * D, dtype=torch.float32).view(n, D) + 1000 * rank + 7 * qs
        adist.gather_counts([0, n, 1])                       # class 1 carries this case's counts
        mine = full[max(0, n - qs):]                           # the last min(n, qs) local rows
        tail = adist.gather_tail_keys(mine, 1, qs)
        ref = adist.gather_keys(mine)
        n_all = sum(n_by_rank)
        old_bank = torch.full((3, D), -1.0)
        banks = [torch.cat((old_bank, got))[-qs:] for got in (tail, ref)]     # CPU emulation of _append's FIFO
        assert torch.equal(banks[0], banks[1]), (rank, qs, n_by_rank)
        assert n_all >= 0
        assert tail.shape[0] == min(qs, sum(min(m, qs) for m in n_by_rank))
    # 7. the all-classes form (one broadcast per contributing rank) returns what the per-class form returns
    for qs, table in ((8, [[20, 3], [0, 5], [2, 3]]), (5, [[5, 5], [0, 0], [9, 1]]), (6, [[1, 0], [0, 2], [0, 0]])):
        counts = [t[rank] for t in table]
        adist.gather_counts(counts)
        mine = [(torch.arange(n * D, dtype=torch.float32).view(n, D) + 1000 * rank + 100 * c)[max(0, n - qs):] for c, n in enumerate(counts)]
        per_class = [adist.gather_tail_keys(m, c, qs) for c, m in enumerate(mine)]
        batched = adist.gather_tail_keys_all(mine, [qs] * len(table))
        assert all(torch.equal(a, b) for a, b in zip(per_class, batched)), (rank, qs, table)
    assert _contrast.tail_gather_all_hook is adist.gather_tail_keys_all
    # 8. counters gathered as a device table (no host-side collective): [3C] -> [world, 3C]; the per-rank key-count table
    #    set from it gives the same sums / tail exchange as the host-side gather_counts
    C3 = torch.tensor([rank, 5 * rank + 1, 0, 11, 7 * rank, 2, 3 + rank, 0, 9 * rank], dtype=torch.int64)      # C = 3: [n_lv | n_anchor | n_neg]
    tab = adist.gather_totals_device(C3)
    assert tuple(tab.shape) == (world, 9) and all(torch.equal(tab[rr], torch.tensor([rr, 5 * rr + 1, 0, 11, 7 * rr, 2, 3 + rr, 0, 9 * rr])) for rr in range(world))
    sums = adist.set_rank_counts([row[6:] for row in tab.tolist()])
    assert sums == [sum(3 + rr for rr in range(world)), 0, sum(9 * rr for rr in range(world))]
    assert sums == adist.gather_counts([3 + rank, 0, 9 * rank])
    assert _contrast.totals_gather_hook is adist.gather_totals_device and _contrast.count_table_hook is adist.set_rank_counts
    # 9. two gradient buckets: the heads' tail is all-reduced from inside the backward (mark_heads_done), the rest in
    #    allreduce_grads; the result is the plain mean
    class Opt2:
        pass
    o2 = Opt2()
    o2.flat_g = torch.zeros(20)
    o2.params = [0, 1]
    o2._touched = set()
    x = torch.ones(3, requires_grad=True)
    (m0,) = adist.mark_heads_done([x * 2.0], o2, 12)
    o2.flat_g[:12] = float(rank + 1); o2.flat_g[12:] = float(10 * (rank + 1))       # "gradients" present before the backward reaches the marker
    m0.sum().backward()
    assert len(adist._pending_buckets) == 1 and torch.equal(x.grad, torch.full((3,), 2.0))
    adist.allreduce_grads(o2)
    mean = sum(rr + 1 for rr in range(world)) / world
    assert torch.allclose(o2.flat_g[:12], torch.full((12,), mean)) and torch.allclose(o2.flat_g[12:], torch.full((8,), 10 * mean))
    assert adist._pending_buckets == []
    # 9b. ADVICE r3 (high): the marker's backward fires only on ranks whose loss reaches the heads - a background-only batch
    #     takes the loss's zero path (train_arco_2d.py: `weight.sum() * 0`) and never touches the marker.  Every rank must still
    #     issue the SAME two collectives, [start:] then [:start]: here rank 0 fires, the other ranks do not; then the reverse;
    #     then nobody fires; a stale handle left by an aborted step and a double backward must not change the sequence either.
    for firing in ([0], [r_ for r_ in range(world) if r_ != 0], [], list(range(world))):
        x = torch.ones(3, requires_grad=True)
        (m0,) = adist.mark_heads_done([x * 2.0], o2, 12)
        assert o2._bucket_start == 12
        o2.flat_g[:12] = float(rank + 1); o2.flat_g[12:] = float(10 * (rank + 1))
        if rank in firing:
            m0.sum().backward(retain_graph=True)
            m0.sum().backward()                                 # a second backward through the marker: ignored
            assert len(adist._pending_buckets) == 1
        else:
            (x * 0.0).sum().backward()                          # the zero path: attached to the graph, not to the marker
            assert adist._pending_buckets == []
        adist.allreduce_grads(o2)
        assert torch.allclose(o2.flat_g[:12], torch.full((12,), mean)) and torch.allclose(o2.flat_g[12:], torch.full((8,), 10 * mean)), firing
        assert adist._pending_buckets == [] and o2._bucket_start is None
    # an aborted step (the hook fired, allreduce_grads never ran - every rank, as an exception in the shared step code would):
    # the next step's mark_heads_done drops the stale handle and the exchange is the usual pair
    x = torch.ones(3, requires_grad=True)
    (m0,) = adist.mark_heads_done([x * 2.0], o2, 12)
    m0.sum().backward()
    assert len(adist._pending_buckets) == 1
    x = torch.ones(3, requires_grad=True)
    (m0,) = adist.mark_heads_done([x * 2.0], o2, 12)
    assert adist._pending_buckets == []
    o2.flat_g[:12] = float(rank + 1); o2.flat_g[12:] = float(10 * (rank + 1))
    m0.sum().backward()
    adist.allreduce_grads(o2)
    assert torch.allclose(o2.flat_g[:12], torch.full((12,), mean)) and torch.allclose(o2.flat_g[12:], torch.full((8,), 10 * mean))
    # without a marker in the step: one collective over the whole buffer
    o2.flat_g[:] = float(rank + 1)
    adist.allreduce_grads(o2)
    assert torch.allclose(o2.flat_g, torch.full((20,), mean))
    # 10. anchors per rank: the split covers num_queries exactly, the loss weights average to one
    for Q in (256, 50, 7):
        if Q >= world:
            q_r = adist.anchors_for_rank(Q, "split")
            t = torch.tensor([float(q_r), adist.anchor_weight(Q, "split")], dtype=torch.float64)
            td.all_reduce(t)
            assert int(t[0]) == Q and abs(float(t[1]) / world - 1.0) < 1e-12
    if world > 2:
        try:
            adist.anchors_for_rank(world - 1, "split")
            raise AssertionError("expected a ValueError")
        except ValueError:
            pass
    td.barrier()
    q.put((rank, "ok"))


def _worker4(rank, world, port, q):
    """world 4, ragged / empty ranks: key all-gather-v, the all-classes tail exchange through the FIFO truncation, the
    device counter table, Q = 50 anchors split 13 / 13 / 12 / 12."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as td
    from arco_amd import dist as adist
    adist.init(backend="gloo")
    D = 4
    for qs, table in ((8, [[20, 0, 3, 1], [0, 0, 0, 5], [2, 3, 0, 0], [0, 0, 0, 0]]), (5, [[5, 5, 5, 5], [0, 9, 0, 1], [1, 0, 0, 0], [3, 3, 3, 3]]),
                      (16, [[1, 2, 3, 4], [0, 0, 7, 0], [16, 0, 0, 1], [40, 1, 0, 0]])):
        counts = [t[rank] for t in table]
        tab = adist.gather_totals_device(torch.tensor([0] * 8 + counts, dtype=torch.int64))        # C = 4: n_neg is the last third
        assert adist.set_rank_counts([row[8:] for row in tab.tolist()]) == [sum(t) for t in table]
        mine = [(torch.arange(n * D, dtype=torch.float32).view(n, D) + 1000 * rank + 100 * c)[max(0, n - qs):] for c, n in enumerate(counts)]
        batched = adist.gather_tail_keys_all(mine, [qs] * len(table))
        for c, got in enumerate(batched):
            ref = adist.gather_keys(mine[c])                                                       # all-gather-v, rank order
            old_bank = torch.full((3, D), -1.0)
            assert torch.equal(torch.cat((old_bank, got))[-qs:], torch.cat((old_bank, ref))[-qs:]), (rank, qs, c)
    assert adist.anchors_for_rank(50, "split") == (13 if rank < 2 else 12)
    w = torch.tensor([adist.anchor_weight(50, "split")], dtype=torch.float64)
    td.all_reduce(w)
    assert abs(float(w) / world - 1.0) < 1e-12
    try:
        adist.anchors_for_rank(3, "split")
        raise AssertionError("expected a ValueError")
    except ValueError:
        pass
    td.barrier()
    q.put((rank, "ok"))


def _run(world, target):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(world)) == [(r, "ok") for r in range(world)]


def test_world4_gloo_ragged_ranks():
    _run(4, _worker4)


def test_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, "ok"), (1, "ok")]
