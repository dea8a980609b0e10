"""Host samplers of the product (arco_amd/samplers.py) vs golden vectors: bit-exact."""
import hashlib
import random

import numpy as np
import pytest
import torch

import fixture_inputs as fx
from arco_amd import samplers as S

FN = {"smc": S.grid_monte_carlo_sample, "asmc": S.grid_as_monte_carlo_sample,
      "mc1d": S.monte_carlo_sample, "asmc1d": S.as_monte_carlo_sample}


def seed_all(s):
    random.seed(s); np.random.seed(s); torch.manual_seed(s)


@pytest.mark.parametrize("name", list(FN))
def test_bit_exact(golden, name):
    g = golden["g1_samplers"]
    for high in fx.SAMPLER_HIGHS:
        for shape in fx.SAMPLER_SHAPES:
            for seed in fx.SAMPLER_SEEDS:
                key = f"{name}_h{high}_s{shape}_r{seed}"
                seed_all(seed)
                idx = FN[name](high, shape)
                assert idx.dtype == torch.int64 and tuple(idx.shape) == (shape,)
                np.testing.assert_array_equal(idx.numpy(), g[key].astype(np.int64), err_msg=key)
                assert [int(torch.randint(1 << 30, (1,))), random.randint(0, 1 << 30)] == g[key + "_probe"].tolist(), key


@pytest.mark.parametrize("name", ["smc", "asmc"])
def test_production_size(golden, name):
    g = golden["g1_samplers"]
    for high in (1, 4096, 29999, 50000):
        key = f"{name}_h{high}_s131072_r3"
        seed_all(3)
        idx = FN[name](high, 131072)
        assert hashlib.sha256(np.ascontiguousarray(idx.numpy()).tobytes()).digest() == g[key + "_sha"].tobytes(), key
        assert [int(torch.randint(1 << 30, (1,))), random.randint(0, 1 << 30)] == g[key + "_probe"].tolist()


def test_range_and_determinism():
    for high in (1, 5, 57, 1000, 65536, 300001):
        seed_all(11); a = S.grid_monte_carlo_sample(high, 4096)
        seed_all(11); b = S.grid_monte_carlo_sample(high, 4096)
        assert torch.equal(a, b) and int(a.min()) >= 0 and int(a.max()) < high


def test_native_and_python_paths_agree(monkeypatch):
    """The native mt19937 replay and the torch-call implementation give the same indices and leave
    the generator in the same state."""
    from arco_amd import _lib
    try:
        _lib.load()
    except RuntimeError:
        pytest.skip("library not built")
    for high in (57, 100, 4096, 5233, 30000, 262144, 1000003):
        for shape in (256, 131072):
            for mirror in (False, True):
                seed_all(high + shape)
                a = S._grid(high, shape, 4, mirror); pa = int(torch.randint(1 << 30, (1,)))
                seed_all(high + shape)
                b = S._grid_native(high, shape, 4, mirror); pb = int(torch.randint(1 << 30, (1,)))
                assert b is not NotImplemented
                assert torch.equal(a, b) and pa == pb, (high, shape, mirror)


@pytest.mark.parametrize("mirror", [False, True])
def test_grid_sample_many_equals_sequential_calls(mirror):
    """One threaded multi-call == the same calls one by one (indices AND final generator state), including
    value-dependent jobs (non-square highs), skipped-ahead jobs (square highs, large shapes) and 1-D fallbacks."""
    import random
    from arco_amd import samplers
    single = samplers.grid_as_monte_carlo_sample if mirror else samplers.grid_monte_carlo_sample
    jobs = [(5233, 256), (4096, 256 * 512), (90000, 256), (4096, 256 * 512), (13, 256), (1024, 16384), (300, 64),
            (4096, 256 * 512), (40, 256), (123457, 256), (65536, 40000)]
    torch.manual_seed(77); random.seed(5)
    ref = [single(h, sh) for h, sh in jobs]
    st_ref = torch.get_rng_state().clone()
    for threads in (0, 1, 8):
        torch.manual_seed(77); random.seed(5)
        got = samplers.grid_sample_many(jobs, mirror, max_threads=threads)
        assert torch.equal(torch.get_rng_state(), st_ref), threads
        for r, g_ in zip(ref, got):
            assert torch.equal(r, g_), threads


def _many_jobs():
    jobs = []
    for na in (572607, 86, 217, 5000):
        jobs += [(na, 256), (4096, 256 * 512)]
    return jobs


@pytest.mark.parametrize("pre", [3 << 20, 700000, 10000])
@pytest.mark.parametrize("threads", [8, 1])
@pytest.mark.parametrize("mirror", [False, True])
def test_pregenerated_generator_blocks_change_nothing(pre, threads, mirror):
    """samplers.pregen (the generator's next state blocks computed ahead of time, while the trainer waits for the GPU's
    counters) is pure acceleration: the sampler sequence of a step draws the same indices and leaves the generator in the
    same state - with enough blocks, with blocks that run out inside a negative call (700 000 draws) and inside the first
    anchor call (10 000), with and without the parallel paths."""
    from arco_amd import samplers
    jobs = _many_jobs()
    tot = sum(s for _, s in jobs)

    def run(n_pre):
        torch.manual_seed(11)
        torch.rand(7)                       # somewhere inside a state block
        if n_pre:
            samplers.pregen(n_pre)
        buf = torch.empty(tot, dtype=torch.int64)
        samplers.grid_sample_many(jobs, mirror, out=buf, max_threads=threads)
        return buf, torch.get_rng_state().clone(), torch.rand(3)

    ref, st_ref, nxt_ref = run(0)
    got, st, nxt = run(pre)
    assert torch.equal(ref, got) and torch.equal(st_ref, st) and torch.equal(nxt_ref, nxt)


def test_pregenerated_blocks_of_another_state_are_ignored():
    """Blocks are only used from exactly the generator state they were computed for: any draw in between makes them stale."""
    from arco_amd import samplers
    jobs = _many_jobs()
    tot = sum(s for _, s in jobs)
    out = []
    for stale in (True, False):
        torch.manual_seed(5)
        if stale:
            samplers.pregen(3 << 20)
        torch.rand(1)
        buf = torch.empty(tot, dtype=torch.int64)
        samplers.grid_sample_many(jobs, False, out=buf)
        out.append((buf, torch.get_rng_state().clone()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_anchor_calls_with_droppable_candidates_run_deferred(seed):
    """Anchor calls of classes with many candidates run in worker threads too (round 3): high is rarely a square, so picked values
    >= high can be dropped - but at most `shape` values are drawn, and then the final shuffle + padding consume shape - 1 draws
    whatever is dropped (loss_helper.py:419-431).  high = 70 000: 225 of 70 225 grid values are droppable, ~0.8 drops per call - the
    seeds cover calls with and without drops; 2 204 346 = the background class of the LA step.  Same indices, same generator state
    and same following draws as the calls made one by one."""
    import random
    from arco_amd import samplers
    # (1, 131072), (9, 300): the 1-D fallback's torch.randint branch (a volume step's banks keep one row) - native, does not cut the sequence
    jobs = [(70000, 256), (4096, 256 * 512), (2204346, 256), (1, 256 * 512), (191444, 256), (1, 256 * 512), (70000, 256), (9, 300), (66000, 256)]
    torch.manual_seed(seed); random.seed(seed)
    ref = [samplers.grid_as_monte_carlo_sample(h, sh) for h, sh in jobs]
    nxt_ref = torch.rand(4)
    st_ref = torch.get_rng_state().clone()
    assert any(int((r >= 0).sum()) == 256 for r in ref)
    for pre in (0, 4 << 20):
        torch.manual_seed(seed); random.seed(seed)
        if pre:
            samplers.pregen(pre)
        buf = torch.empty(sum(sh for _, sh in jobs), dtype=torch.int64)
        got = samplers.grid_sample_many(jobs, True, out=buf, defer=True)
        nxt = torch.rand(4)                                   # drawn while the worker calls may still run
        samplers.finish_many()
        assert torch.equal(nxt, nxt_ref) and torch.equal(torch.get_rng_state(), st_ref), pre
        for r, g_ in zip(ref, got):
            assert torch.equal(r, g_), pre


@pytest.mark.parametrize("pre", [3 << 20, 0])
def test_deferred_sampler_sequence(pre):
    """grid_sample_many(defer=True) returns once the generator holds its final state (the worker calls' draw counts are
    fixed); the caller may consume the generator before finish_many() - same indices, same generator draws afterwards; a
    following sampler call waits for the pending one by itself."""
    from arco_amd import samplers
    jobs = _many_jobs()
    tot = sum(s for _, s in jobs)

    def run(defer):
        torch.manual_seed(23)
        if pre:
            samplers.pregen(pre)
        buf = torch.empty(tot, dtype=torch.int64)
        samplers.grid_sample_many(jobs, True, out=buf, defer=defer)
        nxt = torch.rand(5)                                   # the next consumer (the equivariance warp in the trainers)
        extra = samplers.grid_as_monte_carlo_sample(5233, 256)     # a sampler call while workers may still run
        if defer:
            samplers.finish_many()
        return buf, nxt, extra, torch.get_rng_state().clone()

    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_forked_child_with_a_sequence_in_flight_does_not_hang():
    """a process forked while a deferred sampler sequence / a pregeneration is in flight (DataLoader workers)
    inherits `pending` counters whose threads do not exist in the child; the library's atfork child handler clears them,
    so the child's first sampler call returns (and gives the plain sequential answer)."""
    import os
    import signal
    from arco_amd import samplers
    torch.manual_seed(5)
    samplers.pregen(1 << 20)                                        # worker thread computing state blocks
    pid = os.fork()
    if pid == 0:                                                    # child: would spin forever in Group::wait without the handler
        signal.alarm(20)
        try:
            torch.manual_seed(9)
            out = samplers.grid_monte_carlo_sample(30000, 131072)
            os._exit(0 if int(out.numel()) == 131072 else 3)
        except BaseException:
            os._exit(4)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
    torch.manual_seed(9)
    a = samplers.grid_monte_carlo_sample(30000, 131072)            # the parent is unaffected
    assert int(a.numel()) == 131072


@pytest.mark.parametrize("numel", [16, 17, 1000, 1003, 624 * 5, 624 * 5 + 1, 1 << 20, (1 << 22) + 5])
def test_skip_randn_matches_torch_randn_consumption(numel):
    """samplers.skip_randn(n) leaves the CPU generator exactly where torch.randn(n) leaves it (the reference's random_pool
    draw, train_arco_2d.py:156, skipped without producing 4.7 GB of normals): same next draws of every kind."""
    from arco_amd import samplers
    for seed in (0, 1337):
        torch.manual_seed(seed)
        torch.randint(100, (7,))                                    # not at a block boundary
        st0 = torch.get_rng_state()
        torch.randn(numel)
        ref_state = torch.get_rng_state()
        ref_next = (torch.randint(1 << 30, (5,)), torch.rand(3), torch.randperm(11))
        torch.set_rng_state(st0)
        samplers.skip_randn(numel)
        assert torch.equal(torch.get_rng_state(), ref_state)
        got = (torch.randint(1 << 30, (5,)), torch.rand(3), torch.randperm(11))
        assert all(torch.equal(a, b) for a, b in zip(ref_next, got))


def test_skip_randn_shape_of_the_reference_pool():
    """K x 496 x H x W as the trainer calls it (small H, W here), drawn as a 4-D tensor like the reference."""
    from arco_amd import samplers
    torch.manual_seed(1337)
    torch.randn(6, 496, 8, 8)
    ref = torch.randint(1 << 30, (4,))
    torch.manual_seed(1337)
    samplers.skip_randn(6 * 496 * 8 * 8)
    assert torch.equal(torch.randint(1 << 30, (4,)), ref)
