"""Host side of the stratified samplers.

Bit-exact sample indices are DEFINED by the torch CPU default generator call sequence
of the reference (loss_helper_3d.py:35-268), so index generation is host logic by
construction: these functions issue exactly the same generator calls (randperm per
block, randint per block, one randperm shuffle, one draw per padded element - a
vector randint of n draws equals n single draws) but with closed-form block geometry
instead of materialised index images.  The device consumes the returned indices.
"""
import math
import random

import numpy as np
import torch

__all__ = ["grid_monte_carlo_sample", "grid_as_monte_carlo_sample", "monte_carlo_sample",
           "as_monte_carlo_sample", "grid_sample_many", "finish_many", "pregen"]


def _one_dim(high, shape, patch, mirror):
    # loss_helper_3d.py:35-80 (mirror=True) / :83-117
    if high // patch > shape or high < patch:
        return torch.randint(high, size=(shape,))
    blocks = high // patch
    per = shape // blocks
    vals = []
    for b in range(blocks):
        lo = b * patch
        if mirror:
            half = [random.randint(lo, lo + patch - 1) for _ in range(per // 2)]
            vals += half
            vals += [2 * lo + patch - 1 - v for v in half]
        else:
            vals += [random.randint(lo, lo + patch - 1) for _ in range(per)]
    vals += [random.randint(0, high - 1) for _ in range(shape - len(vals))]
    out = torch.tensor(vals, dtype=torch.float32).long()          # float32 round trip (:74/:111)
    return out[torch.randperm(shape)]


def monte_carlo_sample(high=5233, shape=256, patch=16):
    return _one_dim(high, shape, patch, False)


def as_monte_carlo_sample(high=5233, shape=256, patch=16):
    return _one_dim(high, shape, patch, True)


def _grid(high, shape, cut, mirror):
    # loss_helper_3d.py:120-184 / :187-268.  Only the generator calls go through torch (they
    # define the sequence); the index arithmetic is numpy (no intra-op thread fan-out on tiny arrays).
    edge = round(math.sqrt(high))
    side = edge // cut
    if side <= 1:
        # first block empty or single element: the reference raises before drawing anything
        # (randint(0,..) / scalar .shape) and falls back to the 1-D sampler
        return None
    per_block = shape * edge * edge // high // (cut * cut)
    take = per_block // 2 if mirror else per_block
    last = edge - (cut - 1) * side
    nblk = cut * cut
    local = np.empty((nblk, take), dtype=np.int64)
    wid = np.empty((nblk, 1), dtype=np.int64)
    org = np.empty((nblk, 1), dtype=np.int64)
    fin = np.empty((nblk, 1), dtype=np.int64)
    i = 0
    for bi in range(cut):
        h = last if bi == cut - 1 else side
        for bj in range(cut):
            w = last if bj == cut - 1 else side
            n = h * w
            perm = torch.randperm(n).numpy()
            local[i] = perm[torch.randint(n, (take,)).numpy()]
            wid[i, 0] = w
            org[i, 0] = (bi * side) * edge + bj * side
            # int64(2*mean(block)) == first + last element of the rectangular block (exact)
            fin[i, 0] = org[i, 0] + (bi * side + h - 1) * edge + bj * side + w - 1
            i += 1
    val = org + (local // wid) * edge + local % wid
    if mirror:
        val = np.stack((val, fin - val), axis=1)          # block0 picks, block0 mirrors, block1 picks, ...
    vals = val.reshape(-1).astype(np.float32).astype(np.int64)     # float32 round trip (:163 / :245-246)
    vals = vals[vals < high]
    vals = vals[torch.randperm(vals.shape[0]).numpy()]
    if vals.shape[0] < shape:
        vals = np.concatenate([vals, torch.randint(high, (shape - vals.shape[0],)).numpy()])
    return torch.from_numpy(np.ascontiguousarray(vals[:shape]))


def _grid_native(high, shape, cut, mirror):
    """Same draws through the native mt19937 replay (csrc/sampler_host.hip) on the serialized
    generator state; returns NotImplemented when the library / state layout is unavailable."""
    try:
        from . import _lib
        lib = _lib.load()
    except (RuntimeError, OSError):
        return NotImplemented
    if high >= 2 ** 31:
        return NotImplemented
    st = torch.get_rng_state()
    out = torch.empty(shape, dtype=torch.int64)
    rc = lib.arco_grid_sample(st.data_ptr(), st.numel(), int(high), int(shape), int(cut), int(mirror), out.data_ptr())
    if rc == 0:
        return None                      # reference falls back to the 1-D sampler, nothing drawn
    if rc != shape:
        return NotImplemented
    torch.set_rng_state(st)
    return out


def _grid_any(high, shape, cut, mirror):
    out = _grid_native(high, shape, cut, mirror)
    return _grid(high, shape, cut, mirror) if out is NotImplemented else out


@torch.no_grad()
def grid_monte_carlo_sample(high=5233, shape=256, cut_count=4):
    out = _grid_any(high, shape, cut_count, False)
    return monte_carlo_sample(high, shape) if out is None else out


@torch.no_grad()
def grid_as_monte_carlo_sample(high=5233, shape=256, cut_count=4):
    out = _grid_any(high, shape, cut_count, True)
    return as_monte_carlo_sample(high, shape) if out is None else out


@torch.no_grad()
def pregen(n_draws=3 << 20, background=True):
    """Compute the torch CPU generator's next state blocks for >= n_draws draws ahead of time (native worker thread; the
    generator itself is not touched).  The trainers call this right before they block on the GPU's per-class counters:
    the next grid_sample_many call from the unchanged generator state then reads the blocks instead of regenerating -
    skip-ahead becomes O(1), the grid blocks of the anchor calls and the negative calls run in parallel.  Pure
    acceleration: same draws, same final generator state; from any other generator state the blocks are ignored."""
    from . import _lib
    st = torch.get_rng_state()
    rc = _lib.load().arco_mt_pregen(st.data_ptr(), st.numel(), int(n_draws), 1 if background else 0)
    if rc < 0:
        raise RuntimeError(f"arco_mt_pregen failed ({rc})")
    return int(rc)


def finish_many():
    """Wait for the worker calls of a grid_sample_many(..., defer=True) sequence: its output buffer is complete afterwards."""
    from . import _lib
    _lib.load().arco_grid_sample_many_finish()


_MAX_THREADS = int(__import__('os').environ.get('ARCO_SAMPLER_THREADS', '8'))


def grid_sample_many(jobs, mirror, cut_count=4, out=None, max_threads=None, defer=False):
    """[(high, shape), ...] -> list of index tensors, the SAME draws as calling grid_(as_)monte_carlo_sample
    for each job in order, through ONE native call: jobs whose generator consumption does not depend on the drawn
    values (the negative draws: high = bank length = a perfect square) run in worker threads on copies of the
    generator state while the generator is skipped ahead (csrc/sampler_host.hip).  `out`: optional int64 CPU
    tensor of sum(shape) elements (e.g. pinned) that receives the jobs back to back.
    defer=True: return as soon as the generator holds its final state (the worker calls' draw counts are known before
    they run) - the outputs are complete only after finish_many(); the caller may consume the generator meanwhile."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    single = grid_as_monte_carlo_sample if mirror else grid_monte_carlo_sample
    n = len(jobs)
    total = sum(int(sh) for _, sh in jobs)
    buf = out if out is not None else torch.empty(total, dtype=torch.int64)
    assert buf.dtype == torch.int64 and buf.numel() >= total and buf.is_contiguous()
    views, off = [], 0
    for _, sh in jobs:
        views.append(buf[off:off + int(sh)])
        off += int(sh)
    if any(int(h) >= 2 ** 31 for h, _ in jobs):
        for v, (h, sh) in zip(views, jobs):
            v.copy_(single(int(h), int(sh), cut_count))
        return views
    first = 0
    while first < n:
        m = n - first
        highs = (ctypes.c_long * m)(*[int(h) for h, _ in jobs[first:]])
        shapes = (ctypes.c_long * m)(*[int(sh) for _, sh in jobs[first:]])
        outs = (ctypes.c_void_p * m)(*[v.data_ptr() for v in views[first:]])
        st = torch.get_rng_state()
        fn = lib.arco_grid_sample_many_async if defer else lib.arco_grid_sample_many
        rc = fn(st.data_ptr(), st.numel(), m, highs, shapes, int(cut_count), int(bool(mirror)), outs, int(max_threads or _MAX_THREADS))
        if rc < 0:
            raise RuntimeError(f"arco_grid_sample_many failed ({rc})")
        torch.set_rng_state(st)
        first += int(rc)
        if first < n:                     # this job takes the reference's 1-D fallback (python `random` + torch)
            lib.arco_grid_sample_many_finish()
            h, sh = jobs[first]
            views[first].copy_(single(int(h), int(sh), cut_count))
            first += 1
    return views


def skip_randn(numel):
    """Advance torch's CPU default generator exactly as `torch.randn(numel)` (float32, contiguous, numel >= 16) would,
    without producing the values: ATen's normal_fill draws one 32-bit word per element (uniform_real_distribution<float>)
    and, when numel is not a multiple of 16, 16 more for the recomputed tail - aten/src/ATen/native/cpu/
    DistributionTemplates.h.  Used for the reference's 4.7 GB `random_pool` draw (train_arco_2d.py:156) when the
    revisiting term is off, so a seeded run keeps the reference's generator sequence (weight initialisation, samplers,
    warps) without the pool.  Pinned against torch.randn itself in tests/test_samplers_host.py."""
    from . import _lib
    numel = int(numel)
    assert numel >= 16, "the serial path of small tensors draws differently (normal_distribution<double>)"
    st = torch.get_rng_state()
    rc = _lib.load().arco_mt_skip(st.data_ptr(), st.numel(), numel + (16 if numel % 16 else 0))
    if rc != 0:
        raise RuntimeError(f"arco_mt_skip failed ({rc})")
    torch.set_rng_state(st)
