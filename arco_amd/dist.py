"""Data-parallel plumbing: one process per GPU, torch.distributed backend "nccl" (= RCCL over
xGMI on ROCm).  SURVEY §8e.  The path shards by batch; three exchange steps exist:

 1. gradient all-reduce (mean) - ONE collective on the optimiser's flat fp32 gradient buffer
    (12.8 MB for the 2-D model).  xGMI is point-to-point (7 links/GPU): a single large message
    lets RCCL drive all links instead of paying per-bucket latency.
 2. all-gather-v of the new negative keys of each class (counts first, then rows padded to the
    max count), concatenated in rank order before the FIFO truncation, so every rank holds a
    bit-identical bank - the `gather_together(keys)` the reference left commented out
    (loss_helper_3d.py:16-17).
 3. (optional) all-reduce of prototype partial sums - not enabled: prototypes are per-rank means,
    like the per-replica statistics under the reference's nn.DataParallel.
Works unchanged with backend "gloo" on CPU tensors for the world_size-2 tests.
"""
import os

import torch
import torch.distributed as td

from . import _contrast


def is_dist():
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def local_rank():
    """Device index of this rank.  ARCO_FORCE_DEVICE pins every rank to one device (single-GPU
    rehearsal of the multi-rank path, together with ARCO_DIST_BACKEND=gloo)."""
    if "ARCO_FORCE_DEVICE" in os.environ:
        return int(os.environ["ARCO_FORCE_DEVICE"])
    return int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise from the torchrun environment (RANK/WORLD_SIZE/MASTER_*).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not td.is_initialized():
        if backend is None:
            backend = os.environ.get("ARCO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank())
        td.init_process_group(backend=backend)
    if is_dist():
        _contrast.key_gather_hook = gather_keys
        _contrast.count_gather_hook = gather_counts
        return td.get_rank(), td.get_world_size()
    return 0, 1


def broadcast_module_states(modules, src=0):
    """Same initial weights/buffers on every rank."""
    if not is_dist():
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            td.broadcast(t.data, src)


def allreduce_grads(optimizer):
    """Mean of the flat gradient buffer over ranks - one RCCL all-reduce."""
    if not is_dist():
        return
    g = optimizer.flat_g
    td.all_reduce(g, op=td.ReduceOp.SUM)
    g.mul_(1.0 / td.get_world_size())
    optimizer._touched.update(range(len(optimizer.params)))


@torch.no_grad()
def gather_keys(keys):
    """All-gather-v of key rows [n_r, D] -> [sum n_r, D], rank order.  Two collectives:
    counts (tiny), then rows padded to the largest count."""
    world = td.get_world_size()
    n = torch.tensor([keys.shape[0]], dtype=torch.int64, device=keys.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    td.all_gather(counts, n)
    counts = [int(c) for c in counts]
    mx = max(counts)
    if mx == 0:
        return keys
    pad = torch.zeros((mx, keys.shape[1]), dtype=keys.dtype, device=keys.device)
    pad[:keys.shape[0]] = keys
    out = [torch.empty_like(pad) for _ in range(world)]
    td.all_gather(out, pad)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def gather_counts(counts):
    """Per-class key counts summed over ranks (host ints in, host ints out): bank lengths and queue
    pointers are then identical on every rank."""
    dev = torch.device("cuda", local_rank()) if td.get_backend() == "nccl" else torch.device("cpu")   # tiny, host-side
    t = torch.tensor([int(c) for c in counts], dtype=torch.int64, device=dev)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]
