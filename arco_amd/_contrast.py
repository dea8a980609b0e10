"""Memory-bank pixel-contrastive loss on MI355X - host orchestration over the C-ABI kernels.

Mirrors compute_contra_memobank_loss / dequeue_and_enqueue of the reference
(loss_helper_3d.py:12-32, 271-513 for 4-D image batches; loss_helper.py:142-162,
442-686 for 5-D volumes - the two differ only in the number of spatial dims, this
implementation is rank-generic).  Same signature, same in-place mutation of
`memobank` / `queue_prtlis`, same return values, same degenerate-batch behaviour.

The work is cut into four stages so a trainer can overlap the host sampler with GPU work
(compute_contra_memobank_loss simply runs them back to back):

  contrast_masks     GPU   per-pixel class masks + counts (mask_codes, scan); async D2H of 3*C counters
  contrast_sample    host  wait for the counters, replay the torch-CPU-generator samplers (bit-exact
                           indices; needs only counts and the post-enqueue bank lengths)
  contrast_enqueue   GPU   stable row lists, prototypes (masked mean of teacher rows), key gather,
                           FIFO-by-truncation bank append (banks stay resident in HBM)
  contrast_infonce   GPU   anchors [n,D] -> normalise -> scores vs the whole bank on the fp32 matrix
                           cores -> multiplicity-weighted softmax-CE -> anchor gradient (second GEMM)
Only the anchor ROWS of `rep` enter the loss, so `rep` is touched through a row gather whose
backward is a row scatter; a trainer may instead feed anchor rows computed lazily
(arco_amd.head) - same values, no dense 496-channel tensor.
"""
import ctypes

import torch

from . import _lib as L
from . import samplers

GROUPED = True      # all classes of the InfoNCE loop per launch (False: the per-class launch sequence; tests compare the two)
NCE_FUSED = int(__import__('os').environ.get('ARCO_NCE_FUSED', '1'))     # round 6: score GEMM with the softmax-CE in its epilogue (0: the staged route of rounds 2-5; tests compare the two)
EPS = 1e-8          # torch.cosine_similarity eps
DELTA_P = 0.3       # current_class_threshold      (loss_helper_3d.py:316)
LOW_RANK, HIGH_RANK = 3, 20                        # (loss_helper_3d.py:318)

# hook for data-parallel training: callable(keys[n,D]) -> keys gathered from all ranks
# (rank order); installed by arco_amd.dist.  None = single process.
key_gather_hook = None
# companion hook: callable(list[int] per-class local counts) -> list[int] global counts
count_gather_hook = None
# callable(keys[n,D], cls, queue_size) -> only the rows that survive the FIFO truncation (rank order)
tail_gather_hook = None
tail_gather_all_hook = None  # the same for all classes at once: (list of key rows per class, queue sizes) -> list (one broadcast per rank)
totals_gather_hook = None    # data parallel: device [3C] counters -> device [world, 3C] (dist.gather_totals_device), queued before the host sync
count_table_hook = None      # data parallel: host table [world][C] of new-key counts -> sums over ranks (dist.set_rank_counts)
proto_reduce_hook = None     # data parallel: (proto [C, D], counts [C]) -> count-weighted mean over ranks (dist.reduce_prototypes)


def _ceil(x, m):
    return (x + m - 1) // m * m


def rows_view(x):
    """[B, D, *spatial] -> ([B*P, D] tensor with unit channel stride, row stride).

    Zero-copy when x is already channels-last (or a channel slice of such a tensor);
    otherwise one channels-last copy is made."""
    y = x.movedim(1, -1)
    D = y.shape[-1]
    try:
        r = y.view(-1, D)
    except RuntimeError:
        r = y.contiguous().view(-1, D)
    keep = r.dtype in (torch.float32, torch.float16)       # f16: the V-Net body's activation storage (ops.ACT_HALF)
    if r.stride(1) != 1 or r.stride(0) % 4 != 0 or r.data_ptr() % 16 != 0 or not keep:
        r = (r if keep else r.to(torch.float32)).contiguous()
    return r, r.stride(0)


@torch.no_grad()
def _append(keys, n_all, queue, queue_ptr, queue_size):
    """queue[0] <- cat(queue[0], all n_all new keys)[-queue_size:], given (at least) the last
    min(n_all, queue_size) of them in `keys`; pointer arithmetic of loss_helper_3d.py:24-30."""
    D = int(keys.shape[1])
    take = min(int(keys.shape[0]), queue_size)
    if take < keys.shape[0]:
        keys = keys[keys.shape[0] - take:]
    old = queue[0].to(device=keys.device, dtype=torch.float32).contiguous()
    len_old = int(old.shape[0])
    out_len = min(len_old + take, queue_size)
    out = torch.empty((out_len, D), device=keys.device, dtype=torch.float32)
    L.call("arco_bank_append", L.ptr(old), len_old, L.ptr(keys.contiguous()), take, queue_size, D, L.ptr(out))
    queue[0] = out
    if len_old + n_all >= queue_size:
        queue_ptr[0] = queue_size
    else:
        queue_ptr[0] = (int(queue_ptr) + n_all) % queue_size


@torch.no_grad()
def dequeue_and_enqueue(keys, queue, queue_ptr, queue_size):
    """Device-resident version of loss_helper_3d.py:12-32.  `queue[0]` is replaced by a GPU tensor;
    `queue_ptr[0]` gets the reference's pointer value.  With the data-parallel hook installed the
    keys of all ranks are concatenated in rank order first (the reference's commented-out
    gather_together, loss_helper_3d.py:16-17)."""
    L.require_gpu(keys)
    keys = keys.detach().to(torch.float32).contiguous()
    n = int(keys.shape[0])
    if key_gather_hook is not None:
        keys = key_gather_hook(keys)
    _append(keys, int(keys.shape[0]), queue, queue_ptr, queue_size)
    return n


class GatherRowsFn(torch.autograd.Function):
    """A[j] = rows(rep)[pix[j]]; backward scatters (duplicates add) into a zero tensor shaped like rep."""

    @staticmethod
    def forward(ctx, rep, pix):
        R, ld = rows_view(rep)
        D = int(rep.shape[1])
        n = int(pix.shape[0])
        A = torch.empty((n, D), dtype=torch.float32, device=rep.device)
        L.call("arco_gather_rows", L.ptr(R), ld, D, None, L.ptr(pix), None, 0, n, L.ptr(A), D)
        ctx.save_for_backward(pix)
        ctx.rep_shape = tuple(rep.shape)
        return A

    @staticmethod
    def backward(ctx, dA):
        (pix,) = ctx.saved_tensors
        B, D = ctx.rep_shape[0], ctx.rep_shape[1]
        sp = ctx.rep_shape[2:]
        n_pix = B
        for s in sp:
            n_pix *= s
        grad = torch.zeros((n_pix, D), device=dA.device, dtype=torch.float32)
        dA = dA.contiguous()
        L.call("arco_scatter_add_rows", L.ptr(dA), D, D, None, L.ptr(pix), int(pix.shape[0]), None, 1.0,
               L.ptr(grad), D)
        return grad.view(B, *sp, D).movedim(-1, 1), None


class _CompactGrad(torch.autograd.Function):
    """loss value with the precomputed d loss / d A attached to the compact anchor matrix A."""

    @staticmethod
    def forward(ctx, A, loss_val, dA):
        ctx.save_for_backward(dA)
        return loss_val.clone()

    @staticmethod
    def backward(ctx, g):
        (dA,) = ctx.saved_tensors
        return dA * g, None, None


class ContrastPlan:
    pass


# ----------------------------------------------------------------------------------------------
# stage 1 (GPU): masks + counts                                    loss_helper_3d.py:341-401
# ----------------------------------------------------------------------------------------------
@torch.no_grad()
def contrast_masks(label_l, label_u, prob_l, prob_u, low_mask, high_mask, delta_n=1.0):
    L.load()
    L.require_gpu(label_l, label_u, prob_l, prob_u, low_mask, high_mask)
    pl = ContrastPlan()
    dev = label_l.device
    C = int(label_l.shape[1])
    n_l, n_u = int(label_l.shape[0]), int(label_u.shape[0])
    P = 1
    for s in label_l.shape[2:]:
        P *= int(s)
    n_pix = (n_l + n_u) * P
    lab_l = label_l.to(torch.int64).contiguous()
    lab_u = label_u.to(torch.int64).contiguous()
    p_l = prob_l.detach().to(torch.float32).contiguous()
    p_u = prob_u.detach().to(torch.float32).contiguous()
    lowm = low_mask.to(torch.float32).contiguous()
    highm = high_mask.to(torch.float32).contiguous()
    assert lowm.numel() == n_pix and highm.numel() == n_pix
    nblocks = (n_pix + 255) // 256
    pl.codes = torch.empty(n_pix, dtype=torch.int64, device=dev)
    counts = torch.empty(3 * C * nblocks, dtype=torch.int32, device=dev)
    pl.offsets = torch.empty(3 * C * nblocks, dtype=torch.int32, device=dev)
    pl.totals = torch.empty(3 * C, dtype=torch.int64, device=dev)
    L.call("arco_mask_codes", L.ptr(lab_l), L.ptr(lab_u), L.ptr(p_l), L.ptr(p_u), L.ptr(lowm), L.ptr(highm),
           n_l, n_u, C, P, DELTA_P, float(delta_n), LOW_RANK, HIGH_RANK, L.ptr(pl.codes), L.ptr(counts),
           L.ptr(pl.offsets), L.ptr(pl.totals))
    pl.totals_host = torch.empty(3 * C, dtype=torch.int64).pin_memory()
    pl.totals_host.copy_(pl.totals, non_blocking=True)
    pl.all_totals_host = None
    if totals_gather_hook is not None:        # every rank's counters, gathered on the device and copied with the local ones
        allt = totals_gather_hook(pl.totals)
        pl.all_totals_host = torch.empty(tuple(allt.shape), dtype=torch.int64).pin_memory()
        pl.all_totals_host.copy_(allt, non_blocking=True)
    pl.ready = torch.cuda.Event()
    pl.ready.record()
    pl.C, pl.P, pl.n_pix, pl.dev = C, P, n_pix, dev
    pl.spatial = tuple(int(s) for s in label_l.shape[2:])
    pl.n_img = n_l + n_u
    return pl


# ----------------------------------------------------------------------------------------------
# stage 2 (host): counters -> bank bookkeeping -> sampler replay      loss_helper_3d.py:413-476
# ----------------------------------------------------------------------------------------------
_pin_ring = {"bufs": [], "i": 0}


def _pinned_i64(n, depth=3):
    """Rotating pinned staging buffers for the sampled indices (a buffer is reused `depth` steps later, long after
    its asynchronous upload has completed)."""
    r = _pin_ring
    if len(r["bufs"]) < depth:
        r["bufs"].append(torch.empty(max(n, 1 << 20), dtype=torch.int64).pin_memory())
        return r["bufs"][-1]
    r["i"] = (r["i"] + 1) % depth
    if r["bufs"][r["i"]].numel() < n:
        r["bufs"][r["i"]] = torch.empty(n, dtype=torch.int64).pin_memory()
    return r["bufs"][r["i"]]


@torch.no_grad()
def contrast_sample(pl, memobank, queue_size, func='asmc', num_queries=256, num_negatives=512, _trace=None):
    """stage 2 = contrast_counts (host sync + bookkeeping) then contrast_draw (sampler replay + upload).  The
    trainers call the two halves separately and queue sample-independent GPU work in between."""
    contrast_counts(pl, memobank, queue_size, num_queries, num_negatives)
    return contrast_draw(pl, func, _trace)


SAMPLER_PREGEN = int(__import__('os').environ.get('ARCO_SAMPLER_PREGEN', '1'))    # A/B switch (0: no pregenerated generator blocks)


@torch.no_grad()
def contrast_counts(pl, memobank, queue_size, num_queries=256, num_negatives=512):
    if SAMPLER_PREGEN:
        # While this thread waits for the counters: the CPU generator's next state blocks for the draws the samplers are
        # about to make (anchor calls: <= one draw per candidate pixel; negative calls: ~2.03 draws per index), computed
        # by a native worker thread - the replay in contrast_draw then finds them ready (samplers.pregen).
        samplers.pregen(min(pl.n_pix + pl.C * (int(2.1 * num_queries * num_negatives) + 8192) + 65536, 8 << 20))
    pl.ready.synchronize()                                  # the one device->host dependency of the loss
    C = pl.C
    tot = pl.totals_host.tolist()
    pl.n_lv, pl.n_anchor, pl.n_neg = tot[:C], tot[C:2 * C], tot[2 * C:]
    if getattr(pl, "all_totals_host", None) is not None and count_table_hook is not None:
        n_neg_all = count_table_hook([row[2 * C:] for row in pl.all_totals_host.tolist()])      # no collective here
    else:
        n_neg_all = pl.n_neg if count_gather_hook is None else count_gather_hook(pl.n_neg)
    pl.n_neg_all = n_neg_all
    # bank lengths after this step's enqueue (loss_helper_3d.py:23-26)
    pl.bank_len = [min(int(memobank[c][0].shape[0]) + int(n_neg_all[c]), int(queue_size[c])) for c in range(C)]
    pl.valid_classes = [c for c in range(C) if pl.n_lv[c] > 0]        # :413-415
    pl.valid_seg = len(pl.valid_classes)
    pl.Q, pl.Nn = int(num_queries), int(num_negatives)
    return pl


@torch.no_grad()
def contrast_draw(pl, func='asmc', _trace=None, defer=False):
    """defer=True (the trainers): the native replay returns once the CPU generator holds its final state; the negative
    draws finish in worker threads and the index upload happens in contrast_draw_finish (called by contrast_anchor_pix /
    contrast_infonce) - the caller draws the equivariance warp and queues that pass in between."""
    if func == 'asmc':                                                  # :327-338
        draw, q_arg, n_arg = samplers.grid_as_monte_carlo_sample, pl.Q, pl.Q * pl.Nn
    elif func == 'smc':
        draw, q_arg, n_arg = samplers.grid_monte_carlo_sample, pl.Q, pl.Q * pl.Nn
    else:
        draw, q_arg, n_arg = torch.randint, (pl.Q,), (pl.Q * pl.Nn,)
    pl.entries = []
    pl._pending_upload = None
    pl.idx_all, pl.idx_stride = None, 0       # packed [anchors(Q) | negatives(Q*Nn)] per entry, when drawn in one native call
    if pl.valid_seg > 1:
        ks = [k for k in range(pl.valid_seg)                               # :435-476, k = LOOP COUNTER
              if not (pl.n_anchor[k] == 0 or pl.bank_len[pl.valid_classes[k]] == 0)]
        if func in ('asmc', 'smc') and ks:
            # all 2*len(ks) sampler calls of the step in ONE native call (generator order: anchors k, negatives k,
            # ...; the big negative draws run in worker threads), written back to back into one pinned buffer
            # and uploaded with one copy
            jobs = []
            for k in ks:
                jobs += [(int(pl.n_anchor[k]), pl.Q), (int(pl.bank_len[pl.valid_classes[k]]), pl.Q * pl.Nn)]
            total = sum(sh for _, sh in jobs)
            host = _pinned_i64(total)
            defer = bool(defer) and _trace is None
            views = samplers.grid_sample_many(jobs, func == 'asmc', out=host, defer=defer)
            dev_all = torch.empty(total, dtype=torch.int64, device=pl.dev)
            pl._host_idx = host                                           # keep the staging buffer alive until used
            pl._pending_upload = (dev_all, total)
            if not defer:
                contrast_draw_finish(pl)
            pl.idx_all, pl.idx_stride = dev_all, pl.Q + pl.Q * pl.Nn
            off = 0
            for i, k in enumerate(ks):
                a_dev = dev_all[off:off + pl.Q]; off += pl.Q
                n_dev = dev_all[off:off + pl.Q * pl.Nn]; off += pl.Q * pl.Nn
                pl.entries.append((k, pl.valid_classes[k], a_dev, n_dev))
                if _trace is not None:
                    _trace.setdefault("anchor_idx", []).append(views[2 * i].clone())
                    _trace.setdefault("neg_idx", []).append(views[2 * i + 1].clone())
        else:
            for k in ks:
                vc = pl.valid_classes[k]
                a_idx = draw(int(pl.n_anchor[k]), q_arg)
                n_idx = draw(int(pl.bank_len[vc]), n_arg)
                pl.entries.append((k, vc, a_idx.pin_memory().to(pl.dev, non_blocking=True),
                                   n_idx.pin_memory().to(pl.dev, non_blocking=True)))
                if _trace is not None:
                    _trace.setdefault("anchor_idx", []).append(a_idx)
                    _trace.setdefault("neg_idx", []).append(n_idx)
    return pl


def contrast_draw_finish(pl):
    """Collect a deferred contrast_draw: wait for the worker calls, upload the indices (one copy).  Idempotent."""
    pend = getattr(pl, "_pending_upload", None)
    if pend is not None:
        samplers.finish_many()
        dev_all, total = pend
        dev_all.copy_(pl._host_idx[:total], non_blocking=True)
        pl._pending_upload = None


# ----------------------------------------------------------------------------------------------
# stage 3 (GPU): row lists, prototypes, key enqueue                 loss_helper_3d.py:376-411
# ----------------------------------------------------------------------------------------------
@torch.no_grad()
def contrast_lists_protos(pl, rep_teacher=None, lazy_teacher=None):
    """The part of stage 3 that needs neither the counters on the host nor the samples: per-class row lists
    (arco_compact_rows: offsets are on the device) and the class prototypes (class masks and totals are on the device).  The
    trainers queue it BEFORE they block on the counters; contrast_enqueue runs it itself when nobody has."""
    C, n_pix, dev = pl.C, pl.n_pix, pl.dev
    pl.lists = torch.empty((2 * C, n_pix), dtype=torch.int32, device=dev)
    L.call("arco_compact_rows", L.ptr(pl.codes), n_pix, C, L.ptr(pl.offsets), L.ptr(pl.lists))
    if lazy_teacher is None:
        L.require_gpu(rep_teacher)
        T, ldt = rows_view(rep_teacher.detach())
        D = int(rep_teacher.shape[1])
        if D % 4 != 0:
            raise RuntimeError("arco_amd: feature dim must be a multiple of 4")
        pl.proto = torch.empty((C, D), dtype=torch.float32, device=dev)
        ws = torch.empty(L.query("arco_proto_ws_floats", n_pix, C, D), dtype=torch.float32, device=dev)
        L.call("arco_masked_proto", L.ptr(T), ldt, L.ptr(pl.codes), n_pix, C, D, L.ptr(pl.totals), L.ptr(ws),
               L.ptr(pl.proto))
    else:
        pl.proto = lazy_teacher.prototypes(pl)
    if proto_reduce_hook is not None:        # class means over the GLOBAL batch (one [C, D+1] all-reduce)
        pl.proto = proto_reduce_hook(pl.proto, pl.totals[:C])
    pl.D = int(pl.proto.shape[1])
    pl._lists_protos_done = True


@torch.no_grad()
def contrast_enqueue(pl, rep_teacher, memobank, queue_prtlis, queue_size, _trace=None, lazy_teacher=None,
                     defer_anchor_pix=False):
    """Row lists, prototypes and key enqueue.  Teacher rows come either from the dense `rep_teacher`
    [B,D,*spatial] (public API) or lazily from a `lazy_teacher` object (arco_amd.head): the FeatureExtractor
    is linear, so prototype_c = W_fea4 . mean_c(fea4 input) with the class mask pushed through the
    bi/trilinear adjoint to the low-res level, and only the <= queue_size key rows per class are evaluated.
    Same values up to fp32 re-association of the mean."""
    C, n_pix, dev = pl.C, pl.n_pix, pl.dev
    if not getattr(pl, "_lists_protos_done", False):
        contrast_lists_protos(pl, rep_teacher, lazy_teacher)
    D = pl.D
    takes = [min(int(pl.n_neg[c]), int(queue_size[c])) for c in range(C)]   # only these can survive the truncation
    if lazy_teacher is None:
        T, ldt = rows_view(rep_teacher.detach())
        key_rows = []
        for c in range(C):
            n, take = int(pl.n_neg[c]), takes[c]
            keys = torch.empty((take, D), dtype=torch.float32, device=dev)
            L.call("arco_gather_rows", L.ptr(T), ldt, D, None, None, L.ptr(pl.lists[C + c]), n - take, take,
                   L.ptr(keys), D)
            key_rows.append(keys)
    else:
        # lazy teacher (arco_amd.head.LazyTeacher2D / LazyTeacher3D): only the key rows that can survive the truncation
        pix = torch.cat([pl.lists[C + c][int(pl.n_neg[c]) - takes[c]:int(pl.n_neg[c])] for c in range(C)]).to(torch.int64)
        allk = lazy_teacher.rows(pix) if int(pix.shape[0]) > 0 else torch.empty((0, D), dtype=torch.float32, device=dev)
        key_rows, off = [], 0
        for c in range(C):
            key_rows.append(allk[off:off + takes[c]])
            off += takes[c]
    pl.new_keys = []
    if tail_gather_all_hook is not None:     # data parallel: every class's surviving rows in one broadcast per contributing rank
        key_rows = tail_gather_all_hook([k.contiguous() for k in key_rows], [int(q) for q in queue_size])
    for c in range(C):
        keys = key_rows[c]
        if tail_gather_all_hook is not None:
            pass
        elif tail_gather_hook is not None:
            keys = tail_gather_hook(keys.contiguous(), c, int(queue_size[c]))
        elif key_gather_hook is not None:
            keys = key_gather_hook(keys.contiguous())
        _append(keys, int(pl.n_neg_all[c]), memobank[c], queue_prtlis[c], int(queue_size[c]))
        assert int(memobank[c][0].shape[0]) == pl.bank_len[c]
        pl.new_keys.append(int(pl.n_neg[c]))
    if _trace is not None:
        _trace["lists"], _trace["totals"], _trace["proto"] = pl.lists, pl.totals_host.tolist(), pl.proto
    if not defer_anchor_pix:                 # needs the sampled indices (contrast_draw); everything above does not
        contrast_anchor_pix(pl)
    return pl


def _pack_indices(pl):
    """Sampled indices of all entries in one int64 buffer, [anchors(Q) | negatives(Q*Nn)] per entry (the native draw already
    produces this layout; the torch.randint / trace-replay paths are packed here)."""
    if getattr(pl, "idx_all", None) is None:
        pl.idx_all = torch.cat([torch.cat((a_dev.reshape(-1), n_dev.reshape(-1))) for (k, vc, a_dev, n_dev) in pl.entries]).to(torch.int64)
        pl.idx_stride = pl.Q + pl.Q * pl.Nn


@torch.no_grad()
def contrast_anchor_pix(pl):
    """global pixel id of every sampled anchor, entries concatenated"""
    contrast_draw_finish(pl)
    if pl.entries:
        _pack_indices(pl)
        E = len(pl.entries)
        pl.anchor_pix = torch.empty(E * pl.Q, dtype=torch.int64, device=pl.dev)
        ks = (ctypes.c_int * E)(*[int(k) for (k, vc, a_dev, n_dev) in pl.entries])
        L.call("arco_anchor_pix", L.ptr(pl.lists), pl.n_pix, ks, E, L.ptr(pl.idx_all), pl.idx_stride, pl.Q, L.ptr(pl.anchor_pix))
    else:
        pl.anchor_pix = torch.empty(0, dtype=torch.int64, device=pl.dev)
    return pl


# ----------------------------------------------------------------------------------------------
# stage 4 (GPU): InfoNCE over the compact anchor matrix             loss_helper_3d.py:478-513
# ----------------------------------------------------------------------------------------------
def _infonce_grouped(pl, A_all, memobank, temp, need_grad):
    """All entries of the per-class loop (loss_helper_3d.py:435-509) in one launch per stage: normalise anchors /
    prototypes / banks, batched score GEMM A_c . Bank_c^T on the fp32 matrix cores, fused multiplicity + softmax-CE,
    batched split-K anchor-gradient GEMM, gradient through the cosine normalisation.  9 launches for any number of classes."""
    dev, D, Q, Nn, C = pl.dev, pl.D, pl.Q, pl.Nn, pl.C
    E = len(pl.entries)
    Dp = _ceil(D, 16)
    _pack_indices(pl)
    banks = [memobank[vc][0] for (k, vc, a_dev, n_dev) in pl.entries]
    lens = [int(b.shape[0]) for b in banks]
    Lp = _ceil(max(lens), 16)
    bank_ptrs = (ctypes.c_void_p * E)(*[b.data_ptr() for b in banks])
    lens_c = (ctypes.c_int * E)(*lens)
    prow = (ctypes.c_int * E)(*[int(k) for (k, vc, a_dev, n_dev) in pl.entries])      # positive = prototype of LOOP COUNTER k (:480-486)
    A_det = A_all.detach().contiguous()
    n = E * Q
    An = torch.empty((n, Dp), dtype=torch.float32, device=dev)
    invA = torch.empty(n, dtype=torch.float32, device=dev)
    if NCE_FUSED and D % 4 == 0 and temp >= 0.05 and all(b.is_contiguous() and b.data_ptr() % 16 == 0 for b in banks):
        # Round 6: the score GEMM with the softmax-CE in its epilogue (include/arco_hip.h, arco_nce_score) - five launches, no score
        # matrix, no normalised copy of the banks.  (temp < 0.05: exp((cos - 1) / T) may underflow for every negative of a row;
        # the staged route below shifts by the row's actual maximum)
        proto = pl.proto.contiguous()
        nP = int(proto.shape[0])
        Pn = torch.empty((nP, Dp), dtype=torch.float32, device=dev)
        M = torch.empty((n, Lp), dtype=torch.int16, device=dev)
        L.call("arco_nce_prep", L.ptr(A_det), n, L.ptr(proto), nP, D, Dp, EPS, L.ptr(An), L.ptr(invA), L.ptr(Pn), lens_c, E,
               L.ptr(pl.idx_all), Q, pl.idx_stride, Q, Nn, Lp, L.ptr(M))
        n_lt = int(L.query("arco_nce_score_ltiles", Lp))
        Wu = torch.empty((E, Q, Lp), dtype=torch.float32, device=dev)
        Zp = torch.empty((n, n_lt), dtype=torch.float32, device=dev)
        Bt = torch.empty((E, Dp, Lp), dtype=torch.float32, device=dev) if need_grad else None
        pos = torch.empty(n, dtype=torch.float32, device=dev)
        L.call("arco_nce_score", L.ptr(An), Dp, D, bank_ptrs, lens_c, prow, E, Lp, Q, L.ptr(M), L.ptr(Pn), float(temp), EPS, L.ptr(Wu),
               L.ptr(Zp), L.ptr(pos), L.ptr(Bt))
        gpos = torch.empty(n, dtype=torch.float32, device=dev)
        gscale = torch.empty(n, dtype=torch.float32, device=dev)
        loss_q = torch.empty(n, dtype=torch.float32, device=dev)
        loss_acc = torch.empty(1, dtype=torch.float32, device=dev)
        L.call("arco_nce_finish", L.ptr(pos), n, L.ptr(Zp), Lp, float(temp), 1.0 / (Q * pl.valid_seg),
               L.ptr(gpos), L.ptr(gscale), L.ptr(loss_q), L.ptr(loss_acc))
        dA_all = None
        if need_grad:
            splits = max(1, min(16, Lp // 256))
            ws = torch.empty((E, splits, Q, Dp), dtype=torch.float32, device=dev) if splits > 1 else None
            G = torch.empty((n, Dp), dtype=torch.float32, device=dev)
            L.call("arco_gemm_batched", L.ptr(Wu), Lp, Lp, L.ptr(Bt), Dp, L.ptr(G), Dp, Q, E, Q * Lp, Dp * Lp, Q * Dp, splits, L.ptr(ws))
            dA_all = torch.empty((n, D), dtype=torch.float32, device=dev)
            L.call("arco_nce_anchor_grad_scaled", L.ptr(G), L.ptr(An), L.ptr(Pn), prow, E, L.ptr(gpos), L.ptr(invA), L.ptr(gscale), Q, D, Dp,
                   EPS, 1.0 / (Q * pl.valid_seg), L.ptr(dA_all), D)
        return loss_acc[0], dA_all
    L.call("arco_normalize_rows_pad", L.ptr(A_det), D, n, D, Dp, EPS, L.ptr(An), Dp, L.ptr(invA))
    proto = pl.proto.contiguous()
    Pn = torch.empty((int(proto.shape[0]), Dp), dtype=torch.float32, device=dev)
    L.call("arco_normalize_rows_pad", L.ptr(proto), D, int(proto.shape[0]), D, Dp, EPS, L.ptr(Pn), Dp, None)
    Bn = torch.empty((E, Lp, Dp), dtype=torch.float32, device=dev)
    Bt = torch.empty((E, Dp, Lp), dtype=torch.float32, device=dev) if need_grad else None
    L.call("arco_nce_normalize_banks", bank_ptrs, lens_c, E, D, Dp, Lp, EPS, L.ptr(Bn), L.ptr(Bt))
    S = torch.empty((E, Q, Lp), dtype=torch.float32, device=dev)
    L.call("arco_gemm_batched", L.ptr(An), Dp, Dp, L.ptr(Bn), Lp, L.ptr(S), Lp, Q, E, Q * Dp, Lp * Dp, Q * Lp, 1, None)
    W = torch.empty((E, Q, Lp), dtype=torch.float32, device=dev) if need_grad else None
    gpos = torch.empty(n, dtype=torch.float32, device=dev)
    loss_q = torch.empty(n, dtype=torch.float32, device=dev)
    L.call("arco_nce_fused", L.ptr(S), Lp, lens_c, prow, E, L.ptr(pl.idx_all), Q, pl.idx_stride, Q, Nn, L.ptr(An), L.ptr(Pn), Dp,
           float(temp), L.ptr(W), L.ptr(gpos), L.ptr(loss_q))
    loss_acc = torch.empty(1, dtype=torch.float32, device=dev)
    L.call("arco_sum_scale", L.ptr(loss_q), n, 1.0 / (Q * pl.valid_seg), L.ptr(loss_acc), 0)
    dA_all = None
    if need_grad:
        splits = max(1, min(16, Lp // 256))               # Q x D outputs per class, K = bank length: split-K fills the GPU
        ws = torch.empty((E, splits, Q, Dp), dtype=torch.float32, device=dev) if splits > 1 else None
        G = torch.empty((n, Dp), dtype=torch.float32, device=dev)
        L.call("arco_gemm_batched", L.ptr(W), Lp, Lp, L.ptr(Bt), Dp, L.ptr(G), Dp, Q, E, Q * Lp, Dp * Lp, Q * Dp, splits, L.ptr(ws))
        dA_all = torch.empty((n, D), dtype=torch.float32, device=dev)
        L.call("arco_nce_anchor_grad", L.ptr(G), L.ptr(An), L.ptr(Pn), prow, E, L.ptr(gpos), L.ptr(invA), Q, D, Dp, EPS,
               1.0 / (Q * pl.valid_seg), L.ptr(dA_all), D)
    return loss_acc[0], dA_all


def contrast_infonce(pl, A_all, memobank, temp=0.5, momentum_prototype=None, i_iter=0):
    """A_all [len(entries)*Q, D]: the sampled anchor rows (student), entries in plan order.
    Returns (loss, prototype-or-None).  The gradient w.r.t. A_all is computed here (it only needs
    forward quantities) and attached through _CompactGrad."""
    contrast_draw_finish(pl)
    dev, D, Q, Nn, C = pl.dev, pl.D, pl.Q, pl.Nn, pl.C
    Dp = _ceil(D, 16)
    valid_seg = pl.valid_seg
    need_grad = A_all.requires_grad and torch.is_grad_enabled()
    if (momentum_prototype is None and pl.entries and GROUPED and Nn < 65536
            and max(int(memobank[vc][0].shape[0]) for (k, vc, a_, n_) in pl.entries) <= L.query("arco_nce_max_len")):
        loss, dA_all = _infonce_grouped(pl, A_all, memobank, temp, need_grad)
        if need_grad:
            loss = _CompactGrad.apply(A_all, loss, dA_all)
        return loss, None
    prototype = torch.zeros((C, Q, 1, D), device=dev) if momentum_prototype is not None else None
    loss_acc = torch.zeros(1, dtype=torch.float32, device=dev)
    dA_all = torch.zeros((A_all.shape[0], D), dtype=torch.float32, device=dev) if need_grad else None
    A_det = A_all.detach().contiguous()
    norm_cache = {}
    for e, (k, vc, a_dev, n_dev) in enumerate(pl.entries):
        bank = memobank[vc][0]
        Lb = int(bank.shape[0])
        Lp = _ceil(Lb, 16)
        A = A_det[e * Q:(e + 1) * Q]
        An = torch.zeros((Q, Dp), dtype=torch.float32, device=dev)
        invA = torch.empty(Q, dtype=torch.float32, device=dev)
        L.call("arco_normalize_rows", L.ptr(A), D, Q, D, EPS, L.ptr(An), Dp, None, 0, L.ptr(invA))
        if vc not in norm_cache:                                         # bank normalised once per class
            Bn = torch.zeros((Lp, Dp), dtype=torch.float32, device=dev)
            Bt = torch.zeros((Dp, Lp), dtype=torch.float32, device=dev) if need_grad else None
            L.call("arco_normalize_rows", L.ptr(bank), D, Lb, D, EPS, L.ptr(Bn), Dp, L.ptr(Bt), Lp, None)
            norm_cache[vc] = (Bn, Bt)
        Bn, Bt = norm_cache[vc]
        pos = pl.proto[k].view(1, D)              # prototype of LOOP-COUNTER class k    (:480-486)
        if momentum_prototype is not None:                               # :488-497
            pos = pos.view(1, 1, D).repeat(Q, 1, 1)
            if not bool((momentum_prototype == 0).all()):
                decay = min(1 - 1 / i_iter, 0.999)
                pos = (1 - decay) * pos + decay * momentum_prototype[vc].to(dev)
            prototype[vc] = pos.clone()
            pos = pos.view(Q, D)
        pos = pos.contiguous()
        nP = int(pos.shape[0])
        Pn = torch.zeros((nP, Dp), dtype=torch.float32, device=dev)
        L.call("arco_normalize_rows", L.ptr(pos), D, nP, D, EPS, L.ptr(Pn), Dp, None, 0, None)
        ldp = Dp if nP > 1 else 0
        # scores vs the whole bank on the matrix cores, multiplicity-weighted softmax-CE     (:503-509)
        S = torch.empty((Q, Lp), dtype=torch.float32, device=dev)
        L.call("arco_conv_fwd", L.ptr(An), Dp, Dp, L.ptr(Bn), Lb, L.ptr(S), Lp, None, None, 0, None, None,
               1, 1, 1, Q)
        M = torch.empty((Q, Lp), dtype=torch.int32, device=dev)
        L.call("arco_neg_multiplicity", L.ptr(n_dev), Q, Nn, Lb, Lp, L.ptr(M))
        W = torch.zeros((Q, Lp), dtype=torch.float32, device=dev)
        gpos = torch.empty(Q, dtype=torch.float32, device=dev)
        loss_q = torch.empty(Q, dtype=torch.float32, device=dev)
        L.call("arco_infonce_fwd", L.ptr(S), Lp, L.ptr(M), Lb, L.ptr(An), L.ptr(Pn), ldp, Q, Dp, float(temp),
               L.ptr(W), L.ptr(gpos), L.ptr(loss_q))
        L.call("arco_sum_scale", L.ptr(loss_q), Q, 1.0 / (Q * valid_seg), L.ptr(loss_acc), 1)
        if need_grad:
            G = torch.empty((Q, Dp), dtype=torch.float32, device=dev)
            splits = max(1, min(16, Lp // 256))               # 256 x 496 outputs, K = bank length: split-K fills the GPU
            ws = torch.empty((splits, Q, Dp), dtype=torch.float32, device=dev)
            L.call("arco_gemm_splitk", L.ptr(W), Lp, Lp, L.ptr(Bt), Dp, L.ptr(G), Dp, Q, splits, L.ptr(ws))
            dA = torch.empty((Q, Dp), dtype=torch.float32, device=dev)
            L.call("arco_infonce_anchor_grad", L.ptr(G), L.ptr(An), L.ptr(Pn), ldp, L.ptr(gpos), L.ptr(invA),
                   Q, Dp, EPS, 1.0 / (Q * valid_seg), L.ptr(dA))
            dA_all[e * Q:(e + 1) * Q] = dA[:, :D]
    loss = loss_acc[0]
    if need_grad:
        loss = _CompactGrad.apply(A_all, loss, dA_all)
    return loss, prototype


def compute_contra_memobank_loss(rep, label_l, label_u, prob_l, prob_u, low_mask, high_mask, memobank,
                                 queue_prtlis, queue_size, rep_teacher, momentum_prototype=None, i_iter=0,
                                 delta_n=1.0, func='asmc', num_queries=256, num_negatives=512, temp=0.5,
                                 _trace=None):
    L.load()
    L.require_gpu(rep, rep_teacher)
    pl = contrast_masks(label_l, label_u, prob_l, prob_u, low_mask, high_mask, delta_n)
    contrast_sample(pl, memobank, queue_size, func, num_queries, num_negatives, _trace)
    contrast_enqueue(pl, rep_teacher, memobank, queue_prtlis, queue_size, _trace)
    if pl.valid_seg <= 1:                                               # :417-424
        zero = rep.sum() * 0.0
        return (pl.new_keys, zero) if momentum_prototype is None else (momentum_prototype, pl.new_keys, zero)
    if pl.entries:
        A_all = GatherRowsFn.apply(rep, pl.anchor_pix)                  # :455-457 (grad flows to rep rows)
        loss, prototype = contrast_infonce(pl, A_all, memobank, temp, momentum_prototype, i_iter)
    else:
        loss = rep.sum() * 0.0
        prototype = torch.zeros((pl.C, int(num_queries), 1, int(rep.shape[1])), device=rep.device) \
            if momentum_prototype is not None else None
    if momentum_prototype is None:
        return pl.new_keys, loss
    return prototype, pl.new_keys, loss
