"""Memory-bank pixel-contrastive loss on MI355X - host orchestration over the C-ABI kernels.

Mirrors compute_contra_memobank_loss / dequeue_and_enqueue of the reference
(loss_helper_3d.py:12-32, 271-513 for 4-D image batches; loss_helper.py:142-162,
442-686 for 5-D volumes - the two differ only in the number of spatial dims, this
implementation is rank-generic).  Same signature, same in-place mutation of
`memobank` / `queue_prtlis`, same return values, same degenerate-batch behaviour.

Device pipeline (all on the caller's current HIP stream):
  mask_codes -> scan -> compact_rows   per-pixel class masks, counts, stable row lists
  masked_proto                         prototypes = masked mean of teacher rows
  gather_rows + bank_append            FIFO-by-truncation banks, resident in HBM
  [host]  samplers.*                   torch-CPU-generator index replay (bit-exact)
  gather_rows / normalize_rows / conv_fwd(1x1 MFMA GEMM) / neg_multiplicity /
  infonce_fwd / conv_fwd / infonce_anchor_grad      InfoNCE value + anchor gradient
One device->host copy per call (the 3*C counters), none per class.
"""
import torch

from . import _lib as L
from . import samplers

EPS = 1e-8          # torch.cosine_similarity eps
DELTA_P = 0.3       # current_class_threshold      (loss_helper_3d.py:316)
LOW_RANK, HIGH_RANK = 3, 20                        # (loss_helper_3d.py:318)

# hook for data-parallel training: callable(keys[n,D], class_id) -> keys gathered from all ranks
# (rank order); installed by arco_amd.dist.  None = single process.
key_gather_hook = None


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
    if r.stride(1) != 1 or r.stride(0) % 4 != 0 or r.data_ptr() % 16 != 0 or r.dtype != torch.float32:
        r = r.to(torch.float32).contiguous()
    return r, r.stride(0)


@torch.no_grad()
def dequeue_and_enqueue(keys, queue, queue_ptr, queue_size):
    """Device-resident version of loss_helper_3d.py:12-32: queue[0] <- cat(queue[0], keys)[-queue_size:].
    `queue[0]` is replaced by a GPU tensor; `queue_ptr[0]` gets the reference's pointer value."""
    L.require_gpu(keys)
    keys = keys.detach().to(torch.float32).contiguous()
    n, D = int(keys.shape[0]), int(keys.shape[1])
    old = queue[0].to(device=keys.device, dtype=torch.float32).contiguous()
    if key_gather_hook is not None:
        keys = key_gather_hook(keys)
        n_all = int(keys.shape[0])
    else:
        n_all = n
    take = min(n_all, queue_size)
    if take < n_all:
        keys = keys[n_all - take:]
    len_old = int(old.shape[0])
    out_len = min(len_old + take, queue_size)
    out = torch.empty((out_len, D), device=keys.device, dtype=torch.float32)
    L.call("arco_bank_append", L.ptr(old), len_old, L.ptr(keys), take, queue_size, D, L.ptr(out))
    queue[0] = out
    if len_old + n_all >= queue_size:
        queue_ptr[0] = queue_size
    else:
        queue_ptr[0] = (int(queue_ptr) + n_all) % queue_size
    return n


class _AnchorGrad(torch.autograd.Function):
    """Attaches the precomputed anchor-row gradient of the loss to `rep`."""

    @staticmethod
    def forward(ctx, rep, loss_val, lists, pieces):
        ctx.lists, ctx.pieces = lists, pieces
        ctx.rep_shape = rep.shape
        return loss_val.clone()

    @staticmethod
    def backward(ctx, g):
        B, D = ctx.rep_shape[0], ctx.rep_shape[1]
        sp = tuple(ctx.rep_shape[2:])
        n_pix = B
        for s in sp:
            n_pix *= s
        grad = torch.zeros((n_pix, D), device=g.device, dtype=torch.float32)
        gs = g.to(torch.float32).contiguous()
        for (cls, idx, dA, ld) in ctx.pieces:
            lst = ctx.lists[cls]
            L.call("arco_scatter_add_rows", L.ptr(dA), ld, D, L.ptr(lst), L.ptr(idx), int(idx.shape[0]),
                   L.ptr(gs), 1.0, L.ptr(grad), D)
        return grad.view(B, *sp, D).movedim(-1, 1), None, None, None


def compute_contra_memobank_loss(rep, label_l, label_u, prob_l, prob_u, low_mask, high_mask, memobank,
                                 queue_prtlis, queue_size, rep_teacher, momentum_prototype=None, i_iter=0,
                                 delta_n=1.0, func='asmc', num_queries=256, num_negatives=512, temp=0.5,
                                 _trace=None):
    L.load()
    L.require_gpu(rep, label_l, label_u, prob_l, prob_u, low_mask, high_mask, rep_teacher)
    dev = rep.device
    D = int(rep.shape[1])
    C = int(label_l.shape[1])
    n_l, n_u = int(label_l.shape[0]), int(label_u.shape[0])
    P = 1
    for s in label_l.shape[2:]:
        P *= int(s)
    n_pix = (n_l + n_u) * P
    if D % 4 != 0:
        raise RuntimeError("arco_amd: feature dim must be a multiple of 4")

    if func == 'asmc':                                                  # loss_helper_3d.py:327-338
        draw, q_arg, n_arg = samplers.grid_as_monte_carlo_sample, num_queries, num_queries * num_negatives
    elif func == 'smc':
        draw, q_arg, n_arg = samplers.grid_monte_carlo_sample, num_queries, num_queries * num_negatives
    else:
        draw, q_arg, n_arg = torch.randint, (num_queries,), (num_queries * num_negatives,)

    R, ldr = rows_view(rep)
    T, ldt = rows_view(rep_teacher.detach())
    lab_l = label_l.to(torch.int64).contiguous()
    lab_u = label_u.to(torch.int64).contiguous()
    pl = prob_l.detach().to(torch.float32).contiguous()
    pu = prob_u.detach().to(torch.float32).contiguous()
    lowm = low_mask.to(torch.float32).contiguous()
    highm = high_mask.to(torch.float32).contiguous()
    assert lowm.numel() == n_pix and highm.numel() == n_pix

    # ---- L1/L2: masks, counts, stable row lists, prototypes
    nblocks = (n_pix + 255) // 256
    codes = torch.empty(n_pix, dtype=torch.int64, device=dev)
    counts = torch.empty(3 * C * nblocks, dtype=torch.int32, device=dev)
    offsets = torch.empty(3 * C * nblocks, dtype=torch.int32, device=dev)
    totals = torch.empty(3 * C, dtype=torch.int64, device=dev)
    L.call("arco_mask_codes", L.ptr(lab_l), L.ptr(lab_u), L.ptr(pl), L.ptr(pu), L.ptr(lowm), L.ptr(highm),
           n_l, n_u, C, P, DELTA_P, float(delta_n), LOW_RANK, HIGH_RANK, L.ptr(codes), L.ptr(counts),
           L.ptr(offsets), L.ptr(totals))
    lists = torch.empty((2 * C, n_pix), dtype=torch.int32, device=dev)
    L.call("arco_compact_rows", L.ptr(codes), n_pix, C, L.ptr(offsets), L.ptr(lists))
    proto = torch.empty((C, D), dtype=torch.float32, device=dev)
    ws = torch.empty(L.query("arco_proto_ws_floats", n_pix, C, D), dtype=torch.float32, device=dev)
    L.call("arco_masked_proto", L.ptr(T), ldt, L.ptr(codes), n_pix, C, D, L.ptr(totals), L.ptr(ws), L.ptr(proto))
    tot = totals.cpu().tolist()                                         # the one D2H sync of the call
    n_lv, n_anchor, n_neg = tot[:C], tot[C:2 * C], tot[2 * C:]

    # ---- L3: enqueue the new negative keys of every class (loss_helper_3d.py:403-411)
    new_keys, seg_num, valid_classes = [], [], []
    for c in range(C):
        n = n_neg[c]
        take = min(n, queue_size[c]) if key_gather_hook is None else n
        keys = torch.empty((take, D), dtype=torch.float32, device=dev)
        L.call("arco_gather_rows", L.ptr(T), ldt, D, None, None, L.ptr(lists[C + c]), n - take, take,
               L.ptr(keys), D)
        if take < n:        # only the last queue_size rows can survive the truncation
            pad_n = n
            new_keys.append(_enqueue_counted(keys, pad_n, memobank[c], queue_prtlis[c], queue_size[c]))
        else:
            new_keys.append(dequeue_and_enqueue(keys, memobank[c], queue_prtlis[c], queue_size[c]))
        if n_lv[c] > 0:                                                 # :413-415
            seg_num.append(int(n_lv[c]))
            valid_classes.append(c)
    if _trace is not None:
        _trace["lists"], _trace["totals"], _trace["proto"] = lists, tot, proto

    if len(seg_num) <= 1:                                               # :417-424
        zero = rep.sum() * 0.0
        return (new_keys, zero) if momentum_prototype is None else (momentum_prototype, new_keys, zero)

    valid_seg = len(seg_num)
    Q, Nn = int(num_queries), int(num_negatives)
    Dp = _ceil(D, 16)
    prototype = None
    if momentum_prototype is not None:
        prototype = torch.zeros((C, Q, 1, D), device=dev)
    loss_acc = torch.zeros(1, dtype=torch.float32, device=dev)
    pieces = []
    need_grad = rep.requires_grad and torch.is_grad_enabled()
    norm_cache = {}

    # ---- L4: host index generation in the reference's call order (anchor, then negatives, per class)
    plan = []
    for k in range(valid_seg):                                          # :435-509, k = LOOP COUNTER
        bank = memobank[valid_classes[k]][0]
        if n_anchor[k] == 0 or bank.shape[0] == 0:
            continue
        a_idx = draw(int(n_anchor[k]), q_arg)
        n_idx = draw(int(bank.shape[0]), n_arg)
        plan.append((k, a_idx, n_idx))
        if _trace is not None:
            _trace.setdefault("anchor_idx", []).append(a_idx)
            _trace.setdefault("neg_idx", []).append(n_idx)

    for (k, a_idx, n_idx) in plan:
        vc = valid_classes[k]
        bank = memobank[vc][0]
        Lb = int(bank.shape[0])
        Lp = _ceil(Lb, 16)
        a_dev = a_idx.to(dev, non_blocking=True)
        n_dev = n_idx.to(dev, non_blocking=True)
        # anchors (student rows; gradient flows)                          :455-457
        A = torch.empty((Q, D), dtype=torch.float32, device=dev)
        L.call("arco_gather_rows", L.ptr(R), ldr, D, L.ptr(lists[k]), L.ptr(a_dev), None, 0, Q, L.ptr(A), D)
        An = torch.zeros((Q, Dp), dtype=torch.float32, device=dev)
        invA = torch.empty(Q, dtype=torch.float32, device=dev)
        L.call("arco_normalize_rows", L.ptr(A), D, Q, D, EPS, L.ptr(An), Dp, None, 0, L.ptr(invA))
        # bank, normalised once per class                                  :466
        if vc not in norm_cache:
            Bn = torch.zeros((Lp, Dp), dtype=torch.float32, device=dev)
            Bt = torch.zeros((Dp, Lp), dtype=torch.float32, device=dev) if need_grad else None
            L.call("arco_normalize_rows", L.ptr(bank), D, Lb, D, EPS, L.ptr(Bn), Dp, L.ptr(Bt), Lp, None)
            norm_cache[vc] = (Bn, Bt)
        Bn, Bt = norm_cache[vc]
        # positive = prototype of LOOP-COUNTER class k                     :480-486
        pos = proto[k].view(1, D)
        if momentum_prototype is not None:                               # :488-497
            pos = pos.view(1, 1, D).repeat(Q, 1, 1)
            if not bool((momentum_prototype == 0).all()):
                decay = min(1 - 1 / i_iter, 0.999)
                pos = (1 - decay) * pos + decay * momentum_prototype[vc].to(dev)
            prototype[vc] = pos.clone()
            pos = pos.view(Q, D).contiguous()
        nP = int(pos.shape[0])
        Pn = torch.zeros((nP, Dp), dtype=torch.float32, device=dev)
        L.call("arco_normalize_rows", L.ptr(pos.contiguous()), D, nP, D, EPS, L.ptr(Pn), Dp, None, 0, None)
        ldp = Dp if nP > 1 else 0
        # scores vs the whole bank on the matrix cores, then multiplicity-weighted softmax-CE   :503-509
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
            L.call("arco_conv_fwd", L.ptr(W), Lp, Lp, L.ptr(Bt), Dp, L.ptr(G), Dp, None, None, 0, None, None,
                   1, 1, 1, Q)
            dA = torch.empty((Q, Dp), dtype=torch.float32, device=dev)
            L.call("arco_infonce_anchor_grad", L.ptr(G), L.ptr(An), L.ptr(Pn), ldp, L.ptr(gpos), L.ptr(invA),
                   Q, Dp, EPS, 1.0 / (Q * valid_seg), L.ptr(dA))
            pieces.append((k, a_dev, dA, Dp))

    loss = loss_acc[0]
    if need_grad:
        loss = _AnchorGrad.apply(rep, loss, lists, pieces)
    if momentum_prototype is None:
        return new_keys, loss
    return prototype, new_keys, loss


@torch.no_grad()
def _enqueue_counted(keys_tail, n_total, queue, queue_ptr, queue_size):
    """Enqueue when only the last `queue_size` of `n_total` new keys were gathered
    (the rest cannot survive cat(...)[-queue_size:]); returns n_total like the reference."""
    D = int(keys_tail.shape[1])
    old = queue[0].to(device=keys_tail.device, dtype=torch.float32).contiguous()
    out = torch.empty((queue_size, D), device=keys_tail.device, dtype=torch.float32)
    L.call("arco_bank_append", L.ptr(old), int(old.shape[0]), L.ptr(keys_tail), int(keys_tail.shape[0]),
           queue_size, D, L.ptr(out))
    queue[0] = out
    queue_ptr[0] = queue_size
    return n_total
