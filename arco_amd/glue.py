"""Step glue of the trainers on the GPU: softmax / pseudo-labels /
one-hot / entropy-percentile masks, mirroring train_arco_2d.py:284-286,342-393,492-498
(3-D: train_arco_3d.py:463-469).  Thin wrappers over csrc/glue.hip."""
import torch

from . import _lib as L
from . import ops
from ._contrast import rows_view


def _geom(pred):
    b, C = int(pred.shape[0]), int(pred.shape[1])
    P = 1
    for s in pred.shape[2:]:
        P *= int(s)
    r, ld = rows_view(pred.detach())
    return r, ld, b, C, P


@torch.no_grad()
def softmax(pred):
    """torch.softmax(pred, dim=1) as NC[spatial]-contiguous planes (what the loss consumes)."""
    r, ld, b, C, P = _geom(pred)
    out = torch.empty(pred.shape, dtype=torch.float32, device=pred.device)
    L.call("arco_softmax_rows", L.ptr(r), ld, b * P, C, P, L.ptr(out), None, None, None)
    return out


@torch.no_grad()
def softmax_max(pred):
    """torch.max(torch.softmax(pred, dim=1), dim=1) -> (pseudo_logits, pseudo_labels)  (train_arco_2d.py:286)."""
    r, ld, b, C, P = _geom(pred)
    sp = tuple(pred.shape[2:])
    mp = torch.empty((b, *sp), dtype=torch.float32, device=pred.device)
    am = torch.empty((b, *sp), dtype=torch.int64, device=pred.device)
    L.call("arco_softmax_rows", L.ptr(r), ld, b * P, C, P, None, L.ptr(mp), L.ptr(am), None)
    return mp, am


@torch.no_grad()
def label_onehot(inputs, num_segments):
    """train_arco_2d.py:492-498; returns int64 like the `.long()` the trainer applies (:394)."""
    lab = inputs.to(torch.int64).contiguous()
    b = int(lab.shape[0])
    P = lab.numel() // b
    out = torch.empty((b, num_segments, *lab.shape[1:]), dtype=torch.int64, device=lab.device)
    L.call("arco_label_onehot", L.ptr(lab), b * P, num_segments, P, L.ptr(out))
    return out


state_reduce_hook = None      # data parallel: in-place sum over ranks of an integer device tensor (dist.allreduce_sum)


@torch.no_grad()
def entropy_masks(pred_u, label_l_raw, label_u_raw, alpha_t):
    """low_mask_all / high_mask_all of train_arco_2d.py:352-393: entropy of softmax(pred_u), exact
    np.percentile(alpha_t) / (100-alpha_t) over valid pixels (device radix select, no host sync)."""
    r, ld, b, C, P = _geom(pred_u)
    dev = pred_u.device
    ent = torch.empty(b * P, dtype=torch.float32, device=dev)
    L.call("arco_softmax_rows", L.ptr(r), ld, b * P, C, P, None, None, None, L.ptr(ent))
    ll = label_l_raw.to(torch.int64).contiguous()
    lu = label_u_raw.to(torch.int64).contiguous()
    n_l, n_u = ll.numel(), lu.numel()
    state = torch.empty(L.query("arco_sel_state_bytes"), dtype=torch.uint8, device=dev)
    sp = tuple(pred_u.shape[2:])
    low = torch.empty((int(ll.shape[0]) + b, 1, *sp), dtype=torch.float32, device=dev)
    high = torch.empty_like(low)
    if state_reduce_hook is None:
        L.call("arco_entropy_masks", L.ptr(ent), L.ptr(ll), L.ptr(lu), n_l, n_u, float(alpha_t), float(100 - alpha_t),
               L.ptr(state), L.ptr(low), L.ptr(high))
        return low, high
    # data parallel: percentiles of the GLOBAL batch - the valid count and every digit histogram are summed over ranks
    o_cnt, o_hist = L.query("arco_sel_state_offset", 0), L.query("arco_sel_state_offset", 1)
    cnt = state[o_cnt:o_cnt + 8].view(torch.int64)
    hist = state[o_hist:o_hist + 4 * 256 * 4].view(torch.int32)

    def phase(ph, ps=0):
        L.call("arco_entropy_masks_phase", ph, ps, L.ptr(ent), L.ptr(ll), L.ptr(lu), n_l, n_u, float(alpha_t),
               float(100 - alpha_t), L.ptr(state), L.ptr(low), L.ptr(high))
    phase(0)
    state_reduce_hook(cnt)
    phase(1)
    for ps in range(4):
        phase(2, ps)
        state_reduce_hook(hist)
        phase(3, ps)
    phase(4)
    return low, high


class _SupLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, label):
        r, ld, b, C, P = _geom(pred)
        lab = label.to(torch.int64).contiguous()
        ws = torch.empty(L.query("arco_seg_ws_doubles", b * P, C, b), dtype=torch.float64, device=pred.device)
        out = torch.empty(2, dtype=torch.float32, device=pred.device)
        L.call("arco_sup_loss_fwd", L.ptr(r), ld, b * P, C, L.ptr(lab), L.ptr(ws), L.ptr(out))
        ctx.save_for_backward(pred, lab, ws)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_ce, g_dice):
        pred, lab, ws = ctx.saved_tensors
        r, ld, b, C, P = _geom(pred)
        dx = torch.empty((b, *pred.shape[2:], C), dtype=torch.float32, device=pred.device)
        L.call("arco_sup_loss_bwd", L.ptr(r), ld, b * P, C, L.ptr(lab), L.ptr(ws), L.ptr(g_ce.contiguous().float()),
               L.ptr(g_dice.contiguous().float()), L.ptr(dx), C)
        return dx.movedim(-1, 1), None


def supervised_loss(pred_l, label):
    """(CrossEntropyLoss()(pred_l, label), DiceLoss(C)(softmax(pred_l), label)) of train_arco_2d.py:336-339."""
    return _SupLossFn.apply(pred_l, label)


class _UnsupLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, conf, thr):
        r, ld, b, C, P = _geom(pred)
        lab = target.to(torch.int64).contiguous()
        cf = conf.to(torch.float32).contiguous()
        ws = torch.empty(L.query("arco_loss_slabs", b) * 4 * b + b + 1, dtype=torch.float64, device=pred.device)
        out = torch.empty(1, dtype=torch.float32, device=pred.device)
        L.call("arco_unsup_loss_fwd", L.ptr(r), ld, b, P, C, L.ptr(lab), L.ptr(cf), float(thr), L.ptr(ws), L.ptr(out))
        ctx.save_for_backward(pred, lab, ws)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        pred, lab, ws = ctx.saved_tensors
        r, ld, b, C, P = _geom(pred)
        dx = torch.empty((b, *pred.shape[2:], C), dtype=torch.float32, device=pred.device)
        L.call("arco_unsup_loss_bwd", L.ptr(r), ld, b, P, C, L.ptr(lab), L.ptr(ws), L.ptr(g.contiguous().float()), L.ptr(dx), C)
        return dx.movedim(-1, 1), None, None, None


def compute_unsupervised_loss(predict, target, logits, strong_threshold):
    """train_arco_2d.py:482-489 (same name and arguments)."""
    return _UnsupLossFn.apply(predict, target, logits, strong_threshold)


@torch.no_grad()
def eqv_mask(labels, logits, weak_threshold):
    """train_arco_2d.py:406-410: [B,1,H,W] float mask, 1 where the (pseudo-)label is foreground and its confidence
    reaches weak_threshold."""
    return ((labels != 0) & (logits >= weak_threshold)).to(torch.float32).unsqueeze(1)


class _EqvLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_tps, pred_tps_org, mask_tps):
        L.require_gpu(pred_tps, pred_tps_org, mask_tps)
        p, ldp, b, C, P = _geom(pred_tps)
        q, ldq, _, _, _ = _geom(pred_tps_org.detach())
        m = mask_tps.detach().to(torch.float32).contiguous().view(-1)
        ws = torch.empty(L.query("arco_loss_slabs", b) * 2 * b + b, dtype=torch.float64, device=pred_tps.device)
        out = torch.empty(1, dtype=torch.float32, device=pred_tps.device)
        L.call("arco_eqv_loss_fwd", L.ptr(p), ldp, L.ptr(q), ldq, L.ptr(m), b, P, C, L.ptr(ws), L.ptr(out))
        ctx.save_for_backward(pred_tps, pred_tps_org, m, ws)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        pred_tps, pred_tps_org, m, ws = ctx.saved_tensors
        p, ldp, b, C, P = _geom(pred_tps)
        q, ldq, _, _, _ = _geom(pred_tps_org)
        d = ops.new_act_nd(b, C, tuple(int(v) for v in pred_tps.shape[2:]), pred_tps.device)
        gg = g.reshape(1).to(torch.float32).contiguous()
        L.call("arco_eqv_loss_bwd", L.ptr(p), ldp, L.ptr(q), ldq, L.ptr(m), b, P, C, L.ptr(ws), L.ptr(gg), L.ptr(d), C)
        return d, None, None


def eqv_loss(pred_tps, pred_tps_org, mask_tps):
    """train_arco_2d.py:419-423: mean over images of the masked mean of KL(softmax(pred_tps_org) || softmax(pred_tps))."""
    return _EqvLossFn.apply(pred_tps, pred_tps_org, mask_tps)


# ---------------------------------------------------------------------------------------------------------------
# revisiting loss (train_arco_2d.py:108-136,156-159,334,398-400).  It has NO gradient path (the
# distances that enter the loss come from the teacher's representation), so it only changes the logged loss value; it
# needs the DENSE student and teacher representations of the unlabeled images, which the default row-sparse step
# never materialises - the trainers compute it only under --revisit 1 (which switches the dense head / teacher on).
# ---------------------------------------------------------------------------------------------------------------
class RevisitPool:
    """random_pool of train_arco_2d.py:156-159: K unit vectors of length D*prod(spatial), drawn with torch.randn on the
    CPU generator in the reference's [K, D, *spatial] order, kept on the GPU in the channels-last order of the
    representations ([K, prod(spatial), D] flattened; a dot product does not care as long as both sides agree) with the
    row count padded to a multiple of 16 for the GEMM kernel (extra rows are zero)."""

    def __init__(self, K, D, spatial, device, values=None):
        self.K, self.D, self.spatial = int(K), int(D), tuple(int(v) for v in spatial)
        n = self.D
        for v in self.spatial:
            n *= v
        self.n = n
        if values is None:
            g = torch.randn(self.K, self.D, *self.spatial)                       # :156 (same draw from the CPU generator)
            g = torch.nn.functional.normalize(g.view(self.K, -1), dim=1)         # :157-158
            nd = len(self.spatial)
            values = g.view(self.K, self.D, *self.spatial).permute(0, *range(2, 2 + nd), 1).reshape(self.K, -1)
        kpad = (self.K + 15) // 16 * 16
        self.rows = torch.zeros((kpad, n), dtype=torch.float32, device=device)
        self.rows[:self.K].copy_(values)
        self.ptr = torch.zeros(1, dtype=torch.long)                              # random_pool_ptr (:159)

    def channels_first(self):
        """[K, D*prod(spatial)] in the reference's flattening order (for checks)."""
        nd = len(self.spatial)
        return self.rows[:self.K].view(self.K, *self.spatial, self.D).permute(0, 1 + nd, *range(1, 1 + nd)).reshape(self.K, -1)


def _flat_rows(rep):
    """[b, n] view of a channels-last dense representation [b, D, *spatial] (rows of one image are contiguous)."""
    r, ld = rows_view(rep.detach())
    b = int(rep.shape[0])
    if ld != int(rep.shape[1]) or not r.is_contiguous():
        raise RuntimeError("arco_amd: the revisiting loss needs dense, unpadded channels-last representations")
    return r.reshape(b, -1)


@torch.no_grad()
def _pool_dots(pool, flat):
    """dots[b, K] = flat[b, :] . pool.rows[k, :] - one split-K GEMM launch (a [b x K] output over a 10^7-long K)."""
    b, n = int(flat.shape[0]), int(flat.shape[1])
    assert n == pool.n and n % 16 == 0, (n, pool.n)
    kpad = int(pool.rows.shape[0])
    splits = max(1, min(2048, n // 8192))
    out = torch.empty((b, kpad), dtype=torch.float32, device=flat.device)
    ws = torch.empty((splits, b, kpad), dtype=torch.float32, device=flat.device)
    L.call("arco_gemm_splitk", L.ptr(flat), n, n, L.ptr(pool.rows), pool.K, L.ptr(out), kpad, b, splits, L.ptr(ws))
    return out[:, :pool.K]


@torch.no_grad()
def get_revisiting_loss(random_pool, rep_u, rep_u_teacher, topk=5):
    """train_arco_2d.py:126-136: normalise the flattened student / teacher representations of the unlabeled images,
    dist = 2 - 2 <rep, pool_k>; the student picks its topk nearest pool entries, the loss is the teacher's mean distance
    to those.  Returns a 0-d tensor without a graph (the reference's value has no gradient path either)."""
    L.require_gpu(rep_u, rep_u_teacher)
    fu, ft = _flat_rows(rep_u), _flat_rows(rep_u_teacher)
    eps = 1e-12
    dist_t = 2 - 2 * _pool_dots(random_pool, fu) / torch.linalg.vector_norm(fu, dim=1).clamp_min(eps)[:, None]
    dist_q = 2 - 2 * _pool_dots(random_pool, ft) / torch.linalg.vector_norm(ft, dim=1).clamp_min(eps)[:, None]
    _, nn_index = dist_t.topk(topk, dim=1, largest=False)
    nn_dist_q = torch.gather(dist_q, 1, nn_index)
    return (nn_dist_q.sum(dim=1) / topk).mean()


@torch.no_grad()
def revisit_enqueue(rep_u_teacher, random_pool):
    """_dequeue_and_enqueue of train_arco_2d.py:108-119 with the normalised teacher rows (:398-400): overwrite pool rows
    [ptr, ptr + b), ptr = (ptr + b) % K; like the reference it requires K % b == 0."""
    ft = _flat_rows(rep_u_teacher)
    b = int(ft.shape[0])
    ptr = int(random_pool.ptr)
    assert random_pool.K % b == 0
    torch.div(ft, torch.linalg.vector_norm(ft, dim=1).clamp_min(1e-12)[:, None], out=random_pool.rows[ptr:ptr + b])
    random_pool.ptr[0] = (ptr + b) % random_pool.K
