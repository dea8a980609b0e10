"""CPU oracle for the ARCO stratified pixel-contrastive hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under arco_amd/ may import this module; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
as the checker.  It is a restatement (plain PyTorch-CPU fp32 / numpy) of the
reference algorithm, written in "flatten pixels + GEMM + gather" form; every
function cites the reference file:line it follows (paths relative to
/root/reference/code).  It is pinned against golden vectors produced by running
the real reference in the build container (oracle/gen_golden.py ->
tests/golden/*.npz); see tests/test_oracle_golden.py.

The reference has two loss files that differ only in the number of spatial
dims (loss_helper_3d.py = 4-D/2-D images, loss_helper.py = 5-D/3-D volumes);
the restatement below is rank-generic and serves both.
"""
import math
import random

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# L3  memory bank                                   loss_helper_3d.py:12-32
# --------------------------------------------------------------------------


@torch.no_grad()
def dequeue_and_enqueue(keys, queue, queue_ptr, queue_size):
    """FIFO-by-truncation bank append (loss_helper_3d.py:12-32; loss_helper.py:142-162)."""
    n = int(keys.shape[0])
    grown = torch.cat((queue[0], keys.detach().clone().cpu()), dim=0)
    if grown.shape[0] >= queue_size:
        queue[0] = grown[grown.shape[0] - queue_size:, :]
        queue_ptr[0] = queue_size
    else:
        queue[0] = grown
        queue_ptr[0] = (int(queue_ptr) + n) % queue_size
    return n


# --------------------------------------------------------------------------
# L4  samplers                                      loss_helper_3d.py:35-268
# --------------------------------------------------------------------------


def _strata_1d(high, shape, patch, antithetic):
    """1-D stratified fallback (loss_helper_3d.py:35-80 asmc / :83-117 smc).

    Uses python `random` for the picks and one torch.randperm for the shuffle.
    """
    if high // patch > shape or high < patch:
        return torch.randint(high, size=(shape,))
    blocks = (high - high % patch) // patch
    per = shape // blocks
    out = []
    if high % patch != 0 and blocks > shape:          # :44-48 / :92-96 (unreachable: guarded above)
        b = 0
        while len(out) < shape:
            out.append(random.randint(b * patch, (b + 1) * patch - 1))
            b += 1
    else:
        for b in range(blocks):
            lo, hi = b * patch, (b + 1) * patch - 1
            if antithetic:
                first = [random.randint(lo, hi) for _ in range(per // 2)]
                out.extend(first)
                out.extend([(2 * b + 1) * patch - 1 - v for v in first])
            else:
                out.extend([random.randint(lo, hi) for _ in range(per)])
        while len(out) < shape:
            out.append(random.randint(0, high - 1))
    vals = torch.Tensor(out).reshape(shape,).long()      # float32 round trip, :74 / :111
    return vals[torch.randperm(shape).long()]


def monte_carlo_sample(high=5233, shape=256, patch=16):
    return _strata_1d(high, shape, patch, antithetic=False)


def as_monte_carlo_sample(high=5233, shape=256, patch=16):
    return _strata_1d(high, shape, patch, antithetic=True)


def _grid_blocks(edge, cut):
    """Yield the row-major flattened pixel ids of each of the cut*cut blocks
    (loss_helper_3d.py:140-155): block side edge//cut, last row/col absorb the rest."""
    side = edge // cut
    grid = np.arange(edge * edge).reshape(edge, edge)
    for bi in range(cut):
        r0 = bi * side
        r1 = edge if bi == cut - 1 else (bi + 1) * side
        for bj in range(cut):
            c0 = bj * side
            c1 = edge if bj == cut - 1 else (bj + 1) * side
            yield grid[r0:r1, c0:c1].flatten()


def _grid_sample(high, shape, cut, antithetic):
    """2-D stratified sampler (loss_helper_3d.py:120-184 smc, :187-268 asmc).

    RNG call order on the torch CPU default generator: per block randperm(n_blk)
    then randint(n_blk, (k,)); then randperm(n_kept); then one randint(high,(1,1))
    per missing element.  Any exception in the reference drops to the 1-D sampler.
    Two are reachable, both at the FIRST block and before any generator draw
    (randperm(0)/randperm(1) consume nothing): an empty block (edge < cut) makes
    randint(0, ...) raise; a 1-element block is indexed by a 1-element tensor,
    numpy returns a scalar, and `.shape[0]` raises.  So edge//cut <= 1
    (high <= 56 at cut=4) always takes the 1-D path.
    """
    edge = round(math.sqrt(high))
    per_block = shape * edge ** 2 // high // (cut ** 2)
    take = per_block // 2 if antithetic else per_block
    rows = []
    for blk in _grid_blocks(edge, cut):
        n = blk.shape[0]
        if antithetic:
            center = (2 * np.mean(blk)).astype(np.int64) if n > 0 else None
        if n <= 1:
            return None                                  # reference raises -> fallback
        blk = blk[torch.randperm(n).numpy()]
        pick = blk[torch.randint(n, (take,)).numpy()]
        rows.append(pick)
        if antithetic:
            rows.append(center - pick)
    vals = torch.from_numpy(np.array(rows)).to(torch.float32).flatten().long()   # :163 / :245-246
    vals = vals[vals < high]
    vals = vals[torch.randperm(vals.shape[0])]
    if vals.shape[0] < shape:
        pad = torch.cat([torch.randint(high, (1, 1)) for _ in range(shape - vals.shape[0])]).flatten()
        vals = torch.cat([vals, pad])
    return vals[:shape]


@torch.no_grad()
def grid_monte_carlo_sample(high=5233, shape=256, cut_count=4):
    out = _grid_sample(high, shape, cut_count, antithetic=False)
    return monte_carlo_sample(high, shape) if out is None else out


@torch.no_grad()
def grid_as_monte_carlo_sample(high=5233, shape=256, cut_count=4):
    out = _grid_sample(high, shape, cut_count, antithetic=True)
    return as_monte_carlo_sample(high, shape) if out is None else out


# --------------------------------------------------------------------------
# L1/L2/L5/L6  contrastive loss            loss_helper_3d.py:271-513 (2-D)
#                                           loss_helper.py:442-686   (3-D)
# --------------------------------------------------------------------------


def class_rank(prob_rows):
    """rank[p, c] = position of class c in a descending sort of prob_rows[p, :]
    (loss_helper_3d.py:352-358).  torch's CPU sort is stable for <=16 classes:
    ties keep the lower class index first."""
    pi = prob_rows.unsqueeze(2)                # [n, C(i), 1]
    pj = prob_rows.unsqueeze(1)                # [n, 1, C(j)]
    C = prob_rows.shape[1]
    earlier = torch.tril(torch.ones(C, C, dtype=torch.bool), -1)      # [i, j] : j < i
    return ((pj > pi) | ((pj == pi) & earlier)).sum(2)


def _rows(x):
    """[B, K, *spatial] -> [B*prod(spatial), K] in (b, spatial...) row-major order
    (the reference's permute(0,2,3,1) / (0,2,3,4,1), loss_helper_3d.py:344-345)."""
    return x.movedim(1, -1).reshape(-1, x.shape[1])


def compute_contra_memobank_loss(rep, label_l, label_u, prob_l, prob_u, low_mask, high_mask,
                                 memobank, queue_prtlis, queue_size, rep_teacher,
                                 momentum_prototype=None, i_iter=0, delta_n=1.0, func='asmc',
                                 num_queries=256, num_negatives=512, temp=0.5, trace=None):
    """Restatement of compute_contra_memobank_loss (loss_helper_3d.py:271-513).

    `trace`, if a dict, receives the per-class masks' row lists and the sampled
    index tensors (test instrumentation; not in the reference).
    """
    delta_p, low_rank, high_rank = 0.3, 3, 20                     # :316-318
    D = rep.shape[1]
    C = label_l.shape[1]
    n_lab_rows = label_l.shape[0] * int(np.prod(label_l.shape[2:]))

    if func == 'asmc':                                             # :327-338
        draw = grid_as_monte_carlo_sample
        q_arg, n_arg = num_queries, num_queries * num_negatives
    elif func == 'smc':
        draw = grid_monte_carlo_sample
        q_arg, n_arg = num_queries, num_queries * num_negatives
    else:
        draw = torch.randint
        q_arg, n_arg = (num_queries,), (num_queries * num_negatives,)

    R = _rows(rep)                                                 # student rows (grad)
    T = _rows(rep_teacher).detach()
    lab = _rows(torch.cat((label_l, label_u), 0))
    prob = _rows(torch.cat((prob_l, prob_u), 0))
    low_valid = lab * low_mask.reshape(-1, 1)                      # :341
    high_valid = lab * high_mask.reshape(-1, 1)                    # :342
    rank = class_rank(prob)
    labeled_row = torch.arange(lab.shape[0]) < n_lab_rows

    anchor_rows, protos, new_keys, seg_num, valid_classes = [], [], [], [], []
    for c in range(C):                                             # :364-415
        lv = low_valid[:, c].bool()
        anchor_m = (prob[:, c] > delta_p) & lv
        hard_m = (prob[:, c] < delta_n) & high_valid[:, c].bool()
        anchor_rows.append(torch.nonzero(anchor_m).flatten())
        protos.append(T[lv].mean(0, keepdim=True))
        cls_u = (rank[:, c] >= low_rank) & (rank[:, c] < high_rank)
        cls_l = (rank[:, c] < low_rank) & (lab[:, c] == 0)
        neg_m = hard_m & torch.where(labeled_row, cls_l, cls_u)
        new_keys.append(dequeue_and_enqueue(T[neg_m], memobank[c], queue_prtlis[c], queue_size[c]))
        if low_valid[:, c].sum() > 0:
            seg_num.append(int(low_valid[:, c].sum().item()))
            valid_classes.append(c)
        if trace is not None:
            trace.setdefault('anchor_rows', []).append(anchor_rows[-1].clone())
            trace.setdefault('neg_rows', []).append(torch.nonzero(neg_m).flatten())

    if len(seg_num) <= 1:                                          # :417-424
        zero = torch.tensor(0.0) * rep.sum()
        return (new_keys, zero) if momentum_prototype is None else (momentum_prototype, new_keys, zero)

    loss = torch.tensor(0.0)
    proto = torch.cat(protos)                                      # indexed by LOOP COUNTER below (:481)
    prototype = torch.zeros((C, num_queries, 1, D))
    for k in range(len(seg_num)):                                  # :435-509  (k is the loop counter)
        bank = memobank[valid_classes[k]][0]
        if anchor_rows[k].shape[0] == 0 or bank.shape[0] == 0:     # :436-462
            loss = loss + 0 * rep.sum()
            continue
        a_idx = draw(anchor_rows[k].shape[0], q_arg)
        A = R[anchor_rows[k][a_idx]]                               # [Q, D], grad flows
        with torch.no_grad():
            n_idx = draw(bank.shape[0], n_arg)
            pos = proto[k].view(1, 1, D).repeat(num_queries, 1, 1)
            if momentum_prototype is not None:                     # :488-497
                if not (momentum_prototype == 0).all():
                    decay = min(1 - 1 / i_iter, 0.999)
                    pos = (1 - decay) * pos + decay * momentum_prototype[valid_classes[k]]
                prototype[valid_classes[k]] = pos.clone()
        if trace is not None:
            trace.setdefault('anchor_idx', []).append(a_idx.clone())
            trace.setdefault('neg_idx', []).append(n_idx.clone())
        # cosine similarity in GEMM + gather form (:503-505): each vector is divided
        # by max(||x||, 1e-8) (torch>=2 cosine_similarity), then dotted.
        eps = 1e-8
        An = A / A.norm(dim=1, keepdim=True).clamp_min(eps)
        Bn = bank / bank.norm(dim=1, keepdim=True).clamp_min(eps)
        Pn = pos[:, 0] / pos[:, 0].norm(dim=1, keepdim=True).clamp_min(eps)
        S = An @ Bn.t()                                            # [Q, len(bank)]
        neg_logit = S.gather(1, n_idx.view(num_queries, num_negatives))
        pos_logit = (An * Pn).sum(1, keepdim=True)
        logits = torch.cat((pos_logit, neg_logit), 1)
        loss = loss + F.cross_entropy(logits / temp, torch.zeros(num_queries).long())   # :507-509
    loss = loss / len(seg_num)
    return (new_keys, loss) if momentum_prototype is None else (prototype, new_keys, loss)


# --------------------------------------------------------------------------
# N1-N3  2-D network pieces (functional, parameters by reference state_dict key)
# --------------------------------------------------------------------------


def bn_train(x, w, b, rm=None, rv=None, momentum=0.1, eps=1e-5):
    """Train-mode BatchNorm over all dims but channel (nn.BatchNorm2d/3d default)."""
    return F.batch_norm(x, rm, rv, w, b, True, momentum, eps)


def conv_block_2d(x, sd, pre, slope=0.01, train=True, track=False):
    """networks/unetWithArgs.py:31-47 with Dropout disabled (p forced to 0).  train=False: net.eval()
    (BatchNorm on the running statistics, test_2D.py:78).  track=True: train mode that also makes the momentum
    update of sd's running statistics in place, as nn.BatchNorm2d does."""
    for k in ("0", "4"):
        x = F.conv2d(x, sd[f"{pre}.conv_conv.{k}.weight"], sd[f"{pre}.conv_conv.{k}.bias"], padding=1)
        bnk = str(int(k) + 1)
        if train:
            rm = sd[f"{pre}.conv_conv.{bnk}.running_mean"] if track else None
            rv = sd[f"{pre}.conv_conv.{bnk}.running_var"] if track else None
            x = bn_train(x, sd[f"{pre}.conv_conv.{bnk}.weight"], sd[f"{pre}.conv_conv.{bnk}.bias"], rm, rv)
        else:
            x = F.batch_norm(x, sd[f"{pre}.conv_conv.{bnk}.running_mean"], sd[f"{pre}.conv_conv.{bnk}.running_var"],
                             sd[f"{pre}.conv_conv.{bnk}.weight"], sd[f"{pre}.conv_conv.{bnk}.bias"], False, 0.1, 1e-5)
        x = F.leaky_relu(x, slope)
    return x


def unet_forward(x, sd, train=True, track=False):
    """networks/unetWithArgs.py:109-116 (Encoder), :142-158 (Decoder), :345-348 (UNet);
    dropout off.  NB the Decoder constructs UpBlock WITHOUT passing `bilinear`
    (:130-137), so UpBlock's default bilinear=True applies (:66): the up-path is
    conv1x1 (bias) -> x2 bilinear upsample align_corners=True (:72-75,80-83), not
    ConvTranspose2d, whatever params['bilinear'] (:317) says."""
    feats = [conv_block_2d(x, sd, "encoder.in_conv", train=train, track=track)]
    for i in range(1, 5):
        feats.append(conv_block_2d(F.max_pool2d(feats[-1], 2), sd, f"encoder.down{i}.maxpool_conv.1", train=train, track=track))
    x = feats[4]
    fmap = [x]
    for i, skip in zip(range(1, 5), (feats[3], feats[2], feats[1], feats[0])):
        up = F.conv2d(x, sd[f"decoder.up{i}.conv1x1.weight"], sd[f"decoder.up{i}.conv1x1.bias"])
        up = F.interpolate(up, scale_factor=2, mode='bilinear', align_corners=True)
        x = conv_block_2d(torch.cat([skip, up], 1), sd, f"decoder.up{i}.conv", train=train, track=track)
        fmap.append(x)
    out = F.conv2d(x, sd["decoder.out_conv.weight"], sd["decoder.out_conv.bias"], padding=1)
    return out, feats[4], fmap


def feature_extractor_forward(fmap, sd, mode='bilinear'):
    """model_2D.py:35-55 / model_3D.py:37-63: residual 1x1 convs + align_corners upsample + cat."""
    conv = F.conv2d if mode == 'bilinear' else F.conv3d
    x = conv(fmap[0], sd["fea0.weight"]) + fmap[0]
    for i in range(1, 5):
        x = F.interpolate(x, size=fmap[i].shape[2:], mode=mode, align_corners=True)
        x = torch.cat((x, fmap[i]), 1)
        y = conv(x, sd[f"fea{i}.weight"])
        x = y + x if i < 4 else y
    return x


# --------------------------------------------------------------------------
# V1  3-D V-Net (networks/vnetWithArgs.py:145-252), batchnorm, dropout off
# --------------------------------------------------------------------------


def _vnet_bn(x, sd, key, train, track=False):
    """track=True: train mode that also makes the momentum update of sd's running statistics in place (nn.BatchNorm3d)."""
    if train:
        rm = sd[f"{key}.running_mean"] if track else None
        rv = sd[f"{key}.running_var"] if track else None
        return bn_train(x, sd[f"{key}.weight"], sd[f"{key}.bias"], rm, rv)
    return F.batch_norm(x, sd[f"{key}.running_mean"], sd[f"{key}.running_var"], sd[f"{key}.weight"], sd[f"{key}.bias"],
                        False, 0.1, 1e-5)


def _vnet_stage(x, sd, pre, n, train=True, track=False, tap=None):
    for s in range(n):
        z = F.conv3d(x, sd[f"{pre}.conv.{3 * s}.weight"], sd[f"{pre}.conv.{3 * s}.bias"], padding=1)
        a = F.relu(_vnet_bn(z, sd, f"{pre}.conv.{3 * s + 1}", train, track))
        if tap is not None:      # tests: (conv key, input, pre-BN output, activation) of every 3x3x3 stage, in forward order
            z.retain_grad(); a.retain_grad()
            tap.append((f"{pre}.conv.{3 * s}", x, z, a))
        x = a
    return x


def _vnet_down(x, sd, pre, train=True, track=False):
    x = F.conv3d(x, sd[f"{pre}.conv.0.weight"], sd[f"{pre}.conv.0.bias"], stride=2)
    return F.relu(_vnet_bn(x, sd, f"{pre}.conv.1", train, track))


def _vnet_up(x, sd, pre, train=True, track=False):
    x = F.conv_transpose3d(x, sd[f"{pre}.conv.0.weight"], sd[f"{pre}.conv.0.bias"], stride=2)
    return F.relu(_vnet_bn(x, sd, f"{pre}.conv.1", train, track))


def vnet_forward(x, sd, train=True, track=False, tap=None):
    """train=False: net.eval() (BatchNorm on the running statistics), as the evaluation uses it; track=True: train mode
    with the running-statistics momentum updates of the BatchNorm3d layers made in place, in forward order."""
    t, k = train, track
    x1 = _vnet_stage(x, sd, "block_one", 1, t, k, tap)
    x2 = _vnet_stage(_vnet_down(x1, sd, "block_one_dw", t, k), sd, "block_two", 2, t, k, tap)
    x3 = _vnet_stage(_vnet_down(x2, sd, "block_two_dw", t, k), sd, "block_three", 3, t, k, tap)
    x4 = _vnet_stage(_vnet_down(x3, sd, "block_three_dw", t, k), sd, "block_four", 3, t, k, tap)
    x5 = _vnet_stage(_vnet_down(x4, sd, "block_four_dw", t, k), sd, "block_five", 3, t, k, tap)
    u5 = _vnet_up(x5, sd, "block_five_up", t, k) + x4
    u6 = _vnet_up(_vnet_stage(u5, sd, "block_six", 3, t, k, tap), sd, "block_six_up", t, k) + x3
    u7 = _vnet_up(_vnet_stage(u6, sd, "block_seven", 3, t, k, tap), sd, "block_seven_up", t, k) + x2
    u8 = _vnet_up(_vnet_stage(u7, sd, "block_eight", 2, t, k, tap), sd, "block_eight_up", t, k) + x1
    x9 = _vnet_stage(u8, sd, "block_nine", 1, t, k, tap)
    out = F.conv3d(x9, sd["out_conv.weight"], sd["out_conv.bias"])
    fmap = [u5, u6, u7, u8, x9]
    return out, fmap[0], fmap


# --------------------------------------------------------------------------
# T1 / O1 / N5  trainer glue (train_arco_2d.py:284-286,342-393,425-435,492-498)
# --------------------------------------------------------------------------


def label_onehot(inputs, num_segments):
    """train_arco_2d.py:492-498 (3-D: train_arco_3d.py:463-469): negatives clamp to class 0."""
    idx = inputs.clamp_min(0).to(torch.int64).unsqueeze(1)
    out = torch.zeros([inputs.shape[0], num_segments, *inputs.shape[1:]])
    return out.scatter_(1, idx, 1.0)


def compute_unsupervised_loss(predict, target, logits, strong_threshold):
    """train_arco_2d.py:482-489."""
    b = predict.shape[0]
    valid = (target >= 0).float().view(b, -1).sum(-1)
    weight = logits.view(b, -1).ge(strong_threshold).sum(-1) / valid
    ce = F.cross_entropy(predict, target, reduction='none', ignore_index=-1)
    w = weight.view(b, *([1] * (ce.dim() - 1))) * ce
    return torch.mean(torch.masked_select(w, ce > 0))


def supervised_loss(pred_l, label, n_cls):
    """CrossEntropyLoss + DiceLoss(softmax(pred)) of train_arco_2d.py:336-339 (utils/losses.py:173-209):
    dice = 1/C sum_c (1 - (2 sum(p_c t_c) + s) / (sum p_c^2 + sum t_c^2 + s)), s = 1e-5."""
    ce = F.cross_entropy(pred_l, label.long())
    p = torch.softmax(pred_l, dim=1)
    dice = 0.0
    for c in range(n_cls):
        t = (label == c).float()
        inter, z, y = (p[:, c] * t).sum(), (p[:, c] * p[:, c]).sum(), (t * t).sum()
        dice = dice + (1 - (2 * inter + 1e-5) / (z + y + 1e-5))
    return ce, dice / n_cls


def entropy_masks(pred_u, label_l_raw, label_u_raw, alpha_t):
    """Entropy-percentile masks, train_arco_2d.py:352-393 (F.interpolate to the same
    size is the identity and is omitted).  Returns (low_mask_all, high_mask_all, entropy)."""
    prob = torch.softmax(pred_u, dim=1)
    entropy = -torch.sum(prob * torch.log(prob + 1e-10), dim=1)
    valid = label_u_raw >= 0
    ent_valid = entropy[valid].cpu().numpy().flatten()
    low_t = np.percentile(ent_valid, alpha_t)
    high_t = np.percentile(ent_valid, 100 - alpha_t)
    low_u = entropy.le(low_t).float() * valid
    high_u = entropy.ge(high_t).float() * valid
    lab_ok = (label_l_raw.unsqueeze(1) >= 0).float()
    return (torch.cat((lab_ok, low_u.unsqueeze(1))), torch.cat((lab_ok, high_u.unsqueeze(1))), entropy)


def ema_update(teacher_params, student_params, m=0.99):
    """model_2D.py:176-182: k = m*k + (1-m)*q over parameters() (buffers untouched)."""
    return [k * m + q * (1.0 - m) for k, q in zip(teacher_params, student_params)]


def poly_lr(base_lr, it, max_it):
    """train_arco_2d.py:433."""
    return base_lr * (1.0 - it / max_it) ** 0.9


def sgd_nesterov_step(p, g, buf, lr, momentum=0.9, wd=1e-4):
    """torch.optim.SGD(nesterov=True, weight_decay) single-tensor step
    (train_arco_2d.py:248). buf None on first step."""
    g = g + wd * p
    buf = g.clone() if buf is None else momentum * buf + g
    return p - lr * (g + momentum * buf), buf


# --------------------------------------------------------------------------
# V  evaluation: test_2D.py:52-103
# --------------------------------------------------------------------------


def dice_jaccard(pred, gt):
    """calculate_metric_percase (test_2D.py:52-66) restricted to the overlap metrics: medpy's binary dc / jc
    (2|A&B| / (|A|+|B|), |A&B| / |A|B|) with the reference's empty-set conventions; hd95 / asd need medpy."""
    pred = np.asarray(pred) > 0
    gt = np.asarray(gt) > 0
    if pred.sum() > 0 and gt.sum() > 0:
        inter = float(np.logical_and(pred, gt).sum())
        return 2.0 * inter / float(pred.sum() + gt.sum()), inter / float(np.logical_or(pred, gt).sum())
    if pred.sum() > 0 and gt.sum() == 0:
        return 1.0, 1.0
    return 0.0, 0.0


def test_single_volume(image, label, sd, classes, patch=(256, 256)):
    """test_2D.py:67-92 on arrays: per slice zoom(order=0) to `patch`, eval-mode net, argmax(softmax), zoom back;
    then (dice, jaccard) for classes 1..classes-1."""
    from scipy.ndimage import zoom
    prediction = np.zeros_like(label)
    for ind in range(image.shape[0]):
        sl = image[ind]
        x, y = sl.shape
        sl = zoom(sl, (patch[0] / x, patch[1] / y), order=0)
        inp = torch.from_numpy(sl).unsqueeze(0).unsqueeze(0).float()
        with torch.no_grad():
            out = torch.argmax(torch.softmax(unet_forward(inp, sd, train=False)[0], dim=1), dim=1).squeeze(0).numpy()
        prediction[ind] = zoom(out, (x / patch[0], y / patch[1]), order=0)
    return [dice_jaccard(prediction == i, label == i) for i in range(1, classes)], prediction


def test_single_case(net_fn, image, stride_xy, stride_z, patch_size, num_classes=1):
    """test_util.py:139-211: sliding-window inference of one [w,h,d] volume.  `net_fn(patch [1,1,px,py,pz]) -> logits`
    (eval mode).  Pads volumes smaller than the patch symmetrically with zeros, visits windows in (x, y, z) order with
    the last window clamped to the border, averages the softmax scores over the windows covering a voxel, arg-max."""
    import math
    w, h, d = image.shape
    pads = []
    for sz, p in zip((w, h, d), patch_size):
        tot = max(p - sz, 0)
        pads.append((tot // 2, tot - tot // 2))
    add_pad = any(a + b > 0 for a, b in pads)
    if add_pad:
        image = np.pad(image, pads, mode='constant', constant_values=0)
    ww, hh, dd = image.shape
    sx = math.ceil((ww - patch_size[0]) / stride_xy) + 1
    sy = math.ceil((hh - patch_size[1]) / stride_xy) + 1
    sz = math.ceil((dd - patch_size[2]) / stride_z) + 1
    score_map = np.zeros((num_classes,) + image.shape, dtype=np.float32)
    cnt = np.zeros(image.shape, dtype=np.float32)
    for x in range(sx):
        xs = min(stride_xy * x, ww - patch_size[0])
        for y in range(sy):
            ys = min(stride_xy * y, hh - patch_size[1])
            for z in range(sz):
                zs = min(stride_z * z, dd - patch_size[2])
                sl = (slice(xs, xs + patch_size[0]), slice(ys, ys + patch_size[1]), slice(zs, zs + patch_size[2]))
                patch = torch.from_numpy(image[sl][None, None].astype(np.float32))
                with torch.no_grad():
                    prob = torch.softmax(net_fn(patch), dim=1)[0].numpy()
                score_map[(slice(None),) + sl] += prob
                cnt[sl] += 1
    score_map = score_map / cnt[None]
    label_map = np.argmax(score_map, axis=0)
    if add_pad:
        (wl, _), (hl, _), (dl, _) = pads
        label_map = label_map[wl:wl + w, hl:hl + h, dl:dl + d]
        score_map = score_map[:, wl:wl + w, hl:hl + h, dl:dl + d]
    return label_map, score_map


def surface_metrics(pred, gt):
    """(hd95, asd) of medpy.metric.binary (medpy==0.4.0, environment.yml:301; absent here - restated from its published
    algorithm, PARITY UNPINNED against medpy itself; pinned only by analytic cases in tests/).  Border = X ^ erode(X)
    with the cross structuring element; distances from distance_transform_edt of the other border's complement."""
    from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure
    a, b = np.asarray(pred).astype(bool), np.asarray(gt).astype(bool)
    fp = generate_binary_structure(a.ndim, 1)

    def sd(u, v):
        ub, vb = u ^ binary_erosion(u, structure=fp, iterations=1), v ^ binary_erosion(v, structure=fp, iterations=1)
        return distance_transform_edt(~vb)[ub]
    d1, d2 = sd(a, b), sd(b, a)
    return float(np.percentile(np.hstack((d1, d2)), 95)), float(d1.mean())


# --------------------------------------------------------------------------
# R  revisiting loss: train_arco_2d.py:108-136
# --------------------------------------------------------------------------


def get_revisiting_loss(random_pool, rep_u, rep_u_teacher, topk=5):
    """train_arco_2d.py:126-136.  random_pool [K, n] unit rows; rep_* [b, D, *spatial]."""
    ru = F.normalize(rep_u.reshape(rep_u.shape[0], -1), dim=-1)
    rt = F.normalize(rep_u_teacher.reshape(rep_u_teacher.shape[0], -1), dim=-1)
    dist_t = 2 - 2 * ru @ random_pool.t()
    dist_q = 2 - 2 * rt @ random_pool.t()
    _, nn_index = dist_t.topk(topk, dim=1, largest=False)
    return (torch.gather(dist_q, 1, nn_index).sum(dim=1) / topk).mean()


def pool_enqueue(keys, queue, queue_ptr, K):
    """_dequeue_and_enqueue (train_arco_2d.py:108-119): queue[ptr:ptr+b] = keys; ptr = (ptr + b) % K (K % b == 0)."""
    b = keys.shape[0]
    ptr = int(queue_ptr)
    assert K % b == 0
    queue[ptr:ptr + b] = keys
    queue_ptr[0] = (ptr + b) % K


# --------------------------------------------------------------------------
# A  mixing strategies: augment.py:230-252,284-313; augment_3d.py:182-206,228-257
# --------------------------------------------------------------------------


def cutout_mask(img_size, ratio=2):
    """generate_cutout_mask (augment.py:230-244) / generate_cutout_mask_3d (augment_3d.py:182-198): ones with a zero
    box of area H*W/ratio (10 slices deep for volumes); draws from numpy's global RandomState: w, x_start, y_start
    [, z_start]."""
    area = img_size[0] * img_size[1] / ratio
    w = np.random.randint(img_size[1] / ratio + 1, img_size[1])
    h = np.round(area / w)
    x0 = np.random.randint(0, img_size[1] - w + 1)
    y0 = np.random.randint(0, img_size[0] - h + 1)
    mask = np.ones(img_size, dtype=np.float32)
    if len(img_size) == 3:
        z0 = np.random.randint(0, img_size[2] - 20 + 1)
        mask[int(y0):int(y0 + h), int(x0):int(x0 + w), int(z0):int(z0 + 10)] = 0
    else:
        mask[int(y0):int(y0 + h), int(x0):int(x0 + w)] = 0
    return mask


def class_mask(labels_map):
    """generate_class_mask (augment.py:247-252): a random half (torch.randperm on the CPU generator) of the labels
    present keeps its pixels."""
    labels = np.unique(labels_map)
    perm = torch.randperm(len(labels)).numpy()
    chosen = labels[perm][:len(labels) // 2]
    return np.isin(labels_map, chosen).astype(np.float32)


def generate_unsup_data(data, target, logits, mode='cutout'):
    """generate_unsup_data / _3d on numpy arrays: data [b,c,*sp] float32, target [b,*sp] int64, logits [b,*sp] float32
    -> (new_data, new_target int64, new_logits).  cutout also writes the -1s into `target` (augment.py:292)."""
    b = data.shape[0]
    sp = list(data.shape[2:])
    nd, nt, nl = [], [], []
    for i in range(b):
        j = (i + 1) % b
        if mode == 'cutout':
            m = cutout_mask(sp, ratio=2)
            target[i][m == 0] = -1
            nd.append(data[i] * m); nt.append(target[i].copy()); nl.append(logits[i] * m)
            continue
        if mode == 'cutmix':
            m = cutout_mask(sp)
        elif mode == 'classmix':
            m = class_mask(target[i])
        else:
            m = np.ones(sp, dtype=np.float32)
        nd.append(data[i] * m + data[j] * (1 - m))
        nt.append(target[i] * m + target[j] * (1 - m))
        nl.append(logits[i] * m + logits[j] * (1 - m))
    return np.stack(nd).astype(np.float32), np.stack(nt).astype(np.int64), np.stack(nl).astype(np.float32)


# --------------------------------------------------------------------------
# E  equivariance loss: tps/rand_tps.py:48-153, tps_stn_pytorch/tps_grid_gen.py:9-71,
#    tps/grid_sample.py:11-20, train_arco_2d.py:404-423
# --------------------------------------------------------------------------


def _tps_u(points, controls):
    """U(r) = 0.5 * r^2 * log(r^2) between two point sets, 0 where r == 0 (tps_grid_gen.py:9-21)."""
    d = points.view(-1, 1, 2) - controls.view(1, -1, 2)
    d2 = d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1]
    u = 0.5 * d2 * torch.log(d2)
    u[u != u] = 0
    return u


def tps_constants(height, width):
    """(target control points [25,2], inverse kernel [28,28], target coordinate representation [H*W, 28]):
    rand_tps.py:101-104 and TPSGridGen.__init__ (tps_grid_gen.py:25-57).  Pixel (y, x) -> (X, Y) in [-1, 1]."""
    ticks = torch.arange(-1.0, 1.00001, 2.0 / 4)
    tcp = torch.Tensor([(float(a), float(b)) for a in ticks for b in ticks])
    n = tcp.shape[0]
    k = torch.zeros(n + 3, n + 3)
    k[:n, :n] = _tps_u(tcp, tcp)
    k[:n, -3] = 1
    k[-3, :n] = 1
    k[:n, -2:] = tcp
    k[-2:, :n] = tcp.t()
    inv = torch.inverse(k)
    yy, xx = torch.meshgrid(torch.arange(height, dtype=torch.float32), torch.arange(width, dtype=torch.float32), indexing="ij")
    Y = yy.reshape(-1, 1) * 2 / (height - 1) - 1
    X = xx.reshape(-1, 1) * 2 / (width - 1) - 1
    coord = torch.cat([X, Y], dim=1)
    rep = torch.cat([_tps_u(coord, tcp), torch.ones(height * width, 1), coord], dim=1)
    return tcp, inv, rep


def rand_tps_source_points(tcp, batch, sigma, random_scale=(0.8, 1.2), translate=0.1, rotate=60, mirror=True):
    """RandTPS.reset_control_points up to the control points (rand_tps.py:110-139), mode 'affine'.  Generator calls,
    in order: one torch uniform_ of [batch,25,2]; four numpy uniforms of [batch] (angle, scale, shift x, shift y);
    one python random.randint(0, 1)."""
    src = tcp.unsqueeze(0).repeat(batch, 1, 1)
    src = src + torch.Tensor(src.size()).uniform_(-sigma, sigma)
    inv_scale = (1.0 / random_scale[1], 1.0 / random_scale[0])          # applied target -> source (rand_tps.py:88)
    ang = np.random.uniform(size=[batch], low=-rotate, high=rotate) / 180.0 * np.pi
    sc = np.random.uniform(size=[batch], low=inv_scale[0], high=inv_scale[1])
    sx = np.random.uniform(size=(batch,), low=-translate, high=translate).reshape(-1, 1)
    sy = np.random.uniform(size=(batch,), low=-translate, high=translate).reshape(-1, 1)
    half = np.float32(np.float32(2.0) / 2.0)                           # img_sz = 2.0
    cos_v = (sc * np.cos(ang)).reshape(-1, 1)
    sin_v = (sc * np.sin(ang)).reshape(-1, 1)
    theta = np.concatenate([cos_v, -sin_v, sx * half, sin_v, cos_v, sy * half], axis=1)
    t = torch.from_numpy(theta.reshape(-1, 2, 3).copy()).type(torch.FloatTensor).transpose(1, 2)    # [B,3,2]
    src = torch.matmul(torch.cat((src, torch.ones(batch, src.shape[1], 1)), dim=2), t)
    if mirror and random.randint(0, 1):
        src[:, :, 0] = -src[:, :, 0]
    return src


def tps_grid(src, inv, rep, height, width):
    """TPSGridGen.forward (tps_grid_gen.py:59-71): grid [B, H, W, 2] of source (x, y) in [-1, 1]."""
    Y = torch.cat([src, torch.zeros(src.shape[0], 3, 2)], 1)
    return torch.matmul(rep, torch.matmul(inv, Y)).view(-1, height, width, 2)


def grid_sample(x, grid, padding_mode='zeros'):
    """tps/grid_sample.py:11-12 (canvas=None): bilinear, align_corners=True."""
    return F.grid_sample(x, grid, mode='bilinear', padding_mode=padding_mode, align_corners=True)


def eqv_mask(labels, logits, weak_threshold):
    """train_arco_2d.py:406-410: 1 where the (pseudo-)label is foreground and its confidence >= weak_threshold."""
    m = torch.ones(labels.shape, dtype=torch.float32)
    m[labels == 0] = 0
    m[logits < weak_threshold] = 0
    return m.unsqueeze(1)


def eqv_loss(pred_tps, pred_tps_org, mask_tps):
    """train_arco_2d.py:419-423: per-image masked mean of KL(softmax(pred_tps_org) || softmax(pred_tps)) summed over
    classes, averaged over images."""
    kl = F.kl_div(F.log_softmax(pred_tps, dim=1), F.softmax(pred_tps_org, dim=1), reduction='none')
    per = (kl * mask_tps).flatten(1).sum(1) / (mask_tps.flatten(1).sum(1) + 1e-7)
    return per.mean()


# --------------------------------------------------------------------------
# A2  AdvMorph: adv_morph.py:184-207 (base grid), 260-307 (scaling and squaring, composition),
#     445-532 (Gaussian smoothing, DemonsCompose), 363-388 / 559-577 (forward, transform)
# --------------------------------------------------------------------------


def _morph_base_grid(b, h, w):
    y, x = torch.meshgrid([torch.linspace(-1, 1, h), torch.linspace(-1, 1, w)], indexing='ij')
    return torch.stack((x, y), 0).unsqueeze(0).repeat(b, 1, 1, 1)                     # [B, 2, H, W] = (x, y)


def _morph_gaussian(x, kernel_size=3, sigma=1):
    """gaussian_smooth + get_gaussian_kernel: depthwise filter, zero padding; kernel_size is raised to 2*int(3.5*sigma)+1."""
    import math
    if kernel_size < 2 * int(3.5 * sigma) + 1:
        kernel_size = 2 * int(3.5 * sigma) + 1
    c = torch.arange(kernel_size)
    xg = c.repeat(kernel_size).view(kernel_size, kernel_size)
    xy = torch.stack([xg, xg.t()], dim=-1).float()
    mean, var = (kernel_size - 1) / 2., sigma ** 2.
    k = (1. / (2. * math.pi * var)) * torch.exp(-torch.sum((xy - mean) ** 2., dim=-1) / (2 * var))
    k = (k / torch.sum(k)).view(1, 1, kernel_size, kernel_size).repeat(x.shape[1], 1, 1, 1)
    return F.conv2d(x, k, padding=kernel_size // 2, groups=x.shape[1])


def _morph_compose(f1, f2):
    return F.grid_sample(f1, f2.permute(0, 2, 3, 1), padding_mode='border', align_corners=True)


def adv_morph_grid(param, data_size, epsilon=1.5, num_steps=8):
    """AdvMorph.get_deformation_displacement_field(duv = epsilon * param): the sampling grid [B, 2, H, W], clamped to [-1, 1]."""
    b, h, w = data_size[0], data_size[2], data_size[3]
    base = _morph_base_grid(b, h, w)
    duv = _morph_gaussian(epsilon * param)
    duv = F.interpolate(duv, size=(h, w), mode='bilinear', align_corners=False)
    phi0 = base + duv / (2.0 ** num_steps)
    phi = phi0
    for _ in range(num_steps):
        phi = _morph_compose(phi, phi)
    # reference quirk: integrate_by_add (adv_morph.py:246-259) adds IN PLACE, so the `grid_wh` that
    # vectorFieldExponentiation2D subtracts at the end (:294) is the initial phi = grid + duv / 2^n, not the identity grid
    off = phi - phi0
    comp = _morph_compose(base, off + base)
    comp = _morph_gaussian(comp - base) + base
    return torch.clamp(comp, -1, 1)


def adv_morph_forward(data, param, epsilon=1.5):
    grid = adv_morph_grid(param, list(data.shape), epsilon)
    return F.grid_sample(data, grid.permute(0, 2, 3, 1), mode='bilinear', align_corners=True)


# --------------------------------------------------------------------------
# A3  photometric augmentation of batch_transform (augment.py:148-225): torchvision 0.13.1's ColorJitter on PIL images
#     (transforms.ColorJitter.forward / functional_pil.adjust_brightness|contrast|saturation|hue, published source) and
#     Pillow's ImageEnhance / ImagingConvert (L, HSV) / ImagingGaussianBlur, restated in numpy INTEGER arithmetic.
#     Pinned bit for bit against Pillow itself (tests/golden/g12_jitter.npz).
# --------------------------------------------------------------------------


def q8(x):
    """to_pil_image of a float tensor: pic.mul(255).byte() (truncation)."""
    return np.clip(np.asarray(x, np.float32) * np.float32(255.0), 0, 255).astype(np.uint8)


def _blend8(d, v, f):
    t = d.astype(np.float32) + np.float32(f) * (v.astype(np.float32) - d.astype(np.float32))
    return np.clip(t, 0, 255).astype(np.uint8)


def _lum8(a):
    return ((a[0].astype(np.int64) * 19595 + a[1].astype(np.int64) * 38470 + a[2].astype(np.int64) * 7471 + 0x8000) >> 16).astype(np.uint8)


def _rgb2hsv8(a):
    f32, f64 = np.float32, np.float64
    r, g, b = (a[i].astype(np.int32) for i in range(3))
    maxc, minc = np.maximum(np.maximum(r, g), b), np.minimum(np.minimum(r, g), b)
    cr = (maxc - minc).astype(f32)
    with np.errstate(all='ignore'):
        s = cr / maxc.astype(f32)
        rc, gc, bc = ((maxc - c).astype(f32) / cr for c in (r, g, b))
        h = np.where(r == maxc, bc.astype(f64) - gc.astype(f64),
                     np.where(g == maxc, 2.0 + rc.astype(f64) - bc.astype(f64), 4.0 + gc.astype(f64) - rc.astype(f64))).astype(f32)
        t = np.fmod(h.astype(f64) / 6.0 + 1.0, 1.0).astype(f32)
        uh = np.nan_to_num(t.astype(f64) * 255.0).astype(np.int32)
        us = np.nan_to_num(s.astype(f64) * 255.0).astype(np.int32)
    grey = maxc == minc
    return (np.where(grey, 0, np.clip(uh, 0, 255)).astype(np.uint8), np.where(grey, 0, np.clip(us, 0, 255)).astype(np.uint8),
            maxc.astype(np.uint8))


def _hsv2rgb8(h, s, v):
    f32 = np.float32
    vi = v.astype(np.float64)
    fs = s.astype(f32) / f32(255.0)
    hh = h.astype(f32) * f32(6.0) / f32(255.0)
    fi = np.floor(hh)
    f = hh - fi
    rnd = lambda t: np.clip(np.rint(t), 0, 255).astype(np.int32)
    p = rnd(vi * (f32(1.0) - fs).astype(np.float64))
    q = rnd(vi * (f32(1.0) - fs * f).astype(np.float64))
    t = rnd(vi * (f32(1.0) - fs * (f32(1.0) - f)).astype(np.float64))
    i = fi.astype(np.int32) % 6
    vv = v.astype(np.int32)
    out = np.stack([np.choose(i, [vv, q, p, p, t, vv]), np.choose(i, [t, vv, vv, q, p, p]), np.choose(i, [p, p, t, vv, vv, q])])
    return np.where((s == 0)[None], vv[None].repeat(3, 0), out).astype(np.uint8)


def color_jitter_u8(a, order, factors):
    """a: uint8 [C, H, W] (C = 1: mode 'L', 3: 'RGB'); order: fn_idx of ColorJitter.get_params; factors = (b, c, s, h)."""
    a = a.copy()
    C = a.shape[0]
    for op in order:
        f = factors[op]
        if op == 0:
            a = _blend8(np.zeros_like(a), a, f)
        elif op == 1:
            lum = _lum8(a) if C == 3 else a[0]
            mean = int(lum.astype(np.float64).mean() + 0.5)
            a = _blend8(np.full_like(a, mean), a, f)
        elif op == 2 and C == 3:
            a = _blend8(_lum8(a)[None].repeat(3, 0), a, f)
        elif op == 3 and C == 3:
            h, s, v = _rgb2hsv8(a)
            h = (h.astype(np.int32) + (int(np.float32(f) * np.float32(255.0)) & 0xff)).astype(np.uint8)      # uint8 wrap
            a = _hsv2rgb8(h, s, v)
    return a


def gaussian_blur_radius(sigma, passes=3):
    """Pillow's _gaussian_blur_radius: fractional box radius of the 3-pass box approximation."""
    import math
    sigma2 = float(np.float32(sigma)) * float(np.float32(sigma)) / passes
    L = math.sqrt(12.0 * sigma2 + 1.0)
    l = math.floor((L - 1.0) / 2.0)
    a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2)
    a /= 6 * (sigma2 - (l + 1) * (l + 1))
    return float(np.float32(l + a))


def _box_pass(x, fr):
    r = int(fr)
    ww = int(np.float32(1 << 24) / (np.float32(fr) * np.float32(2.0) + np.float32(1.0)))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    n = x.shape[-1]
    idx = np.arange(n)
    acc = np.zeros(x.shape, np.int64)
    for d in range(-r, r + 1):
        acc += x[..., np.clip(idx + d, 0, n - 1)].astype(np.int64) * ww
    acc += (x[..., np.clip(idx - r - 1, 0, n - 1)].astype(np.int64) + x[..., np.clip(idx + r + 1, 0, n - 1)].astype(np.int64)) * fw
    return ((acc + (1 << 23)) >> 24).astype(np.uint8)


def gaussian_blur_u8(a, sigma):
    """ImageFilter.GaussianBlur(radius=sigma) on uint8 [C, H, W]: three horizontal box passes, then three vertical."""
    fr = gaussian_blur_radius(sigma)
    x = a
    for _ in range(3):
        x = _box_pass(x, fr)
    x = np.swapaxes(x, -1, -2)
    for _ in range(3):
        x = _box_pass(x, fr)
    return np.ascontiguousarray(np.swapaxes(x, -1, -2))


def batch_transform_params(n_images, apply_augmentation, scale_size=(1.0, 1.0)):
    """The generator draws of batch_transform (augment.py:255-281 -> transform :131-225 -> torchvision ColorJitter.get_params),
    in order, for same-size crops: per image python random.uniform(scale), [torch.rand(1) -> randperm(4) + 4 uniform_],
    [torch.rand(1) -> python random.uniform(0.15, 1.15)]; after the loop torch.rand(1) for AdvMorph (drawn even when
    apply_augmentation is False: `torch.rand(1) > 0.5 and apply_augmentation`).  Returns (per-image dicts, morph flag)."""
    out = []
    for _ in range(n_images):
        random.uniform(scale_size[0], scale_size[1])
        d = dict(order=None, factors=None, sigma=None)
        if apply_augmentation:
            if float(torch.rand(1)) > 0.5:
                d["order"] = [int(v) for v in torch.randperm(4)]
                d["factors"] = tuple(float(torch.empty(1).uniform_(lo, hi)) for lo, hi in ((0.75, 1.25), (0.75, 1.25), (0.75, 1.25), (-0.25, 0.25)))
            if float(torch.rand(1)) > 0.5:
                d["sigma"] = random.uniform(0.15, 1.15)
        out.append(d)
    morph = bool(float(torch.rand(1)) > 0.5) and bool(apply_augmentation)
    return out, morph


# --------------------------------------------------------------------------
# S1  stage-1 pre-training: ISD.forward of model_2D.py:215-305 and the losses / optimizer of
#     pretrain_2D.py:235-252, functional over plain state dicts (dropout off)
# --------------------------------------------------------------------------
def isd_compute_logits(z_anchor, z_positive, temp_fac):
    """model_2D.py:320-331"""
    return torch.matmul(F.normalize(z_anchor, dim=1), F.normalize(z_positive, dim=1).T) / temp_fac


def isd_mlp(x, hd, pooling=1):
    """MLP.forward / MLP_3d.forward, model_2D.py:105-112 (hd: f1.weight / f1.bias / f2.weight / f2.bias)."""
    y = (F.adaptive_avg_pool2d if x.dim() == 4 else F.adaptive_avg_pool3d)(x, pooling).reshape(x.shape[0], -1)
    return F.linear(F.linear(y, hd["f1.weight"], hd["f1.bias"]), hd["f2.weight"], hd["f2.bias"])


def isd_projection(x, hd, pool):
    """ProjectionHead 'convmlp', model_2D.py:66-85 (hd: proj.1.* / proj.2.*)."""
    conv = F.conv2d if x.dim() == 4 else F.conv3d
    y = (F.adaptive_avg_pool2d if x.dim() == 4 else F.adaptive_avg_pool3d)(x, pool)
    return conv(conv(y, hd["proj.1.weight"], hd["proj.1.bias"]), hd["proj.2.weight"], hd["proj.2.bias"])


def isd_stage1_forward(st, im_q, im_k, Ts, Tt, patch_size, pool, K, m=0.99, net=None):
    """ISD.forward / ISD_3d.forward (model_3D.py:309-403; net = the V-Net forward, 5-D inputs) in training mode.  st: dict with 'model' / 'ema_model' (U-Net state dicts; the student's parameters
    may require grad), 'q_latent_head', 'k_latent_head', 'latent_predictor' ({'0.weight', ...}), 'q_outputs_head',
    'k_outputs_head', 'outputs_predictor', 'queue' [K, F], 'queue_mask' [K, 49, C*pool^2], 'queue_ptr', 'mask_queue_ptr'.
    Teacher parameters, BN running statistics of both nets and the queues are updated in place, as the module does.
    The shuffle permutation comes from the torch CPU generator (get_shuffle_ids, :308-318)."""
    b = im_q.shape[0]
    if net is None:
        net = lambda x, sd: unet_forward(x, sd, train=True, track=True)
    conv = F.conv2d if im_q.dim() == 4 else F.conv3d
    outputs, latent, _ = net(im_q, st["model"])
    with torch.no_grad():
        ema_output_tmp, _, _ = net(im_k, st["ema_model"])
        # _momentum_update_key_encoder (:176-182): parameters only, three module pairs
        for q_name, k_name in (("model", "ema_model"), ("q_outputs_head", "k_outputs_head"), ("q_latent_head", "k_latent_head")):
            for key, q in st[q_name].items():
                if "running_" in key or "num_batches" in key:
                    continue
                rg = st[k_name][key].requires_grad          # param_k.data = ...: the Parameter (and its requires_grad) stays
                st[k_name][key] = (st[k_name][key].detach() * m + q.detach() * (1.0 - m)).requires_grad_(rg)
        fwd = torch.randperm(b).long()
        bwd = torch.zeros(b).long()
        bwd.index_copy_(0, fwd, torch.arange(b).long())
        ema_output, ema_latent, _ = net(im_k[fwd], st["ema_model"])
        ema_latent, ema_output = ema_latent[bwd], ema_output[bwd]
    queue = st["queue"].clone()
    queue_mask = st["queue_mask"].clone().transpose(0, 1).contiguous()
    step = patch_size // 2
    stu, tea = [], []
    import itertools
    for org in itertools.product(*[range(0, outputs.shape[2 + d] - patch_size + 1, step) for d in range(outputs.dim() - 2)]):
        win = (slice(None), slice(None)) + tuple(slice(o, o + patch_size) for o in org)
        y = isd_projection(outputs[win], st["q_outputs_head"], pool)
        y = conv(conv(y, st["outputs_predictor"]["0.weight"], st["outputs_predictor"]["0.bias"]),
                 st["outputs_predictor"]["1.weight"], st["outputs_predictor"]["1.bias"])
        stu.append(y)
        # NOT under no_grad in the reference (:268): the key heads sit in the autograd graph of the KLD *targets*
        tea.append(isd_projection(ema_output[win], st["k_outputs_head"], pool))
    shp = tuple(stu[0].shape[1:])
    stu = torch.cat(stu).reshape(b, -1, *shp).contiguous()
    tea = torch.cat(tea).reshape(b, -1, *shp).contiguous()
    lat_k = isd_mlp(ema_latent, st["k_latent_head"])                 # (:282, with gradient, see above)
    lat_q = isd_mlp(latent, st["q_latent_head"])
    lat_q = F.linear(F.linear(lat_q, st["latent_predictor"]["0.weight"], st["latent_predictor"]["0.bias"]),
                     st["latent_predictor"]["1.weight"], st["latent_predictor"]["1.bias"])
    tea_tmp = tea.reshape(tea.shape[0], tea.shape[1], -1).contiguous()
    stu = stu.reshape((stu.shape[1], b, -1)).contiguous()
    tea = tea.reshape((tea.shape[1], b, -1)).contiguous()
    stu = stu.reshape(-1, stu.shape[0]).contiguous()
    tea = tea.reshape(-1, tea.shape[0]).contiguous()
    queue_mask = queue_mask.reshape(-1, queue_mask.shape[0]).contiguous()
    out = (outputs, ema_output_tmp, isd_compute_logits(lat_k, queue, Tt), isd_compute_logits(lat_q, queue, Ts),
           isd_compute_logits(tea, queue_mask, Tt), isd_compute_logits(stu, queue_mask, Ts))
    for keys, qn, pn in ((lat_k, "queue", "queue_ptr"), (tea_tmp, "queue_mask", "mask_queue_ptr")):
        ptr = int(st[pn])
        assert K % b == 0
        st[qn][ptr:ptr + b] = keys.detach()
        st[pn][0] = (ptr + b) % K
    return out


def kld_batchmean(inputs, targets):
    """KLD of pretrain_2D.py:99-103"""
    return F.kl_div(F.log_softmax(inputs, dim=1), F.softmax(targets, dim=1), reduction='batchmean')


def sgd_momentum_step(p, g, buf, lr, momentum=0.9, wd=1e-4):
    """torch.optim.SGD(momentum, weight_decay), nesterov off (pretrain_2D.py:193-195). buf None on the first step."""
    g = g + wd * p
    buf = g.clone() if buf is None else momentum * buf + g
    return p - lr * buf, buf


# every parameter with requires_grad goes to the optimizer (pretrain_2D.py:192) - that includes the KEY heads: they are
# EMA-updated from the query heads inside the forward AND trained through the KLD targets (F.kl_div differentiates its
# target), weight decay and momentum included.  Only ema_model is frozen (create_model(ema=True) detaches it).
STAGE1_TRAINED = ("model", "k_latent_head", "q_latent_head", "latent_predictor", "k_outputs_head", "q_outputs_head", "outputs_predictor")


def isd_stage1_step(st, bufs, im_q, im_k, label, labeled_bs, n_cls, lr, Ts, Tt, patch_size, pool, K, k1=1.0, k2=1.0, net=None,
                    sup_scale=1.0):
    """One iteration of pretrain_2D.py:235-252 with train_encoder = train_decoder = 1: forward, supervised CE + Dice on
    the labeled part, the two KLD terms, SGD(momentum) on every trainable tensor.  Returns the loss terms and the
    forward's outputs; st / bufs (momentum buffers by (group, key)) are updated in place."""
    trained = [(g, k) for g in STAGE1_TRAINED for k in st[g] if "running_" not in k and "num_batches" not in k]
    for g, k in trained:
        st[g][k] = st[g][k].detach().requires_grad_(True)
    out = isd_stage1_forward(st, im_q, im_k, Ts, Tt, patch_size, pool, K, net=net)
    outputs, _, ema_latent_logits, latent_logits, ema_output_logits, output_logits = out
    ce, dice = supervised_loss(outputs[:labeled_bs], label[:labeled_bs], n_cls)
    l_lat = kld_batchmean(latent_logits, ema_latent_logits)
    l_out = kld_batchmean(output_logits, ema_output_logits)
    loss = sup_scale * (dice + ce) + k1 * l_lat + k2 * l_out          # (pretrain_3D.py:218: sup_scale = 0.5)
    grads = torch.autograd.grad(loss, [st[g][k] for g, k in trained], allow_unused=True)
    with torch.no_grad():
        for (g, k), gr in zip(trained, grads):
            if gr is None:
                st[g][k] = st[g][k].detach()
                continue
            p, bufs[(g, k)] = sgd_momentum_step(st[g][k].detach(), gr, bufs.get((g, k)), lr)
            st[g][k] = p
    return dict(loss=float(loss.detach()), ce=float(ce.detach()), dice=float(dice.detach()), latent=float(l_lat.detach()), output=float(l_out.detach())), out, \
        {gk: gr for gk, gr in zip(trained, grads)}
