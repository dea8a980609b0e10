"""Stage-1 (pre-training) forward shared by ISD (model_2D.py:215-305) and ISD_3d (model_3D.py:309-403): student and
momentum-teacher passes through the HIP U-Net / V-Net, latent and patch-wise output embeddings compared with two queues.
The heads are a few 2- to 256-wide layers on pooled maps and run as plain tensor ops around the two networks."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def get_shuffle_ids(bsz, device=None):
    """ShuffleBN permutation and its inverse (model_2D.py:308-318).  Drawn from the torch CPU generator, like the
    reference's torch.randperm(bsz).long().cuda()."""
    forward_inds = torch.randperm(bsz).long()
    backward_inds = torch.zeros(bsz).long()
    backward_inds.index_copy_(0, forward_inds, torch.arange(bsz).long())
    return forward_inds.to(device), backward_inds.to(device)


def compute_logits(z_anchor, z_positive, temp_fac):
    """Cosine similarities of every anchor row with every queue row over the temperature (model_2D.py:320-331)."""
    z_anchor = nn.functional.normalize(z_anchor, dim=1)
    z_positive = nn.functional.normalize(z_positive, dim=1)
    return torch.matmul(z_anchor, z_positive.T) / temp_fac


def dequeue_and_enqueue(K, keys, queue, queue_ptr):
    """model_2D.py:201-212: keys overwrite queue[ptr : ptr + batch], the pointer wraps at K."""
    batch_size = keys.shape[0]
    ptr = int(queue_ptr)
    assert K % batch_size == 0
    queue[ptr:ptr + batch_size] = keys
    queue_ptr[0] = (ptr + batch_size) % K


def _unwrap(m):
    return m.module if hasattr(m, "module") else m


def _pool_size(pool_layer):
    o = pool_layer.output_size
    return o if isinstance(o, int) else o[0]


def patch_heads(x, patch, head, predictor):
    """predictor(head(x[..., i:i+p, j:j+p(, k:k+p)])) for every patch of the p/2-stride grid, in the reference's loop
    order (first spatial axis outermost; model_2D.py:263-268, model_3D.py:355-359).  The head is adaptive average pooling
    followed by 1x1 convolutions: when the pooling windows tile the patch stride, the pooled grid of the WHOLE map is
    computed once (avg_pool with the same windows), the pointwise layers run once on it, and a patch is a window of that
    grid - the same values, hundreds of small launches less (700 patches per volume in 3-D); otherwise the patches are
    taken one by one like the reference."""
    nd = x.dim() - 2
    conv = F.conv2d if nd == 2 else F.conv3d
    proj = _unwrap(head).proj
    pool_out = _pool_size(proj[0])
    layers = list(proj[1:]) + (list(_unwrap(predictor)) if predictor is not None else [])
    step = patch // 2
    starts = [range(0, x.shape[2 + d] - patch + 1, step) for d in range(nd)]
    import itertools
    if patch % pool_out == 0 and step % (patch // pool_out) == 0:
        w = patch // pool_out
        crop = tuple(slice(0, (x.shape[2 + d] // w) * w) for d in range(nd))
        g = (F.avg_pool2d if nd == 2 else F.avg_pool3d)(x[(slice(None), slice(None)) + crop], w)
        for l in layers:
            g = conv(g, l.weight, l.bias)
        return [g[(slice(None), slice(None)) + tuple(slice(o // w, o // w + pool_out) for o in org)]
                for org in itertools.product(*starts)]
    out = []
    pool = F.adaptive_avg_pool2d if nd == 2 else F.adaptive_avg_pool3d
    for org in itertools.product(*starts):
        y = pool(x[(slice(None), slice(None)) + tuple(slice(o, o + patch) for o in org)], pool_out)
        for l in layers:
            y = conv(y, l.weight, l.bias)
        out.append(y)
    return out


def mlp(head, x):
    """MLP.forward / MLP_3d.forward (model_2D.py:105-112): global average pooling, two linear layers."""
    m = _unwrap(head)
    y = m.gap(x).reshape(x.shape[0], -1)
    return F.linear(F.linear(y, m.f1.weight, m.f1.bias), m.f2.weight, m.f2.bias)


def isd_forward(isd, im_q, im_k=None, Ts=None, Tt=None):
    """Returns (outputs, ema_output, ema_latent_logits, latent_logits, ema_output_logits, output_logits) in training mode,
    (outputs, latent) in eval mode.  The reshapes below are the reference's, view for view (they interleave patch and
    batch indices; the queues are stored in the layout they produce).  The KEY heads are not under no_grad in the
    reference (:268,282): they are EMA-updated and, being part of the KLD targets' graph, trained by the optimizer too."""
    Ts = Ts if Ts else isd.Ts
    Tt = Tt if Tt else isd.Tt
    batch_size = im_q.shape[0]
    if not isd.training:
        outputs, latent_vector, _ = isd.model(im_q)
        return outputs, latent_vector
    outputs, latent_vector, _ = isd.model(im_q)
    with torch.no_grad():
        ema_output_tmp, _, _ = isd.ema_model(im_k)
        isd._momentum_update_key_encoder()
        shuffle_ids, reverse_ids = get_shuffle_ids(im_k.shape[0], im_k.device)          # ShuffleBN
        ema_output, ema_latent_vector, _ = isd.ema_model(im_k[shuffle_ids])
        ema_latent_vector = ema_latent_vector[reverse_ids]
        ema_output = ema_output[reverse_ids]
    queue = isd.queue.clone().detach()
    queue_mask = isd.queue_mask.clone().detach().transpose(0, 1).contiguous()
    stu = patch_heads(outputs, isd.patch_size, isd.q_outputs_head, isd.outputs_predictor)
    tea = patch_heads(ema_output, isd.patch_size, isd.k_outputs_head, None)
    shp = tuple(stu[0].shape[1:])
    stu = torch.cat(stu).reshape(batch_size, -1, *shp).contiguous()
    tea = torch.cat(tea).reshape(batch_size, -1, *shp).contiguous()
    lat_k = mlp(isd.k_latent_head, ema_latent_vector)
    lat_q = mlp(isd.q_latent_head, latent_vector)
    for l in _unwrap(isd.latent_predictor):
        lat_q = F.linear(lat_q, l.weight, l.bias)
    tea_tmp = tea.reshape(tea.shape[0], tea.shape[1], -1).contiguous()
    stu = stu.reshape((stu.shape[1], batch_size, -1)).contiguous()
    tea = tea.reshape((tea.shape[1], batch_size, -1)).contiguous()
    stu = stu.reshape(-1, stu.shape[0]).contiguous()
    tea = tea.reshape(-1, tea.shape[0]).contiguous()
    queue_mask = queue_mask.reshape(-1, queue_mask.shape[0]).contiguous()
    ema_latent_logits = compute_logits(lat_k, queue, Tt)
    latent_logits = compute_logits(lat_q, queue, Ts)
    ema_output_logits = compute_logits(tea, queue_mask, Tt)
    output_logits = compute_logits(stu, queue_mask, Ts)
    with torch.no_grad():
        dequeue_and_enqueue(isd.K, lat_k, isd.queue, isd.queue_ptr)
        dequeue_and_enqueue(isd.K, tea_tmp, isd.queue_mask, isd.mask_queue_ptr)
    return outputs, ema_output_tmp, ema_latent_logits, latent_logits, ema_output_logits, output_logits
