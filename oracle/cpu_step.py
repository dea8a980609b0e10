"""CPU oracle of one hot-path training step (same op sequence as arco_amd.train_arco_2d.ArcoStep2D.step),
built from the restated pieces in arco_oracle.py.  TEST INFRASTRUCTURE: used by bench.py's
cpu_baseline leg (timed on the GPU box's host cores) and by tests as the checker."""
import time

import numpy as np
import torch
import torch.nn.functional as F

import arco_oracle as orc


def make_state(unet_sd, fe_sd, qrep_w):
    st = dict(
        student={k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in unet_sd.items()},
        teacher={k: v.clone() for k, v in unet_sd.items()},
        q_fe={k: v.clone().requires_grad_(True) for k, v in fe_sd.items()},
        k_fe={k: v.clone() for k, v in fe_sd.items()},
        q_rep=[w.clone().requires_grad_(True) for w in qrep_w],
        mom={}, it=0)
    return st


def batch_transform(data, label, logits, apply_augmentation, morph_velocity=None):
    """augment.batch_transform (augment.py:255-281) for the trainers' same-size configuration, from the oracle's pieces:
    generator draws (orc.batch_transform_params), 8-bit round trip, ColorJitter / GaussianBlur in Pillow's integer
    arithmetic, AdvMorph with the velocity field supplied by `morph_velocity(B, h, w)` (the reference draws it on the
    device generator, which has no CPU counterpart)."""
    params, morph = orc.batch_transform_params(int(data.shape[0]), apply_augmentation)
    out = []
    for k, p in enumerate(params):
        a = orc.q8(data[k].numpy())
        if p["order"] is not None:
            a = orc.color_jitter_u8(a, p["order"], p["factors"])
        if p["sigma"] is not None:
            a = orc.gaussian_blur_u8(a, p["sigma"])
        out.append(torch.from_numpy(a.astype(np.float32) / np.float32(255.0)))
    data_t = torch.stack(out)
    logits_t = torch.from_numpy(orc.q8(logits.numpy()).astype(np.float32) / np.float32(255.0))
    if morph:
        B, _, H, W = data_t.shape
        v = morph_velocity(B, W // 8, W // 8)
        v = v / (v.reshape(B, -1).norm(dim=1).view(-1, 1, 1, 1) + 1e-20)
        data_t = orc.adv_morph_forward(data_t, v)
    return data_t, label, logits_t


def step(st, l_data, l_label, u_data, memobank, ptrs, qsize, n_cls=4, alpha_t=20.0, k1=0.01, lr=0.01,
         delta_n=0.97, func='smc', nq=256, nn_=512, k2=0.0, tps_sigma=0.01, weak_threshold=0.7, apply_aug='none', pool=None, k4=1.0, topk=5,
         bt=False, morph_velocity=None, force=None):
    """force (round 6, as cpu_step3d.step): the step's gradient-free DECISION inputs as another implementation computed them
    (pseudo_labels / pseudo_logits before the mixing, low / high entropy masks, teacher probabilities).  Each is compared with this
    function's own value (agreement counts in st["agree"]) and then used in its place, so that every threshold / arg-max decision of
    the step - 10^5 pixels at 256 x 256, each a hair-trigger for the sampler arguments and through them the CPU generator stream -
    is taken on identical numbers and everything continuous downstream can be compared strictly."""
    agree = {}

    def forced(name, mine):
        if force is None or name not in force:
            return mine
        other = force[name].to(mine.dtype)
        if mine.is_floating_point():
            agree[name] = dict(max_abs_diff=float((other - mine).abs().max()), n_diff=int((other != mine).sum()), n=mine.numel())
        else:
            agree[name] = dict(n_diff=int((other != mine).sum()), n=mine.numel())
        return other
    st["agree"] = agree
    with torch.no_grad():
        pred_u0, _, _ = orc.unet_forward(u_data, st["teacher"], track=True)
        pseudo_logits, pseudo_labels = torch.max(torch.softmax(pred_u0, 1), 1)
        pseudo_logits, pseudo_labels = forced("pseudo_logits", pseudo_logits), forced("pseudo_labels", pseudo_labels)
        if apply_aug in ('cutout', 'cutmix', 'classmix'):      # train_arco_2d.py:296-297 (generate_unsup_data)
            mixed = orc.generate_unsup_data(u_data.numpy(), pseudo_labels.numpy().copy(), pseudo_logits.numpy(), apply_aug)
            u_data, pseudo_labels, pseudo_logits = (torch.from_numpy(v) for v in mixed)
        cj2_l, cj2_u = l_data, None
        if bt:       # train_arco_2d.py:287-304: batch_transform x2 on the labeled (no augmentation), x2 on the mixed unlabeled batch
            batch_transform(l_data, l_label, torch.ones(l_label.shape), False)
            cj2_l, _, _ = batch_transform(l_data, l_label, torch.ones(l_label.shape), False)
        if bt:
            cj2_u, _, _ = batch_transform(u_data, pseudo_labels, pseudo_logits, True, morph_velocity)
            u_data, pseudo_labels, pseudo_logits = batch_transform(u_data, pseudo_labels, pseudo_logits, True, morph_velocity)
        for k in st["k_fe"]:
            st["k_fe"][k] = st["k_fe"][k] * 0.99 + st["q_fe"][k].detach() * 0.01
    # train-mode forwards in the reference's order (train_arco_2d.py:310-315); track=True: the BatchNorm running
    # statistics receive their momentum updates in that order (they do not commute)
    pred_l, _, l_fm = orc.unet_forward(l_data, st["student"], track=True)
    with torch.no_grad():
        orc.unet_forward(cj2_l, st["student"], track=True)           # images_cj2_l (:311): running statistics only
    pred_u, _, u_fm = orc.unet_forward(u_data, st["student"], track=True)
    with torch.no_grad():
        pred_l_t, _, l_fm_t = orc.unet_forward(l_data, st["teacher"], track=True)
        pred_u_t, _, u_fm_t = orc.unet_forward(u_data, st["teacher"], track=True)
        rep_t = orc.feature_extractor_forward([torch.cat((a, b)) for a, b in zip(l_fm_t, u_fm_t)], st["k_fe"])
    feat = orc.feature_extractor_forward([torch.cat((a, b)) for a, b in zip(l_fm, u_fm)], st["q_fe"])
    rep = F.conv2d(F.conv2d(feat, st["q_rep"][0]), st["q_rep"][1])
    with torch.no_grad():
        label_l = orc.label_onehot(l_label, n_cls).long()
        label_u = orc.label_onehot(pseudo_labels, n_cls).long()
        low, high, _ = orc.entropy_masks(pred_u, l_label, pseudo_labels, alpha_t)
        pl, pu = torch.softmax(pred_l_t, 1), torch.softmax(pred_u_t, 1)
        low, high, pl, pu = forced("low", low), forced("high", high), forced("prob_l_t", pl), forced("prob_u_t", pu)
    _, reco = orc.compute_contra_memobank_loss(rep, label_l, label_u, pl, pu, low, high, memobank, ptrs, qsize, rep_t,
                                               delta_n=delta_n, func=func, num_queries=nq, num_negatives=nn_)
    ce, dice = orc.supervised_loss(pred_l, l_label, n_cls)
    unsup = orc.compute_unsupervised_loss(pred_u, pseudo_labels, pseudo_logits, 0.97)
    loss = k1 * reco + unsup + (ce + dice)
    loss_q = None
    if pool is not None:         # revisiting loss (:334) and pool update (:398-400); pool = dict(rows [K, n], ptr)
        nb = int(l_data.shape[0])
        with torch.no_grad():
            loss_q = orc.get_revisiting_loss(pool["rows"], rep[nb:].detach(), rep_t[nb:], topk)
            orc.pool_enqueue(F.normalize(rep_t[nb:].reshape(rep_t.shape[0] - nb, -1), dim=-1), pool["rows"], pool["ptr"],
                             int(pool["rows"].shape[0]))
        loss = loss + k4 * loss_q
    eqv = None
    if k2 != 0:      # train_arco_2d.py:404-423; the warp is drawn after the samplers (same torch-generator order)
        # the trainer builds RandTPS(patch_size[0], patch_size[1]) = RandTPS(width, height) (:255-256): the grid has
        # HEIGHT patch_size[1] and WIDTH patch_size[0], so a non-square patch comes out of the warp transposed in size
        H, W = int(l_data.shape[3]), int(l_data.shape[2])
        if "tps" not in st:
            st["tps"] = orc.tps_constants(H, W)
        tcp, inv, rep = st["tps"]
        nb2 = int(l_data.shape[0]) + int(u_data.shape[0])
        with torch.no_grad():
            labels_all = torch.cat((l_label, pseudo_labels))
            logits_all = torch.cat((torch.ones(l_label.shape), pseudo_logits))
            mask = orc.eqv_mask(labels_all, logits_all, weak_threshold)
            grid = orc.tps_grid(orc.rand_tps_source_points(tcp, nb2, tps_sigma), inv, rep, H, W)
            images_tps = orc.grid_sample(torch.cat((cj2_l, u_data if cj2_u is None else cj2_u)), grid)
            mask_tps = orc.grid_sample(mask, grid)
            org = orc.grid_sample(torch.cat((pred_l.detach(), pred_u.detach())), grid)
        pred_tps = orc.unet_forward(images_tps, st["student"], track=True)[0]
        eqv = orc.eqv_loss(pred_tps, org, mask_tps)
        loss = loss + k2 * eqv
    leaves = [v for v in st["student"].values() if v.requires_grad] + list(st["q_rep"]) + list(st["q_fe"].values())
    grads = torch.autograd.grad(loss, leaves, allow_unused=True)
    with torch.no_grad():
        for i, (p, g) in enumerate(zip(leaves, grads)):
            if g is None:
                continue
            newp, st["mom"][i] = orc.sgd_nesterov_step(p, g, st["mom"].get(i), lr)
            p.copy_(newp)
        for k, v in st["student"].items():
            if v.requires_grad:
                st["teacher"][k] = st["teacher"][k] * 0.99 + v.detach() * 0.01
    st["it"] += 1
    st["last_terms"] = dict(ce=float(ce.detach()), dice=float(dice.detach()), unsup=float(unsup.detach()), reco=float(reco.detach()))
    if eqv is not None:
        st["last_terms"]["eqv"] = float(eqv.detach())
    if loss_q is not None:
        st["last_terms"]["loss_q"] = float(loss_q)
    return float(loss.detach()), float(reco.detach())


def timed_sample(b=2, patch=(256, 256), n_cls=4, seed=1337, qsize=4096, steps=1, k2=0.0, bt=False):
    """`steps` chained full CPU steps at --batch_size b (fresh synthetic batches, the banks fill as they would in
    training, so the later steps run the real grid samplers); returns (mean seconds per step, threads)."""
    import fixture_inputs as fx
    torch.manual_seed(seed)
    st = make_state(fx.unet_state(21, 1, n_cls), fx.fe_state(31), [fx.fe_state(32)["fea4.weight"], fx.fe_state(33)["fea4.weight"]])
    rs = np.random.RandomState(seed)
    bank, ptr, qs = fx.fresh_bank(n_cls, 496, qsize, 'zeros')
    total = 0.0
    for _ in range(steps):
        l = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32))
        u = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32))
        lab = torch.from_numpy(fx.blob_labels(rs, b, patch, n_cls))
        t0 = time.time()
        vel = lambda B, h, w: torch.from_numpy(rs.uniform(-1, 1, size=(B, 2, h, w)).astype(np.float32))
        step(st, l, lab, u, bank, ptr, qs, n_cls, apply_aug='cutmix', k2=k2, bt=bt, morph_velocity=vel)      # trainer defaults: cutmix, k2 = 1, batch_transform
        total += time.time() - t0
    return total / steps, torch.get_num_threads()
