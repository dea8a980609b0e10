"""CPU oracle of one 3-D hot-path training step: a restatement of the loop body of the reference's
code/train_arco_3d.py:255-400 from the pieces of arco_oracle.py that are pinned to the imported reference (g2 5-D loss,
g3 V-Net / FeatureExtractor_3d, g4 glue, g5 slice-wise TPS, g7 3-D mixing).

TEST INFRASTRUCTURE: imported by tests/ only (tests/test_step3d_parity_gpu.py), as the checker of
arco_amd.train_arco_3d.ArcoStep3D.step.  Dropout3d is off on both sides (the product's masks come from its own
counter hash)."""
import numpy as np
import torch
import torch.nn.functional as F

import arco_oracle as orc


def make_state(vnet_sd, fe_sd, qrep_w, base_lr=0.01, max_iterations=6000):
    """Student / teacher V-Net by state_dict key, q / k FeatureExtractor_3d, the two 1x1x1 q_representation weights
    (train_arco_3d.py:195-225: both nets load one checkpoint, the k extractor is a copy of the q extractor)."""
    return dict(
        student={k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in vnet_sd.items()},
        teacher={k: v.clone() for k, v in vnet_sd.items()},
        q_fe={k: v.clone().requires_grad_(True) for k, v in fe_sd.items()},
        k_fe={k: v.clone() for k, v in fe_sd.items()},
        q_rep=[w.clone().requires_grad_(True) for w in qrep_w],
        mom={}, it=0, lr=base_lr, base_lr=base_lr, max_iterations=max_iterations)


def warp_volume(x, grid, padding_mode='zeros'):
    """RandTPS.forward of tps/rand_tps_3d.py:140-158 on a 5-D tensor: ONE 2-D warp applied to every slice x[..., d]."""
    res = torch.zeros_like(x)
    for d in range(x.shape[-1]):
        res[..., d] = orc.grid_sample(x[..., d], grid, padding_mode)
    return res


def step(st, l_data, l_label, u_data, memobank, ptrs, qsize, n_cls=2, epoch_num=0, max_epoch=1, k1=0.01, k3=1.0, k4=1.0,
         delta_n=0.97, strong_threshold=0.97, weak_threshold=0.7, func='asmc', nq=256, nn_=512, tps_sigma=0.01,
         apply_aug='cutmix', eqv_pass=True, pool=None, topk=5, trace=None, force=None):
    """One iteration of train_arco_3d.py:255-400.  Train-mode forwards run in the reference's order with track=True, so
    the BatchNorm running statistics of both nets receive their (non-commuting) momentum updates in that order:
    teacher u (:260), student l, u_aug (:283-284), teacher l, u_aug (:286-287), student warped (:380).

    force: the step's gradient-free DECISION inputs as another implementation computed them (pseudo_labels / pseudo_logits
    before the mixing, low / high entropy masks, teacher probabilities).  Each is first compared with this function's own
    value (agreement counts in st["agree"]) and then used in its place: every threshold, arg-max and rank decision of the
    step - 10^5 voxels, each a hair-trigger for the index sets, the sampler arguments and through them the CPU generator
    stream - is then taken on identical numbers, and everything continuous downstream can be compared strictly."""
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
    with torch.no_grad():
        pred_u0, _, _ = orc.vnet_forward(u_data, st["teacher"], track=True)                       # :259-261
        pseudo_logits, pseudo_labels = torch.max(torch.softmax(pred_u0, 1), 1)
        pseudo_logits, pseudo_labels = forced("pseudo_logits", pseudo_logits), forced("pseudo_labels", pseudo_labels)
        # :262-277: batch_transform of augment_3d.py:133-159 returns its inputs (every transform is commented out)
        if apply_aug in ('cutout', 'cutmix', 'classmix'):                                         # :270-271
            mixed = orc.generate_unsup_data(u_data.numpy(), pseudo_labels.numpy().copy(), pseudo_logits.numpy(), apply_aug)
            u_data, pseudo_labels, pseudo_logits = (torch.from_numpy(v) for v in mixed)
        for k in st["k_fe"]:                                                                      # :279-281
            st["k_fe"][k] = st["k_fe"][k] * 0.99 + st["q_fe"][k].detach() * 0.01
    pred_l, _, l_fm = orc.vnet_forward(l_data, st["student"], track=True)                         # :283
    pred_u, _, u_fm = orc.vnet_forward(u_data, st["student"], track=True)                         # :284
    with torch.no_grad():
        pred_l_t, _, l_fm_t = orc.vnet_forward(l_data, st["teacher"], track=True)                 # :286
        pred_u_t, _, u_fm_t = orc.vnet_forward(u_data, st["teacher"], track=True)                 # :287
        rep_l_t = orc.feature_extractor_forward(l_fm_t, st["k_fe"], mode='trilinear')             # :292-293
        rep_u_t = orc.feature_extractor_forward(u_fm_t, st["k_fe"], mode='trilinear')
        rep_t = torch.cat((rep_l_t, rep_u_t))                                                     # :302 pred_all_teacher
    l_feat = orc.feature_extractor_forward(l_fm, st["q_fe"], mode='trilinear')                    # :289-290
    u_feat = orc.feature_extractor_forward(u_fm, st["q_fe"], mode='trilinear')
    rep_u = F.conv3d(F.conv3d(u_feat, st["q_rep"][0]), st["q_rep"][1])                            # :295-296
    rep_l = F.conv3d(F.conv3d(l_feat, st["q_rep"][0]), st["q_rep"][1])
    rep_all = torch.cat((rep_l, rep_u))
    pred_all = torch.cat((pred_l, pred_u))
    loss_q = None
    if pool is not None:                                                                          # :303
        with torch.no_grad():
            loss_q = orc.get_revisiting_loss(pool["rows"], rep_u.detach(), rep_u_t, topk)
    ce, dice = orc.supervised_loss(pred_l, l_label, n_cls)                                        # :305-308
    unsup = orc.compute_unsupervised_loss(pred_u, pseudo_labels, pseudo_logits, strong_threshold)  # :309
    alpha_t = 20 * (1 - epoch_num / max_epoch)                                                    # :311-313
    with torch.no_grad():
        label_l = orc.label_onehot(l_label, n_cls).long()                                         # :315-317
        label_u = orc.label_onehot(pseudo_labels, n_cls).long()
        low, high, _ = orc.entropy_masks(pred_u, l_label, pseudo_labels, alpha_t)                 # :319-354
        pl, pu = torch.softmax(pred_l_t, 1), torch.softmax(pred_u_t, 1)                           # :322-323
        low, high, pl, pu = forced("low", low), forced("high", high), forced("prob_l_t", pl), forced("prob_u_t", pu)
    _, reco = orc.compute_contra_memobank_loss(rep_all, label_l, label_u, pl, pu, low, high, memobank, ptrs, qsize, rep_t,
                                               delta_n=delta_n, func=func, num_queries=nq, num_negatives=nn_, trace=trace)   # :357-361
    if pool is not None:                                                                          # :364-366
        with torch.no_grad():
            nb = int(rep_u_t.shape[0])
            orc.pool_enqueue(F.normalize(rep_u_t.reshape(nb, -1), dim=-1), pool["rows"], pool["ptr"], int(pool["rows"].shape[0]))
    eqv = None
    if eqv_pass:                                                                                  # :369-388
        # RandTPS(patch[0], patch[1], patch[2]) = RandTPS(width, height, depth) (:227): grid height patch[1], width patch[0]
        H, W = int(l_data.shape[3]), int(l_data.shape[2])
        if "tps" not in st:
            st["tps"] = orc.tps_constants(H, W)
        tcp, inv, crep = st["tps"]
        nb2 = int(l_data.shape[0]) + int(u_data.shape[0])
        with torch.no_grad():
            labels = torch.cat((l_label, pseudo_labels))
            logits = torch.cat((torch.ones(l_label.shape) * 255, pseudo_logits))                  # images_cj1_logits_l = 255 (:263)
            mask = orc.eqv_mask(labels, logits, weak_threshold)                                   # :371-375
            grid = orc.tps_grid(orc.rand_tps_source_points(tcp, nb2, tps_sigma), inv, crep, H, W)  # :377 reset_control_points
            images_tps = warp_volume(torch.cat((l_data, u_data)), grid)                           # :376,378 (images_cj2 = inputs)
            mask_tps = warp_volume(mask, grid)                                                    # :379
            org = warp_volume(pred_all.detach(), grid)                                            # :381-383
        pred_tps = orc.vnet_forward(images_tps, st["student"], track=True)[0]                     # :380
        eqv = orc.eqv_loss(pred_tps, org, mask_tps)                                               # :384-388
        if trace is not None:
            trace.update(grid=grid, images_tps=images_tps, mask_tps=mask_tps, org=org, pred_tps=pred_tps.detach(), eqv_mask=mask)
    first = st["it"] == 0                                                                         # :390 iter_num / max_iterations > 0.0
    if first and eqv is not None:
        loss = unsup + (dice + ce) + eqv                                                          # :393
    else:
        loss = k1 * reco + k3 * unsup + (dice + ce)                                               # :391
        if loss_q is not None:
            loss = loss + k4 * loss_q
    leaves = [v for v in st["student"].values() if v.requires_grad] + list(st["q_rep"]) + list(st["q_fe"].values())
    grads = torch.autograd.grad(loss, leaves, allow_unused=True)                                  # :394-396
    with torch.no_grad():
        for i, (p, g) in enumerate(zip(leaves, grads)):
            if g is None:     # torch.optim.SGD skips parameters whose .grad is None (zero_grad(set_to_none) + no path)
                continue
            newp, st["mom"][i] = orc.sgd_nesterov_step(p, g, st["mom"].get(i), st["lr"])
            p.copy_(newp)
        for k, v in st["student"].items():                                                        # :397 isd._momentum_update_key_encoder
            if v.requires_grad:
                st["teacher"][k] = st["teacher"][k] * 0.99 + v.detach() * 0.01
    st["lr"] = orc.poly_lr(st["base_lr"], st["it"], st["max_iterations"])                         # :399-401
    st["it"] += 1
    st["agree"] = agree
    st["last_terms"] = dict(ce=float(ce.detach()), dice=float(dice.detach()), unsup=float(unsup.detach()), reco=float(reco.detach()))
    if eqv is not None:
        st["last_terms"]["eqv"] = float(eqv.detach())
    if loss_q is not None:
        st["last_terms"]["loss_q"] = float(loss_q)
    if trace is not None:
        trace.update(low=low, high=high, pseudo_labels=pseudo_labels, pred_u=pred_u.detach(), pred_l_t=pred_l_t, pred_u_t=pred_u_t,
                     rep_all=rep_all.detach(), rep_t=rep_t)
    return float(loss.detach()), float(reco.detach())
