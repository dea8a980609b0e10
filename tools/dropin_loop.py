"""The drop-in boundary exercised the way the reference uses it (run by tests/test_dropin_loop_gpu.py in a fresh interpreter with
`dropin/` first on sys.path): the loop body of the reference's code/train_arco_2d.py:284-435, statement by statement, over the
names its own import block binds - `from model_2D import *`, `from loss_helper_3d import *`, `from utils import losses, ramps`,
`from augment import *`, `from tps.rand_tps import RandTPS` - with the objects the reference builds itself: `q_representation` is a
plain torch `nn.Sequential(nn.Conv2d(496, 496, 1), nn.Conv2d(496, 496, 1))`, the optimiser is `torch.optim.SGD(..., nesterov=True)`
(no flat buffers, no PackPlan, no graphs), the k-FeatureExtractor EMA is the reference's `param_k.data = ...` statement, banks start
as CPU tensors.  What is NOT taken from the reference's loop: data loading, tensorboard / logging, and `batch_transform` (its PIL
arithmetic is pinned separately; `--bt 1` switches it on).  Two steps; prints one JSON line with the loss terms, bank lengths,
pointers and a few weight checksums - the test compares them with the CPU oracle step (oracle/cpu_step.py)."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dropin"))          # ahead of everything: the reference's import statements bind arco_amd
sys.path.insert(1, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim
from torch.nn.modules.loss import CrossEntropyLoss

# ---- the reference trainer's own import block (train_arco_2d.py:18-24), verbatim
from utils import losses, metrics, ramps            # noqa: F401,E402
from tps.rand_tps import RandTPS                    # noqa: E402
from augment import *                               # noqa: F401,F403,E402
from loss_helper_3d import *                        # noqa: F401,F403,E402
from model_2D import *                              # noqa: F401,F403,E402

import fixture_inputs as fx                         # noqa: E402


def compute_unsupervised_loss(predict, target, logits, strong_threshold):          # train_arco_2d.py:482-489, verbatim
    batch_size = predict.shape[0]
    valid_mask = (target >= 0).float()   # only count valid pixels
    weighting = logits.view(batch_size, -1).ge(strong_threshold).sum(-1) / valid_mask.view(batch_size, -1).sum(-1)
    loss = F.cross_entropy(predict, target, reduction='none', ignore_index=-1)
    weighted_loss = torch.mean(torch.masked_select(weighting[:, None, None] * loss, loss > 0))
    return weighted_loss


def label_onehot(inputs, num_segments):                                           # :492-498 (the .cpu() hop dropped: INTEGRATION.md)
    batch_size, im_h, im_w = inputs.shape
    inputs = torch.relu(inputs).data.type(torch.int64)
    outputs = torch.zeros([batch_size, num_segments, im_h, im_w]).to(inputs.device)
    return outputs.scatter_(1, inputs.unsqueeze(1), 1.0)


def main():
    num_classes, batch_size, patch, Q, Nn, qs, base_lr = 4, 2, (64, 64), 64, 32, 300, 0.01
    timing = int(os.environ.get("DROPIN_TIME_STEPS", "0"))        # > 0: the headline workload (8 + 8 images of 256 x 256, 4096-key queues,
    if timing:                                                    # 256 x 512 samples), `timing` timed steps after 3 untimed ones
        batch_size, patch, Q, Nn, qs = 8, (256, 256), 256, 512, 4096
    k1, k2, k3 = 1.0, float(os.environ.get("K2", "1.0")), 1.0
    apply_aug = os.environ.get("APPLY_AUG", "cutmix")
    memobank, queue_ptrlis, queue_size = [], [], []
    for i in range(num_classes):                                                  # :147-154
        memobank.append([torch.zeros(1, 496)])
        queue_size.append(qs)
        queue_ptrlis.append(torch.zeros(1, dtype=torch.long))
    isd = ISD(K=36, m=0.99, Ts=0.01, Tt=0.1, num_classes=num_classes, latent_pooling_size=1, latent_feature_size=512,
              output_pooling_size=8, train_encoder=True, train_decoder=True).cuda()      # :217-219
    unet_sd = fx.unet_state(21, 1, num_classes)
    isd.model.load_state_dict(unet_sd)                                            # :223-226 (the stage-1 checkpoint)
    isd.ema_model.load_state_dict(unet_sd)
    ema_model, model = isd.ema_model, isd.model
    q_representation = nn.Sequential(nn.Conv2d(496, 496, kernel_size=1, bias=False),
                                     nn.Conv2d(496, 496, kernel_size=1, bias=False)).cuda()     # :231-234: TORCH modules
    k_feature_extractor = FeatureExtractor(fea_dim=[256, 128, 64, 32, 16], output_dim=496).cuda()
    q_feature_extractor = FeatureExtractor(fea_dim=[256, 128, 64, 32, 16], output_dim=496).cuda()
    fe_sd = fx.fe_state(31)
    q_feature_extractor.load_state_dict(fe_sd)
    with torch.no_grad():
        q_representation[0].weight.copy_(fx.fe_state(32)["fea4.weight"])
        q_representation[1].weight.copy_(fx.fe_state(33)["fea4.weight"])
    for m in (model, ema_model):                                                  # dropout off (the oracle has none)
        for mod in m.modules():
            if isinstance(mod, nn.Dropout):
                mod.p = 0.0
    params = [p for p in model.parameters() if p.requires_grad]                   # :245-248
    params_rep = [p for p in q_representation.parameters() if p.requires_grad]
    params_fea = [p for p in q_feature_extractor.parameters() if p.requires_grad]
    optimizer = optim.SGD(params + params_rep + params_fea, lr=base_lr, weight_decay=0.0001, momentum=0.9, nesterov=True)
    with torch.no_grad():                                                         # :250-253
        for t_params, s_params in zip(k_feature_extractor.parameters(), q_feature_extractor.parameters()):
            t_params.data.copy_(s_params.data)
            t_params.requires_grad = False
    tps = RandTPS(patch[0], patch[1], batch_size=batch_size * 2, sigma=0.01, border_padding=False, random_mirror=True,
                  random_scale=(0.8, 1.2), mode='affine').cuda()                  # :255-261
    model.train(); ema_model.train(); q_representation.train(); k_feature_extractor.train(); q_feature_extractor.train()
    ce_loss = CrossEntropyLoss()
    dice_loss = losses.DiceLoss(num_classes)
    iter_num, max_iterations, epoch_num, max_epoch = 0, 30000, 0, 100
    rs = np.random.RandomState(3)
    out, resident = [], []
    import time
    t_start = None
    for it in range(2 if not timing else 3 + timing):
        if timing and it == 3:
            torch.cuda.synchronize(); t_start = time.perf_counter()
        if timing and it >= 3:          # timed steps: resident batches (the data pipeline is not part of the metric)
            train_l_data, train_u_data, train_l_label = resident[it % 3]
        else:
            train_l_data = torch.from_numpy(rs.uniform(size=(batch_size, 1, *patch)).astype(np.float32)).cuda()
            train_u_data = torch.from_numpy(rs.uniform(size=(batch_size, 1, *patch)).astype(np.float32)).cuda()
            train_l_label = torch.from_numpy(fx.blob_labels(rs, batch_size, patch, num_classes)).cuda()
            resident.append((train_l_data, train_u_data, train_l_label))
        random.seed(10 + it); np.random.seed(10 + it); torch.manual_seed(10 + it)
        with torch.no_grad():                                                     # :284-286
            pred_u, _, _ = ema_model(train_u_data)
        pseudo_logits, pseudo_labels = torch.max(torch.softmax(pred_u, dim=1), dim=1)
        train_u_aug_data, train_u_aug_label, train_u_aug_logits = train_u_data, pseudo_labels, pseudo_logits
        train_u_aug_data, train_u_aug_label, train_u_aug_logits = \
            generate_unsup_data(train_u_aug_data, train_u_aug_label, train_u_aug_logits, mode=apply_aug)      # :296-297
        images_cj2_l, images_cj2_u = train_l_data, train_u_aug_data               # (batch_transform left out, see the docstring)
        with torch.no_grad():                                                     # :306-308, the reference's statement
            for param_q, param_k in zip(q_feature_extractor.parameters(), k_feature_extractor.parameters()):
                param_k.data = param_k.data * 0.99 + param_q.data * 0.01
        pred_l, _, l_feature_map = model(train_l_data)                            # :310-315
        _, _, l_feature_map_2 = model(images_cj2_l)
        pred_u, _, u_feature_map = model(train_u_aug_data)
        pred_l_teacher, _, l_feature_map_teacher = ema_model(train_l_data)
        pred_u_teacher, _, u_feature_map_teacher = ema_model(train_u_aug_data)
        l_feature_all = q_feature_extractor(l_feature_map)                        # :317-322
        u_feature_all = q_feature_extractor(u_feature_map)
        l_feature_all_teacher = k_feature_extractor(l_feature_map_teacher)
        u_feature_all_teacher = k_feature_extractor(u_feature_map_teacher)
        rep_u = q_representation(u_feature_all)                                   # :324-326
        rep_l = q_representation(l_feature_all)
        rep_u_teacher, rep_l_teacher = u_feature_all_teacher, l_feature_all_teacher
        rep_all = torch.cat((rep_l, rep_u))
        pred_all = torch.cat((pred_l, pred_u))
        pred_all_teacher = torch.cat((rep_l_teacher, rep_u_teacher))
        outputs_soft = torch.softmax(pred_l, dim=1)                               # :336-340
        loss_ce = ce_loss(pred_l, train_l_label.long())
        loss_dice = dice_loss(outputs_soft, train_l_label.unsqueeze(1))
        supervised_loss = (loss_dice + loss_ce)
        unsup_loss = compute_unsupervised_loss(pred_u, train_u_aug_label, train_u_aug_logits, 0.97)
        alpha_t = 20 * (1 - epoch_num / max_epoch)
        with torch.no_grad():                                                     # :342-393
            label_l = label_onehot(train_l_label, num_classes)
            label_u = label_onehot(train_u_aug_label, num_classes)
            prob_l_teacher = torch.softmax(pred_l_teacher, dim=1)
            prob_u_teacher = torch.softmax(pred_u_teacher, dim=1)
            prob = torch.softmax(pred_u, dim=1)
            entropy = -torch.sum(prob * torch.log(prob + 1e-10), dim=1)
            ent_valid = entropy[train_u_aug_label >= 0].cpu().numpy().flatten()
            low_thresh = np.percentile(ent_valid, alpha_t)
            low_entropy_mask = (entropy.le(low_thresh).float() * (train_u_aug_label >= 0).bool())
            high_thresh = np.percentile(ent_valid, 100 - alpha_t)
            high_entropy_mask = (entropy.ge(high_thresh).float() * (train_u_aug_label >= 0).bool())
            low_mask_all = torch.cat(((train_l_label.unsqueeze(1) >= 0).float(), low_entropy_mask.unsqueeze(1)))
            high_mask_all = torch.cat(((train_l_label.unsqueeze(1) >= 0).float(), high_entropy_mask.unsqueeze(1)))
        reco_loss = compute_contra_memobank_loss(rep_all, label_l.cuda().long(), label_u.cuda().long(),
                                                 prob_l_teacher.detach(), prob_u_teacher.detach(), low_mask_all.cuda(), high_mask_all.cuda(),
                                                 memobank, queue_ptrlis, queue_size, pred_all_teacher.detach(), delta_n=0.97,
                                                 func='smc', num_queries=Q, num_negatives=Nn)[-1]        # :394-398
        loss = k1 * reco_loss + k3 * unsup_loss + supervised_loss
        loss_eqv = None
        if k2 != 0:                                                               # :404-423
            labels = torch.cat((train_l_label, train_u_aug_label), dim=0)
            logits = torch.cat((torch.ones_like(train_l_label).float(), train_u_aug_logits), dim=0)
            mask = torch.ones((rep_all.shape[0], rep_all.shape[2], rep_all.shape[3]), requires_grad=False).cuda()
            neg = torch.zeros((rep_all.shape[0], rep_all.shape[2], rep_all.shape[3]), requires_grad=False).cuda()
            mask = torch.where(labels == 0, neg, mask)
            mask = torch.where(logits < 0.7, neg, mask)
            mask = mask.unsqueeze(1)
            images_cj2 = torch.cat((images_cj2_l, images_cj2_u), dim=0)
            tps.reset_control_points()
            images_tps = tps(images_cj2)
            mask_tps = tps(mask.float(), padding_mode='zeros')
            pred_tps = model(images_tps)[0]
            pred_d = pred_all.detach()
            pred_d.requires_grad = False
            pred_tps_org = tps(pred_d.cuda(), padding_mode='zeros')
            kl = nn.KLDivLoss(reduction='none').cuda()
            loss_eqv = kl(F.log_softmax(pred_tps, dim=1), F.softmax(pred_tps_org, dim=1))
            loss_eqv = (loss_eqv * mask_tps).flatten(1).sum(1) / (mask_tps.flatten(1).sum(1) + 1e-7)
            loss_eqv = loss_eqv.mean()
            loss = loss + k2 * loss_eqv
        optimizer.zero_grad()                                                     # :429-435
        loss.backward()
        optimizer.step()
        isd._momentum_update_key_encoder()
        lr_ = base_lr * (1.0 - iter_num / max_iterations) ** 0.9
        for param_group in optimizer.param_groups:
            param_group['lr'] = lr_
        iter_num += 1
        if timing:
            continue
        out.append(dict(ce=float(loss_ce), dice=float(loss_dice), unsup=float(unsup_loss), reco=float(reco_loss),
                        eqv=None if loss_eqv is None else float(loss_eqv),
                        bank_len=[int(b[0].shape[0]) for b in memobank], ptr=[int(p) for p in queue_ptrlis],
                        bank_sum=[float(b[0].double().abs().sum()) for b in memobank]))
    if timing:
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t_start) / timing * 1e3
        print("DROPIN_TIME " + json.dumps(dict(ms_per_step=round(ms, 2), steps=timing, peak_mem_gb=round(torch.cuda.max_memory_allocated() / 1e9, 2),
                                                last=dict(ce=float(loss_ce), dice=float(loss_dice), unsup=float(unsup_loss), reco=float(reco_loss)))))
        return
    sd = model.state_dict()
    tail = dict(w_first=float(sd["encoder.in_conv.conv_conv.0.weight"].double().abs().sum()),
                w_last=float(sd["decoder.out_conv.weight"].double().abs().sum()),
                w_deep=float(sd["encoder.down4.maxpool_conv.1.conv_conv.4.weight"].double().abs().sum()),
                qrep0=float(q_representation[0].weight.double().abs().sum()), qrep1=float(q_representation[1].weight.double().abs().sum()),
                qfe4=float(q_feature_extractor.fea4.weight.double().abs().sum()), kfe4=float(k_feature_extractor.fea4.weight.double().abs().sum()),
                t_first=float(ema_model.state_dict()["encoder.in_conv.conv_conv.0.weight"].double().abs().sum()),
                rm=float(sd["encoder.in_conv.conv_conv.1.running_mean"].double().abs().sum()),
                arco_modules=sorted(k for k in sys.modules if k.startswith("arco_amd"))[:2],
                model_file=sys.modules["model_2D"].__file__)
    print("DROPIN_LOOP " + json.dumps(dict(steps=out, tail=tail)))


if __name__ == "__main__":
    main()
