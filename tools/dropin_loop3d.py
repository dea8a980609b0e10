"""3-D twin of tools/dropin_loop.py: the loop body of the reference's code/train_arco_3d.py:255-400 over the names its import
block binds through `dropin/` (`from augment_3d import *`, `from loss_helper import *`, `from model_3D import *`,
`from tps.rand_tps_3d import RandTPS`), with torch's own `nn.Conv3d` q_representation and `torch.optim.SGD`.  Three steps at 32^3
(iteration 0 optimises unsup + supervised + loss_eqv, :393); prints one JSON line."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dropin"))
sys.path.insert(1, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim
from torch.nn.modules.loss import CrossEntropyLoss

# ---- the reference trainer's import block (train_arco_3d.py:18-24), verbatim
from utils import losses, metrics, ramps            # noqa: F401,E402
from tps.rand_tps_3d import RandTPS                 # noqa: E402
from augment_3d import *                            # noqa: F401,F403,E402
from loss_helper import *                           # noqa: F401,F403,E402
from model_3D import *                              # noqa: F401,F403,E402

import fixture_inputs as fx                         # noqa: E402


def compute_unsupervised_loss(predict, target, logits, strong_threshold):          # train_arco_3d.py:453-460, verbatim
    batch_size = predict.shape[0]
    valid_mask = (target >= 0).float()
    weighting = logits.view(batch_size, -1).ge(strong_threshold).sum(-1) / valid_mask.view(batch_size, -1).sum(-1)
    loss = F.cross_entropy(predict, target, reduction='none', ignore_index=-1)
    weighted_loss = torch.mean(torch.masked_select(weighting[:, None, None, None] * loss, loss > 0))
    return weighted_loss


def label_onehot(inputs, num_segments):                                           # :463-469 (the .cpu() hop dropped)
    batch_size, im_h, im_w, im_d = inputs.shape
    inputs = torch.relu(inputs).data.type(torch.int64)
    outputs = torch.zeros([batch_size, num_segments, im_h, im_w, im_d]).to(inputs.device)
    return outputs.scatter_(1, inputs.unsqueeze(1), 1.0)


def main():
    num_classes, batch_size, patch, Q, Nn, base_lr = 4, 2, (32, 32, 32), 48, 16, 0.01
    memobank, queue_ptrlis, queue_size = [], [], []
    random.seed(3); np.random.seed(3); torch.manual_seed(3)
    for i in range(num_classes):                                                  # :144-151
        memobank.append([torch.randn(1, 16)])
        queue_size.append(200)
        queue_ptrlis.append(torch.zeros(1, dtype=torch.long))
    isd = ISD_3d(K=36, m=0.99, Ts=0.01, Tt=0.1, num_classes=num_classes, latent_pooling_size=1, latent_feature_size=512,
                 output_pooling_size=8, train_encoder=True, train_decoder=True).cuda()
    vnet_sd = fx.vnet_state(52, 1, num_classes)
    isd.model.load_state_dict(vnet_sd); isd.ema_model.load_state_dict(vnet_sd)
    ema_model, model = isd.ema_model, isd.model
    for m in (model, ema_model):
        m.has_dropout = False
    q_representation = nn.Sequential(nn.Conv3d(16, 16, kernel_size=1, bias=False), nn.Conv3d(16, 16, kernel_size=1, bias=False)).cuda()
    k_feature_extractor = FeatureExtractor_3d(fea_dim=[128, 64, 32, 16, 16], output_dim=16).cuda()
    q_feature_extractor = FeatureExtractor_3d(fea_dim=[128, 64, 32, 16, 16], output_dim=16).cuda()
    q_feature_extractor.load_state_dict(fx.fe_state(61, (128, 64, 32, 16, 16), 16, nd=3))
    params = [p for p in model.parameters() if p.requires_grad]
    params_rep = [p for p in q_representation.parameters() if p.requires_grad]
    params_fea = [p for p in q_feature_extractor.parameters() if p.requires_grad]
    optimizer = optim.SGD(params + params_rep + params_fea, lr=base_lr, weight_decay=0.0001, momentum=0.9, nesterov=True)
    with torch.no_grad():
        for t_params, s_params in zip(k_feature_extractor.parameters(), q_feature_extractor.parameters()):
            t_params.data.copy_(s_params.data)
            t_params.requires_grad = False
    tps = RandTPS(patch[0], patch[1], patch[2], batch_size=batch_size * 2, sigma=0.01, border_padding=False, random_mirror=True,
                  random_scale=(0.8, 1.2), mode='affine').cuda()
    model.train(); ema_model.train(); q_representation.train(); k_feature_extractor.train(); q_feature_extractor.train()
    ce_loss = CrossEntropyLoss()
    dice_loss = losses.DiceLoss(num_classes)
    iter_num, max_iterations, epoch_num, max_epoch = 0, 6000, 0, 100
    rs = np.random.RandomState(13)
    w0 = model.state_dict()["block_one.conv.0.weight"].clone()
    out = []
    for it in range(3):
        train_l_data = torch.from_numpy(rs.uniform(size=(batch_size, 1, *patch)).astype(np.float32))
        train_u_data = torch.from_numpy(rs.uniform(size=(batch_size, 1, *patch)).astype(np.float32)).cuda()
        train_l_label = torch.from_numpy(fx.blob_labels(rs, batch_size, patch, num_classes))
        train_l_data = (train_l_data + 0.5 * (train_l_label > 0).unsqueeze(1).float()).cuda()
        train_l_label = train_l_label.cuda()
        random.seed(10 + it); np.random.seed(10 + it); torch.manual_seed(10 + it)
        with torch.no_grad():
            pred_u, _, _ = ema_model(train_u_data)
        pseudo_logits, pseudo_labels = torch.max(torch.softmax(pred_u, dim=1), dim=1)
        _, _, images_cj1_logits_l = batch_transform(train_l_data, train_l_label, logits=torch.ones_like(train_l_label) * 255,
                                                    scale_size=(1.0, 1.0), apply_augmentation=False)
        images_cj2_l, _, _ = batch_transform(train_l_data, train_l_label, logits=torch.ones_like(train_l_label) * 255,
                                             scale_size=(1.0, 1.0), apply_augmentation=False)
        train_u_aug_data, train_u_aug_label, train_u_aug_logits = generate_unsup_data_3d(train_u_data, pseudo_labels, pseudo_logits, mode='cutmix')
        images_cj2_u, _, _ = batch_transform(train_u_aug_data, train_u_aug_label, logits=train_u_aug_logits, scale_size=(1.0, 1.0),
                                             apply_augmentation=True)
        train_u_aug_data, train_u_aug_label, train_u_aug_logits = batch_transform(train_u_aug_data, train_u_aug_label, logits=train_u_aug_logits,
                                                                                  scale_size=(1.0, 1.0), apply_augmentation=True)
        with torch.no_grad():
            for param_q, param_k in zip(q_feature_extractor.parameters(), k_feature_extractor.parameters()):
                param_k.data = param_k.data * 0.99 + param_q.data * 0.01
        pred_l, _, l_feature_map = model(train_l_data)
        pred_u, _, u_feature_map = model(train_u_aug_data)
        pred_l_teacher, _, l_feature_map_teacher = ema_model(train_l_data)
        pred_u_teacher, _, u_feature_map_teacher = ema_model(train_u_aug_data)
        l_feature_all = q_feature_extractor(l_feature_map)
        u_feature_all = q_feature_extractor(u_feature_map)
        l_feature_all_teacher = k_feature_extractor(l_feature_map_teacher)
        u_feature_all_teacher = k_feature_extractor(u_feature_map_teacher)
        rep_u = q_representation(u_feature_all)
        rep_l = q_representation(l_feature_all)
        rep_all = torch.cat((rep_l, rep_u))
        pred_all = torch.cat((pred_l, pred_u))
        pred_all_teacher = torch.cat((l_feature_all_teacher, u_feature_all_teacher))
        outputs_soft = torch.softmax(pred_l, dim=1)
        loss_ce = ce_loss(pred_l, train_l_label.long())
        loss_dice = dice_loss(outputs_soft, train_l_label.unsqueeze(1))
        supervised_loss = (loss_dice + loss_ce)
        unsup_loss = compute_unsupervised_loss(pred_u, train_u_aug_label, train_u_aug_logits, 0.3)
        alpha_t = 20 * (1 - epoch_num / max_epoch)
        with torch.no_grad():
            label_l = label_onehot(train_l_label, num_classes)
            label_u = label_onehot(train_u_aug_label, num_classes)
            prob_l_teacher = torch.softmax(pred_l_teacher, dim=1)
            prob_u_teacher = torch.softmax(pred_u_teacher, dim=1)
            prob = torch.softmax(pred_u, dim=1)
            entropy = -torch.sum(prob * torch.log(prob + 1e-10), dim=1)
            ent_valid = entropy[train_u_aug_label >= 0].cpu().numpy().flatten()
            low_entropy_mask = (entropy.le(np.percentile(ent_valid, alpha_t)).float() * (train_u_aug_label >= 0).bool())
            high_entropy_mask = (entropy.ge(np.percentile(ent_valid, 100 - alpha_t)).float() * (train_u_aug_label >= 0).bool())
            low_mask_all = torch.cat(((train_l_label.unsqueeze(1) >= 0).float(), low_entropy_mask.unsqueeze(1)))
            high_mask_all = torch.cat(((train_l_label.unsqueeze(1) >= 0).float(), high_entropy_mask.unsqueeze(1)))
        reco_loss = compute_contra_memobank_loss(rep_all, label_l.cuda().long(), label_u.cuda().long(), prob_l_teacher.detach(),
                                                 prob_u_teacher.detach(), low_mask_all.cuda(), high_mask_all.cuda(), memobank, queue_ptrlis,
                                                 queue_size, pred_all_teacher.detach(), delta_n=0.97, func='asmc', num_queries=Q,
                                                 num_negatives=Nn)[-1]
        labels = torch.cat((train_l_label, train_u_aug_label), dim=0)
        logits = torch.cat((images_cj1_logits_l.float(), train_u_aug_logits), dim=0)
        mask = torch.ones((rep_all.shape[0], rep_all.shape[2], rep_all.shape[3], rep_all.shape[4]), requires_grad=False).cuda()
        neg = torch.zeros_like(mask)
        mask = torch.where(labels == 0, neg, mask)
        mask = torch.where(logits < 0.7, neg, mask)
        mask = mask.unsqueeze(1)
        images_cj2 = torch.cat((images_cj2_l, images_cj2_u), dim=0)
        tps.reset_control_points()
        images_tps = tps(images_cj2)
        mask_tps = tps(mask.float(), padding_mode='zeros')
        pred_tps = model(images_tps)[0]
        pred_d = pred_all.detach()
        pred_tps_org = tps(pred_d.cuda(), padding_mode='zeros')
        kl = nn.KLDivLoss(reduction='none').cuda()
        loss_eqv = kl(F.log_softmax(pred_tps, dim=1), F.softmax(pred_tps_org, dim=1))
        loss_eqv = (loss_eqv * mask_tps).flatten(1).sum(1) / (mask_tps.flatten(1).sum(1) + 1e-7)
        loss_eqv = loss_eqv.mean()
        if iter_num / max_iterations > 0.0:                                       # :390-393
            loss = 1.0 * reco_loss + 1.0 * unsup_loss + supervised_loss
        else:
            loss = unsup_loss + supervised_loss + loss_eqv
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        isd._momentum_update_key_encoder()
        lr_ = base_lr * (1.0 - iter_num / max_iterations) ** 0.9
        for param_group in optimizer.param_groups:
            param_group['lr'] = lr_
        iter_num += 1
        out.append(dict(ce=float(loss_ce), dice=float(loss_dice), unsup=float(unsup_loss), reco=float(reco_loss), eqv=float(loss_eqv),
                        loss=float(loss), bank_len=[int(b[0].shape[0]) for b in memobank], ptr=[int(p) for p in queue_ptrlis],
                        banks_on_gpu=all(b[0].is_cuda for b in memobank)))
    sd = model.state_dict()
    tail = dict(moved=float((sd["block_one.conv.0.weight"] - w0).abs().max()), finite=bool(all(torch.isfinite(p).all() for p in model.parameters())),
                qrep_finite=bool(all(torch.isfinite(p).all() for p in q_representation.parameters())),
                model_file=sys.modules["model_3D"].__file__)
    print("DROPIN_LOOP3D " + json.dumps(dict(steps=out, tail=tail)))


if __name__ == "__main__":
    main()
