"""The VOLUME twin of tests/dropin_user.py: a user of the drop-in boundary for the 3-D trainer, in this repository's own form:
`dropin/` first on sys.path, train_arco_3d.py's import statements (`from model_3D import *`, `from loss_helper import *`,
`from utils import losses`, `from augment_3d import *`, `from tps.rand_tps_3d import RandTPS`) bind arco_amd, and everything a reference-style
trainer builds ITSELF is plain torch: `q_representation` is `nn.Sequential(nn.Conv3d, nn.Conv3d)`, the optimiser `torch.optim.SGD` over
the drop-in modules' parameters, the k-FeatureExtractor EMA re-points `.data`, the banks start as CPU tensors.  None of arco_amd's
trainer machinery (flat buffers, PackPlans, graphs, row-sparse head) is involved.

The step is a sequence of small stages over a state dict (the shape of oracle/cpu_step.py), in the order the algorithm dictates
(train_arco_3d.py:259-400: iteration 0 optimises `unsup + supervised + loss_eqv`, later ones the contrastive objective); its
numbers are compared with the 'v' entries of tests/golden/g19_trainer_loop.npz - the reference's own loop body
executed from the reference's text over the reference's modules on CPU (oracle/gen_golden.py g19) - at north_star's 1e-3.
Prints one JSON line.   python tests/dropin_user3d.py"""
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
import torch.optim as optim          # bound BEFORE the star imports, as in the reference trainers (the shadowing regression)

from utils import losses             # noqa: E402
from tps.rand_tps_3d import RandTPS  # noqa: E402
from augment_3d import *             # noqa: F401,F403,E402
from loss_helper import *            # noqa: F401,F403,E402
from model_3D import *               # noqa: F401,F403,E402

import fixture_inputs as fx          # noqa: E402

DEV = "cuda:0"


def build(C, b, patch, qs, K, rs, bank0):
    """Everything a reference-style volume trainer constructs before its loop (train_arco_3d.py:144-237)."""
    isd = ISD_3d(K=K, m=0.99, Ts=0.01, Tt=0.1, num_classes=C, latent_pooling_size=1, latent_feature_size=128, output_pooling_size=4,
                 train_encoder=True, train_decoder=True).cuda()
    sd = fx.vnet_state(52, 1, C)
    isd.model.load_state_dict(sd); isd.ema_model.load_state_dict(sd)
    for net in (isd.model, isd.ema_model):
        net.has_dropout = False
    q_rep = nn.Sequential(nn.Conv3d(16, 16, 1, bias=False), nn.Conv3d(16, 16, 1, bias=False)).cuda()
    rsq = np.random.RandomState(62)
    with torch.no_grad():
        for layer in q_rep:
            layer.weight.copy_(torch.from_numpy((rsq.standard_normal((16, 16, 1, 1, 1)) / 4).astype(np.float32)))
    k_fe = FeatureExtractor_3d(fea_dim=[128, 64, 32, 16, 16], output_dim=16).cuda()
    q_fe = FeatureExtractor_3d(fea_dim=[128, 64, 32, 16, 16], output_dim=16).cuda()
    q_fe.load_state_dict(fx.fe_state(61, (128, 64, 32, 16, 16), 16, nd=3))
    with torch.no_grad():
        for pk, pq in zip(k_fe.parameters(), q_fe.parameters()):
            pk.data.copy_(pq.data); pk.requires_grad = False
    assert optim is torch.optim, "a star import re-bound the trainer's `optim`"
    opt = optim.SGD([p for grp in (isd.model, q_rep, q_fe) for p in grp.parameters() if p.requires_grad],
                    lr=0.01, weight_decay=0.0001, momentum=0.9, nesterov=True)
    random.seed(6); np.random.seed(6); torch.manual_seed(6)
    tps = RandTPS(patch[0], patch[1], patch[2], batch_size=2 * b, sigma=0.01, border_padding=False, random_mirror=True,
                  random_scale=(0.8, 1.2), mode='affine').cuda()
    for m in (isd.model, isd.ema_model, q_rep, k_fe, q_fe):
        m.train()
    banks = dict(memobank=[[torch.from_numpy(bank0[c:c + 1].copy())] for c in range(C)], ptr=[torch.zeros(1, dtype=torch.long) for _ in range(C)],
                 size=[qs] * C)
    pool = F.normalize(torch.from_numpy(rs.standard_normal((K, 16 * patch[0] * patch[1] * patch[2])).astype(np.float32)), dim=1).cuda()
    return dict(isd=isd, student=isd.model, teacher=isd.ema_model, q_rep=q_rep, k_fe=k_fe, q_fe=q_fe, opt=opt, tps=tps, banks=banks,
                pool=pool, pool_ptr=0, dice=losses.DiceLoss(C), it=0, C=C)


# ---- stages
def teacher_pseudo_labels(S, u):
    with torch.no_grad():
        conf, lab = torch.softmax(S["teacher"](u)[0], dim=1).max(dim=1)
    return lab, conf


def ema_key_extractor(S, m=0.99):
    with torch.no_grad():
        for pq, pk in zip(S["q_fe"].parameters(), S["k_fe"].parameters()):
            pk.data = pk.data * m + pq.data * (1.0 - m)


def confidence_weighted_ce(logits, target, conf, thr):
    share = (conf.flatten(1) >= thr).sum(1) / (target >= 0).float().flatten(1).sum(1)
    ce = F.cross_entropy(logits, target, reduction='none', ignore_index=-1)
    return (share.view(-1, 1, 1, 1) * ce)[ce > 0].mean()


def one_hot(lab, C):
    return F.one_hot(lab.clamp(min=0).long(), C).permute(0, 4, 1, 2, 3).float()


def entropy_masks(student_logits_u, lab_l, lab_u, alpha):
    p = torch.softmax(student_logits_u, dim=1)
    ent = -(p * torch.log(p + 1e-10)).sum(1)
    ok = lab_u >= 0
    e = ent[ok].cpu().numpy().ravel()
    lo, hi = np.percentile(e, alpha), np.percentile(e, 100 - alpha)
    keep_l = (lab_l.unsqueeze(1) >= 0).float()
    return (torch.cat((keep_l, (ent.le(lo) & ok).float().unsqueeze(1))), torch.cat((keep_l, (ent.ge(hi) & ok).float().unsqueeze(1))))


def pool_distance_loss(pool, rep_u, rep_u_t, topk):
    q = F.normalize(rep_u.flatten(1), dim=-1); k = F.normalize(rep_u_t.flatten(1), dim=-1)
    near = (2 - 2 * q @ pool.t()).topk(topk, dim=1, largest=False).indices
    return ((2 - 2 * k @ pool.t()).gather(1, near).sum(1) / topk).mean(), k


def equivariance(S, images, mask, pred_all):
    tps = S["tps"]
    tps.reset_control_points()
    warped_pred = S["student"](tps(images))[0]
    m = tps(mask, padding_mode='zeros')
    target = tps(pred_all.detach(), padding_mode='zeros')
    kl = F.kl_div(F.log_softmax(warped_pred, dim=1), F.softmax(target, dim=1), reduction='none')
    return ((kl * m).flatten(1).sum(1) / (m.flatten(1).sum(1) + 1e-7)).mean()


def step(S, l, lab_l, u, cfg):
    C = S["C"]
    lab_u, conf_u = teacher_pseudo_labels(S, u)
    u_mix, lab_u, conf_u = generate_unsup_data_3d(u, lab_u, conf_u, mode=cfg["mix"])
    ema_key_extractor(S)
    student, teacher = S["student"], S["teacher"]
    pred_l, _, fm_l = student(l)
    pred_u, _, fm_u = student(u_mix)
    tl, _, tfm_l = teacher(l)
    tu, _, tfm_u = teacher(u_mix)
    rep_l, rep_u = S["q_rep"](S["q_fe"](fm_l)), S["q_rep"](S["q_fe"](fm_u))
    rep_lt, rep_ut = S["k_fe"](tfm_l), S["k_fe"](tfm_u)
    rep_all, pred_all = torch.cat((rep_l, rep_u)), torch.cat((pred_l, pred_u))
    loss_q, keys = pool_distance_loss(S["pool"], rep_u, rep_ut, cfg["topk"])
    ce = F.cross_entropy(pred_l, lab_l.long())
    dice = S["dice"](torch.softmax(pred_l, dim=1), lab_l.unsqueeze(1))
    unsup = confidence_weighted_ce(pred_u, lab_u, conf_u, 0.97)
    with torch.no_grad():
        low, high = entropy_masks(pred_u, lab_l, lab_u, alpha=20.0)
        oh_l, oh_u = one_hot(lab_l, C), one_hot(lab_u, C)
        pt_l, pt_u = torch.softmax(tl, dim=1), torch.softmax(tu, dim=1)
    B = S["banks"]
    reco = compute_contra_memobank_loss(rep_all, oh_l.long(), oh_u.long(), pt_l, pt_u, low, high, B["memobank"], B["ptr"], B["size"],
                                        torch.cat((rep_lt, rep_ut)).detach(), delta_n=0.97, func="asmc",
                                        num_queries=cfg["Q"], num_negatives=cfg["Nn"])[-1]
    with torch.no_grad():                                # pool update (a ring of K rows)
        n = keys.shape[0]
        S["pool"][S["pool_ptr"]:S["pool_ptr"] + n] = keys
        S["pool_ptr"] = (S["pool_ptr"] + n) % cfg["K"]
    labels = torch.cat((lab_l, lab_u)); conf = torch.cat((torch.full_like(lab_l, 255).float(), conf_u))
    mask = ((labels != 0) & (conf >= 0.7)).float().unsqueeze(1)
    eqv = equivariance(S, torch.cat((l, u_mix)), mask, pred_all)
    if S["it"] > 0:                                      # the objective switches after the first iteration (train_arco_3d.py:390-393)
        loss = cfg["k1"] * reco + cfg["k3"] * unsup + dice + ce + cfg["k4"] * loss_q
    else:
        loss = unsup + dice + ce + eqv
    S["opt"].zero_grad()
    loss.backward()
    S["opt"].step()
    S["isd"]._momentum_update_key_encoder()
    for grp in S["opt"].param_groups:
        grp['lr'] = 0.01 * (1.0 - S["it"] / 30000) ** 0.9
    S["it"] += 1
    return dict(loss_ce=float(ce), loss_dice=float(dice), unsup_loss=float(unsup), reco_loss=float(reco), loss_eqv=float(eqv),
                loss_q=float(loss_q), loss=float(loss), bank_len=[int(m[0].shape[0]) for m in B["memobank"]], ptr=[int(p) for p in B["ptr"]],
                bank_sum=[float(m[0].double().abs().sum()) for m in B["memobank"]], pool_ptr=int(S["pool_ptr"]),
                banks_on_gpu=all(m[0].is_cuda for m in B["memobank"]))


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "g19_trainer_loop.npz"))
    C, b, patch, Q, Nn, qs, K = 4, 2, (32, 32, 32), 48, 16, 200, 4
    cfg = dict(k1=1.0, k3=1.0, k4=0.5, topk=2, K=K, Q=Q, Nn=Nn, mix="cutmix")
    random.seed(5); np.random.seed(5); torch.manual_seed(5)
    rs = np.random.RandomState(13)
    S = build(C, b, patch, qs, K, rs, g["v_bank0"])
    steps = []
    for it in range(3):
        l = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32)).to(DEV)
        u = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32)).to(DEV)
        lab = torch.from_numpy(fx.blob_labels(rs, b, patch, C)).to(DEV)
        random.seed(10 + it); np.random.seed(10 + it); torch.manual_seed(10 + it)
        steps.append(step(S, l, lab, u, cfg))
        steps[-1]["probe"] = [int(torch.randint(1 << 30, (1,))), float(np.random.uniform()), random.random()]
    sd, sde = S["student"].state_dict(), S["teacher"].state_dict()
    absum = lambda t: float(t.detach().double().abs().sum())
    end = dict(w_first=absum(sd["block_one.conv.0.weight"]), w_out=absum(sd["out_conv.weight"]), qrep0=absum(S["q_rep"][0].weight),
               qfe4=absum(S["q_fe"].fea4.weight), kfe4=absum(S["k_fe"].fea4.weight), t_first=absum(sde["block_one.conv.0.weight"]),
               rm=absum(sd["block_one.conv.1.running_mean"]))
    print("DROPIN_USER " + json.dumps(dict(steps=steps, end=end, model_file=sys.modules["model_3D"].__file__,
                                           arco_modules=sorted(k for k in sys.modules if k.startswith("arco_amd"))[:2])))


if __name__ == "__main__":
    main()
