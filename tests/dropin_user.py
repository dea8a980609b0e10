"""A USER of the drop-in boundary, written in this repository's own form (run by tests/test_dropin_user_gpu.py in a fresh interpreter):
`dropin/` first on sys.path, the reference trainers' import statements (`from model_2D import *`, `from loss_helper_3d import *`,
`from utils import losses`, `from augment import *`, `from tps.rand_tps import RandTPS`) bind arco_amd, and everything a reference-style
trainer builds ITSELF is plain torch: `q_representation` is `nn.Sequential(nn.Conv2d, nn.Conv2d)`, the optimiser `torch.optim.SGD` over
the drop-in modules' parameters, the k-FeatureExtractor EMA re-points `.data`, the banks start as CPU tensors.  None of arco_amd's
trainer machinery (flat buffers, PackPlans, graphs, row-sparse head) is involved.

The step is a sequence of small stages over a state dict (the shape of oracle/cpu_step.py), in the order the algorithm dictates
(train_arco_2d.py:283-435); its numbers are compared with tests/golden/g19_trainer_loop.npz - the reference's own loop body
executed from the reference's text over the reference's modules on CPU (oracle/gen_golden.py g19) - at north_star's 1e-3.
Prints one JSON line.   python tests/dropin_user.py <case a|b>"""
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
from tps.rand_tps import RandTPS     # noqa: E402
from augment import *                # noqa: F401,F403,E402
from loss_helper_3d import *         # noqa: F401,F403,E402
from model_2D import *               # noqa: F401,F403,E402

import fixture_inputs as fx          # noqa: E402

DEV = "cuda:0"


def build(C, b, patch, qs, K, rs):
    """Everything a reference-style trainer constructs before its loop."""
    isd = ISD(K=36, m=0.99, Ts=0.01, Tt=0.1, num_classes=C, latent_pooling_size=1, latent_feature_size=512,
              output_pooling_size=8, train_encoder=True, train_decoder=True).cuda()
    sd = fx.unet_state(21, 1, C)
    isd.model.load_state_dict(sd); isd.ema_model.load_state_dict(sd)
    for net in (isd.model, isd.ema_model):
        for m in net.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
    q_rep = nn.Sequential(nn.Conv2d(496, 496, 1, bias=False), nn.Conv2d(496, 496, 1, bias=False)).cuda()
    k_fe = FeatureExtractor(fea_dim=[256, 128, 64, 32, 16], output_dim=496).cuda()
    q_fe = FeatureExtractor(fea_dim=[256, 128, 64, 32, 16], output_dim=496).cuda()
    q_fe.load_state_dict(fx.fe_state(31))
    with torch.no_grad():
        q_rep[0].weight.copy_(fx.fe_state(32)["fea4.weight"]); q_rep[1].weight.copy_(fx.fe_state(33)["fea4.weight"])
        for pk, pq in zip(k_fe.parameters(), q_fe.parameters()):
            pk.data.copy_(pq.data); pk.requires_grad = False
    assert optim is torch.optim, "a star import re-bound the trainer's `optim`"
    opt = optim.SGD([p for grp in (isd.model, q_rep, q_fe) for p in grp.parameters() if p.requires_grad],
                    lr=0.01, weight_decay=0.0001, momentum=0.9, nesterov=True)
    random.seed(6); np.random.seed(6); torch.manual_seed(6)
    tps = RandTPS(patch[0], patch[1], batch_size=2 * b, sigma=0.01, border_padding=False, random_mirror=True,
                  random_scale=(0.8, 1.2), mode='affine').cuda()
    for m in (isd.model, isd.ema_model, q_rep, k_fe, q_fe):
        m.train()
    banks = dict(memobank=[[torch.zeros(1, 496)] for _ in range(C)], ptr=[torch.zeros(1, dtype=torch.long) for _ in range(C)], size=[qs] * C)
    pool = F.normalize(torch.from_numpy(rs.standard_normal((K, 496 * patch[0] * patch[1])).astype(np.float32)), dim=1).cuda()
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
    return (share.view(-1, 1, 1) * ce)[ce > 0].mean()


def one_hot(lab, C):
    return F.one_hot(lab.clamp(min=0).long(), C).permute(0, 3, 1, 2).float()


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
    u_mix, lab_u, conf_u = generate_unsup_data(u, lab_u, conf_u, mode=cfg["mix"])
    ema_key_extractor(S)
    student, teacher = S["student"], S["teacher"]
    pred_l, _, fm_l = student(l)
    student(l)                                           # the second labeled view (identity transform here): BatchNorm statistics only
    pred_u, _, fm_u = student(u_mix)
    tl, _, tfm_l = teacher(l)
    tu, _, tfm_u = teacher(u_mix)
    rep_l, rep_u = S["q_rep"](S["q_fe"](fm_l)), S["q_rep"](S["q_fe"](fm_u))
    rep_lt, rep_ut = S["k_fe"](tfm_l), S["k_fe"](tfm_u)
    rep_all, pred_all = torch.cat((rep_l, rep_u)), torch.cat((pred_l, pred_u))
    if cfg["k4"]:
        loss_q, keys = pool_distance_loss(S["pool"], rep_u, rep_ut, cfg["topk"])
    else:                                                # (timing mode: the revisiting term is opt-in in the trainers, off at the headline)
        loss_q, keys = torch.zeros((), device=DEV), None
    ce = F.cross_entropy(pred_l, lab_l.long())
    dice = S["dice"](torch.softmax(pred_l, dim=1), lab_l.unsqueeze(1))
    unsup = confidence_weighted_ce(pred_u, lab_u, conf_u, 0.97)
    with torch.no_grad():
        low, high = entropy_masks(pred_u, lab_l, lab_u, alpha=20.0)
        oh_l, oh_u = one_hot(lab_l, C), one_hot(lab_u, C)
        pt_l, pt_u = torch.softmax(tl, dim=1), torch.softmax(tu, dim=1)
    B = S["banks"]
    reco = compute_contra_memobank_loss(rep_all, oh_l.long(), oh_u.long(), pt_l, pt_u, low, high, B["memobank"], B["ptr"], B["size"],
                                        torch.cat((rep_lt, rep_ut)).detach(), delta_n=0.97, func="smc",
                                        num_queries=cfg["Q"], num_negatives=cfg["Nn"])[-1]
    if keys is not None:
        with torch.no_grad():                            # pool update (a ring of K rows)
            n = keys.shape[0]
            S["pool"][S["pool_ptr"]:S["pool_ptr"] + n] = keys
            S["pool_ptr"] = (S["pool_ptr"] + n) % cfg["K"]
    labels = torch.cat((lab_l, lab_u)); conf = torch.cat((torch.full_like(lab_l, 255).float(), conf_u))
    mask = ((labels != 0) & (conf >= 0.7)).float().unsqueeze(1)
    eqv = equivariance(S, torch.cat((l, u_mix)), mask, pred_all)
    loss = cfg["k1"] * reco + cfg["k3"] * unsup + dice + ce + cfg["k2"] * eqv + cfg["k4"] * loss_q
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


def time_at_headline_size(steps=6):
    """`python tests/dropin_user.py time`: the same user at BASELINE.json configs[1] size (8 + 8 images of 256 x 256, 256 queries, 512
    negatives, 4096-key banks; revisiting term off as in the trainers' default) - milliseconds per step of a reference-style trainer
    over the drop-in modules, torch's own q_representation / SGD and the CPU-side bank bookkeeping included (bench.py sub-record)."""
    import time
    C, b, patch = 4, 8, (256, 256)
    cfg = dict(k1=1.0, k2=1.0, k3=1.0, k4=0.0, topk=3, K=2, Q=256, Nn=512, mix="cutmix")
    random.seed(5); np.random.seed(5); torch.manual_seed(5)
    rs = np.random.RandomState(3)
    S = build(C, b, patch, 4096, 2, rs)
    batches = []
    for it in range(3):
        batches.append((torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32)).to(DEV),
                        torch.from_numpy(fx.blob_labels(rs, b, patch, C)).to(DEV),
                        torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32)).to(DEV)))
    for it in range(3):
        l, lab, u = batches[it % 3]
        step(S, l, lab, u, cfg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        l, lab, u = batches[it % 3]
        out = step(S, l, lab, u, cfg)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("DROPIN_USER_TIME " + json.dumps(dict(ms_per_step=round(ms, 2), steps=steps, last=out, model_file=sys.modules["model_2D"].__file__,
                                                peak_mem_gb=round(torch.cuda.max_memory_allocated() / 1e9, 2))))


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "a"
    if case == "time":
        return time_at_headline_size(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
    k2, mix = {"a": (1.0, "cutmix"), "b": (0.0, "cutout")}[case]
    C, b, patch, Q, Nn, qs, K = 4, 2, (64, 64), 64, 32, 300, 6
    cfg = dict(k1=1.0, k2=k2, k3=1.0, k4=0.5, topk=3, K=K, Q=Q, Nn=Nn, mix=mix)
    random.seed(5); np.random.seed(5); torch.manual_seed(5)
    rs = np.random.RandomState(3)
    S = build(C, b, patch, qs, K, rs)
    steps = []
    for it in range(2):
        l = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32)).to(DEV)
        u = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32)).to(DEV)
        lab = torch.from_numpy(fx.blob_labels(rs, b, patch, C)).to(DEV)
        random.seed(10 + it); np.random.seed(10 + it); torch.manual_seed(10 + it)
        steps.append(step(S, l, lab, u, cfg))
        steps[-1]["probe"] = [int(torch.randint(1 << 30, (1,))), float(np.random.uniform()), random.random()]
    sd, sde = S["student"].state_dict(), S["teacher"].state_dict()
    absum = lambda t: float(t.detach().double().abs().sum())
    end = dict(w_first=absum(sd["encoder.in_conv.conv_conv.0.weight"]), w_last=absum(sd["decoder.out_conv.weight"]),
               w_deep=absum(sd["encoder.down4.maxpool_conv.1.conv_conv.4.weight"]), qrep0=absum(S["q_rep"][0].weight),
               qrep1=absum(S["q_rep"][1].weight), qfe4=absum(S["q_fe"].fea4.weight), kfe4=absum(S["k_fe"].fea4.weight),
               t_first=absum(sde["encoder.in_conv.conv_conv.0.weight"]), rm=absum(sd["encoder.in_conv.conv_conv.1.running_mean"]),
               pool=absum(S["pool"]))
    print("DROPIN_USER " + json.dumps(dict(steps=steps, end=end, model_file=sys.modules["model_2D"].__file__,
                                           arco_modules=sorted(k for k in sys.modules if k.startswith("arco_amd"))[:2])))


if __name__ == "__main__":
    main()
