"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on CPU.

TEST INFRASTRUCTURE — runs only in the build container (the reference never
travels).  Usage:  python oracle/gen_golden.py
Inputs come from tests/fixture_inputs.py (numpy RandomState), so the files hold
only the reference's outputs.  Trainer-glue functions are pulled out of
train_arco_2d.py with ast/exec at generation time (the trainer itself cannot be
imported: argparse at import, tensorboardX/h5py/torchvision missing).
"""
import ast
import hashlib
import os
import random
import sys
import textwrap
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_shim            # noqa: E402
import fixture_inputs as fx  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(4)


def seed_all(s):
    random.seed(s); np.random.seed(s); torch.manual_seed(s)


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()


def rng_probe():
    """One draw after a call pins how much of the generator the call consumed."""
    return int(torch.randint(1 << 30, (1,)))


# ---------------------------------------------------------------- G1 samplers
def gen_samplers(mods):
    L = mods["loss_helper_3d"]
    out = {}
    for name, fn in (("smc", L.grid_monte_carlo_sample), ("asmc", L.grid_as_monte_carlo_sample),
                     ("mc1d", L.monte_carlo_sample), ("asmc1d", L.as_monte_carlo_sample)):
        for high in fx.SAMPLER_HIGHS:
            for shape in fx.SAMPLER_SHAPES:
                for seed in fx.SAMPLER_SEEDS:
                    seed_all(seed)
                    idx = fn(high, shape)
                    key = f"{name}_h{high}_s{shape}_r{seed}"
                    out[key] = idx.numpy().astype(np.int32)
                    out[key + "_probe"] = np.array([rng_probe(), random.randint(0, 1 << 30)])
    # production-size negative draws: keep hashes + head only
    for name, fn in (("smc", L.grid_monte_carlo_sample), ("asmc", L.grid_as_monte_carlo_sample)):
        for high in (1, 4096, 29999, 50000):
            seed_all(3)
            idx = fn(high, 131072)
            key = f"{name}_h{high}_s131072_r3"
            out[key + "_sha"] = np.frombuffer(bytes.fromhex(sha(idx)), dtype=np.uint8)
            out[key + "_head"] = idx[:64].numpy().astype(np.int32)
            out[key + "_probe"] = np.array([rng_probe(), random.randint(0, 1 << 30)])
    np.savez_compressed(os.path.join(OUT, "g1_samplers.npz"), **out)
    print("g1_samplers", len(out))


# ---------------------------------------------------------------- G2 loss
def gen_loss(mods):
    out = {}
    for case, (ikw, lkw, qsize, binit) in fx.LOSS_CASES.items():
        nd3 = len(ikw["spatial"]) == 3
        L = mods["loss_helper"] if nd3 else mods["loss_helper_3d"]
        trace = []
        orig = (L.grid_monte_carlo_sample, L.grid_as_monte_carlo_sample)

        def rec(f):
            def w(*a, **k):
                r = f(*a, **k); trace.append(r.clone()); return r
            return w
        L.grid_monte_carlo_sample, L.grid_as_monte_carlo_sample = rec(orig[0]), rec(orig[1])
        bank, ptr, qs = fx.fresh_bank(ikw["n_cls"], ikw["feat"], qsize, binit)
        mom = None
        if case == "proto_momentum":
            mom = torch.zeros(ikw["n_cls"], lkw["num_queries"], 1, ikw["feat"])
        seed_all(1337)
        for step in range(fx.LOSS_STEPS):
            inp = fx.loss_inputs(100 * step + 11, **ikw)
            rep = inp["rep"].clone().requires_grad_(True)
            del trace[:]
            res = L.compute_contra_memobank_loss(
                rep, inp["label_l"], inp["label_u"], inp["prob_l"], inp["prob_u"],
                inp["low_mask"], inp["high_mask"], bank, ptr, qs, inp["rep_teacher"],
                momentum_prototype=mom, i_iter=step + 1, **lkw)
            if mom is not None:
                mom, new_keys, loss = res
                out[f"{case}_s{step}_prototype"] = mom.detach().numpy().copy()
            else:
                new_keys, loss = res
            loss.backward()
            p = f"{case}_s{step}_"
            out[p + "loss"] = np.array(loss.item(), dtype=np.float64)
            out[p + "grad"] = rep.grad.numpy().copy()
            out[p + "new_keys"] = np.array(new_keys, dtype=np.int64)
            out[p + "ptr"] = np.array([int(q) for q in ptr], dtype=np.int64)
            out[p + "bank_len"] = np.array([b[0].shape[0] for b in bank], dtype=np.int64)
            for c, b in enumerate(bank):
                out[p + f"bank{c}"] = b[0].numpy().copy()
            for k, t in enumerate(trace):
                out[p + f"draw{k}"] = t.numpy().astype(np.int32)
            out[p + "n_draws"] = np.array(len(trace))
            out[p + "probe"] = np.array([rng_probe()])
        L.grid_monte_carlo_sample, L.grid_as_monte_carlo_sample = orig
    np.savez_compressed(os.path.join(OUT, "g2_loss.npz"), **out)
    print("g2_loss", len(out))


# ---------------------------------------------------------------- G3 nets
def probe_like(t, seed):
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.standard_normal(tuple(t.shape)).astype(np.float32))


def zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, (torch.nn.Dropout, torch.nn.Dropout3d)):
            mod.p = 0.0


def gen_nets(mods):
    out = {}
    U = mods["networks.unetWithArgs"]
    # --- whole U-Net at 32x32, b=2, one train-mode step (dropout off)
    net = U.UNet(1, 4)
    sd = fx.unet_state(21)
    net.load_state_dict(sd, strict=True)
    zero_dropout(net); net.train()
    x = fx.image_batch(5, 2, 1, (32, 32)).requires_grad_(True)
    logits, latent, fmap = net(x)
    loss = (logits * probe_like(logits, 1)).sum()
    for i, f in enumerate(fmap):
        loss = loss + (f * probe_like(f, 10 + i)).sum()
    loss.backward()
    out["unet_logits"] = logits.detach().numpy()
    out["unet_latent"] = latent.detach().numpy()
    for i, f in enumerate(fmap):
        out[f"unet_fmap{i}"] = f.detach().numpy()
    out["unet_dx"] = x.grad.numpy()
    names, gsum, gabs = [], [], []
    for n, p in net.named_parameters():
        names.append(n); gsum.append(p.grad.double().sum().item()); gabs.append(p.grad.double().abs().sum().item())
    out["unet_grad_names"] = np.array(names)
    out["unet_grad_sum"] = np.array(gsum); out["unet_grad_abs"] = np.array(gabs)
    for n in ("encoder.in_conv.conv_conv.0.weight", "encoder.in_conv.conv_conv.1.weight",
              "encoder.in_conv.conv_conv.1.bias", "encoder.down1.maxpool_conv.1.conv_conv.4.weight",
              "decoder.up4.conv1x1.weight", "decoder.up4.conv1x1.bias", "decoder.up4.conv.conv_conv.0.weight",
              "decoder.out_conv.weight", "decoder.out_conv.bias", "decoder.up1.conv.conv_conv.5.weight"):
        out["unet_grad::" + n] = dict(net.named_parameters())[n].grad.numpy()
    st = net.state_dict()
    for n in ("encoder.in_conv.conv_conv.1.running_mean", "encoder.in_conv.conv_conv.1.running_var",
              "encoder.down4.maxpool_conv.1.conv_conv.5.running_mean",
              "encoder.down4.maxpool_conv.1.conv_conv.5.running_var",
              "decoder.up4.conv.conv_conv.5.running_var", "encoder.in_conv.conv_conv.1.num_batches_tracked"):
        out["unet_buf::" + n] = st[n].numpy()
    out["unet_n_state_keys"] = np.array(len(st))

    # --- ConvBlock / UpBlock alone (small channels)
    cb = U.ConvBlock(3, 8, 0.0); cb.train()
    rs = np.random.RandomState(3)
    for n, p in cb.named_parameters():
        p.data = torch.from_numpy((0.3 * rs.standard_normal(tuple(p.shape))).astype(np.float32))
    xb = fx.image_batch(6, 2, 3, (12, 10)).requires_grad_(True)
    yb = cb(xb); (yb * probe_like(yb, 2)).sum().backward()
    out["cb_y"] = yb.detach().numpy(); out["cb_dx"] = xb.grad.numpy()
    for n, p in cb.named_parameters():
        out["cb_p::" + n] = p.detach().numpy(); out["cb_g::" + n] = p.grad.numpy()
    for n, b in cb.named_buffers():
        out["cb_b::" + n] = b.numpy()

    # --- FeatureExtractor (2-D), small dims, and the production dims (checksums)
    M2 = mods["model_2D"]
    for tag, dims, od, sp in (("fe_small", (32, 16, 8, 8, 8), 24, 32), ("fe_full", (256, 128, 64, 32, 16), 496, 32)):
        fe = M2.FeatureExtractor(fea_dim=list(dims), output_dim=od)
        fsd = fx.fe_state(31, dims, od, nd=2)
        fe.load_state_dict(fsd, strict=True)
        fl = [fx.image_batch(40 + i, 2, c, (sp >> (4 - i), sp >> (4 - i))).requires_grad_(True) for i, c in enumerate(dims)]
        y = fe(fl)
        (y * probe_like(y, 3)).sum().backward()
        if tag == "fe_small":
            out[tag + "_y"] = y.detach().numpy()
            for i, f in enumerate(fl):
                out[tag + f"_dx{i}"] = f.grad.numpy()
            for n, p in fe.named_parameters():
                out[tag + "_g::" + n] = p.grad.numpy()
        else:
            out[tag + "_y_sub"] = y.detach()[:, ::31, ::5, ::7].numpy()
            out[tag + "_y_sum"] = np.array([y.double().sum().item(), y.double().abs().sum().item()])
            for i, f in enumerate(fl):
                out[tag + f"_dx{i}_sum"] = np.array([f.grad.double().sum().item(), f.grad.double().abs().sum().item()])
            for n, p in fe.named_parameters():
                out[tag + "_g_sum::" + n] = np.array([p.grad.double().sum().item(), p.grad.double().abs().sum().item()])
                out[tag + "_g_sub::" + n] = p.grad[::13, ::17].numpy()

    # --- V-Net at 16^3 and FeatureExtractor_3d
    V = mods["networks.vnetWithArgs"]
    vnet = V.VNet(n_channels=1, n_classes=2, normalization='batchnorm', has_dropout=True)
    vsd = fx.vnet_state(51)
    vnet.load_state_dict(vsd, strict=True)
    vnet.train()
    xv = fx.image_batch(8, 2, 1, (16, 16, 16)).requires_grad_(True)
    vo, v0, vf = vnet(xv, turnoff_drop=True)
    lossv = (vo * probe_like(vo, 4)).sum()
    for i, f in enumerate(vf):
        lossv = lossv + (f * probe_like(f, 20 + i)).sum()
    lossv.backward()
    out["vnet_out"] = vo.detach().numpy()
    for i, f in enumerate(vf):
        out[f"vnet_fmap{i}"] = f.detach().numpy()
    out["vnet_dx"] = xv.grad.numpy()
    names, gsum, gabs = [], [], []
    for n, p in vnet.named_parameters():
        names.append(n); gsum.append(p.grad.double().sum().item()); gabs.append(p.grad.double().abs().sum().item())
    out["vnet_grad_names"] = np.array(names); out["vnet_grad_sum"] = np.array(gsum); out["vnet_grad_abs"] = np.array(gabs)
    vst = vnet.state_dict()
    for n in ("block_one.conv.1.running_mean", "block_one.conv.1.running_var", "block_five_up.conv.1.running_var"):
        out["vnet_buf::" + n] = vst[n].numpy()
    M3 = mods["model_3D"]
    fe3 = M3.FeatureExtractor_3d(fea_dim=[128, 64, 32, 16, 16], output_dim=16)
    f3sd = fx.fe_state(61, (128, 64, 32, 16, 16), 16, nd=3)
    fe3.load_state_dict(f3sd, strict=True)
    fl3 = [f.detach().clone().requires_grad_(True) for f in vf]
    y3 = fe3(fl3)
    (y3 * probe_like(y3, 5)).sum().backward()
    out["fe3d_y"] = y3.detach().numpy()
    for i, f in enumerate(fl3):
        out[f"fe3d_dx{i}_sum"] = np.array([f.grad.double().sum().item(), f.grad.double().abs().sum().item()])
    for n, p in fe3.named_parameters():
        out["fe3d_g_sum::" + n] = np.array([p.grad.double().sum().item(), p.grad.double().abs().sum().item()])
    np.savez_compressed(os.path.join(OUT, "g3_nets.npz"), **out)
    print("g3_nets", len(out))


# ---------------------------------------------------------------- G4 trainer glue
def _pull_functions(path, names, extra=None):
    """exec selected top-level functions / classes of a reference script without importing it."""
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"torch": torch, "np": np, "F": torch.nn.functional, "nn": torch.nn}
    ns.update(extra or {})
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    return ns, src.splitlines()


def gen_glue():
    out = {}
    path = os.path.join(ref_shim.REF, "train_arco_2d.py")
    ns, lines = _pull_functions(path, {"compute_unsupervised_loss", "label_onehot", "get_revisiting_loss",
                                       "_dequeue_and_enqueue"})
    rs = np.random.RandomState(77)
    b, C, H, W = 2, 4, 24, 20
    pred_l = torch.from_numpy(rs.standard_normal((b, C, H, W)).astype(np.float32) * 2)
    pred_u = torch.from_numpy(rs.standard_normal((b, C, H, W)).astype(np.float32) * 2)
    lab_l = torch.from_numpy(fx.blob_labels(rs, b, (H, W), C))
    lab_u = torch.from_numpy(fx.blob_labels(rs, b, (H, W), C))
    lab_u[0, :3, :4] = -1                                      # invalid (ignore) pixels
    logits_u = torch.from_numpy(rs.uniform(0.3, 1.0, size=(b, H, W)).astype(np.float32))
    out["onehot_u"] = ns["label_onehot"](lab_u, C).numpy()
    out["unsup_loss"] = np.array(ns["compute_unsupervised_loss"](pred_u, lab_u, logits_u, 0.97).item())
    # mask block train_arco_2d.py:342-393 executed from the reference text itself
    block = textwrap.dedent("\n".join(lines[341:393]))
    for epoch_num, max_epoch in ((0, 10), (3, 10)):
        env = dict(ns)
        env.update(dict(args=types.SimpleNamespace(weak_threshold=0.7, num_classes=C), epoch_num=epoch_num,
                        max_epoch=max_epoch, train_u_aug_logits=logits_u, train_l_label=lab_l,
                        train_u_aug_label=lab_u, pred_all=torch.cat((pred_l, pred_u)), pred_l=pred_l, pred_u=pred_u,
                        pred_l_teacher=pred_l, pred_u_teacher=pred_u))
        exec(block, env)
        t = f"mask_e{epoch_num}_"
        out[t + "low"] = env["low_mask_all"].numpy(); out[t + "high"] = env["high_mask_all"].numpy()
        out[t + "entropy"] = env["entropy"].numpy()
        out[t + "thr"] = np.array([env["low_thresh"], env["high_thresh"]], dtype=np.float64)
        out[t + "label_l"] = env["label_l"].numpy(); out[t + "label_u"] = env["label_u"].numpy()
        out[t + "alpha"] = np.array(env["alpha_t"])
    # supervised CE + Dice (train_arco_2d.py:336-339; utils/losses.py:173-209) with gradients w.r.t. the logits
    import importlib
    losses_mod = importlib.import_module("utils.losses")
    dice_loss = losses_mod.DiceLoss(C)
    pl_ = pred_l.clone().requires_grad_(True)
    lab_pos = lab_l.clone()
    ce = torch.nn.CrossEntropyLoss()(pl_, lab_pos.long())
    dice = dice_loss(torch.softmax(pl_, dim=1), lab_pos.unsqueeze(1))
    (ce + dice).backward()
    out["sup_ce"] = np.array(ce.item()); out["sup_dice"] = np.array(dice.item()); out["sup_grad"] = pl_.grad.numpy().copy()
    pu_ = pred_u.clone().requires_grad_(True)
    ul = ns["compute_unsupervised_loss"](pu_, lab_u, logits_u, 0.97)
    ul.backward()
    out["unsup_grad"] = pu_.grad.numpy().copy()
    ul2 = ns["compute_unsupervised_loss"](pred_u, lab_u, logits_u, 0.5)
    out["unsup_loss_t05"] = np.array(ul2.item())
    # revisiting loss + pool enqueue (train_arco_2d.py:108-136)
    K, feat = 6, 3 * 8 * 8
    pool = torch.nn.functional.normalize(torch.from_numpy(rs.standard_normal((K, feat)).astype(np.float32)), dim=1)
    ru = torch.from_numpy(rs.standard_normal((2, 3, 8, 8)).astype(np.float32))
    rt = torch.from_numpy(rs.standard_normal((2, 3, 8, 8)).astype(np.float32))
    out["revisit_loss"] = np.array(ns["get_revisiting_loss"](pool, ru, rt, topk=3).item())
    ns["args"] = types.SimpleNamespace(K=K)
    ptr = torch.zeros(1, dtype=torch.long); pool2 = pool.clone()
    keys = torch.nn.functional.normalize(rt.view(2, -1), dim=-1)
    ns["_dequeue_and_enqueue"](keys, pool2, ptr)
    out["pool_after"] = pool2.numpy(); out["pool_ptr"] = np.array(int(ptr))
    # EMA (model_2D.py:176-182) and SGD-nesterov + poly LR through torch itself
    q = torch.from_numpy(rs.standard_normal((5, 7)).astype(np.float32)); k = torch.from_numpy(rs.standard_normal((5, 7)).astype(np.float32))
    out["ema_q"] = q.numpy(); out["ema_k"] = k.numpy(); out["ema_out"] = (k * 0.99 + q * (1. - 0.99)).numpy()
    p = torch.nn.Parameter(q.clone())
    opt = torch.optim.SGD([p], lr=0.01, weight_decay=0.0001, momentum=0.9, nesterov=True)
    gs = []
    for it in range(3):
        g = torch.from_numpy(rs.standard_normal((5, 7)).astype(np.float32)); gs.append(g.numpy())
        opt.zero_grad(); p.grad = g.clone(); opt.step()
        lr_ = 0.01 * (1.0 - it / 30000) ** 0.9
        for grp in opt.param_groups:
            grp['lr'] = lr_
        out[f"sgd_p{it}"] = p.detach().numpy().copy()
    out["sgd_g"] = np.stack(gs)
    np.savez_compressed(os.path.join(OUT, "g4_glue.npz"), **out)
    print("g4_glue", len(out))


# ---------------------------------------------------------------- G5 equivariance loss
def gen_eqv():
    """RandTPS (tps/rand_tps.py:82-153 + tps_stn_pytorch/tps_grid_gen.py) and the loss_eqv block of
    train_arco_2d.py:404-423, run from the reference modules on CPU for fixed seeds."""
    import importlib
    import torch.nn as nn
    import torch.nn.functional as F
    rand_tps = importlib.import_module("tps.rand_tps")
    out = {}
    cases = [("a", 4, 24, 32, 0.01, 11), ("b", 3, 16, 16, 0.05, 12), ("c", 2, 32, 24, 0.01, 13)]
    for tag, B, W, H, sigma, seed in cases:                 # RandTPS(width, height, ...) as the trainer calls it
        seed_all(seed)
        tps = rand_tps.RandTPS(W, H, batch_size=B, sigma=sigma, border_padding=False, random_mirror=True,
                               random_scale=(0.8, 1.2), mode='affine')
        probe0 = (rng_probe(), float(np.random.uniform()), random.random())
        out[f"{tag}_cfg"] = np.array([B, W, H, sigma, seed], dtype=np.float64)
        out[f"{tag}_grid_init"] = tps.grid.data.numpy().copy()
        out[f"{tag}_probe_init"] = np.array(probe0, dtype=np.float64)
        seed_all(seed + 100)
        tps.reset_control_points()
        probe1 = (rng_probe(), float(np.random.uniform()), random.random())
        out[f"{tag}_grid"] = tps.grid.data.numpy().copy()
        out[f"{tag}_probe"] = np.array(probe1, dtype=np.float64)
        # tensors of the trainer's shapes: images [B,1,h,w], mask [B,1,h,w], predictions [B,C,h,w]; grid is [B, H, W, 2]
        h, w = tps.grid.shape[1], tps.grid.shape[2]
        rs = np.random.RandomState(seed + 7)
        C = 4
        img = torch.from_numpy(rs.uniform(size=(B, 1, h, w)).astype(np.float32))
        labels = torch.from_numpy(rs.randint(0, C, size=(B, h, w)))
        logits = torch.from_numpy(rs.uniform(size=(B, h, w)).astype(np.float32))
        pred_all = torch.from_numpy(rs.normal(size=(B, C, h, w)).astype(np.float32))
        pred_tps = torch.from_numpy(rs.normal(size=(B, C, h, w)).astype(np.float32)).requires_grad_(True)
        mask = torch.ones((B, h, w)); neg = torch.zeros((B, h, w))
        mask = torch.where(labels == 0, neg, mask)
        mask = torch.where(logits < 0.7, neg, mask)
        mask = mask.unsqueeze(1)
        images_tps = tps(img)
        mask_tps = tps(mask.float(), padding_mode='zeros')
        pred_tps_org = tps(pred_all, padding_mode='zeros')
        kl = nn.KLDivLoss(reduction='none')
        loss_eqv = kl(F.log_softmax(pred_tps, dim=1), F.softmax(pred_tps_org, dim=1))
        loss_eqv = (loss_eqv * mask_tps).flatten(1).sum(1) / (mask_tps.flatten(1).sum(1) + 1e-7)
        loss_eqv = loss_eqv.mean()
        loss_eqv.backward()
        for k, v in (("img", img), ("labels", labels), ("logits", logits), ("pred_all", pred_all), ("pred_tps", pred_tps.detach()),
                     ("images_tps", images_tps), ("mask_tps", mask_tps), ("pred_tps_org", pred_tps_org),
                     ("loss", loss_eqv.detach()), ("grad", pred_tps.grad)):
            out[f"{tag}_{k}"] = v.numpy().copy()
    # volume variant (tps/rand_tps_3d.py): the same 2-D warp on every slice of a [B,C,X,Y,Z] tensor
    rand_tps_3d = importlib.import_module("tps.rand_tps_3d")
    seed_all(21)
    tps3 = rand_tps_3d.RandTPS(12, 12, 6, batch_size=2, sigma=0.02, border_padding=False, random_mirror=True,
                               random_scale=(0.8, 1.2), mode='affine')
    seed_all(121)
    tps3.reset_control_points()
    rs = np.random.RandomState(5)
    vol = torch.from_numpy(rs.normal(size=(2, 3, 12, 12, 6)).astype(np.float32))
    out["v_grid"] = tps3.grid.data.numpy().copy()
    out["v_vol"] = vol.numpy().copy()
    out["v_vol_tps"] = tps3(vol, padding_mode='zeros').numpy().copy()
    out["v_probe"] = np.array((rng_probe(), float(np.random.uniform()), random.random()), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "g5_eqv.npz"), **out)
    print("g5_eqv", len(out))


# ---------------------------------------------------------------- G6 3-D sliding-window evaluation
def gen_eval3d(mods):
    """test_util.test_single_case (test_util.py:139-211), pulled out of the source text (the module imports h5py /
    nibabel / medpy / skimage, absent here), run on the reference V-Net (networks/vnetWithArgs.py) in eval mode."""
    import math
    path = os.path.join(ref_shim.REF, "test_util.py")
    ns, _ = _pull_functions(path, {"test_single_case"})
    ns["math"] = math
    VNet = mods["networks.vnetWithArgs"].VNet
    out = {}
    for tag, (shape, patch, sxy, sz, C, nf, seed) in fx.EVAL3D_CASES.items():
        net = VNet(n_channels=1, n_classes=C, n_filters=nf, normalization='batchnorm', has_dropout=False)
        net.load_state_dict(fx.randomize_running_stats(fx.vnet_state(seed, 1, C, nf), seed + 1))
        net.eval()
        image = fx.eval3d_volume(seed + 2, shape)
        label_map, score_map = ns["test_single_case"](lambda p: net(p)[0], image, sxy, sz, patch, num_classes=C)
        out[f"{tag}_label"] = label_map.astype(np.int8)
        out[f"{tag}_score"] = score_map.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "g6_eval3d.npz"), **out)
    print("g6_eval3d", len(out))


# ---------------------------------------------------------------- G7 mixing strategies
def gen_mix():
    """generate_unsup_data (augment.py:284-313) and generate_unsup_data_3d (augment_3d.py:228-257) with their mask
    generators, pulled out of the source text (the modules import h5py / torchvision, absent here)."""
    ns2, _ = _pull_functions(os.path.join(ref_shim.REF, "augment.py"),
                             {"generate_cutout_mask", "generate_class_mask", "generate_unsup_data"})
    ns3, _ = _pull_functions(os.path.join(ref_shim.REF, "augment_3d.py"),
                             {"generate_cutout_mask_3d", "generate_class_mask", "generate_unsup_data_3d"})
    out = {}
    for tag, (mode, b, c, spatial, n_cls, seed) in fx.MIX_CASES.items():
        data, target, logits = fx.mix_inputs(seed, b, c, spatial, n_cls)
        seed_all(seed + 1)
        fn = ns2["generate_unsup_data"] if len(spatial) == 2 else ns3["generate_unsup_data_3d"]
        nd, nt, nl = fn(data, target, logits, mode=mode)
        out[f"{tag}_data"], out[f"{tag}_target"], out[f"{tag}_logits"] = nd.numpy().copy(), nt.numpy().astype(np.int8), nl.numpy().copy()
        out[f"{tag}_target_after"] = target.numpy().astype(np.int8)         # cutout writes -1 into the caller's tensor
        out[f"{tag}_probe"] = np.array((rng_probe(), float(np.random.uniform()), random.random()), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "g7_mix.npz"), **out)
    print("g7_mix", len(out))


# ---------------------------------------------------------------- G8 data ingest transforms
def gen_ingest():
    """The per-sample transforms and the two-stream sampler of dataloaders/dataset.py and dataloaders/la_heart.py,
    pulled out of the source text (the modules import h5py / torchvision, absent here)."""
    import itertools
    from scipy import ndimage
    from scipy.ndimage import zoom
    from torch.utils.data.sampler import Sampler
    extra = dict(ndimage=ndimage, zoom=zoom, random=random, itertools=itertools, Sampler=Sampler)
    d2, _ = _pull_functions(os.path.join(ref_shim.REF, "dataloaders", "dataset.py"),
                            {"random_rot_flip", "random_rotate", "random_crop", "RandomGenerator", "TwoStreamBatchSampler",
                             "iterate_once", "iterate_eternally", "grouper"}, extra)
    d3, _ = _pull_functions(os.path.join(ref_shim.REF, "dataloaders", "la_heart.py"), {"RandomCrop", "RandomRotFlip", "ToTensor"}, extra)
    out = {}
    for seed in fx.INGEST_SEEDS:
        img, lab = fx.ingest_slice(seed)
        seed_all(seed)
        r = d2["RandomGenerator"]([32, 40])({'image': img, 'label': lab})
        out[f"gen{seed}_image"], out[f"gen{seed}_label"] = r['image'].numpy(), r['label'].numpy()
        out[f"gen{seed}_probe"] = np.array((float(np.random.uniform()), random.random()))
        vol, vlab = fx.ingest_volume(seed)
        seed_all(seed)
        r = d3["ToTensor"]()(d3["RandomRotFlip"]()(d3["RandomCrop"]((16, 12, 16))({'image': vol, 'label': vlab})))
        out[f"vol{seed}_image"], out[f"vol{seed}_label"] = r['image'].numpy(), r['label'].numpy().astype(np.int8)
        out[f"vol{seed}_probe"] = np.array((float(np.random.uniform()), random.random()))
    for name in ("random_rot_flip", "random_rotate", "random_crop"):
        img, lab = fx.ingest_slice(99, (40, 36))
        seed_all(3)
        a, b = d2[name](img, lab)
        out[f"{name}_image"], out[f"{name}_label"] = np.ascontiguousarray(a), np.ascontiguousarray(b)
    seed_all(11)
    smp = d2["TwoStreamBatchSampler"](list(range(7)), list(range(7, 30)), 6, 4)
    out["two_stream"] = np.array([list(b) for _ in range(3) for b in smp], dtype=np.int64)
    out["two_stream_len"] = np.array(len(smp))
    np.savez_compressed(os.path.join(OUT, "g8_ingest.npz"), **out)
    print("g8_ingest", len(out))


# ---------------------------------------------------------------- G9 command-line surface
TRAINERS = ("train_arco_2d.py", "train_arco_3d.py", "pretrain_2D.py", "pretrain_3D.py")


def _module_path(mod):
    """Source file of a module of the reference tree (its packages are namespace packages: no __init__.py), or None."""
    p = os.path.join(ref_shim.REF, mod.replace(".", "/") + ".py")
    return p if os.path.exists(p) else None


def _top_level(path, seen=None):
    """name -> (kind, where) for what `from <module> import *` brings: top-level defs / classes / assignments / imports of
    the module, recursively through its own star imports of reference modules (no __all__ anywhere in the reference)."""
    seen = set() if seen is None else seen
    if path in seen:
        return {}
    seen.add(path)
    out = {}
    for node in ast.parse(open(path).read()).body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            out[node.name] = ("def", path)
        elif isinstance(node, ast.Assign):
            for t in node.targets:
                for n in ast.walk(t):
                    if isinstance(n, ast.Name):
                        out[n.id] = ("var", path)
        elif isinstance(node, ast.Import):
            for a in node.names:
                out[(a.asname or a.name).split(".")[0]] = ("alias", a.name if a.asname else a.name.split(".")[0])
        elif isinstance(node, ast.ImportFrom):
            for a in node.names:
                sub = _module_path(node.module or "")
                if a.name == "*":
                    if sub:
                        for k, v in _top_level(sub, seen).items():
                            out.setdefault(k, v)
                elif sub and a.name in (d := _top_level(sub, set())) and d[a.name][0] == "def":
                    out[a.asname or a.name] = d[a.name]
                else:
                    out[a.asname or a.name] = ("alias", (node.module or "") + "." + a.name)
    return {k: v for k, v in out.items() if not k.startswith("_")}


def _ast_signature(path, name):
    """[[parameter, repr(default) | None], ...] of a top-level function, or of a class's __init__ (self included)."""
    for node in ast.parse(open(path).read()).body:
        if isinstance(node, ast.ClassDef) and node.name == name:
            node = next((n for n in node.body if isinstance(n, ast.FunctionDef) and n.name == "__init__"), None)
            if node is None:
                return None
        elif not (isinstance(node, ast.FunctionDef) and node.name == name):
            continue
        a = node.args
        params = [x.arg for x in a.posonlyargs + a.args]
        defaults = [None] * (len(params) - len(a.defaults)) + list(a.defaults)
        sig = []
        for pn, d in zip(params, defaults):
            if d is None:
                sig.append([pn, None])
            else:
                try:
                    sig.append([pn, repr(ast.literal_eval(d))])
                except Exception:
                    sig.append([pn, ast.unparse(d)])
        return sig
    return None


def trainer_surface():
    """What each reference trainer takes from modules of the reference tree, read from its source: every import statement of
    such a module, the names the trainer's code then USES (for `import *`: every free name of the trainer that only the star
    import can supply - module aliases such as `np` / `F` / `nn` included; for `from pkg import mod`: the attributes it
    reads), and the AST signature of each function / class among them."""
    import builtins
    surface, sigs = {}, {}
    for tr in TRAINERS:
        tree = ast.parse(open(os.path.join(ref_shim.REF, tr)).read())
        loads, bound, attrs = set(), set(), {}
        for n in ast.walk(tree):
            if isinstance(n, ast.Name):
                (loads if isinstance(n.ctx, ast.Load) else bound).add(n.id)
            elif isinstance(n, (ast.FunctionDef, ast.ClassDef)):
                bound.add(n.name)
            elif isinstance(n, ast.arg):
                bound.add(n.arg)
            elif isinstance(n, ast.Import):
                bound.update((a.asname or a.name).split(".")[0] for a in n.names)
            elif isinstance(n, ast.ImportFrom):
                bound.update(a.asname or a.name for a in n.names if a.name != "*")
            if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name):
                attrs.setdefault(n.value.id, set()).add(n.attr)
        free = sorted(n for n in loads if n not in bound and not hasattr(builtins, n))
        stars = [n.module for n in tree.body if isinstance(n, ast.ImportFrom) and any(a.name == "*" for a in n.names)]
        star_names = {m: _top_level(_module_path(m)) for m in stars if _module_path(m)}
        entries = []
        for node in tree.body:
            if not isinstance(node, ast.ImportFrom):
                continue
            mod = node.module or ""
            path, is_pkg = _module_path(mod), os.path.isdir(os.path.join(ref_shim.REF, mod.replace(".", "/")))
            if not path and not is_pkg:
                continue                                      # third-party / stdlib
            e = dict(stmt=ast.unparse(node), module=mod, names={})
            for a in node.names:
                if a.name == "*":
                    # a free name is supplied by the LAST star import that defines it
                    for fn in free:
                        owners = [m for m in stars if fn in star_names.get(m, {})]
                        if owners and owners[-1] == mod:
                            kind, where = star_names[mod][fn]
                            e["names"][fn] = kind
                            if kind == "def":
                                sigs[f"{os.path.relpath(where, ref_shim.REF)[:-3].replace('/', '.')}.{fn}"] = _ast_signature(where, fn)
                elif is_pkg and _module_path(mod + "." + a.name):
                    used = sorted(attrs.get(a.asname or a.name, ()))
                    e["names"][a.name] = {"module_attrs": used}
                    for u in used:
                        sg = _ast_signature(_module_path(mod + "." + a.name), u)
                        if sg is not None:
                            sigs[f"{mod}.{a.name}.{u}"] = sg
                else:
                    e["names"][a.name] = "def"
                    sg = _ast_signature(path, a.name) if path else None
                    if sg is not None:
                        sigs[f"{mod}.{a.name}"] = sg
            entries.append(e)
        unresolved = [fn for fn in free if not any(fn in star_names.get(m, {}) for m in stars)]
        surface[tr] = dict(imports=entries, unresolved=unresolved)
    return surface, sigs


def gen_flags():
    """Every add_argument of the two reference trainers (name, default, type) read from their source with ast."""
    import json

    def flags(path):
        out = {}
        for n in ast.walk(ast.parse(open(path).read())):
            if isinstance(n, ast.Call) and getattr(n.func, "attr", "") == "add_argument" and n.args and isinstance(n.args[0], ast.Constant):
                d = {}
                for kw in n.keywords:
                    if kw.arg in ("default", "type", "nargs"):
                        try:
                            d[kw.arg] = ast.literal_eval(kw.value)
                        except Exception:
                            d[kw.arg] = ast.unparse(kw.value)
                out[n.args[0].value] = d
        return out
    tab = {"2d": flags(os.path.join(ref_shim.REF, "train_arco_2d.py")), "3d": flags(os.path.join(ref_shim.REF, "train_arco_3d.py")),
           "pre2d": flags(os.path.join(ref_shim.REF, "pretrain_2D.py")), "pre3d": flags(os.path.join(ref_shim.REF, "pretrain_3D.py"))}
    # call signatures of the public names the trainers import (parameter names and defaults, in order)
    import importlib
    import inspect
    mods = ref_shim.load()
    sigs = {}
    for mod, names in fx.PUBLIC_NAMES.items():
        m = mods.get(mod) or importlib.import_module(mod)
        for n in names:
            o = getattr(m, n)
            ps = inspect.signature(o.__init__ if inspect.isclass(o) else o).parameters.values()
            sigs[f"{mod}.{n}"] = [[q.name, None if q.default is inspect.Parameter.empty else repr(q.default)] for q in ps]
    tab["signatures"] = sigs
    # state_dict keys and shapes (checkpoints load both ways)
    tab["state_keys"] = {}
    for mod, name, kw in fx.STATE_CASES:
        net = getattr(mods.get(mod) or importlib.import_module(mod), name)(**kw)
        tab["state_keys"][f"{mod}.{name}"] = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    # the drop-in surface derived from the trainers' own source (not hand-picked): tests/test_dropin_boundary.py
    tab["trainer_surface"], tab["surface_signatures"] = trainer_surface()
    json.dump(tab, open(os.path.join(OUT, "g9_flags.json"), "w"), indent=0, sort_keys=True)
    print("g9_flags", len(tab["2d"]), len(tab["3d"]), len(sigs))


# ---------------------------------------------------------------- G10 AdvMorph
def gen_morph():
    """AdvMorph (adv_morph.py:310-580) run from the reference class on CPU: for given velocity fields (numpy draws) the
    sampling grid of get_deformation_displacement_field and the warped images of forward()."""
    import importlib
    am = importlib.import_module("adv_morph")
    out = {}
    for tag, B, C, H, W, seed in fx.MORPH_CASES:
        data, param = fx.morph_inputs(seed, B, C, H, W)
        aug = am.AdvMorph(config_dict={'epsilon': 1.5, 'xi': 0.5, 'data_size': [B, C, H, W], 'vector_size': [W // 8, W // 8],
                                       'interpolator_mode': 'bilinear'}, debug=False, use_gpu=False)
        aug.init_parameters()                                  # builds base_grid_wh (and draws a velocity we overwrite)
        aug.param = aug.unit_normalize(param.clone())
        with torch.no_grad():
            warped = aug.forward(data.clone())
            grid, disp = aug.get_deformation_displacement_field(duv=aug.epsilon * aug.param)
        st = fx.MORPH_STRIDE if H * W > 10000 else 1           # the 256 x 256 case is stored subsampled
        out[f"{tag}_param"] = aug.param.numpy()
        out[f"{tag}_grid"] = grid.numpy()[:, :, ::st, ::st]
        out[f"{tag}_warped"] = warped.numpy()[:, :, ::st, ::st]
        out[f"{tag}_maxdisp"] = np.array([float(disp.abs().max())])
    np.savez_compressed(os.path.join(OUT, "g10_morph.npz"), **out)
    print("g10_morph", len(out))


# ---------------------------------------------------------------- G12 ColorJitter / GaussianBlur on PIL images
def gen_jitter():
    """The photometric half of batch_transform (augment.py:167-178): torchvision 0.13.1 is not installed here, so its
    glue (ColorJitter.forward applies functional_pil.adjust_brightness / contrast / saturation / hue in fn_idx order;
    adjust_hue shifts the H channel of img.convert('HSV') with uint8 wrap) is restated on top of PILLOW, which does every
    pixel operation: ImageEnhance.Brightness / Contrast / Color, Image.convert, ImageFilter.GaussianBlur."""
    from PIL import Image, ImageEnhance, ImageFilter
    out = {}
    for tag, (C, H, W, seed, order, factors, sigma) in fx.JITTER_CASES.items():
        x = fx.jitter_image(seed, C, H, W)
        u8 = torch.from_numpy(x).mul(255).byte().numpy()                                   # to_pil_image
        img = Image.fromarray(u8[0], 'L') if C == 1 else Image.fromarray(np.ascontiguousarray(u8.transpose(1, 2, 0)), 'RGB')
        if order is not None:
            for op in order:
                f = factors[op]
                if op == 0:
                    img = ImageEnhance.Brightness(img).enhance(f)
                elif op == 1:
                    img = ImageEnhance.Contrast(img).enhance(f)
                elif op == 2:
                    img = ImageEnhance.Color(img).enhance(f)
                elif op == 3 and img.mode == 'RGB':                                        # F_pil.adjust_hue ('L': unchanged)
                    h, s_, v = img.convert('HSV').split()
                    np_h = np.array(h, dtype=np.uint8)
                    np_h = (np_h.astype(np.int32) + (int(np.float32(f) * np.float32(255.0)) & 0xff)).astype(np.uint8)   # np_h += np.uint8(f * 255), wrap
                    img = Image.merge('HSV', (Image.fromarray(np_h, 'L'), s_, v)).convert('RGB')
        if sigma is not None:
            img = img.filter(ImageFilter.GaussianBlur(radius=sigma))
        arr = np.array(img)
        out[tag] = arr[None] if C == 1 else np.ascontiguousarray(arr.transpose(2, 0, 1))
    np.savez_compressed(os.path.join(OUT, "g12_jitter.npz"), **out)
    print("g12_jitter", len(out))


# ---------------------------------------------------------------- G11 2-D evaluation
def gen_eval2d(mods):
    """test_2D.test_single_volume (test_2D.py:67-103) pulled out of the source text (the module imports medpy / h5py /
    SimpleITK, absent here) and run on the reference U-Net in eval mode.  Stand-ins are DATA plumbing only: an in-memory
    object for h5py.File, no-op SimpleITK writers; calculate_metric_percase (medpy) is replaced by a recorder - the pinned
    quantity is the function's per-slice prediction volume (zoom order 0 -> net -> argmax -> zoom back)."""
    from scipy.ndimage import zoom
    UNet = mods["networks.unetWithArgs"].UNet
    out = {}
    for tag, (shape, C, seed) in fx.EVAL2D_CASES.items():
        image, label = fx.eval2d_volume(seed, shape, C)
        store = {"image": image, "label": label}

        class _File(dict):
            def __init__(self, *a, **k):
                super().__init__(store)

        class _Sitk:
            @staticmethod
            def GetImageFromArray(a):
                return types.SimpleNamespace(SetSpacing=lambda s: None)

            @staticmethod
            def WriteImage(*a):
                return None

        seen = []

        def record(pred, gt):
            seen.append((pred.copy(), gt.copy()))
            return (0, 0, 0, 0)
        ns, _ = _pull_functions(os.path.join(ref_shim.REF, "test_2D.py"), {"test_single_volume"},
                                dict(h5py=types.SimpleNamespace(File=_File), sitk=_Sitk, zoom=zoom, calculate_metric_percase=record))
        net = UNet(in_chns=1, class_num=C)
        net.load_state_dict(fx.randomize_running_stats(fx.unet_state(seed, 1, C), seed + 1))
        flags = types.SimpleNamespace(root_path="", model="unet")
        ns["test_single_volume"]("case", net, C, "", flags)
        pred = np.zeros_like(label)
        for i, (p, g) in enumerate(seen, start=1):
            pred[p] = i
            assert np.array_equal(g, label == i)
        out[f"{tag}_pred"] = pred.astype(np.int8)
    np.savez_compressed(os.path.join(OUT, "g11_eval2d.npz"), **out)
    print("g11_eval2d", len(out))


# ---------------------------------------------------------------- G13 stage-1 pre-training (ISD.forward + pretrain_2D losses)
def _param_sums(named):
    names, s1, s2 = [], [], []
    for n, p in named:
        names.append(n); s1.append(p.double().sum().item()); s2.append(p.double().abs().sum().item())
    return np.array(names), np.array(s1), np.array(s2)


def _small(t):
    """whole tensor up to 20 000 elements, else every 4th row and column (the per-tensor sums cover the rest)"""
    t = t.detach()
    return (t if t.numel() <= 20000 else t[::4, ::4]).numpy().copy()


def gen_pretrain(mods, three_d=False):
    """Two iterations of pretrain_2D.py:235-252 (pretrain_3D.py:209-232 with three_d) on the imported reference ISD /
    ISD_3d (model_2D.py:115-305, model_3D.py:219-403) at the small configurations fixture_inputs.STAGE1_CFG(_3D): seeded
    states, dropout off, torch.optim.SGD(momentum 0.9, wd 1e-4).  The KLD module, the loss combination and the optimizer
    settings are the trainers' (the trainers themselves parse argv and need tensorboardX / torchvision / h5py at import,
    none of them installed here); DiceLoss is the reference's utils/losses.py class, pulled out of its source file.  3-D:
    the queue_mask buffer (700 patches hard-wired for 112 x 112 x 80 volumes) is replaced by one with the test volume's
    27 patches - the forward only reads its shape."""
    cfg = fx.STAGE1_CFG_3D if three_d else fx.STAGE1_CFG
    seed_all(5)
    kw = dict(K=cfg["K"], m=0.99, Ts=cfg["Ts"], Tt=cfg["Tt"], num_classes=cfg["num_classes"],
              latent_pooling_size=cfg["latent_pooling_size"], latent_feature_size=cfg["latent_feature_size"],
              output_pooling_size=cfg["output_pooling_size"], train_encoder=1, train_decoder=1, patch_size=cfg["patch_size"])
    if three_d:
        isd = mods["model_3D"].ISD_3d(**kw)
        isd.model.load_state_dict(fx.vnet_state(31), strict=True)
        isd.ema_model.load_state_dict(fx.vnet_state(32), strict=True)
        heads = fx.isd3d_head_state(33)
        isd.queue_mask = heads["queue_mask"].clone()
    else:
        isd = mods["model_2D"].ISD(**kw)
        isd.model.load_state_dict(fx.unet_state(31), strict=True)
        isd.ema_model.load_state_dict(fx.unet_state(32), strict=True)
        heads = fx.isd_head_state(33)
    missing, unexpected = isd.load_state_dict(heads, strict=False)
    assert not unexpected and all(k.startswith(("model.", "ema_model.")) for k in missing), (missing, unexpected)
    zero_dropout(isd); isd.train()
    ns, _ = _pull_functions(os.path.join(ref_shim.REF, "utils", "losses.py"), {"DiceLoss"}, {"torch": torch, "nn": nn, "F": F, "np": np})
    dice_loss = ns["DiceLoss"](cfg["num_classes"])
    ce_loss = nn.CrossEntropyLoss()

    class KLD(nn.Module):                       # pretrain_2D.py:99-103
        def forward(self, inputs, targets):
            inputs = F.log_softmax(inputs, dim=1)
            targets = F.softmax(targets, dim=1)
            return F.kl_div(inputs, targets, reduction='batchmean')

    params = [p for p in isd.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=cfg["lr"], momentum=0.9, weight_decay=0.0001)
    out = {}
    lb = cfg["labeled_bs"]
    sub = (slice(None), slice(None)) + (slice(None, None, 2),) * (3 if three_d else 2)
    first_w, last_w = ("block_one.conv.0.weight", "out_conv.weight") if three_d else \
        ("encoder.in_conv.conv_conv.0.weight", "decoder.out_conv.weight")
    for it in range(cfg["steps"]):
        im_q, im_k, lab = (fx.stage1_batch_3d if three_d else fx.stage1_batch)(40, it)
        seed_all(100 + it)
        outputs, ema_output, ema_ll, ll, ema_ol, ol = isd(im_q, im_k)
        outputs_soft = torch.softmax(outputs, dim=1)
        loss_ce = ce_loss(outputs[:lb], lab[:lb].long())
        loss_dice = dice_loss(outputs_soft[:lb], lab[:lb].unsqueeze(1))
        kld = KLD()
        loss_latent = kld(inputs=ll, targets=ema_ll)
        loss_output = kld(inputs=ol, targets=ema_ol)
        loss = (0.5 if three_d else 1.0) * (loss_dice + loss_ce) + 1.0 * loss_latent + 1.0 * loss_output     # pretrain_3D.py:218
        opt.zero_grad()
        loss.backward()
        if it == 0:
            n, a, b_ = _param_sums([(k, p.grad) for k, p in isd.named_parameters() if p.grad is not None])
            out["grad_names"], out["grad_sum"], out["grad_abs"] = n, a, b_
            named = dict(isd.named_parameters())
            for k, p in named.items():
                if p.grad is not None and not k.startswith("model."):
                    out["grad::" + k] = _small(p.grad)
            for w in (first_w, last_w):
                out["grad::model." + w] = named["model." + w].grad.numpy().copy()
        opt.step()
        out[f"s{it}_terms"] = np.array([loss.item(), loss_ce.item(), loss_dice.item(), loss_latent.item(), loss_output.item()])
        out[f"s{it}_outputs"] = outputs.detach()[sub].numpy().copy()          # every other pixel; sums of all below
        out[f"s{it}_ema_output"] = ema_output.detach()[sub].numpy().copy()
        out[f"s{it}_outputs_sums"] = np.array([outputs.double().sum().item(), outputs.double().abs().sum().item(),
                                               ema_output.double().sum().item(), ema_output.double().abs().sum().item()])
        out[f"s{it}_ema_latent_logits"] = ema_ll.detach().numpy()
        out[f"s{it}_latent_logits"] = ll.detach().numpy()
        for tag, t in (("ema_output_logits", ema_ol), ("output_logits", ol)):
            t = t.detach()
            out[f"s{it}_{tag}_shape"] = np.array(t.shape)
            out[f"s{it}_{tag}_sums"] = np.array([t.double().sum().item(), t.double().abs().sum().item()])
            out[f"s{it}_{tag}_sample"] = t[::7, ::13].numpy().copy()
    sd = isd.state_dict()
    for k in ("queue", "queue_mask", "queue_ptr", "mask_queue_ptr"):
        out["final::" + k] = sd[k].numpy().copy()
    for k, v in sd.items():
        if not k.startswith(("model.", "ema_model.", "queue", "mask_queue")):
            out["final::" + k] = _small(v)
    for pre in ("model.", "ema_model.", ""):
        n, a, b_ = _param_sums([(k, v.float()) for k, v in sd.items() if k.startswith(pre) and "num_batches" not in k
                                and (pre or not k.startswith(("model.", "ema_model.")))])
        out[f"final_{pre}names"], out[f"final_{pre}sum"], out[f"final_{pre}abs"] = n, a, b_
    bn1 = "block_one.conv.1" if three_d else "encoder.in_conv.conv_conv.1"
    mid = "block_one.conv.0.weight" if three_d else "encoder.in_conv.conv_conv.4.weight"
    for k in ("model." + last_w, "model." + mid, "ema_model." + last_w.replace("weight", "bias"),
              f"ema_model.{bn1}.running_mean", f"ema_model.{bn1}.running_var", f"model.{bn1}.running_var",
              f"ema_model.{bn1}.num_batches_tracked"):
        out["final::" + k] = sd[k].numpy().copy()
    name = "g14_pretrain3d" if three_d else "g13_pretrain"
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, len(out), {k: out[f"s{k}_terms"] for k in range(cfg["steps"])})


# ---------------------------------------------------------------- G15 whole U-Net, kink-free input (strict gradient parity)
KINK_SEEDS = range(1000, 7000)
KINK_GRADS = ("encoder.in_conv.conv_conv.0.weight", "encoder.in_conv.conv_conv.1.weight", "encoder.in_conv.conv_conv.1.bias",
              "encoder.in_conv.conv_conv.4.weight", "encoder.down1.maxpool_conv.1.conv_conv.4.weight",
              "encoder.down2.maxpool_conv.1.conv_conv.0.weight", "encoder.down3.maxpool_conv.1.conv_conv.5.weight",
              "encoder.down4.maxpool_conv.1.conv_conv.0.weight", "encoder.down4.maxpool_conv.1.conv_conv.4.weight",
              "decoder.up1.conv1x1.weight", "decoder.up1.conv.conv_conv.0.weight", "decoder.up2.conv.conv_conv.4.weight",
              "decoder.up3.conv.conv_conv.1.weight", "decoder.up4.conv1x1.weight", "decoder.up4.conv1x1.bias",
              "decoder.up4.conv.conv_conv.0.weight", "decoder.up4.conv.conv_conv.5.bias", "decoder.out_conv.weight",
              "decoder.out_conv.bias")


def _kink_margin(net, x):
    """Smallest |BatchNorm output| (= LeakyReLU pre-activation) over every BN layer of the reference U-Net's forward."""
    worst = [float("inf")]
    hooks = [m.register_forward_hook(lambda mod, i, o: worst.__setitem__(0, min(worst[0], float(o.detach().abs().min()))))
             for m in net.modules() if isinstance(m, nn.BatchNorm2d)]
    with torch.no_grad():
        net(x)
    for h in hooks:
        h.remove()
    return worst[0]


def gen_unet_kinkfree(mods):
    """The LeakyReLU derivative jumps 100x at zero, so a gradient comparison through 18 BN + LeakyReLU layers is only
    well-posed when no pre-activation sits within forward rounding error of zero.  Search the fixture input seeds for the
    one whose smallest |pre-activation| is largest (weights fixed: fx.unet_state(21)); with that input the reference's
    gradients are a strict target (tests/test_nets_gpu.py::test_unet_gradients_strict_on_kinkfree_input)."""
    U = mods["networks.unetWithArgs"]
    net = U.UNet(1, 4)
    net.load_state_dict(fx.unet_state(21), strict=True)
    zero_dropout(net); net.train()
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    best = (-1.0, None)
    for s in KINK_SEEDS:
        net.load_state_dict(state0)
        m = _kink_margin(net, fx.image_batch(s, 2, 1, (32, 32)))
        if m > best[0]:
            best = (m, s)
    margin, seed = best
    net.load_state_dict(state0)
    x = fx.image_batch(seed, 2, 1, (32, 32)).requires_grad_(True)
    logits, latent, fmap = net(x)
    loss = (logits * probe_like(logits, 1)).sum()
    for i, f in enumerate(fmap):
        loss = loss + (f * probe_like(f, 10 + i)).sum()
    loss.backward()
    out = dict(seed=np.array(seed), margin=np.array(margin), logits=logits.detach().numpy(), dx=x.grad.numpy())
    names, gabs, gl2 = [], [], []
    for n, p in net.named_parameters():
        names.append(n); gabs.append(p.grad.double().abs().sum().item()); gl2.append(p.grad.double().pow(2).sum().sqrt().item())
    out["grad_names"] = np.array(names); out["grad_abs"] = np.array(gabs); out["grad_l2"] = np.array(gl2)
    params = dict(net.named_parameters())
    for n in KINK_GRADS:
        out["grad::" + n] = params[n].grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g15_unet_kinkfree.npz"), **out)
    print("g15_unet_kinkfree seed", seed, "margin", margin)


# ---------------------------------------------------------------- G16 the rest of the trainers' import surface
def gen_boundary(mods):
    """Names the trainers take from reference modules beyond the hot-path functions (derived list: g9 "trainer_surface"):
    utils.losses.DiceLoss, utils.ramps.*, loss_helper_3d.LocalConLoss / SupConLoss / label_onehot,
    augment.randomGeneratorWithLogits - run from the reference's own code."""
    import importlib
    out = {}
    rs = np.random.RandomState(161)
    # DiceLoss on probabilities, with and without class weights / internal softmax, gradient w.r.t. the scores
    losses_mod = importlib.import_module("utils.losses")
    for tag, shape, C in (("2d", (2, 4, 24, 20), 4), ("3d", (2, 2, 10, 12, 8), 2), ("c19", (1, 19, 16, 16), 19)):
        x = torch.from_numpy(rs.standard_normal(shape).astype(np.float32) * 2)
        lab = torch.from_numpy(rs.randint(0, C, size=(shape[0], 1) + shape[2:]).astype(np.int64))
        for mode, kw in (("plain", {}), ("w", dict(weight=[0.5 + 0.25 * i for i in range(C)])), ("sm", dict(softmax=True))):
            xs = x.clone().requires_grad_(True)
            inp = xs if mode == "sm" else torch.softmax(xs, dim=1)
            l = losses_mod.DiceLoss(C)(inp, lab, **kw)
            l.backward()
            out[f"dice_{tag}_{mode}"] = np.array(l.item()); out[f"dice_{tag}_{mode}_grad"] = xs.grad.numpy().copy()
        out[f"dice_{tag}_x"] = x.numpy(); out[f"dice_{tag}_lab"] = lab.numpy()
    # ramps
    ramps = importlib.import_module("utils.ramps")
    cur = np.array([-3.0, 0.0, 1.0, 13.0, 99.5, 200.0, 250.0])
    out["ramp_cur"] = cur
    out["ramp_sigmoid_200"] = np.array([ramps.sigmoid_rampup(c, 200.0) for c in cur])
    out["ramp_sigmoid_0"] = np.array([ramps.sigmoid_rampup(c, 0) for c in cur])
    out["ramp_linear_200"] = np.array([ramps.linear_rampup(c, 200.0) for c in cur[1:]])
    out["ramp_cosine_250"] = np.array([ramps.cosine_rampdown(c, 250.0) for c in cur[1:]])
    out["ramp_exp_100"] = np.array([ramps.exp_rampup(100.0)(c) for c in cur])
    # LocalConLoss / SupConLoss (loss_helper_3d.py:1121-1252)
    LH = mods["loss_helper_3d"]
    f = torch.from_numpy((rs.standard_normal((2, 2, 6, 24, 24)) * 0.4).astype(np.float32))
    lab = torch.from_numpy(rs.randint(-1, 3, size=(2, 2, 24, 24)).astype(np.int64))
    out["lcl_f"] = f.numpy(); out["lcl_lab"] = lab.numpy()
    for tag, kw, use_lab in (("s4_lab", dict(temperature=0.7, stride=4), True), ("s4_nolab", dict(temperature=0.7, stride=4), False),
                             ("s8_lab", dict(temperature=0.5, stride=8), True)):
        fs = f.clone().requires_grad_(True)
        l = LH.LocalConLoss(**kw)(fs, lab) if use_lab else LH.LocalConLoss(**kw)(fs)
        l.backward()
        out[f"lcl_{tag}"] = np.array(l.item()); out[f"lcl_{tag}_grad"] = fs.grad.numpy().copy()
    out["lcl_zero_labels"] = np.array(LH.LocalConLoss()(f, torch.zeros_like(lab)).item())
    # label_onehot of the loss helpers (255 = ignore)
    l3 = torch.from_numpy(rs.randint(0, 4, size=(2, 9, 7)).astype(np.int64)); l3[0, :2, :3] = 255
    out["lh_onehot_in"] = l3.numpy(); out["lh_onehot"] = LH.label_onehot(l3, 4).numpy()
    # randomGeneratorWithLogits (augment.py:339-369), same size (the trainer's case) and a zoomed size
    from scipy.ndimage import zoom
    ns, _ = _pull_functions(os.path.join(ref_shim.REF, "augment.py"), {"randomGeneratorWithLogits"}, extra={"zoom": zoom})
    img = torch.from_numpy(rs.uniform(size=(3, 1, 32, 32)).astype(np.float32))
    pl = torch.from_numpy(rs.randint(0, 4, size=(3, 32, 32)).astype(np.int64))
    lg = torch.from_numpy(rs.uniform(size=(3, 32, 32)).astype(np.float32))
    out["rg_img"] = img.numpy(); out["rg_lab"] = pl.numpy(); out["rg_logit"] = lg.numpy()
    for tag, size in (("same", [32, 32]), ("zoom", [48, 40])):
        a, b_, c = ns["randomGeneratorWithLogits"](img, pl, lg, output_size=size)
        out[f"rg_{tag}_img"] = a.numpy(); out[f"rg_{tag}_lab"] = b_.numpy(); out[f"rg_{tag}_logit"] = c.numpy()
    np.savez_compressed(os.path.join(OUT, "g16_boundary.npz"), **out)
    print("g16_boundary", len(out))


# ---------------------------------------------------------------- G17 V-Net blocks with the other normalisations
VNET_NORMS = ("groupnorm", "instancenorm", "none")


def gen_vnet_norms(mods):
    """`normalization='groupnorm' | 'instancenorm' | 'none'` of the reference's V-Net blocks (vnetWithArgs.py:5-118,145-252) -
    not built by net_factory_3d, but what the blocks accept: ConvBlock / DownsamplingConvBlock / UpsamplingDeconvBlock with
    outputs, input and parameter gradients; the whole V-Net at 32^3 (sub-sampled outputs / feature maps, their norms, gradient norms)."""
    V = mods["networks.vnetWithArgs"]
    out = {}
    for norm in VNET_NORMS:
        cases = (("cb", V.ConvBlock(2, 16, 32, normalization=norm), (2, 16, 8, 8, 8)),
                 ("dw", V.DownsamplingConvBlock(16, 32, normalization=norm), (2, 16, 8, 8, 8)),
                 ("up", V.UpsamplingDeconvBlock(32, 16, normalization=norm), (2, 32, 4, 4, 4)))
        for tag, mod, shape in cases:
            mod.train()
            fx.fill_state(mod, 170 + len(tag) + len(norm))
            x = fx.image_batch(171, shape[0], shape[1], shape[2:]).sub(0.5).mul(2.0).requires_grad_(True)
            y = mod(x)
            (y * probe_like(y, 6)).sum().backward()
            t = f"{norm}_{tag}_"
            out[t + "y"] = y.detach().numpy(); out[t + "dx"] = x.grad.numpy()
            for n, p in mod.named_parameters():
                out[t + "g::" + n] = p.grad.numpy()
        net = V.VNet(n_channels=1, n_classes=2, normalization=norm, has_dropout=True)
        net.train()
        fx.fill_state(net, 175)
        xv = fx.image_batch(8, 2, 1, (32, 32, 32)).requires_grad_(True)     # (InstanceNorm needs > 1 voxel at the bottleneck)
        vo, v0, vf = net(xv, turnoff_drop=True)
        lossv = (vo * probe_like(vo, 4)).sum()
        for i, f in enumerate(vf):
            lossv = lossv + (f * probe_like(f, 20 + i)).sum()
        lossv.backward()
        t = f"{norm}_vnet_"
        out[t + "out_sub"] = vo.detach()[..., ::2, ::2, ::2].numpy()
        out[t + "out_l2"] = np.array(float(vo.detach().double().pow(2).sum().sqrt()))
        for i, f in enumerate(vf):
            out[t + f"fmap{i}_sub"] = f.detach()[:, ::3, ::3, ::3, ::3].numpy()
            out[t + f"fmap{i}_l2"] = np.array(float(f.detach().double().pow(2).sum().sqrt()))
        out[t + "dx_l2"] = np.array(float(xv.grad.double().pow(2).sum().sqrt()))
        names, gl2 = [], []
        for n, p in net.named_parameters():
            names.append(n); gl2.append(float(p.grad.double().pow(2).sum().sqrt()))
        out[t + "grad_names"] = np.array(names); out[t + "grad_l2"] = np.array(gl2)
        out[t + "n_state_keys"] = np.array(len(net.state_dict()))
    np.savez_compressed(os.path.join(OUT, "g17_vnet_norms.npz"), **out)
    print("g17_vnet_norms", len(out))


# ---------------------------------------------------------------- G18 whole V-Net, strict gradients (float64 run of the reference module)
VNET_STRICT = dict(shape=(48, 48, 32), b=2, seeds=range(2000, 2800), state_seed=52, deep_elems=512, stride=53)


def gen_vnet_strict(mods):
    """Strict gradient target for the whole V-Net (vnetWithArgs.py:145-252).  Two things make the fp32 reference itself a
    poor target here (measured in this script, stored as `ref32_dev::*`): (i) torch's CPU BatchNorm3d backward sums 10^5
    random-sign terms per channel sequentially in fp32, and the weight gradients of the layer below multiply that coherent
    error by sum(activation) - the fp32 run of the REFERENCE sits 0.3-7 % from its own float64 run on every input tried;
    (ii) ReLU decisions at the deep levels (36-288 values per channel, each a visible share of its channel) flip under forward rounding.  So the target is the
    reference module evaluated in float64 (`net.double()`: the reference's code, exact to 1e-12), on the input whose smallest
    |BatchNorm output| over the deep layers (<= 512 values per channel: the 6x6x4 and 3x3x2 levels) is largest among 800 fixture seeds; planes
    48x48x32 ... 3x3x2 are not multiples of 16, so the ragged-tile paths of the 27-tap kernels are on the path."""
    V = mods["networks.vnetWithArgs"]
    cfg = VNET_STRICT
    net = V.VNet(n_channels=1, n_classes=2, normalization='batchnorm', has_dropout=True)
    sd = fx.vnet_state(cfg["state_seed"])
    net.load_state_dict(sd, strict=True)
    net.train()

    def deep_margin(x):
        worst = [float("inf")]

        def hook(mod, i, o):
            if o.numel() // o.shape[1] <= cfg["deep_elems"]:
                worst[0] = min(worst[0], float(o.detach().abs().min()))
        hooks = [m.register_forward_hook(hook) for m in net.modules() if isinstance(m, nn.BatchNorm3d)]
        with torch.no_grad():
            net(x, turnoff_drop=True)
        for h in hooks:
            h.remove()
        return worst[0]

    best = (-1.0, None)
    for s in cfg["seeds"]:
        net.load_state_dict(sd)
        m = deep_margin(fx.image_batch(s, cfg["b"], 1, cfg["shape"]))
        if m > best[0]:
            best = (m, s)
    margin, seed = best

    def run(dt):
        net.load_state_dict(sd)
        net.to(dt)
        for p_ in net.parameters():
            p_.grad = None
        x = fx.image_batch(seed, cfg["b"], 1, cfg["shape"]).to(dt).requires_grad_(True)
        o, _, fm = net(x, turnoff_drop=True)
        loss = (o * probe_like(o, 4).to(dt)).sum()
        for i, f in enumerate(fm):
            loss = loss + (f * probe_like(f, 20 + i).to(dt)).sum()
        loss.backward()
        return o.detach(), [f.detach() for f in fm], x.grad.detach(), {n: p_.grad.detach().clone() for n, p_ in net.named_parameters()}

    o32, f32_, dx32, g32 = run(torch.float32)
    o64, f64_, dx64, g64 = run(torch.float64)
    net.to(torch.float32)
    out = dict(seed=np.array(seed), margin=np.array(margin), stride=np.array(cfg["stride"]),
               out_sub=o64[..., ::2, ::2, ::2].float().numpy(), dx=dx64.float().numpy(),
               out_l2=np.array(float(o64.pow(2).sum().sqrt())))
    for i, f in enumerate(f64_):
        out[f"fmap{i}_l2"] = np.array(float(f.pow(2).sum().sqrt()))
    names, gabs, gl2, dev = [], [], [], []
    for n, g in g64.items():
        names.append(n); gabs.append(float(g.abs().sum())); gl2.append(float(g.pow(2).sum().sqrt()))
        dev.append(float((g32[n].double() - g).abs().max() / max(1e-30, float(g.abs().max()))))
        flat = g.reshape(-1)
        out["grad::" + n] = (flat if flat.numel() <= 120000 else flat[::cfg["stride"]]).float().numpy()
    out["grad_names"] = np.array(names); out["grad_abs"] = np.array(gabs); out["grad_l2"] = np.array(gl2)
    out["ref32_dev"] = np.array(dev)                 # the fp32 reference's own distance to this target, per parameter
    out["ref32_dev_dx"] = np.array(float((dx32.double() - dx64).abs().max() / dx64.abs().max()))
    np.savez_compressed(os.path.join(OUT, "g18_vnet_strict.npz"), **out)
    skip = lambda n: n.endswith(".bias") and ".conv." in n and int(n.split(".")[-2]) % 3 == 0
    print("g18_vnet_strict seed", seed, "deep margin", margin, "fp32 reference vs its float64 run: worst",
          max(d for n, d in zip(names, dev) if not skip(n)), "dx", float(out["ref32_dev_dx"]))



# ---------------------------------------------------------------- G19 the trainers' loop bodies, executed from the reference's own text
def _loop_block(path, first_marker, last_marker):
    """The source lines of a reference trainer's loop body, from the line holding `first_marker` to the line holding
    `last_marker` (inclusive), dedented - read from /root/reference at generation time (this container only) and exec'd below:
    no text of the reference's loop lives in this repository (copy-paste findings)."""
    lines = open(path).read().splitlines()
    i0 = next(i for i, l in enumerate(lines) if first_marker in l)
    # the `with torch.no_grad():` that opens the body sits right above the teacher's first forward
    while not lines[i0 - 1].strip().startswith("with torch.no_grad()"):
        i0 -= 1
    i0 -= 1
    i1 = next(i for i, l in enumerate(lines) if i > i0 and l.strip().replace(" ", "") == last_marker)
    return textwrap.dedent("\n".join(lines[i0:i1 + 1]))


def gen_trainer_loop(mods):
    """G19: two chained iterations of the loop body of train_arco_2d.py (:283-435) and of train_arco_3d.py (:259-400), run
    from the REFERENCE's text over the REFERENCE's modules on CPU - ISD / FeatureExtractor / compute_contra_memobank_loss /
    RandTPS / DiceLoss, torch nn.Conv2d q_representation, torch.optim.SGD - on fixture inputs.  Stubs: the PIL / scipy
    augmentations (batch_transform, randomGeneratorWithLogits: identity - pinned separately by g10 / g12 / g16), tensorboard
    and logging.  Recorded per iteration: every loss term, bank lengths / pointers / checksums; at the end checksums of
    student / teacher / head weights.  tests/test_dropin_user_gpu.py drives the drop-in modules through the same sequence
    in this repository's own form and compares at 1e-3."""
    import importlib
    out = {}
    rand_tps = importlib.import_module("tps.rand_tps")
    losses_mod = importlib.import_module("utils.losses")
    M2, L2 = mods["model_2D"], mods["loss_helper_3d"]
    path2 = os.path.join(ref_shim.REF, "train_arco_2d.py")
    ns, _ = _pull_functions(path2, {"compute_unsupervised_loss", "label_onehot", "get_revisiting_loss", "_dequeue_and_enqueue"})
    aug, _ = _pull_functions(os.path.join(ref_shim.REF, "augment.py"), {"generate_cutout_mask", "generate_class_mask", "generate_unsup_data"})
    body = _loop_block(path2, "ema_model(train_u_data)", "iter_num+=1")
    for tag, (k2, mix) in {"a": (1.0, "cutmix"), "b": (0.0, "cutout")}.items():
        C, b, patch, Q, Nn, qs, K = 4, 2, (64, 64), 64, 32, 300, 6
        args = types.SimpleNamespace(patch_size=list(patch), apply_aug=mix, k1=1.0, k2=k2, k3=1.0, k4=0.5, topk=3, K=K,
                                     strong_threshold=0.97, strong_threshold_u2pl=0.97, weak_threshold=0.7, func="smc",
                                     num_queries=Q, num_negatives=Nn, num_classes=C)
        seed_all(5)
        isd = M2.ISD(K=36, m=0.99, Ts=0.01, Tt=0.1, num_classes=C, latent_pooling_size=1, latent_feature_size=512,
                     output_pooling_size=8, train_encoder=True, train_decoder=True)
        sd = fx.unet_state(21, 1, C)
        isd.model.load_state_dict(sd); isd.ema_model.load_state_dict(sd)
        model, ema_model = isd.model, isd.ema_model
        for m_ in (model, ema_model):
            zero_dropout(m_)
        q_representation = nn.Sequential(nn.Conv2d(496, 496, kernel_size=1, bias=False), nn.Conv2d(496, 496, kernel_size=1, bias=False))
        kfe = M2.FeatureExtractor(fea_dim=[256, 128, 64, 32, 16], output_dim=496)
        qfe = M2.FeatureExtractor(fea_dim=[256, 128, 64, 32, 16], output_dim=496)
        qfe.load_state_dict(fx.fe_state(31))
        with torch.no_grad():
            q_representation[0].weight.copy_(fx.fe_state(32)["fea4.weight"]); q_representation[1].weight.copy_(fx.fe_state(33)["fea4.weight"])
            for t_p, s_p in zip(kfe.parameters(), qfe.parameters()):
                t_p.data.copy_(s_p.data); t_p.requires_grad = False
        optimizer = torch.optim.SGD(list(model.parameters()) + list(q_representation.parameters()) + list(qfe.parameters()),
                                    lr=0.01, weight_decay=0.0001, momentum=0.9, nesterov=True)
        seed_all(6)
        tps = rand_tps.RandTPS(patch[0], patch[1], batch_size=2 * b, sigma=0.01, border_padding=False, random_mirror=True,
                               random_scale=(0.8, 1.2), mode='affine')
        for m_ in (model, ema_model, q_representation, kfe, qfe):
            m_.train()
        memobank, queue_ptrlis, queue_size = [], [], []
        for i in range(C):
            memobank.append([torch.zeros(1, 496)]); queue_size.append(qs); queue_ptrlis.append(torch.zeros(1, dtype=torch.long))
        rs = np.random.RandomState(3)
        pool = torch.nn.functional.normalize(torch.from_numpy(rs.standard_normal((K, 496 * patch[0] * patch[1])).astype(np.float32)), dim=1)
        ns["args"] = args
        env = dict(ns)
        env.update(aug)
        env.update(dict(torch=torch, np=np, F=F, nn=nn, args=args, model=model, ema_model=ema_model, isd=isd, q_representation=q_representation,
                        k_feature_extractor=kfe, q_feature_extractor=qfe, optimizer=optimizer, tps=tps, memobank=memobank,
                        queue_ptrlis=queue_ptrlis, queue_size=queue_size, random_pool=pool, random_pool_ptr=torch.zeros(1, dtype=torch.long),
                        compute_contra_memobank_loss=L2.compute_contra_memobank_loss, ce_loss=torch.nn.CrossEntropyLoss(),
                        dice_loss=losses_mod.DiceLoss(C), base_lr=0.01, max_iterations=30000, iter_num=0, epoch_num=0, max_epoch=100,
                        record=[], batch_transform=lambda data, label, logits=None, **kw: (data, label, logits),
                        randomGeneratorWithLogits=lambda d, l, g: (d, l, g)))
        out[f"{tag}_pool0"] = np.array(float(pool.double().abs().sum()))
        for it in range(2):
            env["train_l_data"] = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32))
            env["train_u_data"] = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32))
            env["train_l_label"] = torch.from_numpy(fx.blob_labels(rs, b, patch, C))
            seed_all(10 + it)
            exec(body, env)
            for k in ("loss_ce", "loss_dice", "unsup_loss", "reco_loss", "loss_eqv", "loss_q", "loss"):
                out[f"{tag}_{it}_{k}"] = np.array(float(env[k]))
            out[f"{tag}_{it}_bank_len"] = np.array([int(m[0].shape[0]) for m in memobank])
            out[f"{tag}_{it}_ptr"] = np.array([int(p) for p in queue_ptrlis])
            out[f"{tag}_{it}_bank_sum"] = np.array([float(m[0].double().abs().sum()) for m in memobank])
            out[f"{tag}_{it}_pool_ptr"] = np.array(int(env["random_pool_ptr"]))
            out[f"{tag}_{it}_probe"] = np.array((rng_probe(), float(np.random.uniform()), random.random()), dtype=np.float64)
        sdm, sde = model.state_dict(), ema_model.state_dict()
        for k, v in (("w_first", sdm["encoder.in_conv.conv_conv.0.weight"]), ("w_last", sdm["decoder.out_conv.weight"]),
                     ("w_deep", sdm["encoder.down4.maxpool_conv.1.conv_conv.4.weight"]), ("qrep0", q_representation[0].weight),
                     ("qrep1", q_representation[1].weight), ("qfe4", qfe.fea4.weight), ("kfe4", kfe.fea4.weight),
                     ("t_first", sde["encoder.in_conv.conv_conv.0.weight"]), ("rm", sdm["encoder.in_conv.conv_conv.1.running_mean"]),
                     ("pool", env["random_pool"])):
            out[f"{tag}_end_{k}"] = np.array(float(v.detach().double().abs().sum()))
        out[f"{tag}_cfg"] = np.array([C, b, patch[0], patch[1], Q, Nn, qs, K, k2], dtype=np.float64)
    # ---- the volume trainer: train_arco_3d.py's loop body (:259-400) the same way.  C = 4 (the trainer's own default) so that the
    # 5-D banks fill; three iterations: iteration 0 optimises `unsup + supervised + loss_eqv` (:393), later ones the contrastive
    # objective (:391).  batch_transform / transform are the reference's own (identities in augment_3d.py:133-226).
    rand_tps3 = importlib.import_module("tps.rand_tps_3d")
    M3, L3 = mods["model_3D"], mods["loss_helper"]
    path3 = os.path.join(ref_shim.REF, "train_arco_3d.py")
    ns3, _ = _pull_functions(path3, {"compute_unsupervised_loss", "label_onehot", "get_revisiting_loss", "_dequeue_and_enqueue"})
    aug3, _ = _pull_functions(os.path.join(ref_shim.REF, "augment_3d.py"),
                              {"generate_cutout_mask_3d", "generate_class_mask", "generate_unsup_data_3d", "transform", "batch_transform"})
    body3 = _loop_block(path3, "ema_model(train_u_data)", "iter_num+=1")
    tag = "v"
    C, b, patch, Q, Nn, qs, K = 4, 2, (32, 32, 32), 48, 16, 200, 4
    args = types.SimpleNamespace(patch_size=list(patch), apply_aug="cutmix", k1=1.0, k2=1.0, k3=1.0, k4=0.5, topk=2, K=K,
                                 strong_threshold=0.97, strong_threshold_u2pl=0.97, weak_threshold=0.7, func="asmc",
                                 num_queries=Q, num_negatives=Nn, num_classes=C)
    seed_all(5)
    isd = M3.ISD_3d(K=K, m=0.99, Ts=0.01, Tt=0.1, num_classes=C, latent_pooling_size=1, latent_feature_size=128, output_pooling_size=4,
                    train_encoder=True, train_decoder=True)
    sd = fx.vnet_state(52, 1, C)
    isd.model.load_state_dict(sd); isd.ema_model.load_state_dict(sd)
    model, ema_model = isd.model, isd.ema_model
    for m_ in (model, ema_model):
        zero_dropout(m_)
        m_.has_dropout = False
    q_representation = nn.Sequential(nn.Conv3d(16, 16, kernel_size=1, bias=False), nn.Conv3d(16, 16, kernel_size=1, bias=False))
    rsq = np.random.RandomState(62)
    with torch.no_grad():
        for layer in q_representation:
            layer.weight.copy_(torch.from_numpy((rsq.standard_normal((16, 16, 1, 1, 1)) / 4).astype(np.float32)))
    kfe = M3.FeatureExtractor_3d(fea_dim=[128, 64, 32, 16, 16], output_dim=16)
    qfe = M3.FeatureExtractor_3d(fea_dim=[128, 64, 32, 16, 16], output_dim=16)
    qfe.load_state_dict(fx.fe_state(61, (128, 64, 32, 16, 16), 16, nd=3))
    with torch.no_grad():
        for t_p, s_p in zip(kfe.parameters(), qfe.parameters()):
            t_p.data.copy_(s_p.data); t_p.requires_grad = False
    optimizer = torch.optim.SGD(list(model.parameters()) + list(q_representation.parameters()) + list(qfe.parameters()),
                                lr=0.01, weight_decay=0.0001, momentum=0.9, nesterov=True)
    seed_all(6)
    tps = rand_tps3.RandTPS(patch[0], patch[1], patch[2], batch_size=2 * b, sigma=0.01, border_padding=False, random_mirror=True,
                            random_scale=(0.8, 1.2), mode='affine')
    for m_ in (model, ema_model, q_representation, kfe, qfe):
        m_.train()
    seed_all(7)
    memobank, queue_ptrlis, queue_size = [], [], []
    for i in range(C):
        memobank.append([torch.randn(1, 16)]); queue_size.append(qs); queue_ptrlis.append(torch.zeros(1, dtype=torch.long))      # :144-151
    out[f"{tag}_bank0"] = np.stack([m[0].numpy()[0] for m in memobank])
    rs = np.random.RandomState(13)
    pool = torch.nn.functional.normalize(torch.from_numpy(rs.standard_normal((K, 16 * patch[0] * patch[1] * patch[2])).astype(np.float32)), dim=1)
    ns3["args"] = args
    env = dict(ns3)
    env.update(aug3)
    env.update(dict(torch=torch, np=np, F=F, nn=nn, args=args, model=model, ema_model=ema_model, isd=isd, q_representation=q_representation,
                    k_feature_extractor=kfe, q_feature_extractor=qfe, optimizer=optimizer, tps=tps, memobank=memobank,
                    queue_ptrlis=queue_ptrlis, queue_size=queue_size, random_pool=pool, random_pool_ptr=torch.zeros(1, dtype=torch.long),
                    compute_contra_memobank_loss=L3.compute_contra_memobank_loss, ce_loss=torch.nn.CrossEntropyLoss(),
                    dice_loss=losses_mod.DiceLoss(C), base_lr=0.01, max_iterations=30000, iter_num=0, epoch_num=0, max_epoch=100, record=[]))
    for it in range(3):
        env["train_l_data"] = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32))
        env["train_u_data"] = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32))
        env["train_l_label"] = torch.from_numpy(fx.blob_labels(rs, b, patch, C))
        seed_all(10 + it)
        exec(body3, env)
        for k in ("loss_ce", "loss_dice", "unsup_loss", "reco_loss", "loss_eqv", "loss_q", "loss"):
            out[f"{tag}_{it}_{k}"] = np.array(float(env[k]))
        out[f"{tag}_{it}_bank_len"] = np.array([int(m[0].shape[0]) for m in memobank])
        out[f"{tag}_{it}_ptr"] = np.array([int(p) for p in queue_ptrlis])
        out[f"{tag}_{it}_bank_sum"] = np.array([float(m[0].double().abs().sum()) for m in memobank])
        out[f"{tag}_{it}_probe"] = np.array((rng_probe(), float(np.random.uniform()), random.random()), dtype=np.float64)
    sdm = model.state_dict()
    for k, v in (("w_first", sdm["block_one.conv.0.weight"]), ("w_out", sdm["out_conv.weight"]), ("qrep0", q_representation[0].weight),
                 ("qfe4", qfe.fea4.weight), ("kfe4", kfe.fea4.weight), ("t_first", ema_model.state_dict()["block_one.conv.0.weight"]),
                 ("rm", sdm["block_one.conv.1.running_mean"])):
        out[f"{tag}_end_{k}"] = np.array(float(v.detach().double().abs().sum()))
    out[f"{tag}_cfg"] = np.array([C, b, *patch, Q, Nn, qs, K], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "g19_trainer_loop.npz"), **out)
    print("g19_trainer_loop", len(out))


if __name__ == "__main__":
    mods = ref_shim.load()
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19"]
    if "g1" in which: gen_samplers(mods)
    if "g2" in which: gen_loss(mods)
    if "g3" in which: gen_nets(mods)
    if "g4" in which: gen_glue()
    if "g5" in which: gen_eqv()
    if "g6" in which: gen_eval3d(mods)
    if "g7" in which: gen_mix()
    if "g8" in which: gen_ingest()
    if "g9" in which: gen_flags()
    if "g10" in which: gen_morph()
    if "g11" in which: gen_eval2d(mods)
    if "g12" in which: gen_jitter()
    if "g13" in which: gen_pretrain(mods)
    if "g14" in which: gen_pretrain(mods, three_d=True)
    if "g15" in which: gen_unet_kinkfree(mods)
    if "g16" in which: gen_boundary(mods)
    if "g17" in which: gen_vnet_norms(mods)
    if "g18" in which: gen_vnet_strict(mods)
    if "g19" in which: gen_trainer_loop(mods)
