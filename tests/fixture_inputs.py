"""Deterministic synthetic inputs shared by oracle/gen_golden.py and the tests.

Everything is drawn from numpy RandomState (bit-stable across numpy versions), so
the golden files only need to hold the reference's OUTPUTS.
"""
import numpy as np
import torch


def blob_labels(rs, b, spatial, n_cls, absent=()):
    """Piecewise-constant label maps: background 0 plus random boxes of the other
    classes (ACDC-like imbalance).  Classes in `absent` never appear."""
    lab = np.zeros((b, *spatial), dtype=np.int64)
    present = [c for c in range(1, n_cls) if c not in absent]
    for i in range(b):
        for c in present:
            lo = [rs.randint(0, max(1, s - 2)) for s in spatial]
            sz = [rs.randint(max(1, s // 6), max(2, s // 2)) for s in spatial]
            sl = tuple(slice(l, min(s, l + z)) for l, z, s in zip(lo, sz, spatial))
            lab[(i, *sl)] = c
    if 0 in absent:
        # push background away: fill with the first present class
        lab[lab == 0] = present[0]
    return lab


def onehot(lab, n_cls):
    t = torch.from_numpy(lab)
    out = torch.zeros(lab.shape[0], n_cls, *lab.shape[1:], dtype=torch.int64)
    return out.scatter_(1, t.clamp_min(0).unsqueeze(1), 1)


def loss_inputs(seed, b=2, n_cls=4, feat=16, spatial=(16, 16), absent=(), low_p=0.35, high_p=0.5,
                single_class=False, tie_probs=False, label_bonus=0.8):
    """Inputs of compute_contra_memobank_loss for one step (b labeled + b unlabeled)."""
    rs = np.random.RandomState(seed)
    B = 2 * b
    rep = rs.standard_normal((B, feat, *spatial)).astype(np.float32)
    rep_t = rs.standard_normal((B, feat, *spatial)).astype(np.float32)
    if single_class:
        lab_l = np.zeros((b, *spatial), dtype=np.int64)
        lab_u = np.zeros((b, *spatial), dtype=np.int64)
    else:
        lab_l = blob_labels(rs, b, spatial, n_cls, absent)
        lab_u = blob_labels(rs, b, spatial, n_cls, absent)
    lab = np.concatenate([lab_l, lab_u])
    logit = rs.standard_normal((B, n_cls, *spatial)).astype(np.float32)
    logit += label_bonus * np.moveaxis(np.eye(n_cls, dtype=np.float32)[lab], -1, 1)
    if tie_probs:
        logit = np.round(logit)          # many exact ties between classes
    prob = torch.softmax(torch.from_numpy(logit), 1)
    low_u = (rs.uniform(size=(b, 1, *spatial)) < low_p).astype(np.float32)
    high_u = (rs.uniform(size=(b, 1, *spatial)) < high_p).astype(np.float32)
    ones = np.ones((b, 1, *spatial), dtype=np.float32)
    return dict(
        rep=torch.from_numpy(rep), rep_teacher=torch.from_numpy(rep_t),
        label_l=onehot(lab_l, n_cls), label_u=onehot(lab_u, n_cls),
        prob_l=prob[:b].contiguous(), prob_u=prob[b:].contiguous(),
        low_mask=torch.from_numpy(np.concatenate([ones, low_u])),
        high_mask=torch.from_numpy(np.concatenate([ones, high_u])),
    )


def fresh_bank(n_cls, feat, queue_size, init='zeros', seed=0):
    """train_arco_2d.py:147-154 (zeros(1,D)) / train_arco_3d.py:144-151 (randn(1,D))."""
    rs = np.random.RandomState(1000 + seed)
    bank, ptr, qs = [], [], []
    for c in range(n_cls):
        row = np.zeros((1, feat), np.float32) if init == 'zeros' else rs.standard_normal((1, feat)).astype(np.float32)
        bank.append([torch.from_numpy(row)])
        ptr.append(torch.zeros(1, dtype=torch.long))
        qs.append(queue_size[c] if isinstance(queue_size, (list, tuple)) else queue_size)
    return bank, ptr, qs


# loss cases: name -> (kwargs for loss_inputs, loss kwargs, queue_size, bank init)
LOSS_CASES = {
    "d16_smc": (dict(b=2, n_cls=4, feat=16, spatial=(16, 16)),
                dict(func='smc', num_queries=32, num_negatives=16, delta_n=0.97), 96, 'zeros'),
    "d16_asmc": (dict(b=2, n_cls=4, feat=16, spatial=(16, 16)),
                 dict(func='asmc', num_queries=32, num_negatives=16, delta_n=0.97), 96, 'zeros'),
    "d16_randint": (dict(b=2, n_cls=4, feat=16, spatial=(16, 16)),
                    dict(func='rand', num_queries=32, num_negatives=16, delta_n=0.97), 96, 'zeros'),
    "d64_smc_default_q": (dict(b=1, n_cls=4, feat=64, spatial=(24, 24)),
                          dict(func='smc', delta_n=0.97), 512, 'zeros'),
    "absent_class1": (dict(b=2, n_cls=4, feat=16, spatial=(16, 16), absent=(1,)),
                      dict(func='smc', num_queries=32, num_negatives=16, delta_n=0.97), 96, 'zeros'),
    "absent_class0": (dict(b=2, n_cls=4, feat=16, spatial=(16, 16), absent=(0,)),
                      dict(func='asmc', num_queries=32, num_negatives=16, delta_n=0.97), 96, 'zeros'),
    "single_class": (dict(b=2, n_cls=4, feat=16, spatial=(8, 8), single_class=True),
                     dict(func='smc', num_queries=32, num_negatives=16, delta_n=0.97), 96, 'zeros'),
    "binary_3d": (dict(b=1, n_cls=2, feat=16, spatial=(8, 8, 6)),
                  dict(func='asmc', num_queries=32, num_negatives=16, delta_n=0.97), 64, 'randn'),
    "c8_ties": (dict(b=1, n_cls=8, feat=16, spatial=(16, 16), tie_probs=True),
                 dict(func='smc', num_queries=32, num_negatives=16, delta_n=0.97), 128, 'zeros'),
    "tiny_queue_overflow": (dict(b=2, n_cls=4, feat=16, spatial=(16, 16)),
                            dict(func='smc', num_queries=32, num_negatives=16, delta_n=1.0), 20, 'zeros'),
    "proto_momentum": (dict(b=2, n_cls=4, feat=16, spatial=(16, 16)),
                       dict(func='smc', num_queries=32, num_negatives=16, delta_n=0.97), 96, 'zeros'),
    # 5-D cases whose banks FILL (loss_helper.py:442-686 with C >= 4, the 3-D trainer's default --num_classes 4,
    # train_arco_3d.py:44,144-151): with C = 2 or 3 the rank window [3, 20) is empty and no key is ever enqueued.
    # Step 0 meets the one-row randn bank (1-D fallback sampler), step 1 the grid negative sampler on a grown bank,
    # step 2 the FIFO truncation.
    "c4_3d": (dict(b=1, n_cls=4, feat=16, spatial=(12, 12, 8), label_bonus=0.2),
              dict(func='asmc', num_queries=32, num_negatives=16, delta_n=0.97), [100, 40, 40, 40], 'randn'),
    "c4_3d_absent": (dict(b=2, n_cls=4, feat=16, spatial=(10, 12, 8), absent=(2,)),
                     dict(func='smc', num_queries=32, num_negatives=16, delta_n=0.97), [200, 300, 300, 300], 'randn'),
    "c5_3d_default_q": (dict(b=1, n_cls=5, feat=16, spatial=(12, 10, 12), label_bonus=0.0),
                        dict(func='asmc', delta_n=0.97), [260, 30000, 150, 30000, 30000], 'randn'),
}
LOSS_STEPS = 3

SAMPLER_HIGHS = [1, 3, 11, 12, 13, 15, 16, 17, 100, 257, 4096, 5233, 8179, 30000, 300000]
SAMPLER_SHAPES = [256, 8192]
SAMPLER_SEEDS = [0, 7]


def unet_state(seed, in_chns=1, n_cls=4, ft=(16, 32, 64, 128, 256)):
    """Random U-Net parameters keyed like the reference state_dict
    (networks/unetWithArgs.py; 136 keys incl. BN buffers)."""
    rs = np.random.RandomState(seed)
    sd = {}

    def conv(name, co, ci, k):
        fan = ci * k * k
        sd[name + ".weight"] = torch.from_numpy((rs.standard_normal((co, ci, k, k)) / np.sqrt(fan)).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy((0.1 * rs.standard_normal(co)).astype(np.float32))

    def bn(name, c):
        sd[name + ".weight"] = torch.from_numpy((1 + 0.1 * rs.standard_normal(c)).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy((0.1 * rs.standard_normal(c)).astype(np.float32))
        sd[name + ".running_mean"] = torch.zeros(c)
        sd[name + ".running_var"] = torch.ones(c)
        sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def block(pre, ci, co):
        conv(pre + ".conv_conv.0", co, ci, 3)
        bn(pre + ".conv_conv.1", co)
        conv(pre + ".conv_conv.4", co, co, 3)
        bn(pre + ".conv_conv.5", co)

    block("encoder.in_conv", in_chns, ft[0])
    for i in range(1, 5):
        block(f"encoder.down{i}.maxpool_conv.1", ft[i - 1], ft[i])
    for i, (c1, c2) in zip(range(1, 5), ((ft[4], ft[3]), (ft[3], ft[2]), (ft[2], ft[1]), (ft[1], ft[0]))):
        # Decoder builds UpBlock with its default bilinear=True (unetWithArgs.py:66,130-137):
        # 1x1 conv (with bias) then x2 bilinear upsample, align_corners=True (:72-75)
        conv(f"decoder.up{i}.conv1x1", c2, c1, 1)
        block(f"decoder.up{i}.conv", 2 * c2, c2)
    conv("decoder.out_conv", n_cls, ft[0], 3)
    return sd


def fe_state(seed, fea_dim=(256, 128, 64, 32, 16), out_dim=496, nd=2):
    rs = np.random.RandomState(seed)
    sd, cnt = {}, 0
    for i, f in enumerate(fea_dim):
        cnt += f
        co = out_dim if i == 4 else cnt
        w = (rs.standard_normal((co, cnt) + (1,) * nd) / np.sqrt(cnt)).astype(np.float32)
        sd[f"fea{i}.weight"] = torch.from_numpy(w)
    return sd


def vnet_state(seed, in_chns=1, n_cls=2, nf=16):
    """Random V-Net parameters keyed like networks/vnetWithArgs.py (batchnorm variant)."""
    rs = np.random.RandomState(seed)
    sd = {}

    def conv(name, co, ci, k, transpose=False):
        shape = (ci, co, k, k, k) if transpose else (co, ci, k, k, k)
        sd[name + ".weight"] = torch.from_numpy((rs.standard_normal(shape) / np.sqrt(ci * k ** 3)).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy((0.1 * rs.standard_normal(co)).astype(np.float32))

    def bn(name, c):
        sd[name + ".weight"] = torch.from_numpy((1 + 0.1 * rs.standard_normal(c)).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy((0.1 * rs.standard_normal(c)).astype(np.float32))
        sd[name + ".running_mean"] = torch.zeros(c)
        sd[name + ".running_var"] = torch.ones(c)
        sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def stage(pre, n, ci, co):
        for s in range(n):
            conv(f"{pre}.conv.{3 * s}", co, ci if s == 0 else co, 3)
            bn(f"{pre}.conv.{3 * s + 1}", co)

    def down(pre, ci, co):
        conv(f"{pre}.conv.0", co, ci, 2)
        bn(f"{pre}.conv.1", co)

    def up(pre, ci, co):
        conv(f"{pre}.conv.0", co, ci, 2, transpose=True)
        bn(f"{pre}.conv.1", co)

    stage("block_one", 1, in_chns, nf); down("block_one_dw", nf, 2 * nf)
    stage("block_two", 2, 2 * nf, 2 * nf); down("block_two_dw", 2 * nf, 4 * nf)
    stage("block_three", 3, 4 * nf, 4 * nf); down("block_three_dw", 4 * nf, 8 * nf)
    stage("block_four", 3, 8 * nf, 8 * nf); down("block_four_dw", 8 * nf, 16 * nf)
    stage("block_five", 3, 16 * nf, 16 * nf); up("block_five_up", 16 * nf, 8 * nf)
    stage("block_six", 3, 8 * nf, 8 * nf); up("block_six_up", 8 * nf, 4 * nf)
    stage("block_seven", 3, 4 * nf, 4 * nf); up("block_seven_up", 4 * nf, 2 * nf)
    stage("block_eight", 2, 2 * nf, 2 * nf); up("block_eight_up", 2 * nf, nf)
    stage("block_nine", 1, nf, nf)
    conv("out_conv", n_cls, nf, 1)
    return sd


def image_batch(seed, b, c, spatial):
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.uniform(size=(b, c, *spatial)).astype(np.float32))


def randomize_running_stats(sd, seed):
    """Non-trivial BatchNorm running statistics (eval-mode cases), drawn in key order."""
    rs = np.random.RandomState(seed)
    for k in sorted(sd):
        if k.endswith("running_mean"):
            sd[k] = torch.from_numpy((0.2 * rs.standard_normal(sd[k].shape)).astype(np.float32))
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy(rs.uniform(0.5, 1.5, size=sd[k].shape).astype(np.float32))
    return sd


# 3-D sliding-window evaluation cases (g6): tag -> (volume shape, patch, stride_xy, stride_z, classes, nf, seed)
EVAL3D_CASES = {
    "ragged": ((20, 22, 18), (16, 16, 16), 6, 4, 2, 4, 31),       # clamped last windows on every axis
    "padded": ((12, 16, 10), (16, 16, 16), 8, 8, 3, 4, 32),       # volume smaller than the patch: symmetric zero pad
    "exact": ((32, 16, 16), (16, 16, 16), 16, 16, 2, 4, 33),      # non-overlapping windows
}


def eval3d_volume(seed, shape):
    rs = np.random.RandomState(seed)
    return rs.uniform(size=shape).astype(np.float32)


# mixing-strategy cases (g7): tag -> (mode, batch, channels, spatial, classes, seed)
MIX_CASES = {
    "cutmix2d": ("cutmix", 4, 1, (24, 32), 4, 41),
    "cutout2d": ("cutout", 3, 3, (32, 32), 4, 42),
    "classmix2d": ("classmix", 4, 1, (24, 24), 6, 43),
    "none2d": ("none", 2, 1, (8, 8), 3, 44),
    "cutmix3d": ("cutmix", 2, 1, (16, 16, 24), 2, 45),
    "cutout3d": ("cutout", 2, 1, (12, 16, 22), 3, 46),
    "classmix3d": ("classmix", 3, 1, (8, 8, 21), 5, 47),
}


def mix_inputs(seed, b, c, spatial, n_cls):
    rs = np.random.RandomState(seed)
    data = torch.from_numpy(rs.uniform(size=(b, c, *spatial)).astype(np.float32))
    target = torch.from_numpy(rs.randint(0, n_cls, size=(b, *spatial)).astype(np.int64))
    logits = torch.from_numpy(rs.uniform(size=(b, *spatial)).astype(np.float32))
    return data, target, logits


# data-ingest cases (g8): slices / volumes the reference's transforms are run on
INGEST_SEEDS = [0, 1, 2, 3, 4, 5, 6, 7]


def ingest_slice(seed, shape=(50, 44), n_cls=4):
    rs = np.random.RandomState(1000 + seed)
    return rs.uniform(size=shape).astype(np.float32), blob_labels(rs, 1, shape, n_cls)[0].astype(np.uint8)


def ingest_volume(seed, shape=(20, 18, 14)):
    rs = np.random.RandomState(2000 + seed)
    return rs.uniform(size=shape).astype(np.float32), (rs.uniform(size=shape) > 0.7).astype(np.uint8)


# public names whose call signatures are pinned to the reference (g9): reference module -> names
PUBLIC_NAMES = {
    "loss_helper_3d": ["compute_contra_memobank_loss", "grid_monte_carlo_sample", "grid_as_monte_carlo_sample",
                       "monte_carlo_sample", "as_monte_carlo_sample", "dequeue_and_enqueue"],
    "loss_helper": ["compute_contra_memobank_loss", "grid_monte_carlo_sample", "grid_as_monte_carlo_sample",
                    "monte_carlo_sample", "as_monte_carlo_sample", "dequeue_and_enqueue"],
    "networks.unetWithArgs": ["UNet", "ConvBlock", "DownBlock", "UpBlock", "Encoder", "Decoder"],
    "networks.vnetWithArgs": ["VNet", "ConvBlock", "DownsamplingConvBlock", "UpsamplingDeconvBlock"],
    "model_2D": ["FeatureExtractor", "ISD", "create_model"],
    "model_3D": ["FeatureExtractor_3d", "ISD_3d", "create_model_3d"],
    "tps.rand_tps": ["RandTPS"],
    "tps.rand_tps_3d": ["RandTPS"],
}


# modules whose state_dict keys / shapes are pinned to the reference (g9): (module, class, kwargs)
STATE_CASES = [
    ("model_2D", "ISD", dict(K=36, m=0.99, Ts=0.01, Tt=0.1, num_classes=4, latent_pooling_size=1, latent_feature_size=512,
                             output_pooling_size=8, train_encoder=True, train_decoder=True)),
    ("model_2D", "FeatureExtractor", dict(fea_dim=[256, 128, 64, 32, 16], output_dim=496)),
    ("model_3D", "ISD_3d", dict(K=36, m=0.99, Ts=0.01, Tt=0.1, num_classes=2, latent_pooling_size=1, latent_feature_size=512,
                                output_pooling_size=8, train_encoder=True, train_decoder=True)),
    ("model_3D", "FeatureExtractor_3d", dict(fea_dim=[128, 64, 32, 16, 16], output_dim=16)),
    ("networks.vnetWithArgs", "VNet", dict(n_channels=1, n_classes=2, normalization='batchnorm', has_dropout=True)),
    ("networks.unetWithArgs", "UNet", dict(in_chns=1, class_num=4)),
]


# AdvMorph cases: (tag, B, C, H, W, seed); the velocity field has the trainer's size [B, 2, W//8, W//8] (augment.py:274)
MORPH_CASES = [("a", 2, 1, 64, 64, 3), ("b", 1, 3, 48, 80, 4), ("c", 3, 1, 256, 256, 5)]


MORPH_STRIDE = 5


def morph_inputs(seed, B, C, H, W):
    rs = np.random.RandomState(9000 + seed)
    data = torch.from_numpy(rs.uniform(size=(B, C, H, W)).astype(np.float32))
    param = torch.from_numpy((rs.uniform(size=(B, 2, W // 8, W // 8)) * 2 - 1).astype(np.float32))
    return data, param


# 2-D evaluation cases: tag -> (volume shape [slices, x, y], classes, seed); slices are zoomed to 256 x 256 by the function
EVAL2D_CASES = {"a": ((3, 40, 36), 4, 61), "b": ((2, 300, 280), 4, 62)}


def eval2d_volume(seed, shape, n_cls):
    rs = np.random.RandomState(7000 + seed)
    image = rs.uniform(size=shape).astype(np.float32)
    label = blob_labels(rs, shape[0], shape[1:], n_cls).astype(np.uint8)
    image += 0.5 * (label > 0)                   # some structure for the network to follow
    return image, label


# ColorJitter / GaussianBlur cases (g12): tag -> (C, H, W, seed, order, (brightness, contrast, saturation, hue), blur sigma or None)
JITTER_CASES = {
    "l_bc": (1, 40, 52, 1, (0, 1, 2, 3), (1.2, 0.8, 1.1, 0.1), None),
    "l_cb_blur": (1, 64, 48, 2, (1, 3, 0, 2), (0.76, 1.24, 0.9, -0.2), 0.9),
    "l_blur_only": (1, 33, 70, 3, None, None, 0.15),
    "l_blur_big": (1, 50, 50, 4, (2, 0, 3, 1), (1.25, 0.75, 1.0, 0.0), 1.15),
    "rgb_all": (3, 36, 44, 5, (3, 1, 2, 0), (0.9, 1.2, 0.8, -0.25), None),
    "rgb_hue_sat_blur": (3, 48, 40, 6, (2, 3, 1, 0), (1.1, 0.85, 1.25, 0.25), 0.6),
    "rgb_blur_r1": (3, 40, 40, 7, None, None, 2.2),           # integer box radius 1 (outside the trainer's sigma range)
}


def jitter_image(seed, C, H, W):
    """float image in [0, 1] with some exact grey / saturated pixels (HSV corner cases)."""
    rs = np.random.RandomState(8000 + seed)
    x = rs.uniform(size=(C, H, W)).astype(np.float32)
    x[:, 0, :6] = np.array([[0.0, 1.0, 0.5, 0.25, 1.0, 0.0]], dtype=np.float32)
    if C == 3:
        x[:, 1, 0] = (1.0, 0.0, 0.0); x[:, 1, 1] = (0.0, 1.0, 0.0); x[:, 1, 2] = (0.0, 0.0, 1.0); x[:, 1, 3] = (0.2, 0.2, 0.2)
    return x


# ---- stage-1 pre-training (oracle/gen_golden.py g13, tests/test_pretrain_*.py): a small ISD configuration
STAGE1_CFG = dict(K=12, Ts=0.1, Tt=0.1, num_classes=4, latent_pooling_size=1, latent_feature_size=32, output_pooling_size=4,
                  patch_size=16, b=4, labeled_bs=2, size=(64, 64), lr=0.01, steps=2)


def isd_head_state(seed, cfg=STAGE1_CFG):
    """Seeded parameters of the ISD heads and the two queues, keyed like ISD.state_dict() minus the two U-Nets."""
    rs = np.random.RandomState(seed)
    C, Fd, P = cfg["num_classes"], cfg["latent_feature_size"], cfg["output_pooling_size"]

    def t(*shape, scale=0.2):
        return torch.from_numpy((scale * rs.standard_normal(shape)).astype(np.float32))

    sd = {}
    for h in ("k_latent_head", "q_latent_head"):
        sd[h + ".f1.weight"], sd[h + ".f1.bias"] = t(256, 256 * cfg["latent_pooling_size"] ** 2, scale=0.05), t(256)
        sd[h + ".f2.weight"], sd[h + ".f2.bias"] = t(Fd, 256, scale=0.05), t(Fd)
    for i in (0, 1):
        sd[f"latent_predictor.{i}.weight"], sd[f"latent_predictor.{i}.bias"] = t(Fd, Fd), t(Fd)
    for h in ("k_outputs_head", "q_outputs_head"):
        sd[h + ".proj.1.weight"], sd[h + ".proj.1.bias"] = t(2 * C, C, 1, 1, scale=0.5), t(2 * C)
        sd[h + ".proj.2.weight"], sd[h + ".proj.2.bias"] = t(C, 2 * C, 1, 1, scale=0.5), t(C)
    for i in (0, 1):
        sd[f"outputs_predictor.{i}.weight"], sd[f"outputs_predictor.{i}.bias"] = t(C, C, 1, 1, scale=0.5), t(C)
    q = t(cfg["K"], Fd, scale=1.0)
    qm = t(cfg["K"], 49, C * P * P, scale=1.0)
    sd["queue"] = q / q.norm(dim=0, keepdim=True)
    sd["queue_mask"] = qm / qm.norm(dim=0, keepdim=True)
    sd["queue_ptr"] = torch.zeros(1, dtype=torch.long)
    sd["mask_queue_ptr"] = torch.zeros(1, dtype=torch.long)
    return sd


def stage1_batch(seed, it, cfg=STAGE1_CFG):
    """(student images, teacher images, labels) of iteration `it`."""
    rs = np.random.RandomState(seed + 17 * it)
    b, size = cfg["b"], cfg["size"]
    im_q = torch.from_numpy(rs.uniform(size=(b, 1, *size)).astype(np.float32))
    im_k = (im_q + torch.from_numpy((0.05 * rs.standard_normal((b, 1, *size))).astype(np.float32))).clamp(0, 1)
    lab = torch.from_numpy(blob_labels(rs, b, size, cfg["num_classes"]))
    return im_q, im_k, lab


STAGE1_CFG_3D = dict(K=4, Ts=0.1, Tt=0.1, num_classes=2, latent_pooling_size=1, latent_feature_size=16, output_pooling_size=2,
                     patch_size=16, b=2, labeled_bs=1, size=(32, 32, 32), lr=0.01, steps=2, n_patches=27)


def isd3d_head_state(seed, cfg=STAGE1_CFG_3D):
    """Seeded parameters of the ISD_3d heads and queues (queue_mask with the patch count of the small test volume: the
    constructor hard-wires 700 = the 20-voxel patches of a 112 x 112 x 80 volume)."""
    rs = np.random.RandomState(seed)
    C, Fd, P = cfg["num_classes"], cfg["latent_feature_size"], cfg["output_pooling_size"]

    def t(*shape, scale=0.2):
        return torch.from_numpy((scale * rs.standard_normal(shape)).astype(np.float32))

    sd = {}
    for h in ("k_latent_head", "q_latent_head"):
        sd[h + ".f1.weight"], sd[h + ".f1.bias"] = t(128, 128 * cfg["latent_pooling_size"] ** 2, scale=0.05), t(128)
        sd[h + ".f2.weight"], sd[h + ".f2.bias"] = t(Fd, 128, scale=0.05), t(Fd)
    for i in (0, 1):
        sd[f"latent_predictor.{i}.weight"], sd[f"latent_predictor.{i}.bias"] = t(Fd, Fd), t(Fd)
    for h in ("k_outputs_head", "q_outputs_head"):
        sd[h + ".proj.1.weight"], sd[h + ".proj.1.bias"] = t(2 * C, C, 1, 1, 1, scale=0.5), t(2 * C)
        sd[h + ".proj.2.weight"], sd[h + ".proj.2.bias"] = t(C, 2 * C, 1, 1, 1, scale=0.5), t(C)
    for i in (0, 1):
        sd[f"outputs_predictor.{i}.weight"], sd[f"outputs_predictor.{i}.bias"] = t(C, C, 1, 1, 1, scale=0.5), t(C)
    q = t(cfg["K"], Fd, scale=1.0)
    qm = t(cfg["K"], cfg["n_patches"], C * P ** 3, scale=1.0)
    sd["queue"] = q / q.norm(dim=-1, keepdim=True)
    sd["queue_mask"] = qm / qm.norm(dim=-1, keepdim=True)
    sd["queue_ptr"] = torch.zeros(1, dtype=torch.long)
    sd["mask_queue_ptr"] = torch.zeros(1, dtype=torch.long)
    return sd


def stage1_batch_3d(seed, it, cfg=STAGE1_CFG_3D):
    rs = np.random.RandomState(seed + 17 * it)
    b, size = cfg["b"], cfg["size"]
    im_q = torch.from_numpy(rs.uniform(size=(b, 1, *size)).astype(np.float32))
    im_k = (im_q + torch.from_numpy((0.05 * rs.standard_normal((b, 1, *size))).astype(np.float32))).clamp(0, 1)
    lab = torch.from_numpy(blob_labels(rs, b, size, cfg["num_classes"]))
    return im_q, im_k, lab


def fill_state(module, seed):
    """Deterministic parameters for ANY module, by state_dict order and shape (both the reference module and this package's
    twin have the same keys): conv / linear weights ~ N(0, 1/fan_in), norm weights 1 + 0.1 N, biases 0.1 N, running_var in
    [0.5, 1.5], counters untouched.  Returns the state dict it loaded."""
    rs = np.random.RandomState(seed)
    sd = {}
    for k, v in module.state_dict().items():
        shape = tuple(v.shape)
        if not v.is_floating_point():
            sd[k] = v.clone()
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy(rs.uniform(0.5, 1.5, size=shape).astype(np.float32))
        elif v.dim() >= 2:
            fan = 1
            for d in shape[1:]:
                fan *= d
            sd[k] = torch.from_numpy((rs.standard_normal(shape) / np.sqrt(fan)).astype(np.float32))
        elif k.endswith("weight"):
            sd[k] = torch.from_numpy((1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32))
        else:
            sd[k] = torch.from_numpy((0.1 * rs.standard_normal(shape)).astype(np.float32))
    module.load_state_dict(sd, strict=True)
    return sd
