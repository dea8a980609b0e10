"""Mixing strategies of the unlabeled stream on the GPU (first part) - the names of the reference's
code/augment.py: `generate_cutout_mask` (:230-244), `generate_class_mask` (:247-252), `generate_unsup_data` (:284-313),
and of code/augment_3d.py: `generate_cutout_mask_3d` (:182-198), `generate_unsup_data_3d` (:228-257).

The random choices are the reference's, drawn from the same host generators in the same order (numpy's global
RandomState for the boxes, torch's CPU generator for classmix's randperm), so a seeded run mixes the same pixels;
the mixing itself is one launch over the whole batch instead of a Python loop of per-image tensor expressions, and
nothing leaves the GPU (classmix reads back one 64-bit label set per image, where the reference syncs on torch.unique).

`batch_transform` (tensor -> PIL -> ColorJitter / GaussianBlur -> tensor -> AdvMorph, augment.py:133-227,255-281) is at
the end of this file: Pillow's 8-bit integer arithmetic as HIP kernels (csrc/augment.hip), pinned bit for bit against
Pillow (tests/golden/g12_jitter.npz), AdvMorph pinned against the reference class (arco_amd/adv_morph.py, g10)."""
import ctypes

import numpy as np
import torch

from . import _lib as L

_MODES = {"cutmix": 0, "cutout": 1, "classmix": 2}


def _cutout_box(img_size, ratio=2):
    """The zero box of generate_cutout_mask / _3d as (y0, y1, x0, x1[, z0, z1]) - same draws, same order."""
    cutout_area = img_size[0] * img_size[1] / ratio
    w = np.random.randint(img_size[1] / ratio + 1, img_size[1])
    h = np.round(cutout_area / w)
    x_start = np.random.randint(0, img_size[1] - w + 1)
    y_start = np.random.randint(0, img_size[0] - h + 1)
    box = [int(y_start), int(y_start + h), int(x_start), int(x_start + w)]
    if len(img_size) == 3:
        z_start = np.random.randint(0, img_size[2] - 20 + 1)
        box += [int(z_start), int(z_start + 10)]
    return box


def _box_mask(img_size, box):
    mask = torch.ones(list(img_size))
    if len(img_size) == 3:
        mask[box[0]:box[1], box[2]:box[3], box[4]:box[5]] = 0
    else:
        mask[box[0]:box[1], box[2]:box[3]] = 0
    return mask.float()


def generate_cutout_mask(img_size, ratio=2):
    """[H, W] float mask, 0 inside a random rectangle of area H*W/ratio (augment.py:230-244)."""
    return _box_mask(img_size, _cutout_box(img_size, ratio))


def generate_cutout_mask_3d(img_size, ratio=2, dep=80):
    """[H, W, Z] float mask, 0 inside a random box 10 slices deep (augment_3d.py:182-198)."""
    return _box_mask(img_size, _cutout_box(img_size, ratio))


def _label_sets(target):
    """Per image: sorted list of the labels present (torch.unique) - one kernel + one small read-back."""
    B = int(target.shape[0])
    pres = torch.empty(B, dtype=torch.int64, device=target.device)
    L.call("arco_label_presence", L.ptr(target), B, target[0].numel(), L.ptr(pres))
    out = []
    for v in pres.cpu().tolist():
        v &= (1 << 64) - 1
        out.append([c for c in range(64) if (v >> c) & 1])
    return out


def _select_half(labels):
    """labels[torch.randperm(len(labels))][:len(labels) // 2] (augment.py:249) as a bit set."""
    perm = torch.randperm(len(labels)).tolist()
    sel = 0
    for k in perm[:len(labels) // 2]:
        sel |= 1 << labels[k]
    return sel


def generate_class_mask(pseudo_labels):
    """Float mask selecting a random half of the labels present in ONE label map (augment.py:247-252)."""
    L.require_gpu(pseudo_labels)
    t = pseudo_labels.to(torch.int64).contiguous()
    sel = _select_half(_label_sets(t.unsqueeze(0))[0])
    table = torch.tensor([(sel >> c) & 1 for c in range(64)], dtype=torch.float32, device=t.device)
    return table[t.clamp(0, 63)] * (t >= 0)


def draw_boxes(B, spatial):
    """The host draws of the cutout / cutmix strategies (one half-area box per image, augment.py:229-245) as the descriptor table of
    arco_mix_unsup - drawn apart from the mixing so that a trainer can mix the IMAGES (which need the boxes only) before the teacher
    pass that produces the pseudo-labels has finished, and the labels / logits afterwards, with the same boxes (_mix(..., desc=))."""
    desc = np.zeros((B, 8), dtype=np.int32)
    desc[:, 5] = 1
    for i in range(B):
        box = _cutout_box(list(spatial), ratio=2)
        desc[i, :len(box)] = box
    return desc


def mix_images(data, mode, desc):
    """new_data of generate_unsup_data(_3d) alone, for boxes drawn by draw_boxes (cutout / cutmix: the image mix does not read the
    pseudo-labels)."""
    assert mode in ("cutout", "cutmix")
    B = int(data.shape[0])
    sp = tuple(int(v) for v in data.shape[2:])
    dummy_t = torch.zeros((B,) + sp, dtype=torch.int64, device=data.device)
    dummy_l = torch.zeros((B,) + sp, dtype=torch.float32, device=data.device)
    return _mix(data, dummy_t, dummy_l, mode, desc=desc)[0]


def _mix(data, target, logits, mode, desc=None):
    L.require_gpu(data, target, logits)
    if mode not in _MODES:                                   # reference: mask of ones -> the inputs unchanged
        return data, target.long(), logits
    B, Cimg = int(data.shape[0]), int(data.shape[1])
    sp = tuple(int(v) for v in data.shape[2:])
    H, W, Z = sp[0], sp[1], (sp[2] if len(sp) == 3 else 1)
    data = data.to(torch.float32).contiguous()
    logits = logits.to(torch.float32).contiguous()
    tgt = target.to(torch.int64).contiguous()
    if desc is not None:
        assert mode != "classmix" and desc.shape == (B, 8)
    elif mode == "classmix":
        desc = np.zeros((B, 8), dtype=np.int32)
        desc[:, 5] = 1
        sets = _label_sets(tgt)
        for i in range(B):
            sel = _select_half(sets[i])
            desc[i, 6], desc[i, 7] = np.uint32(sel & 0xFFFFFFFF).astype(np.int32), np.uint32(sel >> 32).astype(np.int32)
    else:
        desc = draw_boxes(B, sp)
    odata, otarget, ologits = torch.empty_like(data), torch.empty_like(tgt), torch.empty_like(logits)
    L.call("arco_mix_unsup", L.ptr(data), Cimg, L.ptr(tgt), L.ptr(logits), B, H, W, Z, desc.ctypes.data_as(ctypes.c_void_p), _MODES[mode], L.ptr(odata),
           L.ptr(otarget), L.ptr(ologits))
    if mode == "cutout" and target.dtype == torch.int64 and target.is_contiguous():
        target.copy_(otarget)                                # the reference writes the -1s into the caller's target (:292)
    return odata, otarget, ologits


def generate_unsup_data(data, target, logits, mode='cutout', desc=None):
    """(new_data [b,c,H,W], new_target int64 [b,H,W], new_logits [b,H,W]) - augment.py:284-313.
    cutout: data, logits zeroed and target = -1 inside a random half-area rectangle per image; cutmix: that rectangle
    is filled from image (i+1) % b; classmix: a random half of image i's labels keep their pixels, the rest comes from
    image (i+1) % b.  Any other mode returns the inputs."""
    assert data.dim() == 4, data.shape
    return _mix(data, target, logits, mode, desc=desc)          # desc: boxes drawn earlier (draw_boxes)


def generate_unsup_data_3d(data, target, logits, mode='cutout', desc=None):
    """Volume variant (augment_3d.py:228-257): the box is 10 slices deep along the last axis.  desc: boxes drawn earlier (draw_boxes)."""
    assert data.dim() == 5, data.shape
    return _mix(data, target, logits, mode, desc=desc)


# ---------------------------------------------------------------------------------------------------------------------
# batch_transform (augment.py:255-281): per-image PIL round trip (8-bit quantisation), ColorJitter + GaussianBlur with
# probability 0.5 each when apply_augmentation, then AdvMorph on the whole batch with probability 0.5
# ---------------------------------------------------------------------------------------------------------------------
def _gaussian_blur_radius(sigma, passes=3):
    """Pillow's _gaussian_blur_radius (BoxBlur.c): fractional box radius of the 3-pass box approximation of a Gaussian."""
    import math
    sigma2 = float(np.float32(sigma)) * float(np.float32(sigma)) / passes
    Lb = math.sqrt(12.0 * sigma2 + 1.0)
    l = math.floor((Lb - 1.0) / 2.0)
    a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2)
    a /= 6 * (sigma2 - (l + 1) * (l + 1))
    return float(np.float32(l + a))


def draw_batch_transform_params(n_images, apply_augmentation, scale_size=(1.0, 1.0)):
    """The host draws of batch_transform in the reference's order (python `random`, torch CPU generator): per image
    random.uniform(*scale_size) (transform, augment.py:134), then with apply_augmentation torch.rand(1) [-> torchvision
    ColorJitter.get_params: torch.randperm(4) and four torch.empty(1).uniform_ draws] and torch.rand(1) [-> python
    random.uniform(0.15, 1.15)]; after the loop one torch.rand(1) for AdvMorph - drawn even without augmentation
    (`torch.rand(1) > 0.5 and apply_augmentation`, :271).  Returns (list of per-image dicts, morph flag)."""
    import random
    out = []
    for _ in range(n_images):
        random.uniform(scale_size[0], scale_size[1])
        d = dict(order=None, factors=None, sigma=None)
        if apply_augmentation:
            if float(torch.rand(1)) > 0.5:
                d["order"] = [int(v) for v in torch.randperm(4)]
                d["factors"] = tuple(float(torch.empty(1).uniform_(lo, hi))
                                     for lo, hi in ((0.75, 1.25), (0.75, 1.25), (0.75, 1.25), (-0.25, 0.25)))
            if float(torch.rand(1)) > 0.5:
                d["sigma"] = random.uniform(0.15, 1.15)
        out.append(d)
    morph = bool(float(torch.rand(1)) > 0.5) and bool(apply_augmentation)
    return out, morph


def jitter_blur(data, params):
    """to_tensor(GaussianBlur(ColorJitter(to_pil_image(x)))) for every image of data [B, C, H, W] (C = 1 or 3) with the
    per-image parameters of draw_batch_transform_params - one launch sequence for the batch, Pillow's 8-bit integer
    arithmetic (csrc/augment.hip)."""
    import struct
    L.require_gpu(data)
    x = data.to(torch.float32).contiguous()
    B, C, H, W = (int(v) for v in x.shape)
    assert len(params) == B and L.query("arco_jitter_desc_bytes") == 44
    raw = b""
    for p in params:
        order = p["order"] if p["order"] is not None else [0, 1, 2, 3]
        f = p["factors"] if p["factors"] is not None else (1.0, 1.0, 1.0, 0.0)
        raw += struct.pack("<4i4fiif", *order, *f, int(p["order"] is not None), int(p["sigma"] is not None),
                           _gaussian_blur_radius(p["sigma"]) if p["sigma"] is not None else 0.0)
    out = torch.empty_like(x)
    for b0 in range(0, B, 32):                         # (the descriptor table holds 32 images)
        nb = min(32, B - b0)
        ws = torch.empty(2 * nb + nb * C * H * W, dtype=torch.float32, device=x.device)
        L.call("arco_jitter_blur", L.ptr(x[b0:b0 + nb]), nb, C, H, W, raw[44 * b0:44 * (b0 + nb)], L.ptr(ws), L.ptr(out[b0:b0 + nb]))
    return out


def batch_transform(data, label, logits, crop_size, scale_size, apply_augmentation):
    """augment.batch_transform (augment.py:255-281) on the GPU for the trainers' use (crop_size == image size,
    scale_size == (1.0, 1.0): the resize / pad / crop of `transform` are the identity and RandomCrop.get_params draws
    nothing).  Returns (data, label, logits) like the reference: data through the 8-bit PIL round trip (+ ColorJitter /
    GaussianBlur / AdvMorph when apply_augmentation), labels unchanged (their /255 -> byte -> *255 round trip is exact,
    255 -> -1), logits quantised to k/255.  Generator consumption = the reference's (draw_batch_transform_params)."""
    from .adv_morph import AdvMorph
    B = int(data.shape[0])
    if tuple(scale_size) != (1.0, 1.0) or (crop_size != -1 and tuple(int(v) for v in crop_size) != tuple(int(v) for v in data.shape[-2:])):
        raise NotImplementedError("arco_amd.batch_transform: only the trainers' configuration (same-size crop, scale 1.0)")
    params, morph = draw_batch_transform_params(B, apply_augmentation, scale_size)
    data_t = jitter_blur(data, params)
    # a byte image (the reference passes `torch.ones_like(uint8 label) * 255` for the labeled stream, train_arco_2d.py:288)
    # becomes k/255 through ToTensor; float confidences are quantised to the 8-bit grid by the PIL round trip
    lg = (logits.to(torch.float32) / 255.0 if not logits.is_floating_point() else logits.to(torch.float32)).contiguous()
    logits_t = torch.empty_like(lg)
    L.call("arco_quantize8", L.ptr(lg), lg.numel(), L.ptr(logits_t))
    if morph:                                                          # :271-279
        ds = list(data.shape)
        aug = AdvMorph(config_dict={'epsilon': 1.5, 'xi': 0.5, 'data_size': ds, 'vector_size': [ds[-1] // 8, ds[-1] // 8],
                                    'interpolator_mode': 'bilinear'}, debug=False, use_gpu=True)
        aug.init_parameters()
        data_t = aug.forward(data_t).contiguous()
    return data_t, label, logits_t


def randomGeneratorWithLogits(image, label, logit, output_size=[256, 256]):
    """augment.py:339-369 (train_arco_2d.py:292-293): per-image scipy `zoom(order=0)` of image / pseudo-label / confidence to
    `output_size`, the labels through a uint8 cast (values wrap modulo 256, like the reference's astype(np.uint8)).
    At the trainers' configuration the zoom factor is 1.0 - nearest-neighbour resampling at the same size is the
    identity - and everything stays on the GPU; other sizes take the reference's host round trip through scipy."""
    B, _, x, y = image.shape
    if (int(output_size[0]), int(output_size[1])) == (int(x), int(y)):
        img = image[:, :1].to(torch.float32)
        return img, (label.to(torch.int64) & 255), logit
    from scipy.ndimage import zoom
    fac = (output_size[0] / x, output_size[1] / y)
    dev = image.device
    img = np.stack([zoom(image[i, 0].detach().cpu().numpy(), fac, order=0) for i in range(B)]).astype(np.float32)
    lab = np.stack([zoom(label[i].detach().cpu().numpy(), fac, order=0) for i in range(B)]).astype(np.uint8)
    lg = np.stack([zoom(logit[i].detach().cpu().numpy(), fac, order=0) for i in range(B)])
    return torch.from_numpy(img).unsqueeze(1).to(dev), torch.from_numpy(lab).long().to(dev), torch.from_numpy(lg).to(dev)
