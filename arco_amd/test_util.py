"""Evaluation of the 3-D model - drop-in for the array-level part of the reference's
code/test_util.py: `test_single_case` (:139-211, sliding-window inference), `calculate_metric_percase` (:214-220),
`getLargestCC` (:11-15) and the loop of `test_all_case` (:39-79) over already-loaded (image, label) arrays (the h5 /
nibabel file handling around it is I/O and stays with the caller).

The window loop is the reference's: windows are visited in (x, y, z) order, the last one of each axis is clamped
to the border, volumes smaller than the patch are zero-padded symmetrically.  The volume, the score map and the
window counter stay on the GPU: each window is one eval-mode forward (HIP kernels), one softmax and one
accumulate launch; a final launch divides by the counter and takes the arg-max.  Several windows can share one
forward (`batch`): eval-mode results do not depend on the batch, and the accumulate launches keep the reference's
order, so the fp32 sums are the same."""
import math

import numpy as np
import torch

from . import _lib as L
from . import glue
from .utils.metrics import binary as _binary


def getLargestCC(segmentation):
    """Largest connected component (full connectivity, like skimage.measure.label's default), test_util.py:11-15."""
    from scipy.ndimage import generate_binary_structure, label
    seg = np.asarray(segmentation)
    labels, n = label(seg, structure=generate_binary_structure(seg.ndim, seg.ndim))
    assert n != 0                        # assume at least 1 CC
    return labels == np.argmax(np.bincount(labels.flat)[1:]) + 1


def _window_starts(size, patch, stride):
    n = math.ceil((size - patch) / stride) + 1
    return [min(stride * i, size - patch) for i in range(n)]


@torch.no_grad()
def test_single_case(net, image, stride_xy, stride_z, patch_size, num_classes=1, batch=4, device="cuda:0"):
    """(label_map int64 [w,h,d], score_map float32 [C,w,h,d]) of one volume - test_util.py:139-211."""
    image = np.asarray(image)
    w, h, d = image.shape
    pads = []
    for s, p in zip((w, h, d), patch_size):
        tot = max(p - s, 0)
        pads.append((tot // 2, tot - tot // 2))
    add_pad = any(a + b > 0 for a, b in pads)
    if add_pad:
        image = np.pad(image, pads, mode='constant', constant_values=0)
    ww, hh, dd = image.shape
    px, py, pz = (int(v) for v in patch_size)
    vol = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).to(device)
    C = int(num_classes)
    score = torch.zeros((C, ww, hh, dd), dtype=torch.float32, device=device)
    cnt = torch.zeros((ww, hh, dd), dtype=torch.float32, device=device)
    starts = [(xs, ys, zs) for xs in _window_starts(ww, px, stride_xy) for ys in _window_starts(hh, py, stride_xy)
              for zs in _window_starts(dd, pz, stride_z)]
    was_training = net.training
    net.eval()
    for s0 in range(0, len(starts), batch):
        grp = starts[s0:s0 + batch]
        patches = torch.stack([vol[xs:xs + px, ys:ys + py, zs:zs + pz] for xs, ys, zs in grp]).unsqueeze(1)
        y1 = net(patches)
        if isinstance(y1, (tuple, list)):
            y1 = y1[0]
        assert int(y1.shape[1]) == C, (y1.shape, C)
        y = glue.softmax(y1)                                     # [n, C, px, py, pz] planes
        for i, (xs, ys, zs) in enumerate(grp):
            L.call("arco_window_accumulate", L.ptr(y[i]), C, px, py, pz, L.ptr(score), L.ptr(cnt), ww, hh, dd, xs, ys, zs)
    if was_training:
        net.train()
    label = torch.empty((ww, hh, dd), dtype=torch.int64, device=device)
    L.call("arco_score_finalize", L.ptr(score), L.ptr(cnt), C, ww * hh * dd, L.ptr(label))
    label_map, score_map = label.cpu().numpy(), score.cpu().numpy()
    if add_pad:
        (wl, _), (hl, _), (dl, _) = pads
        label_map = label_map[wl:wl + w, hl:hl + h, dl:dl + d]
        score_map = score_map[:, wl:wl + w, hl:hl + h, dl:dl + d]
    return label_map, score_map


def calculate_metric_percase(pred, gt):
    """(dice, jc, hd95, asd), test_util.py:214-220."""
    return _binary.dc(pred, gt), _binary.jc(pred, gt), _binary.hd95(pred, gt), _binary.asd(pred, gt)


def test_all_case(model, cases, num_classes, patch_size=(112, 112, 80), stride_xy=18, stride_z=4, preproc_fn=None,
                  metric_detail=0, nms=0, device="cuda:0"):
    """Mean (dice, jc, hd95, asd) over `cases` = iterable of (image, label) arrays - the loop of test_util.py:39-79
    (an all-background prediction scores (0, 0, 0, 0), :57-58)."""
    total, n = np.zeros(4), 0
    for ith, (image, label) in enumerate(cases):
        if preproc_fn is not None:
            image = preproc_fn(image)
        prediction, _ = test_single_case(model, image, stride_xy, stride_z, patch_size, num_classes=num_classes, device=device)
        if nms:
            prediction = getLargestCC(prediction)
        single = (0, 0, 0, 0) if np.sum(prediction) == 0 else calculate_metric_percase(prediction, np.asarray(label))
        if metric_detail:
            print('%02d,\t%.5f, %.5f, %.5f, %.5f' % (ith, *single))
        total += np.asarray(single, dtype=np.float64)
        n += 1
    avg = total / max(n, 1)
    print('average metric is {}'.format(avg))
    return avg
