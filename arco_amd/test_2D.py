"""Evaluation of the 2-D model - drop-in for the array-level part of the reference's
code/test_2D.py: `calculate_metric_percase` (:52-66) and `test_single_volume` (:67-92; the h5 / SimpleITK file
handling around it is I/O and stays with the caller).

Inference runs in eval mode (BatchNorm on running statistics) on the HIP kernels, ALL slices of a volume in one
batch (eval-mode results do not depend on the batch); the zoom(order=0) round trip is scipy on the host exactly as
in the reference; the overlap counts behind Dice / Jaccard are one integer-atomics kernel.  hd95 / asd are medpy's
surface distances (host scipy code) restated in utils/metrics.py."""
import numpy as np
import torch

from . import _lib as L
from . import glue

from .utils import metrics as _medpy_metric       # medpy.metric.binary's dc / jc / hd95 / asd on scipy


def overlap_counts(pred, gt, classes):
    """int64 [classes, 3]: |pred == c|, |gt == c|, |both| for two integer label maps on the GPU."""
    L.require_gpu(pred, gt)
    p = pred.to(torch.int64).contiguous().view(-1)
    g = gt.to(torch.int64).contiguous().view(-1)
    out = torch.empty((classes, 3), dtype=torch.int64, device=p.device)
    L.call("arco_overlap_counts", L.ptr(p), L.ptr(g), p.numel(), classes, L.ptr(out))
    return out


def _dice_jc(n_pred, n_gt, n_both):
    if n_pred > 0 and n_gt > 0:
        return 2.0 * n_both / (n_pred + n_gt), n_both / (n_pred + n_gt - n_both)
    if n_pred > 0 and n_gt == 0:
        return 1.0, 1.0                 # the reference's convention (test_2D.py:62-63)
    return 0.0, 0.0


def calculate_metric_percase(pred, gt):
    """(dice, jc, hd95, asd) of two binary masks (numpy or torch), test_2D.py:52-66."""
    pred = np.asarray(pred.cpu() if torch.is_tensor(pred) else pred) > 0
    gt = np.asarray(gt.cpu() if torch.is_tensor(gt) else gt) > 0
    dice, jc = _dice_jc(int(pred.sum()), int(gt.sum()), int(np.logical_and(pred, gt).sum()))
    hd95 = asd = 0.0
    if pred.sum() > 0 and gt.sum() > 0:
        asd = _medpy_metric.binary.asd(pred, gt)
        hd95 = _medpy_metric.binary.hd95(pred, gt)
    return dice, jc, hd95, asd


@torch.no_grad()
def predict_volume(image, net, patch_size=(256, 256), batch=32, device="cuda:0"):
    """Label map of a [S, X, Y] volume: per slice zoom(order=0) to patch_size, eval-mode net, argmax(softmax), zoom
    back (test_2D.py:72-88).  Returns an int64 numpy array of image.shape."""
    from scipy.ndimage import zoom
    was_training = net.training
    net.eval()
    S, x, y = image.shape
    zoomed = np.stack([zoom(image[i], (patch_size[0] / x, patch_size[1] / y), order=0) for i in range(S)])
    prediction = np.zeros((S, x, y), dtype=np.int64)
    for s0 in range(0, S, batch):
        inp = torch.from_numpy(zoomed[s0:s0 + batch]).unsqueeze(1).float().to(device)
        logits = net(inp)[0]
        _, amax = glue.softmax_max(logits)                       # argmax(softmax(.)) == argmax of the logits' softmax
        out = amax.cpu().numpy()
        for i in range(out.shape[0]):
            prediction[s0 + i] = zoom(out[i], (x / patch_size[0], y / patch_size[1]), order=0)
    if was_training:
        net.train()
    return prediction


@torch.no_grad()
def test_single_volume(image, label, net, classes, patch_size=(256, 256), device="cuda:0"):
    """metric_list of test_2D.py:90-92: (dice, jc, hd95, asd) for every class 1..classes-1 of one case."""
    prediction = predict_volume(image, net, patch_size, device=device)
    cnt = overlap_counts(torch.from_numpy(prediction).to(device), torch.from_numpy(np.asarray(label, dtype=np.int64)).to(device),
                         classes).cpu().numpy()
    metric_list = []
    for c in range(1, classes):
        dice, jc = _dice_jc(int(cnt[c, 0]), int(cnt[c, 1]), int(cnt[c, 2]))
        hd95 = asd = 0.0
        if cnt[c, 0] > 0 and cnt[c, 1] > 0:
            asd = _medpy_metric.binary.asd(prediction == c, label == c)
            hd95 = _medpy_metric.binary.hd95(prediction == c, label == c)
        metric_list.append((dice, jc, hd95, asd))
    return metric_list
