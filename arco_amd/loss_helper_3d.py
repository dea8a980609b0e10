"""Drop-in for the reference's code/loss_helper_3d.py - the 4-D (B,C,H,W) = 2-D image
version of the contrastive loss that train_arco_2d.py imports (train_arco_2d.py:24).
Same public names and signatures (loss_helper_3d.py:12,35,83,120,187,271); plus every other name the reference trainer
resolves through `from loss_helper_3d import *` (derived from the reference source: tests/golden/g9_flags.json
"trainer_names"): `LocalConLoss` (:1194, constructed at train_arco_2d.py:270), `label_onehot` (:892), and the module
aliases `F`, `nn`, `np`, `torch` the trainer uses without importing them itself."""
import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401
import torch.nn.functional as F  # noqa: F401

from ._contrast import compute_contra_memobank_loss, dequeue_and_enqueue
from ._supcon import LocalConLoss, SupConLoss
from .samplers import (as_monte_carlo_sample, grid_as_monte_carlo_sample, grid_monte_carlo_sample,
                       monte_carlo_sample)


def label_onehot(inputs, num_segments):
    """loss_helper_3d.py:892-901 / loss_helper.py:1065-1074, quirk included: the reference scatters along dim 0 of a
    [C, B, H, W] buffer with a [B, 1, H, W] index, which writes only into sample 0 - whose "one-hot" is the UNION over the
    batch of the samples' labels - and leaves the other samples zero; label 255 = ignore (that sample's column zeroed).
    Returns float [B, C, H, W].  Pinned to the reference function (tests/golden/g16_boundary.npz).  The trainers never
    reach it: they define their own `label_onehot` below their imports (train_arco_2d.py:492-498, negatives clamped to
    class 0), which is `arco_amd.glue.label_onehot`."""
    from . import glue
    ignore = inputs == 255
    hot = glue.label_onehot(inputs.masked_fill(ignore, 0), num_segments)
    out = torch.zeros(hot.shape, dtype=torch.float32, device=hot.device)
    out[0] = hot.amax(dim=0).to(torch.float32)
    return out.masked_fill(ignore.unsqueeze(1), 0.0)


__all__ = ["compute_contra_memobank_loss", "dequeue_and_enqueue", "grid_monte_carlo_sample",
           "grid_as_monte_carlo_sample", "monte_carlo_sample", "as_monte_carlo_sample", "LocalConLoss", "SupConLoss",
           "label_onehot", "F", "nn", "np", "torch"]
