"""`LocalConLoss` / `SupConLoss` of the reference's loss_helper_3d.py:1121-1252 (the same classes sit in loss_helper.py).

train_arco_2d.py:270 constructs `LocalConLoss(temperature=0.7, stride=8)` and never calls it - the name must resolve for
the trainer to start, so it is part of the drop-in surface.  The forward is restated in GEMM form (the reference builds
the same [N, N] similarity table with F.conv2d over 1x1 'kernels'): N = bsz * views * h * w rows R of c channels in
(view-major sample, y, x) order, logits = R R^T / T, self-pairs masked out, positives = equal labels (or the same
pixel of the other views when no labels are given), loss = - mean over positives of log-softmax, averaged over the
foreground rows (labels other than 0 and -1) or over all rows.  Off the training step's path: plain tensor ops on
whatever device the inputs live on.  Pinned to the reference classes in tests/golden/g16_boundary.npz."""
import torch
import torch.nn as nn


class SupConLoss(nn.Module):
    def __init__(self, temperature=0.07, contrast_mode='all', base_temperature=0.07):
        super(SupConLoss, self).__init__()
        self.temperature = temperature
        self.contrast_mode = contrast_mode
        self.base_temperature = base_temperature

    def forward(self, features, labels=None):
        if features.dim() < 3:
            raise ValueError('`features` needs to be [bsz, n_views, ...],at least 3 dimensions are required')
        bsz, views, c = features.shape[:3]
        rows = features.transpose(0, 1).reshape(views * bsz, c, -1).transpose(1, 2).reshape(-1, c)   # (view, sample, y, x) rows
        n = rows.shape[0]
        logits = rows @ rows.t() / self.temperature
        off_diag = 1.0 - torch.eye(n, dtype=logits.dtype, device=logits.device)
        if labels is not None:
            lab = labels.transpose(0, 1).reshape(-1, 1)
            positives = (lab == lab.t()).to(logits.dtype) * off_diag
            foreground = ((lab.view(-1) != -1) & (lab.view(-1) != 0)).to(torch.int32)
        else:
            per_view = n // views
            positives = torch.eye(per_view, dtype=logits.dtype, device=logits.device).repeat(views, views) * off_diag
        log_prob = logits - torch.log((torch.exp(logits) * off_diag).sum(1, keepdim=True))
        loss = -(positives * log_prob).sum(1) / positives.sum(1)
        if labels is not None:
            return (loss * foreground).sum() / foreground.sum()
        return loss.mean()


class LocalConLoss(nn.Module):
    def __init__(self, temperature=0.7, stride=4):
        super(LocalConLoss, self).__init__()
        self.temp = temperature
        self.device = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')
        self.supconloss = SupConLoss(temperature=self.temp)
        self.stride = stride

    def forward(self, features, labels=None):
        s = self.stride
        features = features[:, :, :, ::s, ::s]               # [bsz, views, c, h, w] subsampled (memory / time)
        if labels is None:
            return self.supconloss(features)
        labels = labels[:, :, ::s, ::s]
        if labels.sum() == 0:
            return torch.tensor(0).float().to(self.device)
        return self.supconloss(features, labels)
