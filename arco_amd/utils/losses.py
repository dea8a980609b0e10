"""Drop-in for the one class the reference trainers take from code/utils/losses.py: `losses.DiceLoss`
(train_arco_2d.py:17,269,338; train_arco_3d.py:245,308; pretrain_2D.py:199; pretrain_3D.py:172).

Same constructor and call signature as utils/losses.py:173-209: `DiceLoss(n_classes)(inputs, target, weight=None,
softmax=False)` with `inputs` [B, C, *spatial] scores and `target` [B, 1, *spatial] integer labels; the three per-class
sums and the gradient are HIP kernels (`arco_dice_probs_fwd/bwd`, fp64 fixed-order reduction).  The package's own
trainers use the fused CE + Dice kernel on the logits instead (`glue.supervised_loss`, one pass)."""
import torch
import torch.nn as nn

from .. import _lib as L


class _DiceProbsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, lab, wgt):
        M, C = rows.shape
        ws = torch.empty(L.query("arco_seg_ws_doubles", M, C, 1), dtype=torch.float64, device=rows.device)
        out = torch.empty(1, dtype=torch.float32, device=rows.device)
        L.call("arco_dice_probs_fwd", L.ptr(rows), C, M, C, L.ptr(lab), L.ptr(wgt), L.ptr(ws), L.ptr(out))
        ctx.save_for_backward(rows, lab, ws, wgt if wgt is not None else out.new_empty(0))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        rows, lab, ws, wgt = ctx.saved_tensors
        M, C = rows.shape
        d = torch.empty_like(rows)
        L.call("arco_dice_probs_bwd", L.ptr(rows), C, M, C, L.ptr(lab), L.ptr(wgt if wgt.numel() else None), L.ptr(ws),
               L.ptr(g.contiguous().float()), L.ptr(d), C)
        return d, None, None


class DiceLoss(nn.Module):
    def __init__(self, n_classes):
        super(DiceLoss, self).__init__()
        self.n_classes = n_classes

    def forward(self, inputs, target, weight=None, softmax=False):
        L.require_gpu(inputs, target)
        if softmax:
            inputs = torch.softmax(inputs, dim=1)
        C = self.n_classes
        assert inputs.shape[1] == C and inputs.shape[0] == target.shape[0] and inputs.shape[2:] == target.shape[2:] \
            and target.shape[1] == 1, 'predict & target shape do not match'
        rows = inputs.movedim(1, -1)                     # channels-last rows [M, C]: free for this package's network outputs
        shape = rows.shape
        rows = rows.reshape(-1, C).float().contiguous()
        lab = target.reshape(-1).to(torch.int64).contiguous()
        wgt = None if weight is None else torch.as_tensor(weight, dtype=torch.float32, device=rows.device).contiguous()
        return _DiceProbsFn.apply(rows, lab, wgt)
