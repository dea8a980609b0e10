"""Drop-in for the reference's code/model_2D.py on MI355X: FeatureExtractor (:20-55),
create_model (:57-64), ISD (:115-198).  Same names, constructor arguments, attribute names
(.model, .ema_model, ._momentum_update_key_encoder(), .data_parallel()) and state_dict keys.

FeatureExtractor is five bias-free 1x1 convolutions = fp32 MFMA GEMMs over channels-last
pixels with the residual add fused in the GEMM epilogue, chained by align_corners bilinear
resizes.  The MoCo-style heads ISD owns (latent/outputs heads, predictors, queues) are used by
stage-1 pre-training only (ISD.forward, arco_amd/pretrain_2D.py): a few 4- and 256-wide layers
on pooled maps, run as plain tensor ops around the two HIP U-Nets; _momentum_update_key_encoder
(model_2D.py:176-182) EMA-updates them every step in both stages.
"""
import numpy as np  # noqa: F401  (np / F / nn reach the reference trainers through `from model_xD import *`)
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401

from . import ops, stage1
from . import optim as _optim      # NOT `optim`: the reference trainers do `import torch.optim as optim` and then `from model_2D import *`
from .networks.net_factory_args import net_factory


class FeatureExtractor(nn.Module):
    def __init__(self, fea_dim=[256, 128, 64, 32, 16], output_dim=256) -> None:
        super().__init__()
        assert len(fea_dim) == 5, 'input_dim is not correct'
        cnt = fea_dim[0]
        self.fea0 = nn.Conv2d(in_channels=cnt, out_channels=cnt, kernel_size=1, bias=False)
        cnt += fea_dim[1]
        self.fea1 = nn.Conv2d(in_channels=cnt, out_channels=cnt, kernel_size=1, bias=False)
        cnt += fea_dim[2]
        self.fea2 = nn.Conv2d(in_channels=cnt, out_channels=cnt, kernel_size=1, bias=False)
        cnt += fea_dim[3]
        self.fea3 = nn.Conv2d(in_channels=cnt, out_channels=cnt, kernel_size=1, bias=False)
        cnt += fea_dim[4]
        self.fea4 = nn.Conv2d(in_channels=cnt, out_channels=output_dim, kernel_size=1, bias=False)

    def forward_lowres(self, fea_list):
        """Everything up to (not including) the last upsample: returns (fea3(x)+x at the
        second-finest level, finest feature map).  Used by the row-sparse head (arco_amd.head)."""
        f = [ops.to_channels_last(t) for t in fea_list]
        x = ops.conv(f[0], self.fea0.weight, None, residual=True)          # fea0(f0) + f0
        for i, fea in enumerate((self.fea1, self.fea2, self.fea3), start=1):
            x = ops.bilinear(x, f[i].shape[-2:])
            x = torch.cat((x, f[i]), dim=1)
            x = ops.conv(x, fea.weight, None, residual=True)               # fea_i(x) + x
        return x, f[4]

    def forward_lowres1(self, fea_list):
        """Up to fea1 (second level): returns (fea1(x)+x, f2, f3, f4) for the three-level row-sparse head
        (arco_amd.head.lazy_head3); the same commuted evaluation as forward_lowres2's first level."""
        f = [ops.to_channels_last(t) for t in fea_list]
        x = ops.conv(f[0], self.fea0.weight, None, residual=True)
        c = int(x.shape[1])
        n = c + int(f[1].shape[1])
        w_lo, w_hi = ops.fold_residual(self.fea1.weight, c)          # (W + I)[:, :c], (W + I)[:, c:] in one launch
        lo = ops.conv(x, w_lo)
        x = ops.conv(f[1], w_hi, None, residual=ops.bilinear(lo, f[1].shape[-2:]))
        return x, f[2], f[3], f[4]

    def forward_lowres2(self, fea_list):
        """Up to fea2 (third level): returns (fea2(x)+x, f3, f4) for the two-level row-sparse head.

        Same function as the reference order  x <- fea_i(cat(up(x), f_i)) + cat(up(x), f_i)  (model_2D.py:43-50),
        evaluated with the 1x1 conv pushed BELOW the upsample: a bias-free 1x1 conv acts per pixel and bilinear
        interpolation per channel, so they commute, and with W' = W_i + I (residual folded into the weights)
            x <- up(W'[:, :c] . x) + W'[:, c:] . f_i              (c = channels of the low-resolution x).
        The wide block of W' (c of the c + c_i input channels) then runs on 4x fewer pixels: 26.3 -> 9.4 GFLOP for
        fea2 at config 2 (and the same factor in its dgrad / wgrad).  Values differ from the reference order by
        fp32 rounding only (tests/test_head_gpu.py)."""
        f = [ops.to_channels_last(t) for t in fea_list]
        x = ops.conv(f[0], self.fea0.weight, None, residual=True)
        for i, fea in enumerate((self.fea1, self.fea2), start=1):
            c = int(x.shape[1])
            n = c + int(f[i].shape[1])
            w_lo, w_hi = ops.fold_residual(fea.weight, c)                            # (W + I)[:, :c], (W + I)[:, c:]
            lo = ops.conv(x, w_lo)                                                   # at the low resolution
            x = ops.conv(f[i], w_hi, None, residual=ops.bilinear(lo, f[i].shape[-2:]))
        return x, f[3], f[4]

    def forward_reference_order(self, fea_list):
        """model_2D.py:43-53 literally: upsample, concatenate, convolve at every level (the [B, 496, 256, 256] concatenation and a
        1 M-row 496 x 496 GEMM at the last one).  Kept as the comparison target of the tests."""
        x, f4 = self.forward_lowres(fea_list)
        x = ops.bilinear(x, f4.shape[-2:])
        x = torch.cat((x, f4), dim=1)
        return ops.conv(x, self.fea4.weight, None, residual=False)         # fea4(x)

    def forward(self, fea_list):
        """The dense representation [B, output_dim, H, W] (model_2D.py:43-53) with every 1x1 convolution pushed BELOW its upsample
        (forward_lowres2's identity: a bias-free 1x1 conv acts per pixel, bilinear interpolation per channel, so they commute):
            x <- up((W_i + I)[:, :c] . x) + (W_i + I)[:, c:] . f_i        (levels 1-3),      rep = up(W_4[:, :c] . x) + W_4[:, c:] . f_4.
        The wide block of every level runs on 4x fewer pixels and no concatenation is materialised: at config 2 the last level is a
        [262 144 x 480 x 496] GEMM + a K = 16 GEMM with the upsampled product as its residual operand instead of a 2 GB concatenation
        and a [1 048 576 x 496 x 496] GEMM (516 -> 141 GFLOP; the same factor in the backward).  Values differ from the reference
        order by fp32 rounding only (tests/test_nets_gpu.py, tests/test_head_gpu.py)."""
        f = [ops.to_channels_last(t) for t in fea_list]
        x = ops.conv(f[0], self.fea0.weight, None, residual=True)
        for i, fea in enumerate((self.fea1, self.fea2, self.fea3), start=1):
            c = int(x.shape[1])
            w_lo, w_hi = ops.fold_residual(fea.weight, c)                            # (W + I)[:, :c], (W + I)[:, c:]
            lo = ops.conv(x, w_lo)                                                   # at the low resolution
            x = ops.conv(f[i], w_hi, None, residual=ops.bilinear(lo, f[i].shape[-2:]))
        c = int(x.shape[1])
        w = self.fea4.weight                                                          # no residual at the last level
        lo = ops.conv(x, w[:, :c].contiguous())
        return ops.boundary(ops.conv(f[4], w[:, c:].contiguous(), None, residual=ops.bilinear(lo, f[4].shape[-2:])))


def create_model(ema=False, num_classes=4, train_encoder=True, train_decoder=True, in_chns=1):
    """model_2D.py:57-64 (the reference hard-codes in_chns=1; `in_chns` is an extension for RGB inputs)."""
    model = net_factory(net_type='unet', in_chns=in_chns, class_num=num_classes, train_encoder=train_encoder,
                        train_decoder=train_decoder)
    if ema:
        for param in model.parameters():
            param.detach_()
    return model


class ProjectionHead(nn.Module):
    def __init__(self, dim_in=4, proj_dim=4, output_pooling_size=16, proj='convmlp'):
        super(ProjectionHead, self).__init__()
        if proj == 'linear':
            self.proj = nn.Conv2d(dim_in, proj_dim, kernel_size=1)
        elif proj == 'convmlp':
            self.proj = nn.Sequential(nn.AdaptiveAvgPool2d(output_pooling_size),
                                      nn.Conv2d(dim_in, dim_in * 2, kernel_size=1),
                                      nn.Conv2d(dim_in * 2, proj_dim, kernel_size=1))


class MLP(nn.Module):
    def __init__(self, input_channels=256, num_class=128, pooling_size=1):
        super().__init__()
        self.gap = nn.AdaptiveAvgPool2d(pooling_size)
        self.f1 = nn.Linear(input_channels * pooling_size ** 2, input_channels)
        self.f2 = nn.Linear(input_channels, num_class)


class ISD(nn.Module):
    def __init__(self, K=48, m=0.99, Ts=0.1, Tt=0.01, num_classes=4, train_encoder=True, train_decoder=True,
                 latent_pooling_size=1, latent_feature_size=256, output_pooling_size=16, patch_size=64, in_chns=1):
        super(ISD, self).__init__()
        self.K, self.m, self.Ts, self.Tt = K, m, Ts, Tt
        self.num_classes = num_classes
        self.patch_size = patch_size
        self.latent_feature_size = latent_feature_size
        self.model = create_model(num_classes=num_classes, train_encoder=train_encoder, train_decoder=train_decoder,
                                  in_chns=in_chns)
        self.ema_model = create_model(ema=True, num_classes=num_classes, train_encoder=False, train_decoder=False,
                                      in_chns=in_chns)
        self.k_latent_head = MLP(256, self.latent_feature_size, latent_pooling_size)
        self.q_latent_head = MLP(256, self.latent_feature_size, latent_pooling_size)
        self.latent_predictor = nn.Sequential(nn.Linear(self.latent_feature_size, self.latent_feature_size),
                                              nn.Linear(self.latent_feature_size, self.latent_feature_size))
        self.k_outputs_head = ProjectionHead(num_classes, num_classes, output_pooling_size)
        self.q_outputs_head = ProjectionHead(num_classes, num_classes, output_pooling_size)
        self.outputs_predictor = nn.Sequential(nn.Conv2d(num_classes, num_classes, kernel_size=1),
                                               nn.Conv2d(num_classes, num_classes, kernel_size=1))
        for param_q, param_k in zip(self.model.parameters(), self.ema_model.parameters()):
            param_k.data.copy_(param_q.data)
            param_k.requires_grad = False
        self.register_buffer('queue', nn.functional.normalize(torch.randn(self.K, self.latent_feature_size), dim=0))
        self.register_buffer('queue_mask', nn.functional.normalize(
            torch.randn(self.K, 49, num_classes * output_pooling_size ** 2), dim=0))
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long))
        self.register_buffer('mask_queue_ptr', torch.zeros(1, dtype=torch.long))
        self._ema_pairs = None

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        """k = m*k + (1-m)*q over parameters() only (BN buffers untouched), model_2D.py:176-182.
        One fused kernel per (student, teacher) module pair on flat parameter buffers."""
        self._ensure_ema_pairs()
        for pair in self._ema_pairs:
            pair.update(self.m)

    def _ensure_ema_pairs(self):
        if self._ema_pairs is None:
            self._ema_pairs = [_optim.EmaPair(q, k) for q, k in (
                (self._unwrap(self.model), self._unwrap(self.ema_model)),
                (self._unwrap(self.q_outputs_head), self._unwrap(self.k_outputs_head)),
                (self._unwrap(self.q_latent_head), self._unwrap(self.k_latent_head)))]
        return self._ema_pairs

    @staticmethod
    def _unwrap(m):
        return m.module if hasattr(m, "module") else m

    def forward(self, im_q, im_k=None, Ts=None, Tt=None):
        """Stage-1 forward (model_2D.py:215-305), see arco_amd.stage1.isd_forward; eval mode: (outputs, latent)."""
        return stage1.isd_forward(self, im_q, im_k, Ts, Tt)

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys, queue, queue_ptr):
        stage1.dequeue_and_enqueue(self.K, keys, queue, queue_ptr)

    @torch.no_grad()
    def data_parallel(self):
        """The reference wraps sub-modules in nn.DataParallel (model_2D.py:188-198).  On MI355X
        scaling is one process per GPU over RCCL (arco_amd.dist); inside one process this is a no-op."""
        return self


from .stage1 import get_shuffle_ids, compute_logits   # noqa: E402,F401  (module-level names of the reference's model_2D.py)
