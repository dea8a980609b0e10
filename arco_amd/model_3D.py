"""Drop-in for the reference's code/model_3D.py on MI355X: FeatureExtractor_3d (:20-63),
create_model_3d (:113-120), ISD_3d (:219-292); FeatureExtractor / create_model are re-exported from
model_2D like the reference file defines both.  Same names, arguments, attributes, state_dict keys."""
import numpy as np  # noqa: F401  (np / F / nn reach the reference trainers through `from model_xD import *`)
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401

from . import ops
from . import optim as _optim      # NOT `optim`: a star import of this module must not shadow the trainer's `import torch.optim as optim`
from .model_2D import FeatureExtractor, create_model  # noqa: F401
from .networks.net_factory_3dArgs import net_factory_3d


class FeatureExtractor_3d(nn.Module):
    def __init__(self, fea_dim=[128, 64, 32, 16, 16], output_dim=128) -> None:
        super().__init__()
        assert len(fea_dim) == 5, 'input_dim is not correct'
        cnt = fea_dim[0]
        self.fea0 = nn.Conv3d(in_channels=cnt, out_channels=cnt, kernel_size=1, bias=False)
        cnt += fea_dim[1]
        self.fea1 = nn.Conv3d(in_channels=cnt, out_channels=cnt, kernel_size=1, bias=False)
        cnt += fea_dim[2]
        self.fea2 = nn.Conv3d(in_channels=cnt, out_channels=cnt, kernel_size=1, bias=False)
        cnt += fea_dim[3]
        self.fea3 = nn.Conv3d(in_channels=cnt, out_channels=cnt, kernel_size=1, bias=False)
        cnt += fea_dim[4]
        self.fea4 = nn.Conv3d(in_channels=cnt, out_channels=output_dim, kernel_size=1, bias=False)

    def forward_lowres2(self, fea_list):
        """Up to fea2 (56x56x40 level): (fea2(x)+x, f3, f4) for the row-sparse head (arco_amd.head).
        Evaluated with the 1x1x1 conv pushed below the trilinear upsample (they commute; residual folded into the
        weights, W' = W + I):  x <- up(W'[:, :c] . x) + W'[:, c:] . f_i  - see model_2D.FeatureExtractor.forward_lowres2;
        the wide block of W' runs on 8x fewer voxels."""
        f = [ops.to_channels_last(t) for t in fea_list]
        x = ops.conv(f[0], self.fea0.weight, None, residual=True)
        for i, fea in enumerate((self.fea1, self.fea2), start=1):
            c = int(x.shape[1])
            n = c + int(f[i].shape[1])
            w_lo, w_hi = ops.fold_residual(fea.weight, c)                    # (W + I)[:, :c], (W + I)[:, c:] in one launch
            lo = ops.conv(x, w_lo)
            x = ops.conv_upres(f[i], w_hi, lo)                                   # one launch: the upsample lives in the GEMM's epilogue
        return x, f[3], f[4]

    def forward_lowres1(self, fea_list):
        """Up to fea1 (the 28x28x20 level): (fea1(x)+x, f2, f3, f4) for the three-level row-sparse head (arco_amd.head.LazyHead3dL3Fn):
        fea2's 224-channel map at 56x56x40 - 450 MB at the LA size, of which a step reads ~1 % of the rows - is not evaluated densely."""
        f = [ops.to_channels_last(t) for t in fea_list]
        x = ops.conv(f[0], self.fea0.weight, None, residual=True)
        c = int(x.shape[1])
        w_lo, w_hi = ops.fold_residual(self.fea1.weight, c)
        x = ops.conv_upres(f[1], w_hi, ops.conv(x, w_lo))
        return x, f[2], f[3], f[4]

    def forward_reference_order(self, fea_list):
        """model_3D.py:43-58 literally: upsample, concatenate, convolve at every level.  Kept as the comparison target of the tests."""
        f = [ops.to_channels_last(t) for t in fea_list]
        x = ops.conv(f[0], self.fea0.weight, None, residual=True)            # fea0(f0) + f0
        for i, fea in enumerate((self.fea1, self.fea2, self.fea3, self.fea4), start=1):
            x = ops.trilinear(x, f[i].shape[-3:])
            x = torch.cat((x, f[i]), dim=1)
            x = ops.conv(x, fea.weight, None, residual=(i < 4))              # fea_i(x) + x ; fea4(x)
        return x

    def forward(self, fea_list):
        """The dense representation (model_3D.py:43-58) with every 1x1x1 convolution below its trilinear upsample, as forward_lowres2
        does for the row-sparse path and model_2D.FeatureExtractor.forward for the 2-D maps: no concatenation is materialised and the
        wide block of each level runs on 8x fewer voxels.  Values differ from the reference order by fp32 rounding only."""
        f = [ops.to_channels_last(t) for t in fea_list]
        x = ops.conv(f[0], self.fea0.weight, None, residual=True)
        for i, fea in enumerate((self.fea1, self.fea2, self.fea3), start=1):
            c = int(x.shape[1])
            w_lo, w_hi = ops.fold_residual(fea.weight, c)                        # (W + I)[:, :c], (W + I)[:, c:]
            x = ops.conv_upres(f[i], w_hi, ops.conv(x, w_lo))
        c = int(x.shape[1])
        w = self.fea4.weight                                                      # no residual at the last level
        lo = ops.conv(x, w[:, :c].contiguous())
        if tuple(f[4].shape[2:]) == tuple(lo.shape[2:]):                          # (f3 and f4 share the full resolution: the resize is the identity)
            return ops.boundary(ops.conv(f[4], w[:, c:].contiguous(), None, residual=lo))
        return ops.boundary(ops.conv_upres(f[4], w[:, c:].contiguous(), lo))


def create_model_3d(ema=False, num_classes=4):
    model = net_factory_3d(net_type="vnet", in_chns=1, class_num=num_classes)
    if ema:
        for param in model.parameters():
            param.detach_()
    return model


class ProjectionHead_3d(nn.Module):
    def __init__(self, dim_in=4, proj_dim=4, output_pooling_size=16, proj='convmlp'):
        super(ProjectionHead_3d, self).__init__()
        if proj == 'linear':
            self.proj = nn.Conv3d(dim_in, proj_dim, kernel_size=1)
        elif proj == 'convmlp':
            self.proj = nn.Sequential(nn.AdaptiveAvgPool3d(output_pooling_size),
                                      nn.Conv3d(dim_in, dim_in * 2, kernel_size=1),
                                      nn.Conv3d(dim_in * 2, proj_dim, kernel_size=1))


class MLP_3d(nn.Module):
    def __init__(self, input_channels=256, num_class=128, pooling_size=1):
        super().__init__()
        self.gap = nn.AdaptiveAvgPool3d(pooling_size)
        self.f1 = nn.Linear(input_channels * pooling_size ** 2, input_channels)
        self.f2 = nn.Linear(input_channels, num_class)


class ISD_3d(nn.Module):
    def __init__(self, K=48, m=0.99, Ts=0.1, Tt=0.01, num_classes=4, train_encoder=True, train_decoder=True,
                 latent_pooling_size=1, latent_feature_size=128, output_pooling_size=4, patch_size=64):
        super(ISD_3d, self).__init__()
        self.K, self.m, self.Ts, self.Tt = K, m, Ts, Tt
        self.num_classes = num_classes
        self.patch_size = patch_size
        self.latent_feature_size = latent_feature_size
        self.model = create_model_3d(num_classes=num_classes)
        self.ema_model = create_model_3d(ema=True, num_classes=num_classes)
        self.k_latent_head = MLP_3d(128, self.latent_feature_size, latent_pooling_size)
        self.q_latent_head = MLP_3d(128, self.latent_feature_size, latent_pooling_size)
        self.latent_predictor = nn.Sequential(nn.Linear(self.latent_feature_size, self.latent_feature_size),
                                              nn.Linear(self.latent_feature_size, self.latent_feature_size))
        self.k_outputs_head = ProjectionHead_3d(num_classes, num_classes, output_pooling_size)
        self.q_outputs_head = ProjectionHead_3d(num_classes, num_classes, output_pooling_size)
        self.outputs_predictor = nn.Sequential(nn.Conv3d(num_classes, num_classes, kernel_size=1),
                                               nn.Conv3d(num_classes, num_classes, kernel_size=1))
        for param_q, param_k in zip(self.model.parameters(), self.ema_model.parameters()):
            param_k.data.copy_(param_q.data)
            param_k.requires_grad = False
        self.register_buffer('queue', nn.functional.normalize(torch.randn(self.K, self.latent_feature_size), dim=-1))
        self.register_buffer('queue_mask', nn.functional.normalize(
            torch.randn(self.K, 700, num_classes * output_pooling_size ** 3), dim=-1))
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long))
        self.register_buffer('mask_queue_ptr', torch.zeros(1, dtype=torch.long))
        self._ema_pairs = None

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        """model_3D.py:267-274: k = m*k + (1-m)*q over parameters() of the net and the two head pairs."""
        self._ensure_ema_pairs()
        for pair in self._ema_pairs:
            pair.update(self.m)

    def _ensure_ema_pairs(self):
        if self._ema_pairs is None:
            self._ema_pairs = [_optim.EmaPair(q, k) for q, k in (
                (self.model, self.ema_model), (self.q_outputs_head, self.k_outputs_head),
                (self.q_latent_head, self.k_latent_head))]
        return self._ema_pairs

    @torch.no_grad()
    def data_parallel(self):
        """nn.DataParallel wrapping of the reference (model_3D.py:284-292) is replaced by one process per GPU."""
        return self

    def forward(self, im_q, im_k=None, Ts=None, Tt=None):
        """Stage-1 forward (model_3D.py:309-403), see arco_amd.stage1.isd_forward; eval mode: (outputs, latent)."""
        from . import stage1
        return stage1.isd_forward(self, im_q, im_k, Ts, Tt)

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys, queue, queue_ptr):
        from . import stage1
        stage1.dequeue_and_enqueue(self.K, keys, queue, queue_ptr)
