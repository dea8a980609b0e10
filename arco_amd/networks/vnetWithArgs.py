"""3-D V-Net of the ARCO hot path on MI355X - drop-in for the reference's
code/networks/vnetWithArgs.py (ConvBlock :5-31, DownsamplingConvBlock :67-91,
UpsamplingDeconvBlock :94-118, VNet :145-252): every `normalization` the reference's blocks accept - 'batchnorm' (the one
net_factory_3d builds, net_factory_3dArgs.py:16-18: statistics fused into the conv epilogue), 'groupnorm'
(nn.GroupNorm(16, C)), 'instancenorm', 'none' (ops.GnActFn: one statistics pass + the apply pass).

Same class names, constructor arguments, return values and state_dict keys (the nn.Conv3d /
nn.BatchNorm3d children are parameter containers).  Forward on channels-last-3d activations:
  3x3x3 conv  = fp32 MFMA implicit GEMM, the depth tap is an outer loop over shifted planes,
                BN partial statistics fused in the epilogue, then one BN+ReLU apply pass;
  k2 s2 conv  = space-to-depth (8*C channels) + 1x1x1 GEMM (+fused BN statistics);
  k2 s2 convT = 1x1x1 GEMM to 8*C channels + depth-to-space, then BN+ReLU.
With ops.ACT_HALF (train_arco_3d --act_dtype f16; BASELINE.json configs[4]) the first layer writes f16 and the whole body runs
on f16 activations and activation gradients (csrc/conv_h.hip, the *_h entry points; 'batchnorm' only); the logits and the
feature maps are handed out as fp32.
"""
import torch
from torch import nn

from .. import ops


NORMS = ('batchnorm', 'groupnorm', 'instancenorm', 'none')


def _norm_layer(normalization, c):
    """The parameter container the reference appends after a conv (vnetWithArgs.py:17-24): same class -> same state_dict keys."""
    if normalization == 'batchnorm':
        return nn.BatchNorm3d(c)
    if normalization == 'groupnorm':
        return nn.GroupNorm(num_groups=16, num_channels=c)
    if normalization == 'instancenorm':
        return nn.InstanceNorm3d(c)
    assert normalization == 'none', normalization
    return None


def _norm_act(z, norm, training):
    if z.dtype == torch.float16:
        raise RuntimeError("arco_amd: f16 activation storage is built for normalization='batchnorm' (the V-Net net_factory_3d makes)")
    return _norm_act32(z, norm, training)


def _norm_act32(z, norm, training):
    """ReLU(norm(z)) for a pre-activation z that did not get its statistics from the conv epilogue: GroupNorm(16, C) /
    InstanceNorm3d / nothing (vnetWithArgs.py:19-24).  Batch-independent statistics: the same in train and eval mode."""
    if norm is None:
        return ops.act(z, 0.0)
    if isinstance(norm, nn.GroupNorm):
        return ops.gn_act(z, norm.weight, norm.bias, norm.num_groups, 0.0, norm.eps)
    if isinstance(norm, nn.InstanceNorm3d):
        return ops.in_act(z, 0.0, norm.eps)
    raise AssertionError(type(norm))


def _bn_stage(x, conv, bn, training):
    if training:
        y = ops.conv_bn_act(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                            slope=0.0, p=0.0, momentum=bn.momentum, eps=bn.eps,
                            num_batches_tracked=bn.num_batches_tracked)
        return y
    return ops.conv_bn_act_eval(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                slope=0.0, eps=bn.eps)


def _stage(x, conv, norm, training):
    """conv -> norm -> ReLU.  BatchNorm (the variant net_factory_3d builds): statistics fused into the conv epilogue;
    the other variants: conv, then one statistics pass + the apply pass."""
    if isinstance(norm, nn.BatchNorm3d):
        return _bn_stage(x, conv, norm, training)
    return _norm_act(ops.conv(x, conv.weight, conv.bias), norm, training)


class ConvBlock(nn.Module):
    def __init__(self, n_stages, n_filters_in, n_filters_out, normalization='none'):
        super(ConvBlock, self).__init__()
        assert normalization in NORMS
        ops_ = []
        for i in range(n_stages):
            ops_.append(nn.Conv3d(n_filters_in if i == 0 else n_filters_out, n_filters_out, 3, padding=1))
            norm = _norm_layer(normalization, n_filters_out)
            if norm is not None:
                ops_.append(norm)
            ops_.append(nn.ReLU(inplace=True))
        self.conv = nn.Sequential(*ops_)
        self.n_stages = n_stages
        self.per = 2 if normalization == 'none' else 3

    def forward(self, x):
        if self.training and self.per == 3 and self.n_stages >= 2 and isinstance(self.conv[1], nn.BatchNorm3d):
            stages = [(self.conv[3 * i], self.conv[3 * i + 1]) for i in range(self.n_stages)]
            if ops.block3d_fusable(x, stages):       # gradient-free pass: only the last stage's activation is written
                return ops.conv_block3d_nograd(x, stages)
        for i in range(self.n_stages):
            conv = self.conv[self.per * i]
            x = _stage(x, conv, self.conv[self.per * i + 1] if self.per == 3 else None, self.training)
        return x


class DownsamplingConvBlock(nn.Module):
    def __init__(self, n_filters_in, n_filters_out, stride=2, normalization='none'):
        super(DownsamplingConvBlock, self).__init__()
        assert normalization in NORMS and stride == 2
        layers = [nn.Conv3d(n_filters_in, n_filters_out, stride, padding=0, stride=stride)]
        norm = _norm_layer(normalization, n_filters_out)
        if norm is not None:
            layers.append(norm)
        layers.append(nn.ReLU(inplace=True))
        self.conv = nn.Sequential(*layers)
        self.has_norm = norm is not None

    def forward(self, x, xs=None):
        """xs: space_to_depth3(x) when the caller already made it (VNet.encoder: ops.s2d_skip pairs it with the skip alias of x)."""
        conv, bn = self.conv[0], (self.conv[1] if self.has_norm else None)
        w2, bias = ops.gemm_weight(conv)                                          # [co][(dx,dy,dz), ci]
        if xs is None:
            xs = ops.space_to_depth3(x)
        if not isinstance(bn, nn.BatchNorm3d):
            return _norm_act(ops.conv(xs, w2, bias), bn, self.training)
        if self.training:
            return ops.conv_bn_act(xs, w2, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, slope=0.0,
                                   p=0.0, momentum=bn.momentum, eps=bn.eps, num_batches_tracked=bn.num_batches_tracked)
        return ops.conv_bn_act_eval(xs, w2, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, slope=0.0,
                                    eps=bn.eps)


class UpsamplingDeconvBlock(nn.Module):
    def __init__(self, n_filters_in, n_filters_out, stride=2, normalization='none'):
        super(UpsamplingDeconvBlock, self).__init__()
        assert normalization in NORMS and stride == 2
        layers = [nn.ConvTranspose3d(n_filters_in, n_filters_out, stride, padding=0, stride=stride)]
        norm = _norm_layer(normalization, n_filters_out)
        if norm is not None:
            layers.append(norm)
        layers.append(nn.ReLU(inplace=True))
        self.conv = nn.Sequential(*layers)
        self.has_norm = norm is not None

    def forward(self, x, skip=None):
        """skip: the decoder's `block_x_up(x) + skip` (vnetWithArgs.py:224-236) with the addition folded into the BatchNorm apply
        pass (train-mode batchnorm); the other variants add afterwards."""
        y = self._forward(x, skip)
        return y if (skip is None or self._fused) else y + skip

    def _forward(self, x, skip):
        self._fused = False
        conv, bn = self.conv[0], (self.conv[1] if self.has_norm else None)
        ci, co = conv.weight.shape[0], conv.weight.shape[1]
        w2, bias8 = ops.gemm_weight(conv)                                         # [(dx,dy,dz), co][ci]
        is_bn = isinstance(bn, nn.BatchNorm3d)
        # train-mode BatchNorm: the bias only shifts its input -> analytically zero gradient.  GroupNorm's sets span several
        # channels (a per-channel shift survives), 'none' has no normalisation: the bias gradient is a real column sum there;
        # InstanceNorm removes it like BatchNorm does.
        zero_bias_grad = (self.training and is_bn) or isinstance(bn, nn.InstanceNorm3d)
        y = ops.conv(x, w2, bias8, bias_grad_zero=zero_bias_grad)
        if is_bn and self.training and ops.D2S_FUSE and co % 4 == 0:
            # depth-to-space folded into the BatchNorm passes (and the skip addition with it): ops.BnActD2sFn
            self._fused = skip is not None and skip.dtype == y.dtype
            return ops.bn_act_d2s(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, slope=0.0, momentum=bn.momentum, eps=bn.eps,
                                  num_batches_tracked=bn.num_batches_tracked, residual=skip if self._fused else None)
        z = ops.depth_to_space3(y)
        if not is_bn:
            return _norm_act(z, bn, self.training)
        if self.training:
            self._fused = skip is not None and skip.dtype == z.dtype
            return ops.bn_act(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, slope=0.0, momentum=bn.momentum,
                              eps=bn.eps, num_batches_tracked=bn.num_batches_tracked, residual=skip if self._fused else None)
        return ops.bn_act_eval(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, slope=0.0, eps=bn.eps)


class VNet(nn.Module):
    def __init__(self, n_channels=3, n_classes=2, n_filters=16, normalization='none', has_dropout=False):
        super(VNet, self).__init__()
        self.has_dropout = has_dropout
        nf = n_filters
        self.block_one = ConvBlock(1, n_channels, nf, normalization=normalization)
        self.block_one_dw = DownsamplingConvBlock(nf, 2 * nf, normalization=normalization)
        self.block_two = ConvBlock(2, nf * 2, nf * 2, normalization=normalization)
        self.block_two_dw = DownsamplingConvBlock(nf * 2, nf * 4, normalization=normalization)
        self.block_three = ConvBlock(3, nf * 4, nf * 4, normalization=normalization)
        self.block_three_dw = DownsamplingConvBlock(nf * 4, nf * 8, normalization=normalization)
        self.block_four = ConvBlock(3, nf * 8, nf * 8, normalization=normalization)
        self.block_four_dw = DownsamplingConvBlock(nf * 8, nf * 16, normalization=normalization)
        self.block_five = ConvBlock(3, nf * 16, nf * 16, normalization=normalization)
        self.block_five_up = UpsamplingDeconvBlock(nf * 16, nf * 8, normalization=normalization)
        self.block_six = ConvBlock(3, nf * 8, nf * 8, normalization=normalization)
        self.block_six_up = UpsamplingDeconvBlock(nf * 8, nf * 4, normalization=normalization)
        self.block_seven = ConvBlock(3, nf * 4, nf * 4, normalization=normalization)
        self.block_seven_up = UpsamplingDeconvBlock(nf * 4, nf * 2, normalization=normalization)
        self.block_eight = ConvBlock(2, nf * 2, nf * 2, normalization=normalization)
        self.block_eight_up = UpsamplingDeconvBlock(nf * 2, nf, normalization=normalization)
        self.block_nine = ConvBlock(1, nf, nf, normalization=normalization)
        self.out_conv = nn.Conv3d(nf, n_classes, 1, padding=0)
        self.dropout = nn.Dropout3d(p=0.5, inplace=False)

    def _drop(self, x):
        if self.has_dropout and self.training:
            return ops.dropout3d(x, self.dropout.p)
        return x

    def encoder(self, input):
        x = ops.to_channels_last(input.to(torch.float32))
        # every level's activation feeds the next DownsamplingConvBlock (through space-to-depth) AND the decoder's skip connection: the
        # two gradients meet inside one kernel (ops.S2dSkipFn); the returned features are the skip aliases
        x1s, x1 = ops.s2d_skip(self.block_one(x))
        x2s, x2 = ops.s2d_skip(self.block_two(self.block_one_dw(x1, x1s)))
        x3s, x3 = ops.s2d_skip(self.block_three(self.block_two_dw(x2, x2s)))
        x4s, x4 = ops.s2d_skip(self.block_four(self.block_three_dw(x3, x3s)))
        x5 = self._drop(self.block_five(self.block_four_dw(x4, x4s)))
        return [x1, x2, x3, x4, x5]

    def decoder(self, features):
        x1, x2, x3, x4, x5 = features
        x5_up = self.block_five_up(x5, x4)
        feature_map = [x5_up]
        x6_up = self.block_six_up(self.block_six(x5_up), x3)
        feature_map.append(x6_up)
        x7_up = self.block_seven_up(self.block_seven(x6_up), x2)
        feature_map.append(x7_up)
        x8_up = self.block_eight_up(self.block_eight(x7_up), x1)
        feature_map.append(x8_up)
        x9 = self.block_nine(x8_up)
        feature_map.append(x9)
        x9 = self._drop(x9)
        out = ops.conv(x9, self.out_conv.weight, self.out_conv.bias)
        # f16 activation storage (ops.ACT_HALF): the logits and the feature maps leave the f16 region as fp32 - heads, losses
        # and samplers are fp32; gradients come back through the same boundary with the loss scale
        if ops.FM_CAST == 'lowres':      # ops.fm_rows_half: the two full-resolution maps stay f16 for the row-sparse heads
            return ops.from_half(out), [ops.from_half(f) for f in feature_map[:-2]] + feature_map[-2:]
        return ops.from_half(out), ([ops.from_half(f) for f in feature_map] if ops.FM_CAST else feature_map)

    def forward(self, input, turnoff_drop=False):
        if turnoff_drop:
            has_dropout = self.has_dropout
            self.has_dropout = False
        features = self.encoder(input)
        out, feature_map = self.decoder(features)
        if turnoff_drop:
            self.has_dropout = has_dropout
        return out, feature_map[0], feature_map
