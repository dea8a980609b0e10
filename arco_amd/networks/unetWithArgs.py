"""2-D U-Net of the ARCO hot path on MI355X - drop-in for the reference's
code/networks/unetWithArgs.py (ConvBlock :31-47, DownBlock :50-61, UpBlock :64-85,
Encoder :88-116, Decoder :119-158, UNet :309-348).

Same class names, constructor arguments, return values and state_dict keys/shapes
(the nn.Conv2d / nn.BatchNorm2d children are kept as parameter containers so that
initialisation and checkpoints are identical); the forward pass never calls them -
it runs the hand-written HIP kernels of arco_amd.ops on channels-last activations:
conv3x3 (fp32 MFMA implicit GEMM, BN partial statistics fused in the epilogue) ->
BN finalize -> one fused BN-apply + LeakyReLU + dropout pass.
"""
import torch
import torch.nn as nn

from .. import ops


def _bn_eval_stats(bn):
    return bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)


def _stage(x, conv, bn, act, drop_p, training, drop_mode=1, cat_room=0, pool=False):
    slope = getattr(act, "negative_slope", 0.0)
    if training:
        y = ops.conv_bn_act(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                            slope=slope, p=drop_p, drop_mode=drop_mode, momentum=bn.momentum, eps=bn.eps,
                            num_batches_tracked=bn.num_batches_tracked, cat_room=cat_room, pool=pool)
        return y
    assert not pool
    return ops.conv_bn_act_eval(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                slope=slope, eps=bn.eps)


class ConvBlock(nn.Module):
    """two convolution layers with batch norm and leaky relu (unetWithArgs.py:31-47)"""

    def __init__(self, in_channels, out_channels, dropout_p):
        super(ConvBlock, self).__init__()
        self.conv_conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(out_channels),
            nn.LeakyReLU(),
            nn.Dropout(dropout_p),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(out_channels),
            nn.LeakyReLU(),
        )

    cat_room = 0      # Encoder sets it on the blocks whose output is a skip connection (see ops.upcat)

    def forward(self, x, pool=False):
        """pool=True (train mode): (block output, its 2x2 max-pool) - the pooling rides on the last BN / activation pass."""
        s = self.conv_conv
        if self.training:
            # both stages as one node: the first stage's activation is formed in the second convolution's loader (ops.ConvBlockFn)
            y = ops.conv_block(x, s[0], s[1], s[2], s[3].p, s[4], s[5], s[6], cat_room=self.cat_room, pool=pool)
            if y is not None:
                return y
        x = _stage(x, s[0], s[1], s[2], s[3].p, self.training)
        return _stage(x, s[4], s[5], s[6], 0.0, self.training, cat_room=self.cat_room, pool=pool)


class DownBlock(nn.Module):
    """Downsampling followed by ConvBlock (unetWithArgs.py:50-61)"""

    def __init__(self, in_channels, out_channels, dropout_p):
        super(DownBlock, self).__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), ConvBlock(in_channels, out_channels, dropout_p))

    def forward(self, x):
        return self.maxpool_conv[1](ops.maxpool2(x))

    def forward_skip(self, x):
        """(block output, x as the decoder's skip connection): the two gradients of x are summed inside the max-pool backward."""
        pooled, skip = ops.maxpool2_skip(x)
        return self.maxpool_conv[1](pooled), skip


class UpBlock(nn.Module):
    """Upsampling followed by ConvBlock (unetWithArgs.py:64-85).  The reference Decoder never passes `bilinear`, so the default
    bilinear=True path (1x1 conv + x2 align_corners bilinear) is the one on the hot path.  bilinear=False (:76-77,
    `nn.ConvTranspose2d(in_channels1, in_channels2, kernel_size=2, stride=2)`) is unreachable from UNet but part of the module's
    surface: it runs as the GEMM form of a k2 s2 transposed conv (one 1x1-conv launch over [4 co][ci] weights, as the V-Net's
    UpsamplingDeconvBlock does in 3-D) followed by the pixel shuffle."""

    def __init__(self, in_channels1, in_channels2, out_channels, dropout_p, bilinear=True):
        super(UpBlock, self).__init__()
        self.bilinear = bilinear
        if bilinear:
            self.conv1x1 = nn.Conv2d(in_channels1, in_channels2, kernel_size=1)
            self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        else:
            self.up = nn.ConvTranspose2d(in_channels1, in_channels2, kernel_size=2, stride=2)
        self.conv = ConvBlock(in_channels2 * 2, out_channels, dropout_p)

    def _deconv(self, x1):
        w, b = self.up.weight, self.up.bias                         # [ci][co][2][2]
        ci, co = int(w.shape[0]), int(w.shape[1])
        w2 = w.permute(2, 3, 1, 0).reshape(4 * co, ci, 1, 1)        # rows (dy, dx, co)
        y = ops.conv(x1, w2, b.repeat(4) if b is not None else None)            # [N, 4 co, H, W], channels-last rows
        n, _, h, wd = y.shape
        y = y.permute(0, 2, 3, 1).reshape(n, h, wd, 2, 2, co).permute(0, 1, 3, 2, 4, 5).reshape(n, 2 * h, 2 * wd, co)
        return y.permute(0, 3, 1, 2)                                # logical NCHW over channels-last memory

    def forward(self, x1, x2):
        if not self.bilinear:
            return self.conv(torch.cat([x2, self._deconv(x1)], dim=1))
        x1 = ops.conv(x1, self.conv1x1.weight, self.conv1x1.bias)
        if (x1.shape[2] * 2, x1.shape[3] * 2) == tuple(x2.shape[2:]):
            x = ops.upcat(x1, x2)                       # cat([x2, up(x1)]) written in place behind the skip
        else:
            x1 = ops.bilinear(x1, (x1.shape[2] * 2, x1.shape[3] * 2))
            x = torch.cat([x2, x1], dim=1)
        return self.conv(x)


class Encoder(nn.Module):
    def __init__(self, params):
        super(Encoder, self).__init__()
        self.params = params
        self.in_chns = self.params['in_chns']
        self.ft_chns = self.params['feature_chns']
        self.n_class = self.params['class_num']
        self.bilinear = self.params['bilinear']
        self.dropout = self.params['dropout']
        assert (len(self.ft_chns) == 5)
        self.in_conv = ConvBlock(self.in_chns, self.ft_chns[0], self.dropout[0])
        self.down1 = DownBlock(self.ft_chns[0], self.ft_chns[1], self.dropout[1])
        self.down2 = DownBlock(self.ft_chns[1], self.ft_chns[2], self.dropout[2])
        self.down3 = DownBlock(self.ft_chns[2], self.ft_chns[3], self.dropout[3])
        self.down4 = DownBlock(self.ft_chns[3], self.ft_chns[4], self.dropout[4])
        # x0..x3 are concatenated with an equally wide upsampled tensor in the decoder: write them with room for it
        self.in_conv.cat_room = self.ft_chns[0]
        for blk, c in ((self.down1, self.ft_chns[1]), (self.down2, self.ft_chns[2]), (self.down3, self.ft_chns[3])):
            blk.maxpool_conv[1].cat_room = c

    def forward(self, x):
        x = ops.to_channels_last(x.to(torch.float32))
        if self.training and ops.POOL_FUSE:     # nn.MaxPool2d of each DownBlock fused into the block before it; the two
            x0, p = self.in_conv(x, pool=True)  # gradients of x_i (skip + pooling) are summed in that block's backward
            x1, p = self.down1.maxpool_conv[1](p, pool=True)
            x2, p = self.down2.maxpool_conv[1](p, pool=True)
            x3, p = self.down3.maxpool_conv[1](p, pool=True)
            x4 = self.down4.maxpool_conv[1](p)
            return [x0, x1, x2, x3, x4]
        x0 = self.in_conv(x)
        if torch.is_grad_enabled() and x0.requires_grad:       # training: skip gradients fused into the pooling backward
            x1, x0 = self.down1.forward_skip(x0)
            x2, x1 = self.down2.forward_skip(x1)
            x3, x2 = self.down3.forward_skip(x2)
            x4, x3 = self.down4.forward_skip(x3)
            return [x0, x1, x2, x3, x4]
        x1 = self.down1(x0)
        x2 = self.down2(x1)
        x3 = self.down3(x2)
        x4 = self.down4(x3)
        return [x0, x1, x2, x3, x4]


class Decoder(nn.Module):
    def __init__(self, params):
        super(Decoder, self).__init__()
        self.params = params
        self.in_chns = self.params['in_chns']
        self.ft_chns = self.params['feature_chns']
        self.n_class = self.params['class_num']
        self.bilinear = self.params['bilinear']
        assert (len(self.ft_chns) == 5)
        self.up1 = UpBlock(self.ft_chns[4], self.ft_chns[3], self.ft_chns[3], dropout_p=0.0)
        self.up2 = UpBlock(self.ft_chns[3], self.ft_chns[2], self.ft_chns[2], dropout_p=0.0)
        self.up3 = UpBlock(self.ft_chns[2], self.ft_chns[1], self.ft_chns[1], dropout_p=0.0)
        self.up4 = UpBlock(self.ft_chns[1], self.ft_chns[0], self.ft_chns[0], dropout_p=0.0)
        self.out_conv = nn.Conv2d(self.ft_chns[0], self.n_class, kernel_size=3, padding=1)

    def forward(self, feature):
        x0, x1, x2, x3, x4 = feature
        feature_map = [x4]
        x = self.up1(x4, x3)
        feature_map.append(x)
        x = self.up2(x, x2)
        feature_map.append(x)
        x = self.up3(x, x1)
        feature_map.append(x)
        x = self.up4(x, x0)
        feature_map.append(x)
        output = ops.conv(x, self.out_conv.weight, self.out_conv.bias)
        return output, feature_map


class UNet(nn.Module):
    def __init__(self, in_chns, class_num, train_encoder=True, train_decoder=True, unfreeze_seg=True):
        super(UNet, self).__init__()
        params = {'in_chns': in_chns,
                  'feature_chns': [16, 32, 64, 128, 256],
                  'dropout': [0.05, 0.1, 0.2, 0.3, 0.5],
                  'class_num': class_num,
                  'bilinear': False,
                  'acti_func': 'relu'}
        self.params = params
        self.encoder = Encoder(params)
        self.decoder = Decoder(params)
        self.train_encoder = train_encoder
        self.train_decoder = train_decoder
        if not train_encoder:
            for p in self.encoder.parameters():
                p.requires_grad = False
                p.detach_()
        if not train_decoder:
            for p in self.decoder.parameters():
                p.requires_grad = False
                p.detach_()
        if not unfreeze_seg:
            keep = {id(p) for p in self.decoder.out_conv.parameters()}
            for p in list(self.encoder.parameters()) + list(self.decoder.parameters()):
                if id(p) not in keep:
                    p.requires_grad = False
                    p.detach_()

    def forward(self, x):
        feature = self.encoder(x)
        output, feature_map = self.decoder(feature)
        return output, feature[-1], feature_map
