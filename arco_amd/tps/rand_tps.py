"""Random similarity + thin-plate-spline warp for the equivariance loss - drop-in for the
reference's code/tps/rand_tps.py `RandTPS` (:82-153) with TPSGridGen (code/tps_stn_pytorch/tps_grid_gen.py:23-71) and
grid_sample (code/tps/grid_sample.py:11-12) behind it.

The random control points are drawn on the host with exactly the reference's generator calls (one torch CPU
`uniform_` of [B,25,2], four numpy uniforms of [B], one python `random.randint`) - they define the warp for a given
seed.  Everything heavy runs on the GPU: the [H*W, 28] target-coordinate representation is built once and stays in
HBM (the reference rebuilds it, a 28x28 inverse included, on the CPU at every reset), a reset uploads 28x2 floats per
image and one kernel writes the [B,H,W,2] grid; `forward` is a bilinear, align_corners=True grid-sample kernel on
channels-last rows."""
import random

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from .._contrast import rows_view


def _u(points, controls):
    # U(r) = 0.5 r^2 log r^2, 0 at r = 0 (tps_grid_gen.py:9-21)
    d = points.view(-1, 1, 2) - controls.view(1, -1, 2)
    d2 = d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1]
    u = 0.5 * d2 * torch.log(d2)
    u[u != u] = 0
    return u


class RandTPS(nn.Module):
    def __init__(self, width, height, batch_size=16, sigma=0.01, border_padding=False, random_mirror=True,
                 random_scale=(0.7, 1.1), mode='affine', device=None):
        super().__init__()
        if mode != 'affine':
            raise NotImplementedError("RandTPS(mode='projective') is not on the ARCO path (train_arco_2d.py:255-261)")
        self.width, self.height, self.batch_size, self.sigma = int(width), int(height), int(batch_size), float(sigma)
        self.random_scale = (1.0 / random_scale[1], 1.0 / random_scale[0])      # applied target -> source (:88)
        self.padding_mode = 'border' if border_padding else 'zeros'
        self.rand_mirror, self.mode = random_mirror, mode
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        ticks = torch.arange(-1.0, 1.00001, 2.0 / 4)
        self.target_control_points = torch.Tensor([(float(a), float(b)) for a in ticks for b in ticks])    # 25 x 2
        tcp = self.target_control_points
        n = tcp.shape[0]
        k = torch.zeros(n + 3, n + 3)                                           # TPSGridGen.__init__ (:25-41)
        k[:n, :n] = _u(tcp, tcp)
        k[:n, -3] = 1
        k[-3, :n] = 1
        k[:n, -2:] = tcp
        k[-2:, :n] = tcp.t()
        self.inverse_kernel = torch.inverse(k)
        H, W = self.height, self.width
        yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        coord = torch.cat([xx.reshape(-1, 1) * 2 / (W - 1) - 1, yy.reshape(-1, 1) * 2 / (H - 1) - 1], dim=1)   # (x, y)
        rep = torch.cat([_u(coord, tcp), torch.ones(H * W, 1), coord], dim=1)   # [HW, 28]  (:43-52)
        self._rep = rep.contiguous().to(self.device)
        self.grid = torch.zeros(self.batch_size, H, W, 2, dtype=torch.float32, device=self.device)
        self.reset_control_points()

    def cuda(self, device=None):                                                # the trainer calls .cuda() (:261)
        return self

    @torch.no_grad()
    def reset_control_points(self):
        B, tcp = self.batch_size, self.target_control_points
        src = tcp.unsqueeze(0).repeat(B, 1, 1)
        src = src + torch.Tensor(src.size()).uniform_(-self.sigma, self.sigma)   # torch CPU generator (:116)
        # generate_transformer_matrices(batch_size, img_sz=2.0, random_scale) (:48-79): numpy generator
        ang = np.random.uniform(size=[B], low=-60, high=60) / 180.0 * np.pi
        sc = np.random.uniform(size=[B], low=self.random_scale[0], high=self.random_scale[1])
        sx = np.random.uniform(size=(B,), low=-0.1, high=0.1).reshape(-1, 1)
        sy = np.random.uniform(size=(B,), low=-0.1, high=0.1).reshape(-1, 1)
        half = np.float32(np.float32(2.0) / 2.0)
        cos_v, sin_v = (sc * np.cos(ang)).reshape(-1, 1), (sc * np.sin(ang)).reshape(-1, 1)
        theta = np.concatenate([cos_v, -sin_v, sx * half, sin_v, cos_v, sy * half], axis=1)
        t = torch.from_numpy(theta.reshape(-1, 2, 3).copy()).type(torch.FloatTensor).transpose(1, 2)
        src = torch.matmul(torch.cat((src, torch.ones(B, src.shape[1], 1)), dim=2), t)
        if self.rand_mirror and random.randint(0, 1):                            # python generator (:136-138)
            src[:, :, 0] = -src[:, :, 0]
        mapping = torch.matmul(self.inverse_kernel, torch.cat([src, torch.zeros(B, 3, 2)], 1)).contiguous()   # [B,28,2]
        mp = mapping.to(self.device, non_blocking=True)
        L.call("arco_tps_grid", L.ptr(self._rep), L.ptr(mp), B, self.height * self.width, int(self._rep.shape[1]),
               L.ptr(self.grid))

    @torch.no_grad()
    def forward(self, x, padding_mode=None, mode='bilinear'):
        """grid_sample(x, self.grid, bilinear, align_corners=True); x [B,C,h,w] (any layout) -> channels-last output."""
        if mode != 'bilinear':
            raise NotImplementedError("only bilinear sampling is on the ARCO path")
        if x.requires_grad:
            raise NotImplementedError("RandTPS.forward is used on detached tensors only (train_arco_2d.py:413-418)")
        pm = self.padding_mode if padding_mode is None else padding_mode
        L.require_gpu(x)
        xr, ld = rows_view(x.to(torch.float32))
        nb, c, h, w = (int(v) for v in x.shape[:4])
        d3 = int(x.shape[4]) if x.dim() == 5 else 1          # 5-D: the same warp on every slice x[..., z] (rand_tps_3d.py:155-165)
        Ho, Wo = int(self.grid.shape[1]), int(self.grid.shape[2])
        if nb != self.batch_size:
            raise RuntimeError(f"RandTPS was built for batch {self.batch_size}, got {nb}")
        if d3 > 1 and (Ho, Wo) != (h, w):
            raise RuntimeError("the slice-wise warp of a volume needs a grid of the volume's own (X, Y) size")
        if d3 > 1:
            y = torch.empty((nb, Ho, Wo, d3, c), dtype=torch.float32, device=x.device).permute(0, 4, 1, 2, 3)
        else:
            y = torch.empty((nb, Ho, Wo, c), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
        L.call("arco_grid_sample_fwd", L.ptr(xr), ld, nb, h, w, d3, c, L.ptr(self.grid), Ho, Wo, 1 if pm == 'border' else 0,
               L.ptr(y), c)
        return y
