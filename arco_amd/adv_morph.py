"""AdvMorph on MI355X - the random diffeomorphic warp that the reference's batch_transform applies to the unlabeled
stream with probability 0.5 (code/augment.py:271-279, code/adv_morph.py:310-580).  Same class name, config dict and
methods on the path the trainer uses (init_parameters, forward, get_deformation_displacement_field, transform,
unit_normalize); the adversarial optimisation methods (train / optimize_parameters, never called by the trainers) are
not provided.

Everything runs as HIP kernels on channels-last 2-channel fields [B, H, W, 2]: 7x7 Gaussian smoothing of the velocity
field (sigma 1; the reference's kernel_size 3 is raised to 2*int(3.5*sigma)+1 = 7 by get_gaussian_kernel), bilinear resize
to the image size, scaling-and-squaring integration (8 self-compositions = 8 border-padded grid samples of the field by
itself), composition with the identity grid, a second smoothing, clamp, and the final bilinear warp of the images.
The velocity field is drawn on the DEVICE generator like the reference (`torch.rand(..., device=cuda)`), so the torch CPU
generator - which defines the bit-exact sampler sequence - is not touched."""
import ctypes
import math

import torch

from . import _lib as L


def _gaussian_weights(kernel_size, sigma):
    """get_gaussian_kernel (adv_morph.py:460-497): float32 arithmetic of the reference, on the host."""
    if kernel_size < 2 * int(3.5 * sigma) + 1:
        kernel_size = 2 * int(3.5 * sigma) + 1
    x_coord = torch.arange(kernel_size)
    x_grid = x_coord.repeat(kernel_size).view(kernel_size, kernel_size)
    y_grid = x_grid.t()
    xy_grid = torch.stack([x_grid, y_grid], dim=-1).float()
    mean = (kernel_size - 1) / 2.
    variance = sigma ** 2.
    k = (1. / (2. * math.pi * variance)) * torch.exp(-torch.sum((xy_grid - mean) ** 2., dim=-1) / (2 * variance))
    k = k / torch.sum(k)
    return kernel_size, k.contiguous()


class AdvMorph:
    def __init__(self, config_dict={'epsilon': 1.5, 'data_size': [10, 1, 8, 8], 'vector_size': [4, 4], 'interpolator_mode': 'bilinear'},
                 power_iteration=False, use_gpu=True, debug=False):
        self.config_dict = config_dict
        self.use_gpu, self.debug, self.power_iteration = use_gpu, debug, power_iteration
        self.device = torch.device('cuda')
        self.is_training = False
        self.param = None
        self.align_corners = True
        self.sigma, self.gaussian_ks, self.smooth_iter, self.num_steps = 1, 3, 1, 8      # adv_morph.py:326-331
        self.integration_type = 'ss'
        self.init_config(config_dict)

    def init_config(self, config_dict):
        self.epsilon = config_dict['epsilon']
        self.xi = 0.5
        self.data_size = config_dict['data_size']
        self.vector_size = config_dict['vector_size']
        self.interpolator_mode = config_dict['interpolator_mode']

    def unit_normalize(self, d, p_type='l2'):
        """adv_morph.py:120-146 (l2: per-sample norm over all elements)."""
        assert p_type == 'l2'
        n = torch.norm(d.reshape(d.shape[0], -1), dim=1).view(-1, *([1] * (d.dim() - 1)))
        return d / (n + 1e-20)

    def init_velocity(self, batch_size, height, width, use_zero=False):
        if use_zero:
            return torch.zeros(batch_size, 2, height, width, device=self.device, dtype=torch.float32)
        duv = torch.rand(batch_size, 2, height, width, device=self.device, dtype=torch.float32) * 2 - 1     # device generator
        return self.unit_normalize(duv)

    def init_parameters(self):
        self.init_config(self.config_dict)
        self.param = self.init_velocity(self.data_size[0], self.vector_size[0], self.vector_size[1])
        return self.param

    def set_parameters(self, param):
        self.param = param.detach().clone()

    def get_parameters(self):
        return self.param

    # ---- fields are channels-last [B, H, W, 2] on the device --------------------------------------------------------
    def _smooth(self, f, B, H, W):
        ks, wts = _gaussian_weights(self.gaussian_ks, self.sigma)
        out = torch.empty_like(f)
        host = (ctypes.c_float * (ks * ks))(*[float(v) for v in wts.reshape(-1)])
        L.call("arco_field_smooth", L.ptr(f), B, H, W, 2, ks, host, L.ptr(out))
        return out

    def _axpb(self, f, alpha, beta, B, H, W, clamp=0, f2=None, gamma=0.0):
        """alpha * f + beta * base_grid + gamma * f2 (f / f2 may be None)."""
        out = torch.empty((B, H, W, 2), dtype=torch.float32, device=self.device)
        L.call("arco_field_axpb", L.ptr(f), float(alpha), float(beta), L.ptr(f2), float(gamma), B, H, W, int(clamp), L.ptr(out))
        return out

    def _compose(self, f1, f2, B, H, W):
        """applyComposition2D(f1, f2): f1 sampled at the positions f2, border padding, align_corners=True."""
        out = torch.empty_like(f2)
        L.call("arco_grid_sample_fwd", L.ptr(f1), 2, B, H, W, 1, 2, L.ptr(f2), H, W, 1, L.ptr(out), 2)
        return out

    @torch.no_grad()
    def get_deformation_displacement_field(self, duv=None):
        """DemonsCompose(duv, base_grid, smooth=True) (adv_morph.py:499-532); returns (grid [B,2,H,W] view, displacement [B,H,W,2])."""
        if duv is None:
            duv = self.param
        B, H, W = int(self.data_size[0]), int(self.data_size[2]), int(self.data_size[3])
        h, w = int(duv.shape[2]), int(duv.shape[3])
        v = duv.to(self.device, torch.float32).permute(0, 2, 3, 1).contiguous()              # [B, h, w, 2]
        v = self._smooth(v, B, h, w)
        big = torch.empty((B, H, W, 2), dtype=torch.float32, device=self.device)
        L.call("arco_field_resize", L.ptr(v), B, h, w, 2, H, W, L.ptr(big))
        phi0 = self._axpb(big, 1.0 / (2.0 ** self.num_steps), 1.0, B, H, W)                    # grid + duv / 2^n
        phi = phi0
        for _ in range(self.num_steps):                                                      # scaling and squaring
            phi = self._compose(phi, phi, B, H, W)
        # reference quirk kept: integrate_by_add (adv_morph.py:246-259) adds in place, so vectorFieldExponentiation2D
        # subtracts the INITIAL phi = grid + duv / 2^n at :294, not the identity grid
        off = self._axpb(phi, 1.0, 0.0, B, H, W, f2=phi0, gamma=-1.0)
        pos = self._axpb(off, 1.0, 1.0, B, H, W)
        base = self._axpb(None, 0.0, 1.0, B, H, W)
        comp = self._compose(base, pos, B, H, W)
        sm = self._smooth(self._axpb(comp, 1.0, -1.0, B, H, W), B, H, W)
        grid = self._axpb(sm, 1.0, 1.0, B, H, W, clamp=1)
        disp = self._axpb(grid, 1.0, -1.0, B, H, W)
        return grid.permute(0, 3, 1, 2), disp

    @torch.no_grad()
    def transform(self, data, deformation_dxy, mode='bilinear', padding_mode='border'):
        """F.grid_sample(data, grid, bilinear, align_corners=True) with the default ZERO padding - the reference's
        `padding_mode` argument is not forwarded to grid_sample (adv_morph.py:573-575)."""
        from ._contrast import rows_view
        grid = deformation_dxy.permute(0, 2, 3, 1).contiguous()
        B, C, H, W = (int(v) for v in data.shape)
        rows, ld = rows_view(data.to(torch.float32))
        out = torch.empty((B, H, W, C), dtype=torch.float32, device=data.device)
        L.call("arco_grid_sample_fwd", L.ptr(rows), ld, B, H, W, 1, C, L.ptr(grid), H, W, 0, L.ptr(out), C)
        return out.permute(0, 3, 1, 2)

    @torch.no_grad()
    def forward(self, data, interpolation_mode=None):
        if self.param is None:
            self.init_parameters()
        dxy, disp = self.get_deformation_displacement_field(duv=self.epsilon * self.param)
        self.displacement = disp
        return self.transform(data, dxy)

    predict_forward = forward

    def get_name(self):
        return 'morph'

    def is_geometric(self):
        return 1
