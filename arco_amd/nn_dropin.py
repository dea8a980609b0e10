"""The `nn` the reference trainers pick up through their star imports.

`train_arco_2d.py` / `train_arco_3d.py` never import `torch.nn`: the name `nn` reaches them through `from model_2D import *` /
`from loss_helper_3d import *` (the last star import wins), and they use it for `nn.functional.normalize` (:158), `nn.KLDivLoss` (:419),
`nn.Sequential` and - the one place where it builds part of the hot path - q_representation (`train_arco_2d.py:231-234`:
`nn.Sequential(nn.Conv2d(496, 496, kernel_size=1, bias=False), nn.Conv2d(496, 496, kernel_size=1, bias=False))`, `train_arco_3d.py:206-209`
with `nn.Conv3d(16, 16, 1)`).  The `dropin/` modules therefore export THIS namespace under that name: `torch.nn` unchanged except that
`Conv2d` / `Conv3d` are subclasses whose forward sends a plain 1x1 convolution of an fp32 GPU tensor through `arco_amd.ops.conv` (the
split-bf16 GEMM kernels; backward on the non-zero rows of the incoming gradient when the loss touched few rows) and everything else
through `torch.nn`'s own forward.  Same class name, parameters, `state_dict` keys, `isinstance(m, torch.nn.Conv2d)`; fp32-accurate
results (torch's convolution differs from it by summation order, as two torch backends do).  `ARCO_DROPIN_NN=0`: plain `torch.nn`.
The outputs are channels-last in memory (logical NCHW / NCDHW) and are handed out as `ops.BoundaryTensor`: the reference's
`rep_u.view(rep_u.shape[0], -1)` (`train_arco_2d.py:127`) works on them and flattens in (c, h, w) order as on torch's own convolution
output (it copies where torch's would alias); `tests/test_dropin_user_gpu.py::test_literal_reference_statements_on_the_dropin_nn`.
Not supported on the accelerated route: double backward (`create_graph=True` through the convolution) - `ops.ConvFn` is a first-order
autograd function; a trainer that needs it sets `ARCO_DROPIN_NN=0`.
At the headline size a reference-style user's step spends 32 of 76 ms in torch's own 1x1 convolutions of q_representation (11 forward,
21 in their dense backward); through this namespace: see `bench.py` sub-record `dropin_user_step`."""
import os
import types

import torch

from . import ops

_tnn = torch.nn


def _plain_1x1(m, x, nd):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == nd + 2 and m.weight.dtype == torch.float32
            and tuple(m.kernel_size) == (1,) * nd and tuple(m.stride) == (1,) * nd and tuple(m.dilation) == (1,) * nd and m.groups == 1
            and m.padding_mode == "zeros" and (m.padding == "valid" or tuple(m.padding) == (0,) * nd)
            and m.in_channels % 16 == 0 and m.out_channels % 16 == 0)


class Conv2d(_tnn.Conv2d):
    __doc__ = _tnn.Conv2d.__doc__

    def forward(self, x):
        if _plain_1x1(self, x, 2):
            return ops.boundary(ops.conv(ops.to_channels_last(x), self.weight, self.bias))
        return super().forward(x)


class Conv3d(_tnn.Conv3d):
    __doc__ = _tnn.Conv3d.__doc__

    def forward(self, x):
        if _plain_1x1(self, x, 3):
            return ops.boundary(ops.conv(ops.to_channels_last(x), self.weight, self.bias))
        return super().forward(x)


Conv2d.__name__ = Conv2d.__qualname__ = "Conv2d"
Conv3d.__name__ = Conv3d.__qualname__ = "Conv3d"


class _Namespace(types.ModuleType):
    """torch.nn with the two classes above; every other attribute (functional, init, Module, Sequential, ...) is torch.nn's own."""

    def __getattr__(self, name):
        return getattr(_tnn, name)

    def __dir__(self):
        return sorted(set(dir(_tnn)) | {"Conv2d", "Conv3d"})


if os.environ.get("ARCO_DROPIN_NN", "1") != "0":
    nn = _Namespace("torch.nn")
    nn.__doc__ = _tnn.__doc__
    nn.Conv2d, nn.Conv3d = Conv2d, Conv3d
else:
    nn = _tnn
