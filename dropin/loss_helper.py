"""Drop-in for the reference's code/loss_helper.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from loss_helper import *`) binds the MI355X implementation - every name is re-exported from `arco_amd.loss_helper`."""
import _arco_root  # noqa: F401
from arco_amd.loss_helper import *  # noqa: F401,F403
# the reference trainers take `nn` from this star import and build q_representation with it (train_arco_2d.py:231-234): torch.nn with
# 1x1 Conv2d / Conv3d on the HIP GEMM path (arco_amd/nn_dropin.py; ARCO_DROPIN_NN=0: plain torch.nn)
from arco_amd.nn_dropin import nn  # noqa: F401,E402
