"""Drop-in for the reference's code/loss_helper.py - the 5-D (B,C,H,W,D) = 3-D volume
version of the contrastive loss that train_arco_3d.py imports (train_arco_3d.py:22).
Same public names and signatures (loss_helper.py:142,165,213,250,317,442); the device
implementation is rank-generic and shared with loss_helper_3d; the other names of `from loss_helper import *` likewise."""
from .loss_helper_3d import *  # noqa: F401,F403
from .loss_helper_3d import __all__  # noqa: F401
