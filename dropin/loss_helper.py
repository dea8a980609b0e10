"""Drop-in for the reference's code/loss_helper.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from loss_helper import *`) binds the MI355X implementation - every name is re-exported from `arco_amd.loss_helper`."""
import _arco_root  # noqa: F401
from arco_amd.loss_helper import *  # noqa: F401,F403
