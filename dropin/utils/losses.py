"""Drop-in for the reference's code/utils/losses.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from utils import losses`) binds the MI355X implementation - every name is re-exported from `arco_amd.utils.losses`."""
import _arco_root  # noqa: F401
from arco_amd.utils.losses import *  # noqa: F401,F403
