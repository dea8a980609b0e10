"""Drop-in for the reference's code/augment_3d.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from augment_3d import *`) binds the MI355X implementation - every name is re-exported from `arco_amd.augment_3d`."""
import _arco_root  # noqa: F401
from arco_amd.augment_3d import *  # noqa: F401,F403
