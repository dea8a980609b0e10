"""Drop-in for the reference's code/utils/ramps.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from utils import ramps`) binds the MI355X implementation - every name is re-exported from `arco_amd.utils.ramps`."""
import _arco_root  # noqa: F401
from arco_amd.utils.ramps import *  # noqa: F401,F403
