"""Drop-in for the reference's code/test_2D.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from test_2D import test_single_volume`) binds the MI355X implementation - every name is re-exported from `arco_amd.test_2D`."""
import _arco_root  # noqa: F401
from arco_amd.test_2D import *  # noqa: F401,F403
