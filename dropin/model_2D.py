"""Drop-in for the reference's code/model_2D.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from model_2D import *`) binds the MI355X implementation - every name is re-exported from `arco_amd.model_2D`."""
import _arco_root  # noqa: F401
from arco_amd.model_2D import *  # noqa: F401,F403
