"""Drop-in for the reference's code/utils/metrics.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from utils import metrics`) binds the MI355X implementation - every name is re-exported from `arco_amd.utils.metrics`."""
import _arco_root  # noqa: F401
from arco_amd.utils.metrics import *  # noqa: F401,F403
