"""Drop-in for the reference's code/dataloaders/la_heart.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from dataloaders.la_heart import LAHeartWithIndex, ...`) binds the MI355X implementation - every name is re-exported from `arco_amd.dataloaders.la_heart`."""
import _arco_root  # noqa: F401
from arco_amd.dataloaders.la_heart import *  # noqa: F401,F403
