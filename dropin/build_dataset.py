"""Drop-in for the reference's code/build_dataset.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from build_dataset import BaseDataSetsWithIndex, ...`) binds the MI355X implementation - every name is re-exported from `arco_amd.build_dataset`."""
import _arco_root  # noqa: F401
from arco_amd.build_dataset import *  # noqa: F401,F403
