"""Drop-in for the reference's code/dataloaders/dataset_withAug.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from dataloaders.dataset_withAug import RandomColorJitter, RandomNoise`) binds the MI355X implementation - every name is re-exported from `arco_amd.dataloaders.dataset_withAug`."""
import _arco_root  # noqa: F401
from arco_amd.dataloaders.dataset_withAug import *  # noqa: F401,F403
