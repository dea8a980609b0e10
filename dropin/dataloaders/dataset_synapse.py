"""Drop-in for the reference's code/dataloaders/dataset_synapse.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from dataloaders.dataset_synapse import Synapse_dataset`) binds the MI355X implementation - every name is re-exported from `arco_amd.dataloaders.dataset_synapse`."""
import _arco_root  # noqa: F401
from arco_amd.dataloaders.dataset_synapse import *  # noqa: F401,F403
