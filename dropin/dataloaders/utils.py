"""Drop-in for the reference's code/dataloaders/utils.py (`from dataloaders import utils`, imported and unused by the stage-1
trainers) - re-exported from `arco_amd.dataloaders.utils`."""
import _arco_root  # noqa: F401
from arco_amd.dataloaders.utils import *  # noqa: F401,F403
