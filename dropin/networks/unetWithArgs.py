"""Drop-in for the reference's code/networks/unetWithArgs.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from networks.unetWithArgs import UNet`) binds the MI355X implementation - every name is re-exported from `arco_amd.networks.unetWithArgs`."""
import _arco_root  # noqa: F401
from arco_amd.networks.unetWithArgs import *  # noqa: F401,F403
