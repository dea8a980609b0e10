"""Drop-in for the reference's code/networks/net_factory_args.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from networks.net_factory_args import net_factory`) binds the MI355X implementation - every name is re-exported from `arco_amd.networks.net_factory_args`."""
import _arco_root  # noqa: F401
from arco_amd.networks.net_factory_args import *  # noqa: F401,F403
