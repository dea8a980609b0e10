"""Drop-in for the reference's code/test_util.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from test_util import test_single_case`) binds the MI355X implementation - every name is re-exported from `arco_amd.test_util`."""
import _arco_root  # noqa: F401
from arco_amd.test_util import *  # noqa: F401,F403
