"""Drop-in for the reference's code/adv_morph.py: put `dropin/` ahead of `code/` on PYTHONPATH and the reference's own import
statement (`from adv_morph import AdvMorph`) binds the MI355X implementation - every name is re-exported from `arco_amd.adv_morph`."""
import _arco_root  # noqa: F401
from arco_amd.adv_morph import *  # noqa: F401,F403
