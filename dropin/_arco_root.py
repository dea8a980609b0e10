"""Makes `arco_amd` importable from the drop-in modules: the repository root (the parent of this directory) goes on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
