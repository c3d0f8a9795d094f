"""`import cagym` alias of the package directory `gym-exploration-2d_amd/` (whose name is not a
valid Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("gym-exploration-2d_amd")
sys.modules[__name__] = _pkg
