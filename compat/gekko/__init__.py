"""`import gekko` compatibility package: put this directory's parent (`compat/`) on PYTHONPATH and a
script written against GEKKO for the lunar-ascent model family (e.g. the reference's
Launch_Optimiser.py, which does `from gekko import GEKKO` / `from gekko import *`) runs on libascent.
See lunar_module_ascent_trajectory_optimiser_amd/gekko_shim.py for what is and is not supported."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from lunar_module_ascent_trajectory_optimiser_amd.gekko_shim import GEKKO, ModelNotRecognised  # noqa: E402,F401

__all__ = ["GEKKO"]
