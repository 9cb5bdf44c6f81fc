"""Drop-in shadow of the reference's src/my_utils.py: put this directory ahead of the reference's src/ on sys.path
(or launch through src/run_main.py) and main.py's imports resolve to the MI355X implementation."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from aicovergen_amd.my_utils import *  # noqa: F401,F403
from aicovergen_amd import my_utils as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
