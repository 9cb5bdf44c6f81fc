"""Drop-in shadow of the reference's src/infer_pack/models.py (synthesizer classes on the MI355X kernels)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
from aicovergen_amd.infer_pack.models import *  # noqa: F401,F403
from aicovergen_amd.infer_pack.models import (SynthesizerTrnMs256NSFsid, SynthesizerTrnMs256NSFsid_nono,  # noqa: F401
                                              SynthesizerTrnMs768NSFsid, SynthesizerTrnMs768NSFsid_nono)
