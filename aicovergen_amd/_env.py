"""Environment switches of the Python layer (the C library reads none).

Supported in a user's process:
    AICG_PRECISION=bf16x3        opt-in split precision for the conv / TDF family (the f0 models stay fp32)
    AICG_HALF=1                  opt-in fp16 matrix arithmetic where the caller ALSO passes is_half=True (src/main.py:196 does): HuBERT's and the
                                 synthesizer's LDS-DMA staged layers take fp16 operands, fp32 activations / accumulation (ops.mark_half; the f0
                                 models, SineGen, attention and MDX-Net stay fp32).  Without it .half() is a no-op and everything is fp32
    AICG_FORCE_COLLECTIVES=1     a one-rank process group runs every join through the real collectives (tests/test_rccl_one_rank.py)

Everything else -- kernel-form selectors (AICG_WINOGRAD, AICG_WINOGRAD1D, AICG_W2D_*, AICG_GRU_*), schedule selectors (AICG_F0_SEGMENTS,
AICG_OVERLAP_F0, AICG_OVERLAP_SYNTH, AICG_F0_PRIORITY, AICG_RB_STREAMS, AICG_MDX_BATCH) and reference-path selectors (AICG_FILTFILT,
AICG_GPU_KNN, AICG_KNN) -- changes routing or summation ORDER (never the arithmetic's meaning) and exists for A/B measurements and for
the tests that pin one form against another.  They are DEVELOPMENT switches: read through dev() and honoured only when AICG_DEV=1 is
set too, so that a stray variable in a user's shell cannot silently change which kernels run.  tests/conftest.py
and tools/ set AICG_DEV=1; a development variable that is set without it is reported once and ignored."""
import os
import warnings

_reported = set()


def dev_enabled():
    return os.environ.get("AICG_DEV") == "1"


def dev(name, default=None):
    """os.environ.get(name, default) for a development switch: the default unless AICG_DEV=1."""
    if dev_enabled():
        return os.environ.get(name, default)
    if name in os.environ and name not in _reported:
        _reported.add(name)
        warnings.warn("%s=%r is a development switch of aicovergen_amd and is ignored (set AICG_DEV=1 to honour it; see "
                      "aicovergen_amd/_env.py)" % (name, os.environ[name]), RuntimeWarning, stacklevel=2)
    return default
