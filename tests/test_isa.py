"""Properties of the compiled dominant kernel that round 6 established and that a compiler or source change could silently lose (no GPU:
hipcc cross-compiles gfx950 here): every instantiation of conv_w2d keeps its 192 accumulators in registers -- 0 spilled registers, 0 bytes
of scratch memory -- and every MFMA accumulates IN PLACE (destination = addend, or an item's first k-step from the constant 0).  Before the
MFMAs went through inline asm with a tied destination, hipcc moved accumulator quads with untied v_mfma destinations in straight-line code
and 41-57 registers went through scratch inside the MFMA stream (profiles/NOTES.md R6.2b; VERDICT r5 #2: ".vgpr_spill_count: 0")."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (cross-compiles without a GPU)")
def test_conv_w2d_keeps_its_accumulators_in_registers():
    import isa_check
    res = isa_check.analyse(isa_check.compile_asm(os.path.join(isa_check.CSRC, "conv_w2d_1.hip")))
    kernels = {k: v for k, v in res.items() if "conv_w2d_kernel" in k}
    assert len(kernels) >= 5, list(res)
    for k, v in kernels.items():
        assert v["spill"] == 0 and v["scratch"] == 0 and v["scratch_ops"] == 0, (k, v)
        assert v["untied"] == 0 and v["tied"] > 0 and v["zero"] > 0, (k, v)
        assert v["vgpr"] <= 256 or "ILi4E" in k, (k, v)          # the eight-wave forms: two waves per SIMD
