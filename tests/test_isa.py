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


def test_hazard_scan_flags_what_it_should():
    """tools/isa_hazard_scan.py on a hand-written block: a VALU result read by an MFMA one slot later, an MFMA result read by a VALU
    instruction three slots later -- and neither once the s_nop is there."""
    import isa_hazard_scan
    bad = """
_Z4fake_conv_w2d_kernel:
	v_sub_f32_e32 v5, v1, v2
	v_mfma_f32_16x16x4_f32 v[8:11], v5, v6, v[8:11]
	v_mov_b32_e32 v20, v21
	v_add_f32_e32 v0, v8, v9
	s_endpgm
"""
    (v1, v2, ex), = isa_hazard_scan.scan(bad).values()
    assert v1 == 2 and v2 == 1, (v1, v2, ex)
    good = bad.replace("\tv_mfma", "\ts_nop 1\n\tv_mfma").replace("\tv_mov_b32_e32 v20, v21", "\ts_nop 15")
    (v1, v2, ex), = isa_hazard_scan.scan(good).values()
    assert v1 == 0 and v2 == 0, (v1, v2, ex)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (cross-compiles without a GPU)")
def test_conv_w2d_asm_mfmas_meet_no_hazard():
    """The two hazards hipcc handles for the MFMA builtin and cannot see behind inline asm (a VALU result read by an MFMA within 2 wait
    states, an MFMA result read by anything else within 11) do not occur in any instantiation of the compiled kernel."""
    import isa_check
    import isa_hazard_scan
    res = isa_hazard_scan.scan(isa_check.compile_asm(os.path.join(isa_check.CSRC, "conv_w2d_1.hip")))
    assert len(res) >= 5
    for k, (v1, v2, ex) in res.items():
        assert v1 == 0 and v2 == 0, (k, ex)
