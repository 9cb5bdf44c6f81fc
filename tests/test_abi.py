"""The C-ABI boundary (include/aicg.h <-> aicovergen_amd/libaicg_hip.so): loads without a GPU, exports every declared
entry point, reports errors as codes + aicg_last_error(), and the Python side refuses to run without it."""
import ctypes
import os
import re

import pytest

from aicovergen_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_SO = os.path.join(ROOT, "aicovergen_amd", "libaicg_hip.so")


@pytest.fixture(scope="module")
def hip_lib():
    if not os.path.exists(HIP_SO):  # fresh checkout: cross-compile for gfx950 (no GPU needed)
        from aicovergen_amd.build import build_hip
        build_hip()
    return ctypes.CDLL(HIP_SO)


def _declared():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "aicg.h")).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(aicg_\w+)\s*\(", text)))


def test_header_prototypes_parse():
    protos = _lib.parse_header()
    names = _declared()
    assert len(names) >= 35
    # every function-like name in the header is an `int aicg_*(...)` prototype the ctypes loader understands,
    # except the two `const char*` accessors
    assert set(names) - set(protos) <= {"aicg_last_error", "aicg_last_launch"}


def test_hip_library_exports_every_declared_symbol(hip_lib):
    missing = [n for n in _declared() if not hasattr(hip_lib, n)]
    assert not missing, missing


def test_abi_version_and_error_reporting_without_gpu(hip_lib):
    hip_lib.aicg_abi_version.restype = ctypes.c_int
    assert hip_lib.aicg_abi_version() == 5
    hip_lib.aicg_last_error.restype = ctypes.c_char_p
    # argument validation happens before any HIP call: null pointers -> AICG_E_ARG (-2) and a message, no device needed
    hip_lib.aicg_complex_abs.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p]
    hip_lib.aicg_complex_abs.restype = ctypes.c_int
    rc = hip_lib.aicg_complex_abs(None, None, None, 16, None)
    assert rc < 0
    assert b"aicg_complex_abs" in hip_lib.aicg_last_error()


def test_missing_library_is_a_loud_error(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib._load(str(tmp_path / "libaicg_hip.so"), "hip")


def test_emulator_build_exports_the_same_abi():
    emu = os.path.join(ROOT, "tests", "emu", "libaicg_emu.so")
    if not os.path.exists(emu):
        pytest.skip("emulator library not built yet (conftest builds it on first use)")
    lib = ctypes.CDLL(emu)
    assert not [n for n in _declared() if not hasattr(lib, n)]
