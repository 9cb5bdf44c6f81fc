"""Test configuration.

Two ways the same kernel sources are exercised:
  * "emu": csrc/*.hip compiled for the host against tests/emu (fiber-based HIP stand-in) -- runs in the
    GPU-less container, checks kernel logic (indexing, LDS staging, barriers, MFMA fragment layouts);
  * "hip": the real gfx950 library through the same C ABI on cuda:0 -- marked `gpu`.
Every parity test takes the `dev` fixture and therefore exists in both variants.
"""
import os
import sys

import pytest
import torch

os.environ.setdefault("AICG_DEV", "1")   # tests pin kernel forms and schedules against each other: the development switches are live (aicovergen_amd/_env.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Dev:
    def __init__(self, kind):
        self.kind = kind
        self.device = torch.device("cuda:0") if kind == "hip" else torch.device("cpu")
        self.big = kind == "hip"  # hardware runs use larger shapes

    def t(self, x):
        return x.to(self.device)

    def sync(self):
        if self.kind == "hip":
            torch.cuda.synchronize()


_emu_path = None


def _bind(kind):
    global _emu_path
    from aicovergen_amd import _lib
    if kind == "emu":
        if _emu_path is None:
            sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
            import build_emu
            # AICG_EMU_SANITIZE=1 (tests/emu/run_sanitized.sh): the AddressSanitizer + UBSan build of the emulator
            _emu_path = build_emu.build_emu(sanitize=os.environ.get("AICG_EMU_SANITIZE") == "1")
        _lib._use_library_for_tests(_emu_path, "emu")
    else:
        if not torch.cuda.is_available():
            pytest.skip("no GPU")
        _lib._reset_for_tests()
        _lib.get()  # raises loudly if libaicg_hip.so is missing


@pytest.fixture(params=[pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)])
def dev(request):
    _bind(request.param)
    yield Dev(request.param)
    if request.param == "hip":
        torch.cuda.synchronize()


def rel_rms(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float(((a - b).pow(2).sum() / b.pow(2).sum().clamp_min(1e-300)).sqrt())
