"""Opt-in split-precision convolution (AICG_PRECISION=bf16x3, csrc/conv_ws3s.h) against torch fp32.

Each operand is carried as bf16 hi + bf16 lo (16 significand bits) and each product as hi*hi + hi*lo + lo*hi with fp32
accumulation; the dropped lo*lo term is <= 2^-16 of a product.  Tolerance stated here: relative RMS <= 3e-5 against the fp32
reference (the fp32-MFMA kernels meet 1e-5; the reference's own GPU path is fp16, src/rvc.py:103-104, ~1e-3)."""
import pytest
import torch
import torch.nn.functional as F

from aicovergen_amd import ops
from conftest import rel_rms

TOL = 3e-5

# (Cin, Cout, k, stride, pad, dil, groups, T): every tile of the split dispatcher, ragged channels, taps split over stages
CASES = [
    (32, 32, 3, 1, 1, 1, 1, 300),     # 32-row tile
    (64, 64, 7, 1, 9, 3, 1, 260),     # 64-row tile, dilated taps
    (48, 96, 11, 1, 25, 5, 1, 200),   # 96-row tile, 48 channels = 3 groups of 16
    (192, 384, 5, 1, 2, 1, 1, 150),   # 128-row tile
    (40, 480, 3, 1, 1, 1, 1, 170),    # 480 rows: 96-row tiles here (the fp32 path takes 160), channel tail inside a 16-group
    (130, 200, 3, 1, 1, 1, 1, 70),    # channels not multiples of anything
    (96, 96, 16, 1, 8, 1, 4, 130),    # grouped, 24 channels per group
    (32, 32, 3, 2, 0, 1, 1, 301),     # strided
]


@pytest.fixture
def split():
    old = ops.split_precision
    ops.split_precision = True
    yield
    ops.split_precision = old


@pytest.mark.parametrize("ci,co,k,s,p,d,g,T", CASES)
def test_conv1d_split(dev, split, ci, co, k, s, p, d, g, T):
    torch.manual_seed(ci * 1000 + co)
    if dev.big:
        T *= 9
    x = torch.randn(2, ci, T)
    w = torch.randn(co, ci // g, k) * 0.1
    b = torch.randn(co)
    pc = ops.PackedConv(w, b, stride=s, padding=p, dilation=d, groups=g, device=dev.device)
    assert pc.split
    y = ops.conv(dev.t(x), pc)
    ref = F.conv1d(x, w, b, stride=s, padding=p, dilation=d, groups=g)
    assert y.shape == ref.shape
    err = rel_rms(y, ref)
    assert err < TOL, err


def test_conv2d_split_fused(dev, split):
    """2-D tile with halo, lrelu prologue, residual and accumulate: the fused forms ride on the same epilogue."""
    torch.manual_seed(7)
    H, W = (64, 256) if dev.big else (20, 64)
    x = torch.randn(1, 32, H, W)
    w = torch.randn(64, 32, 3, 3) * 0.1
    b = torch.randn(64)
    r = torch.randn(1, 64, H, W)
    pc = ops.PackedConv(w, b, padding=1, device=dev.device)
    y = ops.conv(dev.t(x), pc, res=dev.t(r), pre_act=ops.ACT_LRELU, pre_slope=0.1, act=ops.ACT_RELU)
    ref = F.relu(F.conv2d(F.leaky_relu(x, 0.1), w, b, padding=1)) + r
    err = rel_rms(y, ref)
    assert err < TOL, err


def test_split_is_not_plain_bf16(dev, split):
    """The lo terms are really there: plain bf16 operands would sit at ~3e-3."""
    torch.manual_seed(11)
    x = torch.randn(1, 64, 500)
    w = torch.randn(64, 64, 3) * 0.1
    pc = ops.PackedConv(w, None, padding=1, device=dev.device)
    y = ops.conv(dev.t(x), pc)
    ref = F.conv1d(x, w, None, padding=1)
    plain = F.conv1d(x.bfloat16().float(), w.bfloat16().float(), None, padding=1)
    assert rel_rms(plain, ref) > 1e-3
    assert rel_rms(y, ref) < TOL


def test_default_is_fp32():
    import os
    if os.environ.get("AICG_PRECISION", "fp32").lower() == "fp32":
        assert ops.split_precision is False
        w = torch.randn(32, 32, 3)
        assert ops.pack_conv_weight(w.unsqueeze(2)).numel() == 2 * 3 * 32 * 32
        assert ops.pack_conv_weight(w.unsqueeze(2), split=True).numel() == 3 * 3 * 32 * 32
