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
    (64, 64, 3, 1, 1, 1, 1, 2100),    # enough positions for the 64 x 256 tile (2 x 2 MFMA tiles per wave, single fragment set)
    (48, 96, 11, 1, 25, 5, 1, 200),   # 96-row tile, 48 channels = 3 groups of 16
    (192, 384, 5, 1, 2, 1, 1, 150),   # 128-row tile
    (40, 480, 3, 1, 1, 1, 1, 170),    # 480 rows: 96-row tiles here (the fp32 path takes 160), channel tail inside a 16-group
    (130, 200, 3, 1, 1, 1, 1, 70),    # channels not multiples of anything
    (96, 96, 16, 1, 8, 1, 4, 130),    # grouped, 24 channels per group
    (32, 32, 3, 2, 0, 1, 1, 301),     # strided
    (48, 48, 3, 1, 1, 1, 1, 300),     # 48 rows (MDX level 0) on the 64-row tile
    (32, 64, 3, 2, 0, 1, 1, 601),     # stride 2 over a 128-position tile: 257-position patch, three 8-channel items per thread
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


def test_fp32_layers_are_never_packed_in_split_precision():
    """Layers packed inside ops.fp32_layers() (RMVPE, CREPE, retrieval: they select indices) stay fp32 whatever the global setting."""
    w = torch.randn(16, 16, 3)
    old = ops.split_precision
    ops.split_precision = True
    try:
        assert ops.PackedConv(w, None).split
        with ops.fp32_layers():
            assert not ops.PackedConv(w, None).split
        assert ops.PackedConv(w, None).split
    finally:
        ops.split_precision = old


def test_default_is_fp32():
    import os
    if os.environ.get("AICG_PRECISION", "fp32").lower() == "fp32":
        assert ops.split_precision is False
        w = torch.randn(32, 32, 3)
        assert ops.pack_conv_weight(w.unsqueeze(2)).numel() == 2 * 3 * 32 * 32
        assert ops.pack_conv_weight(w.unsqueeze(2), split=True).numel() == 3 * 3 * 32 * 32


@pytest.mark.parametrize("R,K,O", [(300, 64, 40), (515, 100, 130), (128, 36, 256), (256, 384, 128)])
def test_gemm_nt_split(dev, R, K, O):
    """aicg_gemm_nt_split: the TDF linear in split precision, ragged rows / K tail inside an 8-chunk / output tile, fused epilogue;
    (256, 384, 128) with 32-row channels takes the float4 epilogue."""
    from aicovergen_amd import _lib
    torch.manual_seed(R + K)
    if dev.big:
        R = R * 40 + (3 if R != 256 else 0)
    n_ch, rows_per_ch = (5, 7) if R % 128 else (4, 32)
    x = torch.randn(R, K)
    w, b = torch.randn(O, K) * 0.2, torch.randn(O)
    sc, sh = torch.rand(n_ch) + 0.5, torch.randn(n_ch)
    res = torch.randn(R, O)
    ch = (torch.arange(R) // rows_per_ch) % n_ch
    ref = torch.relu((x.double() @ w.double().t() + b) * sc[ch, None] + sh[ch, None]).float() + res
    o = dev.t(torch.empty(R, O))
    xd, wd, bd, scd, shd, rd = (dev.t(t) for t in (x, w, b, sc, sh, res))
    st = torch.cuda.current_stream().cuda_stream if dev.kind == "hip" else 0
    _lib.call("aicg_gemm_nt_split", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), scd.data_ptr(), shd.data_ptr(), rd.data_ptr(),
              o.data_ptr(), R, K, O, K, K, O, O, rows_per_ch, n_ch, ops.ACT_RELU, st)
    err = rel_rms(o, ref)
    assert err < TOL, err


def test_large_split_tiles_on_the_emulator():
    """With the fill target lowered (AICG_CONV_WANT=1) small problems take the tiles the bench sizes take: 128 x 128, 96 x 128,
    64 x 256 (single fragment set), 32 x 256, with ragged last tiles, a channel tail and the 2-D halo."""
    from test_conv import _run_child
    code = r'''
ops.split_precision = True
for (ci, co, k, d, T) in [(48, 128, 5, 1, 300), (40, 96, 3, 3, 200), (64, 64, 7, 1, 600), (32, 32, 3, 1, 520), (144, 144, 3, 1, 150)]:
    x = torch.randn(1, ci, T)
    w = torch.randn(co, ci, k) * 0.1
    b, r = torch.randn(co), torch.randn(1, co, T)
    pc = ops.PackedConv(w, b, padding=(k - 1) * d // 2, dilation=d)
    y = ops.conv(x, pc, act=ops.ACT_RELU, res=r)
    e = rel(y, F.relu(F.conv1d(x, w, b, padding=(k - 1) * d // 2, dilation=d)) + r)
    assert 1e-7 < e < 3e-5, (ci, co, k, d, T, e)
x = torch.randn(2, 48, 20, 70)
w = torch.randn(48, 48, 3, 3) * 0.1
pc = ops.PackedConv(w, None, padding=1)
e = rel(ops.conv(x, pc), F.conv2d(x, w, None, padding=1))
assert 1e-7 < e < 3e-5, e
print("large split tiles ok")
'''
    _run_child(code, {"AICG_CONV_WANT": "1"}, "large split tiles ok")
