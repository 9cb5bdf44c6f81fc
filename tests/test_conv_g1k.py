"""The LDS-DMA staged k-tap 1-D convolution (csrc/conv_g1k.h: the vocoder's ResBlock layers, reference src/infer_pack/modules.py:299-312)
against torch fp32, forced per launch through aicg_conv_desc.gemm_tile (12 = 128 x 256, 13 = 64 x 256 tile; development builds only:
the kernel measured 10-25 % slower than conv_ws3 on these layers and is not in the product library, so the hardware variants of these
tests skip unless a development library is bound): every kernel size / dilation of
the ResBlocks and more, window shifts of every residue mod 4, ragged channels, tiles with a tail, several images, every epilogue mode.
Tolerance: relative RMS <= 1e-5 (same fp32 products, different summation order)."""
import random

import pytest
import torch
import torch.nn.functional as F

from aicovergen_amd import _lib, ops
from conftest import rel_rms


def _run(dev, n, ci, co, k, d, T, tile, mode, pad=None, seed=0):
    if dev.kind == "hip" and not _lib.get_path().endswith("_dev.so"):
        pytest.skip("conv_g1k is compiled into development builds only (DESIGN 2.11)")
    torch.manual_seed(seed)
    pad = (k - 1) * d // 2 if pad is None else pad
    pad_end = (k - 1) * d - pad
    x, w, b = torch.randn(n, ci, T), torch.randn(co, ci, k) * 0.2, torch.randn(co)
    pc = ops.PackedConv(w, b, padding=pad, padding_end=pad_end, dilation=d, device=dev.device)
    ref = F.conv1d(F.pad(x, (pad, pad_end)), w, b, dilation=d)
    xd = dev.t(x)
    ops.gemm_tile = tile
    try:
        if mode == "resblock":      # x + conv(lrelu(x)): one ResBlock1 step
            assert ci == co
            got = ops.conv(xd, pc, res=xd, pre_act=ops.ACT_LRELU, pre_slope=0.1)
            ref = F.conv1d(F.pad(F.leaky_relu(x, 0.1), (pad, pad_end)), w, b, dilation=d) + x
        elif mode == "accum":       # xs += resblock(x) / 3 (models.py:506-512)
            y0 = torch.randn_like(ref)
            got = dev.t(y0.clone())
            r = torch.randn_like(ref)
            ops.conv(xd, pc, res=dev.t(r), out=got, pre_act=ops.ACT_LRELU, pre_slope=0.1, accumulate=True, out_scale=1 / 3)
            ref = y0 + (F.conv1d(F.pad(F.leaky_relu(x, 0.1), (pad, pad_end)), w, b, dilation=d) + r) / 3
        elif mode == "act":
            got, ref = ops.conv(xd, pc, act=ops.ACT_LRELU, act_slope=0.2), F.leaky_relu(ref, 0.2)
        else:
            got = ops.conv(xd, pc)
        launched = _lib.last_launch()
    finally:
        ops.gemm_tile = 0
    assert launched == "conv_g1k_kernel", launched
    assert got.shape == ref.shape
    return rel_rms(got, ref)


@pytest.mark.parametrize("tile", [12, 13])
@pytest.mark.parametrize("k,d", [(3, 1), (3, 3), (3, 5), (7, 1), (7, 3), (7, 5), (11, 1), (11, 3), (11, 5), (2, 1), (5, 2), (4, 3)])
def test_g1k_resblock_geometries(dev, tile, k, d):
    """The vocoder's (kernel, dilation) pairs and a few others (even kernels: asymmetric padding), 64 = 64 channels, a map of one full tile
    plus a tail; the ResBlock step x + conv(lrelu(x)) and the accumulating last step."""
    T = 1300 if dev.big else 388
    if k % 2:
        assert _run(dev, 1, 64, 64, k, d, T, tile, "resblock", seed=k * 10 + d) < 1e-5
    assert _run(dev, 2, 48, 72, k, d, T - 128, tile, "accum", seed=k * 10 + d + 1) < 1e-5


@pytest.mark.parametrize("seed", range(24))
def test_g1k_fuzz(dev, seed):
    rng = random.Random(seed)
    n = rng.choice([1, 1, 2])
    ci = rng.choice([16, 24, 40, 64, 100, 130])
    co = rng.choice([33, 40, 64, 128, 200]) if (rng.random() < 0.6 or ci <= 32) else ci   # (<= 32 output channels: not this kernel's)
    k = rng.choice([2, 3, 5, 7, 11])
    d = rng.choice([1, 1, 2, 3, 5])
    T = 4 * rng.choice([3, 16, 64, 65, 97, 130, 257])
    pad = rng.randint(0, (k - 1) * d)
    mode = rng.choice(["plain", "act", "accum"] + (["resblock"] if co == ci else []))
    tile = rng.choice([12, 13])
    err = _run(dev, n, ci, co, k, d, T, tile, mode, pad=pad, seed=seed)
    assert err < 1e-5, ((n, ci, co, k, d, T, pad, mode, tile), err)


def test_g1k_leaves_other_layers_alone(dev):
    """Strided, grouped, unaligned or length-changing layers are not the kernel's: the forced tile is ignored and the usual kernels run."""
    torch.manual_seed(1)
    x, w = torch.randn(1, 32, 258), torch.randn(64, 32, 3) * 0.2          # 258 % 4 != 0
    ops.gemm_tile = 12
    try:
        y = ops.conv(dev.t(x), ops.PackedConv(w, None, padding=1, device=dev.device))
        assert _lib.last_launch() != "conv_g1k_kernel"
        assert rel_rms(y, F.conv1d(x, w, padding=1)) < 1e-5
        x2 = torch.randn(1, 32, 256)
        y2 = ops.conv(dev.t(x2), ops.PackedConv(w, None, stride=2, padding=1, device=dev.device))
        assert _lib.last_launch() != "conv_g1k_kernel"
        assert rel_rms(y2, F.conv1d(x2, w, stride=2, padding=1)) < 1e-5
        y3 = ops.conv(dev.t(x2), ops.PackedConv(w, None, padding=0, device=dev.device))   # output shorter than the input
        assert _lib.last_launch() != "conv_g1k_kernel"
        assert rel_rms(y3, F.conv1d(x2, w)) < 1e-5
    finally:
        ops.gemm_tile = 0
