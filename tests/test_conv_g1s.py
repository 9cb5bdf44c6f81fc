"""The LDS-DMA staged STRIDE-2 k-tap 1-D convolution (csrc/conv_g1s.h: HuBERT's feature extractor -- fairseq's ConvFeatureExtractionModel
behind reference src/rvc.py:98-109, k = 3 / 2, stride 2, no padding, GELU behind) against torch fp32: both kernel sizes, both tiles, odd
and ragged lengths (the extractor's 211 231 -> 105 615 -> ... chain never has a multiple of four), padded row strides on either side,
ragged channel counts, several images, every epilogue mode, and the routing rules (unaligned rows, padding, other strides stay on the
producer / consumer kernels).  Tolerance: relative RMS <= 1e-5 (same fp32 products, different summation order)."""
import random

import pytest
import torch
import torch.nn.functional as F

from aicovergen_amd import _lib, ops
from conftest import rel_rms


def padded(dev, n, c, t, fill=None):
    """(n, c, t) view of a buffer whose rows are 16-byte aligned (row stride rounded up to a multiple of 4), as hubert._frontend lays the
    extractor's activations out; the padding holds `fill` (NaN: a read of it that reached a stored output would show)."""
    tp = (t + 3) // 4 * 4
    buf = torch.full((n, c, tp), float("nan") if fill is None else fill)
    return dev.t(buf), tp


def _run(dev, n, ci, co, k, T, tile, mode="gelu", seed=0, expect="conv_g1s_kernel"):
    torch.manual_seed(seed)
    x, w, b = torch.randn(n, ci, T), torch.randn(co, ci, k) * 0.2, torch.randn(co)
    to = (T - k) // 2 + 1
    pc = ops.PackedConv(w, b, stride=2, device=dev.device)
    xb, tp = padded(dev, n, ci, T)
    xb[:, :, :T] = dev.t(x)
    yb, top = padded(dev, n, co, to, fill=7.0)
    ref = F.conv1d(x, w, b, stride=2)
    ops.gemm_tile = tile
    try:
        if mode == "gelu":
            ops.conv(xb[:, :, :T], pc, act=ops.ACT_GELU, out=yb[:, :, :to])
            ref = F.gelu(ref)
        elif mode == "res":
            r = torch.randn_like(ref)
            rb, _ = padded(dev, n, co, to)
            rb[:, :, :to] = dev.t(r)
            ops.conv(xb[:, :, :T], pc, res=rb[:, :, :to], out=yb[:, :, :to])
            ref = ref + r
        elif mode == "accum":
            y0 = torch.randn_like(ref)
            yb[:, :, :to] = dev.t(y0)
            ops.conv(xb[:, :, :T], pc, out=yb[:, :, :to], act=ops.ACT_LRELU, act_slope=0.1, accumulate=True, out_scale=0.5)
            ref = y0 + 0.5 * F.leaky_relu(ref, 0.1)
        else:
            ops.conv(xb[:, :, :T], pc, out=yb[:, :, :to])
        launched = _lib.last_launch()
    finally:
        ops.gemm_tile = 0
    assert launched == expect, launched
    got = yb[:, :, :to]
    assert got.shape == ref.shape
    if top > to:                                       # the row padding behind the last output is not written
        assert bool((yb[:, :, to:] == 7.0).all())
    return rel_rms(got, ref)


@pytest.mark.parametrize("tile", [2, 3])
@pytest.mark.parametrize("k", [3, 2])
def test_g1s_extractor_geometries(dev, tile, k):
    """512-wide layers are for the hardware; here 64 / 72 channels (one and two M tiles of 64, a ragged last one), lengths that leave every
    residue mod 4 on both sides, more than one column tile, a tail tile."""
    for T in ((1031, 1030, 1029, 517) if dev.big else (531, 530, 277, 129)):
        assert _run(dev, 1, 64, 64, k, T, tile, "gelu", seed=T + k) < 1e-5
    assert _run(dev, 2, 40, 72, k, 601 if dev.big else 301, tile, "res", seed=5) < 1e-5
    assert _run(dev, 1, 24, 130, k, 259, tile, "accum", seed=6) < 1e-5
    assert _run(dev, 1, 16, 40, k, 64, tile, "plain", seed=7) < 1e-5          # one unit pair; 31 outputs
    assert _run(dev, 1, 17, 33, k, 9, tile, "plain", seed=8) < 1e-5           # three units with a ragged last one; 4 outputs


@pytest.mark.gpu
def test_g1s_hubert_base_layers():
    """The benched shapes under the POLICY: 512 -> 512, k = 3 on 211 231 and 26 403 frames, k = 2 on 13 201 (the kernel's) and on 6 600 (104
    workgroups: the producer / consumer kernel keeps it), against torch on the host."""
    import conftest
    conftest._bind("hip")
    dev = conftest.Dev("hip")
    for k, T, want in ((3, 211231, "conv_g1s_kernel"), (3, 26403, "conv_g1s_kernel"), (2, 13201, "conv_g1s_kernel"), (2, 6600, "conv_ws3_kernel")):
        assert _run(dev, 1, 512, 512, k, T, 0, "gelu", seed=k, expect=want) < 1e-5


@pytest.mark.parametrize("seed", range(16))
def test_g1s_fuzz(dev, seed):
    rng = random.Random(seed)
    n = rng.choice([1, 1, 2])
    ci = rng.choice([16, 24, 40, 64, 100])
    co = rng.choice([33, 40, 64, 128, 200])
    k = rng.choice([2, 3])
    T = rng.choice([8, 33, 130, 257, 515, 1026])
    mode = rng.choice(["plain", "gelu", "res", "accum"])
    tile = rng.choice([2, 3])
    err = _run(dev, n, ci, co, k, T, tile, mode, seed=seed)
    assert err < 1e-5, ((n, ci, co, k, T, mode, tile), err)


def test_g1s_leaves_other_layers_alone(dev):
    """Unaligned rows (a contiguous odd-length map), padding, stride 1 / 3, k = 4, a pre-activation: the usual kernels run."""
    torch.manual_seed(1)
    w = torch.randn(64, 32, 3) * 0.2
    x = torch.randn(1, 32, 259)                                     # contiguous, 259 % 4 != 0: rows are not 16-byte aligned
    y = ops.conv(dev.t(x), ops.PackedConv(w, None, stride=2, device=dev.device))
    assert _lib.last_launch() != "conv_g1s_kernel" and rel_rms(y, F.conv1d(x, w, stride=2)) < 1e-5
    x = torch.randn(1, 32, 256)
    y = ops.conv(dev.t(x), ops.PackedConv(w, None, stride=2, padding=1, device=dev.device))
    assert _lib.last_launch() != "conv_g1s_kernel" and rel_rms(y, F.conv1d(x, w, stride=2, padding=1)) < 1e-5
    y = ops.conv(dev.t(x), ops.PackedConv(w, None, stride=3, device=dev.device))
    assert _lib.last_launch() != "conv_g1s_kernel" and rel_rms(y, F.conv1d(x, w, stride=3)) < 1e-5
    w4 = torch.randn(64, 32, 4) * 0.2
    y = ops.conv(dev.t(x), ops.PackedConv(w4, None, stride=2, device=dev.device))
    assert _lib.last_launch() != "conv_g1s_kernel" and rel_rms(y, F.conv1d(x, w4, stride=2)) < 1e-5
    y = ops.conv(dev.t(x), ops.PackedConv(w, None, stride=2, device=dev.device), pre_act=ops.ACT_LRELU, pre_slope=0.1)
    assert _lib.last_launch() != "conv_g1s_kernel" and rel_rms(y, F.conv1d(F.leaky_relu(x, 0.1), w, stride=2)) < 1e-5
    ops.gemm_tile = 1                                                # and the per-launch opt-out
    try:
        y = ops.conv(dev.t(x), ops.PackedConv(w, None, stride=2, device=dev.device))
        assert _lib.last_launch() != "conv_g1s_kernel" and rel_rms(y, F.conv1d(x, w, stride=2)) < 1e-5
    finally:
        ops.gemm_tile = 0
    # an output whose rows are not 16-byte aligned (127 contiguous floats) keeps the layer off the kernel too; in a padded buffer it is the kernel's
    y = ops.conv(dev.t(x), ops.PackedConv(w, None, stride=2, device=dev.device))
    assert _lib.last_launch() != "conv_g1s_kernel" and rel_rms(y, F.conv1d(x, w, stride=2)) < 1e-5
    yb, _ = padded(dev, 1, 64, 127)
    ops.gemm_tile = 3
    try:
        ops.conv(dev.t(x), ops.PackedConv(w, None, stride=2, device=dev.device), out=yb[:, :, :127])
    finally:
        ops.gemm_tile = 0
    assert _lib.last_launch() == "conv_g1s_kernel" and rel_rms(yb[:, :, :127], F.conv1d(x, w, stride=2)) < 1e-5
    # the policy sends a launch of fewer than 160 workgroups to the smaller tiles of the producer / consumer kernels
    ops.conv(dev.t(x), ops.PackedConv(w, None, stride=2, device=dev.device), out=yb[:, :, :127])
    assert _lib.last_launch() != "conv_g1s_kernel" and rel_rms(yb[:, :, :127], F.conv1d(x, w, stride=2)) < 1e-5
