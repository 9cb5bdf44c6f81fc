"""The one-dimensional Winograd F(2, 3) convolution (csrc/conv_g1w.h: the vocoder's k = 3 / 7 / 11 ResBlock layers, dilation 1 / 3 / 5, reference
src/infer_pack/modules.py:299-312) against torch fp32: every kernel size, every tile, the ResBlock step x + conv(lrelu(x)), the accumulating
last step (xs += resblock(x) / 3, models.py:506-512), plain / activated outputs, ragged channel counts, maps of several column tiles with a
tail, several images, channel-slice operands -- and the routing (dilated, strided, unaligned or short layers stay on the direct kernels).
Tolerance: relative RMS <= 2e-6 (the transform constants are +-1 and 1/2: only fp32 rounding of the transformed operands and the
summation order differ from the direct form; measured 3-6e-7)."""
import random

import pytest
import torch
import torch.nn.functional as F

from aicovergen_amd import _lib, ops
from conftest import rel_rms


def test_slot_kernel_reproduces_the_direct_form():
    """ops.winograd1d_kernel in float64 arithmetic: Y = A^T [sum_slots U (.) V] equals the direct k-tap sum for k = 3, 7, 11."""
    torch.manual_seed(0)
    for k in (3, 5, 7, 11):
        w = torch.randn(2, 3, k, dtype=torch.float64)
        u = ops.winograd1d_kernel(w.float()).double()          # (2, 3, S), rounded to fp32 once
        x = torch.randn(3, 40, dtype=torch.float64)
        P = (k - 1) // 2
        xp = F.pad(x, (P, P + 3))
        ref = F.conv1d(x[None], w, padding=P)[0]
        got = torch.zeros_like(ref)
        for n in range(0, 40, 2):
            M = torch.zeros(4, 2, dtype=torch.float64)
            s = 0
            for g in range(k // 3):
                d = [xp[:, n + 3 * g + m] for m in range(4)]
                for q, v in enumerate((d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3])):
                    M[q] += (u[:, :, s + q] * v).sum(1)
                s += 4
            if k % 3 == 2:
                d = [xp[:, n + 3 * (k // 3) + m] for m in range(3)]
                for q, v in ((0, d[0] - d[1]), (1, d[1]), (3, d[1] - d[2])):
                    M[q] += (u[:, :, s] * v).sum(1)
                    s += 1
            elif k % 3 == 1:
                d = [xp[:, n + 3 * (k // 3) + m] for m in range(2)]
                M[0] += (u[:, :, s] * d[0]).sum(1)
                M[3] += (u[:, :, s + 1] * d[1]).sum(1)
            got[:, n] = M[0] + M[1] + M[2]
            got[:, n + 1] = M[1] - M[2] - M[3]
        assert rel_rms(got, ref) < 1e-6, k


def _run(dev, n, ci, co, k, T, tile, mode, seed=0, expect="conv_g1w_kernel", d=1):
    if tile in (2, 5, 6, 7) and dev.kind == "hip" and not _lib.get_path().endswith("_dev.so"):
        pytest.skip("the A/B variants of conv_g1w (64 x 256 tile, explicit interleave, persistent walk) are compiled into development builds only")
    torch.manual_seed(seed)
    x, w, b = torch.randn(n, ci, T), torch.randn(co, ci, k) * 0.2, torch.randn(co)
    pad = (k - 1) // 2 * d
    pc = ops.PackedConv(w, b, padding=pad, dilation=d, device=dev.device)
    assert pc.w_wino1 is not None
    conv = lambda t: F.conv1d(t, w, b, padding=pad, dilation=d)
    ref = conv(x)
    xd = dev.t(x)
    ops.gemm_tile = tile
    old_min, ops.winograd1d_min_positions = ops.winograd1d_min_positions, 1
    try:
        if mode == "resblock":      # x + conv(lrelu(x)): one ResBlock1 step
            got = ops.conv(xd, pc, res=xd, pre_act=ops.ACT_LRELU, pre_slope=0.1)
            ref = conv(F.leaky_relu(x, 0.1)) + x
        elif mode == "accum":       # xs += resblock(x) / 3
            y0, r = torch.randn_like(ref), torch.randn_like(ref)
            got = dev.t(y0.clone())
            ops.conv(xd, pc, res=dev.t(r), out=got, pre_act=ops.ACT_LRELU, pre_slope=0.1, accumulate=True, out_scale=1 / 3)
            ref = y0 + (conv(F.leaky_relu(x, 0.1)) + r) / 3
        elif mode == "act":
            got, ref = ops.conv(xd, pc, act=ops.ACT_LRELU, act_slope=0.2), F.leaky_relu(ref, 0.2)
        elif mode == "slice":       # input and output are channel slices of wider buffers
            xb = dev.t(torch.randn(n, ci + 3, T))
            xb[:, 1:1 + ci] = xd
            yb = dev.t(torch.full((n, co + 2, T), 5.0))
            ops.conv(xb[:, 1:1 + ci], pc, out=yb[:, 1:1 + co])
            got = yb[:, 1:1 + co]
            assert bool((yb[:, 0] == 5.0).all()) and bool((yb[:, -1] == 5.0).all())
        else:
            got = ops.conv(xd, pc)
        launched = _lib.last_launch()
    finally:
        ops.gemm_tile = 0
        ops.winograd1d_min_positions = old_min
    assert launched == expect, launched
    assert got.shape == ref.shape
    return rel_rms(got, ref)


@pytest.mark.parametrize("tile", [0, 2, 3])
@pytest.mark.parametrize("k", [3, 5, 7, 11])
def test_g1w_resblock_layers(dev, tile, k):
    T = 1304 if dev.big else 392
    co = 32 if tile == 3 else 64
    assert _run(dev, 1, co, co, k, T, tile, "resblock", seed=k) < 2e-6
    assert _run(dev, 2, 48, 72 if tile != 3 else 40, k, T - 128, tile, "accum", seed=k + 1) < 2e-6
    assert _run(dev, 1, 16, 33, k, 8, tile, "plain", seed=k + 2) < 2e-6           # two outputs pairs per lane, one channel stage / two


@pytest.mark.parametrize("k", [3, 5, 7, 11])
def test_g1w_persistent_tile_walk(dev, k):
    """The persistent form (aicg_conv_desc.gemm_tile 6; 7 = the same walk on three workgroups): a workgroup takes tiles blockIdx.x,
    + gridDim.x, ... with the unit pipeline running on into the next tile.  Maps of 5-11 tiles over three workgroups (uneven: some walk
    four tiles, some three), two M tiles, several images, every epilogue mode; and the launch form whose grid covers all tiles."""
    T = 2304 if dev.big else 1096
    assert _run(dev, 1, 32, 32, k, 5 * 512, 7, "resblock", seed=k) < 2e-6
    assert _run(dev, 2, 48, 64, k, T, 7, "accum", seed=k + 1) < 2e-6
    assert _run(dev, 1, 16, 40, k, T + 4, 7, "act", seed=k + 2) < 2e-6
    assert _run(dev, 3, 24, 33, k, 516, 7, "slice", seed=k + 3) < 2e-6
    assert _run(dev, 1, 16, 32, k, 8, 7, "plain", seed=k + 4) < 2e-6            # one tile: no walk
    assert _run(dev, 1, 40, 64, k, T, 6, "resblock" if False else "plain", seed=k + 5) < 2e-6


@pytest.mark.parametrize("d", [3, 5])
@pytest.mark.parametrize("k", [3, 7, 11])
def test_g1w_dilated_layers(dev, k, d):
    """Dilations 3 and 5 (the first convolution of the second and third pair of every ResBlock): output pairs (n, n + d), a lane owns the
    progression n + {0, d, 2 d, 3 d}, 30 of 32 lanes = 120 outputs per wave, the epilogue through an LDS tile.  Maps of several 480-output
    column tiles with a tail, lengths that are not a multiple of 120, ragged channels, several images, every epilogue mode."""
    T = 1304 if dev.big else 500
    assert _run(dev, 1, 64, 64, k, T, 0, "resblock", seed=k + d, d=d) < 2e-6
    assert _run(dev, 2, 48, 72, k, T - 128, 0, "accum", seed=k + d + 1, d=d) < 2e-6
    assert _run(dev, 1, 16, 33, k, 8, 0, "plain", seed=k + d + 2, d=d) < 2e-6
    assert _run(dev, 1, 24, 40, k, 124, 0, "slice", seed=k + d + 3, d=d) < 2e-6
    assert _run(dev, 1, 17, 16, k, 480, 0, "act", seed=k + d + 4, d=d) < 2e-6


@pytest.mark.parametrize("seed", range(20))
def test_g1w_fuzz(dev, seed):
    rng = random.Random(seed)
    n = rng.choice([1, 1, 2])
    ci = rng.choice([16, 24, 40, 64, 100, 130])
    co = rng.choice([16, 32, 33, 40, 64, 128, 200])
    k = rng.choice([3, 5, 7, 11])
    T = 4 * rng.choice([1, 3, 16, 64, 65, 97, 130, 257])
    mode = rng.choice(["plain", "act", "accum", "slice"] + (["resblock"] if co == ci else []))
    d = rng.choice([1, 1, 3, 5]) if k != 5 else 1
    tile = rng.choice([0, 2, 3, 6, 7]) if d == 1 else 0
    err = _run(dev, n, ci, co, k, T, tile, mode, seed=seed, d=d)
    assert err < 2e-6, ((n, ci, co, k, d, T, mode, tile), err)


def test_g1w_leaves_other_layers_alone(dev, monkeypatch):
    """Other dilations / paddings, strided, k = 5, unaligned or short layers: the direct kernels run (and the f0 models never carry the slot
    image)."""
    monkeypatch.setattr(ops, "winograd1d_min_positions", 1)
    torch.manual_seed(1)
    x = torch.randn(1, 32, 256)
    w = torch.randn(64, 32, 3) * 0.2
    for kw in (dict(padding=2, dilation=2), dict(padding=1, dilation=3), dict(padding=1, stride=2), dict(padding=0)):
        y = ops.conv(dev.t(x), ops.PackedConv(w, None, device=dev.device, **kw))
        assert _lib.last_launch() != "conv_g1w_kernel" and rel_rms(y, F.conv1d(x, w, **kw)) < 1e-5
    w9 = torch.randn(64, 32, 9) * 0.2
    y = ops.conv(dev.t(x), ops.PackedConv(w9, None, padding=4, device=dev.device))
    assert _lib.last_launch() != "conv_g1w_kernel" and rel_rms(y, F.conv1d(x, w9, padding=4)) < 1e-5
    x2 = torch.randn(1, 32, 258)                                   # 258 % 4 != 0
    y = ops.conv(dev.t(x2), ops.PackedConv(w, None, padding=1, device=dev.device))
    assert _lib.last_launch() != "conv_g1w_kernel" and rel_rms(y, F.conv1d(x2, w, padding=1)) < 1e-5
    with ops.fp32_layers():
        pc = ops.PackedConv(w, None, padding=1, device=dev.device)
    assert pc.w_wino1 is None
    monkeypatch.setattr(ops, "winograd1d_min_positions", 1 << 30)   # a short map: not worth a launch of the Winograd form
    y = ops.conv(dev.t(x), ops.PackedConv(w, None, padding=1, device=dev.device))
    assert _lib.last_launch() != "conv_g1w_kernel" and rel_rms(y, F.conv1d(x, w, padding=1)) < 1e-5
