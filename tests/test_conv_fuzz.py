"""Seeded random convolution geometries (groups, strides, dilations, asymmetric kernels, 1-D / 2-D, fused epilogue modes)
against torch: exercises the dispatcher's tile choices -- small tiles, the wave-specialised kernels on long layers, the
16x16x4 path, the streaming 1x1 form and the narrow-tile retry for oversized patches."""
import random

import pytest
import torch
import torch.nn.functional as F

from aicovergen_amd import ops
from conftest import rel_rms


def _case(dev, seed, big):
    rng = random.Random(seed)
    torch.manual_seed(seed)
    two_d = rng.random() < 0.5
    groups = rng.choice([1, 1, 1, 2] if big else [1, 1, 1, 2, 4, 16])
    cin = groups * rng.choice([3, 4, 8, 16, 20, 33, 48] if big else [1, 2, 3, 4, 6, 8, 12, 20, 33, 48])
    cout = groups * rng.choice([4, 12, 16, 40, 48, 64, 72, 96, 100, 128, 144, 160, 200] if big
                               else [1, 2, 4, 5, 8, 12, 16, 24, 40, 48, 72, 100])
    kh = rng.choice([1, 2, 3]) if two_d else 1
    kw = rng.choice([1, 2, 3, 5, 7] if big else [1, 2, 3, 5, 7, 11])
    sh = rng.choice([1, 1, 2]) if two_d else 1
    sw = rng.choice([1, 1, 1, 2] if big else [1, 1, 1, 2, 3, 5])
    dh = 1 if big else (rng.choice([1, 1, 2]) if two_d else 1)
    dw = rng.choice([1, 1, 3] if big else [1, 1, 2, 3, 5])
    ph = rng.randint(0, (kh - 1) * dh) if two_d else 0
    pw = rng.randint(0, (kw - 1) * dw)
    n = rng.choice([1, 2] if big else [1, 1, 2, 3])
    if big:
        h, w = (rng.choice([40, 70, 130, 260]), rng.choice([300, 520, 1030])) if two_d else (1, rng.choice([70001, 90000, 131072]))
    else:
        h, w = (rng.choice([1, 3, 8, 17, 40]) if two_d else 1), rng.choice([7, 33, 64, 130, 257, 700, 3000])
    if h + 2 * ph - dh * (kh - 1) - 1 < 0 or w + 2 * pw - dw * (kw - 1) - 1 < 0:
        return None
    work = cin // groups * kh * kw * cout * n * (h // sh) * (w // sw)
    if work > (6e9 if big else 2e9) or n * cout * h * w > (2e7 if big else 3e6):
        return None
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin // groups, kh, kw) * 0.2
    b = torch.randn(cout) if rng.random() < 0.7 else None
    kwargs = dict(stride=(sh, sw), padding=(ph, pw), dilation=(dh, dw), groups=groups)
    pc = ops.PackedConv(wt, b, device=dev.device, **kwargs)
    ref = F.conv2d(x, wt, b, **kwargs)
    mode = rng.choice(["plain", "act", "res", "accum", "pre"])
    xd = dev.t(x)
    if mode == "act":
        got, ref = ops.conv(xd, pc, act=ops.ACT_LRELU, act_slope=0.1), F.leaky_relu(ref, 0.1)
    elif mode == "res":
        r = torch.randn_like(ref)
        got, ref = ops.conv(xd, pc, res=dev.t(r), act=ops.ACT_RELU), F.relu(ref) + r
    elif mode == "pre":
        got, ref = ops.conv(xd, pc, pre_act=ops.ACT_LRELU, pre_slope=0.1), F.conv2d(F.leaky_relu(x, 0.1), wt, b, **kwargs)
    elif mode == "accum":
        y0 = torch.randn_like(ref)
        got = dev.t(y0.clone())
        ops.conv(xd, pc, out=got, accumulate=True, out_scale=0.5)
        ref = y0 + 0.5 * ref
    else:
        got = ops.conv(xd, pc)
    return rel_rms(got, ref), dict(d2=two_d, g=groups, cin=cin, cout=cout, k=(kh, kw), s=(sh, sw), d=(dh, dw), p=(ph, pw), n=n,
                                   hw=(h, w), mode=mode)


def test_random_small_geometries(dev):
    ran = 0
    for seed in range(200 if dev.big else 50):
        out = _case(dev, seed, big=False)
        if out is None:
            continue
        ran += 1
        assert out[0] < 2e-5, (seed, out)
    assert ran > 20


@pytest.mark.parametrize("seed", [0, 3, 4, 13, 16, 19, 28])
def test_random_long_layers(dev, seed):
    out = _case(dev, seed, big=True)
    assert out is not None and out[0] < 2e-5, (seed, out)
