"""The LDS-DMA staged 1 x 1 GEMM (csrc/conv_g1.h) against torch fp32: every tile shape, every epilogue form it takes (bias, activation,
residual before / after, accumulate, out_scale, the k = s = 2 transposed-conv scatter with additive / multiplicative skip), ragged M / K /
position tails, several images.  Tolerance: relative RMS <= 1e-5 (same fp32 products, different summation order).

On the emulator the tile is forced through AICG_CONV_G1 (dev switch, read once per process: child processes); on the GPU the product
library's own policy picks the kernel, so those cases are sized past its thresholds."""
import pytest
import torch
import torch.nn.functional as F

from aicovergen_amd import _lib, ops
from conftest import rel_rms
from test_conv import _run_child

CHILD = r'''
from aicovergen_amd import _lib
def gelu(v): return F.gelu(v)
cases = [  # n, ci, co, h, w
    (1, 16, 128, 1, 512), (1, 44, 200, 1, 300), (2, 24, 72, 3, 100), (1, 100, 40, 1, 260), (1, 33, 130, 2, 258 * 2)]
for (n, ci, co, h, w) in cases:
    x, wt, b = torch.randn(n, ci, h, w), torch.randn(co, ci, 1, 1) * 0.2, torch.randn(co)
    r, y0 = torch.randn(n, co, h, w), torch.randn(n, co, h, w)
    pc = ops.PackedConv(wt, b)
    ref = F.conv2d(x, wt, b)
    assert rel(ops.conv(x, pc), ref) < 1e-5, ("plain", n, ci, co, h, w)
    assert _lib.last_launch() == "conv_g1_kernel", _lib.last_launch()
    assert rel(ops.conv(x, pc, act=ops.ACT_GELU, out_scale=0.5), 0.5 * gelu(ref)) < 1e-5, ("gelu", n, ci, co, h, w)
    assert rel(ops.conv(x, pc, res=r, act=ops.ACT_RELU), F.relu(ref) + r) < 1e-5, ("res", n, ci, co, h, w)
    assert rel(ops.conv(x, pc, res=r, act=ops.ACT_LRELU, act_slope=0.1, res_before_act=True), F.leaky_relu(ref + r, 0.1)) < 1e-5, ("res first", n, ci, co, h, w)
    assert rel(ops.conv(x, pc, pre_act=ops.ACT_LRELU, pre_slope=0.1, res=r), F.conv2d(F.leaky_relu(x, 0.1), wt, b) + r) < 1e-5, ("lrelu in", n, ci, co, h, w)
    assert _lib.last_launch() == "conv_g1_kernel", _lib.last_launch()
    y = y0.clone()
    ops.conv(x, pc, out=y, accumulate=True, out_scale=1 / 3)
    assert rel(y, y0 + ref / 3) < 1e-5, ("accumulate", n, ci, co, h, w)
    # output / residual as channel slices of larger buffers (strided views), neighbours untouched
    big = torch.full((n, co + 2, h, w), 7.0)
    ops.conv(x, pc, out=big[:, 1:1 + co], bias=None)
    assert rel(big[:, 1:1 + co], ref) < 1e-5 and (big[:, 0] == 7).all() and (big[:, -1] == 7).all(), ("view", n, ci, co, h, w)
# no bias
x, wt = torch.randn(1, 64, 1, 256), torch.randn(96, 64, 1, 1) * 0.2
assert rel(ops.conv(x, ops.PackedConv(wt, None)), F.conv2d(x, wt)) < 1e-5
# the k = s = 2 transposed convolution of MDX-Net's up-sampling path: GEMM + scatter, additive and multiplicative skip
for (n, ci, co, h, w) in [(2, 24, 12, 5, 64), (1, 40, 50, 3, 132)]:
    wt = torch.randn(ci, co, 2, 2) * 0.2
    pt = ops.PackedConvTranspose(wt, torch.randn(co), stride=2)
    x, skip = torch.randn(n, ci, h, w), torch.randn(n, co, 2 * h, 2 * w)
    ref = F.conv_transpose2d(x, wt, pt.bias, stride=2)
    assert rel(ops.conv_transpose(x, pt, act=ops.ACT_RELU, mul=skip), F.relu(ref) * skip) < 1e-5, ("shuffle mul", n, ci, co, h, w)
    assert _lib.last_launch() == "conv_g1_kernel", _lib.last_launch()
    assert rel(ops.conv_transpose(x, pt, act=ops.ACT_RELU), F.relu(ref)) < 1e-5, ("shuffle", n, ci, co, h, w)
    assert rel(ops.conv_transpose(x, pt, add=skip), ref + skip) < 1e-5, ("shuffle add", n, ci, co, h, w)
print("g1 ok")
'''


@pytest.mark.parametrize("forced", ["2", "3", "4"])
def test_g1_tiles_in_a_subprocess(forced):
    """Each tile shape (128 x 256, 64 x 256, 192 x 256) forced on the emulator."""
    _run_child(CHILD, {"AICG_CONV_G1": forced}, "g1 ok")


@pytest.mark.gpu
@pytest.mark.parametrize("ci,co,t,act", [(768, 3072, 13216, ops.ACT_GELU), (3072, 768, 13216, ops.ACT_NONE), (768, 768, 26432, ops.ACT_NONE),
                                          (96, 256, 49152, ops.ACT_RELU), (200, 192, 60000, ops.ACT_NONE)])
def test_g1_on_gpu_at_policy_sizes(ci, co, t, act):
    """Shapes the product's policy sends to conv_g1 (HuBERT's per-token GEMMs at the benched token count among them), vs torch fp32 on CPU."""
    import conftest
    conftest._bind("hip")
    torch.manual_seed(ci + co)
    dev = torch.device("cuda:0")
    x, w, b, r = torch.randn(1, ci, t), torch.randn(co, ci, 1) * 0.05, torch.randn(co), torch.randn(1, co, t)
    pc = ops.PackedConv(w, b, device=dev)
    y = ops.conv(x.to(dev), pc, act=act, res=r.to(dev))
    assert _lib.last_launch() == "conv_g1_kernel", _lib.last_launch()
    ref = F.conv1d(x, w, b)
    ref = (F.gelu(ref) if act == ops.ACT_GELU else F.relu(ref) if act == ops.ACT_RELU else ref) + r
    assert rel_rms(y, ref) < 1e-5


@pytest.mark.gpu
def test_g1_shuffle_on_gpu_at_mdx_level_size():
    """MDX-Net's level-1 up-sampling layer (96 -> 4 x 48 rows, 128 x 1536 map, multiplicative skip) on a 2-image batch."""
    import conftest
    conftest._bind("hip")
    torch.manual_seed(11)
    dev = torch.device("cuda:0")
    wt = torch.randn(96, 48, 2, 2) * 0.1
    pt = ops.PackedConvTranspose(wt, torch.randn(48), stride=2, device=dev)
    x, skip = torch.randn(2, 96, 128, 1536), torch.randn(2, 48, 256, 3072)
    y = ops.conv_transpose(x.to(dev), pt, act=ops.ACT_RELU, mul=skip.to(dev))
    assert _lib.last_launch() == "conv_g1_kernel", _lib.last_launch()
    assert rel_rms(y, F.relu(F.conv_transpose2d(x, wt, pt.bias.cpu(), stride=2)) * skip) < 1e-5


def _g1_fuzz_case(dev, seed):
    import random
    rng = random.Random(seed)
    torch.manual_seed(seed)
    n = rng.choice([1, 1, 2, 3])
    ci = rng.choice([9, 12, 16, 24, 40, 100, 130])   # (<= 8 channels on one side: the pointwise streaming kernel takes the layer)
    co = rng.choice([33, 40, 64, 72, 128, 192, 200])
    two_d = rng.random() < 0.4
    h = rng.choice([2, 3, 5]) if two_d else 1
    w = 4 * rng.choice([1, 2, 16, 33, 64, 65, 130])
    tile = rng.choice([2, 3, 4])
    x, wt = torch.randn(n, ci, h, w), torch.randn(co, ci, 1, 1) * 0.2
    b = torch.randn(co) if rng.random() < 0.7 else None
    pc = ops.PackedConv(wt, b, device=dev.device)
    ref = F.conv2d(x, wt, b)
    mode = rng.choice(["plain", "gelu", "res", "res_first", "accum", "pre", "tanh"])
    xd = dev.t(x)
    ops.gemm_tile = tile
    try:
        if mode == "gelu":
            got, ref = ops.conv(xd, pc, act=ops.ACT_GELU), F.gelu(ref)
        elif mode == "tanh":
            got, ref = ops.conv(xd, pc, act=ops.ACT_TANH, out_scale=2.0), 2.0 * torch.tanh(ref)
        elif mode == "res":
            r = torch.randn_like(ref)
            got, ref = ops.conv(xd, pc, res=dev.t(r), act=ops.ACT_RELU), F.relu(ref) + r
        elif mode == "res_first":
            r = torch.randn_like(ref)
            got, ref = ops.conv(xd, pc, res=dev.t(r), act=ops.ACT_LRELU, act_slope=0.2, res_before_act=True), F.leaky_relu(ref + r, 0.2)
        elif mode == "pre":
            got, ref = ops.conv(xd, pc, pre_act=ops.ACT_LRELU, pre_slope=0.1), F.conv2d(F.leaky_relu(x, 0.1), wt, b)
        elif mode == "accum":
            y0 = torch.randn_like(ref)
            got = dev.t(y0.clone())
            ops.conv(xd, pc, out=got, accumulate=True, out_scale=0.5)
            ref = y0 + 0.5 * ref
        else:
            got = ops.conv(xd, pc)
        launched = _lib.last_launch()
    finally:
        ops.gemm_tile = 0
    return (n, ci, co, h, w, tile, mode), launched, rel_rms(got, ref)


@pytest.mark.parametrize("seed", range(40))
def test_g1_fuzz(dev, seed):
    """Seeded random 1 x 1 layers (ragged M / K, map sizes from one quad to several tiles with a tail, several images, every epilogue mode)
    on a forced tile of conv_g1 (aicg_conv_desc.gemm_tile: no process-wide switch needed), against torch."""
    desc, launched, err = _g1_fuzz_case(dev, seed)
    assert launched == "conv_g1_kernel", (desc, launched)
    assert err < 1e-5, (desc, err)
