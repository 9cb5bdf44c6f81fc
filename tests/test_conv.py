"""Implicit-GEMM convolution kernel vs torch.nn.functional (fp32).  Tolerance: relative RMS <= 1e-5
(same fp32 products, different summation order -- the MFMA is a k-ordered fmaf chain)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from aicovergen_amd import _lib, ops
from conftest import rel_rms

# (Cin, Cout, k, stride, pad, dil, groups, T)  -- shapes taken from the hot path
CASES_1D = [
    (32, 32, 3, 1, 1, 1, 1, 300),      # ResBlock1 stage 3 (modules.py:229-296)
    (64, 64, 7, 1, 9, 3, 1, 260),      # ResBlock1 k7 d3
    (48, 40, 11, 1, 25, 5, 1, 200),    # k11 d5 (ragged channel counts)
    (192, 384, 5, 1, 2, 1, 1, 150),    # WN in_layer (modules.py:168-176)
    (20, 70, 1, 1, 0, 1, 1, 333),      # 1x1 / linear
    (16, 24, 10, 5, 0, 1, 1, 700),     # HuBERT feature extractor k10 s5
    (32, 32, 3, 2, 0, 1, 1, 301),      # HuBERT k3 s2
    (96, 96, 16, 1, 8, 1, 4, 130),     # grouped (HuBERT positional conv is k128 g16)
    (130, 200, 3, 1, 1, 1, 1, 70),     # FFN k3, channels not multiples of the tile
    (8, 1, 7, 1, 3, 1, 1, 500),        # conv_post: Cout = 1 (models.py:486)
]


def test_desc_struct_matches_c():
    torch.manual_seed(0)
    import conftest
    conftest._bind("emu")
    assert _lib.get().aicg_conv_desc_size() == ctypes.sizeof(ops.ConvDesc)


@pytest.mark.parametrize("ci,co,k,s,p,d,g,T", CASES_1D)
def test_conv1d(dev, ci, co, k, s, p, d, g, T):
    torch.manual_seed(ci * 1000 + co)
    if dev.big:
        T *= 9
    x = torch.randn(2, ci, T)
    w = torch.randn(co, ci // g, k) * 0.1
    b = torch.randn(co)
    pc = ops.PackedConv(w, b, stride=s, padding=p, dilation=d, groups=g, device=dev.device)
    y = ops.conv(dev.t(x), pc)
    ref = F.conv1d(x, w, b, stride=s, padding=p, dilation=d, groups=g)
    assert y.shape == ref.shape
    assert rel_rms(y, ref) < 1e-5


def test_conv_fused_resblock_step(dev):
    """lrelu prologue + bias + residual + 1/3 accumulate: one ResBlock1 conv as GeneratorNSF uses it
    (models.py:506-512: xs += resblock(x); x = xs / num_kernels)."""
    torch.manual_seed(3)
    T = 4000 if dev.big else 400
    x = torch.randn(1, 64, T)
    w = torch.randn(64, 64, 3) * 0.1
    b = torch.randn(64)
    y0 = torch.randn(1, 64, T)
    pc = ops.PackedConv(w, b, padding=3, dilation=3, device=dev.device)
    y = dev.t(y0.clone())
    xd = dev.t(x)
    ops.conv(xd, pc, res=xd, out=y, pre_act=ops.ACT_LRELU, pre_slope=0.1, out_scale=1 / 3, accumulate=True)
    ref = y0 + (F.conv1d(F.leaky_relu(x, 0.1), w, b, padding=3, dilation=3) + x) / 3
    assert rel_rms(y, ref) < 1e-5


@pytest.mark.parametrize("act,fn", [(ops.ACT_GELU, F.gelu), (ops.ACT_RELU, F.relu), (ops.ACT_TANH, torch.tanh),
                                    (ops.ACT_SIGMOID, torch.sigmoid)])
def test_conv_epilogue_activations(dev, act, fn):
    torch.manual_seed(5)
    x = torch.randn(1, 40, 200)
    w = torch.randn(24, 40, 3) * 0.2
    b = torch.randn(24)
    pc = ops.PackedConv(w, b, padding=1, device=dev.device)
    y = ops.conv(dev.t(x), pc, act=act)
    assert rel_rms(y, fn(F.conv1d(x, w, b, padding=1))) < 1e-5


CASES_2D = [  # Cin, Cout, kh, kw, stride, pad, H, W
    (16, 32, 3, 3, 1, 1, 40, 128),   # RMVPE level-0 ConvBlockRes (rmvpe.py:27-45)
    (32, 16, 3, 3, 1, 1, 33, 4),     # deepest RMVPE level: only 4 mel columns left
    (8, 8, 3, 3, 1, 1, 5, 20),
    (24, 48, 2, 2, 2, 0, 16, 64),    # MDX-Net downsample 2x2 s2
    (4, 48, 1, 1, 1, 0, 9, 96),      # MDX-Net first 1x1
]


@pytest.mark.parametrize("ci,co,kh,kw,s,p,H,W", CASES_2D)
def test_conv2d(dev, ci, co, kh, kw, s, p, H, W):
    torch.manual_seed(H * W)
    x = torch.randn(2, ci, H, W)
    w = torch.randn(co, ci, kh, kw) * 0.1
    b = torch.randn(co)
    pc = ops.PackedConv(w, b, stride=s, padding=p, device=dev.device)
    y = ops.conv(dev.t(x), pc, act=ops.ACT_RELU)
    assert rel_rms(y, F.relu(F.conv2d(x, w, b, stride=s, padding=p))) < 1e-5


@pytest.mark.parametrize("n,ci,co,H,W,act", [(2, 8, 32, 4, 64, ops.ACT_RELU), (1, 20, 48, 7, 70, ops.ACT_RELU), (1, 16, 96, 5, 130, ops.ACT_NONE),
                                              (1, 40, 144, 9, 66, ops.ACT_RELU), (1, 24, 64, 3, 2, ops.ACT_NONE), (2, 12, 48, 11, 70, ops.ACT_RELU),
                                              (1, 16, 40, 8, 64, ops.ACT_NONE), (1, 8, 240, 3, 36, ops.ACT_RELU), (1, 16, 48, 9, 200, ops.ACT_RELU),
                                              (1, 12, 96, 5, 260, ops.ACT_NONE)])
def test_conv2d_winograd_rows(dev, monkeypatch, n, ci, co, H, W, act):
    """F(2, 3) along rows (csrc/conv_ws3w.h) on TFC-shaped layers: every tile height (32 / 64 / 96 output channels on the 32 x 32 x 2
    MFMA, 48 -- also 144 = 3 x 48 and 240 = 5 x 48 -- on 16 x 16 x 4 with eight output rows a workgroup), ragged rows,
    column tiles (interior ones take 16-byte patch loads) and channel chunks, output written into a channel slice.  Same products up to the exact 1/2 of G; the sums of the
    transformed operands round differently from the direct form, hence 2e-6 rather than bit equality."""
    monkeypatch.setattr(ops, "winograd_min_positions", 1)
    monkeypatch.setattr(ops, "winograd2d", False)          # the row form (the two-dimensional one: test_conv2d_winograd_2d)
    torch.manual_seed(H * W + ci)
    x = torch.randn(n, ci, H, W)
    w = torch.randn(co, ci, 3, 3) * 0.1
    b = torch.randn(co)
    pc = ops.PackedConv(w, b, padding=1, device=dev.device)
    assert pc.w_wino is not None
    buf = dev.t(torch.full((n, co + 3, H, W), 7.0))
    ops.conv(dev.t(x), pc, act=act, out=buf[:, 2:2 + co])
    ref = F.conv2d(x, w, b, padding=1)
    assert rel_rms(buf[:, 2:2 + co], F.relu(ref) if act == ops.ACT_RELU else ref) < 2e-6
    assert (buf[:, :2] == 7).all() and (buf[:, 2 + co:] == 7).all()
    # the epilogues the form does not cover stay on the direct kernels (residual here)
    r = torch.randn(n, co, H, W)
    y = ops.conv(dev.t(x), pc, res=dev.t(r))
    assert rel_rms(y, ref + r) < 1e-5


@pytest.mark.parametrize("waves,quads", [(8, False), (4, False), (8, True), (4, True), (8, "pairs"), (4, "pairs2"), (8, "untied")])
@pytest.mark.parametrize("n,ci,co,H,W,act", [(1, 8, 48, 8, 64, ops.ACT_RELU), (2, 16, 48, 11, 72, ops.ACT_NONE), (1, 24, 96, 5, 132, ops.ACT_RELU),
                                              (1, 12, 48, 17, 60, ops.ACT_RELU), (3, 40, 144, 3, 8, ops.ACT_NONE), (1, 48, 48, 16, 196, ops.ACT_RELU)])
def test_conv2d_winograd_2d(dev, monkeypatch, waves, quads, n, ci, co, H, W, act):
    """F(2 x 2, 3 x 3) (csrc/conv_w2d.h) on TFC-shaped layers, in both workgroup forms: one and several M units of 48 channels, input
    channels that are not a multiple of the 8-channel chunk (zeros from the buffer range check / the padded image), odd row counts
    (a row pair whose second row does not exist), ragged and sub-tile widths (multiples of 4), several tiles per workgroup in the
    persistent walk, several images, output written into a channel slice.  Same products up to the exact 1/2 and 1/4 of G g G^T; the
    sums of the transformed operands round differently from the direct form, hence 2e-6 rather than bit equality."""
    if quads in ("pairs2", "untied") and dev.kind == "hip":
        from aicovergen_amd import _lib
        if not _lib.get_path().endswith("_dev.so"):
            pytest.skip("the two-workgroups-per-CU form of conv_w2d (wino 16: measured level with the routed one) is compiled into development builds only")
    monkeypatch.setattr(ops, "winograd_min_positions", 1)
    monkeypatch.setattr(ops, "winograd2d", True)
    monkeypatch.setattr(ops, "winograd2d_waves", waves)
    monkeypatch.setattr(ops, "winograd2d_quads", quads is True)      # fragment image [s][p / 4][ks][m][p % 4]: one 16-byte read per four MFMAs
    # "pairs": [s][p / 2][ks][m][p % 2], one 8-byte read per two MFMAs (aicg_conv_desc.wino 12: the routed default of the eight-wave form)
    # "pairs2": four waves, pair fragments, stages of FOUR input channels (one k-step each): two workgroups per CU (wino 16)
    monkeypatch.setattr(ops, "winograd2d_pairs", quads == "pairs")
    # "untied": the pair form with its MFMAs through the builtin instead of inline asm with the destination tied to the addend (wino 14, development builds)
    monkeypatch.setattr(ops, "winograd2d_code", 16 if quads == "pairs2" else 14 if quads == "untied" else 0)
    assert ops._w2d_code() == {(8, False): (2, "dword"), (4, False): (3, "dword"), (8, True): (4, "quads"), (4, True): (5, "quads"),
                               (8, "pairs"): (12, "pairs"), (4, "pairs2"): (16, "pairs"), (8, "untied"): (14, "pairs")}[(waves, quads)]
    torch.manual_seed(H * W + ci)
    x = torch.randn(n, ci, H, W)
    w = torch.randn(co, ci, 3, 3) * 0.1
    b = torch.randn(co)
    pc = ops.PackedConv(w, b, padding=1, device=dev.device)
    assert pc.w_wino2 is not None
    prof = ops.conv_profile = ops.ConvProfile()
    try:
        buf = dev.t(torch.full((n, co + 3, H, W), 7.0))
        ops.conv(dev.t(x), pc, act=act, out=buf[:, 2:2 + co])
    finally:
        ops.conv_profile = None
    ref = F.conv2d(x, w, b, padding=1)
    assert rel_rms(buf[:, 2:2 + co], F.relu(ref) if act == ops.ACT_RELU else ref) < 2e-6
    assert (buf[:, :2] == 7).all() and (buf[:, 2 + co:] == 7).all()
    # a width that is not a multiple of 4 cannot be staged by 16-byte DMA: the row form takes the layer
    x2 = torch.randn(1, ci, 6, 66)
    y2 = ops.conv(dev.t(x2), pc, act=act)
    r2 = F.conv2d(x2, w, b, padding=1)
    assert rel_rms(y2, F.relu(r2) if act == ops.ACT_RELU else r2) < 2e-6


def test_winograd2d_image_layout():
    """ops.winograd2d_image against a literal loop over the definition in include/aicg.h."""
    torch.manual_seed(0)
    w = torch.randn(96, 11, 3, 3)
    img = ops.winograd2d_image(w).view(2, 2, 2, 16, 4, 48)
    G = torch.tensor([[1.0, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1.0]])
    for (mu, ch, s, pnt, ks, m) in [(0, 0, 0, 0, 0, 0), (1, 1, 0, 5, 2, 47), (0, 1, 1, 15, 3, 13), (1, 0, 1, 9, 1, 30)]:
        ci = 8 * ch + 4 * s + ks
        want = (G @ w[48 * mu + m, ci] @ G.t())[pnt // 4, pnt % 4] if ci < 11 else torch.tensor(0.0)
        assert torch.allclose(img[mu, ch, s, pnt, ks, m], want, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["rows", "2d8", "2d4"])
@pytest.mark.parametrize("n,c,h,w", [(2, 48, 256, 3072), (2, 96, 128, 1536), (6, 144, 64, 768), (2, 192, 32, 384), (2, 240, 16, 192)])
def test_winograd_at_mdx_level_sizes(n, c, h, w, form):
    """The Winograd forms -- F(2, 3) along rows (conv_ws3w.h) and F(2 x 2, 3 x 3) with eight or four waves (conv_w2d.h) -- at the
    real MDX-Net level shapes (every tile kind; batch 6 on the 144-channel level so that the map has >= 1024 four-row tiles and the
    row form's 96 + 48-row split really runs on hardware -- ADVICE r3; interior 16-byte patch loads; the persistent walks over > 1
    tile per workgroup; 1 .. 5 M units of 48 channels), against (a) the direct HIP kernels over the whole map and (b) torch's
    fp32 convolution on the host over two sub-maps -- the top-left corner and the bottom-right one, all output channels, so padding,
    ragged last tiles and every channel tile are covered by an independent reference."""
    import conftest
    conftest._bind("hip")
    torch.manual_seed(c)
    x = torch.randn(n, c, h, w, device="cuda")
    wt = torch.randn(c, c, 3, 3) * 0.05
    bias = torch.randn(c) * 0.1
    pc = ops.PackedConv(wt, bias, padding=1, device="cuda")
    assert pc.w_wino is not None and pc.w_wino2 is not None
    old = ops.winograd_min_positions, ops.winograd2d, ops.winograd2d_waves
    try:
        ops.winograd_min_positions = 1 << 60
        ref = ops.conv(x, pc, act=ops.ACT_RELU)
        ops.winograd_min_positions = 1
        ops.winograd2d, ops.winograd2d_waves = form != "rows", 4 if form == "2d4" else 8
        got = ops.conv(x, pc, act=ops.ACT_RELU)
    finally:
        ops.winograd_min_positions, ops.winograd2d, ops.winograd2d_waves = old
    assert not torch.equal(got, ref)          # a different summation order: the other kernel really ran
    e_direct = rel_rms(got, ref)
    rh, rw = min(h, 10), min(w, 72)
    xc = x.cpu()
    tl = F.relu(F.conv2d(F.pad(xc[:, :, :rh + 1, :rw + 1], (1, 0, 1, 0)), wt, bias))[:, :, :rh, :rw]
    br = F.relu(F.conv2d(F.pad(xc[:, :, h - rh - 1:, w - rw - 1:], (0, 1, 0, 1)), wt, bias))
    e_tl, e_br = rel_rms(got[:, :, :rh, :rw], tl), rel_rms(got[:, :, h - rh:, w - rw:], br)
    print("winograd %s C%d %dx%d: vs direct HIP %.2e, vs torch fp32 corner maps %.2e / %.2e" % (form, c, h, w, e_direct, e_tl, e_br))
    assert e_direct < 4e-6 and e_tl < 4e-6 and e_br < 4e-6


def test_conv_strided_views(dev):
    """Outputs may be channel slices of a larger buffer (decoder concat without a copy, rmvpe.py:166)."""
    torch.manual_seed(11)
    x = torch.randn(1, 16, 10, 32)
    w = torch.randn(8, 16, 3, 3) * 0.1
    buf = dev.t(torch.zeros(1, 20, 10, 32))
    pc = ops.PackedConv(w, None, padding=1, device=dev.device)
    ops.conv(dev.t(x), pc, out=buf[:, 4:12])
    assert rel_rms(buf[:, 4:12], F.conv2d(x, w, padding=1)) < 1e-5
    assert buf[:, :4].abs().max() == 0 and buf[:, 12:].abs().max() == 0


def test_conv_rejects_inconsistent_geometry(dev):
    x = dev.t(torch.zeros(1, 8, 16))
    pc = ops.PackedConv(torch.zeros(8, 8, 3), None, padding=1, device=dev.device)
    with pytest.raises(AssertionError):
        ops.conv(x, pc, out=dev.t(torch.zeros(1, 8, 15)))


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 4, 48, 7, 64), (2, 48, 4, 7, 64), (1, 8, 8, 3, 12), (1, 3, 20, 1, 128)])
def test_pointwise_streaming_form(dev, n, ci, co, h, w):
    """1x1 layers with <= 8 channels on one side (MDX-Net stem 4 -> 48 and head 48 -> 4) take the float4 streaming kernel:
    fused bias / residual (either side of the activation) / scale, and channel-slice views on both ends."""
    torch.manual_seed(ci * 100 + co)
    if dev.big:
        h, w = h * 9, w * 8
    x, wt, b, r = torch.randn(n, ci, h, w), torch.randn(co, ci, 1, 1) * 0.3, torch.randn(co), torch.randn(n, co, h, w)
    pc = ops.PackedConv(wt, b, device=dev.device)
    assert rel_rms(ops.conv(dev.t(x), pc, act=ops.ACT_RELU, res=dev.t(r)), F.relu(F.conv2d(x, wt, b)) + r) < 1e-6
    y = ops.conv(dev.t(x), pc, act=ops.ACT_LRELU, act_slope=0.2, res=dev.t(r), res_before_act=True, out_scale=0.5)
    assert rel_rms(y, 0.5 * F.leaky_relu(F.conv2d(x, wt, b) + r, 0.2)) < 1e-6
    big, outbig = dev.t(torch.randn(n, ci + 6, h, w)), dev.t(torch.zeros(n, co + 12, h, w))
    ops.conv(big[:, 2:2 + ci], pc, out=outbig[:, 8:8 + co])
    assert rel_rms(outbig[:, 8:8 + co], F.conv2d(big.cpu()[:, 2:2 + ci], wt, b)) < 1e-6
    assert float(outbig[:, :8].abs().sum()) == 0 and float(outbig[:, 8 + co:].abs().sum()) == 0


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 24, 12, 5, 9), (1, 96, 48, 16, 96), (1, 20, 12, 130, 520), (1, 64, 32, 64, 520)])
def test_conv_transpose_k2s2_fused_epilogue(dev, n, ci, co, h, w):
    """kernel = stride = 2 ConvTranspose2d (MDX-Net `us.*`): the GEMM epilogue scatters to the 2x upsampled grid and applies
    bias + ReLU + the multiplicative U-Net skip (or an additive one); small, 32x32-tile and 16x16-tile dispatches."""
    torch.manual_seed(co)
    x, wt, b = torch.randn(n, ci, h, w), torch.randn(ci, co, 2, 2) * 0.2, torch.randn(co)
    skip = torch.randn(n, co, 2 * h, 2 * w)
    pt = ops.PackedConvTranspose(wt, b, stride=2, device=dev.device)
    ref = F.conv_transpose2d(x, wt, b, stride=2)
    assert rel_rms(ops.conv_transpose(dev.t(x), pt, act=ops.ACT_RELU, mul=dev.t(skip)), F.relu(ref) * skip) < 1e-5
    assert rel_rms(ops.conv_transpose(dev.t(x), pt, add=dev.t(skip)), ref + skip) < 1e-5
    assert rel_rms(ops.conv_transpose(dev.t(x), pt), ref) < 1e-5


def test_single_role_fallback_kernels_in_a_subprocess():
    """conv_mfma_kernel / conv_mfma16_kernel are the fallbacks when a layer does not fit the wave-specialised kernels' LDS
    budget; the dispatch switches are read once per process, so they are exercised in a child process on the emulator."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch, torch.nn.functional as F
sys.path.insert(0, %r); sys.path.insert(0, %r)
import conftest
conftest._bind("emu")
from aicovergen_amd import ops
torch.manual_seed(0)
def rel(a, b): return float(((a - b).pow(2).sum() / b.pow(2).sum()).sqrt())
for (n, ci, co, h, w, k) in [(1, 20, 40, 70, 1000, 3), (1, 16, 16, 1, 70000, 3), (2, 33, 64, 9, 130, 3), (1, 4, 48, 8, 64, 1)]:
    x, wt, b, r = torch.randn(n, ci, h, w), torch.randn(co, ci, k if h > 1 else 1, k) * 0.1, torch.randn(co), torch.randn(n, co, h, w)
    pc = ops.PackedConv(wt, b, padding=(k // 2 if h > 1 else 0, k // 2))
    y = ops.conv(x, pc, act=ops.ACT_RELU, res=r)
    e = rel(y, F.relu(F.conv2d(x, wt, b, padding=(k // 2 if h > 1 else 0, k // 2))) + r)
    assert e < 1e-5, (n, ci, co, h, w, k, e)
wt = torch.randn(24, 12, 2, 2) * 0.2
pt = ops.PackedConvTranspose(wt, torch.randn(12), stride=2)
x, skip = torch.randn(2, 24, 5, 9), torch.randn(2, 12, 10, 18)
assert rel(ops.conv_transpose(x, pt, act=ops.ACT_RELU, mul=skip), F.relu(F.conv_transpose2d(x, wt, pt.bias, stride=2)) * skip) < 1e-5
print("fallback kernels ok")
''' % (root, os.path.join(root, "tests"))
    env = dict(os.environ, AICG_CONV_WS="0", AICG_CONV_POINTWISE="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "fallback kernels ok" in out.stdout, out.stdout + out.stderr


def _run_child(code, env_extra, token):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pre = "import sys, torch, torch.nn.functional as F\nsys.path.insert(0, %r); sys.path.insert(0, %r)\nimport conftest\n" \
          "conftest._bind('emu')\nfrom aicovergen_amd import ops\ntorch.manual_seed(0)\n" \
          "def rel(a, b): return float(((a - b).pow(2).sum() / b.pow(2).sum()).sqrt())\n" % (root, os.path.join(root, "tests"))
    out = subprocess.run([sys.executable, "-c", pre + code], env=dict(os.environ, **env_extra), capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and token in out.stdout, out.stdout + out.stderr


def test_winograd_row_split_in_a_subprocess():
    """144 / 240 output channels on a large map: aicg_conv_forward runs the 96-row Winograd tiles on all but the last 48 rows and the
    48-row kernel on those, through shifted bias / output / weight-image pointers (AICG_WINO_SPLIT_TILES lowers the map-size gate)."""
    code = r'''
ops.winograd_min_positions = 1
for (n, ci, co, h, w) in [(1, 16, 144, 6, 130), (2, 12, 240, 5, 66)]:
    x, wt, b = torch.randn(n, ci, h, w), torch.randn(co, ci, 3, 3) * 0.1, torch.randn(co)
    pc = ops.PackedConv(wt, b, padding=1)
    buf = torch.full((n, co + 2, h, w), 3.0)
    ops.conv(x, pc, act=ops.ACT_RELU, out=buf[:, 1:1 + co])
    e = rel(buf[:, 1:1 + co], F.relu(F.conv2d(x, wt, b, padding=1)))
    assert e < 2e-6 and (buf[:, 0] == 3).all() and (buf[:, -1] == 3).all(), (n, ci, co, h, w, e)
print("winograd split ok")
'''
    _run_child(code, {"AICG_WINO_SPLIT_TILES": "1"}, "winograd split ok")


def test_16x16x4_fragment_kernels_in_a_subprocess():
    """conv_ws3m16_kernel (16-byte fragments on v_mfma_f32_16x16x4_f32; opt-in through AICG_CONV_V3M16=1: 48- and 16-row layers with
    >= 65 536 positions, the MDX-Net / RMVPE level-0 shapes) incl. a channel tail (40 of 48 channels in the last K chunk) and a
    ragged last tile."""
    code = r'''
for (ci, co, h, w, k) in [(16, 16, 1, 70000, 3), (32, 48, 6, 11000, 1), (40, 40, 3, 22000, 3)]:
    x = torch.randn(1, ci, h, w)
    wt = torch.randn(co, ci, k if h > 1 else 1, k) * 0.1
    b, r = torch.randn(co), torch.randn(1, co, h, w)
    pad = (k // 2 if h > 1 else 0, k // 2)
    pc = ops.PackedConv(wt, b, padding=pad)
    y = ops.conv(x, pc, act=ops.ACT_RELU, res=r)
    e = rel(y, F.relu(F.conv2d(x, wt, b, padding=pad)) + r)
    assert e < 1e-5, (ci, co, h, w, k, e)
print("m16 fragment kernels ok")
'''
    _run_child(code, {"AICG_CONV_V3M16": "1", "AICG_CONV_M16H": "0"}, "m16 fragment kernels ok")


def test_classic_wave_specialised_kernels_in_a_subprocess():
    """With AICG_CONV_V3=0 / AICG_CONV_V3M16=0 the dispatcher uses the 4-byte-fragment kernels (conv_ws_kernel, conv_ws16_kernel),
    which otherwise only serve layers with < 8 input channels per group and the 160-row tile."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch, torch.nn.functional as F
sys.path.insert(0, %r); sys.path.insert(0, %r)
import conftest
conftest._bind("emu")
from aicovergen_amd import ops
torch.manual_seed(0)
def rel(a, b): return float(((a - b).pow(2).sum() / b.pow(2).sum()).sqrt())
for (n, ci, co, h, w, k) in [(1, 24, 96, 12, 200, 3), (1, 16, 16, 1, 70000, 3), (2, 33, 64, 9, 130, 3), (1, 64, 144, 1, 900, 5)]:
    x, wt, b, r = torch.randn(n, ci, h, w), torch.randn(co, ci, k if h > 1 else 1, k) * 0.1, torch.randn(co), torch.randn(n, co, h, w)
    pc = ops.PackedConv(wt, b, padding=(k // 2 if h > 1 else 0, k // 2))
    y = ops.conv(x, pc, act=ops.ACT_RELU, res=r)
    e = rel(y, F.relu(F.conv2d(x, wt, b, padding=(k // 2 if h > 1 else 0, k // 2))) + r)
    assert e < 1e-5, (n, ci, co, h, w, k, e)
print("classic kernels ok")
''' % (root, os.path.join(root, "tests"))
    env = dict(os.environ, AICG_CONV_V3="0", AICG_CONV_V3M16="0", AICG_CONV_M16H="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "classic kernels ok" in out.stdout, out.stdout + out.stderr


def test_160_row_fragment_tile_in_a_subprocess():
    """conv_ws3_kernel<160, 128, ...>: 5 accumulator tiles per wave, ONE set of A fragments reloaded row by row behind its MFMAs
    (r2).  Driven on the emulator by lowering the fill target; 144 and 150 output channels (ragged last 32-row tile), taps split over
    stages, channel tail, fused epilogue."""
    code = r'''
for (ci, co, k, d, T) in [(48, 144, 3, 1, 300), (40, 150, 7, 2, 260), (144, 144, 1, 1, 200)]:
    x = torch.randn(2, ci, T)
    w = torch.randn(co, ci, k) * 0.1
    b, r = torch.randn(co), torch.randn(2, co, T)
    pc = ops.PackedConv(w, b, padding=(k - 1) * d // 2, dilation=d)
    y = ops.conv(x, pc, act=ops.ACT_RELU, res=r, pre_act=ops.ACT_LRELU, pre_slope=0.1)
    e = rel(y, F.relu(F.conv1d(F.leaky_relu(x, 0.1), w, b, padding=(k - 1) * d // 2, dilation=d)) + r)
    assert e < 1e-5, (ci, co, k, d, T, e)
x = torch.randn(1, 144, 20, 70)
w = torch.randn(144, 144, 3, 3) * 0.05
pc = ops.PackedConv(w, None, padding=1)
e = rel(ops.conv(x, pc), F.conv2d(x, w, None, padding=1))
assert e < 1e-5, e
print("160-row fragment tile ok")
'''
    _run_child(code, {"AICG_CONV_WANT": "1"}, "160-row fragment tile ok")


def test_16x16x4_on_8_byte_fragments_in_a_subprocess():
    """conv_ws3m16h_kernel (r2 default for 48- / 16-row layers with >= 8 input channels and >= 65 536 positions: MDX-Net / RMVPE level
    0): lane (r16, q) reads half a quad of plane q & 1; channel tail (40 of 48), ragged last tile, 1-D and 2-D, fused epilogue."""
    code = r'''
for (ci, co, h, w, k) in [(16, 16, 1, 70000, 3), (32, 48, 6, 11000, 1), (40, 40, 3, 22000, 3), (8, 48, 2, 33000, 3)]:
    x = torch.randn(1, ci, h, w)
    wt = torch.randn(co, ci, k if h > 1 else 1, k) * 0.1
    b, r = torch.randn(co), torch.randn(1, co, h, w)
    pad = (k // 2 if h > 1 else 0, k // 2)
    pc = ops.PackedConv(wt, b, padding=pad)
    y = ops.conv(x, pc, act=ops.ACT_RELU, res=r)
    e = rel(y, F.relu(F.conv2d(x, wt, b, padding=pad)) + r)
    assert e < 1e-5, (ci, co, h, w, k, e)
print("m16 half-quad kernels ok")
'''
    _run_child(code, {"AICG_CONV_M16H": "1"}, "m16 half-quad kernels ok")
