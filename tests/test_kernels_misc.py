"""Elementwise / normalisation / attention / transposed-conv kernels vs plain torch fp32 statements of the same
reference ops (citations in include/aicg.h).  Tolerances: relative RMS <= 1e-5."""
import math

import pytest
import torch
import torch.nn.functional as F

from aicovergen_amd import ops
from conftest import rel_rms


@pytest.mark.parametrize("ci,co,k,s,T", [(64, 32, 16, 10, 50), (32, 16, 4, 2, 77), (48, 24, 20, 10, 31), (16, 8, 24, 12, 20)])
def test_conv_transpose1d(dev, ci, co, k, s, T):
    """GeneratorNSF.ups (models.py:453-463): k16 s10 / k4 s2 (40k), k20 s10 (32k/48k), k24 s12 (48k)."""
    torch.manual_seed(k * s)
    if dev.big:
        T *= 40
    x = torch.randn(1, ci, T)
    w = torch.randn(ci, co, k) * 0.1
    b = torch.randn(co)
    pt = ops.PackedConvTranspose(w, b, stride=s, padding=(k - s) // 2, device=dev.device)
    add = torch.randn(1, co, pt.out_hw(1, T)[1])
    y = ops.conv_transpose(dev.t(x), pt, add=dev.t(add), pre_act=ops.ACT_LRELU, pre_slope=0.1)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=(k - s) // 2) + add
    assert rel_rms(y, ref) < 1e-5


@pytest.mark.parametrize("n,ci,co,k,s,pad,opad,T", [(1, 16, 8, 4, 2, 1, 0, 300), (2, 16, 5, 16, 10, 3, 0, 257), (1, 16, 8, 5, 3, 1, 1, 513),
                                                    (1, 16, 8, 3, 2, 0, 1, 256), (3, 16, 4, 7, 1, 3, 0, 130), (1, 16, 8, 2, 4, 0, 0, 65)])
def test_conv_transpose1d_col2im_forms(dev, n, ci, co, k, s, pad, opad, T):
    """The input-centric 1-D col2im (csrc/elementwise.hip col2im1d_kernel): tiles of 256 input positions (several, with a tail), batches,
    kernels that are not a multiple of the stride, stride 1, stride > kernel (outputs that only see the bias), padding and
    output_padding, an activation in front of the `add` operand, an output written into a channel slice."""
    torch.manual_seed(k * 100 + s)
    x = torch.randn(n, ci, T)
    w = torch.randn(ci, co, k) * 0.1
    b = torch.randn(co)
    pt = ops.PackedConvTranspose(w, b, stride=s, padding=pad, output_padding=opad, device=dev.device)
    L = pt.out_hw(1, T)[1]
    ref = F.conv_transpose1d(x, w, b, stride=s, padding=pad, output_padding=opad)
    assert ref.shape[2] == L
    add = torch.randn(n, co, L)
    buf = dev.t(torch.full((n, co + 2, L), 3.0))
    y = ops.conv_transpose(dev.t(x), pt, add=dev.t(add), act=ops.ACT_LRELU, act_slope=0.2, out=buf[:, 1:1 + co])
    assert rel_rms(y, F.leaky_relu(ref, 0.2) + add) < 1e-5
    assert bool((buf[:, 0] == 3.0).all()) and bool((buf[:, -1] == 3.0).all())
    assert rel_rms(ops.conv_transpose(dev.t(x), pt), ref) < 1e-5


def test_conv_transpose2d(dev):
    """rmvpe.ResDecoderBlock.conv1 (rmvpe.py:147-155): 3x3 stride 2 pad 1 output_padding 1; MDX 2x2 stride 2."""
    torch.manual_seed(2)
    x = torch.randn(1, 16, 9, 8)
    w = torch.randn(16, 8, 3, 3) * 0.1
    pt = ops.PackedConvTranspose(w, None, stride=(2, 2), padding=(1, 1), output_padding=(1, 1), device=dev.device)
    y = ops.conv_transpose(dev.t(x), pt, act=ops.ACT_RELU)
    assert rel_rms(y, F.relu(F.conv_transpose2d(x, w, None, stride=2, padding=1, output_padding=1))) < 1e-5
    w = torch.randn(16, 8, 2, 2) * 0.1
    b = torch.randn(8)
    pt = ops.PackedConvTranspose(w, b, stride=2, device=dev.device)
    assert rel_rms(ops.conv_transpose(dev.t(x), pt), F.conv_transpose2d(x, w, b, stride=2)) < 1e-5


@pytest.mark.parametrize("C,with_res,T", [(192, True, 131), (768, True, 131), (1000, False, 131), (1100, True, 131),
                                          (192, False, 8219), (768, True, 8219), (1000, True, 8219)])
def test_layernorm_ct(dev, C, with_res, T):
    """Channel LayerNorm of (N, C, T): the register-resident kernel (C <= 256 / 768 / 1024) in its 8-column form (short maps: fewer
    than 512 workgroups of 32 columns) and its 32-column form (T = 8219), and the strided fallback (C = 1100); ragged T."""
    torch.manual_seed(3)
    if dev.big and T < 1000:
        T = 3001
    x, r = torch.randn(2, C, T) + 0.3, torch.randn(2, C, T)
    g, b = torch.rand(C) + 0.5, torch.randn(C)
    y = ops.layernorm_ct(dev.t(x), dev.t(g), dev.t(b), res=dev.t(r) if with_res else None)
    ref = F.layer_norm((x + r if with_res else x).transpose(1, 2), (C,), g, b, 1e-5).transpose(1, 2)
    assert rel_rms(y, ref) < 1e-5


@pytest.mark.parametrize("T", [1000, 4099, 16384 + 1, 40001])
def test_rownorm_gelu(dev, T):
    """HuBERT feature extractor layer 0: GroupNorm(512, 512) over time + GELU.  Rows of >= 4096 elements take the split form
    (partial moments per 16 384-element segment, merged; float4 body with <= 3 head / tail scalars: odd T moves every row's
    alignment); 1000: the one-workgroup-per-row form; a large mean against a small spread checks the moment merge."""
    torch.manual_seed(4)
    if dev.big:
        T = T * 5 + (T & 1)
    x = torch.randn(7, T) * 3 + 1
    x[3] = torch.randn(T) * 0.5 + 8.0
    g, b = torch.rand(7) + 0.5, torch.randn(7)
    y = ops.rownorm_act(dev.t(x), dev.t(g), dev.t(b), act=ops.ACT_GELU)
    ref = F.gelu(F.group_norm(x.double().unsqueeze(0), 7, g.double(), b.double(), 1e-5))[0]
    assert rel_rms(y, ref) < 5e-6
    y0 = ops.rownorm_act(dev.t(x), dev.t(g), dev.t(b), act=ops.ACT_NONE)
    assert rel_rms(y0, F.group_norm(x.double().unsqueeze(0), 7, g.double(), b.double(), 1e-5)[0]) < 5e-6
    assert rel_rms(ops.rownorm_act(dev.t(x), dev.t(g), dev.t(b), act=ops.ACT_RELU), F.relu(F.group_norm(x.unsqueeze(0), 7, g, b, 1e-5))[0]) < 1e-5


def test_gate_and_prior(dev):
    torch.manual_seed(5)
    a = torch.randn(1, 24, 50)
    assert rel_rms(ops.gate_tanh_sigmoid(dev.t(a)), torch.tanh(a[:, :12]) * torch.sigmoid(a[:, 12:])) < 1e-6
    st, nz = torch.randn(1, 8, 33), torch.randn(1, 4, 33)
    assert rel_rms(ops.prior_sample(dev.t(st), dev.t(nz), 0.66666), st[:, :4] + torch.exp(st[:, 4:]) * nz * 0.66666) < 1e-6


def test_feats_prepare(dev):
    """vc_infer_pipeline.py:433-452: nearest x2, clip to p_len, protect blend (bit-exact: copies and one fma)."""
    torch.manual_seed(6)
    f, f0 = torch.randn(20, 70), torch.randn(20, 70)
    pf = torch.rand(40) * 300
    pf[5:15] = 0

    def up(z):
        return F.interpolate(z.t().unsqueeze(0), scale_factor=2)[0].t()

    pff = pf.clone()
    pff[pf > 0] = 1
    pff[pf < 1] = 0.33
    ref = (up(f) * pff[:, None] + up(f0) * (1 - pff[:, None]))[:39].t().unsqueeze(0)
    y = ops.feats_prepare(dev.t(f), 39, dev.t(f0), dev.t(pf), 0.33)
    assert rel_rms(y, ref) < 1e-6
    assert torch.equal(ops.feats_prepare(dev.t(f), 40).cpu(), up(f).t().unsqueeze(0))


def test_sine_source_matches_oracle(dev):
    from oracle import synth
    torch.manual_seed(7)
    T, upp, sr = (600 if dev.big else 50), 400, 40000.0
    f0 = 110 * 2 ** (torch.rand(T) * 2)
    f0[10:20] = 0
    noise = torch.randn(T * upp)
    sd = {"dec.m_source.l_linear.weight": torch.tensor([[0.9]]), "dec.m_source.l_linear.bias": torch.tensor([0.01])}
    ref = synth.sine_source(sd, f0[None], upp, sr, noise[None])[0, 0]
    y = ops.sine_source(dev.t(f0), dev.t(noise), upp, sr, 0.9, 0.01)
    assert (y.cpu() - ref).abs().max() < 2e-6


@pytest.mark.parametrize("H,D,T,win", [(3, 64, 150, 0), (2, 96, 200, 10), (2, 32, 70, 4), (12, 64, 333, 0)])
def test_attention(dev, H, D, T, win):
    """Dense statement of attentions.MultiHeadAttention (banded relative keys/values) and of
    plain softmax attention (HuBERT) vs the fused kernel."""
    torch.manual_seed(T)
    if dev.big:
        T = T * 9 + 5
    C = H * D
    q, k, v = torch.randn(C, T) * 0.5, torch.randn(C, T) * 0.5, torch.randn(C, T)
    qh, kh, vh = (z.view(H, D, T).transpose(1, 2) for z in (q, k, v))
    sc = qh @ kh.transpose(1, 2)
    relk = ev = None
    if win:
        ek, ev = torch.randn(2 * win + 1, D) * D ** -0.5, torch.randn(2 * win + 1, D) * D ** -0.5
        rel = qh @ ek.t()
        relk = rel.transpose(1, 2).contiguous()
        i, j = torch.arange(T).view(T, 1), torch.arange(T).view(1, T)
        m = j - i + win
        band = (m >= 0) & (m <= 2 * win)
        sc = sc + torch.where(band, rel.gather(2, m.clamp(0, 2 * win).expand(H, T, T)), torch.zeros(()))
    pa = F.softmax(sc, -1)
    out = pa @ vh
    if win:
        pb = torch.zeros(H, T, 2 * win + 1)
        for mm in range(2 * win + 1):
            jj = torch.arange(T) + mm - win
            ok = (jj >= 0) & (jj < T)
            pb[:, ok, mm] = pa[:, torch.arange(T)[ok], jj[ok]]
        out = out + pb @ ev
    ref = out.transpose(1, 2).reshape(C, T)
    o = ops.attention(dev.t(q), dev.t(k), dev.t(v), H, relk=None if relk is None else dev.t(relk),
                      relv_emb=None if ev is None else dev.t(ev), window=win)
    assert rel_rms(o, ref) < 1e-5


def test_attention_online_softmax_rescale(dev):
    """A key whose score jumps far above every earlier tile forces the running-max rescale branch."""
    torch.manual_seed(9)
    H, D, T = 1, 64, 200
    q, k, v = torch.randn(D, T) * 0.1, torch.randn(D, T) * 0.1, torch.randn(D, T)
    k[:, 170] = q[:, 5] * 400.0  # spike: query 5 . key 170 >> everything else, 5 tiles in
    qh, kh, vh = q.t(), k.t(), v.t()
    ref = (F.softmax(qh @ kh.t(), -1) @ vh).t()
    o = ops.attention(dev.t(q), dev.t(k), dev.t(v), H)
    assert rel_rms(o, ref) < 1e-5


@pytest.mark.parametrize("win", [0, 10])
def test_attention_key_split_matches_single_pass(dev, win):
    """Split-K over the key range + log-sum-exp merge == the single-pass kernel (also for the lse the relative-value
    term consumes), including a split whose tiles end in the ragged last tile."""
    torch.manual_seed(21)
    H, D, T = 2, 96, (1205 if dev.big else 205)
    q, k, v = torch.randn(H * D, T) * 0.4, torch.randn(H * D, T) * 0.4, torch.randn(H * D, T)
    relk = ev = None
    if win:
        relk = dev.t(torch.randn(H, 2 * win + 1, T) * 0.3)
        ev = dev.t(torch.randn(2 * win + 1, D) * 0.1)
    one = ops.attention(dev.t(q), dev.t(k), dev.t(v), H, relk=relk, relv_emb=ev, window=win, n_splits=1)
    for s in (2, 7):
        many = ops.attention(dev.t(q), dev.t(k), dev.t(v), H, relk=relk, relv_emb=ev, window=win, n_splits=s)
        assert rel_rms(many, one.cpu()) < 2e-6


@pytest.mark.parametrize("R,K,O", [(300, 64, 40), (515, 100, 130), (128, 36, 256)])
def test_linear_last_nt_gemm(dev, R, K, O):
    """TDF linear over the last axis (NT GEMM, wave-specialised): ragged rows / K slab / output tile, with the fused
    bias + per-channel affine (eval BatchNorm2d) + ReLU + residual epilogue."""
    torch.manual_seed(R + K)
    if dev.big:
        R = R * 40 + 3
    n_ch, rows_per_ch = 5, 7
    x = torch.randn(R, K)
    w, b = torch.randn(O, K) * 0.2, torch.randn(O)
    sc, sh = torch.rand(n_ch) + 0.5, torch.randn(n_ch)
    res = torch.randn(R, O)
    ch = (torch.arange(R) // rows_per_ch) % n_ch
    ref = torch.relu((x @ w.t() + b) * sc[ch, None] + sh[ch, None]) + res
    out = torch.empty(R, O)
    from aicovergen_amd import _lib
    o = dev.t(out)
    xd, wd, bd, scd, shd, rd = (dev.t(t) for t in (x, w, b, sc, sh, res))
    st = torch.cuda.current_stream().cuda_stream if dev.kind == "hip" else 0
    _lib.call("aicg_gemm_nt", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), scd.data_ptr(), shd.data_ptr(), rd.data_ptr(),
              o.data_ptr(), R, K, O, K, K, O, O, rows_per_ch, n_ch, ops.ACT_RELU, st)
    assert rel_rms(o, ref) < 1e-5


def test_gelu_erf_over_the_whole_line(dev):
    """The branch-free erf behind every GELU epilogue (csrc/common.h fast_erff; fairseq's exact-erf "gelu", reference
    src/rvc.py:98-109 loads the model that uses it) against float64: 0.5 v (1 + erf(v / sqrt 2)) within 4e-7 x max(1, |result|) over [-9, 9] on a
    dense grid plus normal samples -- fp32 rounding of the product, i.e. erf itself within ~1 ulp."""
    torch.manual_seed(0)
    n = 3_000_000 if dev.big else 300_000
    v = torch.cat([torch.linspace(-9.0, 9.0, n), torch.randn(n) * 2.0]).view(1, 1, -1)
    y = ops.channel_affine(dev.t(v), dev.t(torch.ones(1)), dev.t(torch.zeros(1)), act=ops.ACT_GELU).cpu().double()
    ref = 0.5 * v.double() * (1.0 + torch.erf(v.double() / 2.0 ** 0.5))
    err = (y - ref).abs()
    assert float((err / ref.abs().clamp_min(1.0)).max()) < 4e-7, float((err / ref.abs().clamp_min(1.0)).max())   # (half an ulp of the result above 1)
