"""Opt-in fp16 matrix arithmetic for the RVC half (AICG_HALF=1 + is_half=True; reference: src/main.py:196 asks for is_half, src/rvc.py:103-104
and :137-138 then run HuBERT and the synthesizer in fp16 on a GPU).  Here: aicg_conv_desc.split == 2 -- the LDS-DMA staged kernels
(csrc/conv_g1.h: 1 x 1 GEMMs; csrc/conv_g1w.h: the vocoder's k = 3 / 7 / 11 ResBlock layers in the Winograd F(2, 3) form) round their
operands to fp16 (round to nearest even) in registers in front of v_mfma_f32_32x32x8_f16; activations in HBM, accumulation and epilogue
stay fp32.

Tolerances (written where they are asserted):
  * 1 x 1 GEMM against torch on operands rounded to fp16 the same way: 1e-5 relative rms (only the summation order differs);
  * the same against the fp32 layer: 2e-3 (fp16 has 11 significant bits: 2^-11 / sqrt(3) per operand, two operands);
  * Winograd layers against the fp32 convolution: 4e-3 (the TRANSFORMED operands are what is rounded);
  * a whole synthesizer against its fp32 self: 2e-2 relative rms on the waveform (the reference's own fp16 GPU path is no closer to its
    fp32 path: every activation is fp16 there)."""
import os

import pytest
import torch
import torch.nn.functional as F

from aicovergen_amd import _lib, ops
from conftest import rel_rms


def _h(t):
    return t.half().float()


@pytest.mark.parametrize("ci,co,T,pre", [(64, 96, 520, False), (80, 200, 1028, True), (768, 128, 260, False), (24, 40, 132, False)])
def test_g1_fp16_operands(dev, ci, co, T, pre):
    """conv_g1 with fp16 operands = the GEMM of the rounded operands, to summation order."""
    torch.manual_seed(ci + co)
    x, w, b, r = torch.randn(2, ci, T), torch.randn(co, ci, 1) * 0.1, torch.randn(co), torch.randn(2, co, T)
    pc = ops.PackedConv(w, b, device=dev.device)
    ops.mark_half(pc)
    assert pc.f16
    xd = dev.t(x)
    got = ops.conv(xd, pc, res=dev.t(r), pre_act=ops.ACT_LRELU if pre else ops.ACT_NONE, pre_slope=0.1)
    assert _lib.last_launch() == "conv_g1_kernel"
    xin = F.leaky_relu(x, 0.1) if pre else x
    ref16 = F.conv1d(_h(xin).double(), _h(w).double(), b.double()).float() + r
    ref32 = F.conv1d(xin.double(), w.double(), b.double()).float() + r
    assert rel_rms(got, ref16) < 1e-5
    e = rel_rms(got, ref32)
    assert 1e-5 < e < 2e-3, e                     # (it IS the fp16 arithmetic that ran)
    ops.mark_half(pc, False)
    assert rel_rms(ops.conv(xd, pc, res=dev.t(r), pre_act=ops.ACT_LRELU if pre else ops.ACT_NONE, pre_slope=0.1), ref32) < 2e-6


@pytest.mark.parametrize("k,d", [(3, 1), (7, 1), (11, 1), (5, 1), (3, 3), (7, 5), (11, 3)])
def test_g1w_fp16_operands(dev, k, d):
    """The vocoder's ResBlock step x + conv(lrelu(x)) and the accumulating form, fp16 operands on the Winograd kernel."""
    torch.manual_seed(10 * k + d)
    T = 1304 if dev.big else 392
    old_min, ops.winograd1d_min_positions = ops.winograd1d_min_positions, 1
    try:
        for (n, ci, co, mode) in ((1, 64, 64, "resblock"), (2, 48, 72, "accum"), (1, 16, 33, "plain")):
            x, w, b = torch.randn(n, ci, T), torch.randn(co, ci, k) * 0.2, torch.randn(co)
            pad = (k - 1) // 2 * d
            pc = ops.PackedConv(w, b, padding=pad, dilation=d, device=dev.device)
            assert pc.w_wino1 is not None
            ops.mark_half(pc)
            conv = lambda t: F.conv1d(t.double(), w.double(), b.double(), padding=pad, dilation=d).float()
            xd = dev.t(x)
            if mode == "resblock":
                got, ref = ops.conv(xd, pc, res=xd, pre_act=ops.ACT_LRELU, pre_slope=0.1), conv(F.leaky_relu(x, 0.1)) + x
            elif mode == "accum":
                y0, r = torch.randn(n, co, T), torch.randn(n, co, T)
                got = dev.t(y0.clone())
                ops.conv(xd, pc, res=dev.t(r), out=got, pre_act=ops.ACT_LRELU, pre_slope=0.1, accumulate=True, out_scale=1 / 3)
                ref = y0 + (conv(F.leaky_relu(x, 0.1)) + r) / 3
            else:
                got, ref = ops.conv(xd, pc), conv(x)
            assert _lib.last_launch() == "conv_g1w_kernel"
            e = rel_rms(got, ref)
            assert 1e-5 < e < 4e-3, (mode, e)
            ops.mark_half(pc, False)
            if mode != "accum":
                again = ops.conv(xd, pc, res=xd, pre_act=ops.ACT_LRELU, pre_slope=0.1) if mode == "resblock" else ops.conv(xd, pc)
                assert rel_rms(again, ref) < 2e-6
    finally:
        ops.winograd1d_min_positions = old_min


def test_mark_half_skips_fp32_only_and_split_layers():
    w = torch.randn(48, 32, 3)
    with ops.fp32_layers():
        f0_layer = ops.PackedConv(w, None, padding=1, device=torch.device("cpu"))
    plain = ops.PackedConv(w, None, padding=1, device=torch.device("cpu"))
    tree = {"a": [f0_layer, plain], "b": (ops.PackedConvTranspose(torch.randn(32, 16, 4), None, stride=2, padding=1, device=torch.device("cpu")),)}
    ops.mark_half(tree)
    assert plain.f16 and tree["b"][0].gemm.f16 and not f0_layer.f16
    ops.mark_half(tree, False)
    assert not plain.f16 and not tree["b"][0].gemm.f16


def test_half_is_a_no_op_without_the_switch(dev, monkeypatch):
    """is_half=True alone (what src/main.py:196 passes) changes nothing: .half() marks layers only under AICG_HALF=1."""
    from aicovergen_amd.infer_pack.models import SynthesizerTrnMs768NSFsid
    from synthetic import weights
    cfg = weights.SYNTH_CFG_TINY
    monkeypatch.delenv("AICG_HALF", raising=False)
    net = SynthesizerTrnMs768NSFsid(*cfg, is_half=True)
    del net.enc_q
    net.load_state_dict(weights.synth_state_dict(cfg, 7), strict=False)
    net.eval().to(dev.device).half()
    P = net._prepare()
    marked = []
    ops_walk(P, marked)
    assert marked and not any(pc.f16 for pc in marked)
    monkeypatch.setenv("AICG_HALF", "1")
    net.half()
    assert any(pc.f16 for pc in marked)
    net.float()
    assert not any(pc.f16 for pc in marked)


def ops_walk(tree, out):
    if isinstance(tree, ops.PackedConv):
        out.append(tree)
    elif isinstance(tree, ops.PackedConvTranspose):
        out.append(tree.gemm)
    elif isinstance(tree, dict):
        for v in tree.values():
            ops_walk(v, out)
    elif isinstance(tree, (list, tuple)):
        for v in tree:
            ops_walk(v, out)


def _synth(dev, cfg, T, half, seed=1234):
    from aicovergen_amd.infer_pack.models import SynthesizerTrnMs768NSFsid
    from synthetic import weights
    from synthetic.inputs import synth_inputs
    net = SynthesizerTrnMs768NSFsid(*cfg, is_half=half)
    del net.enc_q
    net.load_state_dict(weights.synth_state_dict(cfg, seed), strict=False)
    net.eval().to(dev.device)
    net = net.half() if half else net.float()
    phone, pitch, f0, nz, ns = synth_inputs(cfg, T, seed + 1)
    o, _, (z, z_p, m_p, logs_p) = net.infer(phone, torch.tensor([T]), pitch, f0, torch.tensor([1]), noise_z=nz, noise_src=ns)
    return o, z


def test_synth_tiny_half_against_fp32(dev, monkeypatch):
    from synthetic import weights
    monkeypatch.setenv("AICG_HALF", "1")
    o32, z32 = _synth(dev, weights.SYNTH_CFG_TINY, 24, False)
    o16, z16 = _synth(dev, weights.SYNTH_CFG_TINY, 24, True)
    e = rel_rms(o16, o32)
    assert e < 2e-2, e


@pytest.mark.gpu
def test_synth_40k_half_against_fp32(monkeypatch):
    """Full-size v2 / 40 kHz synthesizer, 3 s: the fp16-operand run against the fp32 run of the same weights and noise."""
    import conftest
    from synthetic import weights
    conftest._bind("hip")
    dev = conftest.Dev("hip")
    monkeypatch.setenv("AICG_HALF", "1")
    o32, z32 = _synth(dev, weights.SYNTH_CFG_40K_V2, 300, False)
    o16, z16 = _synth(dev, weights.SYNTH_CFG_40K_V2, 300, True)
    ez, eo = rel_rms(z16, z32), rel_rms(o16, o32)
    assert 1e-6 < ez < 5e-3, ez
    assert 1e-5 < eo < 2e-2, eo


@pytest.mark.gpu
def test_hubert_half_against_fp32(monkeypatch):
    import conftest
    from aicovergen_amd import hubert
    from synthetic import weights
    conftest._bind("hip")
    dev = conftest.Dev("hip")
    monkeypatch.setenv("AICG_HALF", "1")
    sd = weights.hubert_state_dict(seed=3)
    torch.manual_seed(5)
    wav = torch.randn(1, 400 + 320 * 199).to(dev.device) * 0.1      # 200 frames: the 1 x 1 GEMMs take conv_g1 (positions % 4 == 0)
    outs = []
    for half in (False, True):
        net = hubert.HubertModel(sd).to(dev.device)
        net = net.half() if half else net.float()
        feats = net.extract_features(source=wav, padding_mask=torch.zeros_like(wav, dtype=torch.bool), output_layer=12)[0]
        outs.append(feats.float().cpu())
    e = rel_rms(outs[1], outs[0])
    assert 1e-6 < e < 1e-2, e


@pytest.mark.gpu
def test_c1_pipeline_half_against_fp32_and_the_reference(monkeypatch):
    """BASELINE C1 (30 s, full-size networks) through VC.pipeline with HuBERT and the synthesizer in the is_half mode: same f0 (the f0
    models stay fp32: every coarse bin equal by construction), the int16 waveform within 5e-3 relative rms of the fp32 run and of the
    REFERENCE's own (fp32, CPU) output of tests/golden/pipeline_c1_30s.npz; reported: the <= 1 LSB rate."""
    import numpy as np
    import conftest
    from synthetic import weights
    from synthetic.inputs import vocal_like
    from test_pipeline import build, noise_fn_for
    conftest._bind("hip")
    dev = conftest.Dev("hip")
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "pipeline_c1_30s.npz"))
    seed, x = int(gold["seed"][0]), tuple(int(v) for v in gold["x"])
    nets = weights.full_model_set(seed)
    audio = vocal_like(float(gold["seconds"][0]), 16000, seed + 5)
    monkeypatch.setenv("AICG_HALF", "1")
    outs = []
    for half in (False, True):
        vc, hub, net_g, tgt_sr = build(dev, nets, x)
        hub, net_g = (hub.half(), net_g.half()) if half else (hub.float(), net_g.float())
        outs.append(vc.pipeline(hub, net_g, 0, audio, "x.wav", [0, 0, 0], 0, "rmvpe", "", 0.5, 1, 3, tgt_sr, 0, 0.25, "v2", 0.33, 128,
                                noise_fn=noise_fn_for(nets)))
    o32, o16, ref = outs[0].astype(np.float64), outs[1].astype(np.float64), gold["audio"].astype(np.float64)
    e = float(np.sqrt(((o16 - o32) ** 2).sum() / (o32 ** 2).sum()))
    er = float(np.sqrt(((o16 - ref) ** 2).sum() / (ref ** 2).sum()))
    print("C1 is_half vs fp32: rel rms %.3e, <= 1 LSB on %.4f; vs the reference's fp32 output: %.3e" % (e, (np.abs(o16 - o32) <= 1).mean(), er))
    assert 1e-5 < e < 5e-3, e          # measured 5.3e-4 (and 6.8e-4 from the reference: inside SURVEY 8d's fp32 bar of 1e-3)
    assert er < 5e-3, er
