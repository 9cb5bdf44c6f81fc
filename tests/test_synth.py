"""SynthesizerTrnMs768NSFsid.infer on the HIP kernels vs (a) the golden output of the REFERENCE module
(tests/golden/synth_*.npz, produced by tests/golden/make_golden.py from /root/reference) and (b) the oracle
restatement.  Same seeded parameters and the same injected noise on both sides.
Tolerance: relative RMS <= 1e-4 on the waveform (<= 1e-4 for conv/GEMM stages)."""
import os

import numpy as np
import pytest
import torch

from aicovergen_amd.infer_pack.models import SynthesizerTrnMs768NSFsid
from conftest import rel_rms
from oracle import synth
from synthetic import weights
from synthetic.inputs import synth_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _run(dev, cfg, T, seed=1234):
    sd = weights.synth_state_dict(cfg, seed)
    net = SynthesizerTrnMs768NSFsid(*cfg, is_half=False)
    del net.enc_q                                   # as src/rvc.py:133 does
    net.load_state_dict(sd, strict=False)
    net.eval().to(dev.device)
    phone, pitch, f0, nz, ns = synth_inputs(cfg, T, seed + 1)
    o, x_mask, (z, z_p, m_p, logs_p) = net.infer(phone, torch.tensor([T]), pitch, f0, torch.tensor([1]), noise_z=nz,
                                                  noise_src=ns)
    return sd, (phone, pitch, f0, nz, ns), o, z, m_p, logs_p


def test_synth_tiny_vs_reference_golden(dev):
    cfg, T = weights.SYNTH_CFG_TINY, 24
    gold = np.load(os.path.join(GOLD, "synth_tiny_T24.npz"))
    _, _, o, z, m_p, logs_p = _run(dev, cfg, T)
    assert o.shape == (1, 1, T * 400)
    assert rel_rms(m_p[0], torch.from_numpy(gold["m_p"])) < 1e-4
    assert rel_rms(logs_p[0], torch.from_numpy(gold["logs_p"])) < 1e-4
    assert rel_rms(z[0], torch.from_numpy(gold["z"])) < 1e-4
    assert rel_rms(o[0, 0], torch.from_numpy(gold["audio"])) < 1e-4


@pytest.mark.gpu
def test_synth_40k_vs_reference_golden():
    """Full-size v2 / 40 kHz synthesizer (27.5 M parameters) against the reference's output."""
    import conftest
    conftest._bind("hip")
    dev = conftest.Dev("hip")
    cfg, T = weights.SYNTH_CFG_40K_V2, 16
    gold = np.load(os.path.join(GOLD, "synth_40k_T16.npz"))
    _, _, o, z, m_p, logs_p = _run(dev, cfg, T)
    assert rel_rms(z[0], torch.from_numpy(gold["z"])) < 1e-4
    assert rel_rms(o[0, 0], torch.from_numpy(gold["audio"])) < 1e-4


@pytest.mark.gpu
def test_synth_40k_long_vs_oracle():
    """A multi-second chunk (T = 300 frames = 3 s) against the oracle restatement run on the host CPU."""
    import conftest
    conftest._bind("hip")
    dev = conftest.Dev("hip")
    cfg, T = weights.SYNTH_CFG_40K_V2, 300
    sd, (phone, pitch, f0, nz, ns), o, z, m_p, logs_p = _run(dev, cfg, T)
    with torch.no_grad():
        ro, (rz, rz_p, rm, rl) = synth.synth_infer(sd, cfg, phone, pitch, f0, torch.tensor([1]), nz, ns)
    assert rel_rms(z, rz) < 1e-4
    assert rel_rms(o, ro) < 1e-4
    assert (o.cpu() - ro).abs().max() < 1e-4


@pytest.mark.gpu
def test_resblock_chains_on_side_streams_are_bit_identical(monkeypatch):
    """The vocoder's ResBlock chains run on their own streams with the accumulating convolutions ordered by events
    (GeneratorNSF.forward, reference src/infer_pack/models.py:506-512): the waveform must equal the one-stream walk BIT FOR BIT,
    call after call (a missed dependency would show as run-to-run differences)."""
    import conftest
    conftest._bind("hip")
    d = conftest.Dev("hip")
    cfg, T = weights.SYNTH_CFG_40K_V2, 900
    sd = weights.synth_state_dict(cfg, 4321)
    net = SynthesizerTrnMs768NSFsid(*cfg, is_half=False)
    del net.enc_q
    net.load_state_dict(sd, strict=False)
    net.eval().to(d.device)
    phone, pitch, f0, nz, ns = synth_inputs(cfg, T, 99)
    run = lambda: net.infer(phone, torch.tensor([T]), pitch, f0, torch.tensor([1]), noise_z=nz, noise_src=ns)[0].clone()
    monkeypatch.setenv("AICG_RB_STREAMS", "0")
    one = run()
    monkeypatch.setenv("AICG_RB_STREAMS", "1")
    assert net._rb_streams(d.device, 3) is not None
    for _ in range(4):
        assert torch.equal(run(), one)


def test_synth_default_noise_path_runs(dev):
    """Without injected noise the drop-in draws from torch's global RNG like the reference (models.py:748,368)."""
    cfg, T = weights.SYNTH_CFG_TINY, 8
    sd = weights.synth_state_dict(cfg, 1)
    net = SynthesizerTrnMs768NSFsid(*cfg, is_half=False)
    del net.enc_q
    net.load_state_dict(sd, strict=False)
    net.eval().to(dev.device)
    phone, pitch, f0, _, _ = synth_inputs(cfg, T, 2)
    o, _, _ = net.infer(phone, torch.tensor([T]), pitch, f0, torch.tensor([0]))
    assert o.shape == (1, 1, T * 400) and torch.isfinite(o).all()


@pytest.mark.parametrize("name,variant", [("synth_tiny_v1_T24", "256f0"), ("synth_tiny_nono_T24", "768nono")])
def test_synth_variants_vs_reference_golden(dev, name, variant):
    """v1 (256-d phone features) and _nono (no f0: plain HiFi-GAN Generator) classes, SURVEY 8f item 4."""
    from aicovergen_amd.infer_pack import models as M
    cfg, T = weights.SYNTH_CFG_TINY, 24
    phone_dim, f0_on = (256 if variant.startswith("256") else 768), variant.endswith("f0")
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    sd = weights.synth_state_dict(cfg, 1234, phone_dim=phone_dim, f0=f0_on)
    cls = {"256f0": M.SynthesizerTrnMs256NSFsid, "768nono": M.SynthesizerTrnMs768NSFsid_nono}[variant]
    net = cls(*cfg, is_half=False)
    del net.enc_q
    net.load_state_dict(sd, strict=False)
    net.eval().to(dev.device)
    phone, pitch, f0, nz, ns = synth_inputs(cfg, T, 1235)
    phone = phone[:, :, :phone_dim].contiguous()
    if f0_on:
        o, _, (z, _, _, _) = net.infer(phone, torch.tensor([T]), pitch, f0, torch.tensor([1]), noise_z=nz, noise_src=ns)
    else:
        o, _, (z, _, _, _) = net.infer(phone, torch.tensor([T]), torch.tensor([1]), noise_z=nz)
    assert rel_rms(z[0], torch.from_numpy(gold["z"])) < 1e-4
    assert rel_rms(o[0, 0], torch.from_numpy(gold["audio"])) < 1e-4


@pytest.mark.parametrize("name,up,upk,sr", [("48k_v2", [12, 10, 2, 2], [24, 20, 4, 4], 48000),
                                            ("32k_v2", [10, 8, 2, 2], [20, 16, 4, 4], 32000)])
def test_other_sample_rate_geometries_match_oracle(dev, name, up, upk, sr):
    """The 48 kHz / 32 kHz v2 voice models (configs/48k_v2.json:39-41, 32k_v2.json:39-41) use other upsampling factors and
    transposed-conv kernels; same code path (GEMM + col2im, stride-phase noise convs), tiny channel counts."""
    cfg = list(weights.SYNTH_CFG_TINY)
    cfg[12], cfg[14], cfg[17] = up, upk, sr
    T = 6
    sd, (phone, pitch, f0, nz, ns), o, z, m_p, logs_p = _run(dev, cfg, T, seed=77)
    upp = int(np.prod(up))
    assert o.shape == (1, 1, T * upp)
    with torch.no_grad():
        ro, (rz, _, _, _) = synth.synth_infer(sd, cfg, phone, pitch, f0, torch.tensor([1]), nz, ns)
    assert rel_rms(z, rz) < 1e-4
    assert rel_rms(o, ro) < 1e-4
