"""Parity at the sizes bench.py times: different sizes select different conv tiles / K splits, the
split-attention merge path, the 16-window MDX batches and the two-workgroup GRU over 24 608 steps.

  * BASELINE config C1: 30 s S16 through VC.pipeline with full-size networks and main.py's (3,10,60,65) preset against the
    int16 output of the REFERENCE's own VC.pipeline (tests/golden/pipeline_c1_30s.npz, made by tests/golden/make_golden.py c1
    from /root/reference/src): one 576 000-sample chunk.
  * one 66 s chunk (T_h = 3300 HuBERT frames, synthesizer T = 6600 -> 2.64 M samples) against the oracle on the host CPU;
  * whole-track RMVPE on 240 s (24 608 frames) against the oracle, disagreeing frames listed with their salience margins;
  * a 16-window MDX batch (8 windows x {+x, -x}) at 3072/256/7680 against the oracle per window, and the window counts of a
    4-minute track.
All `-m gpu`: the oracle side needs minutes of host CPU even on the GPU box.
"""
import os

import numpy as np
import pytest
import torch

from conftest import rel_rms
from oracle import hubert as ohub
from oracle import mdxnet
from oracle import pipeline as opipe
from oracle import rmvpe as orm
from oracle import synth as osynth
from synthetic import weights
from synthetic.inputs import song_like, synth_inputs, vocal_like

GOLD = os.path.join(os.path.dirname(__file__), "golden")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _hip():
    import conftest
    conftest._bind("hip")
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    yield
    torch.cuda.synchronize()


def _margins(sal, frames):
    """top-1 minus top-2 salience of the given frames."""
    top2 = np.sort(sal[frames], axis=1)[:, -2:]
    return top2[:, 1] - top2[:, 0]


def test_c1_pipeline_vs_reference_golden():
    """BASELINE C1.  Waveform bar: relative RMS <= 1e-3 on the int16 output; reported: <= 1 LSB rate.
    f0: every frame whose coarse bin differs from the reference's is listed with its f0 distance and salience margin;
    the bar is <= 0.2 % of frames, each within one bin (a cents value that rounds across a bin edge)."""
    from test_pipeline import build, noise_fn_for
    gold = np.load(os.path.join(GOLD, "pipeline_c1_30s.npz"))
    seed, x = int(gold["seed"][0]), tuple(int(v) for v in gold["x"])
    nets = weights.full_model_set(seed)
    audio = vocal_like(float(gold["seconds"][0]), 16000, seed + 5)
    import conftest
    dev = conftest.Dev("hip")
    vc, hub, net_g, tgt_sr = build(dev, nets, x)
    times = [0, 0, 0]
    out = vc.pipeline(hub, net_g, 0, audio, "x.wav", times, 0, "rmvpe", "", 0.5, 1, 3, tgt_sr, 0, 0.25, "v2", 0.33, 128,
                      noise_fn=noise_fn_for(nets))
    ref = gold["audio"]
    assert out.dtype == np.int16 and out.shape == ref.shape == (1199200,)
    diff = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    rel = np.sqrt(np.sum(diff.astype(np.float64) ** 2) / np.sum(ref.astype(np.float64) ** 2))
    print("C1 vs reference: rel rms %.3e, max |diff| %d of peak %d, <= 1 LSB on %.4f, exact on %.4f"
          % (rel, diff.max(), np.abs(ref).max(), (diff <= 1).mean(), (diff == 0).mean()))
    # Tolerance, triangulated (tests/golden/make_fp64_c1.py): the same pipeline evaluated in float64 stands in for exact arithmetic.
    # The REFERENCE's own fp32 output is 1.20e-4 relative RMS / 10 LSB / <= 1 LSB on 92.6 % away from it, its f0 2.3e-7 -- that is
    # the noise floor of ANY fp32 evaluation (the vocoder's source integrates f0 over the whole 36 s chunk, so the waveform reacts
    # to every reordering of the f0 path's sums: two equally accurate kernel choices measured 2.5e-4 and 4.5e-4 from float64).
    # Gates: SURVEY 8d's relative RMS <= 1e-3 (= 8x the reference's own distance from float64: three fp32 evaluations sampled
    # 1.2e-4, 2.5e-4 and 4.5e-4) against the reference AND against float64; largest deviation within 4x the reference's; f0 as close
    # to float64 as the reference's is (2x), all coarse bins equal (below).
    g64 = np.load(os.path.join(GOLD, "pipeline_c1_30s_fp64.npz"))
    dec = int(g64["decim"][0])
    d64 = np.abs(out[::dec].astype(np.int32) - g64["audio"].astype(np.int32))
    rel64 = np.sqrt(np.sum(d64.astype(np.float64) ** 2) / np.sum(g64["audio"].astype(np.float64) ** 2))
    print("C1 vs float64: rel rms %.3e, max |diff| %d, <= 1 LSB on %.4f   (reference vs float64: %.3e, %d, %.4f)"
          % (rel64, d64.max(), (d64 <= 1).mean(), float(g64["ref_rel_rms"][0]), int(g64["ref_max_lsb"][0]), float(g64["ref_le1"][0])))
    assert rel < 1e-3 and rel64 < 1e-3
    assert d64.max() <= 4 * int(g64["ref_max_lsb"][0]) and diff.max() <= 2e-3 * np.abs(ref).max()
    # f0 bins against the reference's own get_f0 output
    _, audio_pad, opt_ts, p_len = vc.plan(audio)
    assert opt_ts == []                                             # 30 s < x_max: a single chunk
    assert audio_pad.shape[0] == 576000
    coarse, f0 = vc.get_f0("x.wav", audio_pad, p_len, 0, "rmvpe", 3, 128)
    n = min(len(coarse), len(gold["coarse"]))
    bad = np.nonzero(coarse[:n] != gold["coarse"][:n])[0]
    r = vc.model_rmvpe
    mel = r.mel_extractor(audio_pad.float()[None].cuda(), center=True)
    sal = r.mel2hidden(mel)[0].cpu().numpy()
    for t, m in zip(bad, _margins(sal, bad)):
        print("  frame %d: bin %d vs reference %d, f0 %.4f vs %.4f Hz, salience top1-top2 %.3e"
              % (t, coarse[t], gold["coarse"][t], f0[t], gold["f0"][t], m))
    print("C1 coarse-bin agreement %.5f (%d of %d frames differ)" % (1 - len(bad) / n, len(bad), n))
    assert len(bad) <= 0.002 * n
    assert np.all(np.abs(coarse[:n][bad].astype(int) - gold["coarse"][:n][bad].astype(int)) <= 1)
    voiced = (f0[:n] > 0) & (gold["f0"][:n] > 0)
    assert np.array_equal(f0[:n] > 0, gold["f0"][:n] > 0) or (np.sum((f0[:n] > 0) != (gold["f0"][:n] > 0)) <= 0.002 * n)
    assert np.max(np.abs(f0[:n][voiced] / gold["f0"][:n][voiced] - 1)) < 1e-3
    v64 = (f0[:n] > 0) & (g64["f0"][:n] > 0)
    e64 = np.sqrt(np.mean((f0[:n][v64] / g64["f0"][:n][v64] - 1) ** 2))
    print("C1 f0 vs float64: relative rms %.3e (reference vs float64: %.3e)" % (e64, float(g64["ref_f0_rel_rms"][0])))
    assert e64 <= 2.0 * float(g64["ref_f0_rel_rms"][0])
    assert np.array_equal(coarse[:n], g64["coarse"][:n])


@pytest.mark.parametrize("schedule", ["default", "one_launch"])
def test_c1_pipeline_with_the_references_f0_injected(monkeypatch, schedule):
    """BASELINE C1 with the chaotic part held fixed.  The free-running test above has to allow 1e-3: RMVPE's
    f0 agrees with the reference's to 2.5e-7, but the vocoder's harmonic source integrates f0 over the whole 36 s chunk, so
    equally accurate f0 roundings give waveforms 1.2e-4 ... 4.7e-4 apart.  Here the REFERENCE's own f0 track
    (gold["f0"], what its get_f0 returned) replaces the estimator's output; everything else -- filtfilt, HuBERT, feature
    plumbing, coarse bins, text encoder, flow, SineGen, vocoder, RMS mix, int16 -- is this implementation.  A regression in any
    of those can no longer hide under the f0 -> phase sensitivity: the waveform must sit at accumulation-order distance from
    the reference's.
    Both schedules (VERDICT r4 "missing" #5): "default" is what bench.py times -- on a GPU the PROGRESSIVE schedule (recurrence in 8
    segments, pitch published per frame range on the side stream, get_f0 never called); "one_launch" (AICG_F0_SEGMENTS=1) asks get_f0
    for the whole track.  The injection sits in VC._estimated_f0, the seam every estimate passes under either schedule."""
    from test_pipeline import build, noise_fn_for
    gold = np.load(os.path.join(GOLD, "pipeline_c1_30s.npz"))
    seed, x = int(gold["seed"][0]), tuple(int(v) for v in gold["x"])
    nets = weights.full_model_set(seed)
    audio = vocal_like(float(gold["seconds"][0]), 16000, seed + 5)
    import conftest
    dev = conftest.Dev("hip")
    if schedule == "one_launch":
        monkeypatch.setenv("AICG_F0_SEGMENTS", "1")
    else:
        monkeypatch.delenv("AICG_F0_SEGMENTS", raising=False)
    vc, hub, net_g, tgt_sr = build(dev, nets, x)
    ranges = []

    def reference_f0(lo, hi, f0):
        ranges.append((lo, hi))
        g = gold["f0"][lo:hi].astype(np.float64)
        assert len(g) >= min(hi, len(gold["f0"])) - lo
        if torch.is_tensor(f0):      # progressive schedule: a frame range on the device
            out = f0.clone()
            out[: len(g)] = torch.from_numpy(g).to(f0.device)
            return out
        out = np.array(f0, dtype=np.float64)
        out[: len(g)] = g
        return out
    vc._estimated_f0 = reference_f0
    tails = []
    orig_tail = vc._f0_tail

    def spy_tail(f0, factor):
        f0bak, coarse = orig_tail(f0, factor)
        tails.append((ranges[-1], coarse))
        return f0bak, coarse
    vc._f0_tail = spy_tail
    out = vc.pipeline(hub, net_g, 0, audio, "x.wav", [0, 0, 0], 0, "rmvpe", "", 0.5, 1, 3, tgt_sr, 0, 0.25, "v2", 0.33, 128,
                      noise_fn=noise_fn_for(nets))
    assert vc.last_profile["f0_progressive"] == (1.0 if schedule == "default" else 0.0)
    if schedule == "default":
        assert len(ranges) > 2 and ranges[0][0] > 0          # the middle of the track first, in several ranges
    else:
        assert ranges == [(0, ranges[0][1])]
    ref = gold["audio"]
    # same f0 in -> the quantiser must give the same bins: bit-exact, range by range
    n = len(gold["coarse"])
    for (lo, hi), coarse in tails:
        m = min(hi, n) - lo
        if m > 0:
            assert np.array_equal(coarse[:m].cpu().numpy(), gold["coarse"][lo:lo + m])
    diff = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    rel = np.sqrt(np.sum(diff.astype(np.float64) ** 2) / np.sum(ref.astype(np.float64) ** 2))
    print("C1, reference f0 injected, %s schedule (%d f0 range(s)): rel rms %.3e, max |diff| %d of peak %d, <= 1 LSB on %.5f, exact on %.4f"
          % (schedule, len(ranges), rel, diff.max(), np.abs(ref).max(), (diff <= 1).mean(), (diff == 0).mean()))
    # SURVEY 8(d)'s end-to-end bar, reachable once the f0 -> phase path is held fixed: <= 1 LSB on >= 99.9 % of the samples.
    # Measured (round 4): max |diff| 1 LSB, <= 1 LSB on 100.000 %, exact on 93.8 %, rel rms 3.46e-5 -- which IS that 6.2 % of
    # one-LSB flips of the truncating int16 cast (q of the samples off by one LSB give sqrt(q) / rms(ref) = 3.5e-5 at q = 0.062);
    # a relative RMS bar below ~5e-5 would gate the cast's rounding, not the arithmetic
    assert (diff <= 1).mean() >= 0.999
    assert diff.max() <= 2
    assert rel <= 6e-5


@pytest.mark.parametrize("n", [1056160, 640160])
def test_chunk_hubert_and_synth_vs_oracle(n):
    """The chunk sizes of the two presets: (3,10,60,65) -> 1 056 160 samples, T_h = 3300 (3-way split attention, 64x64
    tiles on the QKV/FFN GEMMs), synthesizer T = 6600 -> 2 640 000 output samples (vocoder tiles of the bench); (1,6,38,41) ->
    640 160 samples, T_h = 2000, T = 4000 -> 1 600 000 samples (other tile / attention-split choices)."""
    from aicovergen_amd.hubert import HubertModel
    from aicovergen_amd.infer_pack.models import SynthesizerTrnMs768NSFsid
    th = (n - 400) // 320 + 1
    wav = torch.from_numpy(vocal_like(n / 16000.0 + 0.01, 16000, seed=31)[:n]).unsqueeze(0)
    cfg = weights.HUBERT_BASE
    sd = weights.hubert_state_dict(cfg, 1234)
    m = HubertModel(sd, cfg).to("cuda:0")
    y, _ = m.extract_features(source=wav, padding_mask=None, output_layer=12)
    assert y.shape == (1, th, 768)
    with torch.no_grad():
        ref = ohub.extract_features(sd, cfg, wav, 12)
    e = rel_rms(y, ref)
    print("HuBERT-base T_h=%d: rel rms %.3e" % (th, e))
    assert e < 1e-4
    del m, y, ref
    scfg, T = weights.SYNTH_CFG_40K_V2, 2 * th
    ssd = weights.synth_state_dict(scfg, 1236)
    net = SynthesizerTrnMs768NSFsid(*scfg, is_half=False)
    del net.enc_q
    net.load_state_dict(ssd, strict=False)
    net.eval().to("cuda:0")
    phone, pitch, f0, nz, ns = synth_inputs(scfg, T, 77)
    o, _, (z, z_p, m_p, logs_p) = net.infer(phone, torch.tensor([T]), pitch, f0, torch.tensor([3]), noise_z=nz, noise_src=ns)
    assert o.shape == (1, 1, T * 400)
    with torch.no_grad():
        ro, (rz, rzp, rmp, rlp) = osynth.synth_infer(ssd, scfg, phone, pitch, f0, torch.tensor([3]), nz, ns)
    e_mp, e_z, e_o = rel_rms(m_p, rmp), rel_rms(z, rz), rel_rms(o, ro)
    print("synthesizer T=%d: rel rms m_p %.3e, z %.3e, audio %.3e" % (T, e_mp, e_z, e_o))
    assert e_mp < 1e-4 and e_z < 1e-4 and e_o < 1e-3


def test_240s_rmvpe_vs_oracle():
    """Whole-track f0 of a 4-minute input: 24 001 frames padded to 24 032, the (default) four-workgroup GRU over every step."""
    from aicovergen_amd.rmvpe import RMVPE
    from aicovergen_amd import ops
    sd = weights.rmvpe_state_dict(weights.RMVPE_FULL, 1235)
    r = RMVPE(None, False, "cuda:0", state_dict=sd)
    audio = vocal_like(240.0, 16000, seed=1234)
    audio = np.pad(audio, (48000, 48000), mode="reflect")          # what pipeline() hands get_f0 (t_pad = 3 s)
    f0 = r.infer_from_audio(audio, 0.03)
    assert not ops.gru_timed_out()
    mel = r.mel_extractor(torch.from_numpy(audio)[None].cuda(), center=True)
    hid = r.mel2hidden(mel)[0].cpu().numpy()
    of0, ohid = orm.infer_from_audio(sd, audio, 0.03)
    assert hid.shape == ohid.shape and hid.shape[0] == 24601
    print("RMVPE 246 s: salience max |diff| %.3e" % np.abs(hid - ohid).max())
    assert np.abs(hid - ohid).max() < 1e-3
    bad = np.nonzero(hid.argmax(1) != ohid.argmax(1))[0]
    for t, m in zip(bad[:50], _margins(ohid, bad[:50])):
        print("  frame %d: argmax %d vs oracle %d, oracle top1-top2 margin %.3e" % (t, hid[t].argmax(), ohid[t].argmax(), m))
    print("salience argmax agreement %.5f (%d frames differ)" % (1 - len(bad) / len(hid), len(bad)))
    assert len(bad) <= 0.002 * len(hid)
    if len(bad):
        assert _margins(ohid, bad).max() < 1e-3                    # only near-ties may flip
    ok = np.ones(len(hid), bool)
    ok[bad] = False
    assert np.array_equal((f0 > 0)[ok], (of0 > 0)[ok]) or np.sum((f0 > 0)[ok] != (of0 > 0)[ok]) <= 5
    v = ok & (f0 > 0) & (of0 > 0)
    assert np.max(np.abs(f0[v] / of0[v] - 1)) < 1e-3


def test_mdx_16_window_batch_vs_oracle_and_window_counts():
    """The bench's MDX launches: 8 windows x {+x, -x} = 16 images through stft -> U-Net -> istft in one batch; three of the
    eight windows are checked against the oracle (one U-Net forward costs the host ~0.8 TFLOP).  Window counts of a
    4-minute track: 2 segments of 5 336 100 samples -> 22 windows each, pad 239 580."""
    from aicovergen_amd.mdx import MDX, MDXModel
    cfg = weights.MDX_VOC_FT
    sd = weights.mdx_state_dict(cfg, 1234)
    model = MDXModel("cuda:0", cfg["dim_f"], cfg["dim_t"], cfg["n_fft"])
    sess = MDX(None, model, state_dict=sd)
    n = 240 * 44100
    segs = MDX.segment(np.zeros((2, n), np.float32), False, n // 2)
    assert [s.shape[1] for s in segs] == [5336100, 5336100]
    trim, gen, pad = sess._geometry(5336100)
    assert (gen, pad, (5336100 + pad) // gen) == (253440, 239580, 22)
    assert sess.WINDOW_BATCH == 8
    wave = song_like(8 * gen / 44100.0 + 0.5, 44100, seed=5)[:, :8 * gen - 1000]
    mix, pad8, trim = sess.pad_wave(wave)
    assert mix.shape == (8, 2, model.chunk_size)
    mw = torch.cat([mix, -mix], 0)
    with torch.no_grad():
        got = model.istft_tf(sess.net.forward_tf(model.stft_tf(mw)))
        for w in (0, 5, 15):
            x = mw[w:w + 1].cpu()
            ref = mdxnet.istft(mdxnet.unet(sd, cfg, mdxnet.stft(x, cfg["n_fft"], 1024, cfg["dim_f"])), cfg["n_fft"], 1024)
            e = rel_rms(got[w:w + 1], ref)
            print("MDX window %d of the 16-image batch: rel rms %.3e" % (w, e))
            assert e < 1e-4


@pytest.mark.parametrize("name", ["MDX_KARA2", "MDX_REVERB_HQ"])
def test_other_mdx_geometries_vs_oracle(name):
    """The (dim_f, dim_t, n_fft) classes of the other two separations in main.py's chain (src/main.py:185,188): (2048, 256, 5120)
    -- a radix-5 FFT length -- and (3072, 512, 6144); one window through stft -> U-Net -> istft vs the oracle."""
    from aicovergen_amd.mdx import MDX, MDXModel
    cfg = getattr(weights, name)
    sd = weights.mdx_state_dict(cfg, 4321)
    model = MDXModel("cuda:0", cfg["dim_f"], cfg["dim_t"], cfg["n_fft"])
    sess = MDX(None, model, state_dict=sd)
    assert model.chunk_size == 1024 * (cfg["dim_t"] - 1)
    x = torch.from_numpy(song_like(model.chunk_size / 44100.0 + 0.01, 44100, seed=8)[:, :model.chunk_size]).unsqueeze(0)
    with torch.no_grad():
        ref = mdxnet.istft(mdxnet.unet(sd, cfg, mdxnet.stft(x, cfg["n_fft"], 1024, cfg["dim_f"])), cfg["n_fft"], 1024)
        got = model.istft_tf(sess.net.forward_tf(model.stft_tf(x.cuda())))
    e = rel_rms(got, ref)
    print("%s window: rel rms %.3e" % (name, e))
    assert e < 1e-4
