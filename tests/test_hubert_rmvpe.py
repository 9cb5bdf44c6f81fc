"""HuBERT feature extractor and RMVPE f0 estimator on the HIP kernels vs the oracle restatements
(oracle/hubert.py pinned against transformers.HubertModel, oracle/rmvpe.py pinned against the reference's own
src/rmvpe.py -- see tests/test_oracle_golden.py)."""
import os

import numpy as np
import pytest
import torch

from aicovergen_amd import ops
from aicovergen_amd.hubert import HubertModel, _infer_cfg
from aicovergen_amd.rmvpe import RMVPE, mel_filterbank
from conftest import rel_rms
from oracle import hubert as ohub
from oracle import rmvpe as orm
from synthetic import weights
from synthetic.inputs import vocal_like


def test_hubert_tiny_matches_oracle(dev):
    cfg = weights.HUBERT_TINY
    sd = weights.hubert_state_dict(cfg, 1234)
    m = HubertModel(sd, cfg).to(dev.device)
    torch.manual_seed(0)
    wav = torch.randn(1, 8123) * 0.3
    y, pm = m.extract_features(source=wav, padding_mask=torch.zeros(wav.shape, dtype=torch.bool), output_layer=12)
    with torch.no_grad():
        ref = ohub.extract_features(sd, cfg, wav, 12)
    assert y.shape == ref.shape == (1, (8123 - 400) // 320 + 1, cfg["embed"])
    assert rel_rms(y, ref) < 1e-4
    assert rel_rms(m.final_proj(y), ohub.final_proj(sd, ref)) < 1e-4
    y1, _ = m.extract_features(source=wav, padding_mask=None, output_layer=1)
    with torch.no_grad():
        assert rel_rms(y1, ohub.extract_features(sd, cfg, wav, 1)) < 1e-4


def test_hubert_extractor_on_the_stride2_dma_kernel(dev):
    """A 64-channel extractor: wide enough for csrc/conv_g1s.h, so the five stride-2 layers behind layer 0 run the DMA-staged kernel on
    rows padded to 16 bytes (frame counts 1 999 -> 999 -> 499 -> ...: every residue mod 4) -- against the oracle, and the routing is
    asserted (HUBERT_TINY's 32 channels stay on the producer / consumer kernels)."""
    from aicovergen_amd import _lib
    cfg = dict(weights.HUBERT_TINY, conv_dim=64)
    sd = weights.hubert_state_dict(cfg, 4321)
    m = HubertModel(sd, cfg).to(dev.device)
    torch.manual_seed(1)
    wav = torch.randn(1, 10003) * 0.3
    seen = []
    orig = ops.conv

    def spy(x, pc, *a, **k):
        y = orig(x, pc, *a, **k)
        if pc.stride[1] == 2:
            seen.append(_lib.last_launch())
        return y
    ops.conv = spy
    ops.gemm_tile = 3       # (the policy keeps launches of fewer than 160 workgroups on the producer / consumer kernels: force the tile)
    try:
        y, _ = m.extract_features(source=wav, padding_mask=None, output_layer=12)
    finally:
        ops.conv = orig
        ops.gemm_tile = 0
    assert seen == ["conv_g1s_kernel"] * 6, seen
    with torch.no_grad():
        ref = ohub.extract_features(sd, cfg, wav, 12)
    assert y.shape == ref.shape and rel_rms(y, ref) < 1e-4


def test_hubert_many_chunks_at_once(dev):
    """extract_features_many: the transformer's per-token layers run once over chunks of different lengths laid side by side;
    every chunk must come out as from its own call (other GEMM tiles: fp32 summation order only) and match the oracle."""
    cfg = weights.HUBERT_TINY
    sd = weights.hubert_state_dict(cfg, 1234)
    m = HubertModel(sd, cfg).to(dev.device)
    torch.manual_seed(1)
    wavs = [torch.randn(1, n) * 0.3 for n in (8123, 5000, 9999)]
    many = m.extract_features_many([dev.t(w) for w in wavs], 12)
    assert len(many) == 3
    for w, y in zip(wavs, many):
        alone, _ = m.extract_features(source=w, padding_mask=None, output_layer=12)
        assert y.shape == alone.shape == (1, (w.shape[1] - 400) // 320 + 1, cfg["embed"])
        assert rel_rms(y, alone) < 2e-6
        with torch.no_grad():
            assert rel_rms(y, ohub.extract_features(sd, cfg, w, 12)) < 1e-4
    nine = m.extract_features_many([dev.t(w) for w in wavs[:2]], 9)
    with torch.no_grad():
        assert rel_rms(nine[1], ohub.extract_features(sd, cfg, wavs[1], 9)) < 1e-4


def test_hubert_qkv_as_one_gemm_is_exact(dev):
    """head_dim = 64 (HuBERT-base's): head_dim ** -0.5 = 1/8 is a power of two, q / k / v run as ONE 3 E-row GEMM with the scale folded
    into q's rows of the weights and bias.  Scaling by a power of two commutes with every fp32 rounding: the merged GEMM's q rows equal
    the separate GEMM's (scaled in its epilogue) BIT FOR BIT when both run the same kernel family (same accumulation order: asserted on
    a map large enough for conv_ws3 / conv_g1 either way); end to end the two model forms agree to summation-order noise and both match
    the oracle."""
    from aicovergen_amd import _lib
    torch.manual_seed(3)
    E, T = 64, 6000 if dev.big else 700
    h = dev.t(torch.randn(1, E, T))
    qw, qb, kvw, kvb = torch.randn(E, E) * 0.1, torch.randn(E), torch.randn(2 * E, E) * 0.1, torch.randn(2 * E)
    q = ops.conv(h, ops.PackedConv(qw, qb, device=dev.device), out_scale=0.125)
    k1 = _lib.last_launch()
    qkv = ops.conv(h, ops.PackedConv(torch.cat([qw * 0.125, kvw], 0), torch.cat([qb * 0.125, kvb], 0), device=dev.device))
    if _lib.last_launch() == k1:
        assert torch.equal(q, qkv[:, :E])
    assert rel_rms(q, qkv[:, :E]) < 1e-6
    cfg = dict(weights.HUBERT_TINY, heads=1)
    sd = weights.hubert_state_dict(cfg, 77)
    wav = dev.t(torch.randn(1, 6000) * 0.3)
    one = HubertModel(sd, cfg).to(dev.device)
    two = HubertModel(sd, cfg).to(dev.device)
    two.merge_qkv = False
    assert "qkv" in one._prepare()["layers"][0] and "q" in two._prepare()["layers"][0]
    ya, _ = one.extract_features(source=wav, padding_mask=None, output_layer=12)
    yb, _ = two.extract_features(source=wav, padding_mask=None, output_layer=12)
    assert rel_rms(ya, yb) < 2e-6
    with torch.no_grad():
        assert rel_rms(ya, ohub.extract_features(sd, cfg, wav.cpu(), 12)) < 1e-4


def test_hubert_cfg_inferred_from_shapes():
    sd = weights.hubert_state_dict(weights.HUBERT_TINY, 1)
    cfg = _infer_cfg(sd)
    for k in ("conv_dim", "conv_kernel", "embed", "ffn", "layers", "pos_k", "pos_groups", "final_dim"):
        assert cfg[k] == weights.HUBERT_TINY[k], k


@pytest.mark.gpu
def test_hubert_base_matches_oracle():
    """HuBERT-base sized model (94 M parameters), 4 s of audio, layer 12 and layer 9 + final_proj (v1 path)."""
    import conftest
    conftest._bind("hip")
    cfg = weights.HUBERT_BASE
    sd = weights.hubert_state_dict(cfg, 1234)
    m = HubertModel(sd, cfg).to("cuda:0")
    wav = torch.from_numpy(vocal_like(4.0, 16000, seed=3)).unsqueeze(0)
    y, _ = m.extract_features(source=wav, padding_mask=None, output_layer=12)
    with torch.no_grad():
        ref = ohub.extract_features(sd, cfg, wav, 12)
        ref9 = ohub.final_proj(sd, ohub.extract_features(sd, cfg, wav, 9))
    assert rel_rms(y, ref) < 1e-4
    y9, _ = m.extract_features(source=wav, padding_mask=None, output_layer=9)
    assert rel_rms(m.final_proj(y9), ref9) < 1e-4


def test_mel_filterbank_table():
    fb = mel_filterbank()
    assert fb.shape == (128, 513) and fb.dtype == np.float32
    assert np.array_equal(fb, orm.mel_filterbank())
    assert (fb >= 0).all() and (fb.sum(1) > 0).all()


def test_rmvpe_tiny_matches_oracle(dev):
    sd = weights.rmvpe_state_dict(weights.RMVPE_TINY, 1234)
    r = RMVPE(None, False, dev.device, state_dict=sd)
    audio = vocal_like(1.0, 16000, seed=5)
    mel = r.mel_extractor(torch.from_numpy(audio)[None])
    omel = orm.log_mel(torch.from_numpy(audio)[None], torch.from_numpy(orm.mel_filterbank()))
    assert mel.shape == omel.shape == (1, 128, 101)
    assert (mel.cpu() - omel).abs().max() < 2e-4      # log of clamped small mels amplifies fp32 FFT differences
    hid = r.mel2hidden(mel)[0].cpu().numpy()
    of0, ohid = orm.infer_from_audio(sd, audio, 0.03)
    assert hid.shape == ohid.shape == (101, 360)
    assert np.abs(hid - ohid).max() < 1e-4
    f0 = r.infer_from_audio(audio, 0.03)
    agree = hid.argmax(1) == ohid.argmax(1)
    assert agree.mean() > 0.98, "salience argmax agreement %.3f" % agree.mean()
    assert np.abs(f0 - of0)[agree].max() < 1e-2


def test_salience_decode_bit_exact(dev):
    """Given identical salience the decode is bit-equal to numpy: argmax index, float64 9-bin local average
    (rmvpe.py:385-409).  f0 = 10 * 2^(cents/1200) may differ in the last ulp of libm's pow."""
    rng = np.random.default_rng(1)
    T = 20000 if dev.big else 500
    sal = rng.random((T, 360)).astype(np.float32) ** 4
    sal[3] = 0.01                                     # below threshold -> 0
    sal[7, :5] = [0.9, 0.9, 0.1, 0.2, 0.3]            # tie: first maximum wins; window clipped at the low edge
    sal[8, 355:] = [0.1, 0.2, 0.3, 0.95, 0.95]        # window clipped at the high edge
    cents, f0, center = ops.salience_decode(dev.t(torch.from_numpy(sal)), 0.03, want_center=True)
    assert np.array_equal(center.cpu().numpy(), sal.argmax(1))
    assert np.array_equal(cents.cpu().numpy(), orm.to_local_average_cents(sal, 0.03))
    ref_f0 = orm.decode(sal, 0.03)
    got = f0.cpu().numpy()
    assert np.array_equal(got == 0, ref_f0 == 0)
    assert np.max(np.abs(got - ref_f0) / np.maximum(ref_f0, 1e-9)) < 1e-15


def test_f0_coarse_bit_exact(dev):
    """vc_infer_pipeline.py:346,361-368 incl. round-half-even and the 1 / 255 clamps."""
    rng = np.random.default_rng(2)
    f0 = np.concatenate([rng.random(2000) * 1200, np.zeros(50), [49.9, 50.0, 1100.0, 1100.1, 5000.0]])
    for key in (0, 3, -12):
        out, coarse = ops.f0_coarse(dev.t(torch.from_numpy(f0)), pow(2, key / 12), 1127 * np.log(1 + 50 / 700),
                                    1127 * np.log(1 + 1100 / 700))
        oc, ob = orm.f0_to_coarse(f0, key)
        assert np.array_equal(coarse.cpu().numpy(), oc)
        assert np.array_equal(out.cpu().numpy(), ob)
        assert coarse.min() >= 1 and coarse.max() <= 255


@pytest.mark.gpu
def test_rmvpe_full_matches_oracle():
    """Full-size RMVPE (90 M parameters) on 3 s of audio vs the oracle on the host CPU."""
    import conftest
    conftest._bind("hip")
    sd = weights.rmvpe_state_dict(weights.RMVPE_FULL, 1234)
    r = RMVPE(None, False, "cuda:0", state_dict=sd)
    audio = vocal_like(3.0, 16000, seed=9)
    of0, ohid = orm.infer_from_audio(sd, audio, 0.03)
    mel = r.mel_extractor(torch.from_numpy(audio)[None])
    hid = r.mel2hidden(mel)[0].cpu().numpy()
    assert np.abs(hid - ohid).max() < 1e-3
    agree = hid.argmax(1) == ohid.argmax(1)
    assert agree.mean() > 0.98, "salience argmax agreement %.3f" % agree.mean()
    f0 = r.infer_from_audio(audio, 0.03)
    assert np.abs(f0 - of0)[agree].max() / max(1.0, of0.max()) < 1e-3


@pytest.mark.parametrize("hidden,T", [(64, 50), (256, 33)])
def test_gru_kernels_match_torch(dev, hidden, T):
    """All recurrence kernels (single workgroup per direction; two, and for hidden 256 four, co-resident workgroups exchanging h
    through tagged granules) vs torch.nn.GRU semantics written out in oracle/rmvpe.py::bigru."""
    torch.manual_seed(hidden)
    if dev.big:
        T = T * 40 + 3
    x = torch.randn(1, T, 384)
    sd = {}
    for suf in ("", "_reverse"):
        sd["fc.0.gru.weight_ih_l0" + suf] = torch.randn(3 * hidden, 384) / hidden ** 0.5
        sd["fc.0.gru.weight_hh_l0" + suf] = torch.randn(3 * hidden, hidden) / hidden ** 0.5
        sd["fc.0.gru.bias_ih_l0" + suf] = torch.randn(3 * hidden) * 0.1
        sd["fc.0.gru.bias_hh_l0" + suf] = torch.randn(3 * hidden) * 0.1
    ref = orm.bigru(sd, x)[0].t()                                     # (2*hidden, T)
    gi = torch.cat([torch.nn.functional.linear(x[0], sd["fc.0.gru.weight_ih_l0" + s], sd["fc.0.gru.bias_ih_l0" + s]).t()
                    for s in ("", "_reverse")], 0).contiguous()
    whh_t = torch.stack([sd["fc.0.gru.weight_hh_l0"].t().contiguous(), sd["fc.0.gru.weight_hh_l0_reverse"].t().contiguous()])
    bhh = torch.cat([sd["fc.0.gru.bias_hh_l0"], sd["fc.0.gru.bias_hh_l0_reverse"]])
    old = ops.GRU_WORKGROUPS
    try:
        for multi, nwg in ((False, 0), (True, 2), (True, 4)):
            ops.GRU_WORKGROUPS = nwg
            got = ops.gru_bidir(dev.t(gi), dev.t(whh_t.contiguous()), dev.t(bhh), hidden, two_workgroups=multi)
            ops.gru_check_pending()
            assert rel_rms(got, ref) < 1e-5, "workgroups per direction: %d" % (nwg or 1)
    finally:
        ops.GRU_WORKGROUPS = old


@pytest.mark.parametrize("hidden,multi", [(256, True), (256, False), (64, False)])
def test_gru_in_segments_is_bit_identical(dev, hidden, multi):
    """The recurrence cut into segments of steps with the hidden state carried (ops.GruSegments -> aicg_gru_bidir_4wg_seg /
    aicg_gru_bidir_seg) is the SAME arithmetic step by step: bit-identical to one launch, for uneven segment lengths, and the range
    both directions have passed grows from the middle of the track.  (What the multi-GPU pipeline's progressive f0 rests on)"""
    torch.manual_seed(hidden + int(multi))
    T = 75 if not dev.big else 5003
    gi = dev.t(torch.randn(6 * hidden, T) * 0.5)
    whh = dev.t(torch.randn(2, hidden, 3 * hidden) * 0.05)
    bhh = dev.t(torch.randn(6 * hidden) * 0.1)
    ref = ops.gru_bidir(gi, whh, bhh, hidden, two_workgroups=multi)
    seg = ops.GruSegments(gi, whh, bhh, hidden, two_workgroups=multi)
    cuts = [16, 17, T // 2, T - 11, T] if not dev.big else [32, 1000, 1001, T // 2, 4096, T]
    ranges = []
    for s1 in cuts:
        seg.run(s1)
        ranges.append(seg.ready())
    dev.sync()
    assert not ops.gru_timed_out()
    assert torch.equal(seg.out, ref)
    assert ranges[0][0] >= ranges[0][1]                       # nothing is final before the directions meet ...
    assert ranges[-1] == (0, T)                               # ... everything after the last step
    assert all(a1 <= a0 and b1 >= b0 for (a0, b0), (a1, b1) in zip(ranges, ranges[1:]))


def test_rmvpe_progressive_f0_equals_one_launch(dev):
    """RMVPE.infer_progressive: f0 published range by range as both GRU directions pass the frames, middle of the track first;
    the union is the one-launch result bit for bit on the emulator (on hardware the classifier GEMM over a frame range may take
    other tiles than over the whole track: fp32 summation order, 1e-6)."""
    r = RMVPE(None, False, dev.device, state_dict=weights.small_model_set(1234)["rmvpe_sd"])
    audio = torch.from_numpy(vocal_like(3.0, 16000, 77))
    f0 = r.infer_from_audio_device(audio, 0.03).cpu().numpy()
    got = np.full_like(f0, -1.0)
    calls = []

    def on_f0(lo, hi, v):
        got[lo:hi] = v.cpu().numpy()
        calls.append((lo, hi))
    n = r.infer_progressive(audio, 0.03, 5, on_f0)
    dev.sync()
    assert n == len(f0) and (got >= 0).all()
    assert calls[0][0] > 0 and calls[0][1] < n                # the first range is in the middle
    if dev.kind == "emu":
        assert np.array_equal(got, f0)
    else:
        assert np.array_equal(got > 0, f0 > 0) and np.allclose(got, f0, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.skipif(not os.environ.get("AICG_REAL_HUBERT"), reason="set AICG_REAL_HUBERT=<hubert_base.pt> (and install fairseq) to pin HuBERT")
def test_real_hubert_checkpoint_matches_fairseq():
    """The one-command pin of row a6: fairseq's own HubertModel.extract_features (what the reference calls,
    src/vc_infer_pipeline.py:398-406) against this implementation loaded from the same checkpoint."""
    import conftest
    fairseq = pytest.importorskip("fairseq")
    conftest._bind("hip")
    from aicovergen_amd.rvc import load_hubert
    path = os.environ["AICG_REAL_HUBERT"]
    models, _, _ = fairseq.checkpoint_utils.load_model_ensemble_and_task([path], suffix="")
    ref_model = models[0].float().eval()
    wav = torch.from_numpy(vocal_like(4.0, 16000, seed=3)).unsqueeze(0)
    with torch.no_grad():
        ref = ref_model.extract_features(source=wav, padding_mask=torch.zeros_like(wav, dtype=torch.bool), output_layer=12)[0]
    got = load_hubert("cuda:0", False, path).extract_features(source=wav, padding_mask=None, output_layer=12)[0]
    assert rel_rms(got, ref) < 1e-4
