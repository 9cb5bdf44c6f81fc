"""The oracle restatement (oracle/synth.py) pinned against outputs of the reference itself
(tests/golden/*.npz, generated in the build container from /root/reference by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_rms
from oracle import hubert as ohub
from oracle import rmvpe as orm
from oracle import synth
from synthetic import weights
from synthetic.inputs import synth_inputs, vocal_like

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name,cfg,T", [("synth_tiny_T24", weights.SYNTH_CFG_TINY, 24),
                                        ("synth_40k_T16", weights.SYNTH_CFG_40K_V2, 16)])
def test_oracle_synth_matches_reference_golden(name, cfg, T):
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    sd = weights.synth_state_dict(cfg, int(gold["seed"][0]))
    phone, pitch, f0, nz, ns = synth_inputs(cfg, T, int(gold["seed"][0]) + 1)
    with torch.no_grad():
        o, (z, z_p, m_p, logs_p) = synth.synth_infer(sd, cfg, phone, pitch, f0, torch.tensor([1]), nz, ns)
    assert rel_rms(m_p[0], torch.from_numpy(gold["m_p"])) < 1e-5
    assert rel_rms(z[0], torch.from_numpy(gold["z"])) < 1e-5
    assert rel_rms(o[0, 0], torch.from_numpy(gold["audio"])) < 1e-5


@pytest.mark.parametrize("name,cfg", [("hubert_tiny_1s", weights.HUBERT_TINY), ("hubert_base_1s", weights.HUBERT_BASE)])
def test_oracle_hubert_matches_transformers_golden(name, cfg):
    """fairseq is not installable here; the HuBERT restatement is pinned against transformers.HubertModel."""
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    seed = int(gold["seed"][0])
    sd = weights.hubert_state_dict(cfg, seed)
    wav = torch.from_numpy(vocal_like(float(gold["seconds"][0]), 16000, seed + 2)).unsqueeze(0)
    with torch.no_grad():
        y = ohub.extract_features(sd, cfg, wav, cfg["layers"])
        y9 = ohub.extract_features(sd, cfg, wav, min(9, cfg["layers"]))
    assert rel_rms(y[0], torch.from_numpy(gold["last"])) < 1e-5
    assert rel_rms(y9[0], torch.from_numpy(gold["layer9"])) < 1e-5


@pytest.mark.parametrize("name,cfg", [("rmvpe_tiny_1s", weights.RMVPE_TINY), ("rmvpe_full_1s", weights.RMVPE_FULL)])
def test_oracle_rmvpe_matches_reference_golden(name, cfg):
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    seed = int(gold["seed"][0])
    sd = weights.rmvpe_state_dict(cfg, seed)
    audio = vocal_like(float(gold["seconds"][0]), 16000, seed + 3)
    f0, hidden = orm.infer_from_audio(sd, audio, 0.03)
    assert np.abs(hidden - gold["hidden"].astype(np.float32)).max() < 2e-3   # fixture stored as fp16
    assert np.array_equal(hidden.argmax(1), gold["argmax"])
    assert np.allclose(f0, gold["f0"], rtol=1e-6, atol=1e-6)
    omel = orm.log_mel(torch.from_numpy(audio)[None], torch.from_numpy(orm.mel_filterbank()))
    assert (omel[0] - torch.from_numpy(gold["mel"])).abs().max() < 1e-4


def test_oracle_decode_matches_reference_loop():
    """to_local_average_cents restated vectorised vs the reference's per-frame Python loop semantics
    (window of 9 around the argmax on the zero-padded salience, rmvpe.py:385-409)."""
    rng = np.random.default_rng(0)
    sal = rng.random((64, 360)).astype(np.float32) ** 3
    cm = np.pad(20 * np.arange(360) + 1997.3794084376191, (4, 4))
    sp = np.pad(sal, ((0, 0), (4, 4)))
    want = []
    for i in range(sal.shape[0]):
        c = int(np.argmax(sal[i])) + 4
        s9, m9 = sp[i, c - 4:c + 5], cm[c - 4:c + 5]
        want.append(np.sum(s9 * m9) / np.sum(s9) if sp[i].max() > 0.05 else 0.0)
    assert np.allclose(orm.to_local_average_cents(sal, 0.05), np.array(want), rtol=1e-12)


# ---- MDX-Net separator path: the oracle's restatement of src/mdx.py against outputs of the reference module itself -------
# (tests/golden/make_mdx_golden.py ran the reference's MDXModel / MDX / run_mdx with only onnxruntime, librosa and soundfile
# stubbed; the stub network is the restated U-Net, so everything AROUND the network is pinned here)
def _mdx_gold():
    import numpy as np
    g = np.load(os.path.join(GOLD, "mdx_ref_tiny.npz"))
    cfg = dict(weights.MDX_TINY, n_fft=2048, dim_t=16)
    return g, cfg, weights.mdx_state_dict(weights.MDX_TINY, 7)


def test_oracle_mdx_stft_istft_match_reference():
    from oracle import mdxnet
    g, cfg, _ = _mdx_gold()
    spec = mdxnet.stft(torch.from_numpy(g["stft_in"]), cfg["n_fft"], 1024, cfg["dim_f"])
    assert torch.equal(spec, torch.from_numpy(g["stft_out"]))                      # same torch ops on the same CPU
    assert torch.equal(mdxnet.istft(spec, cfg["n_fft"], 1024), torch.from_numpy(g["istft_out"]))


def test_oracle_mdx_window_bookkeeping_matches_reference():
    import numpy as np
    from oracle import mdxnet
    g, cfg, _ = _mdx_gold()
    mix, pad, trim = mdxnet.pad_wave(g["pad_in"], cfg["n_fft"], 1024 * (cfg["dim_t"] - 1))
    assert [pad, trim] == g["pad_trim"].tolist()
    assert np.array_equal(mix.numpy(), g["pad_windows"])
    segs = mdxnet.segment(g["pad_in"], False, 15000)
    assert [s.shape[-1] for s in segs] == g["seg_lens"].tolist()
    assert mdxnet.segment(segs, True, 15000).shape == g["seg_joined"].shape       # the reference's own short-wave edge case


def test_oracle_run_mdx_matches_reference():
    import numpy as np
    from oracle import mdxnet
    g, cfg, sd = _mdx_gold()
    wave = g["wave"].astype(np.float32)
    norm = wave / max(np.max(wave), abs(np.min(wave)))
    assert np.array_equal(mdxnet.process_wave(sd, cfg, norm, 2).astype(np.float32), g["process_wave"])
    for denoise, tag in ((True, "dn"), (False, "plain")):
        main, inv = mdxnet.run_mdx_arrays(sd, cfg, wave, denoise, 1.021, 2)
        assert np.abs(main - g["main_" + tag]).max() < 1e-6
        assert np.abs(inv - g["inv_" + tag]).max() < 1e-6
