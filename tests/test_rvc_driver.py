"""The file-level RVC driver behind the reference's src/rvc.py surface (Config, load_hubert, get_vc, rvc_infer): checkpoints
on disk -> models -> WAV in -> WAV out, the way src/main.py:voice_change drives it (main.py:186-199)."""
import numpy as np
import torch
from scipy.io import wavfile

from aicovergen_amd import rvc
from synthetic import weights
from synthetic.inputs import vocal_like


def _write_models(tmp_path, nets):
    hub = tmp_path / "hubert_base.pt"
    torch.save({"model": nets["hubert_sd"], "cfg": {"note": "fairseq keeps an omegaconf tree here"}, "args": None}, str(hub))
    pth = tmp_path / "voice.pth"
    cfg = list(nets["synth_cfg"])
    cfg[-3] = 109  # spk_embed_dim as stored by training; get_vc overwrites it from emb_g (src/rvc.py:124)
    torch.save({"config": cfg, "weight": nets["synth_sd"], "f0": 1, "version": "v2", "info": "200epoch"}, str(pth))
    rm = tmp_path / "rmvpe.pt"
    torch.save(nets["rmvpe_sd"], str(rm))
    return str(hub), str(pth), str(rm)


def test_config_presets_follow_the_reference():
    cfg = rvc.Config("cpu", False)
    if not torch.cuda.is_available():
        # "No supported N-card found": the reference falls back to cpu AND flips is_half on (src/rvc.py:68-71)
        assert cfg.device == "cpu" and cfg.is_half is True
    assert (cfg.x_pad, cfg.x_query, cfg.x_center, cfg.x_max) in ((3, 10, 60, 65), (1, 6, 38, 41), (1, 5, 30, 32))


def test_rvc_infer_files_in_files_out(dev, tmp_path):
    nets = weights.small_model_set()
    hub_path, pth_path, rm_path = _write_models(tmp_path, nets)
    config = rvc.Config(dev.device, False)
    config.device, config.is_half = dev.device, False
    config.x_pad, config.x_query, config.x_center, config.x_max = 1, 1, 1, 2   # keep the emulator run short
    hubert = rvc.load_hubert(dev.device, False, hub_path)
    cpt, version, net_g, tgt_sr, vc = rvc.get_vc(dev.device, False, config, pth_path)
    assert version == "v2" and tgt_sr == nets["synth_cfg"][-1]
    assert cpt["config"][-3] == nets["synth_sd"]["emb_g.weight"].shape[0]
    vc.rmvpe_path = rm_path
    audio = (vocal_like(1.3, 16000, 21) * 0.5).astype(np.float32)
    wav_in, wav_out = tmp_path / "in.wav", tmp_path / "out.wav"
    wavfile.write(str(wav_in), 16000, np.rint(audio * 32767).astype(np.int16))

    torch.manual_seed(5)
    rvc.rvc_infer("", 0.5, str(wav_in), str(wav_out), 0, "rmvpe", cpt, version, net_g, 3, tgt_sr, 0.25, 0.33, 128, vc, hubert)
    sr, got = wavfile.read(str(wav_out))
    assert sr == tgt_sr and got.dtype == np.int16 and got.ndim == 1

    # the same call made directly on the pipeline with the same random draws
    from aicovergen_amd.my_utils import load_audio
    torch.manual_seed(5)
    ref = vc.pipeline(hubert, net_g, 0, load_audio(str(wav_in), 16000), str(wav_in), [0, 0, 0], 0, "rmvpe", "", 0.5, 1, 3,
                      tgt_sr, 0, 0.25, version, 0.33, 128)
    assert got.shape == ref.shape and np.array_equal(got, ref)
    assert abs(len(got) / tgt_sr - 1.3) < 0.06 and np.abs(got).max() > 100
