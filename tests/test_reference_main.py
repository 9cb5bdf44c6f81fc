"""The reference's REAL src/main.py driven through the shadow modules (VERDICT r5 "missing" #3; north_star: "drops into src/main.py
unchanged").  Build container only: the test imports /root/reference/src/main.py in place and calls ITS OWN `preprocess_song`
(main.py:165-190: three run_mdx calls -- vocals / instrumental, main / backup vocals, de-reverb) and `voice_change` (main.py:193-203:
Config(device, True) -> load_hubert -> get_vc -> rvc_infer), with <repo>/src ahead of the reference's src/ on sys.path, so that the
`from mdx import run_mdx` / `from rvc import ...` lines of the unmodified file resolve to this implementation.

What is stubbed is what the hot path never touches and this image does not have: gradio, sox, yt_dlp, pedalboard, pydub, soundfile
(UI, download, effects, mixing -- SURVEY 2 out of scope) and librosa.load for main.py's mono check (convert_to_stereo, :125-136).
Models are the seeded miniature set written to disk in the formats main.py's directories hold (.onnx, model_data.json entry,
hubert_base.pt, <voice>/*.pth, rmvpe.pt); kernels run on the host emulator.  Asserted: the files run_mdx writes carry the
reference's names (src/mdx.py:262-283) and are byte-identical to direct calls of aicovergen_amd.mdx.run_mdx with the same arguments,
and the WAV rvc_infer writes is byte-identical to a direct aicovergen_amd.rvc call with the same random draws."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MAIN = "/root/reference/src/main.py"

CHILD = r'''
import filecmp, json, os, shutil, sys, types
ROOT, REF_SRC, TMP = %r, %r, %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import conftest
conftest._bind("emu")
from aicovergen_amd import audio_io
from synthetic import weights
from synthetic.inputs import song_like


class _Absent:
    """Stand-in for a UI / effects / download class main.py names at import time; the hot path never instantiates it."""
    def __init__(self, *a, **k): pass
    def __call__(self, *a, **k): raise AssertionError("out-of-scope dependency reached from the hot path")


def stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m
    return m

stub("gradio", Progress=_Absent, Error=Exception)
stub("sox", Transformer=_Absent)
stub("yt_dlp", YoutubeDL=_Absent)
stub("pedalboard", Pedalboard=_Absent, Reverb=_Absent, Compressor=_Absent, HighpassFilter=_Absent)
stub("pedalboard.io", AudioFile=_Absent)
stub("pydub", AudioSegment=_Absent)
stub("soundfile", read=_Absent(), write=_Absent())
# convert_to_stereo (main.py:125-136) asks librosa for the channel layout: (channels, n) for a stereo file
stub("librosa", load=lambda path, mono=False, sr=44100: audio_io.load_wav(path, sr, mono=mono))
# run_mdx queries the card's memory unconditionally, like the reference (src/mdx.py:244-247); the emulator host has no card
torch.cuda.get_device_properties = lambda d=None: types.SimpleNamespace(total_memory=64 << 30, name="emulated")

# ---- the reference's own main.py, shadows first (what src/run_main.py arranges for `python main.py`)
sys.path[:0] = [os.path.join(ROOT, "src"), REF_SRC]
import main
import aicovergen_amd.mdx, aicovergen_amd.rvc
assert os.path.samefile(main.__file__, os.path.join(REF_SRC, "main.py"))
assert main.run_mdx is aicovergen_amd.mdx.run_mdx and main.rvc_infer is aicovergen_amd.rvc.rvc_infer
assert main.Config is aicovergen_amd.rvc.Config and main.get_vc is aicovergen_amd.rvc.get_vc

# ---- main.py's directories, in a scratch tree
mdx_dir, rvc_dir, out_dir = (os.path.join(TMP, d) for d in ("mdxnet_models", "rvc_models", "song_output"))
for d in (mdx_dir, os.path.join(rvc_dir, "Voice"), out_dir):
    os.makedirs(d)
main.mdxnet_models_dir, main.rvc_models_dir, main.output_dir = mdx_dir, rvc_dir, out_dir
fixture = os.path.join(ROOT, "tests", "golden", "mdx_tiny.onnx")
for name in ("UVR-MDX-NET-Voc_FT.onnx", "UVR_MDXNET_KARA_2.onnx", "Reverb_HQ_By_FoxJoy.onnx"):      # main.py:182,185,188
    shutil.copy(fixture, os.path.join(mdx_dir, name))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_onnx_weights import CFG
params = {aicovergen_amd.mdx.MDX.get_hash(fixture): {"mdx_dim_f_set": CFG["dim_f"], "mdx_dim_t_set": 4, "mdx_n_fft_scale_set": 2048,
                                                      "primary_stem": "Vocals", "compensate": 1.021}}
json.dump(params, open(os.path.join(mdx_dir, "model_data.json"), "w"))
nets = weights.small_model_set()
torch.save({"model": nets["hubert_sd"], "cfg": {"note": "fairseq keeps an omegaconf tree here"}, "args": None}, os.path.join(rvc_dir, "hubert_base.pt"))
cfg = list(nets["synth_cfg"]); cfg[-3] = 109
torch.save({"config": cfg, "weight": nets["synth_sd"], "f0": 1, "version": "v2", "info": "seeded"}, os.path.join(rvc_dir, "Voice", "voice.pth"))
torch.save(nets["rmvpe_sd"], os.path.join(rvc_dir, "rmvpe.pt"))
# the emulator run stays short: this implementation's chunk preset for is_half (what main.py's Config(device, True) selects)
aicovergen_amd.rvc._PRESET_HALF = (1, 1, 1, 2)

song = os.path.join(TMP, "song.wav")
audio_io.write_wav_pcm16(song, (song_like(0.8, 44100, seed=9).astype(np.float32) * 0.6).T, 44100)
with open(os.path.join(mdx_dir, "model_data.json")) as f:        # main.py:246-247
    mdx_model_params = json.load(f)
song_id = main.get_hash(song)
os.makedirs(os.path.join(out_dir, song_id))

# ---- main.py's own preprocess_song: three run_mdx calls
orig, vocals, inst, main_v, backup_v, dereverb = main.preprocess_song(song, mdx_model_params, song_id, 0, "local")
sdir = os.path.join(out_dir, song_id)
assert orig == song and os.path.exists(song)                                            # keep_orig=True for a local file
assert (vocals, inst) == (os.path.join(sdir, "song_Vocals.wav"), os.path.join(sdir, "song_Instrumental.wav"))
assert (backup_v, main_v) == (os.path.join(sdir, "song_Vocals_Backup.wav"), os.path.join(sdir, "song_Vocals_Main.wav"))
assert dereverb == os.path.join(sdir, "song_Vocals_Main_DeReverb.wav")
assert sorted(os.listdir(sdir)) == ["song_Instrumental.wav", "song_Vocals.wav", "song_Vocals_Backup.wav", "song_Vocals_Main.wav",
                                    "song_Vocals_Main_DeReverb.wav"]
assert main.get_audio_paths(sdir) == (os.path.join(sdir, "song.wav"), inst, dereverb, backup_v)   # main.py:105-122 finds them again by suffix
# ... and the same three calls made directly
d2 = os.path.join(TMP, "direct"); os.makedirs(d2)
rm = aicovergen_amd.mdx.run_mdx
v2, i2 = rm(mdx_model_params, d2, os.path.join(mdx_dir, "UVR-MDX-NET-Voc_FT.onnx"), song, denoise=True, keep_orig=True)
b2, m2 = rm(mdx_model_params, d2, os.path.join(mdx_dir, "UVR_MDXNET_KARA_2.onnx"), v2, suffix="Backup", invert_suffix="Main", denoise=True)
_, dr2 = rm(mdx_model_params, d2, os.path.join(mdx_dir, "Reverb_HQ_By_FoxJoy.onnx"), m2, invert_suffix="DeReverb", exclude_main=True, denoise=True)
for a, b in ((vocals, v2), (inst, i2), (backup_v, b2), (main_v, m2), (dereverb, dr2)):
    assert os.path.basename(a) == os.path.basename(b) and filecmp.cmp(a, b, shallow=False), (a, b)
x, sr = audio_io.load_wav(dereverb, 44100, mono=False)
assert sr == 44100 and x.shape[0] == 2 and np.abs(x).max() > 1e-3

# ---- main.py's own voice_change: Config('cuda:0', True) -> load_hubert -> get_vc -> rvc_infer
ai_vocals = os.path.join(sdir, "song_Voice_p0_i0.5_fr3_rms0.25_pro0.33_rmvpe.wav")     # the name main.py:287 builds
torch.manual_seed(11)
main.voice_change("Voice", dereverb, ai_vocals, 0, "rmvpe", 0.5, 3, 0.25, 0.33, 128, 0)
from scipy.io import wavfile
sr_out, got = wavfile.read(ai_vocals)
assert sr_out == nets["synth_cfg"][-1] and got.dtype == np.int16 and got.ndim == 1 and np.abs(got).max() > 100
rvc = aicovergen_amd.rvc
config = rvc.Config("cuda:0", True)
hub = rvc.load_hubert("cuda:0", config.is_half, os.path.join(rvc_dir, "hubert_base.pt"))
cpt, version, net_g, tgt_sr, vc = rvc.get_vc("cuda:0", config.is_half, config, os.path.join(rvc_dir, "Voice", "voice.pth"))
direct = os.path.join(TMP, "direct.wav")
torch.manual_seed(11)
rvc.rvc_infer("", 0.5, dereverb, direct, 0, "rmvpe", cpt, version, net_g, 3, tgt_sr, 0.25, 0.33, 128, vc, hub)
assert filecmp.cmp(ai_vocals, direct, shallow=False)
assert abs(len(got) / sr_out - 0.8) < 0.06
print("reference main.py through the shadows ok")
'''


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference checkout exists in the build container only")
def test_reference_main_preprocess_song_and_voice_change_through_the_shadows(tmp_path):
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, os.path.dirname(REF_MAIN), str(tmp_path))], capture_output=True, text=True,
                       cwd="/", timeout=1500)
    assert r.returncode == 0 and "reference main.py through the shadows ok" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]
