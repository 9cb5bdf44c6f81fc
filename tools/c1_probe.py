"""BASELINE C1 (30 s through VC.pipeline, full-size networks) against the reference's own output: waveform relative RMS / LSB rates
and the f0 track's distance to the reference's, in one line -- for A/B runs over the AICG_* switches (GPU box)."""
import os, sys, numpy as np, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import conftest
conftest._bind("hip")
from test_pipeline import build, noise_fn_for
from synthetic import weights
from synthetic.inputs import vocal_like
gold = np.load("tests/golden/pipeline_c1_30s.npz")
seed, x = int(gold["seed"][0]), tuple(int(v) for v in gold["x"])
nets = weights.full_model_set(seed)
audio = vocal_like(float(gold["seconds"][0]), 16000, seed + 5)
dev = conftest.Dev("hip")
vc, hub, net_g, tgt_sr = build(dev, nets, x)
out = vc.pipeline(hub, net_g, 0, audio, "x.wav", [0, 0, 0], 0, "rmvpe", "", 0.5, 1, 3, tgt_sr, 0, 0.25, "v2", 0.33, 128, noise_fn=noise_fn_for(nets))
ref = gold["audio"]
diff = np.abs(out.astype(np.int32) - ref.astype(np.int32))
rel = np.sqrt(np.sum(diff.astype(np.float64) ** 2) / np.sum(ref.astype(np.float64) ** 2))
_, audio_pad, opt_ts, p_len = vc.plan(audio)
coarse, f0 = vc.get_f0("x.wav", audio_pad, p_len, 0, "rmvpe", 3, 128)
n = min(len(f0), len(gold["f0"])); v = (f0[:n] > 0) & (gold["f0"][:n] > 0)
rd = np.abs(f0[:n][v] / gold["f0"][:n][v] - 1)
print("M16H=%s: rel rms %.3e max %d <=1LSB %.4f | f0 rel diff max %.3e rms %.3e, bins differing %d" % (os.environ.get("AICG_CONV_M16H", "1"), rel, diff.max(), (diff <= 1).mean(), rd.max(), np.sqrt((rd ** 2).mean()), int((coarse[:n] != gold["coarse"][:n]).sum())))
