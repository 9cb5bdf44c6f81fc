"""Pin the reference's own crepe post-processing: run /root/reference/src/vc_infer_pipeline.py's
get_f0_crepe_computation / get_f0_official_crepe_computation / get_f0_hybrid_computation with `torchcrepe.predict` replaced by
a deterministic track generator (synthetic.inputs.fake_crepe_tracks: wandering pitch, sub-0.001 entries, low-periodicity
stretches) and torchcrepe.filter.median / .mean by the restated window filters.  What the fixture pins is the reference's code around torchcrepe: quantile normalisation, the
< 0.001 -> NaN gate, the np.interp resize to p_len, nan_to_num, the periodicity gate, f0[1:], np.nanmedian.

    python tests/golden/make_crepe_golden.py   ->  tests/golden/crepe_post_ref.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import transformers  # noqa: F401,E402  (before librosa is stubbed)
from oracle import pipeline as opipe  # noqa: E402
from oracle import rmvpe as orm  # noqa: E402
from synthetic.inputs import fake_crepe_tracks, vocal_like  # noqa: E402

np.int = int


def window_filter(signals, win, fn):
    out = torch.empty_like(signals)
    for i in range(signals.size(1)):
        lo, hi = max(0, i - win // 2), min(signals.size(1), i + win // 2 + 1)
        w = signals[:, lo:hi]
        vals = []
        for row in w:
            v = row[~torch.isnan(row)]
            vals.append(fn(v) if len(v) else torch.tensor(float("nan")))
        out[:, i] = torch.stack(vals)
    return out


tc = types.ModuleType("torchcrepe")
tc.filter = types.ModuleType("torchcrepe.filter")
tc.filter.median = lambda s, w: window_filter(s, w, lambda v: v.sort().values[(len(v) - 1) // 2])
tc.filter.mean = lambda s, w: window_filter(s, w, lambda v: v.mean())


def predict(audio, sr, hop, fmin, fmax, model, batch_size=None, device=None, pad=True, return_periodicity=False):
    assert sr == 16000 and pad
    n = 1 + audio.shape[1] // hop
    pitch, pd = fake_crepe_tracks(n, 1000 + hop)
    p = torch.from_numpy(pitch)[None]
    return (p, torch.from_numpy(pd)[None]) if return_periodicity else p


tc.predict = predict
sys.modules["torchcrepe"] = tc
sys.modules["torchcrepe.filter"] = tc.filter
for mod in ("faiss", "parselmouth", "pyworld"):
    sys.modules.setdefault(mod, types.ModuleType(mod))
lib = types.ModuleType("librosa")
lib.filters = types.ModuleType("librosa.filters")
lib.feature = types.ModuleType("librosa.feature")
lib.filters.mel = lambda sr, n_fft, n_mels, fmin, fmax, htk: orm.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
lib.feature.rms = lambda y, frame_length, hop_length: opipe.rms_frames(y, frame_length, hop_length)
lib.__spec__ = None
sys.modules["librosa"], sys.modules["librosa.filters"], sys.modules["librosa.feature"] = lib, lib.filters, lib.feature
sys.path.insert(0, "/root/reference/src")
import vc_infer_pipeline as ref_vc  # noqa: E402


class Cfg:
    x_pad, x_query, x_center, x_max, is_half, device = 1, 1, 1, 2, False, "cpu"


vc = ref_vc.VC(40000, Cfg())
vc.get_optimal_torch_device = lambda index=0: torch.device("cpu")
seconds, seed = 2.4, 77
audio = vocal_like(seconds, 16000, seed).astype(np.float64)
audio[9000:14000] *= 0.002            # a near-silent stretch: low periodicity, gated frames
p_len = len(audio) // 160
out = dict(seed=np.array([seed]), seconds=np.array([seconds]), p_len=np.array([p_len]))
out["mangio_hop128"] = vc.get_f0_crepe_computation(audio.copy(), 50, 1100, p_len, 128, "full")
out["mangio_hop160"] = vc.get_f0_crepe_computation(audio.copy(), 50, 1100, p_len, 160, "full")
out["official"] = vc.get_f0_official_crepe_computation(audio.copy(), 50, 1100, "full")
out["hybrid"] = vc.get_f0_hybrid_computation("hybrid[mangio-crepe+crepe]", "x.wav", audio.copy(), 50, 1100, p_len, 3, 160, 10.0)
coarse, f0 = vc.get_f0("x.wav", audio.copy(), p_len, 2, "crepe", 3, 128)
out["getf0_coarse"], out["getf0_f0"] = np.asarray(coarse), np.asarray(f0)
np.savez_compressed(os.path.join(HERE, "crepe_post_ref.npz"), **out)
for k, v in out.items():
    print(k, np.asarray(v).shape, float(np.nanmean(v)))
