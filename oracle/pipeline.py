"""TEST INFRASTRUCTURE (oracle): CPU restatement of VC.pipeline / VC.vc / VC.get_f0(rmvpe)
(reference src/vc_infer_pipeline.py:41-60, 262-370, 372-472, 474-653) on top of the oracle networks.
Pinned against the reference's own VC.pipeline through tests/golden/pipeline_*.npz (make_golden.py runs the
reference class with these same seeded networks).  Never imported by the product."""
import numpy as np
import torch
import torch.nn.functional as F
from scipy import signal

from . import hubert as ohub
from . import rmvpe as orm
from . import synth as osynth

bh, ah = signal.butter(N=5, Wn=48, btype="high", fs=16000)


class Geometry:
    """Config.x_* -> sample counts, as VC.__init__ (vc_infer_pipeline.py:64-80)."""

    def __init__(self, tgt_sr, x_pad, x_query, x_center, x_max):
        self.sr, self.window = 16000, 160
        self.x_pad = x_pad
        self.t_pad = self.sr * x_pad
        self.t_pad_tgt = tgt_sr * x_pad
        self.t_pad2 = self.t_pad * 2
        self.t_query = self.sr * x_query
        self.t_center = self.sr * x_center
        self.t_max = self.sr * x_max


def rms_frames(y, frame_length, hop_length):
    """librosa.feature.rms, librosa 0.9.1 defaults (center=True, reflect padding)."""
    pad = frame_length // 2
    yp = np.pad(y, (pad, pad), mode="reflect")
    n = 1 + (len(yp) - frame_length) // hop_length
    out = np.empty(n, dtype=y.dtype)
    for i in range(n):
        seg = yp[i * hop_length: i * hop_length + frame_length]
        out[i] = np.sqrt(np.mean(np.abs(seg) ** 2))
    return out[None, :]


def change_rms(data1, sr1, data2, sr2, rate):
    rms1 = torch.from_numpy(rms_frames(data1, sr1 // 2 * 2, sr1 // 2))
    rms2 = torch.from_numpy(rms_frames(data2, sr2 // 2 * 2, sr2 // 2))
    rms1 = F.interpolate(rms1.unsqueeze(0), size=data2.shape[0], mode="linear").squeeze()
    rms2 = F.interpolate(rms2.unsqueeze(0), size=data2.shape[0], mode="linear").squeeze()
    rms2 = torch.max(rms2, torch.zeros_like(rms2) + 1e-6)
    return data2 * (torch.pow(rms1, torch.tensor(1 - rate)) * torch.pow(rms2, torch.tensor(rate - 1))).numpy()


def cut_points(geo, audio):
    """Quietest-sample search (vc_infer_pipeline.py:514-528)."""
    audio_pad = np.pad(audio, (geo.window // 2, geo.window // 2), mode="reflect")
    opt_ts = []
    if audio_pad.shape[0] > geo.t_max:
        audio_sum = np.zeros_like(audio)
        for i in range(geo.window):
            audio_sum += audio_pad[i: i - geo.window]
        for t in range(geo.t_center, audio.shape[0], geo.t_center):
            seg = np.abs(audio_sum[t - geo.t_query: t + geo.t_query])
            opt_ts.append(t - geo.t_query + np.where(seg == seg.min())[0][0])
    return opt_ts


def vc_chunk(nets, geo, audio0, pitch, pitchf, sid, protect, noise):
    """VC.vc for a v2 / f0 model without faiss index (vc_infer_pipeline.py:372-472)."""
    wav = torch.from_numpy(np.ascontiguousarray(audio0)).float().view(1, -1)
    with torch.no_grad():
        feats = ohub.extract_features(nets["hubert_sd"], nets["hubert_cfg"], wav, 12)
    feats0 = feats.clone()
    feats = F.interpolate(feats.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)
    feats0 = F.interpolate(feats0.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)
    p_len = audio0.shape[0] // geo.window
    if feats.shape[1] < p_len:
        p_len = feats.shape[1]
        pitch, pitchf = pitch[:, :p_len], pitchf[:, :p_len]
    if protect < 0.5:
        pitchff = pitchf.clone()
        pitchff[pitchf > 0] = 1
        pitchff[pitchf < 1] = protect
        pitchff = pitchff.unsqueeze(-1)
        feats = feats * pitchff + feats0 * (1 - pitchff)
    with torch.no_grad():
        o, _ = osynth.synth_infer(nets["synth_sd"], nets["synth_cfg"], feats, pitch, pitchf, sid, noise[0], noise[1])
    return o[0, 0].numpy()


def chunk_noise(ci, T, inter, upp, seed):
    g = torch.Generator().manual_seed(seed * 1000 + ci)   # float32 draws whatever the default dtype (make_fp64_c1.py casts them)
    return (torch.randn(1, inter, T, generator=g, dtype=torch.float32).to(torch.get_default_dtype()),
            torch.randn(1, T * upp, generator=g, dtype=torch.float32).to(torch.get_default_dtype()))


def chunk_frames(n_samples):
    """Synthesizer frames for a chunk of n 16 kHz samples: 2 * HuBERT frames."""
    return 2 * ((n_samples - 400) // 320 + 1)


def vc_pipeline(nets, geo, audio, f0_up_key=0, rms_mix_rate=0.25, protect=0.33, tgt_sr=40000, noise_seed=7, sid=0,
                f0_method="rmvpe", crepe_hop=128, crepe_dither=None, f0_inject=None):
    """VC.pipeline with file_index='', if_f0=1, resample_sr=0 (vc_infer_pipeline.py:474-653) and f0_method 'rmvpe' (get_f0 :346-353)
    or 'mangio-crepe' (:296-301 -> get_f0_crepe_computation :96-137 on the padded track with p_len and crepe_hop_length; needs
    nets["crepe_sd"]; `crepe_dither`: the cents offsets replacing torchcrepe's random dither, None = none).
    `f0_inject`: an f0 track (Hz per 10 ms frame) that replaces the estimator's output in front of the key shift and the
    quantiser -- for tests that hold f0 fixed.  Returns (int16 audio, dict of intermediates)."""
    cfg = nets["synth_cfg"]
    upp = 1
    for u in cfg[12]:
        upp *= u
    audio = signal.filtfilt(bh, ah, audio)
    opt_ts = cut_points(geo, audio)
    audio_pad = np.pad(audio, (geo.t_pad, geo.t_pad), mode="reflect")
    p_len = audio_pad.shape[0] // geo.window
    hidden = bins = post = None
    if f0_inject is not None:
        f0 = np.asarray(f0_inject, dtype=np.float64).copy()
    elif f0_method == "rmvpe":
        f0, hidden = orm.infer_from_audio(nets["rmvpe_sd"], audio_pad, 0.03)
    elif f0_method == "mangio-crepe":
        from . import crepe as ocr
        f0, bins, post = ocr.mangio_crepe_f0(nets["crepe_sd"], audio_pad, p_len, crepe_hop, dither=crepe_dither)
    else:
        raise ValueError(f0_method)
    coarse, f0bak = orm.f0_to_coarse(f0, f0_up_key)
    pitch = torch.tensor(coarse[:p_len]).unsqueeze(0).long()
    pitchf = torch.tensor(f0bak[:p_len]).unsqueeze(0).float()
    sid_t = torch.tensor([sid]).long()
    out, s, t, ci = [], 0, None, 0
    for t in opt_ts:
        t = t // geo.window * geo.window
        seg = audio_pad[s: t + geo.t_pad2 + geo.window]
        noise = chunk_noise(ci, chunk_frames(len(seg)), cfg[2], upp, noise_seed)
        out.append(vc_chunk(nets, geo, seg, pitch[:, s // geo.window: (t + geo.t_pad2) // geo.window],
                            pitchf[:, s // geo.window: (t + geo.t_pad2) // geo.window], sid_t, protect, noise)[geo.t_pad_tgt: -geo.t_pad_tgt])
        s = t
        ci += 1
    seg = audio_pad[t:] if t is not None else audio_pad
    noise = chunk_noise(ci, chunk_frames(len(seg)), cfg[2], upp, noise_seed)
    out.append(vc_chunk(nets, geo, seg, pitch[:, t // geo.window:] if t is not None else pitch,
                        pitchf[:, t // geo.window:] if t is not None else pitchf, sid_t, protect, noise)[geo.t_pad_tgt: -geo.t_pad_tgt])
    audio_opt = np.concatenate(out)
    pre_rms = audio_opt.copy()
    if rms_mix_rate != 1:
        audio_opt = change_rms(audio, 16000, audio_opt, tgt_sr, rms_mix_rate)
    audio_max = np.abs(audio_opt).max() / 0.99
    max_int16 = 32768
    if audio_max > 1:
        max_int16 /= audio_max
    return (audio_opt * max_int16).astype(np.int16), dict(opt_ts=opt_ts, f0=f0bak, coarse=coarse, float_audio=audio_opt,
                                                           pre_rms=pre_rms, hidden=hidden, crepe_bins=bins, crepe_post=post,
                                                           audio_pad=audio_pad, p_len=p_len)
