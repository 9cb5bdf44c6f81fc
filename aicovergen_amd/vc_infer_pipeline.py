"""RVC voice-conversion pipeline on the gfx950 kernels behind the reference's `VC` class
(reference src/vc_infer_pipeline.py:63-653; `Pipeline` is the name upstream RVC uses and BASELINE.json quotes).

Kept exactly: chunk geometry from Config (x_pad, x_query, x_center, x_max), the 48 Hz zero-phase high-pass, the
quietest-sample cut search, reflect padding, whole-track f0, per-chunk vc(), RMS mixing, peak limiting and the
truncating int16 conversion, the in-place `times = [hubert, f0, synth]` accounting.
Moved to the device: HuBERT, RMVPE (incl. the salience decode and the coarse-pitch quantiser), the feature
upsample / protect blend and the whole synthesizer.  Host numpy keeps the O(N) pre/post steps.
"""
import functools
import os
import traceback
from time import time as ttime

import numpy as np
import torch
from . import _env
import torch.nn.functional as F
from scipy import signal

from . import ops
from . import dist as adist

BASE_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

bh, ah = signal.butter(N=5, Wn=48, btype="high", fs=16000)  # reference :22

input_audio_path2wav = {}

_NO_GROUP = "single"   # sentinel understood by dist.world(): this call must not communicate even if a process group exists


class _HostStream:
    """What the overlapped schedule needs of a torch.cuda.Stream when the kernels run synchronously on the host (the CPU emulator
    of tests/): every hand-over is already ordered."""

    def wait_stream(self, other): pass
    def wait_event(self, ev): pass
    def synchronize(self): pass
    def __enter__(self): return self
    def __exit__(self, *a): return False


class _HostEvent:
    def record(self, stream=None): pass


def _rms_frames(y, frame_length, hop_length):
    """librosa.feature.rms(y=y, frame_length=, hop_length=) of librosa 0.9.1: center=True with reflect padding,
    frame power mean, sqrt; returns (1, n_frames) float32 like librosa does for float32 input."""
    y = np.asarray(y)
    pad = frame_length // 2
    yp = np.pad(y, (pad, pad), mode="reflect")
    n_frames = 1 + (len(yp) - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n_frames)[:, None]
    power = np.mean(np.abs(yp[idx]) ** 2, axis=1)
    return np.sqrt(power)[None, :].astype(y.dtype if y.dtype in (np.float32, np.float64) else np.float32)


def change_rms(data1, sr1, data2, sr2, rate):
    """RMS-envelope mix of input (1) into output (2) (reference :41-60)."""
    rms1 = _rms_frames(data1, sr1 // 2 * 2, sr1 // 2)
    rms2 = _rms_frames(data2, sr2 // 2 * 2, sr2 // 2)
    rms1 = F.interpolate(torch.from_numpy(rms1).unsqueeze(0), size=data2.shape[0], mode="linear").squeeze()
    rms2 = F.interpolate(torch.from_numpy(rms2).unsqueeze(0), size=data2.shape[0], mode="linear").squeeze()
    rms2 = torch.max(rms2, torch.zeros_like(rms2) + 1e-6)
    data2 *= (torch.pow(rms1, torch.tensor(1 - rate)) * torch.pow(rms2, torch.tensor(rate - 1))).numpy()
    return data2


def _bracket_pipeline(fn):
    """While VC.pipeline runs, the f0 estimators may use the job's process group (VC._f0_group) -- and never outside it,
    whatever happens inside the call."""
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        self._in_pipeline = True
        try:
            return fn(self, *args, **kwargs)
        finally:
            self._in_pipeline = False
            self._group = None
    return wrapper


class VC(object):
    def __init__(self, tgt_sr, config):
        self.x_pad, self.x_query, self.x_center, self.x_max, self.is_half = (
            config.x_pad, config.x_query, config.x_center, config.x_max, config.is_half)
        self.sr = 16000      # HuBERT input rate
        self.window = 160    # samples per f0 frame
        self.t_pad = self.sr * self.x_pad
        self.t_pad_tgt = tgt_sr * self.x_pad
        self.t_pad2 = self.t_pad * 2
        self.t_query = self.sr * self.x_query
        self.t_center = self.sr * self.x_center
        self.t_max = self.sr * self.x_max
        self.device = config.device
        self.rmvpe_path = None   # resolved on first use (_default_rmvpe_path), or set by the caller

    def get_optimal_torch_device(self, index: int = 0) -> torch.device:
        if torch.cuda.is_available():
            return torch.device(f"cuda:{index % torch.cuda.device_count()}")
        return torch.device("cpu")

    # ---- f0 ---------------------------------------------------------------------------------------------------
    def get_f0_crepe_computation(self, x, f0_min, f0_max, p_len, hop_length=160, model="full", dither=None):
        """mangio-crepe (reference :96-137).  The CREPE network is built lazily from torchcrepe's bundled weights, or
        injected as `self.model_crepe[model]` (tests / benchmarks use seeded parameters)."""
        from . import crepe
        print("Initiating prediction with a crepe_hop_length of: " + str(hop_length))
        return crepe.mangio_crepe_f0(self._crepe(model), x, p_len, hop_length, dither=dither, group=self._f0_group(False))

    def _crepe(self, model):
        from . import crepe
        if not hasattr(self, "model_crepe"):
            self.model_crepe = {}
        if model not in self.model_crepe:
            self.model_crepe[model] = crepe.load_crepe(model, self.device)
        return self.model_crepe[model]

    def get_f0_official_crepe_computation(self, x, f0_min, f0_max, model="full", dither=None):
        """f0_method 'crepe' / 'crepe-tiny' (reference :139-165): torchcrepe.predict at the 10 ms hop with periodicity, 3-frame
        median of the periodicity, 3-frame mean of f0, unvoiced (periodicity < 0.1) frames zeroed."""
        from . import crepe
        return crepe.official_crepe_f0(self._crepe(model), x, self.window, f0_min, f0_max, dither=dither,
                                       group=self._f0_group(False))

    def get_f0_hybrid_computation(self, methods_str, input_audio_path, x, f0_min, f0_max, p_len, filter_radius,
                                  crepe_hop_length, time_step):
        """hybrid[m1+m2+...] (reference :175-260): nan-median over the listed estimators on the quantile-normalised signal."""
        methods = methods_str.split("hybrid")[1].replace("[", "").replace("]", "").split("+")
        print("Calculating f0 pitch estimations for methods: %s" % str(methods))
        x = x.astype(np.float32)
        x = x / np.quantile(np.abs(x), 0.999)
        stack = []
        for method in methods:
            if method in ("crepe", "crepe-tiny"):
                f0 = self.get_f0_official_crepe_computation(x, f0_min, f0_max, "tiny" if method.endswith("tiny") else "full")[1:]
            elif method in ("mangio-crepe", "mangio-crepe-tiny"):
                f0 = self.get_f0_crepe_computation(x, f0_min, f0_max, p_len, crepe_hop_length,
                                                   "tiny" if method.endswith("tiny") else "full")
            else:
                raise NotImplementedError("hybrid f0: method %r needs parselmouth / pyworld (supported inside hybrid[...]: "
                                          "crepe, crepe-tiny, mangio-crepe, mangio-crepe-tiny)" % method)
            stack.append(np.asarray(f0, dtype=np.float64))
        for fc in stack:
            print(len(fc))
        print("Calculating hybrid median f0 from the stack of: %s" % str(methods))
        if len(stack) == 1:
            return stack[0]
        return np.nanmedian(stack, axis=0)   # like the reference, estimators of different lengths raise here

    def _f0_group(self, explicit=True):
        """The process group the f0 estimators may communicate over.  Inside pipeline() the ranks of the job cut RMVPE's U-Net
        over time (rmvpe.E2E.features_sharded) and CREPE's frames (crepe.predict); a stand-alone get_f0 /
        get_f0_*_computation call (no pipeline in progress) NEVER communicates, whatever torch.distributed state the process
        has -- the other ranks would not join the collective.
        Returns None outside pipeline() or for a single-rank job.  Inside: the job's group; for the default group that is
        td.group.WORLD when `explicit` (RMVPE tests `group is not None`) and None otherwise (crepe hands it to dist.world())."""
        import torch.distributed as td
        if not getattr(self, "_in_pipeline", False) or not (td.is_available() and td.is_initialized()):
            return _NO_GROUP if not explicit else None
        g = getattr(self, "_group", None)
        if adist.single(td.get_world_size(g) if g is not None else td.get_world_size(), g):
            return _NO_GROUP if not explicit else None
        return g if (g is not None or not explicit) else td.group.WORLD

    _rmvpe_group = _f0_group

    @staticmethod
    def _default_rmvpe_path():
        """Where rmvpe.pt lives.  The reference reads <its checkout>/rvc_models/rmvpe.pt (src/vc_infer_pipeline.py:18,327: BASE_DIR is the
        parent of ITS src/); this module lives in another tree, so behind the shadow modules the file is looked up where main.py keeps its
        models: the `rvc_models_dir` global of the running main module (src/main.py:27), then rvc_models/ beside every sys.path entry
        (python src/main.py puts the reference's src/ there), then this repository's own rvc_models/."""
        import sys
        cands = []
        for name in ("__main__", "main"):
            d = getattr(sys.modules.get(name), "rvc_models_dir", None)
            if isinstance(d, str):
                cands.append(os.path.join(d, "rmvpe.pt"))
        for q in sys.path:
            if q:
                cands.append(os.path.join(os.path.dirname(os.path.abspath(q)), "rvc_models", "rmvpe.pt"))
        cands.append(os.path.join(BASE_DIR, "rvc_models", "rmvpe.pt"))
        for c in cands:
            if os.path.exists(c):
                return c
        return cands[-1]

    def _rmvpe(self):
        if not hasattr(self, "model_rmvpe"):
            from .rmvpe import RMVPE
            self.model_rmvpe = RMVPE(self.rmvpe_path or self._default_rmvpe_path(), is_half=self.is_half, device=self.device)
        return self.model_rmvpe

    def get_f0(self, input_audio_path, x, p_len, f0_up_key, f0_method, filter_radius, crepe_hop_length, inp_f0=None,
               _raw_f0=None):
        """-> (f0_coarse int64 (n,), f0 float64 (n,)) (reference :262-370).  `_raw_f0` (internal): the estimator's output
        when pipeline() has already run it on a side stream."""
        f0_min, f0_max = 50, 1100
        f0_mel_min = 1127 * np.log(1 + f0_min / 700)
        f0_mel_max = 1127 * np.log(1 + f0_max / 700)
        def host(a):   # the crepe branches start with host numpy arithmetic (quantile normalisation), like the reference
            return a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)

        if _raw_f0 is not None:
            f0 = _raw_f0
        elif f0_method == "rmvpe":
            f0 = self._rmvpe().infer_from_audio(x, thred=0.03, group=self._rmvpe_group())
        elif f0_method in ("mangio-crepe", "mangio-crepe-tiny"):
            f0 = self.get_f0_crepe_computation(host(x), f0_min, f0_max, p_len, crepe_hop_length,
                                               "tiny" if f0_method.endswith("tiny") else "full")
        elif f0_method in ("crepe", "crepe-tiny"):
            f0 = self.get_f0_official_crepe_computation(host(x), f0_min, f0_max, "tiny" if f0_method.endswith("tiny") else "full")
        elif "hybrid" in f0_method:
            f0 = self.get_f0_hybrid_computation(f0_method, input_audio_path, host(x), f0_min, f0_max, p_len, filter_radius,
                                                crepe_hop_length, self.window / self.sr * 1000)
        else:
            raise NotImplementedError(
                "f0_method %r needs parselmouth / pyworld, which are outside the MI355X hot path "
                "(supported: rmvpe, mangio-crepe[-tiny], crepe[-tiny], hybrid[...] of the crepe methods)" % f0_method)
        tf0 = self.sr // self.window
        f0 = np.asarray(self._estimated_f0(0, len(f0), np.asarray(f0, dtype=np.float64)), dtype=np.float64)
        factor = pow(2, f0_up_key / 12)
        if inp_f0 is not None:
            f0 = f0 * factor
            factor = 1.0
            delta_t = np.round((inp_f0[:, 0].max() - inp_f0[:, 0].min()) * tf0 + 1).astype("int16")
            replace_f0 = np.interp(list(range(delta_t)), inp_f0[:, 0] * 100, inp_f0[:, 1])
            shape = f0[self.x_pad * tf0: self.x_pad * tf0 + len(replace_f0)].shape[0]
            f0[self.x_pad * tf0: self.x_pad * tf0 + len(replace_f0)] = replace_f0[:shape]
        f0bak, coarse = self._f0_tail(torch.from_numpy(f0).to(self.device), factor)
        return coarse.cpu().numpy(), f0bak.cpu().numpy()

    def _estimated_f0(self, lo, hi, f0):
        """Seam behind the estimator: EVERY f0 estimate passes through here before the key shift and the coarse quantiser -- the whole
        track (numpy float64, lo = 0) from get_f0 under the serial / one-launch schedules, frame range [lo, hi) (device float64 tensor)
        as the progressive schedule publishes it.  Identity; tests/test_bench_sizes.py replaces it to hold f0 fixed (the reference's own
        track) under whichever schedule pipeline() takes."""
        return f0

    def _f0_tail(self, f0, factor):
        """Key shift + mel-scale coarse bins of an f0 track (device float64) -> (f0 * factor float64, coarse int64), reference :360-370.
        The one place both schedules quantise: get_f0 (whole track) and the progressive schedule's per-range callback."""
        f0_mel_min = 1127 * np.log(1 + 50 / 700)
        f0_mel_max = 1127 * np.log(1 + 1100 / 700)
        return ops.f0_coarse(f0, factor, f0_mel_min, f0_mel_max)

    # ---- one chunk ----------------------------------------------------------------------------------------------
    def vc(self, model, net_g, sid, audio0, pitch, pitchf, times, index, big_npy, index_rate, version, protect,
           noise=None, keep_on_device=False):
        """One padded chunk -> float32 waveform at tgt_sr (reference :372-472).  `noise` = (noise_z, noise_src)
        replaces the synthesizer's random draws (parity tests); `keep_on_device` returns the waveform as a device
        tensor instead of the reference's numpy array (used by pipeline(), which post-processes on the device)."""
        t0 = ttime()
        feats, feats0 = self._vc_features(model, audio0, index, big_npy, index_rate, version,
                                          protect < 0.5 and pitch is not None and pitchf is not None)
        if self._sync():
            torch.cuda.synchronize()
        t1 = ttime()
        o = self._vc_synth(net_g, sid, audio0.shape[0], feats, feats0, pitch, pitchf, protect, noise)
        if keep_on_device:
            if self._sync():
                torch.cuda.synchronize()
            audio1 = o[0, 0]
        else:
            audio1 = o[0, 0].data.cpu().float().numpy()
        t2 = ttime()
        times[0] += t1 - t0
        times[2] += t2 - t1
        return audio1

    def _vc_features(self, model, audio0, index, big_npy, index_rate, version, use_protect):
        """HuBERT features of one chunk (+ optional faiss index mix), reference :379-431.  Needs no pitch."""
        return self._vc_features_many(model, [audio0], index, big_npy, index_rate, version, use_protect)[0]

    def _vc_features_many(self, model, audios, index, big_npy, index_rate, version, use_protect):
        """_vc_features for several chunks -> [(feats, feats0)].  A model with `extract_features_many` (aicovergen_amd.hubert) runs
        the transformer's per-token layers once over all chunks; any other object with fairseq's `extract_features` is called
        chunk by chunk, as the reference does."""
        srcs = []
        for audio0 in audios:
            # audio0: host array (reference contract) or a slice of the track already resident on the device
            feats = audio0.float() if torch.is_tensor(audio0) else torch.from_numpy(np.ascontiguousarray(audio0)).float()
            if feats.dim() == 2:
                feats = feats.mean(-1)
            assert feats.dim() == 1, feats.dim()
            srcs.append(feats.view(1, -1).to(self.device))
        layer = 9 if version == "v1" else 12
        if len(srcs) > 1 and hasattr(model, "extract_features_many"):
            logits = model.extract_features_many(srcs, layer)
        else:
            logits = [model.extract_features(source=s, padding_mask=torch.zeros(s.shape, dtype=torch.bool), output_layer=layer)[0]
                      for s in srcs]
        return [self._vc_features_post(model, lg, index, big_npy, index_rate, version, use_protect) for lg in logits]

    def _vc_features_post(self, model, logits, index, big_npy, index_rate, version, use_protect):
        feats = model.final_proj(logits) if version == "v1" else logits
        feats0 = feats.clone() if use_protect else None
        if index is not None and hasattr(index, "mix_") and index_rate != 0:
            # device retrieval (aicovergen_amd.retrieval.FeatureIndex): faiss' search for the file's index type (IVF-Flat with
            # its nprobe, or flat) + inverse-square blend, all in HBM
            feats = index.mix_(feats[0].contiguous(), index_rate).unsqueeze(0)
        elif index is not None and big_npy is not None and index_rate != 0:
            npy = feats[0].cpu().numpy().astype("float32")
            score, ix = index.search(npy, k=8)
            weight = np.square(1 / score)
            weight /= weight.sum(axis=1, keepdims=True)
            npy = np.sum(big_npy[ix] * np.expand_dims(weight, axis=2), axis=1)
            feats = torch.from_numpy(npy.astype("float32")).unsqueeze(0).to(self.device) * index_rate + (1 - index_rate) * feats
        return feats, feats0

    def _vc_synth(self, net_g, sid, n_samples, feats, feats0, pitch, pitchf, protect, noise):
        """Features (+ pitch) of one chunk -> synthesizer output (1, 1, T) on the device, reference :433-466."""
        return self._vc_synth_back(net_g, self._vc_synth_front(net_g, sid, n_samples, feats, feats0, pitch, pitchf, protect, noise))

    def _vc_synth_back(self, net_g, st):
        """Vocoder half of a chunk's synthesis (state from _vc_synth_front)."""
        if "front" not in st:
            return st["o"]
        return net_g.infer_back(st["front"], st["pitchf"], st["ns"])[0]

    def _vc_synth_front(self, net_g, sid, n_samples, feats, feats0, pitch, pitchf, protect, noise):
        """Encoder half of a chunk's synthesis: nearest x2 upsample + protect blend, text encoder, prior sample, reverse flow.
        Synthesizers without the front / back split (any object with the reference's `infer`) run whole here."""
        p_len = n_samples // self.window
        if 2 * feats.shape[1] < p_len:
            p_len = 2 * feats.shape[1]
        if pitch is not None and pitchf is not None:
            pitch = pitch[:, :p_len]
            pitchf = pitchf[:, :p_len]
        use_protect = feats0 is not None
        # nearest x2 upsample + protect blend, written channel-major for the synthesizer (:433-452)
        phone_ct = ops.feats_prepare(feats[0], p_len, feats0[0] if use_protect else None,
                                     pitchf[0].float() if use_protect else None, protect)
        nz, ns = noise if noise is not None else (None, None)
        if hasattr(net_g, "infer_front"):
            return {"front": net_g.infer_front(phone_ct, pitch, sid, nz), "pitchf": pitchf, "ns": ns}
        lens = torch.tensor([p_len], device=self.device).long()
        if pitch is not None and pitchf is not None:
            return {"o": net_g.infer(None, lens, pitch, pitchf, sid, noise_z=nz, noise_src=ns, phone_ct=phone_ct)[0]}
        return {"o": net_g.infer(None, lens, sid, noise_z=nz, noise_src=ns, phone_ct=phone_ct)[0]}

    def _sync(self):
        return torch.cuda.is_available() and str(self.device).startswith("cuda")

    # ---- whole track ----------------------------------------------------------------------------------------------
    def plan(self, audio):
        """High-pass, cut search and padding (reference :513-534) -> (audio_hp, audio_pad, opt_ts, p_len); audio_hp / audio_pad are
        float64 tensors on self.device.  `audio`: the reference's host array, or a float tensor already resident on the device
        (the opt-in hand-over from the separation stage).
        The zero-phase Butterworth runs block-parallel on the device (ops.filtfilt_f64; AICG_FILTFILT=host keeps scipy's
        sequential recurrence on the host): it agrees with scipy.signal.filtfilt to ~5e-8 of the signal peak, which is the
        rounding-noise floor of this ill-conditioned direct-form filter itself (tests/test_dsp.py compares both against an
        extended-precision run)."""
        dev = self.device if ops._lib.backend() != "emu" else torch.device("cpu")
        if torch.is_tensor(audio):
            a = audio.detach().to(dev).double().view(-1)
        else:
            a = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float64)).to(dev)
        if _env.dev("AICG_FILTFILT", "device") == "host" or a.numel() <= 3 * max(len(ah), len(bh)):
            audio = torch.from_numpy(np.ascontiguousarray(signal.filtfilt(bh, ah, a.cpu().numpy()))).to(dev)
        else:
            audio = ops.filtfilt_f64(a, bh, ah)
        n = audio.numel()

        def reflect_pad(x, k):   # np.pad(x, (k, k), mode="reflect"): a re-indexing (numpy reflects repeatedly when k >= n)
            if k < n:
                return F.pad(x.view(1, 1, -1), (k, k), mode="reflect").view(-1)
            period = max(2 * (n - 1), 1)
            i = (torch.arange(-k, n + k, device=x.device) % period)
            return x[torch.where(i >= n, period - i, i)]

        opt_ts = []
        if n + self.window > self.t_max:
            # 160-tap box sum in the reference's summation order + first-minimum search, on the device (bit-exact)
            audio_sum = ops.box_sum_f64(reflect_pad(audio, self.window // 2), n, self.window)
            centers = list(range(self.t_center, n, self.t_center))
            starts = [t - self.t_query for t in centers]
            lens = [min(t + self.t_query, n) - (t - self.t_query) for t in centers]
            idx = ops.argmin_abs_f64(audio_sum, starts, lens).cpu().numpy()
            opt_ts = [int(s0 + i) for s0, i in zip(starts, idx)]
        audio_pad = reflect_pad(audio, self.t_pad)
        return audio, audio_pad, opt_ts, audio_pad.shape[0] // self.window

    def chunk_bounds(self, audio_pad, opt_ts):
        """[(start, end)] sample ranges of audio_pad handed to vc(), in order (reference :567-637)."""
        bounds = []
        s = 0
        t = None
        for t in opt_ts:
            t = t // self.window * self.window
            bounds.append((s, t + self.t_pad2 + self.window))
            s = t
        bounds.append((t if t is not None else 0, audio_pad.shape[0]))
        return bounds

    @_bracket_pipeline
    def pipeline(self, model, net_g, sid, audio, input_audio_path, times, f0_up_key, f0_method, file_index, index_rate,
                 if_f0, filter_radius, tgt_sr, resample_sr, rms_mix_rate, version, protect, crepe_hop_length, f0_file=None,
                 noise_fn=None, group=None, noise_seed=None):
        """Same contract as the reference (:474-653): float32 16 kHz mono in, int16 at tgt_sr out.
        `noise_fn(chunk_index, start, end) -> (noise_z, noise_src)` injects the synthesizer noise (tests); `noise_seed`
        instead draws it from a per-chunk seeded device generator, which makes the output independent of how chunks are
        distributed over ranks UP TO fp32 SUMMATION ORDER: the HuBERT transformer runs once over all of a rank's chunks
        (HubertModel.extract_features_many), so the GEMM tiles -- and the order of their fp32 sums -- depend on how many chunks a
        rank owns; outputs of different world sizes agree to ~1e-6 relative (tests/test_dist.py states the bound), ranks of one run
        agree bit for bit; `group` shards the chunk loop over the ranks of a torch.distributed process group."""
        if noise_fn is None and noise_seed is not None:
            inter, upp = net_g.inter_channels, net_g.upp

            def noise_fn(ci, s, e, _seed=int(noise_seed)):
                g = torch.Generator(device=self.device).manual_seed(_seed + ci)
                T = 2 * ((e - s - 400) // 320 + 1)
                return (torch.randn((1, inter, T), generator=g, device=self.device),
                        torch.randn(T * upp, generator=g, device=self.device))
        index = big_npy = None
        if file_index != "" and os.path.exists(file_index) and index_rate != 0:
            try:
                if _env.dev("AICG_GPU_KNN", "1") != "0":
                    from . import retrieval
                    index = retrieval.load_index(file_index, self.device)   # vectors in HBM, search + mix on the device
                else:
                    import faiss   # the reference's host path: faiss search per chunk
                    index = faiss.read_index(file_index)
                    big_npy = index.reconstruct_n(0, index.ntotal)
            except Exception:
                traceback.print_exc()
                index = big_npy = None
        self._group = group   # the crepe f0 methods shard their frames over it, RMVPE its U-Net
        tp0 = ttime()
        audio, audio_pad, opt_ts, p_len = self.plan(audio)
        t1 = ttime()
        inp_f0 = None
        if hasattr(f0_file, "name"):
            try:
                with open(f0_file.name, "r") as f:
                    lines = f.read().strip("\n").split("\n")
                inp_f0 = np.array([[float(i) for i in line.split(",")] for line in lines], dtype="float32")
            except Exception:
                traceback.print_exc()
        sid = torch.tensor(sid, device=self.device).unsqueeze(0).long()
        bounds = self.chunk_bounds(audio_pad, opt_ts)
        rank, world = adist.world(group)
        mine = [ci for ci in range(len(bounds)) if ci % world == rank]
        pitch, pitchf = None, None
        use_protect = protect < 0.5 and if_f0 == 1

        def run_f0(raw=None):
            pc, pcf = self.get_f0(input_audio_path, audio_pad, p_len, f0_up_key, f0_method, filter_radius,
                                  crepe_hop_length, inp_f0, _raw_f0=raw)
            return (torch.tensor(pc[:p_len], device=self.device).unsqueeze(0).long(),
                    torch.tensor(pcf[:p_len], device=self.device).unsqueeze(0).float())

        # The f0 estimate does not depend on the HuBERT features and RMVPE's recurrent part occupies a handful of CUs, so on
        # the GPU its kernels are queued on a side stream first and the feature extraction of every chunk runs on the main
        # stream on top of them; the synthesizer passes start once both are done.  Everything is enqueued from this thread
        # (a helper thread fights the launch loop for the GIL).  AICG_OVERLAP_F0=0 restores the reference's serial order
        # (f0, then per chunk: features, synthesis).
        on_gpu = self._sync()
        # (AICG_F0_SEGMENTS on the host: the CPU tests walk the same progressive schedule, with the streams and events as no-ops;
        #  "1" = the recurrence as one launch.)  One rank on a GPU takes the progressive schedule too since the f0 chain became the
        #  longer branch of the phase (r4: HuBERT 74.9 ms, f0 79.4): the chunk loop then starts when HuBERT is done, on the middle chunks,
        #  while the recurrence finishes the track's ends -- 718.7 -> 713.9 ms per 240 s track at 8 segments (4: 715.5, 16: 715.6).
        env_seg = int(_env.dev("AICG_F0_SEGMENTS", "0"))
        nseg = env_seg or (16 if world > 1 else (8 if on_gpu else 0))
        overlap = (if_f0 == 1 and f0_method == "rmvpe" and (on_gpu or nseg > 1)
                   and _env.dev("AICG_OVERLAP_F0", "1") != "0")
        feats_of = {}
        f0_wait = 0.0
        gru_seg = None        # progressive f0: the recurrence's handle (its exchange-timeout word is polled between chunks)
        f0_bad = False
        marks = []            # progressive f0: (first frame, end frame, event) -- the frame range whose pitch exists once the event has fired
        progressive = False
        if overlap:
            main = torch.cuda.current_stream(self.device) if on_gpu else _HostStream()
            # one side stream per VC object: the caching allocator keeps a block pool per stream, a fresh stream per call
            # would strand RMVPE's working set (a few GB for a 4-minute track) in up to 32 pools
            side = getattr(self, "_f0_stream", None) if on_gpu else _HostStream()
            if side is None:
                # high priority: the f0 branch is a short U-Net followed by a 70 ms recurrence on four CUs; dispatched first it
                # leaves the chip to HuBERT while the GRU runs, dispatched behind HuBERT's launches it finishes 40 ms later
                # (AICG_F0_PRIORITY=0: default priority)
                prio = -1 if _env.dev("AICG_F0_PRIORITY", "1") != "0" else 0
                side = self._f0_stream = torch.cuda.Stream(device=self.device, priority=prio)
            # one upload of the padded track: a pageable host->device copy on the default stream waits for the whole device,
            # side stream included, so the chunk loop below must not issue any
            pad_dev = audio_pad.float()
            # Progressive f0 (see below): the device-resident pitch tracks are created and filled on `main` BEFORE the side stream forks
            # from it -- on_f0 writes slices of them on `side`, and a fill that could run after such a write would erase it
            progressive = nseg > 1 and inp_f0 is None
            if progressive:
                pitch = torch.ones((1, p_len), dtype=torch.long, device=self.device)
                pitchf = torch.zeros((1, p_len), dtype=torch.float32, device=self.device)
            side.wait_stream(main)
            tf0 = ttime()
            # Progressive f0 (multi-GPU: every rank runs the whole-track BiGRU, N x 32 ms for N x 240 s): the recurrence is queued in
            # segments, and the pitch of a frame range exists as soon as BOTH directions have passed it -- the middle of the track at
            # half the recurrence time, its ends last.  The chunk loop below then takes this rank's chunks middle-out and waits per
            # chunk (events), not for the whole track.  Same values as the one-launch form (GruSegments); the host never sees f0.
            # NOTE: this schedule never calls get_f0 (an override of VC.get_f0 is bypassed; the estimate passes _estimated_f0 and the
            # shared _f0_tail instead); an f0 curve file (inp_f0) keeps the one-launch schedule, whose get_f0 splices it.
            if progressive:
                cover = [p_len, 0]

                def on_f0(lo, hi, f0, _factor=pow(2, f0_up_key / 12)):
                    f0bak, coarse = self._f0_tail(self._estimated_f0(lo, hi, f0), _factor)
                    hi2 = min(hi, p_len)
                    if lo < hi2:
                        pitch[0, lo:hi2] = coarse[: hi2 - lo]
                        pitchf[0, lo:hi2] = f0bak[: hi2 - lo].float()
                    cover[0], cover[1] = min(cover[0], lo), max(cover[1], hi2)
                    ev = torch.cuda.Event() if on_gpu else _HostEvent()
                    ev.record(side)
                    marks.append((cover[0], cover[1], ev))

                with (torch.cuda.stream(side) if on_gpu else side):
                    self._rmvpe().infer_progressive(pad_dev, 0.03, nseg, on_f0, group=self._rmvpe_group())
                gru_seg = self._rmvpe().last_segments
            else:
                with (torch.cuda.stream(side) if on_gpu else side):
                    f0_dev = self._rmvpe().infer_from_audio_device(pad_dev, thred=0.03, group=self._rmvpe_group())
            many = self._vc_features_many(model, [pad_dev[bounds[ci][0]:bounds[ci][1]] for ci in mine], index, big_npy, index_rate,
                                          version, use_protect)
            feats_of = dict(zip(mine, many))

            def settle():
                """The host meets the device: every chunk's features exist.  -> (time, f0 already known to be invalid)"""
                main.synchronize()
                # an early exchange timeout: do not synthesise a whole track from invalid pitch
                return ttime(), bool(progressive and gru_seg.timed_out())

            # Progressive: the host first queues the first chunk's encoder half (below) and settles THEN -- its ~140 short launches
            # take the host 6 ms to issue, which now pass under the tail of HuBERT instead of in front of the first vocoder.
            settle_later = progressive
            if not settle_later:
                tf1, f0_bad = settle()
            if not progressive:
                side.synchronize()
                main.wait_stream(side)
                f0_host = f0_dev.cpu().numpy()
                if ops.gru_timed_out():  # multi-workgroup GRU starved of its partners under the HuBERT load: single-workgroup rerun
                    f0_host = self._rmvpe().infer_from_audio_device(pad_dev, thred=0.03, two_workgroups=False).cpu().numpy()
                pitch, pitchf = run_f0(f0_host)
                del f0_dev
            if not settle_later:
                t2 = ttime()
                f0_wait = t2 - tf1
                times[0] += tf1 - tf0
                times[1] += t2 - tf0  # the f0 branch's own wall time (it overlaps times[0])
        else:
            settle_later = False
            if if_f0 == 1:
                pitch, pitchf = run_f0()
            if self._sync():
                torch.cuda.synchronize()
            t2 = ttime()
            times[1] += t2 - t1
        pieces = {}

        def frames_of(ci):
            s, e = bounds[ci]
            return s // self.window, (p_len if ci == len(bounds) - 1 else (e - self.window) // self.window)

        def mark_of(ci):      # index of the first mark whose range covers the chunk's frames
            fa, fb = frames_of(ci)
            for m, (lo, hi, _) in enumerate(marks):
                if lo <= fa and min(fb, p_len) <= hi:
                    return m
            return len(marks) - 1

        order = sorted(mine, key=lambda ci: (mark_of(ci), ci)) if progressive else list(mine)

        def chunk_pitch(ci):
            s, e = bounds[ci]
            if if_f0 != 1:
                return None, None
            pe = None if ci == len(bounds) - 1 else (e - self.window) // self.window
            return pitch[:, s // self.window: pe], pitchf[:, s // self.window: pe]

        # Overlapped schedule: the encoder half of chunk i + 1 (text encoder + flow: ~140 short launches that leave most CUs idle
        # and take the host about as long to queue -- 40 us each -- as the GPU to run) is queued on a second stream underneath the
        # vocoder of chunk i.  The HOST queues the vocoder of chunk i first and the encoder half of chunk i + 1 behind it: the other way
        # round the main stream stood idle for the 6 ms the host needs to issue the short launches (kernel trace of round 5,
        # profiles/r05_timeline.json: 6.5 + 6.1 ms at two chunk boundaries, the first vocoder behind TWO encoder halves).  What that
        # buys is small -- 2 ms per 240 s track: short launches running beside the vocoder cost it nearly what they take alone
        # (one stream: 674.7 ms, two: 667) -- and a stream priority changes nothing.  AICG_OVERLAP_SYNTH=0: one stream.
        two_streams = overlap and on_gpu and hasattr(net_g, "infer_front") and _env.dev("AICG_OVERLAP_SYNTH", "1") != "0"
        fronts = {}
        if two_streams:
            main = torch.cuda.current_stream(self.device)
            enc = getattr(self, "_enc_stream", None)
            if enc is None:
                enc = self._enc_stream = torch.cuda.Stream(device=self.device)
            enc.wait_stream(main)          # features, pitch tracks, sid: everything the fronts read exists on `main` by now

            def queue_front(ci):
                s, e = bounds[ci]
                pc, pcf = chunk_pitch(ci)
                feats, feats0 = feats_of.pop(ci)
                if progressive:
                    enc.wait_event(marks[mark_of(ci)][2])      # this chunk's pitch frames exist
                with torch.cuda.stream(enc):
                    noise = noise_fn(ci, s, e) if noise_fn is not None else None
                    st = self._vc_synth_front(net_g, sid, e - s, feats, feats0, pc, pcf, protect, noise)
                    ev = torch.cuda.Event()
                    ev.record(enc)
                # the inputs stay referenced until the chunk's synchronize below: nothing is recycled under the other stream
                fronts[ci] = (st, ev, (feats, feats0, pc, pcf, noise))

            if order:
                queue_front(order[0])
        if settle_later:
            tf1, f0_bad = settle()
            t2 = tf1
            times[0] += tf1 - tf0
            times[1] += t2 - tf0

        def drain():          # the chunk's work is done (progressive: without waiting for the f0 stream's remaining segments)
            if not on_gpu:
                return
            if progressive:
                torch.cuda.current_stream(self.device).synchronize()
            else:
                torch.cuda.synchronize()

        for k, ci in enumerate(order):
            if f0_bad:
                break
            s, e = bounds[ci]
            pc, pcf = chunk_pitch(ci)
            if two_streams:
                ts0 = ttime()
                st, ev, keep = fronts.pop(ci)
                main.wait_event(ev)
                out = self._vc_synth_back(net_g, st)[0, 0]
                if k + 1 < len(order):
                    queue_front(order[k + 1])
                drain()
                del st, keep
                times[2] += ttime() - ts0
            elif overlap:
                noise = noise_fn(ci, s, e) if noise_fn is not None else None
                ts0 = ttime()
                feats, feats0 = feats_of.pop(ci)
                if progressive and on_gpu:
                    torch.cuda.current_stream(self.device).wait_event(marks[mark_of(ci)][2])
                out = self._vc_synth(net_g, sid, e - s, feats, feats0, pc, pcf, protect, noise)[0, 0]
                drain()
                times[2] += ttime() - ts0
            else:
                noise = noise_fn(ci, s, e) if noise_fn is not None else None
                out = self.vc(model, net_g, sid, audio_pad[s:e], pc, pcf, times, index, big_npy, index_rate, version, protect,
                              noise=noise, keep_on_device=True)
            pieces[ci] = out[self.t_pad_tgt: -self.t_pad_tgt]
            if progressive and gru_seg.timed_out():    # polled per chunk (4 bytes): at most one chunk is synthesised from bad pitch
                f0_bad = True
        if progressive:
            if on_gpu:
                torch.cuda.synchronize()
            fronts.clear()
            if ops.gru_timed_out() or f0_bad:
                # the multi-workgroup recurrence starved of its partners (a busy or shared GPU): this rank's f0 is invalid.  Recompute
                # it locally on the single-workgroup kernel (no collective: the other ranks may be fine) and redo this rank's chunks
                import warnings
                warnings.warn("aicovergen_amd: the multi-workgroup BiGRU timed out waiting for its partner workgroups (busy or shared GPU); "
                              "f0 and this rank's chunks are recomputed on the single-workgroup kernel -- this call costs about twice "
                              "its usual time", RuntimeWarning)
                f0_host = self._rmvpe().infer_from_audio_device(audio_pad.float(), thred=0.03, two_workgroups=False).cpu().numpy()
                pitch, pitchf = run_f0(f0_host)
                many = self._vc_features_many(model, [audio_pad.float()[bounds[ci][0]:bounds[ci][1]] for ci in mine], index, big_npy,
                                              index_rate, version, use_protect)
                for ci, (feats, feats0) in zip(mine, many):
                    s, e = bounds[ci]
                    pc, pcf = chunk_pitch(ci)
                    noise = noise_fn(ci, s, e) if noise_fn is not None else None
                    out = self._vc_synth(net_g, sid, e - s, feats, feats0, pc, pcf, protect, noise)[0, 0]
                    pieces[ci] = out[self.t_pad_tgt: -self.t_pad_tgt]
                if on_gpu:
                    torch.cuda.synchronize()
        tc1 = ttime()
        pieces = adist.gather_pieces(pieces, len(bounds), self.device, group)
        tj1 = ttime()
        audio_opt = torch.cat([pieces[i] for i in range(len(bounds))]).contiguous()
        if resample_sr >= 16000 and tgt_sr != resample_sr:
            # optional output resampling (rvc_infer passes resample_sr=0): host fallback, not on the hot path
            a = audio_opt.cpu().numpy()
            if rms_mix_rate != 1:
                a = change_rms(audio.cpu().numpy(), 16000, a, tgt_sr, rms_mix_rate)
            from scipy.signal import resample_poly
            g = np.gcd(int(tgt_sr), int(resample_sr))
            audio_opt = torch.from_numpy(resample_poly(a, resample_sr // g, tgt_sr // g).astype(np.float32)).to(self.device)
        elif rms_mix_rate != 1:
            # change_rms on the device: frame RMS envelopes (1 s frames, 0.5 s hop), linear interpolation, power mix
            rms1 = ops.frame_rms(audio, 16000 // 2 * 2, 16000 // 2)
            rms2 = ops.frame_rms(audio_opt, tgt_sr // 2 * 2, tgt_sr // 2)
            ops.rms_mix_(audio_opt, rms1, rms2, rms_mix_rate)
        audio_max = float(ops.absmax(audio_opt).item()) / 0.99
        max_int16 = 32768
        if audio_max > 1:
            max_int16 /= audio_max
        audio_opt = ops.to_int16(audio_opt, max_int16).cpu().numpy()
        # wall-clock split of this call (host pre-processing, f0, chunk loop, join + host post-processing)
        # (overlapped schedule: f0_s = features of every chunk with the f0 branch underneath, f0_wait_s of it spent waiting for f0)
        self.last_profile = {"plan_s": t1 - tp0, "f0_s": t2 - t1, "chunks_s": tc1 - t2, "post_s": ttime() - tc1,
                             "f0_wait_s": f0_wait, "overlap_f0": float(bool(overlap)), "join_s": tj1 - tc1,
                             "f0_progressive": float(bool(progressive))}
        return audio_opt


Pipeline = VC  # the name BASELINE.json / upstream RVC use for this class
