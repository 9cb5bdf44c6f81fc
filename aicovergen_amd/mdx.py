"""MDX-Net separation on the gfx950 kernels behind the reference's src/mdx.py surface: MDXModel (stft / istft),
MDX (segment / pad_wave / process / process_wave) and run_mdx(model_params, output_dir, model_path, filename, ...).

What changed underneath (and nothing above):
  * STFT / U-Net / iSTFT all run on the device in one (B, 4, T, F) layout; the reference's four host<->device
    copies per window (mdx.py:77,193-195) and its two Python threads sharing one ORT session (:216-226) are
    replaced by batching windows;
  * the windows of every half (and of the -x pass of `denoise`) are independent work items, which is what
    aicovergen_amd.dist shards across the GPUs of a node.
"""
import gc
import hashlib
import os

import numpy as np
import torch
from . import _env

from . import audio_io, ops
from .mdx_net import ConvTDFNet

stem_naming = {'Vocals': 'Instrumental', 'Other': 'Instruments', 'Instrumental': 'Vocals', 'Drums': 'Drumless',
               'Bass': 'Bassless'}


class MDXModel:
    """Same attributes as the reference (src/mdx.py:19-35); stft/istft keep its tensor layouts (:37-54)."""

    def __init__(self, device, dim_f, dim_t, n_fft, hop=1024, stem_name=None, compensation=1.000):
        self.dim_f, self.dim_t, self.dim_c = dim_f, dim_t, 4
        self.n_fft, self.hop = n_fft, hop
        self.stem_name, self.compensation = stem_name, compensation
        self.n_bins = self.n_fft // 2 + 1
        self.chunk_size = hop * (self.dim_t - 1)
        self.device = torch.device(device)
        self.window = torch.hann_window(window_length=self.n_fft, periodic=True).to(device)
        self.freq_pad = torch.zeros([1, self.dim_c, self.n_bins - self.dim_f, self.dim_t]).to(device)

    def stft(self, x):
        """(B, 2, chunk_size) -> (B, 4, dim_f, dim_t), channels (L.re, L.im, R.re, R.im)."""
        x = x.reshape([-1, self.chunk_size]).to(self.device).float()
        y = ops.stft(x, self.n_fft, self.hop, self.dim_f, window=self.window)  # (2B, 2, dim_f, dim_t)
        return y.reshape([-1, 4, self.dim_f, self.dim_t])

    def stft_tf(self, x):
        """Internal fast layout: (B, 2, chunk_size) -> (B, 4, dim_t, dim_f) with frequency contiguous."""
        x = x.reshape([-1, self.chunk_size])
        y = ops.stft(x, self.n_fft, self.hop, self.dim_f, frame_major=True, window=self.window)
        return y.reshape([-1, 4, self.dim_t, self.dim_f])

    def istft(self, x, freq_pad=None):
        """(B, 4, dim_f, dim_t) -> (B, 2, chunk_size).  A user-supplied freq_pad is concatenated like the
        reference does; the default (zeros) is handled inside the kernel without materialising it."""
        x = x.to(self.device).float()
        if freq_pad is not None:
            x = torch.cat([x, freq_pad.to(self.device)], -2)
        b = x.shape[0]
        y = ops.istft(x.reshape(b * 2, 2, x.shape[2], self.dim_t), self.n_fft, self.hop, self.chunk_size, window=self.window)
        return y.reshape([-1, 2, self.chunk_size])

    def istft_tf(self, x):
        b = x.shape[0]
        y = ops.istft(x.reshape(b * 2, 2, self.dim_t, x.shape[3]), self.n_fft, self.hop, self.chunk_size, frame_major=True,
                      window=self.window)
        return y.reshape([-1, 2, self.chunk_size])


def load_network_state(model_path):
    """Parameters of the TFC-TDF U-Net.  Accepts a torch checkpoint (state_dict, optionally under "state_dict");
    `.onnx` graphs need aicovergen_amd.onnx_weights (initializer reader)."""
    if str(model_path).lower().endswith(".onnx"):
        from .onnx_weights import load_onnx_state_dict
        return load_onnx_state_dict(model_path)
    ckpt = torch.load(model_path, map_location="cpu")
    if isinstance(ckpt, dict) and "state_dict" in ckpt:
        ckpt = ckpt["state_dict"]
    return ckpt


class MDX:
    DEFAULT_SR = 44100
    DEFAULT_CHUNK_SIZE = 0 * DEFAULT_SR
    DEFAULT_MARGIN_SIZE = 1 * DEFAULT_SR
    DEFAULT_PROCESSOR = 0
    WINDOW_BATCH = int(_env.dev("AICG_MDX_BATCH", "8"))  # windows per network launch

    def __init__(self, model_path, params: MDXModel, processor=DEFAULT_PROCESSOR, state_dict=None):
        self.device = params.device if state_dict is not None else (
            torch.device(f'cuda:{processor}') if processor >= 0 else torch.device('cpu'))
        if ops._lib.backend() == "emu":      # tests' host emulator: the reference's unconditional cuda:{processor} has no device to name
            self.device = torch.device("cpu")
        self.provider = ['HIPKernels']
        self.model = params
        sd = state_dict if state_dict is not None else load_network_state(model_path)
        self.net = ConvTDFNet(sd, self.device)
        assert self.net.cfg["dim_f"] == params.dim_f, "network frequency size does not match model_data.json"
        # same contract as the reference's lambda (mdx.py:77): torch spec in, numpy estimate out
        self.process = lambda spec: self.net(spec).cpu().numpy()
        self.prog = None

    _HASH_TAIL = 10000 * 1024

    @staticmethod
    def get_hash(model_path):
        """Key into model_data.json: md5 of the file's last 10 000 KiB, or of the whole file when it is shorter than that
        (the reference seeks from the end and falls back on the OSError, mdx.py:81-90)."""
        size = os.path.getsize(model_path)
        with open(model_path, "rb") as fh:
            if size > MDX._HASH_TAIL:
                fh.seek(size - MDX._HASH_TAIL)
            return hashlib.md5(fh.read()).hexdigest()

    @staticmethod
    def segment(wave, combine=True, chunk_size=DEFAULT_CHUNK_SIZE, margin_size=DEFAULT_MARGIN_SIZE):
        """Split a (2, N) wave into chunks overlapping by `margin_size`, or join such chunks (mdx.py:92-141)."""
        if combine:
            parts = []
            for i, seg in enumerate(wave):
                lo = 0 if i == 0 else margin_size
                hi = None if (i == len(wave) - 1 or margin_size == 0) else -margin_size
                parts.append(seg[:, lo:hi])
            return parts[0] if len(parts) == 1 else np.concatenate(parts, axis=-1)
        n = wave.shape[-1]
        if chunk_size <= 0 or chunk_size > n:
            chunk_size = n
        margin_size = min(margin_size, chunk_size)
        pieces = []
        for i, skip in enumerate(range(0, n, chunk_size)):
            lo = skip - (0 if i == 0 else margin_size)
            hi = min(skip + chunk_size + margin_size, n)
            pieces.append(wave[:, lo:hi].copy())
            if hi == n:
                break
        return pieces

    def _geometry(self, n_sample):
        trim = self.model.n_fft // 2
        gen = self.model.chunk_size - 2 * trim
        pad = gen - n_sample % gen  # a full extra window when n_sample % gen == 0, like the reference
        return trim, gen, pad

    def pad_wave(self, wave):
        """(2, n) -> (windows (k, 2, chunk_size) float32 on device, pad, trim) (mdx.py:143-171)."""
        n_sample = wave.shape[1]
        trim, gen, pad = self._geometry(n_sample)
        w = torch.as_tensor(np.ascontiguousarray(wave), dtype=torch.float32).to(self.device)
        wp = torch.nn.functional.pad(w, (trim, pad + trim))
        k = (n_sample + pad) // gen
        mix = wp.unfold(1, self.model.chunk_size, gen)[:, :k].permute(1, 0, 2).contiguous()  # window re-indexing
        return mix, pad, trim

    def _run_windows(self, mix_waves, trim):
        """(k, 2, chunk) device windows -> (2, k * gen) device signal: stft -> U-Net -> istft -> trim, batched."""
        outs = []
        for s in range(0, mix_waves.shape[0], self.WINDOW_BATCH):
            mw = mix_waves[s:s + self.WINDOW_BATCH]
            spec = self.model.stft_tf(mw)
            est = self.net.forward_tf(spec)
            wav = self.model.istft_tf(est)
            outs.append(wav[:, :, trim:-trim].transpose(0, 1).reshape(2, -1))
            if self.prog is not None:
                self.prog.update(mw.shape[0])
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=-1)

    def _process_wave(self, mix_waves, trim, pad, q, _id):
        """Kept for API compatibility with the reference's thread worker (mdx.py:173-199)."""
        with torch.no_grad():
            sig = self._run_windows(mix_waves, trim)[:, :-pad].cpu().numpy()
        q.put({_id: sig})
        return sig

    def process_wave(self, wave, mt_threads=1):
        """(2, N) numpy -> (2, N) numpy estimate of the primary stem (mdx.py:201-235).  `mt_threads` keeps its
        meaning as the number of overlapping segments the song is cut into; they are processed back to back on the
        device instead of on host threads."""
        try:
            from tqdm import tqdm
            self.prog = tqdm(total=0)
        except Exception:
            self.prog = None
        chunk = wave.shape[-1] // mt_threads
        waves = self.segment(wave, False, chunk)
        processed = []
        with torch.no_grad():
            for batch in waves:
                mix_waves, pad, trim = self.pad_wave(batch)
                if self.prog is not None:
                    self.prog.total = len(mix_waves) * mt_threads
                processed.append(self._run_windows(mix_waves, trim)[:, :-pad].cpu().numpy())
        if self.prog is not None:
            self.prog.close()
        assert len(processed) == len(waves), 'Incomplete processed batches, please reduce batch size!'
        return self.segment(processed, True, chunk)

    def separate(self, wave, denoise, mt_threads=2, shard=None):
        """run_mdx's numeric core on a device-resident normalised wave (2, N) torch tensor:
        0.5 * (-f(-x) + f(x)) when denoise else f(x) (mdx.py:261-265), with f = process_wave.
        Because padding, windowing and trimming are linear re-indexings, the windows of -x are the negated windows
        of x: both passes share one window list.  `shard` = (rank, world) processes a contiguous slice of the window
        list and returns (partial result, meta) for aicovergen_amd.dist to join."""
        n = wave.shape[-1]
        chunk = n // mt_threads
        cs = chunk if (0 < chunk <= n) else n
        margin = min(self.DEFAULT_MARGIN_SIZE, cs)
        bounds = []
        for i, skip in enumerate(range(0, n, cs)):
            lo = skip - (0 if i == 0 else margin)
            hi = min(skip + cs + margin, n)
            bounds.append((lo, hi))
            if hi == n:
                break
        trim = self.model.n_fft // 2
        gen = self.model.chunk_size - 2 * trim
        jobs = []  # (segment index, window index within segment)
        seg_info = []
        for si, (lo, hi) in enumerate(bounds):
            ns = hi - lo
            pad = gen - ns % gen
            k = (ns + pad) // gen
            seg_info.append((lo, hi, pad, k))
            jobs += [(si, wi) for wi in range(k)]
        rank, world = shard if shard is not None else (0, 1)
        per = (len(jobs) + world - 1) // world
        mine = jobs[rank * per:(rank + 1) * per]
        out_windows = torch.zeros((len(mine), 2, gen), dtype=torch.float32, device=wave.device)
        with torch.no_grad():
            for s in range(0, len(mine), self.WINDOW_BATCH):
                batch = mine[s:s + self.WINDOW_BATCH]
                wins = []
                for si, wi in batch:
                    lo, hi, pad, k = seg_info[si]
                    a = lo + wi * gen - trim            # window start in song coordinates (may be < lo)
                    idx0, idx1 = max(a, lo), min(a + self.model.chunk_size, hi)
                    w = torch.zeros((2, self.model.chunk_size), dtype=torch.float32, device=wave.device)
                    if idx1 > idx0:
                        w[:, idx0 - a: idx1 - a] = wave[:, idx0:idx1]
                    wins.append(w)
                mw = torch.stack(wins)
                if denoise:
                    mw = torch.cat([mw, -mw], 0)
                est = self.model.istft_tf(self.net.forward_tf(self.model.stft_tf(mw)))[:, :, trim:-trim]
                if denoise:
                    b = len(batch)
                    est = ops.axpbypcz(est[:b].contiguous(), 0.5, est[b:].contiguous(), -0.5)  # 0.5 * (f(x) - f(-x))
                out_windows[s:s + len(batch)] = est
        return out_windows, dict(jobs=jobs, per=per, seg_info=seg_info, gen=gen, margin=margin, n=n)

    @staticmethod
    def join_windows(all_windows, meta):
        """(total windows, 2, gen) in job order -> (2, N): per segment concat + drop `pad`, then margin join."""
        seg_out = []
        pos = 0
        for (lo, hi, pad, k) in meta["seg_info"]:
            sig = all_windows[pos:pos + k].transpose(0, 1).reshape(2, -1)[:, :-pad]
            pos += k
            seg_out.append(sig)
        parts = []
        m = meta["margin"]
        for i, seg in enumerate(seg_out):
            a = 0 if i == 0 else m
            b = seg.shape[1] if (i == len(seg_out) - 1 or m == 0) else seg.shape[1] - m
            parts.append(seg[:, a:b])
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)


def _stem_path(output_dir, filename, stem):
    base = os.path.splitext(os.path.basename(filename))[0]
    return os.path.join(output_dir, "%s_%s.wav" % (base, stem))


def run_mdx(model_params, output_dir, model_path, filename, exclude_main=False, exclude_inversion=False, suffix=None,
            invert_suffix=None, denoise=False, keep_orig=True, m_threads=2):
    """File in -> stem files out, with the arguments, file names and return value of reference src/mdx.py:238-287:
    model hash -> `model_params` entry -> MDXModel; stereo 44.1 kHz load; peak-normalise; separate (optionally as the odd part
    0.5 * (f(x) - f(-x))); undo the normalisation; write `<name>_<stem>.wav` and the inverted stem
    `-out * compensation + normalised input` (the reference adds the NORMALISED wave, mdx.py:280); PCM-16 like soundfile."""
    on_gpu = torch.cuda.is_available()
    device = torch.device("cuda:0" if on_gpu else "cpu")
    # the reference halves the thread (= segment) count on cards under 8 GB; it queries the device unconditionally
    vram_gb = torch.cuda.get_device_properties(device).total_memory / 1024 ** 3
    m_threads = 2 if vram_gb >= 8 else 1

    entry = model_params.get(MDX.get_hash(model_path))
    model = MDXModel(device, dim_f=entry["mdx_dim_f_set"], dim_t=2 ** entry["mdx_dim_t_set"], n_fft=entry["mdx_n_fft_scale_set"],
                     stem_name=entry["primary_stem"], compensation=entry["compensate"])
    session = MDX(model_path, model)

    wave, sr = audio_io.load_wav(filename, 44100, mono=False)
    if wave.shape[0] == 1:
        wave = np.concatenate([wave, wave], 0)
    peak = max(np.max(wave), abs(np.min(wave)))
    wave /= peak
    separated = run_mdx_arrays(session, wave, denoise, m_threads) * peak

    primary = suffix if suffix is not None else model.stem_name
    main_filepath = invert_filepath = None
    if not exclude_main:
        main_filepath = _stem_path(output_dir, filename, primary)
        audio_io.write_wav_pcm16(main_filepath, separated.T, sr)
    if not exclude_inversion:
        other = invert_suffix if invert_suffix is not None else stem_naming.get(primary)
        invert_filepath = _stem_path(output_dir, filename, other if other is not None else primary + "_diff")
        audio_io.write_wav_pcm16(invert_filepath, wave.T - separated.T * model.compensation, sr)
    if not keep_orig:
        os.remove(filename)
    del session, separated, wave
    gc.collect()
    return main_filepath, invert_filepath


def run_mdx_arrays(mdx_sess, wave, denoise, m_threads=2, group=None):
    """The array-level core of run_mdx: normalised (2, N) numpy in -> separated (2, N) numpy out.  With a
    torch.distributed `group` (or an initialised default group) the window list is sharded across ranks and joined
    by an all-gather (aicovergen_amd.dist)."""
    from . import dist as adist
    w = torch.as_tensor(np.ascontiguousarray(wave), dtype=torch.float32).to(mdx_sess.device)
    out = adist.mdx_separate(mdx_sess, w, denoise, m_threads, group)
    return out.cpu().numpy()
