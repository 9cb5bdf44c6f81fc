"""RMVPE f0 estimator on the gfx950 kernels behind the reference's class surface (reference src/rmvpe.py:328-409):
RMVPE(model_path, is_half, device).infer_from_audio(audio, thred) / mel2hidden / decode / to_local_average_cents,
plus `.mel_extractor(audio, center=True)`.

Pipeline (all arithmetic in libaicg_hip.so):  framed STFT (LDS FFT) -> |.| -> mel GEMM with fused log-clamp ->
DeepUnet (implicit-GEMM 3x3 convs with BatchNorm folded into weights at load, avg-pool, transposed convs as GEMM +
col2im, skip connections written straight into the concat buffers) -> BiGRU (input projection as one GEMM,
persistent recurrence kernel) -> Linear + sigmoid -> wave-per-frame argmax / float64 local-average decode.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops


def mel_filterbank(sr=16000, n_fft=1024, n_mels=128, fmin=30.0, fmax=8000.0):
    """Table of librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=True) (reference src/rmvpe.py:277-284):
    Slaney-normalised triangles on the HTK mel scale, float32 (n_mels, 1 + n_fft // 2).  librosa is not a
    dependency here; this is parameter-table construction at load time, not hot-path arithmetic."""
    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    to_mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    mel_pts = np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2)
    hz = 700.0 * (10.0 ** (mel_pts / 2595.0) - 1.0)
    width = np.diff(hz)
    ramps = hz[:, None] - fftfreqs[None, :]
    fb = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        fb[i] = np.maximum(0, np.minimum(-ramps[i] / width[i], ramps[i + 2] / width[i + 1]))
    fb *= (2.0 / (hz[2:n_mels + 2] - hz[:n_mels]))[:, None]
    return fb.astype(np.float32)


def _fold_bn(w, sd, bn, eps=1e-5, transposed=False):
    """conv (no bias) followed by eval-mode BatchNorm -> (w * s[co], shift)."""
    s = sd[bn + ".weight"].float() / torch.sqrt(sd[bn + ".running_var"].float() + eps)
    shift = sd[bn + ".bias"].float() - sd[bn + ".running_mean"].float() * s
    w = w.float()
    w = w * (s.view(1, -1, 1, 1) if transposed else s.view(-1, 1, 1, 1))
    return w, shift


def _count(sd, fmt):
    n = 0
    while any(k.startswith(fmt % n) for k in sd):
        n += 1
    return n


class _ResBlock:
    """ConvBlockRes (src/rmvpe.py:23-58) with both BatchNorms folded."""

    def __init__(self, sd, name, dev):
        w1, b1 = _fold_bn(sd[name + ".conv.0.weight"], sd, name + ".conv.1")
        w2, b2 = _fold_bn(sd[name + ".conv.3.weight"], sd, name + ".conv.4")
        self.c1 = ops.PackedConv(w1, b1, padding=1, device=dev)
        self.c2 = ops.PackedConv(w2, b2, padding=1, device=dev)
        self.sc = None
        if name + ".shortcut.weight" in sd:
            self.sc = ops.PackedConv(sd[name + ".shortcut.weight"].float(), sd[name + ".shortcut.bias"], device=dev)

    def __call__(self, x, out=None):
        y = ops.conv(x, self.c1, act=ops.ACT_RELU)
        res = x if self.sc is None else ops.conv(x, self.sc)
        return ops.conv(y, self.c2, act=ops.ACT_RELU, res=res, out=out)  # relu(bn(conv)) + shortcut


class MelSpectrogram:
    """rmvpe.MelSpectrogram.forward with keyshift=0, speed=1 (the only way RMVPE calls it, src/rmvpe.py:370)."""

    def __init__(self, is_half, n_mel_channels, sampling_rate, win_length, hop_length, n_fft=None, mel_fmin=0,
                 mel_fmax=None, clamp=1e-5, device="cpu"):
        self.n_fft = win_length if n_fft is None else n_fft
        self.hop_length, self.win_length, self.clamp = hop_length, win_length, clamp
        self.n_mel_channels, self.sampling_rate = n_mel_channels, sampling_rate
        self.device = torch.device(device)
        fb = mel_filterbank(sampling_rate, self.n_fft, n_mel_channels, mel_fmin, mel_fmax)
        self.mel_basis = torch.from_numpy(fb)
        self._pc = None

    def to(self, device):
        self.device = torch.device(device)
        self._pc = None
        return self

    @ops.fp32_only
    def __call__(self, audio, keyshift=0, speed=1, center=True):
        assert keyshift == 0 and speed == 1 and center, "only the configuration RMVPE.infer_from_audio uses"
        if self._pc is None:
            self._pc = ops.PackedConv(self.mel_basis, None, device=self.device)
        audio = audio.to(self.device).float()
        spec = ops.stft(audio, self.n_fft, self.hop_length)           # (1, 2, bins, T)
        mag = ops.complex_abs(spec[:, 0].contiguous(), spec[:, 1].contiguous())  # (1, bins, T)
        return ops.conv(mag, self._pc, act=ops.ACT_LOGCLAMP, act_slope=self.clamp)  # log(clamp(mel @ mag, 1e-5))


class E2E:
    """rmvpe.E2E.forward (src/rmvpe.py:254-258) from the reference state_dict."""

    @ops.fp32_only   # pitch-bin selection stays bit-exact under AICG_PRECISION=bf16x3
    def __init__(self, sd, device):
        dev = torch.device(device)
        self.device = dev
        s = sd["unet.encoder.bn.weight"].float() / torch.sqrt(sd["unet.encoder.bn.running_var"].float() + 1e-5)
        self.in_scale = s.to(dev)
        self.in_shift = (sd["unet.encoder.bn.bias"].float() - sd["unet.encoder.bn.running_mean"].float() * s).to(dev)
        self.enc = []
        for i in range(_count(sd, "unet.encoder.layers.%d.")):
            nb = _count(sd, "unet.encoder.layers.%d.conv." % i + "%d.")
            self.enc.append([_ResBlock(sd, "unet.encoder.layers.%d.conv.%d" % (i, b), dev) for b in range(nb)])
        self.inter = []
        for i in range(_count(sd, "unet.intermediate.layers.%d.")):
            nb = _count(sd, "unet.intermediate.layers.%d.conv." % i + "%d.")
            self.inter += [_ResBlock(sd, "unet.intermediate.layers.%d.conv.%d" % (i, b), dev) for b in range(nb)]
        self.dec = []
        for i in range(_count(sd, "unet.decoder.layers.%d.")):
            p = "unet.decoder.layers.%d." % i
            w, shift = _fold_bn(sd[p + "conv1.0.weight"], sd, p + "conv1.1", transposed=True)
            up = ops.PackedConvTranspose(w, shift, stride=(2, 2), padding=(1, 1), output_padding=(1, 1), device=dev)
            nb = _count(sd, p + "conv2.%d.")
            self.dec.append((up, [_ResBlock(sd, p + "conv2.%d" % b, dev) for b in range(nb)]))
        self.cnn = ops.PackedConv(sd["cnn.weight"].float(), sd["cnn.bias"], padding=1, device=dev)
        # BiGRU: both directions' input projections in one GEMM; W_hh transposed for unit-stride reads
        wi = torch.cat([sd["fc.0.gru.weight_ih_l0"], sd["fc.0.gru.weight_ih_l0_reverse"]], 0).float()
        bi = torch.cat([sd["fc.0.gru.bias_ih_l0"], sd["fc.0.gru.bias_ih_l0_reverse"]], 0).float()
        self.gru_in = ops.PackedConv(wi, bi, device=dev)
        self.hidden = sd["fc.0.gru.weight_hh_l0"].shape[1]
        self.whh_t = torch.stack([sd["fc.0.gru.weight_hh_l0"].float().t().contiguous(),
                                  sd["fc.0.gru.weight_hh_l0_reverse"].float().t().contiguous()]).contiguous().to(dev)
        self.bhh = torch.cat([sd["fc.0.gru.bias_hh_l0"], sd["fc.0.gru.bias_hh_l0_reverse"]]).float().contiguous().to(dev)
        self.fc = ops.PackedConv(sd["fc.1.weight"].float(), sd["fc.1.bias"], device=dev)

    def features(self, mel):
        """mel (1, 128, T) with T % 32 == 0 -> U-Net + cnn output (384, T), row c*128 + f: the BiGRU's input sequence."""
        T = mel.shape[-1]
        x = mel[0].t().contiguous().view(1, 1, T, mel.shape[1])          # mel.transpose(-1,-2).unsqueeze(1)
        x = ops.channel_affine(x, self.in_scale, self.in_shift)
        cats = []
        for blocks in self.enc:
            c_out = blocks[-1].c2.cout
            h, w = x.shape[2], x.shape[3]
            # decoder concat buffer: [up-sampled ; skip]; the encoder writes its skip output in place
            cat = torch.empty((1, 2 * c_out, h, w), dtype=torch.float32, device=x.device)
            for b, blk in enumerate(blocks):
                x = blk(x, out=cat[:, c_out:] if b == len(blocks) - 1 else None)
            cats.append(cat)
            x = ops.avgpool2x2(x)
        for blk in self.inter:
            x = blk(x)
        for i, (up, blocks) in enumerate(self.dec):
            cat = cats[-1 - i]
            c_out = up.cout
            ops.conv_transpose(x, up, out=cat[:, :c_out], act=ops.ACT_RELU)
            x = cat
            for blk in blocks:
                x = blk(x)
        y = ops.conv(x, self.cnn)                                        # (1, 3, T, 128)
        return y[0].permute(0, 2, 1).reshape(-1, T).contiguous()         # (384, T)

    def time_reach(self):
        """Frames on either side that an output frame of `features` can depend on: every 3x3 conv reaches one row of its level
        (2^level frames), a 2x2 average pool stays inside its cell, a k=3 s=2 transposed conv reaches one row of the coarser
        level.  Rounded up to whole pooling cells plus one (cuts must sit on the coarsest pooling grid)."""
        cell = 1 << len(self.enc)
        r = 1                                                            # cnn
        for lvl, blocks in enumerate(self.enc):
            r += 2 * len(blocks) * (1 << lvl) + (1 << lvl)
        r += 2 * len(self.inter) * cell
        for i, (_, blocks) in enumerate(self.dec):
            lvl = len(self.enc) - 1 - i
            r += 2 * (1 << lvl) + 2 * len(blocks) * (1 << lvl)
        return (r + cell - 1) // cell * cell + cell

    def features_sharded(self, mel, group):
        """`features` with the time axis cut over the ranks of `group` (the U-Net is the shardable part of the f0
        branch; the recurrence behind it is not).  Every rank runs its segment plus `time_reach()` frames of real context on
        either side -- inside that margin the segment's zero padding differs from the whole track's, beyond it nothing does --
        keeps the interior and one all_gather of equal-size blocks rebuilds (384, T) everywhere.  Cuts sit on the coarsest
        pooling grid.  Same arithmetic per output frame as the unsharded call; the conv dispatcher may pick other tiles for the
        shorter problem, i.e. another fp32 summation order (~1e-7 relative, like any other tile change)."""
        from . import dist as adist
        rank, world = adist.world(group)
        T = mel.shape[-1]
        cell = 1 << len(self.enc)
        halo = self.time_reach()
        per = ((T + world - 1) // world + cell - 1) // cell * cell      # frames per rank, whole cells
        if adist.single(world, group) or per < 2 * halo:                 # one rank / short tracks: the margins would dominate
            return self.features(mel)
        a, b = min(rank * per, T), min((rank + 1) * per, T)
        block = torch.zeros((mel.shape[1] * self.cnn.cout, per), dtype=torch.float32, device=mel.device)
        if b > a:
            lo, hi = max(0, a - halo), min(T, b + halo)
            seg = self.features(mel[:, :, lo:hi].contiguous())
            block[:, : b - a] = seg[:, a - lo: b - lo]
        allb = adist.all_gather_equal(block.t().contiguous(), group)     # (world * per, 384): rank-major = time-major
        return allb[:T].t().contiguous()

    def __call__(self, mel, two_workgroups=None, group=None):
        """mel (1, 128, T) with T % 32 == 0 -> salience (1, T, 360)."""
        T = mel.shape[-1]
        feat = (self.features(mel) if group is None else self.features_sharded(mel, group)).unsqueeze(0)   # (1, 384, T)
        gi = ops.conv(feat, self.gru_in)                                 # (1, 6*hidden, T)
        hseq = ops.gru_bidir(gi[0], self.whh_t, self.bhh, self.hidden, two_workgroups)   # (2*hidden, T)
        sal = ops.conv(hseq.unsqueeze(0), self.fc, act=ops.ACT_SIGMOID)  # (1, 360, T)
        return sal[0].t().contiguous().unsqueeze(0)                      # (1, T, 360)


    def progressive(self, mel, n_frames, nseg, on_ready, two_workgroups=None, group=None):
        """__call__ with the recurrence in `nseg` segments (ops.GruSegments): after each one, `on_ready(lo, hi, salience)` is called for
        the frame ranges [lo, hi) < n_frames that BOTH directions have now passed -- the track's middle first, its ends last -- with
        their (hi - lo, 360) salience rows; everything is queued on the current stream.  Same arithmetic per frame as __call__ (the
        classifier is a per-frame GEMM; the dispatcher may tile a short range differently, i.e. another fp32 summation order)."""
        T = mel.shape[-1]
        feat = (self.features(mel) if group is None else self.features_sharded(mel, group)).unsqueeze(0)
        gi = ops.conv(feat, self.gru_in)
        seg = ops.GruSegments(gi[0], self.whh_t, self.bhh, self.hidden, two_workgroups)
        step = ((T + nseg - 1) // nseg + 31) // 32 * 32
        lo_prev = hi_prev = None
        for k in range(nseg):
            seg.run((k + 1) * step)
            lo, hi = seg.ready()
            if lo >= hi:
                continue
            new = [(lo, hi)] if lo_prev is None else [(lo, lo_prev), (hi_prev, hi)]
            lo_prev, hi_prev = lo, hi
            for a, b in new:
                b = min(b, n_frames)
                if a < b:
                    sal = ops.conv(seg.out[:, a:b].unsqueeze(0), self.fc, act=ops.ACT_SIGMOID)      # (1, 360, b - a)
                    on_ready(a, b, sal[0].t().contiguous())
        return seg


class RMVPE:
    def __init__(self, model_path, is_half, device=None, state_dict=None):
        self.resample_kernel = {}
        self.is_half = is_half
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = device
        sd = state_dict if state_dict is not None else torch.load(model_path, map_location="cpu")
        self.model = E2E(sd, device)
        self.mel_extractor = MelSpectrogram(is_half, 128, 16000, 1024, 160, None, 30, 8000, device=device)
        cents_mapping = 20 * np.arange(360) + 1997.3794084376191
        self.cents_mapping = np.pad(cents_mapping, (4, 4))

    def mel2hidden(self, mel, two_workgroups=None, group=None):
        n_frames = mel.shape[-1]
        mel = F.pad(mel, (0, 32 * ((n_frames - 1) // 32 + 1) - n_frames), mode="reflect")  # frame re-indexing only
        hidden = self.model(mel, two_workgroups, group)
        return hidden[:, :n_frames]

    def _decode_device(self, hidden, thred):
        cents, f0 = ops.salience_decode(hidden, thred)
        return cents, f0

    def decode(self, hidden, thred=0.03):
        """hidden: (T, 360) numpy or tensor -> f0 float64 numpy (src/rmvpe.py:359-364)."""
        h = torch.as_tensor(hidden).to(self.device)
        return self._decode_device(h, thred)[1].cpu().numpy()

    def to_local_average_cents(self, salience, thred=0.05):
        s = torch.as_tensor(salience).to(self.device)
        return self._decode_device(s, thred)[0].cpu().numpy()

    def infer_from_audio_device(self, audio, thred=0.03, two_workgroups=None, group=None):
        """infer_from_audio without the final device->host copy: everything is queued on the current stream.  The caller
        must consult ops.gru_timed_out() once the stream has drained and, if set, call again with two_workgroups=False.
        `group`: the ranks of a torch.distributed group split the U-Net over time (E2E.features_sharded); every rank must call."""
        if not torch.is_tensor(audio):
            audio = torch.from_numpy(np.asarray(audio))
        audio = audio.float().to(self.device).unsqueeze(0)
        mel = self.mel_extractor(audio, center=True)
        hidden = self.mel2hidden(mel, two_workgroups, group)
        return self._decode_device(hidden[0], thred)[1]

    def infer_progressive(self, audio, thred, nseg, on_f0, two_workgroups=None, group=None):
        """infer_from_audio_device with the recurrence in `nseg` segments: `on_f0(lo, hi, f0)` receives the float64 f0 of frames
        [lo, hi) as soon as both GRU directions have passed them (middle of the track first), queued on the current stream.
        -> number of frames."""
        if not torch.is_tensor(audio):
            audio = torch.from_numpy(np.asarray(audio))
        audio = audio.float().to(self.device).unsqueeze(0)
        mel = self.mel_extractor(audio, center=True)
        n_frames = mel.shape[-1]
        mel = F.pad(mel, (0, 32 * ((n_frames - 1) // 32 + 1) - n_frames), mode="reflect")
        # (last_segments: the recurrence's handle, for callers that poll its error word while it runs -- ops.GruSegments.timed_out)
        self.last_segments = self.model.progressive(mel, n_frames, nseg, lambda a, b, sal: on_f0(a, b, self._decode_device(sal, thred)[1]),
                                                    two_workgroups, group)
        return n_frames

    def infer_from_audio(self, audio, thred=0.03, group=None):
        f0 = self.infer_from_audio_device(audio, thred, group=group).cpu().numpy()
        if ops.gru_timed_out():
            # the partner workgroups of the two-workgroup recurrence were not co-resident in time (busy / shared GPU):
            # recompute on the single-workgroup kernel instead of failing the conversion (locally, without the group: the
            # other ranks may not have timed out and would not join a collective)
            f0 = self.infer_from_audio_device(audio, thred, two_workgroups=False).cpu().numpy()
        return f0
