"""TEST INFRASTRUCTURE (oracle): CPU restatement of the RMVPE f0 estimator (reference src/rmvpe.py) as plain
functions over the reference's own `rmvpe.pt` state_dict.  Pinned against the reference module itself through
tests/golden/rmvpe_*.npz (tests/golden/make_golden.py imports /root/reference/src/rmvpe.py with
librosa.filters.mel supplied by `mel_filterbank` below -- librosa 0.9.1 is not installed; the filterbank formula is
restated from librosa's published source: PARITY UNPINNED for that one table)."""
import numpy as np
import torch
import torch.nn.functional as F


def mel_filterbank(sr=16000, n_fft=1024, n_mels=128, fmin=30.0, fmax=8000.0):
    """librosa.filters.mel(htk=True, norm='slaney') as called at rmvpe.py:277-284: triangular filters on the HTK
    mel scale (2595 log10(1 + f/700)), each scaled by 2 / (f_hi - f_lo); float32 (n_mels, 1 + n_fft/2)."""
    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    mmin = 2595.0 * np.log10(1.0 + fmin / 700.0)
    mmax = 2595.0 * np.log10(1.0 + fmax / 700.0)
    mel_f = 700.0 * (10.0 ** (np.linspace(mmin, mmax, n_mels + 2) / 2595.0) - 1.0)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, np.newaxis]
    return w.astype(np.float32)


def log_mel(audio, mel_basis):
    """MelSpectrogram.forward (rmvpe.py:295-325) with keyshift=0, speed=1, center=True: (1, N) -> (1, 128, T)."""
    fft = torch.stft(audio, n_fft=1024, hop_length=160, win_length=1024, window=torch.hann_window(1024), center=True,
                     return_complex=True)
    mag = torch.sqrt(fft.real.pow(2) + fft.imag.pow(2))
    return torch.log(torch.clamp(torch.matmul(mel_basis.to(mag.dtype), mag), min=1e-5))   # the table itself is float32 (librosa)


def _bn_eval(x, sd, name, eps=1e-5):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                        False, 0.0, eps)


def conv_block_res(sd, name, x):
    """ConvBlockRes.forward (rmvpe.py:54-58)."""
    y = F.relu(_bn_eval(F.conv2d(x, sd[name + ".conv.0.weight"], None, padding=1), sd, name + ".conv.1"))
    y = F.relu(_bn_eval(F.conv2d(y, sd[name + ".conv.3.weight"], None, padding=1), sd, name + ".conv.4"))
    if name + ".shortcut.weight" in sd:
        return y + F.conv2d(x, sd[name + ".shortcut.weight"], sd[name + ".shortcut.bias"])
    return y + x


def _count(sd, fmt):
    n = 0
    while any(k.startswith(fmt % n) for k in sd):
        n += 1
    return n


def e2e_forward(sd, mel):
    """E2E.forward (rmvpe.py:254-258) on mel (1, 128, T), T a multiple of 32 -> salience (1, T, 360)."""
    x = mel.transpose(-1, -2).unsqueeze(1)                       # (1, 1, T, 128)
    x = _bn_eval(x, sd, "unet.encoder.bn")
    n_enc = _count(sd, "unet.encoder.layers.%d.")
    skips = []
    for i in range(n_enc):
        for b in range(_count(sd, "unet.encoder.layers.%d.conv." % i + "%d.")):
            x = conv_block_res(sd, "unet.encoder.layers.%d.conv.%d" % (i, b), x)
        skips.append(x)
        x = F.avg_pool2d(x, 2)
    for i in range(_count(sd, "unet.intermediate.layers.%d.")):
        for b in range(_count(sd, "unet.intermediate.layers.%d.conv." % i + "%d.")):
            x = conv_block_res(sd, "unet.intermediate.layers.%d.conv.%d" % (i, b), x)
    for i in range(_count(sd, "unet.decoder.layers.%d.")):
        p = "unet.decoder.layers.%d." % i
        x = F.conv_transpose2d(x, sd[p + "conv1.0.weight"], None, stride=2, padding=1, output_padding=1)
        x = F.relu(_bn_eval(x, sd, p + "conv1.1"))
        x = torch.cat((x, skips[-1 - i]), dim=1)
        for b in range(_count(sd, p + "conv2.%d.")):
            x = conv_block_res(sd, p + "conv2.%d" % b, x)
    x = F.conv2d(x, sd["cnn.weight"], sd["cnn.bias"], padding=1)  # (1, 3, T, 128)
    x = x.transpose(1, 2).flatten(-2)                              # (1, T, 384)
    x = bigru(sd, x)
    return torch.sigmoid(F.linear(x, sd["fc.1.weight"], sd["fc.1.bias"]))


def bigru(sd, x):
    """torch.nn.GRU(384, 256, bidirectional) written out (rmvpe.py:11-20)."""
    T = x.shape[1]
    outs = []
    for suf, order in (("", range(T)), ("_reverse", range(T - 1, -1, -1))):
        wi, wh = sd["fc.0.gru.weight_ih_l0" + suf], sd["fc.0.gru.weight_hh_l0" + suf]
        bi, bh = sd["fc.0.gru.bias_ih_l0" + suf], sd["fc.0.gru.bias_hh_l0" + suf]
        hd = wh.shape[1]
        gi = F.linear(x[0], wi, bi)                                  # (T, 3*hd)
        h = torch.zeros(hd)
        out = torch.zeros(T, hd)
        for t in order:
            gh = F.linear(h, wh, bh)
            r = torch.sigmoid(gi[t, :hd] + gh[:hd])
            z = torch.sigmoid(gi[t, hd:2 * hd] + gh[hd:2 * hd])
            n = torch.tanh(gi[t, 2 * hd:] + r * gh[2 * hd:])
            h = (1 - z) * n + z * h
            out[t] = h
        outs.append(out)
    return torch.cat(outs, dim=1).unsqueeze(0)


def mel2hidden(sd, mel):
    """RMVPE.mel2hidden (rmvpe.py:350-357): reflect-pad frames to a multiple of 32."""
    n = mel.shape[-1]
    mel = F.pad(mel, (0, 32 * ((n - 1) // 32 + 1) - n), mode="reflect")
    return e2e_forward(sd, mel)[:, :n]


def to_local_average_cents(salience, thred=0.05):
    """RMVPE.to_local_average_cents (rmvpe.py:385-409), vectorised; float32 salience x float64 mapping."""
    salience = np.asarray(salience)
    cents_mapping = np.pad(20 * np.arange(360) + 1997.3794084376191, (4, 4))
    center = np.argmax(salience, axis=1)
    sal = np.pad(salience, ((0, 0), (4, 4)))
    idx = center[:, None] + np.arange(9)[None, :]
    todo_s = np.take_along_axis(sal, idx, axis=1)
    todo_c = cents_mapping[idx]
    devided = np.sum(todo_s * todo_c, 1) / np.sum(todo_s, 1)
    devided[np.max(sal, axis=1) <= thred] = 0
    return devided


def decode(salience, thred=0.03):
    """RMVPE.decode (rmvpe.py:359-364)."""
    cents = to_local_average_cents(salience, thred)
    f0 = 10 * (2 ** (cents / 1200))
    f0[f0 == 10] = 0
    return f0


def f0_to_coarse(f0, f0_up_key=0):
    """VC.get_f0 tail (vc_infer_pipeline.py:346, 361-368)."""
    f0 = f0 * pow(2, f0_up_key / 12)
    f0_mel_min = 1127 * np.log(1 + 50 / 700)
    f0_mel_max = 1127 * np.log(1 + 1100 / 700)
    f0bak = f0.copy()
    f0_mel = 1127 * np.log(1 + f0 / 700)
    f0_mel[f0_mel > 0] = (f0_mel[f0_mel > 0] - f0_mel_min) * 254 / (f0_mel_max - f0_mel_min) + 1
    f0_mel[f0_mel <= 1] = 1
    f0_mel[f0_mel > 255] = 255
    return np.rint(f0_mel).astype(np.int64), f0bak


def infer_from_audio(sd, audio, thred=0.03, mel_basis=None):
    """RMVPE.infer_from_audio (rmvpe.py:366-383)."""
    if mel_basis is None:
        mel_basis = torch.from_numpy(mel_filterbank())
    with torch.no_grad():
        mel = log_mel(torch.from_numpy(np.asarray(audio)).float().unsqueeze(0), mel_basis)
        hidden = mel2hidden(sd, mel)[0].numpy()
    return decode(hidden, thred), hidden
