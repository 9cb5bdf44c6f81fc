"""TEST INFRASTRUCTURE (oracle): CPU restatement of the MDX-Net separator path of the reference
(src/mdx.py): MDXModel.stft / istft, MDX.segment / pad_wave / _process_wave / process_wave and run_mdx's
peak-normalise / denoise / invert arithmetic, plus the TFC-TDF U-Net the ONNX file contains.

The framing / STFT / denoise / inversion code below is pinned against the reference's own src/mdx.py, run in the build
container with only onnxruntime / librosa / soundfile stubbed (tests/golden/make_mdx_golden.py -> tests/golden/
mdx_ref_tiny.npz, replayed bit-exactly by tests/test_oracle_golden.py).  The U-Net itself lives only in the downloaded
.onnx files (not available offline): restated from the published kuielab "ConvTDFNet"; PARITY UNPINNED for the network.
"""
import numpy as np
import torch
import torch.nn.functional as F


def stft(x, n_fft, hop, dim_f):
    """MDXModel.stft (mdx.py:37-43): x (B, 2, chunk) -> (B, 4, dim_f, dim_t) with channels (L.re, L.im, R.re, R.im)."""
    b = x.shape[0]
    x = x.reshape(-1, x.shape[-1])
    s = torch.stft(x, n_fft=n_fft, hop_length=hop, window=torch.hann_window(n_fft, periodic=True), center=True,
                   return_complex=True)
    s = torch.view_as_real(s).permute(0, 3, 1, 2)
    n_bins, dim_t = s.shape[2], s.shape[3]
    return s.reshape(b, 2, 2, n_bins, dim_t).reshape(b, 4, n_bins, dim_t)[:, :, :dim_f]


def istft(x, n_fft, hop):
    """MDXModel.istft (mdx.py:45-54): zero-pad bins above dim_f, inverse STFT -> (B, 2, chunk)."""
    b, _, dim_f, dim_t = x.shape
    n_bins = n_fft // 2 + 1
    x = torch.cat([x, torch.zeros(b, 4, n_bins - dim_f, dim_t)], -2)
    x = x.reshape(b, 2, 2, n_bins, dim_t).reshape(-1, 2, n_bins, dim_t).permute(0, 2, 3, 1).contiguous()
    y = torch.istft(torch.view_as_complex(x), n_fft=n_fft, hop_length=hop, window=torch.hann_window(n_fft, periodic=True),
                    center=True)
    return y.reshape(b, 2, -1)


def segment(wave, combine=True, chunk_size=0, margin_size=44100):
    """MDX.segment (mdx.py:92-141)."""
    if combine:
        out = None
        for i, seg in enumerate(wave):
            start = 0 if i == 0 else margin_size
            end = None if (i == len(wave) - 1 or margin_size == 0) else -margin_size
            out = seg[:, start:end] if out is None else np.concatenate((out, seg[:, start:end]), axis=-1)
        return out
    n = wave.shape[-1]
    if chunk_size <= 0 or chunk_size > n:
        chunk_size = n
    if margin_size > chunk_size:
        margin_size = chunk_size
    out = []
    for i, skip in enumerate(range(0, n, chunk_size)):
        margin = 0 if i == 0 else margin_size
        end = min(skip + chunk_size + margin_size, n)
        out.append(wave[:, skip - margin:end].copy())
        if end == n:
            break
    return out


def pad_wave(wave, n_fft, chunk_size):
    """MDX.pad_wave (mdx.py:143-171): returns (windows (n, 2, chunk_size) float32, pad, trim)."""
    n = wave.shape[1]
    trim = n_fft // 2
    gen = chunk_size - 2 * trim
    pad = gen - n % gen
    wp = np.concatenate((np.zeros((2, trim)), wave, np.zeros((2, pad)), np.zeros((2, trim))), 1)
    wins = [np.array(wp[:, i:i + chunk_size]) for i in range(0, n + pad, gen)]
    return torch.tensor(np.array(wins), dtype=torch.float32), pad, trim


def _bn(x, sd, name):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                        False, 0.0, 1e-5)


def _tfc_tdf(sd, name, x, l):
    for j in range(l):
        p = "%s.tfc.H.%d" % (name, j)
        k = sd[p + ".0.weight"].shape[-1]
        x = F.relu(_bn(F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], padding=k // 2), sd, p + ".1"))
    t = F.relu(_bn(F.linear(x, sd[name + ".tdf.0.weight"], sd[name + ".tdf.0.bias"]), sd, name + ".tdf.1"))
    t = F.relu(_bn(F.linear(t, sd[name + ".tdf.3.weight"], sd[name + ".tdf.3.bias"]), sd, name + ".tdf.4"))
    return x + t


def unet(sd, cfg, spec):
    """ConvTDFNet.forward: spec (B, 4, dim_f, dim_t) -> (B, 4, dim_f, dim_t)."""
    n, l = cfg["n"], cfg["l"]
    x = F.relu(_bn(F.conv2d(spec, sd["first_conv.0.weight"], sd["first_conv.0.bias"]), sd, "first_conv.1"))
    x = x.transpose(-1, -2)
    skips = []
    for i in range(n):
        x = _tfc_tdf(sd, "ds_dense.%d" % i, x, l)
        skips.append(x)
        x = F.relu(_bn(F.conv2d(x, sd["ds.%d.0.weight" % i], sd["ds.%d.0.bias" % i], stride=2), sd, "ds.%d.1" % i))
    x = _tfc_tdf(sd, "mid_dense", x, l)
    for i in range(n):
        x = F.relu(_bn(F.conv_transpose2d(x, sd["us.%d.0.weight" % i], sd["us.%d.0.bias" % i], stride=2), sd, "us.%d.1" % i))
        x = x * skips[-i - 1]
        x = _tfc_tdf(sd, "us_dense.%d" % i, x, l)
    x = x.transpose(-1, -2)
    return F.conv2d(x, sd["final_conv.0.weight"], sd["final_conv.0.bias"])


def process_wave(sd, cfg, wave, mt_threads=2, hop=1024):
    """MDX.process_wave (mdx.py:201-235) with the ORT session replaced by `unet`."""
    n_fft, dim_f, dim_t = cfg["n_fft"], cfg["dim_f"], cfg["dim_t"]
    chunk_size = hop * (dim_t - 1)
    chunk = wave.shape[-1] // mt_threads
    waves = segment(wave, False, chunk)
    outs = []
    for batch in waves:
        mix, pad, trim = pad_wave(batch, n_fft, chunk_size)
        pw = []
        with torch.no_grad():
            for m in mix.split(1):
                spec = stft(m, n_fft, hop, dim_f)
                y = istft(unet(sd, cfg, spec), n_fft, hop)
                pw.append(y[:, :, trim:-trim].transpose(0, 1).reshape(2, -1).numpy())
        outs.append(np.concatenate(pw, axis=-1)[:, :-pad])
    return segment(outs, True, chunk)


def run_mdx_arrays(sd, cfg, wave, denoise=True, compensation=1.0, m_threads=2):
    """run_mdx arithmetic (mdx.py:257-280) on arrays: returns (main stem, inverted stem), each (2, N)."""
    wave = wave.copy()
    peak = max(np.max(wave), abs(np.min(wave)))
    wave /= peak
    if denoise:
        out = -(process_wave(sd, cfg, -wave, m_threads)) + process_wave(sd, cfg, wave, m_threads)
        out *= 0.5
    else:
        out = process_wave(sd, cfg, wave, m_threads)
    out *= peak
    return out, (-out * compensation) + wave  # NB: the reference adds the *normalised* wave (mdx.py:260,280)
