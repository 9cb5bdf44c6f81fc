"""Thin torch-tensor wrappers over the C ABI (include/aicg.h).  No arithmetic happens here: these
functions validate shapes, allocate outputs with torch, and hand raw device pointers + the current HIP
stream to libaicg_hip.so."""
import math

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream(t):
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return 0


def _check(*tensors):
    """All tensors must be fp32/int, contiguous where required by the caller, and live where the bound
    library executes: HIP device memory for the product library, host memory for the test emulator."""
    be = _lib.backend()
    for t in tensors:
        if t is None:
            continue
        if be == "hip" and not t.is_cuda:
            raise RuntimeError("aicovergen_amd: tensor on %s, but the HIP kernels need device memory" % t.device)
        if be == "emu" and t.is_cuda:
            raise RuntimeError("emulator backend needs host tensors")


# ---------------------------------------------------------------------------------------------------
# STFT / iSTFT
# ---------------------------------------------------------------------------------------------------
_fft_tables = {}


def fft_tables(n_fft, device, window=None):
    """(window, tw_half, tw_full) on `device`; twiddles are evaluated in float64 and rounded once."""
    key = (n_fft, str(device))
    if key not in _fft_tables:
        m = n_fft // 2
        k = np.arange(m, dtype=np.float64)
        th = np.exp(-2j * np.pi * k / m)
        k2 = np.arange(m + 1, dtype=np.float64)
        tf = np.exp(-2j * np.pi * k2 / n_fft)
        tw_half = torch.from_numpy(np.stack([th.real, th.imag], -1).astype(np.float32)).contiguous().to(device)
        tw_full = torch.from_numpy(np.stack([tf.real, tf.imag], -1).astype(np.float32)).contiguous().to(device)
        hann = torch.hann_window(n_fft, periodic=True, dtype=torch.float32).to(device)
        _fft_tables[key] = (hann, tw_half, tw_full)
    hann, tw_half, tw_full = _fft_tables[key]
    if window is None:
        window = hann
    return window, tw_half, tw_full


def stft(x, n_fft, hop, n_bins_out=None, frame_major=False, window=None):
    """x: (n_sig, L) fp32.  Returns the complex spectrogram split into planes:
    frame_major=False -> (n_sig, 2, n_bins_out, n_frames)  [torch.stft + view_as_real + permute layout]
    frame_major=True  -> (n_sig, 2, n_frames, n_bins_out)  [bins contiguous: coalesced stores]"""
    assert x.dim() == 2 and x.dtype == torch.float32
    x = x.contiguous()
    n_sig, L = x.shape
    n_frames = 1 + L // hop
    nb = n_fft // 2 + 1 if n_bins_out is None else n_bins_out
    window, tw_half, tw_full = fft_tables(n_fft, x.device, window)
    _check(x, window)
    if frame_major:
        out = torch.empty((n_sig, 2, n_frames, nb), dtype=torch.float32, device=x.device)
        o_bin, o_frame = 1, nb
    else:
        out = torch.empty((n_sig, 2, nb, n_frames), dtype=torch.float32, device=x.device)
        o_bin, o_frame = n_frames, 1
    _lib.call("aicg_stft", _ptr(x), _ptr(out), _ptr(window), _ptr(tw_half), _ptr(tw_full), n_sig, L, n_fft, hop,
              n_frames, nb, 2 * nb * n_frames, nb * n_frames, o_bin, o_frame, _stream(x))
    return out


def istft(spec, n_fft, hop, length, frame_major=False, window=None):
    """spec: (n_sig, 2, n_bins_in, n_frames) (or (n_sig, 2, n_frames, n_bins_in) if frame_major);
    bins above n_bins_in are zero.  Returns (n_sig, length)."""
    assert spec.dim() == 4 and spec.shape[1] == 2 and spec.dtype == torch.float32
    spec = spec.contiguous()
    n_sig = spec.shape[0]
    if frame_major:
        n_frames, nb = spec.shape[2], spec.shape[3]
        i_bin, i_frame = 1, nb
    else:
        nb, n_frames = spec.shape[2], spec.shape[3]
        i_bin, i_frame = n_frames, 1
    window, tw_half, tw_full = fft_tables(n_fft, spec.device, window)
    _check(spec, window)
    frames = torch.empty((n_sig, n_frames, n_fft), dtype=torch.float32, device=spec.device)
    out = torch.empty((n_sig, length), dtype=torch.float32, device=spec.device)
    st = _stream(spec)
    _lib.call("aicg_istft_frames", _ptr(spec), _ptr(frames), _ptr(window), _ptr(tw_half), _ptr(tw_full), n_sig,
              n_fft, n_frames, nb, 2 * nb * n_frames, nb * n_frames, i_bin, i_frame, st)
    _lib.call("aicg_istft_ola", _ptr(frames), _ptr(window), _ptr(out), n_sig, length, n_fft, hop, n_frames, st)
    return out
