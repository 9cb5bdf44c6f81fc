"""Signal-processing kernels either side of the networks (csrc/dsp.hip) against numpy / scipy:
3-tap NaN-aware median / mean, zero-phase IIR (filtfilt), polyphase resampling (retrieval: tests/test_retrieval.py)."""
import numpy as np
import pytest
import torch
from scipy import signal

from aicovergen_amd import ops, retrieval


def _filter3_ref(x, mode):
    out = np.full_like(x, np.nan)
    for i in range(len(x)):
        w = x[max(0, i - 1): i + 2]
        w = w[~np.isnan(w)]
        if len(w) == 0:
            continue
        out[i] = np.sort(w)[(len(w) - 1) // 2] if mode == "median" else w.mean(dtype=np.float32)
        if mode == "mean" and out[i] == 0:
            out[i] = np.nan
    return out


def test_filter3_median_mean(dev):
    """torchcrepe.filter.median / .mean with win_length 3: truncated windows at the ends, NaNs skipped, all-NaN window -> NaN."""
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 64, 65, 1000, 4097):
        x = rng.standard_normal(n).astype(np.float32)
        if n > 10:
            x[[0, 5, 6, 7, n - 1, n // 2]] = np.nan
        xt = dev.t(torch.from_numpy(x))
        for mode in ("median", "mean"):
            got = ops.filter3(xt, mode).cpu().numpy()
            want = _filter3_ref(x, mode)
            assert np.array_equal(np.isnan(got), np.isnan(want)), (n, mode)
            assert np.allclose(got[~np.isnan(want)], want[~np.isnan(want)], rtol=2e-7, atol=0), (n, mode)
    # median of an exact sequence is bit-exact
    x = rng.standard_normal(777).astype(np.float32)
    assert np.array_equal(ops.filter3(dev.t(torch.from_numpy(x)), "median").cpu().numpy(), _filter3_ref(x, "median"))


def _filtfilt_longdouble(b, a, x):
    """scipy.signal.filtfilt's algorithm in extended precision: the yardstick for the rounding noise of the float64 versions."""
    b, a, x = (np.asarray(v, dtype=np.longdouble) for v in (b, a, x))
    zi = np.asarray(signal.lfilter_zi(np.asarray(b, float), np.asarray(a, float)), dtype=np.longdouble)
    p = 3 * max(len(a), len(b))
    ext = np.concatenate([2 * x[0] - x[p:0:-1], x, 2 * x[-1] - x[-2:-p - 2:-1]])

    def lfilt(u, z):
        z = z.copy()
        y = np.empty_like(u)
        for k in range(len(u)):
            y[k] = z[0] + b[0] * u[k]
            for q in range(len(z) - 1):
                z[q] = z[q + 1] + u[k] * b[q + 1] - y[k] * a[q + 1]
            z[-1] = u[k] * b[-1] - y[k] * a[-1]
        return y
    y1 = lfilt(ext, zi * ext[0])
    y2 = lfilt(y1[::-1], zi * y1[-1])
    return y2[::-1][p:-p]


def test_filtfilt_matches_scipy(dev):
    """48 Hz Butterworth high-pass of VC.pipeline (vc_infer_pipeline.py:22,513).  One block = scipy's own sequential recurrence:
    bit-identical.  Block-parallel: within the filter's own rounding noise -- measured against an extended-precision run, the
    device result is no further from the truth than scipy's (both ~1e-8 of the peak: five poles clustered at z = 1)."""
    rng = np.random.default_rng(1)
    b, a = signal.butter(N=5, Wn=48, btype="high", fs=16000)
    x = rng.standard_normal(40000) * 0.1 + 0.25 + 0.2 * np.sin(np.arange(40000) / 900.0)
    want = signal.filtfilt(b, a, x)
    one = ops.filtfilt_f64(dev.t(torch.from_numpy(x)), b, a, block=1 << 20).cpu().numpy()
    assert np.array_equal(one, want)
    got = ops.filtfilt_f64(dev.t(torch.from_numpy(x)), b, a, block=2048).cpu().numpy()
    peak = np.abs(want).max()
    assert np.abs(got - want).max() < 2e-7 * peak
    xs = x[:6000]
    truth = _filtfilt_longdouble(b, a, xs).astype(np.float64)
    e_scipy = np.abs(signal.filtfilt(b, a, xs) - truth).max()
    e_dev = np.abs(ops.filtfilt_f64(dev.t(torch.from_numpy(xs)), b, a, block=512).cpu().numpy() - truth).max()
    print("filtfilt vs extended precision: scipy %.2e, device %.2e (peak %.2f)" % (e_scipy, e_dev, np.abs(truth).max()))
    assert e_dev < 4 * max(e_scipy, 1e-12) + 1e-9 * np.abs(truth).max()
    with pytest.raises(RuntimeError):
        ops.filtfilt_f64(dev.t(torch.zeros(10, dtype=torch.float64)), b, a)   # shorter than scipy's padlen


def test_resample_poly_matches_scipy(dev):
    """44.1 kHz stereo -> 16 kHz mono (the opt-in device hand-over between separation and conversion) and a 48 k -> 16 k case."""
    rng = np.random.default_rng(2)
    for (sr_in, sr_out, ch, n) in [(44100, 16000, 2, 44100 + 377), (48000, 16000, 1, 9000), (16000, 40000, 1, 3000)]:
        x = rng.standard_normal((ch, n)).astype(np.float32)
        g = np.gcd(sr_in, sr_out)
        want = signal.resample_poly(x.mean(0), sr_out // g, sr_in // g)
        got = ops.resample_poly_mono(dev.t(torch.from_numpy(x)), sr_in, sr_out).cpu().numpy()
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 2e-6 * np.abs(want).max(), (sr_in, sr_out)
