"""Seeded synthetic inputs shared by the golden generator, the parity tests and
bench.py (all inputs are synthetic, seed 1234, generated in memory)."""
import numpy as np
import torch


def synth_inputs(cfg, T, seed):
    g = torch.Generator().manual_seed(seed)
    upp = int(np.prod(cfg[12]))
    phone = torch.randn(1, T, 768, generator=g)
    pitch = torch.randint(1, 255, (1, T), generator=g)
    f0 = 110.0 * 2 ** (torch.rand(1, T, generator=g) * 2)
    f0[:, T // 3: T // 3 + max(2, T // 8)] = 0.0  # an unvoiced stretch
    noise_z = torch.randn(1, cfg[2], T, generator=g)
    noise_src = torch.randn(1, T * upp, generator=g)
    return phone, pitch, f0, noise_z, noise_src


def vocal_like(seconds, sr=16000, seed=1234):
    """S16 of SURVEY 8(d): harmonic source (8 partials, 1/k roll-off) with an f0 glide 110 -> 440 Hz + 5.5 Hz
    vibrato, a syllabic envelope with a near-silent gap every ~7 s, white noise at -50 dBFS; peak 0.9; float32."""
    rng = np.random.default_rng(seed)
    n = int(round(seconds * sr))
    t = np.arange(n) / sr
    glide = 110.0 * 2.0 ** (2.0 * (0.5 - 0.5 * np.cos(2 * np.pi * t / 11.0)))
    f0 = glide * (1.0 + 0.02 * np.sin(2 * np.pi * 5.5 * t))
    phase = 2 * np.pi * np.cumsum(f0) / sr
    x = np.zeros(n)
    for k in range(1, 9):
        x += np.sin(k * phase) / k
    env = 0.55 + 0.45 * np.sin(2 * np.pi * 3.1 * t) ** 2
    gap = ((t % 7.0) > 6.6).astype(np.float64)
    env = env * (1.0 - 0.995 * gap)
    x = x * env + 10 ** (-50 / 20) * rng.standard_normal(n)
    return (0.9 * x / np.abs(x).max()).astype(np.float32)


def song_like(seconds, sr=44100, seed=1234):
    """S44 of SURVEY 8(d): centre-panned voice + decorrelated 'accompaniment' (detuned saw chords, band-limited
    by construction) + -60 dBFS noise; stereo float32 (2, N), peak 0.95."""
    rng = np.random.default_rng(seed)
    n = int(round(seconds * sr))
    t = np.arange(n) / sr
    voice = vocal_like(seconds, sr, seed).astype(np.float64)[:n]
    acc = np.zeros(n)
    for base in (130.81, 164.81, 196.0):
        for det in (0.997, 1.0, 1.004):
            f = base * det
            for k in range(1, 20):
                if f * k > 6000:
                    break
                acc += np.sin(2 * np.pi * f * k * t + rng.uniform(0, 6.28)) / k
    acc *= 0.25 * (0.6 + 0.4 * np.sin(2 * np.pi * 0.5 * t))
    d = 11
    left = voice + acc + 10 ** (-60 / 20) * rng.standard_normal(n)
    right = voice + np.concatenate([np.zeros(d), acc[:-d]]) + 10 ** (-60 / 20) * rng.standard_normal(n)
    out = np.stack([left, right])
    return (0.95 * out / np.abs(out).max()).astype(np.float32)


def fake_crepe_tracks(n, seed):
    """Deterministic stand-in for torchcrepe.predict's outputs on n frames: (pitch float32 with some sub-0.001 entries, which the
    reference gates to NaN; periodicity float32 with stretches below the 0.1 voicing gate).  Used where the code AROUND the
    network is under test (tests/golden/make_crepe_golden.py and tests/test_crepe.py share it)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    pitch = (180.0 + 60.0 * np.sin(t / 17.0) + rng.normal(0.0, 3.0, n)).astype(np.float32)
    pitch[rng.random(n) < 0.05] = 0.0005
    pd = rng.random(n).astype(np.float32)
    for s in range(0, n, 53):
        pd[s: s + 9] *= 0.05
    return pitch, pd
