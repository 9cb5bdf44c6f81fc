"""TEST INFRASTRUCTURE (oracle): seeded synthetic inputs shared by the golden generator, the parity tests and
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
