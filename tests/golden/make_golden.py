"""Regenerate tests/golden/*.npz by running the REFERENCE ITSELF (imported from /root/reference/src, build
container only) on seeded parameters from oracle/weights.py and seeded inputs.  The fixtures pin the oracle
(tests/test_oracle_golden.py) and travel to the GPU box, where /root/reference does not exist.

    python tests/golden/make_golden.py
"""
import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
sys.path.insert(0, ROOT)


@contextlib.contextmanager
def injected_noise(draws):
    """Make torch.randn_like return the given tensors in order (reference RNG sites: models.py:748, :368)."""
    orig_randn_like, orig_rand = torch.randn_like, torch.rand
    it = iter(draws)

    def fake_randn_like(x, *a, **k):
        n = next(it)
        assert n.shape == x.shape, (n.shape, x.shape)
        return n.to(x.dtype)

    torch.randn_like = fake_randn_like
    try:
        yield
    finally:
        torch.randn_like, torch.rand = orig_randn_like, orig_rand


def make_synth(name, cfg, T, seed):
    sys.path.insert(0, REF)
    from infer_pack.models import SynthesizerTrnMs768NSFsid
    from oracle import weights
    from oracle.inputs import synth_inputs
    sd = weights.synth_state_dict(cfg, seed)
    net = SynthesizerTrnMs768NSFsid(*cfg, is_half=False)
    del net.enc_q
    missing = net.load_state_dict(sd, strict=False)
    assert not missing.missing_keys and not missing.unexpected_keys, missing
    net.eval()
    phone, pitch, f0, noise_z, noise_src = synth_inputs(cfg, T, seed + 1)
    sid = torch.tensor([1])
    with torch.no_grad(), injected_noise([noise_z, noise_src.unsqueeze(-1)]):
        o, _, (z, z_p, m_p, logs_p) = net.infer(phone, torch.tensor([T]), pitch, f0, sid)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), cfg_T=np.array([T]), seed=np.array([seed]),
                        audio=o[0, 0].numpy(), z=z[0].numpy(), m_p=m_p[0].numpy(), logs_p=logs_p[0].numpy())
    print(name, "audio", tuple(o.shape), float(o.abs().max()), float(o.pow(2).mean().sqrt()))


if __name__ == "__main__":
    from oracle import weights
    make_synth("synth_tiny_T24", weights.SYNTH_CFG_TINY, 24, 1234)
    make_synth("synth_40k_T16", weights.SYNTH_CFG_40K_V2, 16, 1234)
