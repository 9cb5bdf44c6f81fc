"""The oracle restatement (oracle/synth.py) pinned against outputs of the reference itself
(tests/golden/*.npz, generated in the build container from /root/reference by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_rms
from oracle import synth, weights
from oracle.inputs import synth_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name,cfg,T", [("synth_tiny_T24", weights.SYNTH_CFG_TINY, 24),
                                        ("synth_40k_T16", weights.SYNTH_CFG_40K_V2, 16)])
def test_oracle_synth_matches_reference_golden(name, cfg, T):
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    sd = weights.synth_state_dict(cfg, int(gold["seed"][0]))
    phone, pitch, f0, nz, ns = synth_inputs(cfg, T, int(gold["seed"][0]) + 1)
    with torch.no_grad():
        o, (z, z_p, m_p, logs_p) = synth.synth_infer(sd, cfg, phone, pitch, f0, torch.tensor([1]), nz, ns)
    assert rel_rms(m_p[0], torch.from_numpy(gold["m_p"])) < 1e-5
    assert rel_rms(z[0], torch.from_numpy(gold["z"])) < 1e-5
    assert rel_rms(o[0, 0], torch.from_numpy(gold["audio"])) < 1e-5
