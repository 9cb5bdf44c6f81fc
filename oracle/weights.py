"""TEST INFRASTRUCTURE (oracle): seeded random parameters in the reference's own checkpoint formats.

No trained weights exist offline, so parity and benchmarks use random parameters with the
exact state_dict keys / shapes of the reference constructors (checked against the constructors themselves in
tests/test_oracle_vs_reference.py when /root/reference is present).  Only tests/, bench.py's cpu_baseline leg
and __graft_entry__.smoke() may import this package.
"""
import math

import numpy as np
import torch

SYNTH_CFG_40K_V2 = [1025, 32, 192, 192, 768, 2, 6, 3, 0, "1", [3, 7, 11], [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                    [10, 10, 2, 2], 512, [16, 16, 4, 4], 109, 256, 40000]
# a structurally identical but tiny generator for CPU-side tests (same code paths, fewer channels)
SYNTH_CFG_TINY = [1025, 32, 64, 64, 128, 2, 2, 3, 0, "1", [3, 7, 11], [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                  [10, 10, 2, 2], 128, [16, 16, 4, 4], 4, 32, 40000]


class _Gen:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)

    def normal(self, *shape, std=1.0):
        return torch.randn(*shape, generator=self.g) * std

    def uniform(self, *shape, lo=0.0, hi=1.0):
        return torch.rand(*shape, generator=self.g) * (hi - lo) + lo


def _conv(sd, g, name, cout, cin, k, bias=True, gain=1.0):
    sd[name + ".weight"] = g.normal(cout, cin, k, std=gain / math.sqrt(cin * k))
    if bias:
        sd[name + ".bias"] = g.normal(cout, std=0.1)


def _wn_conv(sd, g, name, cout, cin, k, gain=1.0, transposed=False):
    """weight-normalised conv as stored by torch.nn.utils.weight_norm (dim 0): weight_g (d0,1,1), weight_v."""
    shape = (cin, cout, k) if transposed else (cout, cin, k)
    fan = cin * k if not transposed else cin * k / max(1, k // 4)
    v = g.normal(*shape, std=1.0 / math.sqrt(fan))
    nrm = v.flatten(1).norm(dim=1).view(-1, 1, 1)
    sd[name + ".weight_g"] = nrm * g.uniform(shape[0], 1, 1, lo=0.7, hi=1.3) * gain
    sd[name + ".weight_v"] = v
    sd[name + ".bias"] = g.normal(cout, std=0.05)


def synth_state_dict(cfg=SYNTH_CFG_40K_V2, seed=1234, phone_dim=768):
    """state_dict of SynthesizerTrnMs768NSFsid(*cfg) after `del net_g.enc_q` (reference src/rvc.py:129-134)."""
    (_, _, inter, hidden, filt, heads, layers, ksize, _, _, rb_k, rb_d, up_r, up_init, up_k, spk, gin, sr) = cfg
    g = _Gen(seed)
    sd = {}
    dk = hidden // heads
    sd["enc_p.emb_phone.weight"] = g.normal(hidden, phone_dim, std=1.0 / math.sqrt(phone_dim))
    sd["enc_p.emb_phone.bias"] = g.normal(hidden, std=0.05)
    sd["enc_p.emb_pitch.weight"] = g.normal(256, hidden, std=0.3)
    for i in range(layers):
        p = "enc_p.encoder.attn_layers.%d." % i
        sd[p + "emb_rel_k"] = g.normal(1, 21, dk, std=dk ** -0.5)
        sd[p + "emb_rel_v"] = g.normal(1, 21, dk, std=dk ** -0.5)
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            _conv(sd, g, p + n, hidden, hidden, 1, gain=1.5 if n in ("conv_q", "conv_k") else 1.0)
        sd["enc_p.encoder.norm_layers_1.%d.gamma" % i] = g.uniform(hidden, lo=0.8, hi=1.2)
        sd["enc_p.encoder.norm_layers_1.%d.beta" % i] = g.normal(hidden, std=0.1)
        _conv(sd, g, "enc_p.encoder.ffn_layers.%d.conv_1" % i, filt, hidden, ksize)
        _conv(sd, g, "enc_p.encoder.ffn_layers.%d.conv_2" % i, hidden, filt, ksize)
        sd["enc_p.encoder.norm_layers_2.%d.gamma" % i] = g.uniform(hidden, lo=0.8, hi=1.2)
        sd["enc_p.encoder.norm_layers_2.%d.beta" % i] = g.normal(hidden, std=0.1)
    _conv(sd, g, "enc_p.proj", 2 * inter, hidden, 1, gain=0.5)
    # decoder (GeneratorNSF)
    sd["dec.m_source.l_linear.weight"] = torch.tensor([[0.9]])
    sd["dec.m_source.l_linear.bias"] = torch.tensor([0.01])
    ch = up_init
    for i, (u, k) in enumerate(zip(up_r, up_k)):
        cin, cout = up_init // (2 ** i), up_init // (2 ** (i + 1))
        _wn_conv(sd, g, "dec.ups.%d" % i, cout, cin, k, transposed=True)
        if i + 1 < len(up_r):
            s = int(np.prod(up_r[i + 1:]))
            _conv(sd, g, "dec.noise_convs.%d" % i, cout, 1, 2 * s)
        else:
            _conv(sd, g, "dec.noise_convs.%d" % i, cout, 1, 1)
        ch = cout
    _conv(sd, g, "dec.conv_pre", up_init, inter, 7)
    j = 0
    for i in range(len(up_r)):
        c = up_init // (2 ** (i + 1))
        for k, _d in zip(rb_k, rb_d):
            for m in range(3):
                _wn_conv(sd, g, "dec.resblocks.%d.convs1.%d" % (j, m), c, c, k, gain=0.8)
                _wn_conv(sd, g, "dec.resblocks.%d.convs2.%d" % (j, m), c, c, k, gain=0.8)
            j += 1
    _conv(sd, g, "dec.conv_post", 1, ch, 7, bias=False, gain=0.7)
    _conv(sd, g, "dec.cond", up_init, gin, 1)
    # flow: 4 x (ResidualCouplingLayer, Flip); WN: 3 layers, kernel 5
    half = inter // 2
    for f in (0, 2, 4, 6):
        p = "flow.flows.%d." % f
        _conv(sd, g, p + "pre", hidden, half, 1)
        for l in range(3):
            _wn_conv(sd, g, p + "enc.in_layers.%d" % l, 2 * hidden, hidden, 5)
            _wn_conv(sd, g, p + "enc.res_skip_layers.%d" % l, 2 * hidden if l < 2 else hidden, hidden, 1)
        _wn_conv(sd, g, p + "enc.cond_layer", 2 * hidden * 3, gin, 1)
        # the reference zero-initialises `post` (modules.py:437-438); random here so the flow is not an identity
        _conv(sd, g, p + "post", half, hidden, 1, gain=0.5)
    sd["emb_g.weight"] = g.normal(spk, gin, std=0.5)
    return {k: v.contiguous().float() for k, v in sd.items()}


def synth_checkpoint(cfg=SYNTH_CFG_40K_V2, seed=1234):
    """The on-disk dict reference src/rvc.py:113-120 expects from torch.load(model.pth)."""
    return {"config": list(cfg), "weight": synth_state_dict(cfg, seed), "f0": 1, "version": "v2",
            "info": "seeded-random", "sr": "40k"}
