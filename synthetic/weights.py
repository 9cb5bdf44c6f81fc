"""Seeded random parameters in the reference's own checkpoint formats.

No trained weights exist offline, so parity and benchmarks use random parameters with the
exact state_dict keys / shapes of the reference constructors (checked against the constructors themselves in
tests/golden/make_golden.py, which loads them with strict key matching).  Used by tests/, bench.py and
__graft_entry__.smoke().
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


def synth_state_dict(cfg=SYNTH_CFG_40K_V2, seed=1234, phone_dim=768, f0=True):
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
    if not f0:  # SynthesizerTrnMs*_nono: no pitch embedding, plain Generator without the NSF source
        sd = {k: v for k, v in sd.items() if not (k.startswith("enc_p.emb_pitch") or k.startswith("dec.noise_convs")
                                                   or k.startswith("dec.m_source"))}
    return {k: v.contiguous().float() for k, v in sd.items()}


def synth_checkpoint(cfg=SYNTH_CFG_40K_V2, seed=1234):
    """The on-disk dict reference src/rvc.py:113-120 expects from torch.load(model.pth)."""
    return {"config": list(cfg), "weight": synth_state_dict(cfg, seed), "f0": 1, "version": "v2",
            "info": "seeded-random", "sr": "40k"}


# ---------------------------------------------------------------------------------------------------
# HuBERT (fairseq 0.12.2 `HubertModel`, hubert_base.pt key names) -- not vendored by the reference:
# architecture restated from fairseq models/hubert/hubert.py + models/wav2vec/wav2vec2.py and cross-checked
# against transformers.HubertModel (same hyper-parameters) in tests/golden/make_golden.py.
# ---------------------------------------------------------------------------------------------------
HUBERT_BASE = dict(conv_dim=512, conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_stride=(5, 2, 2, 2, 2, 2, 2), embed=768,
                   heads=12, ffn=3072, layers=12, pos_k=128, pos_groups=16, final_dim=256)
HUBERT_TINY = dict(conv_dim=32, conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_stride=(5, 2, 2, 2, 2, 2, 2), embed=64,
                   heads=2, ffn=128, layers=2, pos_k=16, pos_groups=4, final_dim=16)


def hubert_state_dict(cfg=HUBERT_BASE, seed=1234):
    g = _Gen(seed)
    sd = {}
    cd, E = cfg["conv_dim"], cfg["embed"]
    cin = 1
    for i, k in enumerate(cfg["conv_kernel"]):
        sd["feature_extractor.conv_layers.%d.0.weight" % i] = g.normal(cd, cin, k, std=math.sqrt(2.0 / (cin * k)))
        cin = cd
    sd["feature_extractor.conv_layers.0.2.weight"] = g.uniform(cd, lo=0.8, hi=1.2)   # GroupNorm(cd, cd) affine
    sd["feature_extractor.conv_layers.0.2.bias"] = g.normal(cd, std=0.1)
    sd["layer_norm.weight"] = g.uniform(cd, lo=0.8, hi=1.2)
    sd["layer_norm.bias"] = g.normal(cd, std=0.1)
    sd["post_extract_proj.weight"] = g.normal(E, cd, std=1.0 / math.sqrt(cd))
    sd["post_extract_proj.bias"] = g.normal(E, std=0.05)
    pk, pg = cfg["pos_k"], cfg["pos_groups"]
    v = g.normal(E, E // pg, pk, std=1.0 / math.sqrt(pk * E // pg))
    sd["encoder.pos_conv.0.weight_v"] = v
    # weight_norm(dim=2): one gain per kernel tap, norm over the other two dims
    sd["encoder.pos_conv.0.weight_g"] = v.transpose(0, 2).flatten(1).norm(dim=1).view(1, 1, pk) * g.uniform(1, 1, pk, lo=0.7, hi=1.3)
    sd["encoder.pos_conv.0.bias"] = g.normal(E, std=0.05)
    sd["encoder.layer_norm.weight"] = g.uniform(E, lo=0.8, hi=1.2)
    sd["encoder.layer_norm.bias"] = g.normal(E, std=0.1)
    for i in range(cfg["layers"]):
        p = "encoder.layers.%d." % i
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + "self_attn.%s.weight" % n] = g.normal(E, E, std=(1.6 if n in ("q_proj", "k_proj") else 1.0) / math.sqrt(E))
            sd[p + "self_attn.%s.bias" % n] = g.normal(E, std=0.05)
        sd[p + "self_attn_layer_norm.weight"] = g.uniform(E, lo=0.8, hi=1.2)
        sd[p + "self_attn_layer_norm.bias"] = g.normal(E, std=0.1)
        sd[p + "fc1.weight"] = g.normal(cfg["ffn"], E, std=1.0 / math.sqrt(E))
        sd[p + "fc1.bias"] = g.normal(cfg["ffn"], std=0.05)
        sd[p + "fc2.weight"] = g.normal(E, cfg["ffn"], std=1.0 / math.sqrt(cfg["ffn"]))
        sd[p + "fc2.bias"] = g.normal(E, std=0.05)
        sd[p + "final_layer_norm.weight"] = g.uniform(E, lo=0.8, hi=1.2)
        sd[p + "final_layer_norm.bias"] = g.normal(E, std=0.1)
    sd["final_proj.weight"] = g.normal(cfg["final_dim"], E, std=1.0 / math.sqrt(E))
    sd["final_proj.bias"] = g.normal(cfg["final_dim"], std=0.05)
    return {k: v.contiguous().float() for k, v in sd.items()}


# ---------------------------------------------------------------------------------------------------
# RMVPE E2E (reference src/rmvpe.py:221-258): DeepUnet + BiGRU(384 -> 256) + Linear(512 -> 360)
# ---------------------------------------------------------------------------------------------------
RMVPE_FULL = dict(n_blocks=4, en_de_layers=5, inter_layers=4, en_out_channels=16)
RMVPE_TINY = dict(n_blocks=1, en_de_layers=5, inter_layers=1, en_out_channels=2)


def _bn(sd, g, name, c):
    sd[name + ".weight"] = g.uniform(c, lo=0.8, hi=1.2)
    sd[name + ".bias"] = g.normal(c, std=0.05)
    sd[name + ".running_mean"] = g.normal(c, std=0.05)
    sd[name + ".running_var"] = g.uniform(c, lo=0.8, hi=1.25)
    sd[name + ".num_batches_tracked"] = torch.tensor(1)


def _conv_block_res(sd, g, name, cin, cout, gain):
    """ConvBlockRes (rmvpe.py:23-58).  The residual path gets a small gain so that ~60 stacked blocks with
    un-calibrated BatchNorm statistics keep O(1) activations (SURVEY 7 'hard parts')."""
    sd[name + ".conv.0.weight"] = g.normal(cout, cin, 3, 3, std=math.sqrt(2.0 / (cin * 9)))
    _bn(sd, g, name + ".conv.1", cout)
    sd[name + ".conv.3.weight"] = g.normal(cout, cout, 3, 3, std=gain * math.sqrt(2.0 / (cout * 9)))
    _bn(sd, g, name + ".conv.4", cout)
    if cin != cout:
        sd[name + ".shortcut.weight"] = g.normal(cout, cin, 1, 1, std=1.0 / math.sqrt(cin))
        sd[name + ".shortcut.bias"] = g.normal(cout, std=0.05)


def rmvpe_state_dict(cfg=RMVPE_FULL, seed=1234):
    g = _Gen(seed)
    sd = {}
    nb, nl, ni, c0 = cfg["n_blocks"], cfg["en_de_layers"], cfg["inter_layers"], cfg["en_out_channels"]
    gain = 0.35
    _bn(sd, g, "unet.encoder.bn", 1)
    cin, cout = 1, c0
    for i in range(nl):
        for b in range(nb):
            _conv_block_res(sd, g, "unet.encoder.layers.%d.conv.%d" % (i, b), cin if b == 0 else cout, cout, gain)
        cin, cout = cout, cout * 2
    # after the loop: cin = c0 * 2^(nl-1) (deepest encoder width), cout = 2 * cin
    for i in range(ni):
        for b in range(nb):
            _conv_block_res(sd, g, "unet.intermediate.layers.%d.conv.%d" % (i, b), cin if (i == 0 and b == 0) else cout, cout, gain)
    dc = cout
    for i in range(nl):
        oc = dc // 2
        sd["unet.decoder.layers.%d.conv1.0.weight" % i] = g.normal(dc, oc, 3, 3, std=math.sqrt(2.0 / (dc * 9 / 4)))
        _bn(sd, g, "unet.decoder.layers.%d.conv1.1" % i, oc)
        for b in range(nb):
            _conv_block_res(sd, g, "unet.decoder.layers.%d.conv2.%d" % (i, b), oc * 2 if b == 0 else oc, oc, gain)
        dc = oc
    sd["cnn.weight"] = g.normal(3, c0, 3, 3, std=1.0 / math.sqrt(c0 * 9))
    sd["cnn.bias"] = g.normal(3, std=0.05)
    hid = 256
    for suf in ("", "_reverse"):
        sd["fc.0.gru.weight_ih_l0" + suf] = g.uniform(3 * hid, 384, lo=-1, hi=1) / math.sqrt(hid) * 2.0
        sd["fc.0.gru.weight_hh_l0" + suf] = g.uniform(3 * hid, hid, lo=-1, hi=1) / math.sqrt(hid)
        sd["fc.0.gru.bias_ih_l0" + suf] = g.uniform(3 * hid, lo=-1, hi=1) / math.sqrt(hid)
        sd["fc.0.gru.bias_hh_l0" + suf] = g.uniform(3 * hid, lo=-1, hi=1) / math.sqrt(hid)
    sd["fc.1.weight"] = g.normal(360, 512, std=3.0 / math.sqrt(512))
    sd["fc.1.bias"] = g.normal(360, std=0.5) - 2.0
    return {k: (v.contiguous().float() if v.is_floating_point() else v) for k, v in sd.items()}


# ---------------------------------------------------------------------------------------------------
# MDX-Net (TFC-TDF U-Net, "ConvTDFNet" of kuielab/mdx-net as shipped in the UVR-MDX-NET .onnx files).
# The reference only downloads the graphs (src/download_models.py:4,21) and runs them with onnxruntime
# (src/mdx.py:74-77): architecture recalled from the published kuielab code, PARITY UNPINNED (no .onnx offline).
# ---------------------------------------------------------------------------------------------------
MDX_VOC_FT = dict(dim_c=4, g=48, n=5, l=3, k=3, bn=8, dim_f=3072, dim_t=256, n_fft=7680)   # model_data.json Voc_FT class
MDX_TINY = dict(dim_c=4, g=8, n=2, l=2, k=3, bn=4, dim_f=64, dim_t=16, n_fft=160)
# The other two separations of main.py's chain (src/main.py:185,188: UVR_MDXNET_KARA_2, Reverb_HQ_By_FoxJoy).  Which hash of
# model_data.json belongs to which file cannot be derived offline; these are two of the (dim_f, dim_t, n_fft) classes that json
# lists -- (2048, 2^8, 5120) and (3072, 2^9, 6144) -- with the Voc_FT-class channel plan [MEM-EST].
MDX_KARA2 = dict(dim_c=4, g=48, n=5, l=3, k=3, bn=8, dim_f=2048, dim_t=256, n_fft=5120)
MDX_REVERB_HQ = dict(dim_c=4, g=48, n=5, l=3, k=3, bn=8, dim_f=3072, dim_t=512, n_fft=6144)


def _bn2(sd, g, name, c):
    _bn(sd, g, name, c)


def _tfc_tdf(sd, g, name, c, l, f, k, bn):
    for j in range(l):
        sd["%s.tfc.H.%d.0.weight" % (name, j)] = g.normal(c, c, k, k, std=math.sqrt(2.0 / (c * k * k)))
        sd["%s.tfc.H.%d.0.bias" % (name, j)] = g.normal(c, std=0.05)
        _bn2(sd, g, "%s.tfc.H.%d.1" % (name, j), c)
    sd[name + ".tdf.0.weight"] = g.normal(f // bn, f, std=math.sqrt(2.0 / f))
    sd[name + ".tdf.0.bias"] = g.normal(f // bn, std=0.05)
    _bn2(sd, g, name + ".tdf.1", c)
    sd[name + ".tdf.3.weight"] = g.normal(f, f // bn, std=0.5 * math.sqrt(2.0 / (f // bn)))
    sd[name + ".tdf.3.bias"] = g.normal(f, std=0.05)
    _bn2(sd, g, name + ".tdf.4", c)


def mdx_state_dict(cfg=MDX_VOC_FT, seed=1234):
    g = _Gen(seed)
    sd = {}
    gg, n, l, k, bn, f = cfg["g"], cfg["n"], cfg["l"], cfg["k"], cfg["bn"], cfg["dim_f"]
    sd["first_conv.0.weight"] = g.normal(gg, cfg["dim_c"], 1, 1, std=1.0)
    sd["first_conv.0.bias"] = g.normal(gg, std=0.05)
    _bn2(sd, g, "first_conv.1", gg)
    c = gg
    for i in range(n):
        _tfc_tdf(sd, g, "ds_dense.%d" % i, c, l, f, k, bn)
        sd["ds.%d.0.weight" % i] = g.normal(c + gg, c, 2, 2, std=math.sqrt(2.0 / (c * 4)))
        sd["ds.%d.0.bias" % i] = g.normal(c + gg, std=0.05)
        _bn2(sd, g, "ds.%d.1" % i, c + gg)
        f //= 2
        c += gg
    _tfc_tdf(sd, g, "mid_dense", c, l, f, k, bn)
    for i in range(n):
        sd["us.%d.0.weight" % i] = g.normal(c, c - gg, 2, 2, std=math.sqrt(2.0 / c))
        sd["us.%d.0.bias" % i] = g.normal(c - gg, std=0.05)
        _bn2(sd, g, "us.%d.1" % i, c - gg)
        f *= 2
        c -= gg
        _tfc_tdf(sd, g, "us_dense.%d" % i, c, l, f, k, bn)
    sd["final_conv.0.weight"] = g.normal(cfg["dim_c"], c, 1, 1, std=0.5 / math.sqrt(c))
    sd["final_conv.0.bias"] = g.normal(cfg["dim_c"], std=0.01)
    return {kk: (v.contiguous().float() if v.is_floating_point() else v) for kk, v in sd.items()}


# ---------------------------------------------------------------------------------------------------
# small end-to-end model set (CPU-side pipeline tests): 768-d HuBERT so the reference's TextEncoder768 accepts it,
# and a synthesizer that upsamples only 16x (tgt_sr 1600) so the emulator finishes in seconds
# ---------------------------------------------------------------------------------------------------
HUBERT_SMALL768 = dict(conv_dim=32, conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_stride=(5, 2, 2, 2, 2, 2, 2), embed=768,
                       heads=12, ffn=128, layers=1, pos_k=128, pos_groups=16, final_dim=256)
SYNTH_CFG_MICRO = [1025, 32, 64, 64, 128, 2, 1, 3, 0, "1", [3, 7, 11], [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                   [2, 2, 2, 2], 64, [4, 4, 4, 4], 4, 32, 1600]


def small_model_set(seed=1234):
    return dict(hubert_cfg=HUBERT_SMALL768, hubert_sd=hubert_state_dict(HUBERT_SMALL768, seed),
                rmvpe_sd=rmvpe_state_dict(RMVPE_TINY, seed + 1),
                synth_cfg=SYNTH_CFG_MICRO, synth_sd=synth_state_dict(SYNTH_CFG_MICRO, seed + 2))


def full_model_set(seed=1234):
    return dict(hubert_cfg=HUBERT_BASE, hubert_sd=hubert_state_dict(HUBERT_BASE, seed),
                rmvpe_sd=rmvpe_state_dict(RMVPE_FULL, seed + 1),
                synth_cfg=SYNTH_CFG_40K_V2, synth_sd=synth_state_dict(SYNTH_CFG_40K_V2, seed + 2))


# ---------------------------------------------------------------------------------------------------
# CREPE (torchcrepe 0.0.20 `Crepe`, pip dependency of the reference, requirements.txt:19; not vendored):
# 6 x [pad, Conv2d (k,1), ReLU, BatchNorm2d(eps=0.0010000000474974513), MaxPool (2,1)] + Linear(in_features, 360)
# ---------------------------------------------------------------------------------------------------
CREPE_FULL = dict(out_channels=(1024, 128, 128, 128, 256, 512), kernels=(512, 64, 64, 64, 64, 64), stride0=4)
CREPE_TINY = dict(out_channels=(128, 16, 16, 16, 32, 64), kernels=(512, 64, 64, 64, 64, 64), stride0=4)
CREPE_MICRO = dict(out_channels=(16, 8, 8, 8, 8, 8), kernels=(512, 64, 64, 64, 64, 64), stride0=4)   # CPU-side tests


def crepe_state_dict(cfg=CREPE_FULL, seed=1234):
    g = _Gen(seed)
    sd = {}
    cin = 1
    for i, (co, k) in enumerate(zip(cfg["out_channels"], cfg["kernels"])):
        n = "conv%d" % (i + 1)
        sd[n + ".weight"] = g.normal(co, cin, k, 1, std=math.sqrt(2.0 / (cin * k)))
        sd[n + ".bias"] = g.normal(co, std=0.05)
        _bn(sd, g, n + "_BN", co)
        cin = co
    in_features = 4 * cfg["out_channels"][-1]
    sd["classifier.weight"] = g.normal(360, in_features, std=4.0 / math.sqrt(in_features))
    sd["classifier.bias"] = g.normal(360, std=0.5)
    return {kk: (v.contiguous().float() if v.is_floating_point() else v) for kk, v in sd.items()}
