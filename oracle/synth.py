"""TEST INFRASTRUCTURE (oracle): CPU fp32 restatement of SynthesizerTrnMs768NSFsid.infer
(reference src/infer_pack/models.py:745-751) as plain functions over the reference's own state_dict.

Pinned against the reference itself: tests/golden/make_golden.py imports /root/reference/src/infer_pack in the
build container, runs the reference module and this restatement on the same seeded parameters / noise and
stores the reference outputs as fixtures (tests/golden/synth_*.npz); tests/test_oracle_golden.py replays them.
Never imported by the product path.
"""
import math

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # modules.py:17


def wn_weight(sd, name):
    """torch.nn.utils.weight_norm, dim=0: w = g * v / ||v|| with the norm over all other dims."""
    if name + ".weight" in sd:
        return sd[name + ".weight"]
    v, g = sd[name + ".weight_v"], sd[name + ".weight_g"]
    return v * (g / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1))))


def layer_norm_c(x, gamma, beta, eps=1e-5):
    """modules.LayerNorm (modules.py:29-32): normalise over the channel axis of (B, C, T)."""
    return F.layer_norm(x.transpose(1, 2), (x.shape[1],), gamma, beta, eps).transpose(1, 2)


def rel_attention(sd, p, x, n_heads, window=10):
    """attentions.MultiHeadAttention.forward with window_size=10, heads_share=True (attentions.py:216-275),
    written in the banded closed form verified in SURVEY appendix B.3."""
    b, c, t = x.shape
    dk = c // n_heads
    q = F.conv1d(x, sd[p + "conv_q.weight"], sd[p + "conv_q.bias"])
    k = F.conv1d(x, sd[p + "conv_k.weight"], sd[p + "conv_k.bias"])
    v = F.conv1d(x, sd[p + "conv_v.weight"], sd[p + "conv_v.bias"])
    q = q.view(b, n_heads, dk, t).transpose(2, 3) / math.sqrt(dk)
    k = k.view(b, n_heads, dk, t).transpose(2, 3)
    v = v.view(b, n_heads, dk, t).transpose(2, 3)
    scores = q @ k.transpose(2, 3)                           # (b, h, t, t)
    ek, ev = sd[p + "emb_rel_k"][0], sd[p + "emb_rel_v"][0]   # (2w+1, dk)
    rel = q @ ek.t()                                         # (b, h, t, 2w+1): q_i . E^k_m
    i = torch.arange(t).view(t, 1)
    j = torch.arange(t).view(1, t)
    m = j - i + window
    band = (m >= 0) & (m <= 2 * window)
    mc = m.clamp(0, 2 * window)
    scores = scores + torch.where(band, rel.gather(3, mc.expand(b, n_heads, t, t)), torch.zeros(()))
    pa = F.softmax(scores, dim=-1)
    out = pa @ v
    # relative values: sum_m P[i, i+m-w] E^v_m
    pband = torch.zeros(b, n_heads, t, 2 * window + 1)
    for mm in range(2 * window + 1):
        jj = torch.arange(t) + mm - window
        ok = (jj >= 0) & (jj < t)
        pband[:, :, ok, mm] = pa[:, :, torch.arange(t)[ok], jj[ok]]
    out = out + pband @ ev
    out = out.transpose(2, 3).reshape(b, c, t)
    return F.conv1d(out, sd[p + "conv_o.weight"], sd[p + "conv_o.bias"])


def text_encoder(sd, phone, pitch, n_heads, n_layers, ksize):
    """TextEncoder768.forward (models.py:93-108) + attentions.Encoder.forward (attentions.py:61-73);
    single full-length sequence => every mask is all-ones."""
    hidden = sd["enc_p.emb_phone.weight"].shape[0]
    x = F.linear(phone, sd["enc_p.emb_phone.weight"], sd["enc_p.emb_phone.bias"])
    if pitch is not None:
        x = x + sd["enc_p.emb_pitch.weight"][pitch]
    x = F.leaky_relu(x * math.sqrt(hidden), 0.1).transpose(1, 2)
    pad_l, pad_r = (ksize - 1) // 2, ksize // 2
    for i in range(n_layers):
        y = rel_attention(sd, "enc_p.encoder.attn_layers.%d." % i, x, n_heads)
        x = layer_norm_c(x + y, sd["enc_p.encoder.norm_layers_1.%d.gamma" % i], sd["enc_p.encoder.norm_layers_1.%d.beta" % i])
        f = "enc_p.encoder.ffn_layers.%d." % i
        y = F.conv1d(F.pad(x, (pad_l, pad_r)), sd[f + "conv_1.weight"], sd[f + "conv_1.bias"])
        y = F.conv1d(F.pad(torch.relu(y), (pad_l, pad_r)), sd[f + "conv_2.weight"], sd[f + "conv_2.bias"])
        x = layer_norm_c(x + y, sd["enc_p.encoder.norm_layers_2.%d.gamma" % i], sd["enc_p.encoder.norm_layers_2.%d.beta" % i])
    stats = F.conv1d(x, sd["enc_p.proj.weight"], sd["enc_p.proj.bias"])
    m, logs = stats.chunk(2, dim=1)
    return m, logs


def wavenet(sd, p, x, g, hidden, n_layers=3, ksize=5):
    """modules.WN.forward (modules.py:188-213), dilation_rate = 1."""
    out = torch.zeros_like(x)
    cond = F.conv1d(g, wn_weight(sd, p + "cond_layer"), sd[p + "cond_layer.bias"])
    for l in range(n_layers):
        a = F.conv1d(x, wn_weight(sd, p + "in_layers.%d" % l), sd[p + "in_layers.%d.bias" % l], padding=(ksize - 1) // 2)
        a = a + cond[:, 2 * hidden * l: 2 * hidden * (l + 1)]
        acts = torch.tanh(a[:, :hidden]) * torch.sigmoid(a[:, hidden:])
        rs = F.conv1d(acts, wn_weight(sd, p + "res_skip_layers.%d" % l), sd[p + "res_skip_layers.%d.bias" % l])
        if l < n_layers - 1:
            x = x + rs[:, :hidden]
            out = out + rs[:, hidden:]
        else:
            out = out + rs
    return out


def flow_reverse(sd, z, g, hidden):
    """ResidualCouplingBlock.forward(reverse=True) (models.py:150-153): Flip, RCL3, Flip, RCL2, ... with
    mean-only coupling x1 <- x1 - m (modules.py:440-459)."""
    half = z.shape[1] // 2
    for f in (6, 4, 2, 0):
        z = torch.flip(z, [1])
        p = "flow.flows.%d." % f
        x0, x1 = z[:, :half], z[:, half:]
        h = F.conv1d(x0, sd[p + "pre.weight"], sd[p + "pre.bias"])
        h = wavenet(sd, p + "enc.", h, g, hidden)
        m = F.conv1d(h, sd[p + "post.weight"], sd[p + "post.bias"])
        z = torch.cat([x0, x1 - m], 1)
    return z


def sine_source(sd, f0, upp, sr, noise):
    """SineGen.forward + SourceModuleHnNSF.forward (models.py:320-370, 414-419), harmonic_num=0, in the float64
    closed form of SURVEY appendix B.5 (phase = running sum of (f0/sr mod 1), wrap-free)."""
    rad = torch.fmod(f0 / sr, 1.0)                                       # (1, T) fp32 like the reference
    radu = rad.double().repeat_interleave(upp, dim=1)
    phase = torch.cumsum(radu, dim=1)
    sine = (torch.sin(2 * math.pi * torch.frac(phase)) * 0.1).float()
    uv = (f0 > 0).float().repeat_interleave(upp, dim=1)
    namp = uv * 0.003 + (1 - uv) * 0.1 / 3
    waves = sine * uv + namp * noise
    w, b = sd["dec.m_source.l_linear.weight"], sd["dec.m_source.l_linear.bias"]
    return torch.tanh(waves * w[0, 0] + b[0]).unsqueeze(1)              # (1, 1, T*upp)


def generator_nsf(sd, cfg, x, f0, g, noise):
    """GeneratorNSF.forward (models.py:494-516)."""
    rb_k, rb_d, up_r, up_init, up_k = cfg[10], cfg[11], cfg[12], cfg[13], cfg[14]
    sr = cfg[17]
    upp = 1
    for u in up_r:
        upp *= u
    har = sine_source(sd, f0, upp, float(sr), noise) if f0 is not None else None  # _nono models: plain Generator (:253-272)
    x = F.conv1d(x, sd["dec.conv_pre.weight"], sd["dec.conv_pre.bias"], padding=3)
    x = x + F.conv1d(g, sd["dec.cond.weight"], sd["dec.cond.bias"])
    nk = len(rb_k)
    for i, (u, k) in enumerate(zip(up_r, up_k)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, wn_weight(sd, "dec.ups.%d" % i), sd["dec.ups.%d.bias" % i], stride=u, padding=(k - u) // 2)
        if har is not None:
            if i + 1 < len(up_r):
                s = 1
                for uu in up_r[i + 1:]:
                    s *= uu
                xs = F.conv1d(har, sd["dec.noise_convs.%d.weight" % i], sd["dec.noise_convs.%d.bias" % i], stride=s, padding=s // 2)
            else:
                xs = F.conv1d(har, sd["dec.noise_convs.%d.weight" % i], sd["dec.noise_convs.%d.bias" % i])
            x = x + xs
        acc = None
        for j in range(nk):
            r = "dec.resblocks.%d." % (i * nk + j)
            y = x
            for m, d in enumerate(rb_d[j]):
                kk = rb_k[j]
                t = F.conv1d(F.leaky_relu(y, LRELU_SLOPE), wn_weight(sd, r + "convs1.%d" % m), sd[r + "convs1.%d.bias" % m],
                             dilation=d, padding=(kk * d - d) // 2)
                t = F.conv1d(F.leaky_relu(t, LRELU_SLOPE), wn_weight(sd, r + "convs2.%d" % m), sd[r + "convs2.%d.bias" % m],
                             padding=(kk - 1) // 2)
                y = t + y
            acc = y if acc is None else acc + y
        x = acc / nk
    x = F.leaky_relu(x)  # default slope 0.01 (models.py:513)
    x = F.conv1d(x, sd["dec.conv_post.weight"], None, padding=3)
    return torch.tanh(x)


def synth_infer(sd, cfg, phone, pitch, nsff0, sid, noise_z, noise_src):
    """SynthesizerTrnMs768NSFsid.infer (models.py:745-751) with the two torch.randn_like draws
    (:748 and SineGen :368) passed in: noise_z (1, inter, T), noise_src (1, T*upp)."""
    inter, hidden, n_heads, n_layers, ksize = cfg[2], cfg[3], cfg[5], cfg[6], cfg[7]
    g = sd["emb_g.weight"][sid].unsqueeze(-1)                 # (1, gin, 1)
    m_p, logs_p = text_encoder(sd, phone, pitch, n_heads, n_layers, ksize)
    z_p = m_p + torch.exp(logs_p) * noise_z * 0.66666
    z = flow_reverse(sd, z_p, g, hidden)
    o = generator_nsf(sd, cfg, z, nsff0, g, noise_src)
    return o, (z, z_p, m_p, logs_p)
