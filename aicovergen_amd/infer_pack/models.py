"""RVC synthesizers on the gfx950 kernels, behind the reference's class names and call surface
(reference src/infer_pack/models.py: SynthesizerTrnMs{256,768}NSFsid{,_nono}; built by src/rvc.py:122-140 and
driven by VC.vc at src/vc_infer_pipeline.py:454-465 through `.infer`).

Host code here only lays out weights (weight-norm folded once at load: w = g * v / ||v||, exactly the tensor
torch recomputes every forward in the reference) and sequences kernel launches; every arithmetic op of the
forward pass is a kernel of libaicg_hip.so (aicovergen_amd/ops.py).  Activations are fp32 channel-major
(1, C, T), the same layout the reference uses.
"""
import contextlib
import math
import os
from collections import namedtuple

import numpy as np
import torch
from .. import _env
import torch.nn.functional as F

from .. import ops

_Incompatible = namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys"])
LRELU_SLOPE = 0.1  # reference modules.py:17

sr2sr = {"32k": 32000, "40k": 40000, "48k": 48000}


def _fold_weight_norm(sd, name):
    if name + ".weight" in sd:
        return sd[name + ".weight"].float()
    v, g = sd[name + ".weight_v"].float(), sd[name + ".weight_g"].float()
    return v * (g / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1))))


class _SynthesizerBase:
    phone_dim = 768
    use_f0 = True

    def __init__(self, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads,
                 n_layers, kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                 upsample_rates, upsample_initial_channel, upsample_kernel_sizes, spk_embed_dim, gin_channels, sr=None,
                 **kwargs):
        if isinstance(sr, str):
            sr = sr2sr[sr]
        self.spec_channels, self.segment_size = spec_channels, segment_size
        self.inter_channels, self.hidden_channels, self.filter_channels = inter_channels, hidden_channels, filter_channels
        self.n_heads, self.n_layers, self.kernel_size, self.p_dropout = n_heads, n_layers, kernel_size, p_dropout
        self.resblock = str(resblock)
        self.resblock_kernel_sizes = list(resblock_kernel_sizes)
        self.resblock_dilation_sizes = [list(d) for d in resblock_dilation_sizes]
        self.upsample_rates = list(upsample_rates)
        self.upsample_initial_channel = upsample_initial_channel
        self.upsample_kernel_sizes = list(upsample_kernel_sizes)
        self.spk_embed_dim, self.gin_channels = spk_embed_dim, gin_channels
        self.sr = sr
        self.upp = int(np.prod(self.upsample_rates))
        self.is_half = bool(kwargs.get("is_half", False))
        self.enc_q = None  # rvc.get_vc does `del net_g.enc_q` (posterior encoder is training-only)
        self.device = torch.device("cpu")
        self._sd = None
        self._p = None
        assert kernel_size % 2 == 1, "FFN same-padding is symmetric only for odd kernels"
        print("gin_channels:", gin_channels, "self.spk_embed_dim:", self.spk_embed_dim)

    # ---- torch.nn.Module-like surface used by rvc.get_vc -------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        self._sd = {k: v.detach().to("cpu") for k, v in state_dict.items() if not k.startswith("enc_q.")}
        self._p = None
        unexpected = [k for k in state_dict if k.startswith("enc_q.")] if strict else []
        return _Incompatible([], unexpected)

    def eval(self):
        return self

    def to(self, device):
        self.device = torch.device(device)
        self._p = None
        return self

    def half(self):
        # The reference runs fp16 on GPU (src/rvc.py:137-138).  Default: fp32 kernels (>= that precision); with AICG_HALF=1 the layers on
        # the LDS-DMA staged kernels take fp16 operands on the matrix pipe (ops.mark_half; fp32 activations and accumulation)
        self._half = ops.half_requested()
        if self._p is not None:
            ops.mark_half(self._p, self._half)
        return self

    def float(self):
        self._half = False
        if self._p is not None:
            ops.mark_half(self._p, False)
        return self

    def remove_weight_norm(self):
        return None  # already folded at load

    # ---- weight layout ------------------------------------------------------------------------------------
    def _prepare(self):
        if self._p is not None:
            return self._p
        assert self._sd is not None, "load_state_dict() first"
        sd, dev = self._sd, self.device
        P = {}
        H, C = self.n_heads, self.hidden_channels
        dk = C // H
        P["emb_phone"] = ops.PackedConv(sd["enc_p.emb_phone.weight"].float(), sd["enc_p.emb_phone.bias"], device=dev)
        if self.use_f0:
            P["emb_pitch"] = sd["enc_p.emb_pitch.weight"].float().to(dev)
        layers = []
        for i in range(self.n_layers):
            a = "enc_p.encoder.attn_layers.%d." % i
            L = {}
            L["q"] = ops.PackedConv(sd[a + "conv_q.weight"].float(), sd[a + "conv_q.bias"], device=dev)
            L["kv"] = ops.PackedConv(torch.cat([sd[a + "conv_k.weight"], sd[a + "conv_v.weight"]], 0).float(),
                                     torch.cat([sd[a + "conv_k.bias"], sd[a + "conv_v.bias"]], 0), device=dev)
            ek = sd[a + "emb_rel_k"].float()  # (1 or H, 2w+1, dk)
            ev = sd[a + "emb_rel_v"].float()
            assert ek.shape[0] == 1, "heads_share=True is the only configuration the reference instantiates"
            self.window = (ek.shape[1] - 1) // 2
            L["relk"] = ops.PackedConv(ek[0].repeat(H, 1).unsqueeze(-1).contiguous(), None, groups=H, device=dev)
            L["relv"] = ev[0].contiguous().to(dev)
            L["o"] = ops.PackedConv(sd[a + "conv_o.weight"].float(), sd[a + "conv_o.bias"], device=dev)
            for n in (1, 2):
                L["g%d" % n] = sd["enc_p.encoder.norm_layers_%d.%d.gamma" % (n, i)].float().to(dev)
                L["b%d" % n] = sd["enc_p.encoder.norm_layers_%d.%d.beta" % (n, i)].float().to(dev)
            f = "enc_p.encoder.ffn_layers.%d." % i
            pad = (self.kernel_size - 1) // 2
            L["ffn1"] = ops.PackedConv(sd[f + "conv_1.weight"].float(), sd[f + "conv_1.bias"], padding=pad, device=dev)
            L["ffn2"] = ops.PackedConv(sd[f + "conv_2.weight"].float(), sd[f + "conv_2.bias"], padding=pad, device=dev)
            layers.append(L)
        P["enc_layers"] = layers
        P["proj"] = ops.PackedConv(sd["enc_p.proj.weight"].float(), sd["enc_p.proj.bias"], device=dev)
        P["emb_g"] = sd["emb_g.weight"].float().to(dev)
        # flow (reverse order is applied at run time)
        flows = {}
        for fidx in (0, 2, 4, 6):
            p = "flow.flows.%d." % fidx
            Fl = {"pre": ops.PackedConv(sd[p + "pre.weight"].float(), sd[p + "pre.bias"], device=dev)}
            n_wn = 0
            while (p + "enc.in_layers.%d.weight_v" % n_wn) in sd or (p + "enc.in_layers.%d.weight" % n_wn) in sd:
                n_wn += 1
            Fl["n_wn"] = n_wn
            in_bias = []
            for l in range(n_wn):
                w = _fold_weight_norm(sd, p + "enc.in_layers.%d" % l)
                Fl["in%d" % l] = ops.PackedConv(w, None, padding=(w.shape[2] - 1) // 2, device=dev)
                in_bias.append(sd[p + "enc.in_layers.%d.bias" % l].float())
                rs_w = _fold_weight_norm(sd, p + "enc.res_skip_layers.%d" % l)
                rs_b = sd[p + "enc.res_skip_layers.%d.bias" % l].float()
                if l < n_wn - 1:
                    Fl["res%d" % l] = ops.PackedConv(rs_w[:C], rs_b[:C], device=dev)
                    Fl["skip%d" % l] = ops.PackedConv(rs_w[C:], rs_b[C:], device=dev)
                else:
                    Fl["skip%d" % l] = ops.PackedConv(rs_w, rs_b, device=dev)
            # cond_layer(g) + in_layer biases -> the per-call biases of all WN in_layers in one GEMV
            Fl["cond"] = ops.PackedConv(_fold_weight_norm(sd, p + "enc.cond_layer"), sd[p + "enc.cond_layer.bias"], device=dev)
            Fl["in_bias"] = torch.cat(in_bias).view(1, -1, 1).contiguous().to(dev)
            # x1 <- x1 - post(h): negated weights so that the conv epilogue's residual add does the subtraction
            Fl["post_neg"] = ops.PackedConv(-sd[p + "post.weight"].float(), -sd[p + "post.bias"].float(), device=dev)
            flows[fidx] = Fl
        P["flows"] = flows
        # decoder
        up_init = self.upsample_initial_channel
        P["conv_pre"] = ops.PackedConv(sd["dec.conv_pre.weight"].float(), None, padding=3, device=dev)
        P["conv_pre_bias"] = sd["dec.conv_pre.bias"].float().view(1, -1, 1).contiguous().to(dev)
        P["cond"] = ops.PackedConv(sd["dec.cond.weight"].float(), sd["dec.cond.bias"], device=dev)
        ups, noise = [], []
        for i, (u, k) in enumerate(zip(self.upsample_rates, self.upsample_kernel_sizes)):
            ups.append(ops.PackedConvTranspose(_fold_weight_norm(sd, "dec.ups.%d" % i), sd["dec.ups.%d.bias" % i], stride=u,
                                               padding=(k - u) // 2, device=dev))
            if self.use_f0:
                w, b = sd["dec.noise_convs.%d.weight" % i].float(), sd["dec.noise_convs.%d.bias" % i]
                cc = w.shape[0]
                if i + 1 < len(self.upsample_rates):
                    s = int(np.prod(self.upsample_rates[i + 1:]))
                    assert w.shape[2] == 2 * s
                    # Conv1d(1 -> C, k = 2s, stride s, pad s/2) == Conv1d(s -> C, k = 2) over the s-phase
                    # decomposition X[ph][q] = xpad[q*s + ph] of the padded source (pure re-indexing)
                    w2 = w.view(cc, 2, s).permute(0, 2, 1).contiguous()
                    noise.append((s, ops.PackedConv(w2, b, device=dev)))
                else:
                    noise.append((1, ops.PackedConv(w, b, device=dev)))
        P["ups"], P["noise"] = ups, noise
        rbs = []
        nres = len(self.upsample_rates) * len(self.resblock_kernel_sizes)
        for j in range(nres):
            k = self.resblock_kernel_sizes[j % len(self.resblock_kernel_sizes)]
            dil = self.resblock_dilation_sizes[j % len(self.resblock_kernel_sizes)]
            r = "dec.resblocks.%d." % j
            convs = []
            if self.resblock == "1":
                for m, d in enumerate(dil):
                    c1 = ops.PackedConv(_fold_weight_norm(sd, r + "convs1.%d" % m), sd[r + "convs1.%d.bias" % m], dilation=d,
                                        padding=(k * d - d) // 2, device=dev)
                    c2 = ops.PackedConv(_fold_weight_norm(sd, r + "convs2.%d" % m), sd[r + "convs2.%d.bias" % m],
                                        padding=(k - 1) // 2, device=dev)
                    convs.append((c1, c2))
            else:  # ResBlock2 (modules.py:321-359): x = conv_d(lrelu(x)) + x
                for m, d in enumerate(dil):
                    convs.append((ops.PackedConv(_fold_weight_norm(sd, r + "convs.%d" % m), sd[r + "convs.%d.bias" % m],
                                                 dilation=d, padding=(k * d - d) // 2, device=dev), None))
            rbs.append(convs)
        P["resblocks"] = rbs
        P["conv_post"] = ops.PackedConv(sd["dec.conv_post.weight"].float(), None, padding=3, device=dev)
        if self.use_f0:
            P["lin_w"] = float(sd["dec.m_source.l_linear.weight"].reshape(-1)[0])
            P["lin_b"] = float(sd["dec.m_source.l_linear.bias"].reshape(-1)[0])
        if getattr(self, "_half", False):
            ops.mark_half(P, True)
        self._p = P
        return P

    # ---- forward pieces -----------------------------------------------------------------------------------
    def _enc_p(self, P, phone_ct, pitch):
        """TextEncoder{256,768}.forward + attentions.Encoder.forward (models.py:93-108, attentions.py:61-73)."""
        C, H = self.hidden_channels, self.n_heads
        dk = C // H
        T = phone_ct.shape[2]
        res = None
        if pitch is not None:
            res = P["emb_pitch"][pitch[0]].t().contiguous().unsqueeze(0)  # embedding gather (1, C, T)
        # (emb_phone(phone) + emb_pitch(pitch)) * sqrt(C) -> LeakyReLU(0.1); lrelu is positively homogeneous
        x = ops.conv(phone_ct, P["emb_phone"], res=res, res_before_act=True, act=ops.ACT_LRELU, act_slope=0.1,
                     out_scale=math.sqrt(C))
        for L in P["enc_layers"]:
            q = ops.conv(x, L["q"], out_scale=1.0 / math.sqrt(dk))  # query / sqrt(k_channels) (attentions.py:233)
            kv = ops.conv(x, L["kv"])
            relk = ops.conv(q, L["relk"])  # (1, H*(2w+1), T): q_i . E^k_m per head
            o = ops.attention(q[0], kv[0, :C], kv[0, C:], H, relk=relk[0].view(H, 2 * self.window + 1, T),
                              relv_emb=L["relv"], window=self.window)
            y = ops.conv(o.unsqueeze(0), L["o"])
            x = ops.layernorm_ct(x, L["g1"], L["b1"], res=y)
            y = ops.conv(x, L["ffn1"], act=ops.ACT_RELU)
            y = ops.conv(y, L["ffn2"])
            x = ops.layernorm_ct(x, L["g2"], L["b2"], res=y)
        return ops.conv(x, P["proj"])  # (1, 2*inter, T) = [m_p ; logs_p]

    def _flow_reverse(self, P, z, g):
        """ResidualCouplingBlock.forward(reverse=True) (models.py:150-153; modules.py:440-459, 188-213)."""
        C = self.hidden_channels
        half = self.inter_channels // 2
        for fidx in (6, 4, 2, 0):
            Fl = P["flows"][fidx]
            z = torch.flip(z, [1])  # modules.Flip: channel permutation (data movement only)
            x0, x1 = z[:, :half], z[:, half:]
            h = ops.conv(x0, Fl["pre"])
            bias_all = ops.conv(g, Fl["cond"], res=Fl["in_bias"])  # (1, 2C*n_wn, 1): cond slice + in_layer bias
            out = None
            for l in range(Fl["n_wn"]):
                a = ops.conv(h, Fl["in%d" % l], bias=bias_all[0, 2 * C * l: 2 * C * (l + 1), 0].contiguous())
                acts = ops.gate_tanh_sigmoid(a)
                if l < Fl["n_wn"] - 1:
                    if out is None:
                        out = ops.conv(acts, Fl["skip%d" % l])
                    else:
                        ops.conv(acts, Fl["skip%d" % l], out=out, accumulate=True)
                    ops.conv(acts, Fl["res%d" % l], res=h, out=h)  # x = x + res_acts, in place
                else:
                    if out is None:
                        out = ops.conv(acts, Fl["skip%d" % l])
                    else:
                        ops.conv(acts, Fl["skip%d" % l], out=out, accumulate=True)
            ops.conv(out, Fl["post_neg"], res=x1, out=x1)  # x1 <- x1 - (post(h))   (mean-only coupling)
        return z

    def _decoder(self, P, x, f0, g, noise_src):
        """GeneratorNSF.forward / Generator.forward (models.py:494-516, 253-272)."""
        T = x.shape[2]
        har = None
        if self.use_f0:
            if noise_src is None:
                noise_src = torch.randn(T * self.upp, device=x.device)  # reference: torch.randn_like (models.py:368)
            har = ops.sine_source(f0.reshape(-1)[:T], noise_src.to(x.device).reshape(-1), self.upp, float(self.sr), P["lin_w"], P["lin_b"])
        pre_bias = ops.conv(g, P["cond"], res=P["conv_pre_bias"])  # conv_pre.bias + cond(g), per-channel constant
        x = ops.conv(x, P["conv_pre"], bias=pre_bias.reshape(-1))
        nk = len(self.resblock_kernel_sizes)
        for i, pt in enumerate(P["ups"]):
            add = None
            if har is not None:
                s, nc = P["noise"][i]
                L = har.numel()
                if s > 1:
                    xp = F.pad(har, (s // 2, s // 2))                     # zero padding of Conv1d(padding=s//2)
                    phases = xp.view(L // s + 1, s).t().contiguous()      # X[ph][q] = xpad[q*s + ph]
                    add = ops.conv(phases.unsqueeze(0), nc)
                else:
                    add = ops.conv(har.view(1, 1, L), nc)
            x = ops.conv_transpose(x, pt, add=add, pre_act=ops.ACT_LRELU, pre_slope=LRELU_SLOPE)
            acc = torch.empty_like(x)
            # The num_kernels ResBlocks of a stage read the same x and are independent up to their last convolution, which adds into
            # the shared sum (models.py:506-512).  Opt-in (AICG_RB_STREAMS=1): each chain on its own stream, so that a launch's ragged
            # last round (the 256-channel stage: 1 142 tiles on 512 slots = 2.2 rounds) could run under another chain's launch; the
            # accumulating convolutions stay ordered k = 3, 7, 11 by events -- output bit-identical to the one-stream walk
            # (tests/test_synth.py).  Measured (tools/kbench_rb_streams.py, round-robin): chunk loop 238.1 ms against 237.8 -- the
            # launches do not overlap usefully (two resident workgroups per CU either way), so the one-stream walk stays the default.
            streams = self._rb_streams(x.device, nk)
            main = torch.cuda.current_stream(x.device) if streams else None
            bufs = [tuple(torch.empty_like(x) for _ in range(3)) for _ in range(nk if streams else 1)]
            if streams:
                ready = torch.cuda.Event()
                ready.record(main)
            prev_acc = None
            for j in range(nk):
                convs = P["resblocks"][i * nk + j]
                st = streams[j - 1] if (streams and j > 0) else main
                if streams and j > 0:
                    st.wait_event(ready)
                with (torch.cuda.stream(st) if streams else contextlib.nullcontext()):
                    # (buffers of a chain come from the main stream's pool -- allocated before the stream switch -- and stay referenced
                    #  until the main stream has waited for the last chain: no cross-stream reuse while a side stream still reads them)
                    tmp, ya, yb = bufs[j] if streams else bufs[0]
                    y = x
                    for m, (c1, c2) in enumerate(convs):
                        last = m == len(convs) - 1
                        if c2 is None:  # ResBlock2
                            src = c1
                            inp = y
                        else:
                            # c2 reads c1's output only through the leaky ReLU (modules.py:305-309): c1's epilogue applies it once per
                            # element (same multiplication, same bits) and c2 runs without an input activation
                            ops.conv(y, c1, out=tmp, pre_act=ops.ACT_LRELU, pre_slope=LRELU_SLOPE, act=ops.ACT_LRELU, act_slope=LRELU_SLOPE)
                            src, inp = c2, tmp
                        if last:  # xs += resblock(x); x = xs / num_kernels  (models.py:506-512)
                            if streams and prev_acc is not None:
                                st.wait_event(prev_acc)
                            pre = ops.ACT_NONE if c2 is not None else ops.ACT_LRELU
                            ops.conv(inp, src, res=y, out=acc, pre_act=pre, pre_slope=LRELU_SLOPE, out_scale=1.0 / nk, accumulate=j > 0)
                            if streams:
                                prev_acc = torch.cuda.Event()
                                prev_acc.record(st)
                        else:
                            dst = ya if y is not ya else yb
                            ops.conv(inp, src, res=y, out=dst, pre_act=ops.ACT_NONE if c2 is not None else ops.ACT_LRELU, pre_slope=LRELU_SLOPE)
                            y = dst
            if streams:
                main.wait_event(prev_acc)
            x = acc
        # F.leaky_relu default slope 0.01 (models.py:513), conv_post (no bias), tanh
        return ops.conv(x, P["conv_post"], pre_act=ops.ACT_LRELU, pre_slope=0.01, act=ops.ACT_TANH)

    def _rb_streams(self, device, nk):
        """Side streams for the ResBlock chains of a vocoder stage (one per chain but the first, created once per model: the caching
        allocator keeps a pool per stream), or None where the chains run one after the other (the default; CPU / emulator)."""
        if device.type != "cuda" or nk < 2 or _env.dev("AICG_RB_STREAMS", "0") != "1":
            return None
        have = getattr(self, "_rb_side", None)
        if have is None or len(have) != nk - 1:
            have = self._rb_side = [torch.cuda.Stream(device=device) for _ in range(nk - 1)]
        return have

    def infer(self, phone, phone_lengths, pitch=None, nsff0=None, sid=None, max_len=None, noise_z=None, noise_src=None,
              phone_ct=None):
        """Same contract as the reference (models.py:745-751).  `noise_z` (1, inter, T) and `noise_src` (T*upp)
        optionally replace the two torch.randn_like draws (parity tests inject them on both sides).  `phone_ct`
        (1, phone_dim, T) passes the features already channel-major (what aicg_feats_prepare writes)."""
        if not self.use_f0 and sid is None:  # _nono signature: infer(phone, phone_lengths, sid, max_len=None)
            sid, pitch = pitch, None
        P = self._prepare()
        dev = self.device
        if phone_ct is None:
            phone = phone.to(dev).float()
            assert phone.shape[0] == 1, "batch 1 (the reference pipeline never batches chunks)"
            T = phone.shape[1]
            phone_ct = phone[0].t().contiguous().unsqueeze(0)  # (1, phone_dim, T)  layout plumbing
        else:
            T = phone_ct.shape[2]
        front = self.infer_front(phone_ct, pitch, sid, noise_z)
        return self.infer_back(front, nsff0, noise_src, max_len)

    def infer_front(self, phone_ct, pitch, sid, noise_z=None):
        """First half of infer(): speaker embedding, text encoder, prior sample, reverse flow -> state for infer_back.  A few
        hundred short launches over T frames (they leave most of the chip idle): VC.pipeline queues the next chunk's front on a
        second stream underneath the current chunk's vocoder."""
        P = self._prepare()
        dev = self.device
        T = phone_ct.shape[2]
        g = P["emb_g"][sid.to(dev).reshape(-1)[:1]].unsqueeze(-1).contiguous()  # (1, gin, 1)
        stats = self._enc_p(P, phone_ct, None if pitch is None else pitch.to(dev))
        if noise_z is None:
            noise_z = torch.randn((1, self.inter_channels, T), device=dev)  # reference: models.py:748
        z_p = ops.prior_sample(stats, noise_z.to(dev).float().contiguous(), 0.66666)
        z = self._flow_reverse(P, z_p.clone(), g)
        return {"z": z, "z_p": z_p, "stats": stats, "g": g, "T": T}

    def infer_back(self, front, nsff0=None, noise_src=None, max_len=None):
        """Second half of infer(): the (NSF-)HiFiGAN vocoder on the flow's output -> (o, x_mask, (z, z_p, m_p, logs_p))."""
        P = self._prepare()
        dev = self.device
        z, z_p, stats, g, T = front["z"], front["z_p"], front["stats"], front["g"], front["T"]
        zz = z if max_len is None else z[:, :, :max_len].contiguous()
        o = self._decoder(P, zz, None if nsff0 is None else nsff0.to(dev).float(), g, noise_src)
        x_mask = torch.ones((1, 1, T), dtype=torch.float32, device=dev)
        m_p, logs_p = stats[:, : self.inter_channels], stats[:, self.inter_channels:]
        return o, x_mask, (z, z_p, m_p, logs_p)


class SynthesizerTrnMs768NSFsid(_SynthesizerBase):
    phone_dim, use_f0 = 768, True


class SynthesizerTrnMs256NSFsid(_SynthesizerBase):
    phone_dim, use_f0 = 256, True


class SynthesizerTrnMs768NSFsid_nono(_SynthesizerBase):
    phone_dim, use_f0 = 768, False


class SynthesizerTrnMs256NSFsid_nono(_SynthesizerBase):
    phone_dim, use_f0 = 256, False
