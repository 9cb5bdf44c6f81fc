"""HuBERT-base content-feature extractor on the gfx950 kernels, exposing the slice of fairseq's HubertModel API
that the reference uses: `load_hubert` builds it (reference src/rvc.py:98-109) and VC.vc calls
`model.extract_features(source=, padding_mask=, output_layer=)` and `model.final_proj` (src/vc_infer_pipeline.py:398-406).

Everything stays channel-major (C, T): the strided feature-extractor convs, the grouped positional conv and all
Linear layers run through the implicit-GEMM conv kernel; attention through the fused softmax kernel (T x T never
materialised); LayerNorms through layernorm_ct; layer-0 GroupNorm + GELU through rownorm_act.
"""
import io
import math
import pickle

import torch

from . import ops

HUBERT_BASE = dict(conv_dim=512, conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_stride=(5, 2, 2, 2, 2, 2, 2), embed=768,
                   heads=12, ffn=3072, layers=12, pos_k=128, pos_groups=16, final_dim=256)


def _infer_cfg(sd):
    """Hyper-parameters from tensor shapes (hubert_base.pt carries them in a fairseq cfg object we cannot unpickle)."""
    n_conv = 0
    kernels = []
    while "feature_extractor.conv_layers.%d.0.weight" % n_conv in sd:
        kernels.append(sd["feature_extractor.conv_layers.%d.0.weight" % n_conv].shape[2])
        n_conv += 1
    layers = 0
    while "encoder.layers.%d.fc1.weight" % layers in sd:
        layers += 1
    v = sd["encoder.pos_conv.0.weight_v"]
    E = v.shape[0]
    strides = tuple([5] + [2] * (n_conv - 1))  # wav2vec2 / HuBERT default conv_feature_layers strides
    return dict(conv_dim=sd["feature_extractor.conv_layers.0.0.weight"].shape[0], conv_kernel=tuple(kernels),
                conv_stride=strides, embed=E, heads=max(1, E // 64), ffn=sd["encoder.layers.0.fc1.weight"].shape[0],
                layers=layers, pos_k=v.shape[2], pos_groups=E // v.shape[1],
                final_dim=sd["final_proj.weight"].shape[0] if "final_proj.weight" in sd else 0)


class _FinalProj:
    def __init__(self, owner):
        self.owner = owner

    def __call__(self, x):
        """x: (1, T, embed) token-major -> (1, T, final_dim)   (v1 models: feats = model.final_proj(logits[0]))."""
        P = self.owner._prepare()
        y = ops.conv(x[0].t().contiguous().unsqueeze(0), P["final_proj"])
        return y[0].t().contiguous().unsqueeze(0)


class HubertModel:
    def __init__(self, state_dict, cfg=None):
        self._sd = {k: v.detach().to("cpu").float() for k, v in state_dict.items() if torch.is_tensor(v)}
        self.cfg = dict(cfg) if cfg is not None else _infer_cfg(self._sd)
        if "heads" in (cfg or {}):
            self.cfg["heads"] = cfg["heads"]
        self.device = torch.device("cpu")
        self._p = None
        self.final_proj = _FinalProj(self)

    # torch.nn.Module-like surface used by load_hubert
    def to(self, device):
        self.device = torch.device(device)
        self._p = None
        return self

    def half(self):
        # fp32 kernels by default (>= the reference's fp16 GPU path, src/rvc.py:103-104); AICG_HALF=1: fp16 operands on the matrix pipe for
        # the layers on the LDS-DMA staged kernels (ops.mark_half)
        self._half = ops.half_requested()
        if self._p is not None:
            ops.mark_half(self._p, self._half)
        return self

    def float(self):
        self._half = False
        if self._p is not None:
            ops.mark_half(self._p, False)
        return self

    def eval(self):
        return self

    def _prepare(self):
        if self._p is not None:
            return self._p
        sd, cfg, dev = self._sd, self.cfg, self.device
        P = {}
        fe = []
        for i, (k, s) in enumerate(zip(cfg["conv_kernel"], cfg["conv_stride"])):
            w = sd["feature_extractor.conv_layers.%d.0.weight" % i]
            b = sd.get("feature_extractor.conv_layers.%d.0.bias" % i)
            if i == 0 and w.shape[1] == 1 and k == 2 * s:
                # Conv1d(1 -> C, k = 2s, stride s) == Conv1d(s -> C, k = 2) over the s-phase view X[ph][q] = x[q*s + ph]
                w2 = w.view(w.shape[0], 2, s).permute(0, 2, 1).contiguous()
                fe.append(("phase", s, ops.PackedConv(w2, b, device=dev)))
            else:
                fe.append(("conv", s, ops.PackedConv(w, b, stride=s, device=dev)))
        P["fe"] = fe
        P["gn_w"] = sd["feature_extractor.conv_layers.0.2.weight"].to(dev)
        P["gn_b"] = sd["feature_extractor.conv_layers.0.2.bias"].to(dev)
        P["ln_w"], P["ln_b"] = sd["layer_norm.weight"].to(dev), sd["layer_norm.bias"].to(dev)
        P["proj"] = ops.PackedConv(sd["post_extract_proj.weight"], sd["post_extract_proj.bias"], device=dev)
        v, g = sd["encoder.pos_conv.0.weight_v"], sd["encoder.pos_conv.0.weight_g"]
        wpos = v * (g / v.transpose(0, 2).flatten(1).norm(dim=1).view(1, 1, -1))  # weight_norm(dim=2) folded
        P["pos"] = ops.PackedConv(wpos, sd["encoder.pos_conv.0.bias"], padding=cfg["pos_k"] // 2, groups=cfg["pos_groups"],
                                  device=dev)
        P["eln_w"], P["eln_b"] = sd["encoder.layer_norm.weight"].to(dev), sd["encoder.layer_norm.bias"].to(dev)
        layers = []
        scale = (cfg["embed"] // cfg["heads"]) ** -0.5
        for i in range(cfg["layers"]):
            p = "encoder.layers.%d." % i
            qw, qb = sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]
            kvw = torch.cat([sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"]], 0)
            kvb = torch.cat([sd[p + "self_attn.k_proj.bias"], sd[p + "self_attn.v_proj.bias"]], 0)
            if self.merge_qkv and math.frexp(scale)[0] == 0.5:
                # head_dim ** -0.5 is a power of two (HuBERT-base: 64 -> 1/8): scaling q's rows of the weights and its bias instead of the
                # GEMM's result is EXACT in fp32, and q, k, v become one 3 E-row GEMM (2 304 rows: 18 x 52 tiles instead of 6 x 52 + 12 x 52
                # in two launches -- 432 vs 515 us at the benched 13 216 tokens, profiles/r04_kbench_g1.txt)
                proj = {"qkv": ops.PackedConv(torch.cat([qw * scale, kvw], 0), torch.cat([qb * scale, kvb], 0), device=dev)}
            else:
                proj = {"q": ops.PackedConv(qw, qb, device=dev), "kv": ops.PackedConv(kvw, kvb, device=dev)}
            L = {**proj,
                 "o": ops.PackedConv(sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], device=dev),
                 "fc1": ops.PackedConv(sd[p + "fc1.weight"], sd[p + "fc1.bias"], device=dev),
                 "fc2": ops.PackedConv(sd[p + "fc2.weight"], sd[p + "fc2.bias"], device=dev),
                 "ln1": (sd[p + "self_attn_layer_norm.weight"].to(dev), sd[p + "self_attn_layer_norm.bias"].to(dev)),
                 "ln2": (sd[p + "final_layer_norm.weight"].to(dev), sd[p + "final_layer_norm.bias"].to(dev))}
            layers.append(L)
        P["layers"] = layers
        if "final_proj.weight" in sd:
            P["final_proj"] = ops.PackedConv(sd["final_proj.weight"], sd["final_proj.bias"], device=dev)
        if getattr(self, "_half", False):
            ops.mark_half(P, True)
        self._p = P
        return P

    def extract_features(self, source, padding_mask=None, mask=False, ret_conv=False, output_layer=None):
        """fairseq HubertModel.extract_features for one un-padded waveform (1, N): returns (features (1, T, embed),
        padding_mask).  `output_layer` is 1-based like fairseq's (12 for v2 models, 9 for v1)."""
        assert source.dim() == 2 and source.shape[0] == 1, "one chunk at a time, as VC.vc calls it"
        return self.extract_features_many([source], output_layer)[0], padding_mask

    merge_qkv = True              # q, k, v as one GEMM where folding head_dim ** -0.5 into q's weights is exact (tests switch it off)
    max_tokens_per_pass = 32768   # extract_features_many: tokens laid side by side per transformer pass (~0.4 GB of fc1 output)

    def _frontend(self, source):
        """Waveform (1, N) -> (1, embed, T) in front of the encoder LayerNorm: feature-extractor convs, layer-0 GroupNorm + GELU,
        LayerNorm, projection, positional conv.  Everything here sees the chunk's boundaries (strided convs, whole-chunk GroupNorm
        statistics, the k = 128 positional conv's padding): strictly per chunk."""
        P, dev = self._prepare(), self.device
        x = source.to(dev).float().contiguous()
        n = x.shape[1]
        cur = x.view(1, 1, n)

        def rows16(c, t):
            """(1, c, t) view of a buffer whose rows start on 16-byte boundaries (row stride rounded up to a multiple of four floats): the
            extractor's frame counts are odd (211 231 -> 105 615 -> ...), and the stride-2 layers are staged by 16-byte DMA pieces
            (csrc/conv_g1s.h).  The padding is never read into a stored result and never written."""
            return torch.empty((1, c, (t + 3) // 4 * 4), dtype=torch.float32, device=dev)[:, :, :t]

        for i, (kind, s, pc) in enumerate(P["fe"]):
            act = ops.ACT_NONE if i == 0 else ops.ACT_GELU
            if kind == "phase":
                to = (n - 2 * s) // s + 1
                nq = (n + s - 1) // s
                xp = torch.nn.functional.pad(cur.view(-1), (0, nq * s - n))          # zero fill to a multiple of s
                cur = ops.conv(xp.view(nq, s).t().contiguous().unsqueeze(0), pc, act=act, out_len=to, out=rows16(pc.cout, to))
            else:
                cur = ops.conv(cur, pc, act=act, out=rows16(pc.cout, pc.out_hw(1, cur.shape[2])[1]))
            if i == 0:  # GroupNorm(C, C): per-channel statistics over the whole chunk, then GELU
                cur = ops.rownorm_act(cur[0], P["gn_w"], P["gn_b"], act=ops.ACT_GELU).unsqueeze(0)
        T = cur.shape[2]
        h = ops.layernorm_ct(cur.contiguous(), P["ln_w"], P["ln_b"])   # (a copy of 512 x T only when T % 4 != 0)
        h = ops.conv(h, P["proj"])
        # x + GELU(SamePad(pos_conv(x))): the even kernel's extra last frame is simply not computed
        h = ops.conv(h, P["pos"], act=ops.ACT_GELU, res=h, out_len=T)
        return h

    def extract_features_many(self, sources, output_layer=None):
        """extract_features for several independent chunks at once -> [features (1, T_i, embed)].
        The transformer's linears, FFNs and LayerNorms act per token, so the chunks are laid side by side along the time axis
        ((embed, sum T_i), the layout everything already uses) and each GEMM runs ONCE over all of them -- at 3 300 tokens per chunk
        a 768-wide GEMM offers 26 x 6 tiles of 128 x 128 to 256 CUs, at four chunks 104 x 6; attention (and everything in
        `_frontend`) stays per chunk on column slices.  One chunk reproduces the single-chunk call exactly."""
        P, cfg = self._prepare(), self.cfg
        E, H = cfg["embed"], cfg["heads"]
        # Token budget per pass: activation memory grows with the tokens laid side by side (fc1: ffn x tokens floats), so a very
        # long track's chunks go through in groups (the GEMMs are far past chip-filling at this size).  Grouping only changes which
        # chunks share a GEMM launch, i.e. tile selection and with it fp32 summation order: results differ by ~1e-6 relative between
        # groupings, not bit for bit (see VC.pipeline's docstring).
        if len(sources) > 1:
            groups, cur, tok = [], [], 0
            for src in sources:
                t = (src.shape[-1] - 400) // 320 + 1
                if cur and tok + t > self.max_tokens_per_pass:
                    groups.append(cur)
                    cur, tok = [], 0
                cur.append(src)
                tok += t
            groups.append(cur)
            if len(groups) > 1:
                return [y for g in groups for y in self.extract_features_many(g, output_layer)]
        fronts = [self._frontend(src) for src in sources]
        lens = [f.shape[2] for f in fronts]
        offs = [0]
        for t in lens:
            offs.append(offs[-1] + t)
        if len(fronts) == 1:
            h = fronts[0]
            tot = offs[-1]
        else:
            # the row stride of the side-by-side map is rounded up to 32 tokens: every 128-token tile of the GEMMs then starts on a
            # cache line (13 198 -> 13 216: 768>3072 94 -> 100, 768>1536 87 -> 95 TFLOP/s, tools/kbench_gemm_probe.py).  The padding
            # tokens are zeros; every op between here and the final slices acts per token, so they never meet a real one.
            tot = -(-offs[-1] // 32) * 32
            h = torch.zeros((1, E, tot), dtype=torch.float32, device=fronts[0].device)
            for f, o in zip(fronts, offs):
                h[:, :, o:o + f.shape[2]] = f
        del fronts
        h = ops.layernorm_ct(h, P["eln_w"], P["eln_b"])
        n_layers = cfg["layers"] if output_layer is None else min(output_layer, cfg["layers"])
        for L in P["layers"][:n_layers]:
            if "qkv" in L:
                qkv = ops.conv(h, L["qkv"])
                q, kv = qkv[:, :E], qkv[:, E:]
            else:
                q = ops.conv(h, L["q"], out_scale=(E // H) ** -0.5)
                kv = ops.conv(h, L["kv"])
            if len(lens) == 1:
                a = ops.attention(q[0], kv[0, :E], kv[0, E:], H)
            else:
                a = torch.empty((E, tot), dtype=torch.float32, device=h.device)
                a[:, offs[-1]:] = 0.0
                for o, t in zip(offs, lens):
                    a[:, o:o + t] = ops.attention(q[0, :, o:o + t], kv[0, :E, o:o + t], kv[0, E:, o:o + t], H)
            a = ops.conv(a.unsqueeze(0), L["o"])
            h = ops.layernorm_ct(h, L["ln1"][0], L["ln1"][1], res=a)
            f = ops.conv(h, L["fc1"], act=ops.ACT_GELU)
            f = ops.conv(f, L["fc2"])
            h = ops.layernorm_ct(h, L["ln2"][0], L["ln2"][1], res=f)
        return [h[0, :, o:o + t].t().contiguous().unsqueeze(0) for o, t in zip(offs, lens)]  # token-major, as fairseq returns them


class _TolerantUnpickler(pickle.Unpickler):
    """hubert_base.pt pickles fairseq / omegaconf config objects next to the tensors; those packages are not
    needed to read the weights, so unknown classes are replaced by inert placeholders."""

    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except Exception:
            return type(name, (), {"__init__": lambda self, *a, **k: None, "__setstate__": lambda self, s: None})


class _PickleModule:
    Unpickler = _TolerantUnpickler
    load = staticmethod(pickle.load)
    __name__ = "tolerant_pickle"


def load_state(model_path):
    """Read a HuBERT checkpoint: fairseq format ({"model": state_dict, "cfg"/"args": ...}) or a bare state_dict."""
    try:
        ckpt = torch.load(model_path, map_location="cpu", weights_only=False)
    except Exception:
        ckpt = torch.load(model_path, map_location="cpu", weights_only=False, pickle_module=_PickleModule)
    if isinstance(ckpt, dict) and "model" in ckpt and isinstance(ckpt["model"], dict):
        return ckpt["model"]
    return ckpt
