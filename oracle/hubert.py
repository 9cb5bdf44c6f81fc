"""TEST INFRASTRUCTURE (oracle): CPU fp32 restatement of fairseq HubertModel.extract_features as the reference
calls it (src/vc_infer_pipeline.py:398-406; model loaded at src/rvc.py:98-109).

fairseq 0.12.2 is a pip dependency of the reference (requirements.txt:2), absent from /root/reference and from
this image: the algorithm below follows fairseq's published HuBERT-base (ConvFeatureExtractionModel with
GroupNorm on layer 0, LayerNorm + post_extract_proj, weight-normed grouped positional conv + SamePad + GELU,
post-LN Transformer encoder) and is pinned against `transformers.HubertModel` (identical hyper-parameters,
weights mapped key-by-key) through tests/golden/hubert_*.npz.  PARITY UNPINNED with respect to fairseq itself.
"""
import math

import torch
import torch.nn.functional as F


def pos_conv_weight(sd):
    v, g = sd["encoder.pos_conv.0.weight_v"], sd["encoder.pos_conv.0.weight_g"]
    nrm = v.transpose(0, 2).flatten(1).norm(dim=1).view(1, 1, -1)  # weight_norm(dim=2)
    return v * (g / nrm)


def feature_extractor(sd, cfg, wav):
    """wav (1, N) -> (1, conv_dim, T).  Layer 0: conv, GroupNorm(C, C) (per-channel stats over all time), GELU."""
    x = wav.unsqueeze(1)
    for i, (k, s) in enumerate(zip(cfg["conv_kernel"], cfg["conv_stride"])):
        x = F.conv1d(x, sd["feature_extractor.conv_layers.%d.0.weight" % i], None, stride=s)
        if i == 0:
            c = x.shape[1]
            x = F.group_norm(x, c, sd["feature_extractor.conv_layers.0.2.weight"], sd["feature_extractor.conv_layers.0.2.bias"], 1e-5)
        x = F.gelu(x)
    return x


def extract_features(sd, cfg, wav, output_layer=12):
    """Returns (1, T, embed): the output of encoder layer `output_layer` (1-based), post-LN model."""
    E, H = cfg["embed"], cfg["heads"]
    x = feature_extractor(sd, cfg, wav).transpose(1, 2)                       # (1, T, C)
    x = F.layer_norm(x, (x.shape[-1],), sd["layer_norm.weight"], sd["layer_norm.bias"], 1e-5)
    x = F.linear(x, sd["post_extract_proj.weight"], sd["post_extract_proj.bias"])
    pk = cfg["pos_k"]
    pc = F.conv1d(x.transpose(1, 2), pos_conv_weight(sd), sd["encoder.pos_conv.0.bias"], padding=pk // 2, groups=cfg["pos_groups"])
    if pk % 2 == 0:
        pc = pc[:, :, :-1]                                                   # SamePad
    x = x + F.gelu(pc).transpose(1, 2)
    x = F.layer_norm(x, (E,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], 1e-5)
    dh = E // H
    T = x.shape[1]
    for i in range(min(output_layer, cfg["layers"])):
        p = "encoder.layers.%d." % i
        q = F.linear(x, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * dh ** -0.5
        k = F.linear(x, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = F.linear(x, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q, k, v = (z.view(1, T, H, dh).transpose(1, 2) for z in (q, k, v))
        a = F.softmax(q @ k.transpose(2, 3), dim=-1) @ v
        a = a.transpose(1, 2).reshape(1, T, E)
        a = F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        x = F.layer_norm(x + a, (E,), sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"], 1e-5)
        h = F.gelu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
        h = F.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])
        x = F.layer_norm(x + h, (E,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], 1e-5)
    return x


def final_proj(sd, x):
    return F.linear(x, sd["final_proj.weight"], sd["final_proj.bias"])


def to_hf_state_dict(sd):
    """fairseq key names -> transformers.HubertModel key names (used only to pin this file against HF)."""
    out = {}
    for k, v in sd.items():
        if k.startswith("final_proj"):
            continue
        k2 = k
        k2 = k2.replace("feature_extractor.conv_layers.0.2.", "feature_extractor.conv_layers.0.layer_norm.")
        for i in range(7):
            k2 = k2.replace("feature_extractor.conv_layers.%d.0.weight" % i, "feature_extractor.conv_layers.%d.conv.weight" % i)
        if k2.startswith("layer_norm."):
            k2 = "feature_projection." + k2
        k2 = k2.replace("post_extract_proj.", "feature_projection.projection.")
        k2 = k2.replace("encoder.pos_conv.0.weight_g", "encoder.pos_conv_embed.conv.parametrizations.weight.original0")
        k2 = k2.replace("encoder.pos_conv.0.weight_v", "encoder.pos_conv_embed.conv.parametrizations.weight.original1")
        k2 = k2.replace("encoder.pos_conv.0.bias", "encoder.pos_conv_embed.conv.bias")
        k2 = k2.replace(".self_attn.", ".attention.").replace(".self_attn_layer_norm.", ".layer_norm.")
        k2 = k2.replace(".fc1.", ".feed_forward.intermediate_dense.").replace(".fc2.", ".feed_forward.output_dense.")
        out[k2] = v
    return out
