"""ONNX initializer reader for the MDX-Net graphs (aicovergen_amd/onnx_weights.py): protobuf decoding, the graph walk
onto kuielab ConvTDFNet parameter names, and the separator running from an `.onnx` path like the reference's
`MDX(model_path, ...)` does (src/mdx.py:74-77)."""
import os
import warnings

import numpy as np
import pytest
import torch

from aicovergen_amd import onnx_weights
from aicovergen_amd.mdx_net import ConvTDFNet, infer_cfg
from conftest import rel_rms
from oracle import mdxnet, weights

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FIXTURE = os.path.join(GOLD, "mdx_tiny.onnx")
CFG = dict(weights.MDX_TINY)


def test_parse_committed_fixture():
    nodes, inits = onnx_weights.parse_model(FIXTURE)
    ops = [n["op"] for n in nodes]
    assert ops.count("Conv") == 14 and ops.count("ConvTranspose") == 2 and ops.count("MatMul") == 10
    assert ops.count("BatchNormalization") == 12  # tdf linears + transposed convs; Conv+BN pairs were fused by the exporter
    first = nodes[0]
    assert first["op"] == "Conv" and first["attr"]["kernel_shape"] == [1, 1]
    assert inits[first["input"][1]].shape == (CFG["g"], CFG["dim_c"], 1, 1)


def test_state_dict_reproduces_the_network():
    """Parameters read back from the graph drive the restated U-Net to the same output as the parameters that were
    exported (Conv+BN fusion by the exporter only re-associates the arithmetic)."""
    sd0 = weights.mdx_state_dict(CFG, 7)
    sd1 = onnx_weights.load_onnx_state_dict(FIXTURE)
    assert set(sd1) == {k for k in sd0 if not k.endswith("num_batches_tracked")}
    cfg1 = infer_cfg(sd1)
    assert all(cfg1[k] == CFG[k] for k in ("dim_c", "g", "n", "l", "k", "bn", "dim_f"))
    torch.manual_seed(0)
    spec = torch.randn(2, CFG["dim_c"], CFG["dim_f"], CFG["dim_t"])
    with torch.no_grad():
        assert rel_rms(mdxnet.unet(sd1, CFG, spec), mdxnet.unet(sd0, CFG, spec)) < 2e-6


def test_fresh_export_round_trip(tmp_path):
    """Same, for a graph exported now with other parameters and a different depth (n = 1, l = 3, no tdf bias)."""
    import sys
    sys.path.insert(0, GOLD)
    try:
        from make_onnx_fixture import export_unet
        cfg = dict(CFG, n=1, l=3)
        sd0 = weights.mdx_state_dict(cfg, 11)
        for k in list(sd0):
            if ".tdf." in k and k.endswith((".0.bias", ".3.bias")):
                sd0[k] = torch.zeros_like(sd0[k])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            export_unet(sd0, cfg, str(tmp_path / "m.onnx"))
    except Exception as exc:  # exporter not usable on this box: the committed fixture still covers the reader
        pytest.skip("torch.onnx.export unavailable: %r" % (exc,))
    sd1 = onnx_weights.load_onnx_state_dict(str(tmp_path / "m.onnx"))
    spec = torch.randn(1, cfg["dim_c"], cfg["dim_f"], cfg["dim_t"])
    with torch.no_grad():
        assert rel_rms(mdxnet.unet(sd1, cfg, spec), mdxnet.unet(sd0, cfg, spec)) < 2e-6


def test_kernels_run_from_onnx_path(dev):
    """ConvTDFNet built from the `.onnx` file == ConvTDFNet built from the exported state_dict, on the HIP kernels."""
    from aicovergen_amd.mdx import load_network_state
    net1 = ConvTDFNet(load_network_state(FIXTURE), dev.device)
    net0 = ConvTDFNet(weights.mdx_state_dict(CFG, 7), dev.device)
    torch.manual_seed(1)
    x = torch.randn(2, CFG["dim_c"], CFG["dim_t"], CFG["dim_f"])
    assert rel_rms(net1.forward_tf(dev.t(x)), net0.forward_tf(dev.t(x)).cpu()) < 5e-6


def test_not_a_unet_is_a_clear_error(tmp_path):
    p = tmp_path / "bad.onnx"
    p.write_bytes(b"\x08\x07")  # ModelProto with ir_version only
    with pytest.raises(ValueError, match="GraphProto"):
        onnx_weights.load_onnx_state_dict(str(p))
