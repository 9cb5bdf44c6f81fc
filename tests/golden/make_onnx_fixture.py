"""Writes tests/golden/mdx_tiny.onnx: torch.onnx.export (TorchScript exporter, opset 13, eval mode) of the restated TFC-TDF
U-Net (oracle/mdxnet.py) with the seeded parameters of synthetic/weights.py (MDX_TINY, seed 7).  The `onnx` Python package
is not installed in the build container; the exporter only needs it for a post-processing hook that is a no-op for this
graph, so the hook is bypassed."""
import io
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mdxnet  # noqa: E402
from synthetic import weights


def export_unet(sd, cfg, path=None):
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.names = list(sd)
            for i, v in enumerate(sd.values()):
                self.register_buffer("t%d" % i, v.clone())

        def forward(self, spec):
            return mdxnet.unet({k: getattr(self, "t%d" % i) for i, k in enumerate(self.names)}, cfg, spec)

    spec = torch.randn(1, cfg["dim_c"], cfg["dim_f"], cfg["dim_t"])
    f = io.BytesIO()
    torch.onnx.export(Net().eval(), (spec,), f, opset_version=13, dynamo=False, input_names=["input"],
                      output_names=["output"])
    if path:
        open(path, "wb").write(f.getvalue())
    return f.getvalue()


if __name__ == "__main__":
    cfg = dict(weights.MDX_TINY)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mdx_tiny.onnx")
    export_unet(weights.mdx_state_dict(cfg, 7), cfg, out)
    print("wrote", out, os.path.getsize(out), "bytes")
