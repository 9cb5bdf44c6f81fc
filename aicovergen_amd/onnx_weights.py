"""Placeholder for the ONNX initializer reader of the UVR MDX-Net graphs.  No `.onnx` file is
available offline to validate a reader against, so the path fails loudly instead of guessing."""


def load_onnx_state_dict(path):
    raise NotImplementedError(
        "%s: reading MDX-Net parameters from an .onnx graph is not implemented yet; export the network's state_dict "
        "(kuielab ConvTDFNet parameter names) with torch.save and pass that file instead" % path)
