"""Test helper: write refiner_hwc.onnx / scorer_hwc.onnx with PyTorch's own TorchScript ONNX exporter (the tool the published
FoundationPose ONNX files were made with) from the oracle's restatement of the architecture (oracle/nets_torch.py).

The exporter's serialiser is C++ and needs nothing else; the Python package `onnx` (absent here) is only imported by
torch.onnx for models carrying onnxscript custom functions, so that one post-processing hook is stubbed for the duration of
the export.  TEST INFRASTRUCTURE (uses oracle/): the product reads ONNX files with foundationpose_cpp_amd/onnx_reader.py."""
import contextlib
import warnings

import torch


@contextlib.contextmanager
def _exporter_without_onnx_package():
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils as opu
    saved = opu._add_onnxscript_fn
    opu._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    fast = torch.backends.mha.get_fastpath_enabled()
    torch.backends.mha.set_fastpath_enabled(False)      # the fused inference path of nn.MultiheadAttention is not exportable
    try:
        yield
    finally:
        opu._add_onnxscript_fn = saved
        torch.backends.mha.set_fastpath_enabled(fast)


def export(model: torch.nn.Module, kind: str, path: str, opset: int = 17, batch: int = 2) -> None:
    """blob names as the reference binds them (detection_6d_foundationpose/src/foundationpose.cpp:78-83)"""
    a = torch.randn(batch, 160, 160, 6)
    b = torch.randn(batch, 160, 160, 6)
    outs = ["trans", "rot"] if kind == "refiner" else ["scores"]
    with _exporter_without_onnx_package(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(model.eval(), (a, b), path, dynamo=False, opset_version=opset,
                          input_names=["render_input", "transf_input"], output_names=outs,
                          dynamic_axes={"render_input": {0: "n"}, "transf_input": {0: "n"}}, do_constant_folding=True)
