"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/foundationpose_amd.h declares,
and it fails loudly (no CPU fallback) when there is no GPU."""
import os
import re

import pytest

from foundationpose_cpp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "foundationpose_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fp_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_all_exported():
    declared = _declared_symbols()
    assert len(declared) >= 20
    assert sorted(_lib.SYMBOLS) == declared, "python binding list and header disagree"
    L = _lib.lib()
    for s in declared:
        assert hasattr(L, s), f"libfoundationpose_amd.so does not export {s}"


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from foundationpose_cpp_amd import FoundationPose, FoundationPoseError, synthetic as syn
    with pytest.raises(FoundationPoseError, match="no HIP device"):
        FoundationPose(syn.make_mesh(1), syn.intrinsics())


def test_product_does_not_touch_oracle():
    # the product path must never import / link the oracle (it is test infrastructure)
    pkg = os.path.join(ROOT, "foundationpose_cpp_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "fp_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
