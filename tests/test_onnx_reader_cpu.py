"""ONNX initialiser reader: ONNX written in the TorchScript exporter's conventions by
tests/onnx_writer.py -> onnx_reader.extract -> must equal weights.fold_batchnorm(state), i.e. exactly the tensors the
synthetic FPW files hold.  Mapping against the REAL refiner_hwc.onnx / scorer_hwc.onnx is unverified (files unavailable)."""
import numpy as np
import pytest

from foundationpose_cpp_amd import onnx_reader as R, weights as W

from onnx_writer import write_model


@pytest.mark.parametrize("kind", ["refiner", "scorer"])
@pytest.mark.parametrize("flavour", ["folded", "named"])
def test_reader_recovers_folded_state(tmp_path, kind, flavour):
    st = W.make_synthetic_state(kind)
    p = str(tmp_path / f"{kind}_hwc.onnx")
    write_model(p, kind, st, flavour)
    got = R.extract(p, kind)
    want = W.fold_batchnorm(st)
    assert set(got) == set(want), (sorted(set(want) - set(got)), sorted(set(got) - set(want)))
    for k, v in want.items():
        assert got[k].shape == v.shape, k
        if flavour == "folded":
            np.testing.assert_array_equal(got[k], v, err_msg=k)
        else:     # the reader folds BatchNorm itself, with the file's fp32 epsilon attribute (1e-5 rounded to fp32)
            np.testing.assert_allclose(got[k], v, rtol=1e-6, atol=1e-9, err_msg=k)
    out = str(tmp_path / "w.fpw")
    R.convert(p, kind, out)
    back = W.read_fpw(out)
    assert all(np.array_equal(back[k], got[k]) for k in want)
    assert "Conv x15" in R.describe(p)


def test_reader_fails_loudly(tmp_path):
    st = W.make_synthetic_state("scorer")
    p = str(tmp_path / "s.onnx")
    write_model(p, "scorer", st)
    with pytest.raises(ValueError, match="two outputs"):
        R.extract(p, "refiner")
    bad = str(tmp_path / "bad.onnx")
    open(bad, "wb").write(b"\x08\x08")
    with pytest.raises(ValueError, match="no GraphProto"):
        R.read_graph(bad)
    st2 = dict(st)
    st2["linear.weight"] = np.zeros((2, 512), np.float32)
    st2["linear.bias"] = np.zeros(2, np.float32)
    write_model(p, "scorer", st2)
    with pytest.raises(ValueError, match="expected a 1x512"):
        R.extract(p, "scorer")
