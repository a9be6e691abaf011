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


@pytest.mark.parametrize("kind", ["refiner", "scorer"])
def test_reader_takes_exporter_variants(tmp_path, kind):
    """Round-3 review #9: what the published files may look like beyond the two TorchScript forms -- fp16 initialisers behind Cast nodes
    with anonymous names (no constant folding), Linear weights stored [out, in] behind a Transpose, LayerNorm decomposed with anonymous
    constants (found by the Div -> Mul(const) -> Add(const) tail), onnxruntime's fused `Attention` node with packed [in, 3 * hidden]
    weights.  The recovered tensors equal the fp16 rounding of the folded state."""
    st = W.make_synthetic_state(kind)
    p = str(tmp_path / f"{kind}_variant.onnx")
    write_model(p, kind, st, "variant")
    got = R.extract(p, kind)
    want = W.fold_batchnorm(st)
    assert set(got) == set(want), (sorted(set(want) - set(got)), sorted(set(got) - set(want)))
    for k, v in want.items():
        assert got[k].shape == v.shape, k
        np.testing.assert_array_equal(got[k], v.astype(np.float16).astype(np.float32), err_msg=k)
    rep = R.check(p, kind)
    text = rep if isinstance(rep, str) else str(rep)
    assert "DIFF" not in text.upper() or "0 diff" in text.lower(), text


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


def test_check_reports_a_structural_diff_instead_of_asserting(tmp_path):
    """`python -m foundationpose_cpp_amd.onnx_reader --check`: variations the reader absorbs are NOTES (leading Transpose, unfused
    BatchNorm, split q/k/v), what breaks the conversion is a DIFF that names the layer -- for the day the published files arrive"""
    st = W.make_synthetic_state("refiner")
    p = str(tmp_path / "r.onnx")
    write_model(p, "refiner", st, "named")
    good, lines = R.check(p)
    text = "\n".join(lines)
    assert good and "DIFF" not in text
    assert "feeds Transpose" in text and "unfused BatchNormalization" in text and "separate 512x512 projections" in text
    assert "opset 17" in text and "convert: 58 tensors" in text
    # a different trunk: wrong channel count in one convolution -> the diff names the layer and both shapes
    st2 = {k: v.copy() for k, v in st.items()}
    st2["encodeAB.2.conv.weight"] = np.zeros((256, 256, 3, 3), np.float32)
    for k in ("conv.bias", "bn.weight", "bn.bias", "bn.running_mean", "bn.running_var"):
        st2["encodeAB.2." + k] = np.ones(256, np.float32)
    write_model(p, "refiner", st2, "folded")
    good, lines = R.check(p, "refiner")
    text = "\n".join(lines)
    assert not good and "conv #10 (encodeAB.2): found (256, 256, 3, 3), expected (512, 256, 3, 3)" in text
    # a scorer handed in as a refiner, and the CLI's exit code
    ss = W.make_synthetic_state("scorer")
    write_model(p, "scorer", ss)
    good, lines = R.check(p, "refiner")
    assert not good and any(l.startswith("DIFF") and "output" in l for l in lines)
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "foundationpose_cpp_amd.onnx_reader", "--check", p], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0 and "RESULT: convertible" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-m", "foundationpose_cpp_amd.onnx_reader", "--check", p, "refiner"], cwd=root, capture_output=True, text=True)
    assert r.returncode == 1 and "NOT convertible" in r.stdout
