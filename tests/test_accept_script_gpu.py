"""Dry run of tools/accept_real_assets.sh (the one-command acceptance run for the day the real ONNX files, the mustard sequence and a
pose log of the reference arrive: onnx_reader --check -> weights --onnx -> fp_demo -> compare_pose_log.py, PASS / FAIL exit code) on
synthetic stand-ins: ONNX files written by tests/onnx_writer.py in the exporter's conventions, a synthetic sequence in the reference's
dataset layout, and as the "reference log" (a) the demo's own pose log re-written in the reference's glog / Eigen format -> PASS,
(b) the same with one frame moved by 3 mm -> FAIL (exit 1), (c) a file of another architecture -> the structural check stops it (exit 2).
Mirrors simple_tests/src/test_foundationpose.cpp:48-104."""
import os
import subprocess
import sys

import numpy as np
import pytest

from foundationpose_cpp_amd import dataset as D, synthetic as syn, weights as W

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from onnx_writer import write_model  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tools", "accept_real_assets.sh")


def _glog(poses, path):
    """the reference's log format: `first Pose : <row 0>` + three more rows, then `Track pose : ...` per frame"""
    with open(path, "w") as f:
        for i, p in enumerate(poses):
            rows = ["%.9g %.9g %.9g %.9g" % tuple(r) for r in p]
            f.write(f"W0000 00:00:00.000000 1 test_foundationpose.cpp:{62 if i == 0 else 89}] {'first Pose' if i == 0 else 'Track pose'} : {rows[0]}\n")
            f.write("\n".join(rows[1:]) + "\n")


def test_usage_error_exits_2():
    res = subprocess.run(["bash", SCRIPT, "--data", "/nonexistent"], capture_output=True, text=True)
    assert res.returncode == 2 and "accept_real_assets.sh" in res.stderr


@pytest.mark.gpu
def test_acceptance_script_dry_run(tmp_path):
    root = str(tmp_path / "synthetic0")
    D.write_synthetic_sequence(root, n_frames=4)
    onnx = {}
    for kind in ("refiner", "scorer"):
        onnx[kind] = str(tmp_path / f"{kind}_hwc.onnx")
        write_model(onnx[kind], kind, W.make_synthetic_state(kind), "named")
    out = str(tmp_path / "out")
    dummy = str(tmp_path / "none.log")
    open(dummy, "w").write("first Pose : 1 0 0 0\n0 1 0 0\n0 0 1 0\n0 0 0 1\n")
    base = ["bash", SCRIPT, "--refiner-onnx", onnx["refiner"], "--scorer-onnx", onnx["scorer"], "--data", root, "--out", out]
    # first run: any reference log (it fails the gate or the frame count) -- what matters is that steps 1-3 produce poses.txt
    res = subprocess.run(base + ["--reference-log", dummy], capture_output=True, text=True, timeout=900)
    assert res.returncode in (1, 2) and os.path.exists(os.path.join(out, "poses.txt")), res.stdout + res.stderr
    rows = [l.split() for l in open(os.path.join(out, "poses.txt"))]
    poses = syn.from_colmajor(np.array([[float(v) for v in r[1:]] for r in rows], np.float32))
    assert len(poses) == 4
    good, bad = str(tmp_path / "ref_good.log"), str(tmp_path / "ref_bad.log")
    _glog(poses, good)
    moved = poses.copy()
    moved[2, 0, 3] += 0.003
    _glog(moved, bad)
    fpw = ["--refiner-fpw", os.path.join(out, "refiner.fpw"), "--scorer-fpw", os.path.join(out, "scorer.fpw")]      # converted by the first run
    res = subprocess.run(["bash", SCRIPT, "--data", root, "--out", out, "--reference-log", good] + fpw, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "PASS" in res.stdout, res.stdout + res.stderr
    res = subprocess.run(["bash", SCRIPT, "--data", root, "--out", out, "--reference-log", bad] + fpw, capture_output=True, text=True, timeout=900)
    assert res.returncode == 1 and "FAIL: pose gate" in res.stdout, res.stdout + res.stderr
    # a scorer file handed in as the refiner: the structural check stops the run before anything is converted
    res = subprocess.run(["bash", SCRIPT, "--refiner-onnx", onnx["scorer"], "--scorer-onnx", onnx["scorer"], "--data", root, "--out", str(tmp_path / "o2"),
                          "--reference-log", good], capture_output=True, text=True, timeout=900)
    assert res.returncode == 2 and "not the architecture" in res.stdout, res.stdout + res.stderr
