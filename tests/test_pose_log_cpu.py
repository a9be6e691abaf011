"""tools/compare_pose_log.py: both log formats (fp_demo's poses.txt, the reference's glog / Eigen print of
simple_tests/src/test_foundationpose.cpp:62,89), the 1 deg / 1 mm gate and the exit codes."""
import os
import subprocess
import sys

import numpy as np

from foundationpose_cpp_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "compare_pose_log.py")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import compare_pose_log as CPL  # noqa: E402


def _poses(n, seed=0):
    return np.stack([syn.pose_matrix(syn.random_rotation(seed + i), [0.01 * i, -0.02, 0.7 + 0.001 * i]) for i in range(n)])


def _write_ours(path, poses):
    with open(path, "w") as f:
        for i, p in enumerate(poses):
            f.write(f"{i:06d} " + " ".join(f"{x:.9g}" for x in syn.to_colmajor(p[None].astype(np.float32))[0]) + "\n")


def _write_reference(path, poses):
    """what glog + Eigen's operator<< print: label and first row on one line, three more rows below"""
    with open(path, "w") as f:
        f.write("I0523 10:00:00.000000 12 test_foundationpose.cpp:40] some other line 1 2 3 4\n")
        for i, p in enumerate(poses):
            label = "first Pose : " if i == 0 else "Track pose : "
            rows = ["  ".join(f"{x:.6g}" for x in r) for r in p]
            f.write(f"W0523 10:00:0{i % 10}.123456 12 test_foundationpose.cpp:{62 if i == 0 else 89}] {label}{rows[0]}\n")
            f.write("\n".join(" " + r for r in rows[1:]) + "\n")


def test_formats_gate_and_exit_codes(tmp_path):
    poses = _poses(6)
    a, b = str(tmp_path / "poses.txt"), str(tmp_path / "ref.log")
    _write_ours(a, poses)
    _write_reference(b, poses)
    ids, pa = CPL.parse_pose_log(a)
    idb, pb = CPL.parse_pose_log(b)
    assert ids[0] == "000000" and idb[0] == "register#0" and idb[1] == "track#1"
    np.testing.assert_allclose(pa, poses, atol=1e-6)
    np.testing.assert_allclose(pb, poses, atol=1e-5)
    r = subprocess.run([sys.executable, TOOL, a, b], capture_output=True, text=True)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr
    # frame 3 off by 1.5 degrees, frame 4 by 2 mm
    off = poses.copy()
    off[3] = syn.perturb_pose(poses[3].astype(np.float32), deg=1.5, trans=0.0, seed=1)
    off[4, 2, 3] += 0.002
    _write_reference(b, off)
    ang, dist = CPL.pose_errors(*[CPL.parse_pose_log(x)[1] for x in (a, b)])
    assert abs(ang[3] - 1.5) < 1e-3 and abs(dist[4] - 0.002) < 1e-6 and ang[0] < 1e-3
    r = subprocess.run([sys.executable, TOOL, a, b, "--rot-deg", "1", "--trans-mm", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "FAIL (2 frames)" in r.stdout and r.stdout.count("exceeds the gate") == 2
    r = subprocess.run([sys.executable, TOOL, a, b, "--rot-deg", "2", "--trans-mm", "3", "--quiet"], capture_output=True, text=True)
    assert r.returncode == 0
    # different lengths / garbage -> exit 2
    _write_ours(a, poses[:5])
    assert subprocess.run([sys.executable, TOOL, a, b], capture_output=True).returncode == 2
    open(a, "w").write("000000 1 2 3\n")
    assert subprocess.run([sys.executable, TOOL, a, b], capture_output=True).returncode == 2
