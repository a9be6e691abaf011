"""Acceptance harness: the C++ demo (examples/fp_demo.cpp, the twin of the reference's
simple_tests/src/test_foundationpose.cpp:48-105) on a synthetic sequence in the reference's dataset layout:
  * its pose log equals the Python API's on the same files (same library, same decode -> bit-identical);
  * every tracked pose equals the CPU oracle's Track from the previous pose within 0.1 degree / 0.1 mm
    (north-star bar: 1 degree / 1 mm)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from foundationpose_cpp_amd import dataset as D, load_mesh, synthetic as syn, weights as W
from oracle import fp_oracle as fo
from oracle import nets_torch as NT

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pose_err(a, b):
    dR = a[:3, :3] @ b[:3, :3].T
    return np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))), np.linalg.norm(a[:3, 3] - b[:3, 3])


def test_cpp_demo_sequence_matches_python_and_oracle(tmp_path):
    root = str(tmp_path / "synthetic0")
    D.write_synthetic_sequence(root, n_frames=5)
    rp, sp = str(tmp_path / "r.fpw"), str(tmp_path / "s.fpw")
    rs = W.pack_synthetic("refiner", rp)
    W.pack_synthetic("scorer", sp)
    exe = str(tmp_path / "fp_demo")
    libdir = os.path.join(ROOT, "foundationpose_cpp_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "fp_demo.cpp"),
                           "-o", exe, "-L", libdir, "-lfoundationpose_amd", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib",
                           "-Wl,-rpath,/opt/rocm/lib"])
    out_c = str(tmp_path / "out_c")
    res = subprocess.run([exe, "--data", root, "--refiner", rp, "--scorer", sp, "--out", out_c, "--plots"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    rows = [l.split() for l in open(os.path.join(out_c, "poses.txt"))]
    assert len(rows) == 5
    poses_c = syn.from_colmajor(np.array([[float(v) for v in r[1:]] for r in rows], np.float32))

    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import demo_sequence
    poses_py = demo_sequence.run(root, rp, sp, str(tmp_path / "out_py"))
    np.testing.assert_allclose(poses_c, poses_py, rtol=0, atol=1e-7)        # %.9g round trip of identical results

    seq = D.Sequence(root)
    assert [r[0] for r in rows] == seq.ids
    mesh = load_mesh("mustard", seq.mesh_path())
    om = fo.OracleMesh(mesh)
    refiner = NT.build("refiner", rs)
    for i in range(1, 5):
        rgb, depth = seq.frame(i)
        prev = syn.to_colmajor(poses_c[i - 1][None])
        a = fo.render(om, prev, seq.K, (seq.H, seq.W), 1.2)
        b = fo.crop(rgb, depth, seq.K, prev, 1.2, mesh.diameter)
        with torch.no_grad():
            t, r = refiner(torch.from_numpy(a), torch.from_numpy(b))
        ref = syn.from_colmajor(fo.refine_post_process(prev, t.numpy(), r.numpy(), mesh.diameter))[0]
        ang, dist = _pose_err(poses_c[i], ref)
        assert ang < 0.1 and dist < 1e-4, (i, ang, dist)
    # overlays exist and contain the green box
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(out_c, seq.ids[0] + "_plot.png")))
    assert ((img == np.array([0, 255, 0], np.uint8)).all(-1)).sum() > 200
    # speed modes run (tiny rep counts)
    for mode in ("speed_register", "speed_track"):
        res = subprocess.run([exe, "--data", root, "--refiner", rp, "--scorer", sp, "--mode", mode, "--reps", "3"],
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0 and "average fps" in res.stdout, res.stdout + res.stderr


def test_native_sharded_register_demo(tmp_path):
    """examples/fp_demo_mgpu.cpp: fp_create_on + fp_register_sharded (the library issues the ncclAllGather itself, RCCL bound with dlopen)
    in a plain C++ process -- no torch, no Python.  One rank here (the GPU box has one device): the communicator, the persistent
    exchange buffers and the packed begin / finish halves all run; the result must equal the unsharded fp_register and the Python API's."""
    from foundationpose_cpp_amd import FoundationPose
    root = str(tmp_path / "synthetic0")
    D.write_synthetic_sequence(root, n_frames=1)
    rp, sp = str(tmp_path / "r.fpw"), str(tmp_path / "s.fpw")
    W.pack_synthetic("refiner", rp)
    W.pack_synthetic("scorer", sp)
    exe = str(tmp_path / "fp_demo_mgpu")
    libdir = os.path.join(ROOT, "foundationpose_cpp_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-w", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "fp_demo_mgpu.cpp"),
                           "-o", exe, "-L", libdir, "-lfoundationpose_amd", f"-Wl,-rpath,{libdir}", "-lrccl", "-lpthread"])
    res = subprocess.run([exe, "--data", root, "--refiner", rp, "--scorer", sp, "--ranks", "1", "--reps", "3"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    print(res.stdout)
    line = [l for l in res.stdout.splitlines() if l.startswith("pose ")][0]
    pose_c = syn.from_colmajor(np.array([float(v) for v in line.split()[1:]], np.float32)[None])[0]
    assert "max |sharded - unsharded| pose element 0" in res.stdout          # bit-identical to fp_register in the same process
    seq = D.Sequence(root)
    mesh = load_mesh("object", seq.mesh_path())
    m = FoundationPose(mesh, seq.K, rp, sp, device=0)
    try:
        assert m.device == 0
        rgb, depth, mask = seq.frame(0, with_mask=True)
        ok, pose_py = m.Register(rgb, depth, mask, mesh.name)
        assert ok, m.last_error
        np.testing.assert_allclose(pose_c, pose_py, rtol=0, atol=1e-6)
    finally:
        m.close()
