"""include/detection_6d_foundationpose_amd.hpp -- the reference's own API (detection_6d::CreateFoundationPoseModel,
Base6DofDetectionModel::Register / Track, BaseMeshLoader, ConvertPoseMesh2BBox; foundationpose.hpp:16-105,
mesh_loader.hpp:15-93) on top of the C ABI.  The header needs Eigen + OpenCV, which this image lacks, so it is compiled
here against TEST-ONLY minimal stand-ins (tests/mock_include) that provide just the members the header touches."""
import os
import subprocess

import numpy as np
import pytest

from foundationpose_cpp_amd import dataset as D, synthetic as syn, weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include <fstream>
#include <vector>
#include "detection_6d_foundationpose_amd.hpp"
using namespace detection_6d;
static std::vector<char> slurp(const std::string &p) { std::ifstream f(p, std::ios::binary); return {std::istreambuf_iterator<char>(f), {}}; }
int main(int argc, char **argv) {
  try {
    auto loader = CreateAssimpMeshLoader("mustard", argv[1]);            // test_foundationpose.cpp:37
    std::printf("MESH %zu %zu %.9g\n", loader->GetMeshNumVertices(), loader->GetMeshNumFaces(), loader->GetMeshDiameter());
    Eigen::Matrix4f eye = Eigen::Matrix4f::Identity();
    Eigen::Matrix4f box = ConvertPoseMesh2BBox(eye, loader);
    std::printf("BOX %.9g %.9g %.9g\n", box(0, 3), box(1, 3), box(2, 3));
    if (argc < 5) return 0;
    auto refiner = inference_core::CreateAmdInferCore(argv[2]);          // :24-35 with the two factory calls swapped
    auto scorer = inference_core::CreateAmdInferCore(argv[3]);
    Eigen::Matrix3f K;
    K(0, 0) = 320; K(0, 1) = 0; K(0, 2) = 320; K(1, 0) = 0; K(1, 1) = 320; K(1, 2) = 240; K(2, 0) = 0; K(2, 1) = 0; K(2, 2) = 1;
    auto model = CreateFoundationPoseModel(refiner, scorer, {loader}, K);  // :42-43
    std::string d = argv[4];
    auto rgb = slurp(d + "/rgb.bin"), depth = slurp(d + "/depth.bin"), mask = slurp(d + "/mask.bin");
    cv::Mat mrgb(480, 640, CV_8UC3, rgb.data()), mdepth(480, 640, CV_32FC1, depth.data()), mmask(480, 640, CV_8UC1, mask.data());
    Eigen::Matrix4f pose, tracked;
    if (!model->Register(mrgb, mdepth, mmask, "mustard", pose)) { std::printf("Register failed\n"); return 1; }   // :62
    if (!model->Track(mrgb, mdepth, pose, "mustard", tracked)) { std::printf("Track failed\n"); return 1; }       // :89
    std::printf("POSE"); for (int i = 0; i < 16; i++) std::printf(" %.9g", pose.data()[i]); std::printf("\n");
    std::printf("TRACK"); for (int i = 0; i < 16; i++) std::printf(" %.9g", tracked.data()[i]); std::printf("\n");
    Eigen::Matrix4f bad;
    std::printf("UNKNOWN %d\n", (int)model->Register(mrgb, mdepth, mmask, "nope", bad));
  } catch (const std::exception &e) { std::printf("threw: %s\n", e.what()); return 3; }
  return 0;
}
'''


def _build(tmp_path):
    src = tmp_path / "shim.cpp"
    src.write_text(SRC)
    exe = str(tmp_path / "shim")
    libdir = os.path.join(ROOT, "foundationpose_cpp_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "tests", "mock_include"), "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", exe, "-L", libdir, "-lfoundationpose_amd", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib",
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_shim_compiles_and_its_mesh_loader_matches(tmp_path):
    mesh = syn.make_mesh(subdiv=2, offset=(0.01, -0.02, 0.03))
    obj = D.write_obj(str(tmp_path / "mesh"), mesh)
    res = subprocess.run([_build(tmp_path), obj], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    tok = dict(l.split(None, 1) for l in res.stdout.strip().splitlines())
    v, f, diam = tok["MESH"].split()
    assert int(v) == len(mesh.vertices) and int(f) == len(mesh.faces) and abs(float(diam) - mesh.diameter) < 1e-6
    from foundationpose_cpp_amd import load_mesh
    m = load_mesh("mustard", obj)
    ref = D.convert_pose_mesh2bbox(np.eye(4, dtype=np.float32), m)[:3, 3]     # mesh_loader.hpp:75-81
    np.testing.assert_allclose([float(x) for x in tok["BOX"].split()], ref, atol=1e-6)
    # error behaviour of the factory: throws like the reference (assimp_mesh_loader.cpp:162-185)
    res = subprocess.run([_build(tmp_path), str(tmp_path / "missing.obj")], capture_output=True, text=True, timeout=120)
    assert res.returncode == 3 and "Failed to read mesh file" in res.stdout


@pytest.mark.gpu
def test_shim_register_and_track_equal_the_c_abi(tmp_path, syn_scene):
    from foundationpose_cpp_amd import FoundationPose, load_mesh
    mesh0 = syn.make_mesh()
    obj = D.write_obj(str(tmp_path / "mesh"), mesh0)
    rp, sp = str(tmp_path / "r.fpw"), str(tmp_path / "s.fpw")
    W.pack_synthetic("refiner", rp)
    W.pack_synthetic("scorer", sp)
    syn_scene.rgb.tofile(tmp_path / "rgb.bin")
    syn_scene.depth.astype(np.float32).tofile(tmp_path / "depth.bin")
    syn_scene.mask.tofile(tmp_path / "mask.bin")
    res = subprocess.run([_build(tmp_path), obj, rp, sp, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    tok = dict(l.split(None, 1) for l in res.stdout.strip().splitlines())
    pose = syn.from_colmajor(np.array([float(x) for x in tok["POSE"].split()], np.float32)[None])[0]
    tracked = syn.from_colmajor(np.array([float(x) for x in tok["TRACK"].split()], np.float32)[None])[0]
    assert tok["UNKNOWN"].strip() == "0"
    mesh = load_mesh("mustard", obj)
    m = FoundationPose(mesh, syn.intrinsics(), rp, sp)
    ok, ref = m.Register(syn_scene.rgb, syn_scene.depth, syn_scene.mask, "mustard")
    ok2, tref = m.Track(syn_scene.rgb, syn_scene.depth, ref, "mustard")
    assert ok and ok2
    np.testing.assert_allclose(pose, ref, rtol=0, atol=1e-7)
    np.testing.assert_allclose(tracked, tref, rtol=0, atol=1e-7)
    m.close()
