"""include/infer_core_amd.hpp: the deploy_core `inference_core` contract the reference's orchestrator drives
(GetBuffer / GetTensor / SetBufferLocation / RawPtr / SetShape / Cast / SyncInfer, D6F/src/foundationpose.cpp:126-139,
331-354,410-436) on top of the C ABI's fp_net_*.  CPU: the header compiles stand-alone and the error behaviour that needs no
GPU; GPU: a C++ program drives a refiner and a scorer core exactly like RefinePreProcess / ScorePreprocess do and its
outputs equal the Python API's on the same blobs."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include <cstring>
#include <fstream>
#include "infer_core_amd.hpp"
using namespace inference_core;
static std::vector<char> slurp(const std::string &p) { std::ifstream f(p, std::ios::binary); return {std::istreambuf_iterator<char>(f), {}}; }
int main(int argc, char **argv) {
  if (argc < 2) { std::printf("COMPILED\n"); return 0; }
  try {
    const std::string dir = argv[1];
    const uint64_t N = std::stoull(argv[2]);
    // simple_tests/src/test_foundationpose.cpp:24-35 with the factory name swapped
    auto refiner = CreateAmdInferCore(dir + "/r.fpw", {{"transf_input", {252, 160, 160, 6}}, {"render_input", {252, 160, 160, 6}}},
                                      {{"trans", {252, 3}}, {"rot", {252, 3}}}, 1);
    auto scorer = CreateAmdInferCore(dir + "/s.fpw", {{"transf_input", {252, 160, 160, 6}}, {"render_input", {252, 160, 160, 6}}},
                                     {{"scores", {252, 1}}}, 1);
    // foundationpose.cpp:126-139: the constructor's blob-name check
    auto rb = refiner->GetBuffer(true), sb = scorer->GetBuffer(true);
    rb->GetTensor("render_input"); rb->GetTensor("transf_input"); sb->GetTensor("render_input"); sb->GetTensor("transf_input");
    bool threw = false;
    try { rb->GetTensor("nope"); } catch (const std::exception &) { threw = true; }
    std::printf("UNKNOWN_BLOB_THROWS %d\n", (int)threw);
    auto a = slurp(dir + "/a.bin"), b = slurp(dir + "/b.bin");
    // RefinePreProcess (:331-354): inputs live on the device in the reference; here the host copies are filled and used
    for (auto *blobs : {rb.get(), sb.get()}) {
      blobs->GetTensor("render_input")->SetBufferLocation(DataLocation::HOST);
      blobs->GetTensor("transf_input")->SetBufferLocation(DataLocation::HOST);
      std::memcpy(blobs->GetTensor("render_input")->RawPtr(), a.data(), a.size());
      std::memcpy(blobs->GetTensor("transf_input")->RawPtr(), b.data(), b.size());
      blobs->GetTensor("render_input")->SetShape({N, 160, 160, 6});
      blobs->GetTensor("transf_input")->SetShape({N, 160, 160, 6});
    }
    if (!refiner->SyncInfer(rb.get())) { std::printf("refiner SyncInfer failed: %s\n", fp_last_error()); return 1; }   // :207
    const float *t = rb->GetTensor("trans")->Cast<float>(), *r = rb->GetTensor("rot")->Cast<float>();                 // :364-365
    std::printf("TRANS"); for (uint64_t i = 0; i < 3 * N; i++) std::printf(" %.9g", t[i]); std::printf("\n");
    std::printf("ROT"); for (uint64_t i = 0; i < 3 * N; i++) std::printf(" %.9g", r[i]); std::printf("\n");
    sb->GetTensor("scores")->SetBufferLocation(DataLocation::HOST);
    if (!scorer->SyncInfer(sb.get())) { std::printf("scorer SyncInfer failed: %s\n", fp_last_error()); return 1; }     // :219
    const float *s = sb->GetTensor("scores")->Cast<float>();                                                            // :436
    std::printf("SCORES"); for (uint64_t i = 0; i < N; i++) std::printf(" %.9g", s[i]); std::printf("\n");
    threw = false;
    try { rb->GetTensor("render_input")->SetShape({300, 160, 160, 6}); } catch (const std::exception &) { threw = true; }
    std::printf("OVERSIZE_THROWS %d\n", (int)threw);
    // the device copies are real device pointers
    rb->GetTensor("render_input")->SetBufferLocation(DataLocation::DEVICE);
    std::printf("DEVPTR %d\n", (int)(rb->GetTensor("render_input")->RawPtr() != nullptr));
  } catch (const std::exception &e) { std::printf("threw: %s\n", e.what()); return 3; }
  return 0;
}
'''


def _build(tmp_path):
    src = tmp_path / "core.cpp"
    src.write_text(SRC)
    exe = str(tmp_path / "core")
    libdir = os.path.join(ROOT, "foundationpose_cpp_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe, "-L", libdir,
                           "-lfoundationpose_amd", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_header_compiles_without_eigen_or_opencv(tmp_path):
    exe = _build(tmp_path)
    assert subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip() == "COMPILED"


@pytest.mark.gpu
def test_cores_driven_like_the_reference_orchestrator(tmp_path, syn_mesh, syn_scene):
    from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
    exe = _build(tmp_path)
    rp, sp = str(tmp_path / "r.fpw"), str(tmp_path / "s.fpw")
    W.pack_synthetic("refiner", rp)
    W.pack_synthetic("scorer", sp)
    m = FoundationPose(syn_mesh, syn.intrinsics(), rp, sp)
    try:
        N = 5
        m.upload_frame(syn_scene.rgb, syn_scene.depth)
        poses = m.get_hyp_poses(syn_scene.mask)[:N]
        a, b = m.render_and_transform(syn_mesh.name, poses, 1.2)
        a.tofile(tmp_path / "a.bin")
        b.tofile(tmp_path / "b.bin")
        out = subprocess.run([exe, str(tmp_path), str(N)], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = {l.split()[0]: l.split()[1:] for l in out.stdout.splitlines() if l}
        assert lines["UNKNOWN_BLOB_THROWS"] == ["1"] and lines["OVERSIZE_THROWS"] == ["1"] and lines["DEVPTR"] == ["1"]
        t, r = m.refiner_infer(a, b)
        s = m.scorer_infer(a, b)
        np.testing.assert_array_equal(np.array(lines["TRANS"], np.float32).reshape(N, 3), t)     # the same kernels on the same data
        np.testing.assert_array_equal(np.array(lines["ROT"], np.float32).reshape(N, 3), r)
        np.testing.assert_array_equal(np.array(lines["SCORES"], np.float32), s)
    finally:
        m.close()
