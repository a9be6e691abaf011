// fp_demo.cpp -- acceptance / demo harness on the dependency-free C++ wrapper.
//
// Mirrors the reference's simple_tests/src/test_foundationpose.cpp on the same dataset layout
// (test_data/download.md:6-15): `test` (:48-105) = Register the first frame with its mask, Track every following frame
// feeding the previous pose back, 3-D box overlay per frame; `speed_register` (:107-130) = 50 x Register with an
// FPSCounter; `speed_track` (:132-158) = Register once, 5000 x Track.  Instead of cv::imshow / an mp4 it writes
//   <out>/poses.txt          one line per frame: id + the 16 column-major floats (a numeric log the reference lacks)
//   <out>/<id>_plot.png      rgb with the green oriented-bounding-box overlay (first / last frame, or all with --plots)
//
// build:  g++ -std=c++17 -Iinclude examples/fp_demo.cpp -o fp_demo -Lfoundationpose_cpp_amd -lfoundationpose_amd
//             -Wl,-rpath,$PWD/foundationpose_cpp_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
// run:    ./fp_demo --data test_data/mustard0 --refiner refiner.fpw --scorer scorer.fpw --out out [--mode test]
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "foundationpose_amd.hpp"

namespace {

// tests/fps_counter.h equivalent: wall clock around the API calls (host upload included, like the reference's numbers)
struct FpsCounter {
  std::chrono::steady_clock::time_point t0;
  size_t n = 0;
  void Start() { t0 = std::chrono::steady_clock::now(); n = 0; }
  void Count(size_t k) { n += k; }
  double GetFPS() const { return n / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

std::vector<std::string> sorted_png_stems(const std::string &dir) {
  std::vector<std::string> out;
  if (DIR *d = opendir(dir.c_str())) {
    while (dirent *e = readdir(d)) {
      std::string n = e->d_name;
      if (n.size() > 4 && n.substr(n.size() - 4) == ".png") out.push_back(n.substr(0, n.size() - 4));
    }
    closedir(d);
  }
  std::sort(out.begin(), out.end());
  return out;
}

std::string first_obj(const std::string &dir) {
  std::vector<std::string> objs;
  if (DIR *d = opendir(dir.c_str())) {
    while (dirent *e = readdir(d)) {
      std::string n = e->d_name;
      if (n.size() > 4 && n.substr(n.size() - 4) == ".obj") objs.push_back(n);
    }
    closedir(d);
  }
  std::sort(objs.begin(), objs.end());
  return objs.empty() ? std::string() : dir + "/" + objs[0];
}

struct Frame {
  int H = 0, W = 0;
  std::vector<uint8_t> rgb, mask;
  std::vector<float> depth;
};

bool read_frame(const std::string &root, const std::string &id, int H, int W, bool with_mask, Frame &f) {
  f.H = H; f.W = W;
  f.rgb.resize((size_t)H * W * 3);
  f.depth.resize((size_t)H * W);
  if (with_mask) f.mask.resize((size_t)H * W);
  const std::string r = root + "/rgb/" + id + ".png", d = root + "/depth/" + id + ".png", m = root + "/masks/" + id + ".png";
  return fp_read_rgb_depth_mask(r.c_str(), d.c_str(), with_mask ? m.c_str() : nullptr, H, W, f.rgb.data(), f.depth.data(),
                                with_mask ? f.mask.data() : nullptr) == 0;
}

// ConvertPoseMesh2BBox (mesh_loader.hpp:75-81) on column-major arrays: pose * T(-centre) * orient_bounds
fp_amd::Pose pose_mesh2bbox(const fp_amd::Pose &pose, const float center[3], const float ob[16]) {
  auto mul = [](const float *A, const float *B, float *Cm) {
    for (int c = 0; c < 4; c++)
      for (int r = 0; r < 4; r++) {
        float s = 0;
        for (int k = 0; k < 4; k++) s += A[k * 4 + r] * B[c * 4 + k];
        Cm[c * 4 + r] = s;
      }
  };
  float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -center[0], -center[1], -center[2], 1}, tmp[16];
  fp_amd::Pose out;
  mul(pose.data(), T, tmp);
  mul(tmp, ob, out.data());
  return out;
}

const char *arg(int argc, char **argv, const char *key, const char *dflt) {
  for (int i = 1; i + 1 < argc; i++)
    if (!std::strcmp(argv[i], key)) return argv[i + 1];
  return dflt;
}
bool flag(int argc, char **argv, const char *key) {
  for (int i = 1; i < argc; i++)
    if (!std::strcmp(argv[i], key)) return true;
  return false;
}

}  // namespace

int main(int argc, char **argv) {
  const std::string root = arg(argc, argv, "--data", ""), out = arg(argc, argv, "--out", "fp_demo_out");
  const std::string refiner = arg(argc, argv, "--refiner", ""), scorer = arg(argc, argv, "--scorer", "");
  const std::string mode = arg(argc, argv, "--mode", "test"), name = arg(argc, argv, "--name", "mustard");
  const int refine_itr = std::atoi(arg(argc, argv, "--refine-itr", "1"));
  const int reps = std::atoi(arg(argc, argv, "--reps", mode == "speed_track" ? "5000" : "50"));
  if (root.empty() || refiner.empty() || scorer.empty()) {
    std::fprintf(stderr, "usage: fp_demo --data DIR --refiner R.fpw --scorer S.fpw [--mesh M.obj] [--out DIR] "
                         "[--mode test|speed_register|speed_track] [--reps N] [--refine-itr N] [--plots]\n");
    return 2;
  }
  float K[9];
  if (fp_read_cam_k((root + "/cam_K.txt").c_str(), K)) { std::fprintf(stderr, "%s\n", fp_last_error()); return 1; }
  std::string mesh_path = arg(argc, argv, "--mesh", "");
  if (mesh_path.empty()) mesh_path = first_obj(root + "/mesh");
  float ob[16], dim[3];
  std::vector<fp_amd::Mesh> meshes;
  try {
    meshes.push_back(fp_amd::LoadObjMesh(name, mesh_path, ob, dim));
  } catch (const std::exception &e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  const auto ids = sorted_png_stems(root + "/rgb");
  if (ids.empty()) { std::fprintf(stderr, "no frames under %s/rgb\n", root.c_str()); return 1; }
  int H = 0, W = 0;
  if (fp_frame_size((root + "/rgb/" + ids[0] + ".png").c_str(), &H, &W)) { std::fprintf(stderr, "%s\n", fp_last_error()); return 1; }

  std::unique_ptr<fp_amd::FoundationPose> fpm;
  try {
    fpm.reset(new fp_amd::FoundationPose(meshes, K, refiner, scorer, std::max(H, 1080), std::max(W, 1920)));
  } catch (const std::exception &e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  Frame f0;
  if (!read_frame(root, ids[0], H, W, true, f0)) { std::fprintf(stderr, "%s\n", fp_last_error()); return 1; }
  const fp_amd::ImageU8 rgb0{f0.rgb.data(), H, W, 3}, mask0{f0.mask.data(), H, W, 1};
  const fp_amd::ImageF32 depth0{f0.depth.data(), H, W};

  if (mode == "speed_register") {
    FpsCounter c;
    fp_amd::Pose p;
    fpm->Register(rgb0, depth0, mask0, name, p, refine_itr);  // warm-up (allocations)
    c.Start();
    for (int i = 0; i < reps; i++) {
      if (!fpm->Register(rgb0, depth0, mask0, name, p, refine_itr)) { std::fprintf(stderr, "%s\n", fpm->last_error().c_str()); return 1; }
      c.Count(1);
    }
    std::printf("average fps: %.3f  (Register, %d hypotheses -> %.1f hypotheses/s)\n", c.GetFPS(), fp_num_hypotheses(fpm->handle()),
                c.GetFPS() * fp_num_hypotheses(fpm->handle()));
    return 0;
  }
  fp_amd::Pose pose;
  if (!fpm->Register(rgb0, depth0, mask0, name, pose, refine_itr)) { std::fprintf(stderr, "%s\n", fpm->last_error().c_str()); return 1; }
  if (mode == "speed_track") {
    FpsCounter c;
    fp_amd::Pose tp;
    fpm->Track(rgb0, depth0, pose, name, tp, refine_itr);
    c.Start();
    for (int i = 0; i < reps; i++) {
      if (!fpm->Track(rgb0, depth0, pose, name, tp, refine_itr)) { std::fprintf(stderr, "%s\n", fpm->last_error().c_str()); return 1; }
      c.Count(1);
    }
    std::printf("average fps: %.3f  (Track)\n", c.GetFPS());
    return 0;
  }

  mkdir(out.c_str(), 0755);
  FILE *log = std::fopen((out + "/poses.txt").c_str(), "w");
  if (!log) { std::fprintf(stderr, "cannot write %s/poses.txt\n", out.c_str()); return 1; }
  const bool all_plots = flag(argc, argv, "--plots");
  auto emit = [&](const std::string &id, const fp_amd::Pose &p, Frame &f, bool plot) {
    std::fprintf(log, "%s", id.c_str());
    for (float v : p) std::fprintf(log, " %.9g", v);
    std::fprintf(log, "\n");
    if (plot) {
      fp_amd::Pose box = pose_mesh2bbox(p, meshes[0].center, ob);
      fp_draw_bbox3d(f.rgb.data(), H, W, K, box.data(), dim);
      fp_image_write_png_rgb((out + "/" + id + "_plot.png").c_str(), f.rgb.data(), H, W);
    }
  };
  std::printf("first Pose (%s): t = %.5f %.5f %.5f\n", ids[0].c_str(), pose[12], pose[13], pose[14]);
  emit(ids[0], pose, f0, true);
  FpsCounter c;
  c.Start();
  Frame f;
  for (size_t i = 1; i < ids.size(); i++) {
    if (!read_frame(root, ids[i], H, W, false, f)) { std::fprintf(stderr, "%s\n", fp_last_error()); return 1; }
    fp_amd::Pose tp;
    if (!fpm->Track({f.rgb.data(), H, W, 3}, {f.depth.data(), H, W}, pose, name, tp, refine_itr)) {
      std::fprintf(stderr, "%s\n", fpm->last_error().c_str());
      return 1;
    }
    c.Count(1);
    emit(ids[i], tp, f, all_plots || i + 1 == ids.size());
    pose = tp;  // out_pose = track_pose (test_foundationpose.cpp:101)
  }
  std::fclose(log);
  if (ids.size() > 1) std::printf("tracked %zu frames, %.1f fps including PNG decode\n", ids.size() - 1, c.GetFPS());
  std::printf("wrote %s/poses.txt\n", out.c_str());
  return 0;
}
