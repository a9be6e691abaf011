// fp_demo_mgpu.cpp -- the sharded Register natively: ONE process, one host thread per GPU, RCCL over xGMI, no Python / torch.
//
// the N pose hypotheses (252 at Register) are the data-parallel axis.  Every rank owns the frame, the mesh and
// the weights and a contiguous slice of the hypothesis grid; the only exchange is ONE ncclAllGather of a [512 + 16]-float row per
// hypothesis, issued by fp_register_sharded on the model's own stream, after which every rank evaluates the cross-hypothesis head and
// the arg-max redundantly.  This program
//   * creates one model per rank with fp_create_on(device) and one communicator per rank with ncclCommInitAll,
//   * runs the sharded Register from `ranks` threads at once and checks that every rank returns the same pose and winner,
//   * compares with the unsharded fp_register of rank 0's model (same winner; pose equal to the last bit for the default precision),
//   * times `--reps` sharded Registers (hypotheses/s over all ranks).
// RCCL cannot place two ranks of one communicator on one device, so `--ranks` must not exceed the GPU count; `--ranks 1` runs the
// whole path (communicator, persistent exchange buffers, packed begin / finish) on a single GPU.
//
// build:  hipcc -std=c++17 -Iinclude examples/fp_demo_mgpu.cpp -o fp_demo_mgpu -Lfoundationpose_cpp_amd -lfoundationpose_amd \
//               -Wl,-rpath,$PWD/foundationpose_cpp_amd -lrccl -lpthread
// run:    ./fp_demo_mgpu --data test_data/mustard0 --refiner refiner.fpw --scorer scorer.fpw [--ranks 8] [--hyps 252|1008] [--reps 20]
#include <dirent.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "foundationpose_amd.h"

namespace {
std::string first_with_ext(const std::string &dir, const char *ext) {
  std::vector<std::string> names;
  if (DIR *d = opendir(dir.c_str())) {
    while (dirent *e = readdir(d)) {
      std::string n = e->d_name;
      if (n.size() > 4 && n.substr(n.size() - 4) == ext) names.push_back(n);
    }
    closedir(d);
  }
  std::sort(names.begin(), names.end());
  return names.empty() ? std::string() : dir + "/" + names[0];
}
// a reusable barrier for the rank threads (C++17: no std::barrier)
struct Barrier {
  explicit Barrier(int n) : n_(n) {}
  void wait() {
    const int gen = gen_.load();
    if (count_.fetch_add(1) + 1 == n_) { count_.store(0); gen_.fetch_add(1); }
    else while (gen_.load() == gen) std::this_thread::yield();
  }
  int n_;
  std::atomic<int> count_{0}, gen_{0};
};
}  // namespace

int main(int argc, char **argv) {
  std::string data, refiner, scorer;
  int ranks = 0, hyps = 252, reps = 10;
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string k = argv[i], v = argv[i + 1];
    if (k == "--data") data = v;
    else if (k == "--refiner") refiner = v;
    else if (k == "--scorer") scorer = v;
    else if (k == "--ranks") ranks = std::atoi(v.c_str());
    else if (k == "--hyps") hyps = std::atoi(v.c_str());
    else if (k == "--reps") reps = std::atoi(v.c_str());
  }
  if (data.empty() || refiner.empty() || scorer.empty() || hyps % 42 != 0) {
    std::fprintf(stderr, "usage: fp_demo_mgpu --data DIR --refiner R.fpw --scorer S.fpw [--ranks N] [--hyps 252] [--reps 10]\n");
    return 2;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::fprintf(stderr, "no HIP device\n"); return 1; }
  if (ranks <= 0) ranks = ndev;

  // the dataset of the reference's demo (test_data/download.md:6-15): first frame + mask, cam_K.txt, mesh/*.obj
  const std::string rgb_path = first_with_ext(data + "/rgb", ".png"), stem = rgb_path.substr(rgb_path.rfind('/') + 1);
  float K[9];
  int H = 0, W = 0;
  if (fp_read_cam_k((data + "/cam_K.txt").c_str(), K) || fp_frame_size(rgb_path.c_str(), &H, &W)) { std::fprintf(stderr, "%s\n", fp_last_error()); return 1; }
  std::vector<uint8_t> rgb((size_t)H * W * 3), mask((size_t)H * W);
  std::vector<float> depth((size_t)H * W);
  if (fp_read_rgb_depth_mask(rgb_path.c_str(), (data + "/depth/" + stem).c_str(), (data + "/masks/" + stem).c_str(), H, W, rgb.data(), depth.data(), mask.data())) {
    std::fprintf(stderr, "%s\n", fp_last_error());
    return 1;
  }
  fp_loaded_mesh *lm = fp_mesh_load_obj("object", first_with_ext(data + "/mesh", ".obj").c_str());
  if (!lm) { std::fprintf(stderr, "%s\n", fp_last_error()); return 1; }

  // one communicator per rank (ncclCommInitAll: single process, one rank per device)
  std::vector<int> devs(ranks);
  for (int r = 0; r < ranks; r++) devs[r] = r % ndev;
  std::vector<ncclComm_t> comms(ranks, nullptr);
  const bool distinct = ranks <= ndev;
  if (distinct && ranks > 1 && ncclCommInitAll(comms.data(), ranks, devs.data()) != ncclSuccess) { std::fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }
  if (!distinct && ranks > 1) { std::fprintf(stderr, "%d ranks on %d device(s): RCCL cannot put two ranks of one communicator on one device\n", ranks, ndev); return 2; }
  if (ranks == 1) {
    ncclUniqueId id;
    (void)hipSetDevice(devs[0]);
    if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&comms[0], 1, id, 0) != ncclSuccess) { std::fprintf(stderr, "ncclCommInitRank failed\n"); return 1; }
  }

  std::vector<fp_model *> models(ranks, nullptr);
  std::vector<std::vector<float>> poses(ranks, std::vector<float>(16, 0.f));
  std::vector<int> winners(ranks, -1), rcs(ranks, 0);
  std::vector<std::string> errors(ranks);
  Barrier bar(ranks);
  double seconds = 0;
  auto worker = [&](int r) {
    // (fp_create_on makes the device current itself; nothing here calls hipSetDevice)
    models[r] = fp_create_on(devs[r], fp_mesh_view(lm), 1, K, refiner.c_str(), scorer.c_str(), 0, 0);
    if (!models[r] || fp_set_inplane_steps(models[r], hyps / 42)) { rcs[r] = 1; errors[r] = fp_last_error(); }
    bar.wait();
    for (int r2 = 0; r2 < ranks; r2++) if (rcs[r2]) return;   // (every thread sees the same verdict after the barrier)
    for (int it = 0; it < 2 && !rcs[r]; it++)   // eager call, then the captured graph
      rcs[r] = fp_register_sharded(models[r], comms[r], rgb.data(), depth.data(), mask.data(), FP_HOST, H, W, "object", 1, poses[r].data(), &winners[r]);
    if (rcs[r]) errors[r] = fp_last_error();
    bar.wait();
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<float> p(16);
    for (int it = 0; it < reps && !rcs[r]; it++)
      rcs[r] = fp_register_sharded(models[r], comms[r], rgb.data(), depth.data(), mask.data(), FP_HOST, H, W, "object", 1, p.data(), nullptr);
    bar.wait();
    if (r == 0) seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  std::vector<std::thread> threads;
  for (int r = 0; r < ranks; r++) threads.emplace_back(worker, r);
  for (auto &t : threads) t.join();
  int rc = 0;
  for (int r = 0; r < ranks; r++)
    if (rcs[r]) { std::fprintf(stderr, "rank %d: %s\n", r, errors[r].c_str()); rc = 1; }
  if (!rc) {
    for (int r = 1; r < ranks; r++)
      if (winners[r] != winners[0] || std::memcmp(poses[r].data(), poses[0].data(), 64) != 0) { std::fprintf(stderr, "rank %d disagrees with rank 0\n", r); rc = 1; }
    float ref[16];
    if (fp_register(models[0], rgb.data(), depth.data(), mask.data(), H, W, "object", 1, ref)) { std::fprintf(stderr, "%s\n", fp_last_error()); rc = 1; }
    float dmax = 0;
    for (int i = 0; i < 16; i++) dmax = std::max(dmax, std::fabs(ref[i] - poses[0][i]));
    std::printf("ranks %d (devices %d)  hypotheses %d  winner %d  max |sharded - unsharded| pose element %.3g\n", ranks, std::min(ranks, ndev), hyps, winners[0], dmax);
    std::printf("pose");
    for (int i = 0; i < 16; i++) std::printf(" %.9g", poses[0][i]);
    std::printf("\n%d sharded Registers: %.3f ms each, %.1f hypotheses/s over %d rank(s)\n", reps, seconds / reps * 1e3, hyps * reps / seconds, ranks);
    if (dmax > 1e-3f) { std::fprintf(stderr, "sharded and unsharded Register disagree\n"); rc = 1; }
  }
  for (int r = 0; r < ranks; r++) {
    if (models[r]) fp_destroy(models[r]);
    if (comms[r]) ncclCommDestroy(comms[r]);
  }
  fp_mesh_free(lm);
  return rc;
}
