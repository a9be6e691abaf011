// fp_api.hip -- the C ABI (include/foundationpose_amd.h) and the host-side orchestration that mirrors
// detection_6d::FoundationPose (reference D6F/src/foundationpose.cpp:108-458), MI355X-first:
//   * all per-hypothesis host work of the reference (crop windows, bbox, projection, pose update, arg-max) runs
//     on the device, so a Register is one stream of launches with a single D2H of the winning pose;
//   * render/crop kernels write the networks' fp16 input tensor directly (no fp32 blob, no concat pass);
//   * device buffers are allocated once per model and grown on demand (no per-call hipMalloc).
#include "../../include/foundationpose_amd.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <cstring>
#include <dlfcn.h>
#include <link.h>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>

#include "fp_internal.h"
#include "fp_nn.h"

namespace fp {

static thread_local std::string g_last_error;
// Concurrency contract: a model is NOT re-entrant (like the reference, foundationpose.cpp:103-105: one renderer and its
// scratch per target), but different models may be driven from different host threads at the same time, each on its own
// non-blocking stream; their kernels overlap on the GPU.  (Round 1 serialised all models behind a process-wide lock to
// hide wrong results under exactly that overlap; the cause -- packed-f32 VALU instructions returning wrong values in lanes
// 48-63 while waves of another queue's kernel share the SIMD -- and the fix are in DESIGN.md section 9.)
void set_error(const std::string &msg) { g_last_error = msg; }

// One non-blocking utility stream per DEVICE, shared by all host threads under a lock (creation / loading / calibration paths only:
// nothing on the serving path uses it).  Per device, not per thread: a stream belongs to the device that was current when it was
// created, and a stream made by a transient host thread would outlive that thread inside the runtime.
static std::mutex g_util_mu;
static hipError_t utility_stream(hipStream_t *out) {
  static hipStream_t streams[64] = {nullptr};
  int d = 0;
  hipError_t e = hipGetDevice(&d);
  if (e != hipSuccess) return e;
  if (d < 0 || d >= 64) return hipErrorInvalidDevice;
  if (!streams[d] && (e = hipStreamCreateWithFlags(&streams[d], hipStreamNonBlocking)) != hipSuccess) { streams[d] = nullptr; return e; }
  *out = streams[d];
  return hipSuccess;
}
// Host <-> device copies of the creation / loading / calibration paths go through a PINNED staging buffer (two 4 MB halves,
// double-buffered) instead of handing the caller's pageable memory to hipMemcpyAsync: the runtime never has to lock or stage user
// pages, and it is the faster copy.  (Also tried against the fp_create SIGSEGV of DESIGN.md section 9; the fault is elsewhere.)
static hipError_t staging(unsigned char **buf, size_t *half) {
  static unsigned char *p = nullptr;   // (under g_util_mu)
  constexpr size_t HALF = (size_t)4 << 20;
  if (!p) {
    const hipError_t e = hipHostMalloc((void **)&p, 2 * HALF, hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) { p = nullptr; return e; }
  }
  *buf = p; *half = HALF;
  return hipSuccess;
}
// Host -> device through a KERNEL that reads the pinned, device-mapped staging block [r5]: no copy command on the creation / loading /
// calibration paths.  (The one native backtrace of the fp_create crash under a concurrent PyTorch stream -- DESIGN.md section 9 --
// ends inside hipMemcpyAsync's signal wait in the HIP 7.0 runtime PyTorch bundles; a kernel launch does not take that path.  The same
// trick carries Track's window upload, window_fetch_kernel.)
__global__ __launch_bounds__(256) void staged_upload_kernel(const unsigned char *__restrict__ src_mapped, unsigned char *__restrict__ dst, size_t n) {
  const size_t n16 = n / 16;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src_mapped)[i];
  if (blockIdx.x == 0 && threadIdx.x < (n & 15)) dst[n16 * 16 + threadIdx.x] = src_mapped[n16 * 16 + threadIdx.x];
}
// [r6] the Register's read-back {winner index, sampler status, pose} written straight into the model's pinned, device-mapped result
// block: one 18-lane kernel instead of three device -> host copy commands in front of the call's only synchronisation
__global__ void publish_result_kernel(const int *__restrict__ argmax, const int *__restrict__ sampler_status, const float *__restrict__ best_pose,
                                      int *__restrict__ out_mapped) {
  const int i = threadIdx.x;
  int v = 0;
  if (i == 0) v = argmax[0];
  else if (i == 1) v = sampler_status ? sampler_status[0] : 0;
  else if (i < 18) v = __float_as_int(best_pose[i - 2]);
  if (i < 18) out_mapped[i] = v;
  __threadfence_system();
}
hipError_t memcpy_sync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  std::lock_guard<std::mutex> lk(g_util_mu);
  hipStream_t s = nullptr;
  hipError_t e = utility_stream(&s);
  if (e != hipSuccess) return e;
  if (kind != hipMemcpyHostToDevice && kind != hipMemcpyDeviceToHost) {
    e = hipMemcpyAsync(dst, src, bytes, kind, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
  }
  unsigned char *stage = nullptr;
  size_t half = 0;
  if ((e = staging(&stage, &half)) != hipSuccess) return e;
  hipEvent_t done[2] = {nullptr, nullptr};
  for (auto &ev : done) if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) { if (done[0]) (void)hipEventDestroy(done[0]); return e; }
  size_t off = 0;
  int slot = 0;
  bool used[2] = {false, false};
  while (off < bytes && e == hipSuccess) {
    const size_t n = std::min(half, bytes - off);
    unsigned char *h = stage + slot * half;
    if (used[slot]) e = hipEventSynchronize(done[slot]);   // the copy that last used this half has finished
    if (e != hipSuccess) break;
    if (kind == hipMemcpyHostToDevice) {
      std::memcpy(h, (const unsigned char *)src + off, n);
      unsigned char *h_dev = nullptr;
      if (((uintptr_t)((unsigned char *)dst + off) & 15) == 0 && hipHostGetDevicePointer((void **)&h_dev, h, 0) == hipSuccess) {
        const unsigned blocks = (unsigned)std::min<size_t>(1024, (n / 16 + 255) / 256 + 1);
        hipLaunchKernelGGL(staged_upload_kernel, dim3(blocks), dim3(256), 0, s, h_dev, (unsigned char *)dst + off, n);
        e = hipGetLastError();
      } else e = hipMemcpyAsync((unsigned char *)dst + off, h, n, hipMemcpyHostToDevice, s);
      if (e == hipSuccess) e = hipEventRecord(done[slot], s);
      used[slot] = true;
    } else {   // device -> host: wait for the chunk, then hand it over
      e = hipMemcpyAsync(h, (const unsigned char *)src + off, n, hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e == hipSuccess) std::memcpy((unsigned char *)dst + off, h, n);
    }
    off += n;
    slot ^= 1;
  }
  const hipError_t e2 = hipStreamSynchronize(s);
  for (auto &ev : done) (void)hipEventDestroy(ev);
  return e != hipSuccess ? e : e2;
}
hipError_t memset_sync(void *dst, int value, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_util_mu);
  hipStream_t s = nullptr;
  hipError_t e = utility_stream(&s);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(dst, value, bytes, s);
  return e != hipSuccess ? e : hipStreamSynchronize(s);
}
std::atomic<unsigned long> g_alloc_epoch{0};

// ---- streams are RECYCLED, never destroyed [r4]: destroyed models park their stream here, new models take one.  (Tried as a remedy for
// the fp_create SIGSEGV under a concurrent PyTorch stream -- a signal wait inside the HIP 7.0 runtime PyTorch bundles, DESIGN.md
// section 9 -- where it changed nothing; kept because a serving process that adds and removes objects should not churn HSA queues.)
namespace {
std::mutex g_stream_pool_mu;
std::vector<std::pair<int, hipStream_t>> g_stream_pool;   // (device, stream)
}  // namespace
hipError_t stream_acquire(hipStream_t *out) {
  int d = 0;
  hipError_t e = hipGetDevice(&d);
  if (e != hipSuccess) return e;
  {
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    for (size_t i = 0; i < g_stream_pool.size(); i++)
      if (g_stream_pool[i].first == d) {
        *out = g_stream_pool[i].second;
        g_stream_pool.erase(g_stream_pool.begin() + (long)i);
        return hipSuccess;
      }
  }
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);   // non-blocking: no implicit synchronisation with the legacy null stream
}
void stream_release(hipStream_t s) {   // (the caller has synchronised it; its device is current)
  int d = 0;
  if (!s || hipGetDevice(&d) != hipSuccess) return;
  std::lock_guard<std::mutex> lk(g_stream_pool_mu);
  g_stream_pool.emplace_back(d, s);
}

// ---- roctx (fp_internal.h): bound once, no-ops when the library is not there
namespace {
struct RoctxApi {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
};
const RoctxApi &roctx_api() {
  static const RoctxApi api = [] {
    RoctxApi a;
    void *h = nullptr;
    for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"})
      if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) {
      a.push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
      a.pop = (int (*)())dlsym(h, "roctxRangePop");
      if (!a.push || !a.pop) a.push = nullptr, a.pop = nullptr;
    }
    return a;
  }();
  return api;
}
}  // namespace
void roctx_push(const char *name) { if (roctx_api().push) (void)roctx_api().push(name); }
void roctx_pop() { if (roctx_api().pop) (void)roctx_api().pop(); }
#ifdef FP_TEST_HOOKS
static int g_upload_cols = 1;   // A/B (fpt_set_upload_cols): Track from host frames uploads the crop window's rectangle, not whole rows (1: packed into pinned memory with its frame record and fetched by a kernel; 3: the same with one 1-D copy command; 2: two 2-D copies straight from the caller's pageable frame)
#else
static constexpr int g_upload_cols = 1;
#endif
#ifdef FP_TEST_HOOKS
static int g_calib_sweeps = 2, g_calib_tok = 1, g_calib_out = 1;   // A/B (fpt_set_calib_opts): steps of the 8-bit calibration
static int g_vertex_crop = 1;   // A/B (fpt_set_vertex_crop): Track's crop warp (and, unless 2, the triangles' row ranges) inside the vertex launch
static int g_tri_rows_all_batches = 0;   // A/B (fpt_set_tri_rows(2)): size the buffer for large batches too
static int g_tri_rows = 1;      // A/B (fpt_set_tri_rows): per-triangle row ranges, so that a strip of the rasteriser skips the triangles that miss it
#else
static constexpr int g_calib_sweeps = 2, g_calib_tok = 1, g_calib_out = 1;
static constexpr int g_vertex_crop = 1;
static constexpr int g_tri_rows = 1;
static constexpr int g_tri_rows_all_batches = 0;
#endif

// ------------------------------------------------------------------------------------------------
// Profiler
// ------------------------------------------------------------------------------------------------
ProfEntry &Profiler::get(const char *name) {
  for (auto &e : entries)
    if (e.name == name) return e;
  entries.emplace_back();
  entries.back().name = name;
  return entries.back();
}
hipEvent_t Profiler::ev() {
  if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
void Profiler::begin(hipStream_t s, const char *name, double flops, double bytes) {
  // entries is a vector: take the index, pointers may move
  ProfEntry &e = get(name);
  e.calls++; e.flops += flops; e.bytes += bytes;
  cur = &e;
  cur_start = ev();
  (void)hipEventRecord(cur_start, s);
}
void Profiler::end(hipStream_t s) {
  hipEvent_t stop = ev();
  (void)hipEventRecord(stop, s);
  cur->pending.emplace_back(cur_start, stop);
  cur = nullptr;
}
void Profiler::collect() {
  for (auto &e : entries) {
    for (auto &pr : e.pending) {
      (void)hipEventSynchronize(pr.second);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, pr.first, pr.second);
      e.ms += ms;
      pool.push_back(pr.first);
      pool.push_back(pr.second);
    }
    e.pending.clear();
  }
}
void Profiler::reset() {
  collect();
  entries.clear();
}

// ------------------------------------------------------------------------------------------------
// a7: rotation grid (D6F/src/foundationpose_sampling.cpp:56-121,178-237); 42 icosphere views x inplane_steps
// ------------------------------------------------------------------------------------------------
namespace {
struct V3 { float x, y, z; };
V3 normalized(V3 p) {
  float n2 = p.x * p.x + p.y * p.y + p.z * p.z;
  if (n2 > 0.0f) { float n = std::sqrt(n2); p.x /= n; p.y /= n; p.z /= n; }
  return p;
}
std::vector<V3> icosphere(unsigned min_views) {
  std::vector<V3> verts;
  std::vector<std::array<int, 3>> faces;
  std::map<int64_t, int> cache;
  float t = (float)((1.0 + std::sqrt(5.0)) / 2.0);
  const float init[12][3] = {{-1, t, 0}, {1, t, 0}, {-1, -t, 0}, {1, -t, 0}, {0, -1, t}, {0, 1, t},
                             {0, -1, -t}, {0, 1, -t}, {t, 0, -1}, {t, 0, 1}, {-t, 0, -1}, {-t, 0, 1}};
  for (auto &p : init) verts.push_back(normalized(V3{p[0], p[1], p[2]}));
  const int f0[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4},
                         {11, 10, 2}, {10, 7, 6}, {7, 1, 8}, {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8},
                         {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
  for (auto &f : f0) faces.push_back({f[0], f[1], f[2]});
  auto middle = [&](int i, int j) {
    int64_t sm = std::min(i, j), gr = std::max(i, j), key = (sm << 32) + gr;
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    V3 a = verts[i], b = verts[j];
    verts.push_back(normalized(V3{(a.x + b.x) / 2.0f, (a.y + b.y) / 2.0f, (a.z + b.z) / 2.0f}));
    cache[key] = (int)verts.size() - 1;
    return (int)verts.size() - 1;
  };
  while (verts.size() < min_views) {
    std::vector<std::array<int, 3>> nf;
    for (auto &f : faces) {
      int a = f[0], b = f[1], c = f[2];
      int ab = middle(a, b), bc = middle(b, c), ca = middle(c, a);
      nf.push_back({a, ab, ca}); nf.push_back({b, bc, ab}); nf.push_back({c, ca, bc}); nf.push_back({ab, bc, ca});
    }
    faces.swap(nf);
  }
  return verts;
}
}  // namespace

std::vector<float> make_rotation_grid(int min_views, int inplane_steps) {
  std::vector<float> out;
  auto verts = icosphere((unsigned)min_views);
  const double step = 360.0 / inplane_steps;
  for (auto &pos : verts) {
    V3 z = normalized(V3{-pos.x, -pos.y, -pos.z});
    V3 up{0, 0, 1};
    V3 x{up.y * z.z - up.z * z.y, up.z * z.x - up.x * z.z, up.x * z.y - up.y * z.x};
    if (x.x == 0.0f && x.y == 0.0f && x.z == 0.0f) x = V3{1, 0, 0};
    x = normalized(x);
    V3 y = normalized(V3{z.y * x.z - z.z * x.y, z.z * x.x - z.x * x.z, z.x * x.y - z.y * x.x});
    // cam_in_ob rotation columns x,y,z ; translation pos
    for (int k = 0; k < inplane_steps; k++) {
      float a = (float)((k * step) * M_PI / 180.0f);
      float s = std::sin(a), c = std::cos(a);
      // R = [x y z] * Rz(a): columns
      double cx[3] = {(double)(x.x * c + y.x * s), (double)(x.y * c + y.y * s), (double)(x.z * c + y.z * s)};
      double cy[3] = {(double)(x.x * -s + y.x * c), (double)(x.y * -s + y.y * c), (double)(x.z * -s + y.z * c)};
      double cz[3] = {(double)z.x * ((1.0f - c) + c), (double)z.y * ((1.0f - c) + c), (double)z.z * ((1.0f - c) + c)};
      double tt[3] = {pos.x, pos.y, pos.z};
      // ob_in_cam = inverse: rotation R^T (rows = columns of R), translation -R^T t
      float o[16];
      const double *cols[3] = {cx, cy, cz};
      for (int r = 0; r < 3; r++)
        for (int cc = 0; cc < 3; cc++) o[cc * 4 + r] = (float)cols[r][cc];
      for (int r = 0; r < 3; r++) o[12 + r] = (float)(-(cols[r][0] * tt[0] + cols[r][1] * tt[1] + cols[r][2] * tt[2]));
      o[3] = o[7] = o[11] = 0; o[15] = 1;
      out.insert(out.end(), o, o + 16);
    }
  }
  return out;
}

// GuessTranslation (D6F/src/foundationpose_sampling.cpp:250-298) on host buffers
template <typename T>
static int dev_alloc(T **p, size_t count) {
  FP_HIP_OK(hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)));
  return 0;
}
template <typename T>
static void dev_free(T *&p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

}  // namespace fp

using namespace fp;

struct Target {
  std::string name;
  DeviceMesh mesh;
};

// a calibration frame kept on the host between fp_calibrate_add_frame and fp_calibrate_finish
struct CalibFrame {
  std::vector<uint8_t> rgb, mask;
  std::vector<float> depth;
  int H = 0, W = 0;
  std::string target;
};

struct fp_model {
  int device = 0;            // the HIP device the model lives on (fp_create_on); every entry point makes it current for its duration
  hipStream_t stream = nullptr;
  Profiler prof;
  float K[9];
  int max_h = 1080, max_w = 1920;
  int inplane_steps = 6;
  std::vector<Target> targets;
  std::vector<float> grid_host;  // [n_hyp*16]

  // frame
  int H = 0, W = 0;
  uint8_t *rgb_own = nullptr;
  float *depth_own = nullptr;
  FrameRef *frame_dev = nullptr;   // what kernels inside graphs read the frame through
  // [r6] a ring of 8 pinned, device-MAPPED record slots of 64 bytes each: a changed record reaches frame_dev through a kernel
  // (window_fetch_kernel), not a copy command
  uint8_t *frame_pinned = nullptr, *frame_pinned_dev = nullptr;
  // [r6] Register from HOST frames: rgb | depth | mask are packed into this pinned, device-mapped block (grown on demand) and fetched by
  // staged_upload_kernel -- no copy command and no runtime-internal staging of the caller's pageable pages on a serving path
  uint8_t *host_stage = nullptr, *host_stage_dev = nullptr;
  size_t host_stage_cap = 0;
  hipEvent_t host_stage_done = nullptr;   // recorded behind the last fetch: the block is rewritten only after it
  bool host_stage_pending = false;
  bool host_stage_fresh = false;   // this call's frame upload has just claimed the block (the mask goes behind it without waiting again)
  // [r4] frame_dev heads a device block [FrameRef, padded to 64 bytes | packed window]: Track's crop window of a host frame is packed
  // (record, rgb rows, depth rows) into the pinned twin win_stage and fetched from there by window_fetch_kernel, record included
  uint8_t *win_stage = nullptr, *win_stage_dev = nullptr;   // (host address / the device's mapping of it)
  size_t win_cap = 0;             // bytes of either block (0: no window path)
  FrameRef frame_pub = {nullptr, nullptr};                 // last published value
  unsigned frame_pub_count = 0;
  float *multi_io = nullptr, *multi_io_dev = nullptr;  // host-pinned [K poses in | K poses out] of fp_track_multi
  int multi_io_cap = 0;
  std::vector<Target *> mg_sig;                        // the object sequence the multi-object graph was captured for
  bool track_pending = false;  // fp_track_submit without its fp_track_wait
  bool frame_partial = false;  // the model's copy of a host frame holds only the rows Track needed (stage operators refuse it)
  const uint8_t *rgb = nullptr;   // device
  const float *depth = nullptr;   // device
  float *erode = nullptr, *bilat = nullptr, *xyz = nullptr;
  // device-side sampler (GuessTranslation): rotation grid, scan state (int[8]), compacted depth list, mask copy
  float *grid_dev = nullptr;
  int *samp_state = nullptr;
  float *samp_vals = nullptr;
  uint8_t *mask_dev = nullptr;
  size_t samp_px_cap = 0;
  size_t frame_cap = 0;

  // per-hypothesis scratch
  int cap = 0;
  size_t vert_cap = 0;
  PoseRec *recs = nullptr;
  float *poses_dev = nullptr;
  float4 *clip = nullptr, *attr = nullptr;
  unsigned *tri_rows = nullptr;   // [cap, max_faces] row range of every triangle of every hypothesis (launch_tri_rows)
  size_t tri_cap = 0, max_faces = 0;
  __half *nn_in = nullptr;                  // [2*cap,84,84,32] (s2d, zero border 2)
  float *blob_a = nullptr, *blob_b = nullptr;  // [cap,160,160,6] f32 (blob-mode entry points only)
  float *trans_dev = nullptr, *rot_dev = nullptr, *scores_dev = nullptr, *feat_dev = nullptr;
  int *argmax_dev = nullptr;
  // one read-back per Register: device [pose16] + pinned host mirror {idx, sampler status, pose16}
  float *best_pose_dev = nullptr;
  int *result_pinned = nullptr, *result_pinned_dev = nullptr;  // 18 words, device-mapped [r6]: publish_result_kernel writes them (no D2H copy command)
  bool defer_begin_sync = false;  // fp_register_ex: shard_begin leaves its synchronisation to shard_finish
  bool shard_sampler_pending = false;  // packed shard protocol: begin ran the sampler, finish reports its verdict
  float *scores_all = nullptr;  // scores of the gathered hypotheses of every rank (sharded Register)
  int scores_all_cap = 0;
  float *gath_feat = nullptr, *gath_poses = nullptr;  // [n_total,512] / [n_total,16] unpacked from the all-gathered rows
  int gath_cap = 0;
  float *shard_send = nullptr, *shard_recv = nullptr;  // fp_register_sharded: persistent exchange buffers [per,528] / [world*per,528]
  size_t shard_send_cap = 0, shard_recv_cap = 0;
  int32_t *dbg_tri = nullptr;
  float *dbg_rast = nullptr;

  // networks per precision (loaded on demand from the weight files), activation arenas per precision (the border
  // positions of a tensor depend on its element size, so an arena serves one precision)
  std::string refiner_path, scorer_path;
  Net *refiner_p[N_PREC] = {nullptr}, *scorer_p[N_PREC] = {nullptr};
  NNScratch *ws_p[N_PREC] = {nullptr};
  int prec = PREC_F16;
  // float model of the rendering stage: true = multiply-adds contracted like the reference's nvcc -fmad=true build
  // (fp_geometry.hip "float model"), false = every operation separately rounded
  bool fmad = true;
  Net *refiner = nullptr, *scorer = nullptr;  // = refiner_p[prec], scorer_p[prec]
  NNScratch *ws = nullptr;                    // = ws_p[prec]
  bool calibrating = false;
  // calibration of the 8-bit precisions (fp_calibrate): [refiner, scorer] per-channel |max| / mean of the 15 trunk activations of
  // the f16 networks on the calibration frame ([15][512] each), and per 8-bit precision the solved corrections (bias [13][512], token
  // [512]); they are applied to a precision's networks when those are loaded / re-calibrated
  int calib_session_prec = -1;                 // fp_calibrate_begin .. fp_calibrate_finish: the precision being calibrated (-1: no session)
  std::vector<CalibFrame> calib_frames; // host copies of the session's frames
  // the |max| record each 8-bit precision was quantised with (empty = not calibrated).  Per precision: two precisions may have been
  // calibrated on different frames, and a blob must carry the statistics ITS corrections were solved against
  std::vector<float> calib_amax_q[N_PREC][2];
  bool calibrated(int prec) const { return prec >= 0 && prec < N_PREC && !calib_amax_q[prec][0].empty(); }
  std::vector<float> calib_bias_fix[N_PREC][2], calib_tok_fix[N_PREC][2];
  std::vector<float> calib_out_fix[N_PREC][2];   // output-layer correction: refiner [8] (trans 3 | rot 3 | 0 0), scorer [512]
  // [r5] per-frame channel means of the f16 trunk activations ([slots][15][512], slots <= FP_CALIB_SLOTS: frame f goes to slot f % slots):
  // the INT8 weights are rounded with error feedback against them (fp_nn.hip quantise_q8), so a record must carry them
  std::vector<float> calib_fmeans[N_PREC][2];
  int calib_slots[N_PREC] = {0};
  // pinned staging for hypothesis poses: Register returns from its asynchronous section while the H2D copy may still be
  // queued, so the source must outlive the call (a local std::vector did not: found by the two-model serving test)
  float *track_io = nullptr, *track_io_dev = nullptr;  // host-pinned [hypothesis 16 | refined pose 16 | done flag] of Track and its device address
  bool track_flag_armed = false;   // the submitted Track ends in the kernel that raises the done flag (fp_track_wait polls it)
  float *poses_pinned = nullptr;
  int poses_pinned_cap = 0;
  unsigned long long *digests = nullptr;  // [16] device, debug checkpoints (null = off)

  // Track is launch-bound (~60 short kernels): after one eager call (allocations settle) the launch chain is captured
  // into a hipGraph and replayed.  The graph bakes buffer addresses, so it is keyed by g_alloc_epoch.
  // (Register's ~110 launches are replayed the same way: inter-kernel gaps are ~4 % of a 12 ms Register.)
  struct GraphSlot {
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph = nullptr;
    Target *target = nullptr;
    int H = 0, W = 0, itr = 0, n = 0, prec = 0;
    unsigned long epoch = 0;
    int eager_calls = 0;
  } tg, rg, mg;  // Track body / Register body / multi-object Track body
  bool use_graphs = true;

  Target *find(const char *name) {
    for (auto &t : targets)
      if (t.name == name) return &t;
    return nullptr;
  }
  int n_hyp() const { return 42 * inplane_steps; }
};


// ---- debug checkpoints (tools/dbg_concurrent.py): order-independent 64-bit digest of a device buffer per pipeline stage
__global__ void fp_digest_kernel(const uint32_t *p, size_t n, unsigned long long *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long acc = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += (unsigned long long)p[i] * (2654435761ull + 2 * (i % 1000003ull));
  if (acc) atomicAdd(out, acc);
}
static void checkpoint(fp_model *m, int slot, const void *buf, size_t bytes);

static int ensure_capacity(fp_model *m, int N, size_t V) {
  if (N > m->cap) {
    g_alloc_epoch++;
    dev_free(m->recs); dev_free(m->poses_dev); dev_free(m->nn_in); dev_free(m->blob_a); dev_free(m->blob_b);
    dev_free(m->trans_dev); dev_free(m->rot_dev); dev_free(m->scores_dev); dev_free(m->feat_dev);
    dev_free(m->clip); dev_free(m->attr); dev_free(m->dbg_tri); dev_free(m->dbg_rast);
    m->vert_cap = 0;
    int cap = std::max(N, 8);
    if (dev_alloc(&m->recs, cap)) return 1;
    if (dev_alloc(&m->poses_dev, (size_t)cap * 16)) return 1;
    if (dev_alloc(&m->nn_in, (size_t)2 * cap * FP_NN_IN_IMG_HALFS)) return 1;
    // zero borders are written once here; the raster / crop kernels only ever store image interiors
    FP_HIP_OK(hipMemsetAsync(m->nn_in, 0, (size_t)2 * cap * FP_NN_IN_IMG_HALFS * sizeof(__half), m->stream));
    if (dev_alloc(&m->trans_dev, (size_t)cap * 3)) return 1;
    if (dev_alloc(&m->rot_dev, (size_t)cap * 3)) return 1;
    if (dev_alloc(&m->scores_dev, (size_t)cap)) return 1;
    if (dev_alloc(&m->feat_dev, (size_t)cap * 512)) return 1;
    m->cap = cap;
  }
  if ((size_t)m->cap * V > m->vert_cap) {
    g_alloc_epoch++;
    dev_free(m->clip); dev_free(m->attr);
    if (dev_alloc(&m->clip, (size_t)m->cap * V)) return 1;
    if (dev_alloc(&m->attr, (size_t)m->cap * V)) return 1;
    m->vert_cap = (size_t)m->cap * V;
  }
  // row ranges: for the largest mesh of the model (any target may be rendered) and for the batch sizes that are rendered in short
  // strips (raster_wants_tri_rows: below 100 hypotheses; a larger batch falls back to the full walk when the buffer is too small)
  const size_t tri_need = (size_t)(g_tri_rows_all_batches ? m->cap : std::min(m->cap, 99)) * m->max_faces;
  if (V > 0 && tri_need > m->tri_cap) {
    g_alloc_epoch++;
    dev_free(m->tri_rows);
    m->tri_cap = 0;
    if (dev_alloc(&m->tri_rows, tri_need)) return 1;
    m->tri_cap = tri_need;
  }
  return 0;
}

static int ensure_blobs(fp_model *m) {
  size_t n = (size_t)m->cap * FP_CROP_HW * FP_CROP_HW * 6;
  if (!m->blob_a && dev_alloc(&m->blob_a, n)) return 1;
  if (!m->blob_b && dev_alloc(&m->blob_b, n)) return 1;
  return 0;
}

static int check_frame_args(fp_model *m, int H, int W, const char *target_name, Target **t) {
  // CheckInputArguments (D6F/src/foundationpose.cpp:155-179)
  FP_CHECK(m != nullptr, "[FoundationPose] null model");
  FP_CHECK(H > 0 && W > 0 && H <= m->max_h && W <= m->max_w, "[FoundationPose] Got rgb/depth/mask with unexpected size !");
  if (target_name) {
    *t = m->find(target_name);
    FP_CHECK(*t != nullptr,
             "[FoundationPose] Register Got Invalid `target_name` which was not provided to FoundationPose instance!!!");
  }
  return 0;
}

// render + crop for N poses already in m->poses_dev; writes the fp16 network input (both halves) or fp32 blobs
// n_crop: number of observed crops to produce (N, or 1 when every hypothesis shares the same translation)
static int render_and_crop(fp_model *m, Target *t, int N, float crop_ratio, OutMode mode, void *out_a, void *out_b,
                           int32_t *dbg_tri, float *dbg_rast, int n_crop = -1, const float *poses_src = nullptr, PoseRec *recs = nullptr) {
  if (n_crop < 0) n_crop = N;
  if (!recs) recs = m->recs;   // (fp_track_multi: the records of one object group inside the batch)
  hipStream_t s = m->stream;
  const size_t out_bytes = (mode == OUT_F32X6 ? 24.0 : 16.0) * FP_CROP_HW * FP_CROP_HW;  // both 2-byte modes: 16 B per pixel
  if (!out_a) {
    ProfScope ps(&m->prof, s, "pose_setup");
    launch_pose_setup(s, poses_src ? poses_src : m->poses_dev, N, m->K, m->H, m->W, crop_ratio, t->mesh.diameter, recs);
  }
  unsigned *const rows_small = g_tri_rows && m->tri_rows && raster_wants_tri_rows(N) && (size_t)N * t->mesh.F <= m->tri_cap ? m->tri_rows : nullptr;
  if (out_a && out_b && N <= 4 && !m->prof.on && !dbg_tri && !dbg_rast && g_vertex_crop &&
      launch_setup_vertex_crop(s, t->mesh, poses_src ? poses_src : m->poses_dev, N, m->K, m->H, m->W, crop_ratio, t->mesh.diameter, recs, m->clip,
                               m->attr, m->fmad, m->frame_dev, n_crop, mode, out_b, g_vertex_crop == 2 ? nullptr : rows_small)) {
    // tiny batches (Track): set-up + vertex stage + crop warp + the triangles' row ranges were ONE launch; the rasteriser follows
    unsigned *rows = rows_small;
    if (rows && g_vertex_crop == 2) launch_tri_rows(s, t->mesh, N, m->clip, rows);   // A/B: the row ranges as their own launch
    launch_raster_shade(s, t->mesh, recs, N, m->clip, m->attr, mode, out_a, nullptr, nullptr, m->fmad, rows);
    FP_HIP_OK(hipGetLastError());
    return 0;
  }
  if (out_a) {
    {   // pose set-up (crop window, bounding box, projection) is computed inside the vertex kernel: one launch less per render
      ProfScope ps(&m->prof, s, "vertex", 0, (double)N * t->mesh.V * 32.0 + t->mesh.V * 24.0);
      launch_setup_vertex(s, t->mesh, poses_src ? poses_src : m->poses_dev, N, m->K, m->H, m->W, crop_ratio, t->mesh.diameter, recs,
                          m->clip, m->attr, m->fmad);
    }
    unsigned *rows = g_tri_rows && m->tri_rows && raster_wants_tri_rows(N) && (size_t)N * t->mesh.F <= m->tri_cap ? m->tri_rows : nullptr;
    if (rows) {
      ProfScope ps(&m->prof, s, "tri_rows", 0, (double)N * t->mesh.F * (12.0 + 48.0 + 4.0));
      launch_tri_rows(s, t->mesh, N, m->clip, rows);
    }
    ProfScope ps(&m->prof, s, "raster_shade", 0, (double)N * (out_bytes + t->mesh.V * 32.0 + t->mesh.F * 12.0));
    launch_raster_shade(s, t->mesh, recs, N, m->clip, m->attr, mode, out_a, dbg_tri, dbg_rast, m->fmad, rows);
  }
  if (out_b) {
    ProfScope ps(&m->prof, s, "crop_warp", 0, (double)n_crop * out_bytes);
    launch_crop(s, m->frame_dev, m->H, m->W, m->K, recs, n_crop, t->mesh.diameter, mode, out_b);
  }
  FP_HIP_OK(hipGetLastError());
  return 0;
}

static void checkpoint(fp_model *m, int slot, const void *buf, size_t bytes) {
#ifndef FP_TEST_HOOKS
  (void)m; (void)slot; (void)buf; (void)bytes;
  return;
#endif
  if (!m->digests || !buf) return;
  hipLaunchKernelGGL(fp_digest_kernel, dim3(512), dim3(256), 0, m->stream, (const uint32_t *)buf, bytes / 4, m->digests + slot);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) std::fprintf(stderr, "checkpoint %d (%p, %zu B): %s\n", slot, buf, bytes, hipGetErrorString(e));
}

static int upload_frame_async(fp_model *m, const void *rgb, const void *depth, int memspace, int H, int W, int row0 = 0, int row1 = -1, int col0 = 0, int col1 = -1);
static int set_rotation_grid(fp_model *m, int steps);

static void drop_graph(fp_model::GraphSlot &g) {
  if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (g.graph) (void)hipGraphDestroy(g.graph);
  g.exec = nullptr; g.graph = nullptr; g.eager_calls = 0;
}

// every captured body bakes the float model, the precision and (FP8) the per-tensor activation scales in as kernel arguments:
// anything that changes one of them drops ALL three graphs
static void invalidate_graphs(fp_model *m) {
  drop_graph(m->tg);
  drop_graph(m->rg);
  drop_graph(m->mg);
}

// Runs `body` (a chain of launches on m->stream reading / writing only model-owned buffers): eagerly the first time a
// (target, H, W, itr, n) configuration is seen, captured into a hipGraph on the second call once allocations have
// settled (the graph bakes buffer addresses, so it is keyed by g_alloc_epoch), replayed from then on.
template <class Body>
static int run_graphed(fp_model *m, fp_model::GraphSlot &g, Target *t, int H, int W, int itr, int n, bool graphable, Body body) {
  const bool same = g.target == t && g.H == H && g.W == W && g.itr == itr && g.n == n && g.prec == m->prec && g.epoch == g_alloc_epoch;
  if (graphable && same && g.exec) {
    FP_HIP_OK(hipGraphLaunch(g.exec, m->stream));
  } else if (graphable && same && g.eager_calls >= 1) {
    drop_graph(g);
    FP_HIP_OK(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
    int rc = body();
    hipError_t e = hipStreamEndCapture(m->stream, &g.graph);
    if (rc || e != hipSuccess || hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0) != hipSuccess) {
      drop_graph(g);
      m->use_graphs = false;  // fall back to eager launches for good
      if (body()) return 1;
    } else {
      g.eager_calls = 1;
      FP_HIP_OK(hipGraphLaunch(g.exec, m->stream));
    }
  } else {
    if (!same) { drop_graph(g); g.target = t; g.H = H; g.W = W; g.itr = itr; g.n = n; g.prec = m->prec; }
    if (body()) return 1;
    g.epoch = g_alloc_epoch;  // allocations made by this eager call are now settled
    g.eager_calls = graphable ? g.eager_calls + 1 : 0;
  }
  return 0;
}

// ---- packed exchange buffers of a sharded Register: row = [pooled score feature 512 | refined pose 16]
// sampler_status: the device-side verdict word of THIS rank's sampler run (0 = poses valid): anything else poisons the rows with
// NaNs, so that every rank's finish reports the failure -- a rank whose sampler failed must not feed garbage rows to ranks that
// would otherwise succeed (the verdict itself reaches the host only with the finish half's single synchronisation)
__global__ void pack_shard_kernel(const float *__restrict__ feat, const float *__restrict__ poses, int count, int per,
                                  float *__restrict__ packed, const int *__restrict__ sampler_status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per * 528) return;
  const int r = i / 528, c = i - r * 528;
  if (sampler_status && *sampler_status != 0) { packed[i] = __int_as_float(0x7fc00000); return; }
  packed[i] = r >= count ? 0.f : (c < 512 ? feat[(size_t)r * 512 + c] : poses[(size_t)r * 16 + (c - 512)]);
}
// gathered rows are already in global hypothesis order (contiguous shards of `per` rows, padding only behind row n_total)
__global__ void unpack_shards_kernel(const float *__restrict__ gathered, int n_total, float *__restrict__ feat,
                                     float *__restrict__ poses) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total * 528) return;
  const int r = i / 528, c = i - r * 528;
  if (c < 512) feat[(size_t)r * 512 + c] = gathered[i];
  else poses[(size_t)r * 16 + (c - 512)] = gathered[i];
}

// Escape hatch (off by default): FP_SERIALIZE_MODELS=1 in the environment makes the synchronous entry points of ALL models in the
// process (Register, Track, fp_track_multi, FP8 calibration) take one process-wide lock for their whole duration, so that kernels
// of two models are never co-resident.  This is the round-1 behaviour; it exists because the packed-f32 erratum behind the
// removal of that lock was mitigated, not reproduced in isolation.  The pipelined fp_track_submit /
// fp_track_wait pair is not covered (it exists to overlap models).
// Lifetime lock [r3]: every compute entry point holds it SHARED for the duration of the call, fp_create / fp_destroy / fp_net_create /
// fp_net_destroy hold it EXCLUSIVE -- a model is never built or torn down (hundreds of allocations, uploads, a stream) while another
// thread of the process is inside a Register / Track call capturing or replaying its graphs.  (Two of ~20 runs of the widened
// concurrency test died with SIGSEGV inside fp_register_shard_begin on two threads while the main thread was inside fp_create;
// the calls themselves share nothing but the HIP runtime.)  Calls still run concurrently with each other; GPU work already enqueued
// keeps running during a creation.  Nested entry points (fp_register_ex -> shard_begin) take it once (thread-local depth).
static std::shared_mutex g_life_rw;
static thread_local int g_life_depth = 0;
struct LifeExclusive {   // (entry points nested inside -- fp_calibrate_fp8 runs a Register -- see depth > 0 and take nothing)
  LifeExclusive() { g_life_rw.lock(); ++g_life_depth; }
  ~LifeExclusive() { --g_life_depth; g_life_rw.unlock(); }
  LifeExclusive(const LifeExclusive &) = delete;
  LifeExclusive &operator=(const LifeExclusive &) = delete;
};
// Device affinity [r4]: a model remembers the device it was created on and every entry point makes that device current for the
// duration of the call (and puts the caller's device back), so a single-process multi-GPU host -- one thread per GPU, or one thread
// walking over the models -- cannot launch a model's work on the wrong device.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  explicit DeviceScope(int dev) {
    if (dev < 0) return;
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess) { prev = cur; switched = true; }
  }
  ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
  DeviceScope(const DeviceScope &) = delete;
  DeviceScope &operator=(const DeviceScope &) = delete;
};
// fp_register_sharded with more than one rank runs WITHOUT the FP_SERIALIZE_MODELS lock: a host with one thread per rank would
// deadlock on it (the thread inside the finish half's stream synchronisation holds the lock while its all-gather waits for ranks
// whose threads cannot enqueue theirs), and the ranks of a collective live on different devices, where the co-residency the lock
// exists to prevent cannot happen.
static thread_local bool g_no_serial = false;
struct SerialGuard {
  std::unique_lock<std::recursive_mutex> lk;
  bool shared_held = false;
  DeviceScope dev;
  explicit SerialGuard(int device) : dev(device) {
    if (g_life_depth++ == 0) { g_life_rw.lock_shared(); shared_held = true; }
    static const bool on = [] { const char *e = std::getenv("FP_SERIALIZE_MODELS"); return e && *e && *e != '0'; }();
    static std::recursive_mutex mu;
    if (on && !g_no_serial) lk = std::unique_lock<std::recursive_mutex>(mu);
  }
  ~SerialGuard() {
    if (lk.owns_lock()) lk.unlock();
    --g_life_depth;
    if (shared_held) g_life_rw.unlock_shared();
  }
  SerialGuard(const SerialGuard &) = delete;
  SerialGuard &operator=(const SerialGuard &) = delete;
};

// Exception barrier of the C ABI: every entry point below that can allocate on the host (std::vector / std::string / new) is a
// function-try-block ending in one of these; nothing is thrown across the extern "C" boundary.
static int caught_exception() noexcept {
  try {
    try { throw; }
    catch (const std::bad_alloc &) { set_error("[FoundationPose] out of host memory"); }
    catch (const std::exception &e) { set_error(std::string("[FoundationPose] internal error: ") + e.what()); }
    catch (...) { set_error("[FoundationPose] internal error (unknown exception)"); }
  } catch (...) {
  }
  return 1;
}
#define FP_CATCH_INT catch (...) { return caught_exception(); }
#define FP_CATCH_PTR catch (...) { caught_exception(); return nullptr; }

extern "C" {

#ifdef FP_TEST_HOOKS
void fpt_set_calib_opts(int sweeps, int tok, int out) { g_calib_sweeps = sweeps; g_calib_tok = tok; g_calib_out = out; }
void fpt_set_vertex_crop(int v) { g_vertex_crop = v; }
void fpt_set_tri_rows(int v) { g_tri_rows = v != 0; g_tri_rows_all_batches = v == 2; }
void fpt_set_upload_cols(int v) { g_upload_cols = v; }
// A/B hook: hipGraph replay of the Track / Register bodies on or off for one model
int fpt_model_use_graphs(fp_model *m, int on) {
  m->use_graphs = on != 0;
  invalidate_graphs(m);
  return 0;
}

// bit 0: graphs still enabled (a failed capture disables them), bit 1: Track graph instantiated, bit 2: Register graph
int fpt_model_graph_state(fp_model *m) { return (m->use_graphs ? 1 : 0) | (m->tg.exec ? 2 : 0) | (m->rg.exec ? 4 : 0); }

// debug: enable the stage digests and read them back (16 slots; zeroed by every read)
int fpt_digests(fp_model *m, unsigned long long out[16]) {
  if (!m->digests) {
    FP_HIP_OK(hipMalloc((void **)&m->digests, 16 * 8));
    FP_HIP_OK(fp::memset_sync(m->digests, 0, 16 * 8));
    return 0;
  }
  FP_HIP_OK(hipMemcpyAsync(out, m->digests, 16 * 8, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipMemsetAsync(m->digests, 0, 16 * 8, m->stream));
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  return 0;
}

// debug: digests of every long-lived device buffer of an (idle) model: recs, poses, clip, attr, nn_in, trans, rot,
// scores, feat, arena, arena f32 (11 values) -- to detect writes coming from OTHER models' kernels
int fpt_digest_buffers(fp_model *m, unsigned long long out[16]) {
  unsigned long long *d = nullptr;
  FP_HIP_OK(hipMalloc((void **)&d, 16 * 8));
  FP_HIP_OK(hipMemsetAsync(d, 0, 16 * 8, m->stream));
  const void *ab = nullptr, *af = nullptr;
  size_t abytes = 0, afbytes = 0;
  if (m->ws) nn_scratch_debug_info(m->ws, &ab, &abytes, &af, &afbytes);
  size_t V = m->targets.empty() ? 0 : (size_t)m->targets[0].mesh.V;
  const DeviceMesh *dm = m->targets.empty() ? nullptr : &m->targets[0].mesh;
  struct { const void *p; size_t n; } bufs[16] = {
      {m->recs, (size_t)m->cap * sizeof(PoseRec)}, {m->poses_dev, (size_t)m->cap * 64}, {m->clip, m->vert_cap * 16},
      {m->attr, m->vert_cap * 16}, {m->nn_in, (size_t)2 * m->cap * FP_NN_IN_IMG_HALFS * 2}, {m->trans_dev, (size_t)m->cap * 12},
      {m->rot_dev, (size_t)m->cap * 12}, {m->scores_dev, (size_t)m->cap * 4}, {m->feat_dev, (size_t)m->cap * 2048},
      {ab, abytes}, {af, afbytes},
      {dm ? dm->verts : nullptr, V * 12}, {dm ? dm->normals : nullptr, V * 12}, {dm ? dm->uvs : nullptr, V * 8},
      {dm ? (const void *)dm->faces : nullptr, dm ? (size_t)dm->F * 12 : 0}, {dm ? (const void *)dm->tex : nullptr, dm ? (size_t)dm->TH * dm->TW * 3 : 0}};
  for (int i = 0; i < 16; i++)
    if (bufs[i].p && bufs[i].n >= 4)
      hipLaunchKernelGGL(fp_digest_kernel, dim3(1024), dim3(256), 0, m->stream, (const uint32_t *)bufs[i].p, bufs[i].n / 4, d + i);
  FP_HIP_OK(hipMemcpyAsync(out, d, 16 * 8, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  (void)hipFree(d);
  return 0;
}

// debug: allocate the vertex-stage intermediates buffer (launches with N == 64 fill it) / read it back
long long fpt_vertex_dbg(fp_model *m, void *dst, long long bytes) {
  if (!fp::g_vertex_dbg) {
    float4 *p = nullptr;
    if (hipMalloc((void **)&p, (size_t)bytes) != hipSuccess) return -1;
    fp::g_vertex_dbg = p;
    return 0;
  }
  if (hipMemcpyAsync(dst, fp::g_vertex_dbg, (size_t)bytes, hipMemcpyDeviceToHost, m->stream) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess) return -1;
  return bytes;
}
// debug: copy one of the model's device buffers to the host (0 recs, 1 clip, 2 attr, 3 nn_in, 4 poses); returns bytes copied
long long fpt_read_buffer(fp_model *m, int which, void *dst, long long max_bytes) {
  size_t V = m->targets.empty() ? 0 : (size_t)m->targets[0].mesh.V;
  const void *src[5] = {m->recs, m->clip, m->attr, m->nn_in, m->poses_dev};
  size_t n[5] = {(size_t)m->cap * sizeof(PoseRec), (size_t)m->cap * V * 16, (size_t)m->cap * V * 16,
                 (size_t)2 * m->cap * FP_NN_IN_IMG_HALFS * 2, (size_t)m->cap * 64};
  if (which < 0 || which > 4 || !src[which]) return -1;
  size_t b = std::min<size_t>(n[which], (size_t)max_bytes);
  if (hipMemcpyAsync(dst, src[which], b, hipMemcpyDeviceToHost, m->stream) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess) return -1;
  return (long long)b;
}

#endif  // FP_TEST_HOOKS

const char *fp_last_error(void) { return g_last_error.c_str(); }

// Host phase of loading the networks of a precision that are not in memory yet (file I/O + five weight re-layouts per layer: 0.3-1 s)
// -- runs BEFORE the caller takes the exclusive lifetime lock, so other models keep serving meanwhile; adopt() (under the lock) does
// the device phase (one allocation + upload per network) and hands them to the model.
struct PreparedNets {
  Net *r = nullptr, *s = nullptr;
  int prec = -1;
  PreparedNets() = default;
  PreparedNets(const PreparedNets &) = delete;
  PreparedNets &operator=(const PreparedNets &) = delete;
  ~PreparedNets() { if (r) net_free(r); if (s) net_free(s); }
  int prepare(const std::string &refiner_path, bool have_r, const std::string &scorer_path, bool have_s, int precision) {
    prec = precision;
    std::string err;
    if (!refiner_path.empty() && !have_r) {
      r = net_prepare(refiner_path.c_str(), false, precision, &err);
      FP_CHECK(r != nullptr, "[FoundationPose] Failed to load refiner weights: " + err);
    }
    if (!scorer_path.empty() && !have_s) {
      s = net_prepare(scorer_path.c_str(), true, precision, &err);
      FP_CHECK(s != nullptr, "[FoundationPose] Failed to load scorer weights: " + err);
    }
    return 0;
  }
  int adopt(fp_model *m) {
    std::string err;
    if (r && !m->refiner_p[prec]) {
      FP_CHECK(net_commit(r, &err) == 0, "[FoundationPose] Failed to load refiner weights: " + err);
      m->refiner_p[prec] = r; r = nullptr;
    }
    if (s && !m->scorer_p[prec]) {
      FP_CHECK(net_commit(s, &err) == 0, "[FoundationPose] Failed to load scorer weights: " + err);
      m->scorer_p[prec] = s; s = nullptr;
    }
    return 0;
  }
};

// networks of precision `prec` (loaded on first use) become the model's current ones
static int select_precision(fp_model *m, int prec) {
  FP_CHECK(prec == PREC_F16 || prec == PREC_BF16 || prec == PREC_FP8 || prec == PREC_INT8, "[FoundationPose] unknown precision");
  std::string err;
  if (!m->refiner_path.empty() && !m->refiner_p[prec]) {
    m->refiner_p[prec] = net_load(m->refiner_path.c_str(), false, prec, &err);
    FP_CHECK(m->refiner_p[prec] != nullptr, "[FoundationPose] Failed to load refiner weights: " + err);
  }
  if (!m->scorer_path.empty() && !m->scorer_p[prec]) {
    m->scorer_p[prec] = net_load(m->scorer_path.c_str(), true, prec, &err);
    FP_CHECK(m->scorer_p[prec] != nullptr, "[FoundationPose] Failed to load scorer weights: " + err);
  }
  if ((prec == PREC_FP8 || prec == PREC_INT8) && m->calibrated(prec)) {
    Net *nets[2] = {m->refiner_p[prec], m->scorer_p[prec]};
    for (int k = 0; k < 2; k++)
      if (nets[k] && !net_q8_ready(nets[k]) &&
          net_apply_q8(nets[k], m->calib_amax_q[prec][k].data(), m->calib_bias_fix[prec][k].empty() ? nullptr : m->calib_bias_fix[prec][k].data(),
                       m->calib_tok_fix[prec][k].empty() ? nullptr : m->calib_tok_fix[prec][k].data(), true,
                       m->calib_slots[prec] ? m->calib_fmeans[prec][k].data() : nullptr, m->calib_slots[prec]))
        return 1;
    for (int k = 0; k < 2; k++)
      if (nets[k] && !m->calib_out_fix[prec][k].empty() && net_q8_set_out_fix(nets[k], m->calib_out_fix[prec][k].data())) return 1;
  }
  if (!m->ws_p[prec]) m->ws_p[prec] = nn_scratch_create(prec);
  m->prec = prec;
  m->refiner = m->refiner_p[prec];
  m->scorer = m->scorer_p[prec];
  m->ws = m->ws_p[prec];
  return 0;
}
// element type of the networks' input tensor in the current precision
static OutMode nn_mode(const fp_model *m) {
  const Net *n = m->refiner ? m->refiner : m->scorer;
  return n && net_input_dt(n) == DT_BF16 ? OUT_BF16X8 : OUT_F16X8;
}

static void destroy_model_impl(fp_model *m);
fp_model *fp_create_on(int device, const fp_mesh *meshes, int n_meshes, const float K[9], const char *refiner_weights,
                       const char *scorer_weights, int max_h, int max_w) try {
  if (!meshes || n_meshes <= 0 || !K) { set_error("[FoundationPose] fp_create: invalid arguments"); return nullptr; }
  // host phase first, WITHOUT the lock [r5]: reading both weight files and building every kernel layout is most of a creation's time
  PreparedNets nets;
  if (nets.prepare(refiner_weights ? refiner_weights : "", false, scorer_weights ? scorer_weights : "", false, PREC_F16)) return nullptr;
  LifeExclusive life;   // device phase: not while another thread is inside a call (see SerialGuard)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    set_error("[FoundationPose] no HIP device available (this library has no CPU path)");
    return nullptr;
  }
  if (device < 0 && hipGetDevice(&device) != hipSuccess) { set_error("[FoundationPose] hipGetDevice failed"); return nullptr; }   // -1: the caller's current device
  if (device >= ndev) { set_error("[FoundationPose] fp_create_on: no such device"); return nullptr; }
  DeviceScope on_device(device);
  std::unique_ptr<fp_model, void (*)(fp_model *)> m(new fp_model(), destroy_model_impl);  // a failed construction releases what it had allocated
  m->device = device;
  std::memcpy(m->K, K, sizeof(float) * 9);
  if (max_h > 0) m->max_h = max_h;
  if (max_w > 0) m->max_w = max_w;
  // non-blocking: no implicit synchronisation with the legacy null stream (other models' threads, the caller's framework)
  if (stream_acquire(&m->stream) != hipSuccess) { set_error("[FoundationPose] Failed to create stream"); return nullptr; }
  static_assert(sizeof(FrameRef) <= 64, "the packed window starts 64 bytes into the frame block");
  // the frame record's device block [FrameRef, padded to 64 bytes | packed window]: the window part and its pinned twin are allocated
  // on the first host-frame Track that needs them and grown on demand (ensure_window) [r5] -- a model that is only ever handed device
  // frames, or never tracks, pins no host memory
  if (hipMalloc((void **)&m->frame_dev, 64) != hipSuccess ||
      hipHostMalloc((void **)&m->frame_pinned, 8 * 64, hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void **)&m->frame_pinned_dev, m->frame_pinned, 0) != hipSuccess) {
    set_error("[FoundationPose] Failed to allocate the frame record");
    return nullptr;
  }
  for (int i = 0; i < n_meshes; i++) {
    const fp_mesh &src = meshes[i];
    if (!src.vertices || !src.normals || !src.texcoords || !src.faces || !src.texture || src.num_vertices <= 0 ||
        src.num_faces <= 0 || src.tex_height <= 0 || src.tex_width <= 0) {
      set_error("[FoundationPose Renderer] Failed to load textured mesh!!!");
      destroy_model_impl(m.release());
      return nullptr;
    }
    Target t;
    t.name = src.name ? src.name : "";
    DeviceMesh &d = t.mesh;
    d.V = src.num_vertices; d.F = src.num_faces; d.TH = src.tex_height; d.TW = src.tex_width;
    d.diameter = src.diameter;
    std::memcpy(d.center, src.center, sizeof(float) * 3);
    // LoadTexturedMesh (D6F/src/foundationpose_render.cpp:381-509): centre vertices, flip v
    std::vector<float> v((size_t)d.V * 3), uv((size_t)d.V * 2);
    for (int k = 0; k < d.V; k++) {
      for (int c = 0; c < 3; c++) v[(size_t)k * 3 + c] = src.vertices[(size_t)k * 3 + c] - src.center[c];
      uv[(size_t)k * 2] = src.texcoords[(size_t)k * 2];
      uv[(size_t)k * 2 + 1] = 1 - src.texcoords[(size_t)k * 2 + 1];
    }
    std::vector<int32_t> f((size_t)d.F * 3);
    for (size_t k = 0; k < f.size(); k++) f[k] = (int32_t)src.faces[k];
    bool ok = !dev_alloc(&d.verts, v.size()) && !dev_alloc(&d.normals, v.size()) && !dev_alloc(&d.uvs, uv.size()) &&
              !dev_alloc(&d.faces, f.size()) && !dev_alloc(&d.tex, (size_t)d.TH * d.TW * 3);
    ok = ok && fp::memcpy_sync(d.verts, v.data(), v.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && fp::memcpy_sync(d.normals, src.normals, v.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && fp::memcpy_sync(d.uvs, uv.data(), uv.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && fp::memcpy_sync(d.faces, f.data(), f.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && fp::memcpy_sync(d.tex, src.texture, (size_t)d.TH * d.TW * 3, hipMemcpyHostToDevice) == hipSuccess;
    m->targets.push_back(t);
    m->max_faces = std::max(m->max_faces, (size_t)t.mesh.F);
    if (!ok) {
      set_error("[FoundationPose Renderer] Failed to prepare buffer!!!");
      destroy_model_impl(m.release());
      return nullptr;
    }
  }
  if (set_rotation_grid(m.get(), m->inplane_steps)) { destroy_model_impl(m.release()); return nullptr; }
  if (refiner_weights) m->refiner_path = refiner_weights;
  if (scorer_weights) m->scorer_path = scorer_weights;
  if (nets.adopt(m.get()) || select_precision(m.get(), PREC_F16)) { destroy_model_impl(m.release()); return nullptr; }
  if (hipMalloc((void **)&m->argmax_dev, sizeof(int)) != hipSuccess) { destroy_model_impl(m.release()); return nullptr; }
  return m.release();
} FP_CATCH_PTR

fp_model *fp_create(const fp_mesh *meshes, int n_meshes, const float K[9], const char *refiner_weights,
                    const char *scorer_weights, int max_h, int max_w) try {
  return fp_create_on(-1, meshes, n_meshes, K, refiner_weights, scorer_weights, max_h, max_w);
} FP_CATCH_PTR
int fp_device(const fp_model *m) { return m ? m->device : -1; }

static void destroy_model_impl(fp_model *m) {
  if (!m) return;
  DeviceScope on_device(m->device);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  invalidate_graphs(m);
  if (m->multi_io) (void)hipHostFree(m->multi_io);
  m->prof.reset();
  for (auto &t : m->targets) {
    dev_free(t.mesh.verts); dev_free(t.mesh.normals); dev_free(t.mesh.uvs); dev_free(t.mesh.faces); dev_free(t.mesh.tex);
  }
  dev_free(m->rgb_own); dev_free(m->depth_own); dev_free(m->erode); dev_free(m->bilat); dev_free(m->xyz);
  dev_free(m->recs); dev_free(m->poses_dev); dev_free(m->clip); dev_free(m->attr); dev_free(m->tri_rows); dev_free(m->nn_in);
  dev_free(m->blob_a); dev_free(m->blob_b); dev_free(m->trans_dev); dev_free(m->rot_dev); dev_free(m->scores_dev);
  dev_free(m->feat_dev); dev_free(m->argmax_dev); dev_free(m->scores_all); dev_free(m->best_pose_dev);
  dev_free(m->gath_feat); dev_free(m->gath_poses); dev_free(m->shard_send); dev_free(m->shard_recv);
  if (m->result_pinned) (void)hipHostFree(m->result_pinned); dev_free(m->dbg_tri); dev_free(m->dbg_rast);
  if (m->poses_pinned) (void)hipHostFree(m->poses_pinned);
  if (m->track_io) (void)hipHostFree(m->track_io);
  if (m->frame_pinned) (void)hipHostFree(m->frame_pinned);
  if (m->win_stage) (void)hipHostFree(m->win_stage);
  if (m->host_stage) (void)hipHostFree(m->host_stage);
  if (m->host_stage_done) (void)hipEventDestroy(m->host_stage_done);
  if (m->frame_dev) (void)hipFree(m->frame_dev);
  dev_free(m->grid_dev); dev_free(m->samp_state); dev_free(m->samp_vals); dev_free(m->mask_dev);
  if (m->digests) (void)hipFree(m->digests);
  for (int i = 0; i < N_PREC; i++) {
    if (m->refiner_p[i]) net_free(m->refiner_p[i]);
    if (m->scorer_p[i]) net_free(m->scorer_p[i]);
    if (m->ws_p[i]) nn_scratch_free(m->ws_p[i]);
  }
  if (m->stream) { (void)hipStreamSynchronize(m->stream); stream_release(m->stream); }
  delete m;
}
void fp_destroy(fp_model *m) {
  LifeExclusive life;   // not while another thread is inside a call (see SerialGuard)
  destroy_model_impl(m);
}

int fp_set_inplane_steps(fp_model *m, int steps) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && steps >= 1 && steps <= 360, "[FoundationPose] fp_set_inplane_steps: invalid arguments");
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  return set_rotation_grid(m, steps);
} FP_CATCH_INT
int fp_num_hypotheses(const fp_model *m) { return m ? m->n_hyp() : 0; }
void *fp_stream(fp_model *m) { return m ? (void *)m->stream : nullptr; }
int fp_synchronize(fp_model *m) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m, "null model");
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  return 0;
} FP_CATCH_INT

// asynchronous: the caller of this helper synchronises m->stream before the host frame can go away
// row0 / row1 (host frames): only rows [row0, row1) are needed by the caller (Track: the observed-crop window) -- the rest of the
// model's copy keeps whatever an earlier frame left there
// [r6] host -> device on a SERVING path without a copy command: the bytes are packed into the model's pinned, device-mapped block at
// `stage_off` and a kernel fetches them (staged_upload_kernel; dst must be 16-byte aligned).  The block is rewritten only after the fetches of
// the previous call have run (host_stage_done), so the asynchronous shard entry points may return before the GPU has read it; the caller's
// buffer is free on return.
static int ensure_host_stage(fp_model *m, size_t bytes) {
  if (m->host_stage_pending) {   // the last call's fetch kernels have read the block
    FP_HIP_OK(hipEventSynchronize(m->host_stage_done));
    m->host_stage_pending = false;
  }
  if (bytes <= m->host_stage_cap) return 0;
  const size_t cap = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  uint8_t *stage = nullptr, *stage_dev = nullptr;
  if (hipHostMalloc((void **)&stage, cap, hipHostMallocMapped) != hipSuccess) { set_error("[FoundationPose] out of pinned host memory for the frame"); return 1; }
  if (hipHostGetDevicePointer((void **)&stage_dev, stage, 0) != hipSuccess) { (void)hipHostFree(stage); set_error("[FoundationPose] the pinned frame block is not device-mapped"); return 1; }
  if (m->host_stage) (void)hipHostFree(m->host_stage);
  m->host_stage = stage; m->host_stage_dev = stage_dev; m->host_stage_cap = cap;
  if (!m->host_stage_done) FP_HIP_OK(hipEventCreateWithFlags(&m->host_stage_done, hipEventDisableTiming));
  return 0;
}
static int stage_fetch(fp_model *m, void *dst_dev, const void *src_host, size_t bytes, size_t stage_off) {
  FP_CHECK(stage_off % 16 == 0 && stage_off + bytes <= m->host_stage_cap && ((uintptr_t)dst_dev & 15) == 0, "[FoundationPose] internal: misaligned staged upload");
  std::memcpy(m->host_stage + stage_off, src_host, bytes);
  const unsigned blocks = (unsigned)std::min<size_t>(1024, (bytes / 16 + 255) / 256 + 1);
  hipLaunchKernelGGL(fp::staged_upload_kernel, dim3(blocks), dim3(256), 0, m->stream, m->host_stage_dev + stage_off, (unsigned char *)dst_dev, bytes);
  FP_HIP_OK(hipGetLastError());
  return 0;
}
static int stage_fetch_done(fp_model *m) {
  FP_HIP_OK(hipEventRecord(m->host_stage_done, m->stream));
  m->host_stage_pending = true;
  return 0;
}
static size_t stage_mask_offset(size_t px) { return ((px * 3 + 63) & ~(size_t)63) + ((px * 4 + 63) & ~(size_t)63); }   // [rgb | depth | mask]

// room for a packed window of `total` bytes in the model's pinned block and its device twin (both hold the frame record in front)
static int ensure_window(fp_model *m, size_t total) {
  if (total <= m->win_cap) return 0;
  const size_t limit = (64 + (size_t)m->max_h * m->max_w * 7 / 2 + 256 + 15) & ~(size_t)15;   // a window that takes this path is at most half the frame wide
  size_t cap = std::min(limit, std::max<size_t>(total + total / 2, (size_t)256 << 10));
  cap = (cap + 15) & ~(size_t)15;   // whole 16-byte units for window_fetch_kernel
  if (total > cap) return 1;        // (larger than any window of this model's frame limits: the caller falls back to the 2-D copies)
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  uint8_t *stage = nullptr, *stage_dev = nullptr;
  FrameRef *block = nullptr;
  if (hipHostMalloc((void **)&stage, cap, hipHostMallocCoherent) != hipSuccess) { set_error("[FoundationPose] out of pinned host memory for the Track window"); return 1; }
  if (hipHostGetDevicePointer((void **)&stage_dev, stage, 0) != hipSuccess || hipMalloc((void **)&block, cap) != hipSuccess) {
    (void)hipHostFree(stage);
    set_error("[FoundationPose] out of device memory for the Track window");
    return 1;
  }
  if (m->win_stage) (void)hipHostFree(m->win_stage);
  if (m->frame_dev) (void)hipFree(m->frame_dev);
  m->win_stage = stage; m->win_stage_dev = stage_dev; m->frame_dev = block; m->win_cap = cap;
  m->frame_pub = FrameRef{};   // the new block holds no record yet
  g_alloc_epoch++;             // captured graphs read the frame through the old block's address
  return 0;
}

static int upload_frame_async(fp_model *m, const void *rgb, const void *depth, int memspace, int H, int W, int row0, int row1, int col0, int col1) {
  Target *t = nullptr;
  if (check_frame_args(m, H, W, nullptr, &t)) return 1;
  FP_CHECK(rgb && depth, "[FoundationPose] Got INVALID rgb/depth ptr");
  size_t px = (size_t)H * W;
  if (px > m->frame_cap) {
    g_alloc_epoch++;
    dev_free(m->rgb_own); dev_free(m->depth_own); dev_free(m->erode); dev_free(m->bilat); dev_free(m->xyz);
    if (dev_alloc(&m->rgb_own, px * 3) || dev_alloc(&m->depth_own, px) || dev_alloc(&m->erode, px) ||
        dev_alloc(&m->bilat, px))
      return 1;
    m->frame_cap = px;
  }
  m->H = H; m->W = W;
  m->host_stage_fresh = false;
  if (memspace == FP_DEVICE) {
    m->rgb = (const uint8_t *)rgb;
    m->depth = (const float *)depth;
    m->frame_partial = false;
  } else {
    if (row1 < 0 || row1 > H) row1 = H;
    row0 = std::max(0, std::min(row0, row1));
    m->frame_partial = row0 > 0 || row1 < H;
    if (col1 < 0 || col1 > W) col1 = W;
    col0 = std::max(0, std::min(col0, col1));
    const size_t o = (size_t)row0 * W, n = (size_t)(row1 - row0) * W;
    const size_t cw = (size_t)(col1 - col0);
    ProfScope ps(&m->prof, m->stream, "h2d_frame", 0, (double)(row1 - row0) * cw * 7);
    if (n && cw && g_upload_cols && cw * 2 <= (size_t)W) {
      // the window is less than half the frame wide: a 2-D copy of the rectangle (same device pitch: the kernels index whole frames)
      m->frame_partial = true;
      const size_t oc = o + col0, nr = (size_t)(row1 - row0);
      const size_t rgb_bytes = (nr * cw * 3 + 63) & ~(size_t)63, total = 64 + rgb_bytes + nr * cw * 4;
      if ((g_upload_cols == 1 || g_upload_cols == 3) && ensure_window(m, total) == 0) {
        // [r4] The caller's frame is pageable: a 2-D copy from it is staged inside the runtime and holds the calling thread until it
        // is done (two of them: ~48 us of a 260 us Track).  The window is packed here into the model's own pinned block instead (a
        // few hundred short memcpys, ~0.1 MB) TOGETHER with the frame record that describes it, and a small kernel fetches the block
        // over PCIe (window_fetch_kernel: a copy command of this size costs ~27 us before the graph behind it can start, the kernel
        // ~6; 2-D copies from pinned memory are no alternative at all: 1.07 ms per Track); crop_body reads the packed window
        // through the record's pitch and virtual origins.  The caller's buffers are free again when this function returns.  The
        // pinned block is reused by the next call: a model's Track is waited for (fp_track_wait) before its next submission.
        uint8_t *dev_block = reinterpret_cast<uint8_t *>(m->frame_dev);
        uint8_t *sr = m->win_stage + 64;
        float *sd = reinterpret_cast<float *>(m->win_stage + 64 + rgb_bytes);
        const uint8_t *ur = (const uint8_t *)rgb + oc * 3;
        const float *ud = (const float *)depth + oc;
        for (size_t r = 0; r < nr; r++) {
          std::memcpy(sr + r * cw * 3, ur + r * (size_t)W * 3, cw * 3);
          std::memcpy(sd + r * cw, ud + r * (size_t)W, cw * 4);
        }
        FrameRef rec;
        const long long org = (long long)row0 * (long long)cw + col0;   // packed index of frame pixel (0, 0)
        // (virtual origins lie outside the block: formed as integers, only ever dereferenced inside the window)
        rec.rgb = reinterpret_cast<const uint8_t *>(reinterpret_cast<uintptr_t>(dev_block) + 64 - (uintptr_t)(org * 3));
        rec.depth = reinterpret_cast<const float *>(reinterpret_cast<uintptr_t>(dev_block) + 64 + rgb_bytes - (uintptr_t)(org * 4));
        rec.pitch = (int)cw;
        rec.wx0 = col0; rec.wx1 = col1; rec.wy0 = row0; rec.wy1 = row1;
        std::memcpy(m->win_stage, &rec, sizeof(rec));
        if (g_upload_cols == 3) FP_HIP_OK(hipMemcpyAsync(dev_block, m->win_stage, total, hipMemcpyHostToDevice, m->stream));   // A/B: a copy command
        else {
          launch_window_fetch(m->stream, m->win_stage_dev, dev_block, total);
          FP_HIP_OK(hipGetLastError());
        }
        m->frame_pub = rec;
        m->rgb = m->rgb_own; m->depth = m->depth_own;   // (hold nothing of this frame: frame_partial)
        return 0;
      }
      FP_HIP_OK(hipMemcpy2DAsync(m->rgb_own + oc * 3, (size_t)W * 3, (const uint8_t *)rgb + oc * 3, (size_t)W * 3, cw * 3, nr,
                                 hipMemcpyHostToDevice, m->stream));
      FP_HIP_OK(hipMemcpy2DAsync(m->depth_own + oc, (size_t)W * 4, (const float *)depth + oc, (size_t)W * 4, cw * 4, nr,
                                 hipMemcpyHostToDevice, m->stream));
    } else if (n) {
      // [r6] whole rows [row0', row1) through the pinned block + a fetch kernel (row0 rounded down so that both device offsets are
      // 16-byte aligned: row0' * W % 16 == 0)
      int ra = row0;
      while (ra > 0 && ((size_t)ra * W) % 16 != 0) ra--;
      const size_t oa = (size_t)ra * W, na = (size_t)(row1 - ra) * W;
      if (ensure_host_stage(m, stage_mask_offset(px) + px)) return 1;
      if (stage_fetch(m, m->rgb_own + oa * 3, (const uint8_t *)rgb + oa * 3, na * 3, 0)) return 1;
      if (stage_fetch(m, m->depth_own + oa, (const float *)depth + oa, na * 4, (px * 3 + 63) & ~(size_t)63)) return 1;
      if (stage_fetch_done(m)) return 1;
      m->host_stage_fresh = true;
    }
    m->rgb = m->rgb_own;
    m->depth = m->depth_own;
  }
  if (m->frame_pub.rgb != m->rgb || m->frame_pub.depth != m->depth) {
    // (a ring of pinned records: the asynchronous shard entry points return before the copy has run)
    const unsigned k = m->frame_pub_count++ & 7;
    FrameRef *slot = reinterpret_cast<FrameRef *>(m->frame_pinned + 64 * k);
    *slot = FrameRef{};   // whole frame: no pitch, no window
    slot->rgb = m->rgb; slot->depth = m->depth;
    launch_window_fetch(m->stream, m->frame_pinned_dev + 64 * k, m->frame_dev, 64);   // [r6] 64 bytes by a kernel, not a copy command
    FP_HIP_OK(hipGetLastError());
    m->frame_pub = *slot;
  }
  return 0;
}

int fp_upload_frame(fp_model *m, const void *rgb, const void *depth, int memspace, int H, int W) try {
  SerialGuard serial(m ? m->device : -1);
  if (upload_frame_async(m, rgb, depth, memspace, H, W)) return 1;
  if (memspace == FP_HOST) FP_HIP_OK(hipStreamSynchronize(m->stream));  // the host frame may be released on return
  return 0;
} FP_CATCH_INT

int fp_get_xyz_map(fp_model *m, float *xyz_host) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && m->depth && xyz_host, "[FoundationPose] fp_get_xyz_map: no frame uploaded");
  FP_CHECK(!m->frame_partial, "[FoundationPose] the last call (Track from a host frame) uploaded only its crop window: call fp_upload_frame first");
  size_t px = (size_t)m->H * m->W;
  if (!m->xyz && dev_alloc(&m->xyz, m->frame_cap * 3)) return 1;
  launch_depth_to_xyz(m->stream, m->depth, m->H, m->W, m->K, m->xyz);
  FP_HIP_OK(hipMemcpyAsync(xyz_host, m->xyz, px * 12, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  return 0;
} FP_CATCH_INT

static int run_depth_filters(fp_model *m) {
  {
    ProfScope ps(&m->prof, m->stream, "erode_depth", 0, (double)m->H * m->W * 8);
    launch_erode(m->stream, m->depth, m->erode, m->H, m->W);
  }
  {
    ProfScope ps(&m->prof, m->stream, "bilateral_depth", 0, (double)m->H * m->W * 8);
    launch_bilateral(m->stream, m->erode, m->bilat, m->H, m->W);
  }
  return 0;
}

int fp_filter_depth(fp_model *m, float *eroded_out, float *bilateral_out) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && m->depth, "[FoundationPose] fp_filter_depth: no frame uploaded");
  FP_CHECK(!m->frame_partial, "[FoundationPose] the last call (Track from a host frame) uploaded only its crop window: call fp_upload_frame first");
  size_t px = (size_t)m->H * m->W;
  run_depth_filters(m);
  if (eroded_out) FP_HIP_OK(hipMemcpyAsync(eroded_out, m->erode, px * 4, hipMemcpyDeviceToHost, m->stream));
  if (bilateral_out) FP_HIP_OK(hipMemcpyAsync(bilateral_out, m->bilat, px * 4, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  return 0;
} FP_CATCH_INT

// MakeRotationGrid result, host + device copies (device: the sampler kernel stamps the translation into it)
static int set_rotation_grid(fp_model *m, int steps) {
  m->inplane_steps = steps;
  m->grid_host = make_rotation_grid(40, steps);
  g_alloc_epoch++;
  dev_free(m->grid_dev);
  if (dev_alloc(&m->grid_dev, m->grid_host.size())) return 1;
  FP_HIP_OK(fp::memcpy_sync(m->grid_dev, m->grid_host.data(), m->grid_host.size() * 4, hipMemcpyHostToDevice));
  return 0;
}

// GetHypPoses (D6F/src/foundationpose_sampling.cpp:344-394), entirely on the device: erode + bilateral, then
// GuessTranslation (bounding box of the mask, exact median of the filtered depth under it) and the hypothesis poses
// [first, first+N) of the rotation grid written to m->poses_dev.  No read-back, no synchronisation: the status word
// (sampler_status) is fetched together with the results.
static int sample_hypotheses_async(fp_model *m, Target *t, const void *mask, int memspace, int first, int N) {
  FP_CHECK(m->depth != nullptr && mask != nullptr, "[FoudationPoseSampler] Got INVALID depth/mask ptr on device!!!");
  FP_CHECK(!m->frame_partial, "[FoundationPose] the last call (Track from a host frame) uploaded only its crop window: call fp_upload_frame first");
  const size_t px = (size_t)m->H * m->W;
  if (ensure_capacity(m, N, t ? (size_t)t->mesh.V : 0)) return 1;
  if (px > m->samp_px_cap) {
    g_alloc_epoch++;
    dev_free(m->samp_vals); dev_free(m->mask_dev);
    m->samp_px_cap = 0;
    if (dev_alloc(&m->samp_vals, px) || dev_alloc(&m->mask_dev, px)) return 1;
    m->samp_px_cap = px;
  }
  if (!m->samp_state) {
    if (dev_alloc(&m->samp_state, 8)) return 1;
    const int init[8] = {0x7fffffff, -1, 0x7fffffff, -1, 0, 0, 3, 0};
    FP_HIP_OK(fp::memcpy_sync(m->samp_state, init, sizeof(init), hipMemcpyHostToDevice));
    g_alloc_epoch++;
  }
  run_depth_filters(m);
  const uint8_t *mask_d = (const uint8_t *)mask;
  if (memspace != FP_DEVICE) {  // the caller's host mask stays valid until the entry point returns (it synchronises)
    // [r6] through the pinned block (behind the frame's rgb | depth, which a host-frame call has just staged; a host mask over a DEVICE
    // frame finds the block free or waits for its last use)
    if (!m->host_stage_fresh && ensure_host_stage(m, stage_mask_offset(px) + px)) return 1;   // (fresh: this call's frame upload sized the block and waited for its last use)
    m->host_stage_fresh = false;
    if (stage_fetch(m, m->mask_dev, mask, px, stage_mask_offset(px))) return 1;
    if (stage_fetch_done(m)) return 1;
    mask_d = m->mask_dev;
  }
  ProfScope ps(&m->prof, m->stream, "sampler", 0, (double)px * 5);
  launch_sampler(m->stream, m->bilat, mask_d, m->H, m->W, FP_MIN_DEPTH, m->K, m->grid_dev, first, N, m->samp_state, m->samp_vals,
                 m->poses_dev);
  return 0;
}

// after a synchronisation: the reference's failure modes (foundationpose_sampling.cpp:269,278)
static int sampler_status(fp_model *m) {
  int st = 3;
  FP_HIP_OK(hipMemcpyAsync(&st, m->samp_state + 6, 4, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  FP_CHECK(st != 1, "[FoundationposeSampling] Mask is all zero.");
  FP_CHECK(st != 2, "[FoundationposeSampling] No valid value in mask.");
  FP_CHECK(st == 0, "[FoundationposeSampling] sampler did not run");
  return 0;
}

int fp_get_hyp_poses(fp_model *m, const void *mask, int memspace, float *poses_out, int *n_out) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && poses_out, "[FoundationPose] fp_get_hyp_poses: invalid arguments");
  const int n = m->n_hyp();
  if (sample_hypotheses_async(m, m->targets.empty() ? nullptr : &m->targets[0], mask, memspace, 0, n)) return 1;
  if (sampler_status(m)) return 1;
  FP_HIP_OK(hipMemcpyAsync(poses_out, m->poses_dev, (size_t)n * 64, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  if (n_out) *n_out = n;
  return 0;
} FP_CATCH_INT

static int upload_poses(fp_model *m, Target *t, const float *poses, int N) {
  if (ensure_capacity(m, N, (size_t)t->mesh.V)) return 1;
  if (N > m->poses_pinned_cap) {
    FP_HIP_OK(hipStreamSynchronize(m->stream));
    if (m->poses_pinned) (void)hipHostFree(m->poses_pinned);
    m->poses_pinned = nullptr; m->poses_pinned_cap = 0;
    FP_HIP_OK(hipHostMalloc((void **)&m->poses_pinned, (size_t)std::max(N, 256) * 64, hipHostMallocDefault));
    m->poses_pinned_cap = std::max(N, 256);
  }
  // every entry point synchronises the stream before it returns or before it calls this again
  std::memcpy(m->poses_pinned, poses, (size_t)N * 64);
  FP_HIP_OK(hipMemcpyAsync(m->poses_dev, m->poses_pinned, (size_t)N * 64, hipMemcpyHostToDevice, m->stream));
  return 0;
}

int fp_render_and_transform(fp_model *m, const char *target_name, const float *poses, int N, float crop_ratio,
                            float *render_out, float *transf_out, int out_memspace) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && poses && N > 0, "[FoundationposeRender] The transform matrix vector is empty");
  FP_CHECK(m->depth != nullptr, "[FoundationPose] fp_render_and_transform: no frame uploaded");
  FP_CHECK(!m->frame_partial, "[FoundationPose] the last call (Track from a host frame) uploaded only its crop window: call fp_upload_frame first");
  Target *t = m->find(target_name ? target_name : "");
  FP_CHECK(t != nullptr, "[FoundationPose] unknown target_name");
  if (upload_poses(m, t, poses, N)) return 1;
  const size_t bytes = (size_t)N * FP_CROP_HW * FP_CROP_HW * 6 * sizeof(float);
  float *a = render_out, *b = transf_out;
  if (out_memspace == FP_HOST) {
    if (ensure_blobs(m)) return 1;
    a = render_out ? m->blob_a : nullptr;
    b = transf_out ? m->blob_b : nullptr;
  }
  if (render_and_crop(m, t, N, crop_ratio, OUT_F32X6, a, b, nullptr, nullptr)) return 1;
  if (out_memspace == FP_HOST) {
    if (render_out) FP_HIP_OK(hipMemcpyAsync(render_out, a, bytes, hipMemcpyDeviceToHost, m->stream));
    if (transf_out) FP_HIP_OK(hipMemcpyAsync(transf_out, b, bytes, hipMemcpyDeviceToHost, m->stream));
  }
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  return 0;
} FP_CATCH_INT

int fp_debug_rasterize(fp_model *m, const char *target_name, const float *poses, int N, float crop_ratio,
                       int32_t *tri_id, float *rast_out) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && poses && N > 0, "[FoundationposeRender] The transform matrix vector is empty");
  FP_CHECK(m->H > 0, "[FoundationPose] fp_debug_rasterize: no frame uploaded (image size unknown)");
  Target *t = m->find(target_name ? target_name : "");
  FP_CHECK(t != nullptr, "[FoundationPose] unknown target_name");
  if (upload_poses(m, t, poses, N)) return 1;
  if (ensure_blobs(m)) return 1;
  size_t px = (size_t)m->cap * FP_CROP_HW * FP_CROP_HW;
  if (!m->dbg_tri && dev_alloc(&m->dbg_tri, px)) return 1;
  if (!m->dbg_rast && dev_alloc(&m->dbg_rast, px * 4)) return 1;
  if (render_and_crop(m, t, N, crop_ratio, OUT_F32X6, m->blob_a, nullptr, m->dbg_tri, m->dbg_rast)) return 1;
  size_t n = (size_t)N * FP_CROP_HW * FP_CROP_HW;
  if (tri_id) FP_HIP_OK(hipMemcpyAsync(tri_id, m->dbg_tri, n * 4, hipMemcpyDeviceToHost, m->stream));
  if (rast_out) FP_HIP_OK(hipMemcpyAsync(rast_out, m->dbg_rast, n * 16, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  return 0;
} FP_CATCH_INT

// blob-mode network entry points: f32 NHWC [N,160,160,6] -> packed fp16 input
static int pack_blobs(fp_model *m, const float *render_input, const float *transf_input, int memspace, int N) {
  FP_CHECK(render_input && transf_input && N > 0, "[FoundationPose] null network input");
  if (ensure_capacity(m, N, 0)) return 1;
  const size_t px = (size_t)N * FP_CROP_HW * FP_CROP_HW;
  const float *a = render_input, *b = transf_input;
  if (memspace == FP_HOST) {
    if (ensure_blobs(m)) return 1;
    FP_HIP_OK(hipMemcpyAsync(m->blob_a, render_input, px * 24, hipMemcpyHostToDevice, m->stream));
    FP_HIP_OK(hipMemcpyAsync(m->blob_b, transf_input, px * 24, hipMemcpyHostToDevice, m->stream));
    a = m->blob_a; b = m->blob_b;
  }
  launch_pack_f32x6(m->stream, a, m->nn_in, px, nn_mode(m));
  launch_pack_f32x6(m->stream, b, m->nn_in + (size_t)N * FP_NN_IN_IMG_HALFS, px, nn_mode(m));
  return 0;
}

int fp_refiner_infer(fp_model *m, const float *render_input, const float *transf_input, int memspace, int N,
                     float *trans_out, float *rot_out) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && m->refiner, "[FoundationPose] refiner weights not loaded");
  if (pack_blobs(m, render_input, transf_input, memspace, N)) return 1;
  if (refiner_forward(m->stream, &m->prof, m->refiner, m->ws, m->nn_in, N, m->trans_dev, m->rot_dev)) return 1;
  FP_HIP_OK(hipMemcpyAsync(trans_out, m->trans_dev, (size_t)N * 12, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipMemcpyAsync(rot_out, m->rot_dev, (size_t)N * 12, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  return 0;
} FP_CATCH_INT

int fp_scorer_infer(fp_model *m, const float *render_input, const float *transf_input, int memspace, int N,
                    float *scores_out) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && m->scorer, "[FoundationPose] scorer weights not loaded");
  if (pack_blobs(m, render_input, transf_input, memspace, N)) return 1;
  if (scorer_features(m->stream, &m->prof, m->scorer, m->ws, m->nn_in, N, m->feat_dev)) return 1;
  if (scorer_head(m->stream, &m->prof, m->scorer, m->ws, m->feat_dev, N, m->scores_dev)) return 1;
  FP_HIP_OK(hipMemcpyAsync(scores_out, m->scores_dev, (size_t)N * 4, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  return 0;
} FP_CATCH_INT

int fp_refine_post_process(fp_model *m, const char *target_name, const float *poses, const float *trans,
                           const float *rot, int N, float *poses_out) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && poses && trans && rot && poses_out && N > 0, "[FoundationPose] fp_refine_post_process: invalid arguments");
  Target *t = m->find(target_name ? target_name : "");
  FP_CHECK(t != nullptr, "[FoundationPose] unknown target_name");
  if (ensure_capacity(m, N, 0)) return 1;
  FP_HIP_OK(hipMemcpyAsync(m->poses_dev, poses, (size_t)N * 64, hipMemcpyHostToDevice, m->stream));
  FP_HIP_OK(hipMemcpyAsync(m->trans_dev, trans, (size_t)N * 12, hipMemcpyHostToDevice, m->stream));
  FP_HIP_OK(hipMemcpyAsync(m->rot_dev, rot, (size_t)N * 12, hipMemcpyHostToDevice, m->stream));
  launch_pose_update(m->stream, m->poses_dev, m->trans_dev, m->rot_dev, N, t->mesh.diameter);
  FP_HIP_OK(hipMemcpyAsync(poses_out, m->poses_dev, (size_t)N * 64, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  return 0;
} FP_CATCH_INT

int fp_argmax(fp_model *m, const float *scores, int N, int *index_out) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && scores && index_out && N > 0, "[FoundationPose] fp_argmax: invalid arguments");
  if (ensure_capacity(m, N, 0)) return 1;
  FP_HIP_OK(hipMemcpyAsync(m->scores_dev, scores, (size_t)N * 4, hipMemcpyHostToDevice, m->stream));
  launch_argmax(m->stream, m->scores_dev, N, m->argmax_dev);
  int idx = -1;
  FP_HIP_OK(hipMemcpyAsync(&idx, m->argmax_dev, 4, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  // the kernel flags non-finite scores with -2 (used by the sharded Register); the reference's getMaxScoreIndex always returns an
  // index in [0, N): here a NaN score is an error, never an out-of-range index
  FP_CHECK(idx >= 0 && idx < N, "[FoundationPose] fp_argmax: scores are not finite");
  *index_out = idx;
  return 0;
} FP_CATCH_INT

// one refine iteration over m->poses_dev[0..N): RefinePreProcess + SyncInfer + RefinePostProcess, all on device
// shared_b: all N poses have the same translation (fresh sampler output), so the observed crop -- which depends only on
// the translation (foundationpose_render.cpp:59, foundationpose_render.cu:78-80) -- is computed and encoded once.
// poses_in / result_out (Track): the poses are read from / the refined poses also written to host-pinned memory
static int refine_iteration(fp_model *m, Target *t, int N, bool shared_b, const float *poses_in = nullptr, float *result_out = nullptr,
                            unsigned *done_flag = nullptr) {
  RoctxRange range("refine_iteration (render + crop @1.2, refine-net, pose update)");
  const size_t half = (size_t)N * FP_NN_IN_IMG_HALFS;
  if (render_and_crop(m, t, N, 1.2f /* refine_mode_crop_ratio_ foundationpose.cpp:87 */, nn_mode(m), m->nn_in,
                      m->nn_in + half, nullptr, nullptr, shared_b ? 1 : N, poses_in))
    return 1;
  checkpoint(m, 0, m->recs, (size_t)N * sizeof(PoseRec));
  checkpoint(m, 1, m->clip, (size_t)N * t->mesh.V * 16);
  checkpoint(m, 2, m->attr, (size_t)N * t->mesh.V * 16);
  checkpoint(m, 3, m->nn_in, (size_t)N * FP_NN_IN_IMG_HALFS * 2);
  checkpoint(m, 4, m->nn_in + half, (size_t)(shared_b ? 1 : N) * FP_NN_IN_IMG_HALFS * 2);
  // N == 1 (Track): the head kernel applies RefinePostProcess itself (one launch less); profiling / digests keep the stages apart
  const PoseUpdateFuse fuse{m->poses_dev, t->mesh.diameter, poses_in, result_out, done_flag};
  bool fused = false;
  if (refiner_forward(m->stream, &m->prof, m->refiner, m->ws, m->nn_in, N, m->trans_dev, m->rot_dev, shared_b ? 1 : 0,
                      (N == 1 && !m->prof.on && !m->digests) ? &fuse : nullptr, &fused))
    return 1;
  checkpoint(m, 5, m->trans_dev, (size_t)N * 12);
  checkpoint(m, 6, m->rot_dev, (size_t)N * 12);
  if (!fused) {
    ProfScope ps(&m->prof, m->stream, "pose_update");
    launch_pose_update(m->stream, m->poses_dev, m->trans_dev, m->rot_dev, N, t->mesh.diameter, poses_in, result_out);
  }
  checkpoint(m, 7, m->poses_dev, (size_t)N * 64);
  return 0;
}

int fp_register_shard_begin(fp_model *m, const void *rgb, const void *depth, const void *mask, int memspace, int H,
                            int W, const char *target_name, int refine_itr, int shard_begin, int shard_count,
                            float **feat_dev, float **poses_dev) try {
  SerialGuard serial(m ? m->device : -1);
  RoctxRange range("fp_register_shard_begin (sampler + refine + score trunk)");
  Target *t = nullptr;
  if (check_frame_args(m, H, W, target_name ? target_name : "", &t)) return 1;
  FP_CHECK(m->refiner && m->scorer, "[FoundationPose] refiner/scorer weights not loaded");
  FP_CHECK(mask != nullptr, "[FoundationPose] Register needs a mask");
  const int n_all = m->n_hyp();
  FP_CHECK(shard_begin >= 0 && shard_count > 0 && shard_begin + shard_count <= n_all,
           "[FoundationPose] hypothesis shard out of range");
  const bool graphable = m->use_graphs && !m->prof.on && !m->digests && !m->calibrating && refine_itr >= 1;
  if (upload_frame_async(m, rgb, depth, memspace, H, W)) return 1;
  const int N = shard_count;
  if (sample_hypotheses_async(m, t, mask, memspace, shard_begin, N)) return 1;
  const size_t half = (size_t)N * FP_NN_IN_IMG_HALFS;
  if (run_graphed(m, m->rg, t, H, W, refine_itr, N, graphable, [&]() {
        for (int it = 0; it < refine_itr; it++)
          if (refine_iteration(m, t, N, it == 0 && N > 1)) return 1;  // sampler output: one translation for all hypotheses
        {
          RoctxRange r2("render + crop @1.1 (score mode)");
          if (render_and_crop(m, t, N, 1.1f /* score_mode_crop_ratio_ foundationpose.cpp:88 */, nn_mode(m), m->nn_in,
                              m->nn_in + half, nullptr, nullptr))
            return 1;
        }
        RoctxRange r3("score-net trunk + self-attention");
        checkpoint(m, 8, m->clip, (size_t)N * t->mesh.V * 16);
        checkpoint(m, 9, m->attr, (size_t)N * t->mesh.V * 16);
        checkpoint(m, 10, m->nn_in, (size_t)N * FP_NN_IN_IMG_HALFS * 2);
        checkpoint(m, 11, m->nn_in + half, (size_t)N * FP_NN_IN_IMG_HALFS * 2);
        if (scorer_features(m->stream, &m->prof, m->scorer, m->ws, m->nn_in, N, m->feat_dev)) return 1;
        checkpoint(m, 12, m->feat_dev, (size_t)N * 512 * 4);
        return 0;
      }))
    return 1;
  if (feat_dev) *feat_dev = m->feat_dev;
  if (poses_dev) *poses_dev = m->poses_dev;
  if (m->defer_begin_sync) return 0;  // fp_register_ex: one synchronisation at the very end
  // synchronises: the returned buffers are complete (and the GPU lock may be released); reports the sampler's verdict
  if (sampler_status(m)) {
    set_error(std::string("[FoundationPose] Failed to generate hyp poses!!! ") + g_last_error);
    return 1;
  }
  return 0;
} FP_CATCH_INT

int fp_register_shard_finish(fp_model *m, const float *all_feat_dev, const float *all_poses_dev, int N_total,
                             float out_pose[16], int *best_index, float *scores_host) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && m->scorer && all_feat_dev && all_poses_dev && N_total > 0 && out_pose,
           "[FoundationPose] fp_register_shard_finish: invalid arguments");
  RoctxRange range("fp_register_shard_finish (cross-hypothesis head + arg-max)");
  float *scores = m->scores_dev;
  if (N_total > m->cap) {  // gathered hypotheses of all ranks: a persistent buffer, not a malloc/free per Register
    if (N_total > m->scores_all_cap) {
      dev_free(m->scores_all);
      m->scores_all_cap = 0;
      if (dev_alloc(&m->scores_all, (size_t)N_total)) return 1;
      m->scores_all_cap = N_total;
    }
    scores = m->scores_all;
  }
  if (!m->best_pose_dev && dev_alloc(&m->best_pose_dev, 16)) return 1;
  if (!m->result_pinned) {
    FP_HIP_OK(hipHostMalloc((void **)&m->result_pinned, 64 * sizeof(int), hipHostMallocMapped));
    FP_HIP_OK(hipHostGetDevicePointer((void **)&m->result_pinned_dev, m->result_pinned, 0));
  }
  int rc = scorer_head(m->stream, &m->prof, m->scorer, m->ws, all_feat_dev, N_total, scores);
  if (!rc) checkpoint(m, 13, scores, (size_t)N_total * 4);
  if (!rc) {
    ProfScope ps(&m->prof, m->stream, "argmax");
    launch_argmax(m->stream, scores, N_total, m->argmax_dev, all_poses_dev, m->best_pose_dev);
  }
  // ONE synchronisation: winner index, the sampler's status word (when this call also ran the sampler) and the pose
  // [r6] written by a kernel into the pinned, device-mapped block: no device -> host copy command in front of the synchronisation
  int *res = m->result_pinned;
  res[0] = -3; res[1] = 3;   // (overwritten by the kernel; what a kernel that never ran would leave is an error, not a stale winner)
  if (!rc) {
    hipLaunchKernelGGL(fp::publish_result_kernel, dim3(1), dim3(64), 0, m->stream, m->argmax_dev, m->defer_begin_sync ? m->samp_state + 6 : nullptr, m->best_pose_dev, m->result_pinned_dev);
    if (hipGetLastError() != hipSuccess) rc = 1;
  }
  if (!rc && scores_host &&
      hipMemcpyAsync(scores_host, scores, (size_t)N_total * 4, hipMemcpyDeviceToHost, m->stream) != hipSuccess)
    rc = 1;
  if (!rc && hipStreamSynchronize(m->stream) != hipSuccess) rc = 1;
  if (!rc) {
    // the sampler's verdict first: on failure the caller's pose is left untouched, like the reference, which returns
    // false before it writes out_pose_in_mesh (foundationpose.cpp:196-201)
    if (res[1] == 1) { set_error("[FoundationPose] Failed to generate hyp poses!!! [FoundationposeSampling] Mask is all zero."); rc = 1; }
    else if (res[1] == 2) { set_error("[FoundationPose] Failed to generate hyp poses!!! [FoundationposeSampling] No valid value in mask."); rc = 1; }
    else if (res[1] != 0) { set_error("[FoundationPose] Failed to generate hyp poses!!! sampler did not run"); rc = 1; }
    else if (res[0] == -2) { set_error("[FoundationPose] scores are not finite (a rank of a sharded Register reported a failed shard, or the weights are broken)"); rc = 1; }
  }
  if (!rc) {
    std::memcpy(out_pose, &res[2], 64);
    if (best_index) *best_index = res[0];
  }
  if (rc && g_last_error.empty()) set_error("[FoundationPose] fp_register_shard_finish failed");
  return rc;
} FP_CATCH_INT

int fp_download(fp_model *m, void *dst_host, const void *src_dev, size_t bytes) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && dst_host && src_dev, "[FoundationPose] fp_download: invalid arguments");
  if (bytes) FP_HIP_OK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  return 0;
} FP_CATCH_INT

// Sharded Register without host stalls: everything is enqueued on the model's stream, nothing is allocated or synchronised.
int fp_register_shard_begin_packed(fp_model *m, const void *rgb, const void *depth, const void *mask, int memspace, int H,
                                   int W, const char *target_name, int refine_itr, int shard_begin, int shard_count,
                                   float *packed_dev, int rows_per_rank) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && packed_dev && rows_per_rank >= shard_count && shard_count >= 0, "[FoundationPose] fp_register_shard_begin_packed: invalid arguments");
  float *feat = nullptr, *poses = nullptr;
  if (shard_count > 0) {
    m->defer_begin_sync = true;
    int rc = fp_register_shard_begin(m, rgb, depth, mask, memspace, H, W, target_name, refine_itr, shard_begin, shard_count, &feat, &poses);
    m->defer_begin_sync = false;
    if (rc) { (void)hipStreamSynchronize(m->stream); return 1; }
    m->shard_sampler_pending = true;
  } else {
    // an empty shard still runs the (cheap) sampler, so that a bad mask fails on EVERY rank alike
    Target *t = nullptr;
    if (check_frame_args(m, H, W, target_name ? target_name : "", &t)) return 1;
    FP_CHECK(mask != nullptr, "[FoundationPose] Register needs a mask");
    if (upload_frame_async(m, rgb, depth, memspace, H, W) || sample_hypotheses_async(m, t, mask, memspace, 0, 1)) {
      (void)hipStreamSynchronize(m->stream);
      return 1;
    }
    m->shard_sampler_pending = true;
  }
  const int n = rows_per_rank * 528;
  hipLaunchKernelGGL(pack_shard_kernel, dim3((n + 255) / 256), dim3(256), 0, m->stream, feat, poses, shard_count, rows_per_rank, packed_dev, m->samp_state + 6);
  FP_HIP_OK(hipGetLastError());
  return 0;
} FP_CATCH_INT

int fp_register_shard_finish_packed(fp_model *m, const float *gathered_dev, int n_total, float out_pose[16], int *best_index) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && gathered_dev && n_total > 0 && out_pose, "[FoundationPose] fp_register_shard_finish_packed: invalid arguments");
  if (n_total > m->gath_cap) {
    FP_HIP_OK(hipStreamSynchronize(m->stream));
    dev_free(m->gath_feat); dev_free(m->gath_poses);
    m->gath_cap = 0;
    if (dev_alloc(&m->gath_feat, (size_t)n_total * 512) || dev_alloc(&m->gath_poses, (size_t)n_total * 16)) return 1;
    m->gath_cap = n_total;
  }
  const int n = n_total * 528;
  hipLaunchKernelGGL(unpack_shards_kernel, dim3((n + 255) / 256), dim3(256), 0, m->stream, gathered_dev, n_total, m->gath_feat, m->gath_poses);
  // the sampler's verdict of this rank's begin (if it had a shard) is fetched together with the result
  m->defer_begin_sync = m->shard_sampler_pending;
  m->shard_sampler_pending = false;
  int rc = fp_register_shard_finish(m, m->gath_feat, m->gath_poses, n_total, out_pose, best_index, nullptr);
  m->defer_begin_sync = false;
  return rc;
} FP_CATCH_INT

// ---- native sharded Register: begin -> ONE ncclAllGather (RCCL over xGMI) on the model's stream -> finish.  No torch, no events
// across runtimes: the collective is ordered by the stream itself.  RCCL is bound at first use with dlopen / dlsym -- first an
// already loaded copy (a Python host has torch's bundled librccl in the process: a communicator made by torch.distributed belongs to
// THAT copy), otherwise librccl.so.1 of the ROCm installation -- so that the library itself has no link-time RCCL dependency and
// geometry-only / single-GPU users never load it.
namespace {
typedef int (*nccl_allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
typedef const char *(*nccl_errstr_fn)(int);
typedef int (*nccl_count_fn)(void *, int *);
typedef int (*nccl_abort_fn)(void *);
struct RcclApi {
  nccl_abort_fn abort = nullptr;   // optional: lets a rank that cannot join the collective release the others
  nccl_allgather_fn all_gather = nullptr;
  nccl_errstr_fn err = nullptr;
  nccl_count_fn count = nullptr, user_rank = nullptr;
  std::string why;
};
const RcclApi &rccl_api() {
  static const RcclApi api = [] {
    RcclApi a;
    void *h = nullptr;
    // a copy that is already in the process, whatever its file name / soname (torch wheels ship `torch/lib/librccl.so`): the
    // communicator the caller hands over was made by THAT copy
    std::string loaded;
    dl_iterate_phdr([](struct dl_phdr_info *info, size_t, void *out) {
      if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl")) { *(std::string *)out = info->dlpi_name; return 1; }
      return 0;
    }, &loaded);
    if (!loaded.empty()) h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char *name : {"librccl.so.1", "librccl.so"}) if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      const char *why = dlerror();   // (one call: dlerror clears the message it returns)
      a.why = std::string("cannot load librccl: ") + (why ? why : "?");
      return a;
    }
    a.all_gather = (nccl_allgather_fn)dlsym(h, "ncclAllGather");
    a.err = (nccl_errstr_fn)dlsym(h, "ncclGetErrorString");
    a.count = (nccl_count_fn)dlsym(h, "ncclCommCount");
    a.user_rank = (nccl_count_fn)dlsym(h, "ncclCommUserRank");
    a.abort = (nccl_abort_fn)dlsym(h, "ncclCommAbort");
    if (!a.all_gather || !a.err || !a.count || !a.user_rank) { a.why = "librccl lacks ncclAllGather / ncclCommCount / ncclCommUserRank"; a.all_gather = nullptr; }
    return a;
  }();
  return api;
}
__global__ void poison_rows_kernel(float *p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = __int_as_float(0x7fc00000);
}
}  // namespace

int fp_register_sharded(fp_model *m, void *nccl_comm, const void *rgb, const void *depth, const void *mask, int memspace, int H, int W,
                        const char *target_name, int refine_itr, float out_pose[16], int *best_index) try {
  FP_CHECK(m && nccl_comm && out_pose, "[FoundationPose] fp_register_sharded: invalid arguments");
  const RcclApi &nccl = rccl_api();
  FP_CHECK(nccl.all_gather != nullptr, "[FoundationPose] fp_register_sharded: " + nccl.why);
  int world = 0, rank = -1, rc;
  if ((rc = nccl.count(nccl_comm, &world)) != 0 || (rc = nccl.user_rank(nccl_comm, &rank)) != 0 || world < 1 || rank < 0 || rank >= world) {
    set_error(std::string("[FoundationPose] fp_register_sharded: bad communicator: ") + (rc ? nccl.err(rc) : "rank / size out of range"));
    return 1;
  }
  struct NoSerial {   // (see g_no_serial; restored on every return path)
    bool prev = g_no_serial;
    explicit NoSerial(bool off) { if (off) g_no_serial = true; }
    ~NoSerial() { g_no_serial = prev; }
  } no_serial(world > 1);
  SerialGuard serial(m->device);
  const int n_total = m->n_hyp();
  const int per = (n_total + world - 1) / world;
  const int begin = std::min(rank * per, n_total), count = std::min(per, n_total - begin);
  // Everything that can fail on THIS rank alone happens behind a promise to the other ranks: they are already heading for the
  // collective.  A failure of the begin half joins it with NaN rows (every rank's finish reports them) and returns the error
  // afterwards; a rank that cannot even hold its exchange buffers cannot join -- it aborts the communicator (ncclCommAbort: the
  // others' collective returns an error instead of hanging) and says so.
  auto cannot_join = [&](const std::string &why) {
    const bool aborted = world > 1 && nccl.abort && nccl.abort(nccl_comm) == 0;
    set_error("[FoundationPose] fp_register_sharded: " + why + (world > 1 ? (aborted ? " -- the communicator was aborted so that the other ranks do not wait for this one (it must be re-created)"
                                                                                      : " -- this rank could not join the collective and ncclCommAbort is unavailable: the other ranks may wait") : ""));
    return 1;
  };
  // persistent exchange buffers (grown on demand; a Register never allocates in steady state)
  const size_t need_send = (size_t)per * 528, need_recv = (size_t)world * per * 528;
  if (need_send > m->shard_send_cap || need_recv > m->shard_recv_cap) {
    if (hipStreamSynchronize(m->stream) != hipSuccess) return cannot_join("the model's stream is in an error state");
    if (need_send > m->shard_send_cap) {
      dev_free(m->shard_send); m->shard_send_cap = 0;
      if (hipMalloc((void **)&m->shard_send, need_send * sizeof(float)) != hipSuccess) { m->shard_send = nullptr; return cannot_join("out of device memory for the exchange buffers"); }
      m->shard_send_cap = need_send;
    }
    if (need_recv > m->shard_recv_cap) {
      dev_free(m->shard_recv); m->shard_recv_cap = 0;
      if (hipMalloc((void **)&m->shard_recv, need_recv * sizeof(float)) != hipSuccess) { m->shard_recv = nullptr; return cannot_join("out of device memory for the exchange buffers"); }
      m->shard_recv_cap = need_recv;
    }
  }
  const int rc_begin = fp_register_shard_begin_packed(m, rgb, depth, mask, memspace, H, W, target_name, refine_itr, begin, count, m->shard_send, per);
  std::string begin_error;
  if (rc_begin) {
    begin_error = g_last_error;
    m->shard_sampler_pending = false;
    hipLaunchKernelGGL(poison_rows_kernel, dim3((unsigned)((need_send + 255) / 256)), dim3(256), 0, m->stream, m->shard_send, need_send);
  }
  if (world > 1) {
    rc = nccl.all_gather(m->shard_send, m->shard_recv, need_send, 7 /* ncclFloat32 */, nccl_comm, m->stream);
    if (rc != 0) {
      (void)hipStreamSynchronize(m->stream);
      set_error(std::string("[FoundationPose] ncclAllGather failed: ") + nccl.err(rc) + (rc_begin ? " (after: " + begin_error + ")" : ""));
      return 1;
    }
  } else {
    FP_HIP_OK(hipMemcpyAsync(m->shard_recv, m->shard_send, need_send * sizeof(float), hipMemcpyDeviceToDevice, m->stream));
  }
  if (rc_begin) {
    (void)hipStreamSynchronize(m->stream);
    set_error(begin_error);
    return 1;
  }
  return fp_register_shard_finish_packed(m, m->shard_recv, n_total, out_pose, best_index);
} FP_CATCH_INT

int fp_register_ex(fp_model *m, const void *rgb, const void *depth, const void *mask, int memspace, int H, int W,
                   const char *target_name, int refine_itr, float out_pose[16]) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m != nullptr, "[FoundationPose] null model");
  float *feat = nullptr, *poses = nullptr;
  m->defer_begin_sync = true;  // begin + finish back to back on one stream: a single synchronisation, at the end
  int rc = fp_register_shard_begin(m, rgb, depth, mask, memspace, H, W, target_name, refine_itr, 0, m->n_hyp(), &feat, &poses);
  if (!rc) rc = fp_register_shard_finish(m, feat, poses, m->n_hyp(), out_pose, nullptr, nullptr);
  else (void)hipStreamSynchronize(m->stream);
  m->defer_begin_sync = false;
  return rc;
} FP_CATCH_INT

int fp_register(fp_model *m, const uint8_t *rgb, const float *depth, const uint8_t *mask, int H, int W,
                const char *target_name, int refine_itr, float out_pose[16]) try {
  return fp_register_ex(m, rgb, depth, mask, FP_HOST, H, W, target_name, refine_itr, out_pose);
} FP_CATCH_INT

// Track in two halves: everything is ENQUEUED by fp_track_submit (frame upload, the replayed graph); fp_track_wait synchronises the
// model's stream and hands the pose over.  One host thread can so keep several models (objects) in flight at once.
static int track_submit_impl(fp_model *m, const void *rgb, const void *depth, int memspace, int H, int W, const float hyp_pose[16],
                             const char *target_name, int refine_itr) {
  RoctxRange range("fp_track_submit");
  Target *t = nullptr;
  if (check_frame_args(m, H, W, target_name ? target_name : "", &t)) return 1;
  FP_CHECK(m->refiner, "[FoundationPose] refiner weights not loaded");
  FP_CHECK(hyp_pose, "[FoundationPose] Track: null pose");
  FP_CHECK(!m->track_pending, "[FoundationPose] fp_track_submit: the previous submission has not been waited for");
  const bool graphable = m->use_graphs && !m->prof.on && !m->digests && !m->calibrating && refine_itr >= 1;
  // a single refine iteration reads the frame only inside the observed-crop window of the hypothesis (ComputeCropWindowTF,
  // foundationpose_render.cpp:25-70, restated on the host in double with a margin): host frames upload just those rows
  int row0 = 0, row1 = -1, col0 = 0, col1 = -1;
  if (memspace != FP_DEVICE && refine_itr == 1) {
    const double r = (double)t->mesh.diameter * 1.2 / 2, tx = hyp_pose[12], ty = hyp_pose[13], tz = hyp_pose[14];
    auto proj_v = [&](double x, double y, double z) {
      const double q1 = m->K[3] * x + m->K[4] * y + m->K[5] * z, q2 = m->K[6] * x + m->K[7] * y + m->K[8] * z;
      return q1 / q2;
    };
    auto proj_u = [&](double x, double y, double z) {
      const double q0 = m->K[0] * x + m->K[1] * y + m->K[2] * z, q2 = m->K[6] * x + m->K[7] * y + m->K[8] * z;
      return q0 / q2;
    };
    if (tz > 1e-6) {
      const double v0 = proj_v(tx, ty, tz);
      double rad = 0;
      const double offs[4][2] = {{r, 0}, {-r, 0}, {0, r}, {0, -r}};
      for (auto &o : offs) rad = std::max(rad, std::fabs(proj_v(tx + o[0], ty + o[1], tz) - v0));
      // far outside [-H, 2H] (tiny tz, huge translation) the casts below would overflow: such a window either misses the frame
      // (decided in double) or the whole frame is uploaded
      if (std::isfinite(v0) && std::isfinite(rad) && rad < 4.0 * H) {
        if (v0 + rad + 5 <= 0 || v0 - rad - 4 >= H) { row0 = 0; row1 = 0; }   // window outside the frame: nothing is read
        else if (v0 > -(double)H && v0 < 2.0 * H) {
          row0 = (int)std::floor(v0 - rad) - 4;
          row1 = (int)std::ceil(v0 + rad) + 5;
          // the window is a square of the same radius around (u0, v0): its columns, with the same margin
          const double u0 = proj_u(tx, ty, tz);
          if (std::isfinite(u0) && u0 > -(double)W && u0 < 2.0 * W) {
            col0 = (int)std::floor(u0 - rad) - 4;
            col1 = (int)std::ceil(u0 + rad) + 5;
            if (col1 <= 0 || col0 >= W) { col0 = 0; col1 = -1; }   // (a window beside the frame: keep whole rows)
          }
        }
      }
    }
  }
  if (upload_frame_async(m, rgb, depth, memspace, H, W, row0, row1, col0, col1)) return 1;
  if (!m->track_io) {
    FP_HIP_OK(hipHostMalloc((void **)&m->track_io, 48 * sizeof(float), hipHostMallocCoherent));   // (fine-grained: the done flag is read while the graph is still running)
    FP_HIP_OK(hipHostGetDevicePointer((void **)&m->track_io_dev, m->track_io, 0));
  }
  m->track_flag_armed = false;
  if (refine_itr <= 0) {  // no refinement requested: the hypothesis is the answer (the reference's loop runs zero times)
    std::memcpy(m->track_io + 16, hyp_pose, 64);
    m->track_pending = true;
    return 0;
  }
  if (ensure_capacity(m, 1, (size_t)t->mesh.V)) return 1;
  // the hypothesis goes in and the refined pose comes out through host-pinned memory the kernels address directly: no copy
  // kernels around the graph (two of the ~50 launches of a Track)
  std::memcpy(m->track_io, hyp_pose, 64);
  // completion flag (own cache line of the pinned block): cleared here, raised by the last kernel after it stored the refined pose.
  // Armed only when that kernel is the fused head + pose-update kernel (not while profiling / taking digests: separate stages there)
  volatile unsigned *flag = reinterpret_cast<volatile unsigned *>(m->track_io + 32);
  *flag = 0u;
  const bool armed = !m->prof.on && !m->digests && refiner_fuses_pose(m->refiner);
  if (run_graphed(m, m->tg, t, H, W, refine_itr, 1, graphable, [&]() {
        for (int it = 0; it < refine_itr; it++) {
          const bool last = it == refine_itr - 1;
          if (refine_iteration(m, t, 1, false, it == 0 ? m->track_io_dev : nullptr, last ? m->track_io_dev + 16 : nullptr,
                               last ? reinterpret_cast<unsigned *>(m->track_io_dev + 32) : nullptr))
            return 1;
        }
        return 0;
      }))
    return 1;
  m->track_flag_armed = armed;
  m->track_pending = true;
  return 0;
}

int fp_track_submit(fp_model *m, const void *rgb, const void *depth, int memspace, int H, int W, const float hyp_pose[16],
                    const char *target_name, int refine_itr) try {
  SerialGuard serial(m ? m->device : -1);
  const int rc = track_submit_impl(m, rgb, depth, memspace, H, W, hyp_pose, target_name, refine_itr);
  // a failure after the upload was enqueued must not leave H2D copies of the caller's host frame in flight
  if (rc && m && m->stream) (void)hipStreamSynchronize(m->stream);
  return rc;
} FP_CATCH_INT

int fp_track_wait(fp_model *m, float out_pose[16]) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m && out_pose, "[FoundationPose] fp_track_wait: invalid arguments");
  FP_CHECK(m->track_pending, "[FoundationPose] fp_track_wait: nothing was submitted");
  m->track_pending = false;
  if (m->track_flag_armed) {
    // the last kernel of the Track raises a flag in host-pinned memory right after it stored the pose there: poll that instead of the
    // stream (later work on the model's stream is ordered behind the graph anyway).  Bounded: a fault, or a graph captured under
    // other settings, falls through to the stream wait, which also reports the error.
    m->track_flag_armed = false;
    volatile unsigned *flag = reinterpret_cast<volatile unsigned *>(m->track_io + 32);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    // (a raised flag IS the proof of success: the kernel that raises it is the last of the chain and a faulted stream never runs it;
    // spin briefly -- a Track takes ~0.2 ms -- then yield the core between looks)
    bool yielding = false;
    while (*flag == 0u) {
      if (yielding) std::this_thread::yield();
      else {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        asm volatile("yield");
#endif
      }
      if ((++spins & 1023u) == 0 || yielding) {
        const auto waited = std::chrono::steady_clock::now() - t0;
        if (waited > std::chrono::milliseconds(20)) break;
        yielding = waited > std::chrono::microseconds(500);
      }
    }
    if (*flag != 0u) {
      std::atomic_thread_fence(std::memory_order_acquire);
      std::memcpy(out_pose, m->track_io + 16, 64);
      return 0;
    }
  }
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  std::memcpy(out_pose, m->track_io + 16, 64);
  return 0;
} FP_CATCH_INT

// Track of K objects of one frame in ONE batch: the geometry runs per object (its mesh), the refine-net once over all K crops.
// Track is launch-latency-bound at N = 1, so K objects cost little more than one (tools/bench_multi_track.py).
int fp_track_multi(fp_model *m, const void *rgb, const void *depth, int memspace, int H, int W, int K, const float *hyp_poses,
                   const char *const *target_names, int refine_itr, float *out_poses) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m != nullptr, "[FoundationPose] null model");
  FP_CHECK(K >= 1 && K <= 64 && hyp_poses && target_names && out_poses, "[FoundationPose] fp_track_multi: invalid arguments (1..64 objects)");
  FP_CHECK(m->refiner, "[FoundationPose] refiner weights not loaded");
  FP_CHECK(!m->track_pending, "[FoundationPose] fp_track_multi: a submitted Track has not been waited for");
  std::vector<Target *> targets(K);
  size_t maxV = 0;
  for (int i = 0; i < K; i++) {
    Target *t = nullptr;
    if (check_frame_args(m, H, W, target_names[i] ? target_names[i] : "", &t)) return 1;
    targets[i] = t;
    maxV = std::max(maxV, (size_t)t->mesh.V);
  }
  if (upload_frame_async(m, rgb, depth, memspace, H, W)) return 1;
  if (refine_itr <= 0) {
    FP_HIP_OK(hipStreamSynchronize(m->stream));
    std::memcpy(out_poses, hyp_poses, (size_t)K * 64);
    return 0;
  }
  if (ensure_capacity(m, K, maxV)) return 1;
  if (K > m->multi_io_cap) {
    FP_HIP_OK(hipStreamSynchronize(m->stream));
    if (m->multi_io) (void)hipHostFree(m->multi_io);
    m->multi_io = nullptr; m->multi_io_cap = 0;
    g_alloc_epoch++;   // graphs bake the pinned addresses
    FP_HIP_OK(hipHostMalloc((void **)&m->multi_io, (size_t)2 * 64 * 16 * sizeof(float), hipHostMallocDefault));
    FP_HIP_OK(hipHostGetDevicePointer((void **)&m->multi_io_dev, m->multi_io, 0));
    m->multi_io_cap = 64;
  }
  std::memcpy(m->multi_io, hyp_poses, (size_t)K * 64);
  const bool graphable = m->use_graphs && !m->prof.on && !m->digests && !m->calibrating;
  if (m->mg_sig != targets) { drop_graph(m->mg); m->mg.target = nullptr; m->mg_sig = targets; }
  float *pin_in = m->multi_io_dev, *pin_out = m->multi_io_dev + 64 * 16;
  const size_t IMG = FP_NN_IN_IMG_HALFS;
  if (run_graphed(m, m->mg, targets[0], H, W, refine_itr, K, graphable, [&]() {
        for (int it = 0; it < refine_itr; it++) {
          for (int o = 0; o < K;) {   // groups of consecutive objects with the same mesh
            int n = 1;
            while (o + n < K && targets[o + n] == targets[o]) n++;
            if (render_and_crop(m, targets[o], n, 1.2f, nn_mode(m), m->nn_in + (size_t)o * IMG, m->nn_in + (size_t)(K + o) * IMG, nullptr, nullptr, n,
                                (it == 0 ? pin_in : m->poses_dev) + (size_t)o * 16, m->recs + o))
              return 1;
            o += n;
          }
          if (refiner_forward(m->stream, &m->prof, m->refiner, m->ws, m->nn_in, K, m->trans_dev, m->rot_dev, 0)) return 1;
          for (int o = 0; o < K;) {
            int n = 1;
            while (o + n < K && targets[o + n] == targets[o]) n++;
            launch_pose_update(m->stream, m->poses_dev + (size_t)o * 16, m->trans_dev + (size_t)o * 3, m->rot_dev + (size_t)o * 3, n,
                               targets[o]->mesh.diameter, it == 0 ? pin_in + (size_t)o * 16 : nullptr,
                               it == refine_itr - 1 ? pin_out + (size_t)o * 16 : nullptr);
            o += n;
          }
        }
        return 0;
      }))
    return 1;
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  std::memcpy(out_poses, m->multi_io + 64 * 16, (size_t)K * 64);
  return 0;
} FP_CATCH_INT

int fp_track_ex(fp_model *m, const void *rgb, const void *depth, int memspace, int H, int W, const float hyp_pose[16],
                const char *target_name, int refine_itr, float out_pose[16]) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(out_pose, "[FoundationPose] Track: null pose");
  // an earlier fp_track_submit that is still in flight is the caller's to wait for: refuse without touching it
  FP_CHECK(!m || !m->track_pending, "[FoundationPose] Track: a submitted Track has not been waited for (fp_track_wait)");
  if (fp_track_submit(m, rgb, depth, memspace, H, W, hyp_pose, target_name, refine_itr)) {
    if (m) m->track_pending = false;   // this call's own submission failed (fp_track_submit synchronised)
    return 1;
  }
  return fp_track_wait(m, out_pose);
} FP_CATCH_INT

int fp_track(fp_model *m, const uint8_t *rgb, const float *depth, int H, int W, const float hyp_pose[16],
             const char *target_name, int refine_itr, float out_pose[16]) try {
  return fp_track_ex(m, rgb, depth, FP_HOST, H, W, hyp_pose, target_name, refine_itr, out_pose);
} FP_CATCH_INT

int fp_set_precision(fp_model *m, int precision) try {
  FP_CHECK(m != nullptr, "[FoundationPose] null model");
  FP_CHECK(precision == PREC_F16 || precision == PREC_BF16 || precision == PREC_FP8 || precision == PREC_INT8, "[FoundationPose] unknown precision");
  // (checked BEFORE anything is prepared: an uncalibrated 8-bit precision must not cost 0.3-1 s of host work, an exclusive section and
  // two resident networks just to be refused.  m->calibrated only changes under the exclusive lock, which a racing calibration of the
  // same model would be a caller error anyway: models are not re-entrant)
  FP_CHECK(!(precision == PREC_FP8 || precision == PREC_INT8) || m->calibrated(precision),
           "[FoundationPose] an 8-bit precision needs its calibration: call fp_calibrate for it (or fp_set_calibration_blob with its record) first");
  PreparedNets nets;   // (host phase of a first selection outside the lock)
  if (nets.prepare(m->refiner_path, m->refiner_p[precision] != nullptr, m->scorer_path, m->scorer_p[precision] != nullptr, precision)) return 1;
  LifeExclusive life;   // may load the networks of a precision: as exclusive as fp_create
  DeviceScope on_device(m->device);
  if (nets.adopt(m)) return 1;
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  return select_precision(m, precision);
} FP_CATCH_INT
int fp_get_precision(const fp_model *m) { return m ? m->prec : -1; }

int fp_set_float_model(fp_model *m, int fmad) try {
  LifeExclusive life;   // destroys the captured graphs: not while another thread is inside a call
  FP_CHECK(m != nullptr, "[FoundationPose] null model");
  DeviceScope on_device(m->device);
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  m->fmad = fmad != 0;
  invalidate_graphs(m);
  return 0;
} FP_CATCH_INT
int fp_get_float_model(const fp_model *m) { return m ? (m->fmad ? 1 : 0) : -1; }

// Post-training static quantisation for the 8-bit precisions (FP_PREC_FP8 / FP_PREC_INT8) over K >= 1 calibration frames
// (fp_calibrate_begin / fp_calibrate_add_frame / fp_calibrate_finish; fp_calibrate = the three with one frame):
//   1. one Register per frame in f16 records, per channel of the 15 trunk activations of both networks, |max| and the mean OVER ALL
//      FRAMES (the statistics buffers accumulate across the Registers of a pass: integer atomics, so the result does not depend on order);
//   2. the 8-bit networks are quantised with per-channel activation scales folded into their weights (net_apply_q8);
//   3. bias correction (Nagel et al., "Data-free quantization through weight equalization and bias correction", 2019 -- here with
//      data): layer by layer, in trunk order, a pass of the 8-bit model over the frames measures the per-channel mean of the layer's
//      output and the difference to the f16 mean goes into its bias (two sweeps over the 13 layers);
//   4. what is left of the mean shift of the TOKEN tensor goes into the positional table;
//   5. what is left at the OUTPUTS (means over hypotheses and frames) goes into the output layers' biases.
// Why several frames [r5]: the common-mode part of the correction (steps 4, 5) measured on ONE frame is partly that frame's own
// (round 4: 0.6-0.7 mm of common-mode shift on other frames, 82-90 % of the refined poses within 1 mm / 1 deg); solved over frames of
// the deployment's scene family it is the family's mean and carries over to frames the calibration never saw.
// The record of the precision is replaced only when every step succeeded; a failure restores the previous record (or leaves the
// precision uncalibrated).  The pose is discarded; the model's precision is unchanged.  ~30 Registers per frame, once per deployment.
static constexpr int kCalibSlots = 16;   // frame-mean slots a record carries (more frames share slots: frame f -> slot f % 16)
struct CalibRecord {   // what fp_get_calibration_blob carries for one precision
  std::vector<float> amax[2], bias_fix[2], tok_fix[2], out_fix[2];
  std::vector<float> fmeans[2];   // [slots][15][512]
  int slots = 0;
  bool valid() const { return !amax[0].empty(); }
};
static CalibRecord record_of(const fp_model *m, int precision) {
  CalibRecord r;
  for (int k = 0; k < 2; k++) { r.amax[k] = m->calib_amax_q[precision][k]; r.bias_fix[k] = m->calib_bias_fix[precision][k]; r.tok_fix[k] = m->calib_tok_fix[precision][k]; r.out_fix[k] = m->calib_out_fix[precision][k]; r.fmeans[k] = m->calib_fmeans[precision][k]; }
  r.slots = m->calib_slots[precision];
  return r;
}
static void commit_record(fp_model *m, int precision, const CalibRecord &r) {
  for (int k = 0; k < 2; k++) { m->calib_amax_q[precision][k] = r.amax[k]; m->calib_bias_fix[precision][k] = r.bias_fix[k]; m->calib_tok_fix[precision][k] = r.tok_fix[k]; m->calib_out_fix[precision][k] = r.out_fix[k]; m->calib_fmeans[precision][k] = r.fmeans[k]; }
  m->calib_slots[precision] = r.slots;
}
// (re-)quantises the LOADED networks of `precision` from a record (weights and corrections)
static int apply_record(fp_model *m, int precision, const CalibRecord &r) {
  Net *loaded[2] = {m->refiner_p[precision], m->scorer_p[precision]};
  for (int k = 0; k < 2; k++)
    if (loaded[k] && (net_apply_q8(loaded[k], r.amax[k].data(), r.bias_fix[k].data(), r.tok_fix[k].data(), true, r.slots ? r.fmeans[k].data() : nullptr, r.slots) ||
                      net_q8_set_out_fix(loaded[k], r.out_fix[k].data()))) return 1;
  return 0;
}

static int calibrate_impl(fp_model *m, const std::vector<CalibFrame> &frames, int precision) {
  FP_CHECK(m != nullptr, "[FoundationPose] null model");
  FP_CHECK(precision == PREC_FP8 || precision == PREC_INT8, "[FoundationPose] fp_calibrate: precision must be FP_PREC_FP8 or FP_PREC_INT8");
  FP_CHECK(!m->refiner_path.empty() && !m->scorer_path.empty(), "[FoundationPose] fp_calibrate needs both networks");
  FP_CHECK(!frames.empty(), "[FoundationPose] fp_calibrate_finish: no calibration frame was added");
  FP_CHECK(m->refiner_p[precision] && m->scorer_p[precision] && m->refiner_p[PREC_F16] && m->scorer_p[PREC_F16], "[FoundationPose] fp_calibrate: networks not loaded");
  const int prev = m->prec;
  if (select_precision(m, PREC_F16)) return 1;
  float pose[16];
  constexpr size_t NS = (size_t)15 * 512;
  const double inv_frames = 1.0 / (double)frames.size();
  // means over the hypotheses of what the networks hand to the pose update / the cross-hypothesis head (still in the model's buffers
  // after a Register): refiner trans | rot, scorer pooled features; accumulated over the frames of a pass
  auto add_output_means = [&](std::vector<double> (&acc)[2]) -> int {
    const int N = m->n_hyp();
    std::vector<float> t((size_t)N * 3), r((size_t)N * 3), f((size_t)N * 512);
    FP_HIP_OK(hipMemcpyAsync(t.data(), m->trans_dev, t.size() * 4, hipMemcpyDeviceToHost, m->stream));
    FP_HIP_OK(hipMemcpyAsync(r.data(), m->rot_dev, r.size() * 4, hipMemcpyDeviceToHost, m->stream));
    FP_HIP_OK(hipMemcpyAsync(f.data(), m->feat_dev, f.size() * 4, hipMemcpyDeviceToHost, m->stream));
    FP_HIP_OK(hipStreamSynchronize(m->stream));
    for (int c = 0; c < 3; c++) {
      double a = 0, b = 0;
      for (int i = 0; i < N; i++) { a += t[(size_t)i * 3 + c]; b += r[(size_t)i * 3 + c]; }
      acc[0][c] += a / N * inv_frames; acc[0][3 + c] += b / N * inv_frames;
    }
    for (int c = 0; c < 512; c++) {
      double a = 0;
      for (int i = 0; i < N; i++) a += f[(size_t)i * 512 + c];
      acc[1][c] += a / N * inv_frames;
    }
    return 0;
  };
  // one pass over the frames: statistics of both networks (mode / only_act: net_calib_begin), optionally the output means
  auto pass = [&](int mode, int only_act, std::vector<float> *amax, std::vector<float> (&mean)[2], std::vector<float> *out_mean) -> int {
    Net *nets[2] = {m->refiner, m->scorer};
    for (Net *n : nets) net_calib_begin(n, m->stream, mode, only_act);
    std::vector<double> acc[2] = {std::vector<double>(8, 0.0), std::vector<double>(512, 0.0)};
    int rc = 0;
    for (const CalibFrame &f : frames) {
      m->calibrating = true;
      rc = fp_register_ex(m, f.rgb.data(), f.depth.data(), f.mask.data(), FP_HOST, f.H, f.W, f.target.c_str(), 1, pose);
      m->calibrating = false;
      if (!rc && out_mean) rc = add_output_means(acc);
      if (rc) break;
    }
    for (int k = 0; k < 2; k++) {
      mean[k].assign(NS, 0.f);
      if (amax) amax[k].assign(NS, 0.f);
      rc |= net_calib_end(nets[k], m->stream, amax ? amax[k].data() : nullptr, mean[k].data());
      if (out_mean) out_mean[k].assign(acc[k].begin(), acc[k].end());
    }
    return rc;
  };
  std::vector<float> f16_mean[2], f16_out_mean[2];
  CalibRecord rec;
  {   // 1. f16 statistics, frame by frame: |max| over all frames, the mean over all frames, and every frame's own channel means (slots)
    rec.slots = (int)std::min<size_t>(frames.size(), kCalibSlots);
    std::vector<double> acc_out[2] = {std::vector<double>(8, 0.0), std::vector<double>(512, 0.0)};
    std::vector<double> mean_acc[2] = {std::vector<double>(NS, 0.0), std::vector<double>(NS, 0.0)};
    std::vector<int> per_slot(rec.slots, 0);
    for (int k = 0; k < 2; k++) { rec.amax[k].assign(NS, 0.f); rec.fmeans[k].assign((size_t)rec.slots * NS, 0.f); }
    Net *nets[2] = {m->refiner, m->scorer};
    for (size_t f = 0; f < frames.size(); f++) {
      for (Net *n : nets) net_calib_begin(n, m->stream, 1, -1);
      m->calibrating = true;
      int rc = fp_register_ex(m, frames[f].rgb.data(), frames[f].depth.data(), frames[f].mask.data(), FP_HOST, frames[f].H, frames[f].W, frames[f].target.c_str(), 1, pose);
      m->calibrating = false;
      if (!rc) rc = add_output_means(acc_out);
      const int slot = (int)(f % (size_t)rec.slots);
      per_slot[slot]++;
      for (int k = 0; k < 2; k++) {
        std::vector<float> am(NS), mn(NS);
        rc |= net_calib_end(nets[k], m->stream, am.data(), mn.data());
        for (size_t i = 0; i < NS; i++) {
          rec.amax[k][i] = std::max(rec.amax[k][i], am[i]);
          mean_acc[k][i] += mn[i];
          rec.fmeans[k][(size_t)slot * NS + i] += mn[i];
        }
      }
      if (rc) return 1;
    }
    for (int k = 0; k < 2; k++) {
      f16_mean[k].resize(NS);
      for (size_t i = 0; i < NS; i++) f16_mean[k][i] = (float)(mean_acc[k][i] * inv_frames);
      for (int sl = 0; sl < rec.slots; sl++)
        for (size_t i = 0; i < NS; i++) rec.fmeans[k][(size_t)sl * NS + i] /= (float)per_slot[sl];
      f16_out_mean[k].assign(acc_out[k].begin(), acc_out[k].end());
    }
  }
  for (int k = 0; k < 2; k++) { rec.bias_fix[k].assign((size_t)13 * 512, 0.f); rec.tok_fix[k].assign(512, 0.f); rec.out_fix[k].assign(k == 0 ? 8 : 512, 0.f); }
  // quantise (or re-quantise) this precision's networks without corrections
  if (apply_record(m, precision, rec)) return 1;
  invalidate_graphs(m);
  // (not select_precision: it would put the PREVIOUS record's output correction back on a network that is being re-calibrated)
  if (!m->ws_p[precision]) m->ws_p[precision] = nn_scratch_create(precision);
  m->prec = precision; m->refiner = m->refiner_p[precision]; m->scorer = m->scorer_p[precision]; m->ws = m->ws_p[precision];
  Net *qn[2] = {m->refiner, m->scorer};
  std::vector<float> mq[2];
  auto target = [&](int k, int a, int c) {   // f16 mean the output channel c of the layer writing activation a should have
    const std::vector<float> &t = f16_mean[k];
    return a == 5 ? 0.5f * (t[a * 512 + c] + t[a * 512 + c + 128]) : t[a * 512 + c];
  };
  const int n_sweeps = g_calib_sweeps;   // 2 (tools/q8_check.py, round 4: 0 / 1 / 2 / 3 sweeps -> 88 / 93 / 95-100 / 98 % of the refined poses within 1 mm / 1 deg of the f16 path)
  for (int sweep = 0; sweep < n_sweeps; sweep++)
    for (int layer = 0; layer < 13; layer++) {
      const int a = layer + 2, C = net_q8_bias_channels(layer);
      if (!net_q8_layer_on(qn[0], layer)) continue;   // [r6] a 2-byte layer of a partly 8-bit trunk has nothing to correct
      if (pass(2, a, nullptr, mq, nullptr)) return 1;
      for (int k = 0; k < 2; k++) {
        std::vector<float> &fix = rec.bias_fix[k];
        for (int c = 0; c < C; c++) {
          const float got = a == 5 ? 0.5f * (mq[k][a * 512 + c] + mq[k][a * 512 + c + 128]) : mq[k][a * 512 + c];
          fix[(size_t)layer * 512 + c] += target(k, a, c) - got;
        }
        if (net_apply_q8(qn[k], rec.amax[k].data(), fix.data(), rec.tok_fix[k].data(), false)) return 1;
      }
    }
  if (g_calib_tok && pass(2, 14, nullptr, mq, nullptr)) return 1;
  for (int k = 0; g_calib_tok && k < 2; k++) {
    for (int c = 0; c < 512; c++) rec.tok_fix[k][c] = f16_mean[k][14 * 512 + c] - mq[k][14 * 512 + c];
    if (net_apply_q8(qn[k], rec.amax[k].data(), rec.bias_fix[k].data(), rec.tok_fix[k].data(), false)) return 1;
  }
  // 5. what is left at the OUTPUTS (the heads are non-linear in the token mean): the mean refiner outputs / pooled score feature of
  //    the 8-bit model over the frames are moved onto the f16 model's through the output layers' biases
  if (g_calib_out) {
    std::vector<float> om[2];
    if (pass(2, 14, nullptr, mq, om)) return 1;
    for (int k = 0; k < 2; k++) {
      for (size_t c = 0; c < om[k].size(); c++) rec.out_fix[k][c] = f16_out_mean[k][c] - om[k][c];
      if (net_q8_set_out_fix(qn[k], rec.out_fix[k].data())) return 1;
    }
  }
  for (int k = 0; k < 2; k++)
    for (const std::vector<float> *v : {&rec.amax[k], &rec.bias_fix[k], &rec.tok_fix[k], &rec.out_fix[k], &rec.fmeans[k]})
      for (float x : *v) FP_CHECK(std::isfinite(x), "[FoundationPose] fp_calibrate: the solved record is not finite (broken weights or frames)");
  commit_record(m, precision, rec);   // only now: every step succeeded
  invalidate_graphs(m);
  return select_precision(m, prev);
}

// Locking [r5]: exclusive (against every other call of the process) only while networks are LOADED -- the Registers of the
// calibration and the re-quantisation uploads into existing buffers run under the shared lifetime lock like any Register, so other
// models keep serving while one calibrates.
static int calibrate_locked(fp_model *m, const std::vector<CalibFrame> &frames, int precision) {
  FP_CHECK(m != nullptr, "[FoundationPose] null model");
  FP_CHECK(precision == PREC_FP8 || precision == PREC_INT8, "[FoundationPose] fp_calibrate: precision must be FP_PREC_FP8 or FP_PREC_INT8");
  FP_CHECK(!m->refiner_path.empty() && !m->scorer_path.empty(), "[FoundationPose] fp_calibrate needs both networks");
  const int prev = m->prec;
  if (!m->refiner_p[precision] || !m->scorer_p[precision] || !m->refiner_p[PREC_F16] || !m->scorer_p[PREC_F16]) {
    PreparedNets q8, f16;   // host phase outside the lock
    if (q8.prepare(m->refiner_path, m->refiner_p[precision] != nullptr, m->scorer_path, m->scorer_p[precision] != nullptr, precision) ||
        f16.prepare(m->refiner_path, m->refiner_p[PREC_F16] != nullptr, m->scorer_path, m->scorer_p[PREC_F16] != nullptr, PREC_F16)) return 1;
    LifeExclusive life;   // device phase: one allocation + upload per network
    DeviceScope on_device(m->device);
    if (q8.adopt(m) || f16.adopt(m)) return 1;
    FP_HIP_OK(hipStreamSynchronize(m->stream));
    int rc = select_precision(m, precision) || select_precision(m, PREC_F16);
    rc = select_precision(m, rc ? PREC_F16 : prev) || rc;
    if (rc) return 1;
  }
  SerialGuard serial(m->device);
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  const CalibRecord before = record_of(m, precision);
  const int rc = calibrate_impl(m, frames, precision);
  if (rc) {   // leave the model usable and the precision's networks consistent with its (previous) record
    const std::string why = g_last_error;
    m->calibrating = false;
    for (Net *n : {m->refiner_p[precision], m->scorer_p[precision], m->refiner_p[PREC_F16], m->scorer_p[PREC_F16]}) if (n) net_calib_abort(n);
    (void)hipStreamSynchronize(m->stream);
    invalidate_graphs(m);
    bool restored = before.valid() && apply_record(m, precision, before) == 0;
    if (!restored) {   // uncalibrated: fp_set_precision refuses the precision until a calibration succeeds
      for (int k = 0; k < 2; k++) { m->calib_amax_q[precision][k].clear(); m->calib_bias_fix[precision][k].clear(); m->calib_tok_fix[precision][k].clear(); m->calib_out_fix[precision][k].clear(); m->calib_fmeans[precision][k].clear(); }
      m->calib_slots[precision] = 0;
      for (Net *n : {m->refiner_p[precision], m->scorer_p[precision]}) if (n) net_q8_unready(n);
    }
    const bool q8_prev = prev == PREC_FP8 || prev == PREC_INT8;
    (void)select_precision(m, (q8_prev && !m->calibrated(prev)) ? PREC_F16 : prev);
    set_error(why);
  }
  return rc;
}

static int copy_frame_in(fp_model *m, CalibFrame &f, const void *rgb, const void *depth, const void *mask, int memspace, int H, int W) {
  const size_t px = (size_t)H * W;
  f.rgb.resize(px * 3); f.depth.resize(px); f.mask.resize(px);
  if (memspace == FP_HOST) {
    std::memcpy(f.rgb.data(), rgb, px * 3); std::memcpy(f.depth.data(), depth, px * 4); std::memcpy(f.mask.data(), mask, px);
    return 0;
  }
  FP_HIP_OK(hipMemcpyAsync(f.rgb.data(), rgb, px * 3, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipMemcpyAsync(f.depth.data(), depth, px * 4, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipMemcpyAsync(f.mask.data(), mask, px, hipMemcpyDeviceToHost, m->stream));
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  return 0;
}

int fp_calibrate_begin(fp_model *m, int precision) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m != nullptr, "[FoundationPose] null model");
  FP_CHECK(precision == PREC_FP8 || precision == PREC_INT8, "[FoundationPose] fp_calibrate_begin: precision must be FP_PREC_FP8 or FP_PREC_INT8");
  FP_CHECK(!m->refiner_path.empty() && !m->scorer_path.empty(), "[FoundationPose] fp_calibrate needs both networks");
  m->calib_frames.clear();
  m->calib_session_prec = precision;
  return 0;
} FP_CATCH_INT
int fp_calibrate_add_frame(fp_model *m, const void *rgb, const void *depth, const void *mask, int memspace, int H, int W, const char *target_name) try {
  SerialGuard serial(m ? m->device : -1);
  FP_CHECK(m != nullptr, "[FoundationPose] null model");
  FP_CHECK(m->calib_session_prec >= 0, "[FoundationPose] fp_calibrate_add_frame without fp_calibrate_begin");
  FP_CHECK(rgb && depth && mask, "[FoundationPose] fp_calibrate_add_frame: a calibration frame needs rgb, depth and mask");
  FP_CHECK(m->calib_frames.size() < 1024, "[FoundationPose] fp_calibrate_add_frame: too many frames (1024)");
  Target *t = nullptr;
  if (check_frame_args(m, H, W, target_name ? target_name : "", &t)) return 1;
  CalibFrame f;
  f.H = H; f.W = W; f.target = t->name;
  if (copy_frame_in(m, f, rgb, depth, mask, memspace, H, W)) return 1;
  m->calib_frames.push_back(std::move(f));
  return 0;
} FP_CATCH_INT
int fp_calibrate_finish(fp_model *m) try {
  FP_CHECK(m != nullptr, "[FoundationPose] null model");
  FP_CHECK(m->calib_session_prec >= 0, "[FoundationPose] fp_calibrate_finish without fp_calibrate_begin");
  std::vector<CalibFrame> frames;
  frames.swap(m->calib_frames);
  const int precision = m->calib_session_prec;
  m->calib_session_prec = -1;
  return calibrate_locked(m, frames, precision);
} FP_CATCH_INT
int fp_calibrate_abort(fp_model *m) try {
  FP_CHECK(m != nullptr, "[FoundationPose] null model");
  m->calib_frames.clear();
  m->calib_frames.shrink_to_fit();
  m->calib_session_prec = -1;
  return 0;
} FP_CATCH_INT
int fp_calibrate_frames(const fp_model *m) { return m && m->calib_session_prec >= 0 ? (int)m->calib_frames.size() : -1; }

int fp_calibrate(fp_model *m, const void *rgb, const void *depth, const void *mask, int memspace, int H, int W, const char *target_name,
                 int precision) try {
  FP_CHECK(m != nullptr, "[FoundationPose] null model");
  FP_CHECK(m->calib_session_prec < 0, "[FoundationPose] fp_calibrate while a fp_calibrate_begin session is open (finish or abort it first)");
  if (fp_calibrate_begin(m, precision)) return 1;
  if (fp_calibrate_add_frame(m, rgb, depth, mask, memspace, H, W, target_name)) { (void)fp_calibrate_abort(m); return 1; }
  return fp_calibrate_finish(m);
} FP_CATCH_INT
int fp_calibrate_fp8(fp_model *m, const void *rgb, const void *depth, const void *mask, int memspace, int H, int W,
                     const char *target_name) try {
  return fp_calibrate(m, rgb, depth, mask, memspace, H, W, target_name, PREC_FP8);
} FP_CATCH_INT

// ---- the calibration record, version 2: [magic "FPQ8", 2, precision, slots] + amax [2][15][512] + bias_fix [2][13][512] + tok_fix [2][512]
// + out_fix [8 | 512] + frame means [2][16 slots][15][512] (f32; unused slots zero).  Version 1 (round 4: no frame means, weights rounded
// to nearest) is still accepted.
static constexpr uint32_t kCalibMagic = 0x38515046u;
static constexpr size_t kCalibFloats = (size_t)2 * 15 * 512 + (size_t)2 * 13 * 512 + (size_t)2 * 512 + 8 + 512;
static constexpr size_t kCalibMeanFloats = (size_t)2 * kCalibSlots * 15 * 512;
static constexpr size_t kCalibSizeV1 = 16 + kCalibFloats * sizeof(float);
size_t fp_calibration_size(void) { return 16 + (kCalibFloats + kCalibMeanFloats) * sizeof(float); }
int fp_get_calibration_blob(const fp_model *m, int precision, void *out, size_t capacity) try {
  FP_CHECK(m && out && (precision == PREC_FP8 || precision == PREC_INT8), "[FoundationPose] fp_get_calibration_blob: invalid arguments");
  FP_CHECK(m->calibrated(precision) && !m->calib_bias_fix[precision][0].empty(), "[FoundationPose] no calibration available for this precision");
  FP_CHECK(capacity >= fp_calibration_size(), "[FoundationPose] fp_get_calibration_blob: buffer too small (fp_calibration_size)");
  const int slots = m->calib_slots[precision];
  uint32_t hdr[4] = {kCalibMagic, 2u, (uint32_t)precision, (uint32_t)slots};
  unsigned char *p = (unsigned char *)out;
  std::memcpy(p, hdr, 16); p += 16;
  for (int k = 0; k < 2; k++) { std::memcpy(p, m->calib_amax_q[precision][k].data(), 15 * 512 * 4); p += 15 * 512 * 4; }
  for (int k = 0; k < 2; k++) { std::memcpy(p, m->calib_bias_fix[precision][k].data(), 13 * 512 * 4); p += 13 * 512 * 4; }
  for (int k = 0; k < 2; k++) { std::memcpy(p, m->calib_tok_fix[precision][k].data(), 512 * 4); p += 512 * 4; }
  for (int k = 0; k < 2; k++) { const size_t n = k == 0 ? 8 : 512; std::memcpy(p, m->calib_out_fix[precision][k].data(), n * 4); p += n * 4; }
  for (int k = 0; k < 2; k++) {
    const size_t n = (size_t)kCalibSlots * 15 * 512, have = (size_t)slots * 15 * 512;
    std::memset(p, 0, n * 4);
    if (have) std::memcpy(p, m->calib_fmeans[precision][k].data(), have * 4);
    p += n * 4;
  }
  return 0;
} FP_CATCH_INT
int fp_set_calibration_blob(fp_model *m, const void *blob, size_t bytes) try {
  LifeExclusive life;
  FP_CHECK(m && blob && (bytes == fp_calibration_size() || bytes == kCalibSizeV1), "[FoundationPose] fp_set_calibration_blob: invalid arguments / size");
  DeviceScope on_device(m->device);
  uint32_t hdr[4];
  const unsigned char *p = (const unsigned char *)blob;
  std::memcpy(hdr, p, 16); p += 16;
  const bool v2 = hdr[1] == 2u && bytes == fp_calibration_size(), v1 = hdr[1] == 1u && bytes == kCalibSizeV1;
  FP_CHECK(hdr[0] == kCalibMagic && (v1 || v2) && (hdr[2] == (uint32_t)PREC_FP8 || hdr[2] == (uint32_t)PREC_INT8) && (v1 ? hdr[3] == 0u : hdr[3] <= (uint32_t)kCalibSlots),
           "[FoundationPose] not a calibration record of this library version");
  const int precision = (int)hdr[2], slots = v2 ? (int)hdr[3] : 0;
  const size_t n_floats = kCalibFloats + (v2 ? kCalibMeanFloats : 0);
  {   // every float of the record must be finite (and the |max| values non-negative) BEFORE anything of the model is touched
    std::vector<float> tmp(n_floats);   // (the caller's buffer need not be aligned)
    std::memcpy(tmp.data(), p, n_floats * sizeof(float));
    const float *f = tmp.data();
    for (size_t i = 0; i < n_floats; i++) FP_CHECK(std::isfinite(f[i]), "[FoundationPose] calibration record holds non-finite values");
    for (size_t i = 0; i < (size_t)2 * 15 * 512; i++) FP_CHECK(f[i] >= 0.f, "[FoundationPose] calibration record holds a negative |max|");
  }
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  CalibRecord rec;
  auto take = [&](std::vector<float> &dst, size_t n) { dst.resize(n); std::memcpy(dst.data(), p, n * 4); p += n * 4; };
  for (int k = 0; k < 2; k++) take(rec.amax[k], 15 * 512);
  for (int k = 0; k < 2; k++) take(rec.bias_fix[k], 13 * 512);
  for (int k = 0; k < 2; k++) take(rec.tok_fix[k], 512);
  for (int k = 0; k < 2; k++) take(rec.out_fix[k], k == 0 ? 8 : 512);
  rec.slots = slots;
  for (int k = 0; v2 && k < 2; k++) { take(rec.fmeans[k], (size_t)kCalibSlots * 15 * 512); rec.fmeans[k].resize((size_t)slots * 15 * 512); }
  const CalibRecord before = record_of(m, precision);
  if (apply_record(m, precision, rec)) {   // (re-quantises the networks of the precision that are already loaded)
    // a failure behind the first network leaves it re-quantised to the NEW record while the model still reports the old one: put the
    // old one back (as calibrate_locked does), or drop the precision's record when that fails too
    const std::string why = fp_last_error();
    if (!(before.valid() && apply_record(m, precision, before) == 0)) {
      for (int k = 0; k < 2; k++) { m->calib_amax_q[precision][k].clear(); m->calib_bias_fix[precision][k].clear(); m->calib_tok_fix[precision][k].clear(); m->calib_out_fix[precision][k].clear(); m->calib_fmeans[precision][k].clear(); }
      m->calib_slots[precision] = 0;
      for (Net *n : {m->refiner_p[precision], m->scorer_p[precision]}) if (n) net_q8_unready(n);
    }
    invalidate_graphs(m);
    set_error(why);
    return 1;
  }
  commit_record(m, precision, rec);
  invalidate_graphs(m);
  return 0;
} FP_CATCH_INT
// legacy per-tensor view (round 2/3 API): |max| over the channels of each trunk activation, [refiner 16 | scorer 16]
int fp_get_calibration(const fp_model *m, float amax_out[32]) try {
  FP_CHECK(m && amax_out && (m->calibrated(PREC_FP8) || m->calibrated(PREC_INT8)), "[FoundationPose] no calibration available");
  const int pr = m->calibrated(PREC_FP8) ? PREC_FP8 : PREC_INT8;   // (the round-2 API knew FP8 only)
  for (int k = 0; k < 2; k++)
    for (int a = 0; a < 16; a++) {
      float v = 0.f;
      if (a < 15) for (int c = 0; c < 512; c++) v = std::max(v, m->calib_amax_q[pr][k][a * 512 + c]);
      amax_out[k * 16 + a] = v;
    }
  return 0;
} FP_CATCH_INT
// legacy: per-tensor |max| only -> every channel of a tensor gets the tensor's scale, no bias / token correction (FP8 networks)
int fp_set_calibration(fp_model *m, const float amax[32]) try {
  LifeExclusive life;
  FP_CHECK(m && amax, "[FoundationPose] fp_set_calibration: invalid arguments");
  DeviceScope on_device(m->device);
  FP_HIP_OK(hipStreamSynchronize(m->stream));
  for (int k = 0; k < 2; k++) {
    std::vector<float> rec((size_t)15 * 512, 0.f);
    for (int a = 0; a < 15; a++)
      for (int c = 0; c < 512; c++) rec[a * 512 + c] = amax[k * 16 + a];
    for (int pr : {PREC_FP8, PREC_INT8}) {
      m->calib_amax_q[pr][k] = rec;
      m->calib_bias_fix[pr][k].assign((size_t)13 * 512, 0.f); m->calib_tok_fix[pr][k].assign(512, 0.f); m->calib_out_fix[pr][k].assign(k == 0 ? 8 : 512, 0.f);
      m->calib_fmeans[pr][k].clear(); m->calib_slots[pr] = 0;   // (per-tensor records carry no frame means: weights rounded to nearest)
    }
  }
  for (int pr : {PREC_FP8, PREC_INT8}) {
    Net *loaded[2] = {m->refiner_p[pr], m->scorer_p[pr]};
    for (int k = 0; k < 2; k++)
      if (loaded[k] && (net_apply_q8(loaded[k], m->calib_amax_q[pr][k].data(), m->calib_bias_fix[pr][k].data(), m->calib_tok_fix[pr][k].data(), true) ||
                        net_q8_set_out_fix(loaded[k], m->calib_out_fix[pr][k].data()))) return 1;
  }
  invalidate_graphs(m);
  return 0;
} FP_CATCH_INT

// ------------------------------------------------------------------------------------------------
// A network on its own, with the blob interface the reference's orchestrator drives through deploy_core's BaseInferCore
// (GetBuffer / GetTensor / SetBufferLocation / RawPtr / SetShape / SyncInfer, D6F/src/foundationpose.cpp:126-139,331-354,
// 410-436); include/infer_core_amd.hpp puts those C++ names on top of these calls.
// ------------------------------------------------------------------------------------------------
struct fp_net {
  int device = 0;
  hipStream_t stream = nullptr;
  Net *net = nullptr;
  NNScratch *ws = nullptr;
  bool scorer = false;
  int max_batch = 0;
  float *in_dev[2] = {nullptr, nullptr};   // render_input, transf_input  [max_batch,160,160,6] f32
  float *in_host[2] = {nullptr, nullptr};  // pinned, allocated on first use
  float *out_dev[2] = {nullptr, nullptr};  // trans, rot [max_batch,3]  |  scores [max_batch] (+ unused)
  float *out_host[2] = {nullptr, nullptr};
  float *feat_dev = nullptr;               // scorer: pooled features
  __half *nn_in = nullptr;                 // packed 2-byte network input
};
static size_t net_blob_elems(const fp_net *n, int idx, bool out) {
  if (!out) return (size_t)n->max_batch * FP_CROP_HW * FP_CROP_HW * 6;
  return (size_t)n->max_batch * (n->scorer ? 1 : 3) * (idx == 0 || !n->scorer ? 1 : 0);
}
static int net_blob_index(const fp_net *n, const char *name, bool *out) {
  const std::string s(name ? name : "");
  if (s == "render_input") { *out = false; return 0; }
  if (s == "transf_input") { *out = false; return 1; }
  if (!n->scorer && s == "trans") { *out = true; return 0; }
  if (!n->scorer && s == "rot") { *out = true; return 1; }
  if (n->scorer && s == "scores") { *out = true; return 0; }
  return -1;
}

static void destroy_net_impl(fp_net *n);
fp_net *fp_net_create(const char *weights_path, int is_scorer, int max_batch) try {
  if (!weights_path || max_batch <= 0) { set_error("[FoundationPose] fp_net_create: invalid arguments"); return nullptr; }
  std::string err;
  std::unique_ptr<Net, void (*)(Net *)> prepared(net_prepare(weights_path, is_scorer != 0, PREC_F16, &err), net_free);   // host phase, outside the lock
  if (!prepared) { set_error("[FoundationPose] Failed to load network weights: " + err); return nullptr; }
  LifeExclusive life;   // (see SerialGuard)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("[FoundationPose] no HIP device available (this library has no CPU path)"); return nullptr; }
  std::unique_ptr<fp_net> n(new fp_net());
  if (hipGetDevice(&n->device) != hipSuccess) { set_error("[FoundationPose] hipGetDevice failed"); return nullptr; }
  n->scorer = is_scorer != 0;
  n->max_batch = max_batch;
  if (net_commit(prepared.get(), &err)) { set_error("[FoundationPose] Failed to load network weights: " + err); return nullptr; }
  n->net = prepared.release();
  n->ws = nn_scratch_create(PREC_F16);
  const size_t in_elems = (size_t)max_batch * FP_CROP_HW * FP_CROP_HW * 6;
  bool ok = stream_acquire(&n->stream) == hipSuccess;
  for (int i = 0; ok && i < 2; i++) ok = !dev_alloc(&n->in_dev[i], in_elems) && !dev_alloc(&n->out_dev[i], (size_t)max_batch * 3);
  ok = ok && !dev_alloc(&n->feat_dev, (size_t)max_batch * 512) && !dev_alloc(&n->nn_in, (size_t)2 * max_batch * FP_NN_IN_IMG_HALFS);
  ok = ok && hipMemsetAsync(n->nn_in, 0, (size_t)2 * max_batch * FP_NN_IN_IMG_HALFS * sizeof(__half), n->stream) == hipSuccess;
  if (!ok) { set_error("[FoundationPose] fp_net_create: device allocation failed"); destroy_net_impl(n.release()); return nullptr; }
  return n.release();
} FP_CATCH_PTR

static void destroy_net_impl(fp_net *n) {
  if (!n) return;
  DeviceScope on_device(n->device);
  if (n->stream) (void)hipStreamSynchronize(n->stream);
  for (int i = 0; i < 2; i++) {
    dev_free(n->in_dev[i]); dev_free(n->out_dev[i]);
    if (n->in_host[i]) (void)hipHostFree(n->in_host[i]);
    if (n->out_host[i]) (void)hipHostFree(n->out_host[i]);
  }
  dev_free(n->feat_dev); dev_free(n->nn_in);
  if (n->net) net_free(n->net);
  if (n->ws) nn_scratch_free(n->ws);
  if (n->stream) { (void)hipStreamSynchronize(n->stream); stream_release(n->stream); }
  delete n;
}
void fp_net_destroy(fp_net *n) {
  LifeExclusive life;
  destroy_net_impl(n);
}

// raw pointer of a blob in the given memory space (BlobsTensor::GetTensor(name)->RawPtr()); NULL + fp_last_error for an
// unknown name (the reference's GetTensor throws)
void *fp_net_blob(fp_net *n, const char *name, int memspace) try {
  bool out = false;
  const int idx = n ? net_blob_index(n, name, &out) : -1;
  if (idx < 0) { set_error(std::string("[FoundationPose] no blob named '") + (name ? name : "") + "'"); return nullptr; }
  if (memspace == FP_DEVICE) return out ? (void *)n->out_dev[idx] : (void *)n->in_dev[idx];
  float **h = out ? &n->out_host[idx] : &n->in_host[idx];
  if (!*h) {
    const size_t elems = out ? (size_t)n->max_batch * 3 : (size_t)n->max_batch * FP_CROP_HW * FP_CROP_HW * 6;
    if (hipHostMalloc((void **)h, elems * sizeof(float), hipHostMallocDefault) != hipSuccess) { set_error("[FoundationPose] pinned allocation failed"); return nullptr; }
  }
  return *h;
} FP_CATCH_PTR
int fp_net_max_batch(const fp_net *n) { return n ? n->max_batch : 0; }

// SyncInfer: inputs are taken from the blobs' FP_HOST or FP_DEVICE copies (render_loc / transf_loc), outputs are left in
// the device blobs and, with out_loc == FP_HOST, copied to the host blobs as well; returns when they are complete.
int fp_net_infer(fp_net *n, int batch, int render_loc, int transf_loc, int out_loc) try {
  SerialGuard serial(n ? n->device : -1);
  FP_CHECK(n && batch > 0 && batch <= n->max_batch, "[FoundationPose] fp_net_infer: batch out of range");
  const size_t px = (size_t)batch * FP_CROP_HW * FP_CROP_HW;
  const int locs[2] = {render_loc, transf_loc};
  for (int i = 0; i < 2; i++)
    if (locs[i] == FP_HOST) {
      FP_CHECK(n->in_host[i] != nullptr, "[FoundationPose] fp_net_infer: host input blob was never requested");
      FP_HIP_OK(hipMemcpyAsync(n->in_dev[i], n->in_host[i], px * 24, hipMemcpyHostToDevice, n->stream));
    }
  launch_pack_f32x6(n->stream, n->in_dev[0], n->nn_in, px, OUT_F16X8);
  launch_pack_f32x6(n->stream, n->in_dev[1], n->nn_in + (size_t)batch * FP_NN_IN_IMG_HALFS, px, OUT_F16X8);
  if (n->scorer) {
    if (scorer_features(n->stream, nullptr, n->net, n->ws, n->nn_in, batch, n->feat_dev)) return 1;
    if (scorer_head(n->stream, nullptr, n->net, n->ws, n->feat_dev, batch, n->out_dev[0])) return 1;
  } else {
    if (refiner_forward(n->stream, nullptr, n->net, n->ws, n->nn_in, batch, n->out_dev[0], n->out_dev[1])) return 1;
  }
  if (out_loc == FP_HOST) {
    const int nout = n->scorer ? 1 : 2;
    for (int i = 0; i < nout; i++) {
      if (!fp_net_blob(n, n->scorer ? "scores" : (i == 0 ? "trans" : "rot"), FP_HOST)) return 1;
      FP_HIP_OK(hipMemcpyAsync(n->out_host[i], n->out_dev[i], (size_t)batch * (n->scorer ? 1 : 3) * sizeof(float), hipMemcpyDeviceToHost, n->stream));
    }
  }
  FP_HIP_OK(hipStreamSynchronize(n->stream));
  return 0;
} FP_CATCH_INT

int fp_profile_enable(fp_model *m, int on) try {
  FP_CHECK(m, "null model");
  SerialGuard serial(m->device);
  m->prof.on = on != 0;
  return 0;
} FP_CATCH_INT
int fp_profile_reset(fp_model *m) try {
  FP_CHECK(m, "null model");
  SerialGuard serial(m->device);
  (void)hipStreamSynchronize(m->stream);
  m->prof.reset();
  return 0;
} FP_CATCH_INT
int fp_profile_report(fp_model *m, char *buf, int buf_len) try {
  FP_CHECK(m && buf && buf_len > 0, "fp_profile_report: invalid arguments");
  SerialGuard serial(m->device);
  (void)hipStreamSynchronize(m->stream);
  m->prof.collect();
  std::string out;
  char line[256];
  for (auto &e : m->prof.entries) {
    std::snprintf(line, sizeof(line), "%s %ld %.6f %.6e %.6e\n", e.name.c_str(), e.calls, e.ms, e.flops, e.bytes);
    out += line;
  }
  std::snprintf(buf, (size_t)buf_len, "%s", out.c_str());
  return 0;
} FP_CATCH_INT

}  // extern "C"
