// Internal declarations shared by the HIP translation units of libfoundationpose_amd.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#define FP_CROP_HW 160
#define FP_NN_IN_BORDER 2  // zero border (in space-to-depth pixels) around each network-input image
// halfs per network-input image: [84,84,32]
#define FP_NN_IN_IMG_HALFS ((size_t)(FP_CROP_HW / 2 + 2 * FP_NN_IN_BORDER) * (FP_CROP_HW / 2 + 2 * FP_NN_IN_BORDER) * 32)
#define FP_MIN_DEPTH 0.001f  // FoundationPoseRenderer min_depth (foundationpose_render.hpp:26)
#define FP_MAX_DEPTH 4.0f    // FoundationPoseRenderer max_depth (foundationpose_render.hpp:27)

namespace fp {

// ---------------------------------------------------------------------------------------------
// error plumbing: C ABI returns 0 / non-zero + fp_last_error()
// ---------------------------------------------------------------------------------------------
void set_error(const std::string &msg);
// fp_image_io.cpp: 8-bit PNG (grey / RGB / palette / alpha variants) -> RGB u8
bool load_png_rgb(const std::string &path, std::vector<uint8_t> &rgb, int &H, int &W);
// any texture container the library decodes (PNG incl. 16-bit / sub-byte / interlaced, baseline + progressive JPEG, BMP, PNM, TGA) -> RGB u8 like cv::imread + BGR2RGB
bool load_texture_rgb(const std::string &path, std::vector<uint8_t> &rgb, int &H, int &W, std::string *why);
// Synchronous copies / fills WITHOUT the legacy (null) stream: hipMemcpy / hipMemset fail with hipErrorStreamCaptureImplicit (906)
// while ANY thread of the process captures a hipGraph (another model replaying its warm-up), so the library never touches the
// legacy stream -- these go through a per-thread non-blocking utility stream and wait for it.
hipError_t memcpy_sync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t memset_sync(void *dst, int value, size_t bytes);
// Per-DEVICE once for launch sites that opt a kernel into > 64 KB of dynamic LDS: hipFuncSetAttribute acts on the current device's
// copy of the function, so a process that drives several GPUs (one model per device) has to repeat it on each.  BLOCKING [r5]: a second
// thread on the same device (another model: the documented concurrency model) waits until the first has applied the attribute --
// a flag that is set before the attribute call would let it launch a > 64 KB kernel without the opt-in (std::call_once per device).
struct PerDeviceOnce {
  std::once_flag once[64];
  template <class F>
  void run(F &&f) {   // f runs exactly once per device, and every caller returns only after it has (devices >= 64: every time, harmless)
    int d = 0;
    (void)hipGetDevice(&d);
    if (d < 0 || d >= 64) { f(); return; }
    std::call_once(once[d], f);
  }
};
// bumped whenever a device buffer that kernels may have baked into a captured hipGraph is (re)allocated
extern std::atomic<unsigned long> g_alloc_epoch;
#define FP_HIP_OK(expr)                                                                              \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess) {                                                                          \
      fp::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e));                       \
      return 1;                                                                                      \
    }                                                                                                \
  } while (0)
#define FP_CHECK(cond, msg)                                                                          \
  do {                                                                                               \
    if (!(cond)) {                                                                                   \
      fp::set_error(msg);                                                                            \
      return 1;                                                                                      \
    }                                                                                                \
  } while (0)

// ---------------------------------------------------------------------------------------------
// profiling: optional HIP-event bracket around every launch, accumulated per kernel family
// ---------------------------------------------------------------------------------------------
struct ProfEntry {
  std::string name;
  long calls = 0;
  double flops = 0, bytes = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double ms = 0;
};
struct Profiler {
  bool on = false;
  std::vector<ProfEntry> entries;
  std::vector<hipEvent_t> pool;
  ProfEntry &get(const char *name);
  hipEvent_t ev();
  void begin(hipStream_t s, const char *name, double flops, double bytes);
  void end(hipStream_t s);
  void collect();
  void reset();
  ProfEntry *cur = nullptr;
  hipEvent_t cur_start = nullptr;
};
// roctx ranges (SURVEY.md section 5 "tracing / profiling": roctx + rocprofv3): every ProfScope -- one per kernel launch site, named
// "<layer>/<kernel>" -- and the stage scopes of fp_api.hip (RoctxRange) push / pop a range, so `rocprofv3 --marker-trace` sees
// labelled stages instead of a stream of 60 anonymous kernels.  The roctx library is bound at first use with dlopen
// (librocprofiler-sdk-roctx, else libroctx64); without it the calls are no-ops.  Host-side markers: inside a replayed hipGraph
// there is no host code per launch, so profile with graphs off (the first, eager call of a configuration) to see per-launch ranges.
void roctx_push(const char *name);
void roctx_pop();
struct RoctxRange {
  explicit RoctxRange(const char *name) { roctx_push(name); }
  ~RoctxRange() { roctx_pop(); }
  RoctxRange(const RoctxRange &) = delete;
  RoctxRange &operator=(const RoctxRange &) = delete;
};
struct ProfScope {
  Profiler *p;
  hipStream_t s;
  ProfScope(Profiler *p_, hipStream_t s_, const char *name, double flops = 0, double bytes = 0) : p(p_), s(s_) {
    roctx_push(name);
    if (p && p->on) p->begin(s, name, flops, bytes);
  }
  ~ProfScope() {
    if (p && p->on) p->end(s);
    roctx_pop();
  }
};

// ---------------------------------------------------------------------------------------------
// geometry kernels (fp_geometry.hip)
// ---------------------------------------------------------------------------------------------

// Per-hypothesis record produced by the pose-setup kernel; everything the reference computes on the host per
// pose (ComputeCropWindowTF, ConstructBBox2D, ProjectMatrixFromIntrinsics, foundationpose_render.cpp:25-186,590).
struct PoseRec {
  float M[16];     // Proj * GLcam * pose, column-major
  float pose[16];  // column-major
  float a00, a11, a30, a31;  // generate_pose_clip remap (foundationpose_render.cu:382-384)
  float m0, m2, m4, m5;      // inverse crop transform: src = (m0*x + m2, m4*y + m5)
  float tf[9];               // forward crop transform, row-major (debug / tests)
  float bbox[4];
};

struct DeviceMesh {
  int V = 0, F = 0, TH = 0, TW = 0;
  float diameter = 0;
  float center[3] = {0, 0, 0};
  float *verts = nullptr;    // [V,3] centred
  float *normals = nullptr;  // [V,3]
  float *uvs = nullptr;      // [V,2] (u, 1-v)
  int32_t *faces = nullptr;  // [F,3]
  uint8_t *tex = nullptr;    // [TH,TW,3]
};

enum OutMode { OUT_F32X6 = 0, OUT_F16X8 = 1, OUT_BF16X8 = 2 };  // F32X6: the reference's blob; *X8: the networks' s2d input tensor

void launch_pose_setup(hipStream_t s, const float *poses_dev, int N, const float *K9_host, int img_h, int img_w,
                       float crop_ratio, float diameter, PoseRec *recs);
#ifdef FP_TEST_HOOKS
extern float4 *g_vertex_dbg;  // race hunt: per-vertex intermediates (null = off)
#endif
// fmad: float model of the vertex stage / shader / interpolator / texture unit (fp_geometry.hip "float model")
void launch_vertex(hipStream_t s, const DeviceMesh &m, const PoseRec *recs, int N, float4 *clip, float4 *attr, bool fmad);
// the two above in ONE launch (recs is an output)
void launch_setup_vertex(hipStream_t s, const DeviceMesh &m, const float *poses_dev, int N, const float *K9_host, int img_h, int img_w,
                         float crop_ratio, float diameter, PoseRec *recs, float4 *clip, float4 *attr, bool fmad);
#ifdef __HIPCC__
// RefinePostProcess for hypothesis i (foundationpose.cpp:360-406): one function for pose_update_kernel and for the Track head
// kernel of fp_nn.hip that applies it in place (one launch less per frame); identical arithmetic in both.
__device__ __forceinline__ void pose_update_one(float *poses, const float *__restrict__ trans, const float *__restrict__ rot, int i, float diameter,
                                                const float *poses_in, float *extra_out) {
  // fp_geometry.hip is compiled with -ffp-contract=off (the oracle's float model), fp_nn.hip is not: pin it
  // here so that both translation units evaluate exactly the same operations
#pragma clang fp contract(off)
  const float NORM = 0.349065850398865f;
  float P[16];
  for (int k = 0; k < 16; k++) P[k] = poses_in[(size_t)i * 16 + k];
  float td[3], v[3];
  for (int k = 0; k < 3; k++) { td[k] = trans[i * 3 + k] * (diameter / 2); v[k] = tanhf(rot[i * 3 + k]) * NORM; }
  float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  float ang = sqrtf(n2);
  float ax[3] = {v[0], v[1], v[2]};
  if (n2 > 0.0f) { ax[0] /= ang; ax[1] /= ang; ax[2] /= ang; }
  float s = sinf(ang), c = cosf(ang);
  float sa[3] = {s * ax[0], s * ax[1], s * ax[2]}, ca[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
  float R[9], tmp;
  tmp = ca[0] * ax[1]; R[1] = tmp - sa[2]; R[3] = tmp + sa[2];
  tmp = ca[0] * ax[2]; R[2] = tmp + sa[1]; R[6] = tmp - sa[1];
  tmp = ca[1] * ax[2]; R[5] = tmp - sa[0]; R[7] = tmp + sa[0];
  R[0] = ca[0] * ax[0] + c; R[4] = ca[1] * ax[1] + c; R[8] = ca[2] * ax[2] + c;
  float O[16];
  for (int k = 0; k < 16; k++) O[k] = P[k];
  O[12] = P[12] + td[0]; O[13] = P[13] + td[1]; O[14] = P[14] + td[2];
  for (int r = 0; r < 3; r++)
    for (int cc = 0; cc < 3; cc++) {
      float sacc = R[0 * 3 + r] * P[cc * 4 + 0];
      sacc = sacc + R[1 * 3 + r] * P[cc * 4 + 1];
      sacc = sacc + R[2 * 3 + r] * P[cc * 4 + 2];
      O[cc * 4 + r] = sacc;
    }
  for (int k = 0; k < 16; k++) poses[(size_t)i * 16 + k] = O[k];
  if (extra_out)
    for (int k = 0; k < 16; k++) extra_out[(size_t)i * 16 + k] = O[k];
}
#endif
struct FrameRef;
// tiny batches: the same + the observed-crop warp of hypotheses [0, n_crop) in ONE launch (false: not available for this output mode);
// tri_rows non-null: the launch also leaves the [N, F] row ranges launch_tri_rows would compute from its clip coordinates
bool launch_setup_vertex_crop(hipStream_t s, const DeviceMesh &m, const float *poses_dev, int N, const float *K9_host, int img_h, int img_w,
                              float crop_ratio, float diameter, PoseRec *recs, float4 *clip, float4 *attr, bool fmad, const FrameRef *frame,
                              int n_crop, OutMode mode, void *out_b, unsigned *tri_rows);
// tri_rows: [N, F] row ranges from launch_tri_rows of the same clip coordinates (a strip then skips the triangles that miss it), or null
void launch_tri_rows(hipStream_t s, const DeviceMesh &m, int N, const float4 *clip, unsigned *rows);
bool raster_wants_tri_rows(int N);   // false for batches that are rendered in two tall strips per crop
void launch_raster_shade(hipStream_t s, const DeviceMesh &m, const PoseRec *recs, int N, const float4 *clip,
                         const float4 *attr, OutMode mode, void *out, int32_t *tri_id_dbg, float *rast_dbg, bool fmad,
                         const unsigned *tri_rows = nullptr);
#ifdef FP_TEST_HOOKS
void set_raster_strip_rows(int rows);  // 0 = automatic (A/B hook)
void set_tri_rows_tall(int v);
void set_raster_strip_threads(int threads);  // 0 = by batch size; 256 / 512 / 1024 (A/B hook, 8-row strips)
#endif
// the frame a replayed hipGraph reads: kernels inside graphs take the frame through this device-resident record, so a caller's
// device frame is used in place (no copy into model-owned buffers) and the graph stays valid when the pointers change
// how the kernels of a (replayed) graph find the current frame.  Whole frames: pitch = 0 (rows are W pixels apart), window = all.
// [r4] Track's packed crop window of a host frame: rgb / depth are VIRTUAL origins (the address pixel (0, 0) would have if the packed
// window were part of a frame with rows `pitch` pixels apart), and only pixels inside [wx0, wx1) x [wy0, wy1) exist.
struct FrameRef {
  const uint8_t *rgb;
  const float *depth;
  int pitch = 0;
  int wx0 = 0, wy0 = 0, wx1 = 0x7fffffff, wy1 = 0x7fffffff;
};
void launch_crop(hipStream_t s, const FrameRef *frame_dev, int H, int W, const float *K9_host,
                 const PoseRec *recs, int N, float diameter, OutMode mode, void *out);
void launch_depth_to_xyz(hipStream_t s, const float *depth, int H, int W, const float *K9_host, float *xyz);
// copies `bytes` (rounded up to 16) from a host-pinned, device-mapped block into device memory with a kernel
void launch_window_fetch(hipStream_t s, const void *src_host_mapped, void *dst, size_t bytes);
void launch_erode(hipStream_t s, const float *depth, float *out, int H, int W);
void launch_bilateral(hipStream_t s, const float *depth, float *out, int H, int W);
// device-side refine post process: poses updated in place from trans/rot [N,3] (foundationpose.cpp:360-406)
// poses_in (optional): read the poses from there instead of `poses`; extra_out (optional): a second copy of the result (Track: both
// are host-pinned, so no copy kernels run around the graph)
void launch_pose_update(hipStream_t s, float *poses, const float *trans, const float *rot, int N, float diameter,
                        const float *poses_in = nullptr, float *extra_out = nullptr);
// first-max arg-max over scores[N] -> *index (foundationpose_decoder.cu:24-35)
void launch_argmax(hipStream_t s, const float *scores, int N, int *index_dev, const float *poses = nullptr,
                   float *best_pose_dev = nullptr);
// GuessTranslation + hypothesis poses on the device; state = int[8] (status in state[6]: 0 ok, 1 empty mask, 2 no valid depth)
void launch_sampler(hipStream_t s, const float *filtered_depth, const uint8_t *mask_dev, int H, int W, float min_depth,
                    const float *K9_host, const float *grid_dev, int first, int N, int *state, float *vals, float *poses);
// f32 [N,160,160,6] -> the networks' 2-byte s2d input tensor (blob-mode entry points); mode = OUT_F16X8 / OUT_BF16X8
void launch_pack_f32x6(hipStream_t s, const float *in, void *out, size_t pixels, OutMode mode);

// host helpers (fp_host.cpp part of fp_api.hip)
std::vector<float> make_rotation_grid(int min_views, int inplane_steps);

}  // namespace fp
