// fp_geometry.hip -- per-hypothesis mesh rasteriser, RGB-D crop/warp, sampler filters, pose update, arg-max.
// gfx950 (MI355X) only.  Compiled with -ffp-contract=off: float expressions follow the operand order of the
// reference kernels so the tri-id buffer is bit-exact against the oracle and the float tensors agree to ulps.
//
// Reference semantics followed (path:line under zz990099/foundationpose_cpp, D6F = detection_6d_foundationpose/src,
// CR = D6F/nvdiffrast/common/cudaraster/impl, NVDR = D6F/nvdiffrast/common):
//   pose setup      D6F/foundationpose_render.cpp:25-75 (crop tf), :123-149 (bbox2d), :151-186 (projection), :590
//   vertex stage    D6F/foundationpose_render.cu:321-341 (K4), :363-398 (K5), :418-443 (K13)
//   rasteriser      CR/TriangleSetup.inl:11-24,42-58,120-177,181-391; CR/Util.inl:101-160,184-210,304-309;
//                   CR/FineRaster.inl:75-101,152-172,330-351; CR/Constants.hpp:22-27,78-80
//   shading         NVDR/rasterize.cu:15-90, NVDR/interpolate.cu:16-84, NVDR/texture.cu:20-96,132-179,
//                   D6F/foundationpose_render.cu:459-501 (K14), :30-39 (K15), foundationpose_render.cpp:676-680 (flip),
//                   D6F/foundationpose_render.cu:61-118 (K17), :121-140 (K18)
//   crop / warp     D6F/foundationpose_render.cpp:731-812 (cvcuda WarpPerspective/ConvertTo), foundationpose_utils.cu:3-32
//   sampler filters D6F/foundationpose_sampling.cu:21-82, :84-164
//   pose update     D6F/foundationpose.cpp:360-406;  arg-max D6F/foundationpose_decoder.cu:24-35
//
// MI355X design: one fused raster+shade kernel replaces K6-K18 of the render branch.  A workgroup owns one
// (hypothesis, 40-row strip): the strip's z-buffer lives in LDS as 64-bit keys (depth<<32 | ~colour) resolved with
// ds_min_u64, so "nearest wins, ties -> later triangle" (the CudaRaster ROP rule) is order independent; the
// shading pass reads the keys back from LDS and writes the final NHWC tensor once (fp16x8 or fp32x6), instead of the
// reference's ~14 full-tensor passes.  Triangles are tiny (~5 px) so lanes map to triangles, not pixels.

#include "fp_internal.h"
#include <cstdlib>

namespace fp {

#define CR_SUBPIXEL_LOG2 4
#define CR_MAXVIEWPORT_LOG2 11
#define CR_DEPTH_MIN (2200u << 3)
#define CR_DEPTH_MAX (0xFFFFFFFFu - (2200u << 3))

static constexpr int CROP = FP_CROP_HW;
// rows of the 160-row viewport owned by one workgroup: 20 (8 strips/hypothesis) for large batches, 8 (20 strips) when
// the batch alone cannot fill the chip (Track: N = 1).  The shading pass is a chain of dependent loads per pixel, so
// its latency is hidden by workgroup count, not by work per thread.

struct K9 { float k[9]; };

__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }

// ---- float model -----------------------------------------------------------------------------------------------
// This file is compiled with -ffp-contract=off: the compiler never fuses a multiply with an add on its own.  The reference's
// kernels are built by nvcc with its default -fmad=true (D6F/CMakeLists.txt:5 only sets -O3), i.e. its binary DOES contract.
// FMAD = true restates that with explicit fmaf calls under one documented rule [EXT: ptxas' actual choices are not
// published]: in a sum of products evaluated left to right, every `acc + a*b` whose product has no other use becomes
// fmaf(a, b, acc); the FIRST product of a chain is a plain multiply; `x - a*b` is fmaf(-a, b, x); `a*b - c*d` is
// fmaf(a, b, -(c*d)).  FMAD = false keeps every operation separately rounded (round 1's model).  the CPU oracle (oracle/, test infrastructure) carries
// the same two models expression by expression, so integer results (triangle ids) are bit-exact in either.
template <bool FMAD> __device__ __forceinline__ float mad(float a, float b, float c) { return FMAD ? fmaf(a, b, c) : a * b + c; }
template <bool FMAD> __device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
  return mad<FMAD>(a2, b2, mad<FMAD>(a1, b1, a0 * b0));
}
template <bool FMAD> __device__ __forceinline__ float diffprod(float a, float b, float c, float d) {  // a*b - c*d
  return FMAD ? fmaf(a, b, -(c * d)) : a * b - c * d;
}

// fp16 network-input layout: space-to-depth(2x2) of NHWC [N,160,160,8] -> [N,80,80,32], stored with a physical zero
// border of FP_NN_IN_BORDER s2d-pixels ([N,84,84,32]); in 16-byte units the pixel (n,y,x) lands at
// ((n*84 + y/2 + 2)*84 + x/2 + 2)*4 + (y&1)*2 + (x&1).  This turns the 7x7 stride-2 stem convolution into a 4x4
// stride-1 convolution with Cin = 32 that the generic MFMA implicit-GEMM kernel runs without bounds checks.
// six channel values -> one 16-byte network-input pixel (r,g,b,x,y,z,0,0) in the 2-byte type of the output mode
template <int MODE>
__device__ __forceinline__ uint4 pack6(const float (&o)[6]) {
  if constexpr (MODE == OUT_BF16X8) {
    union { __bf16 h[8]; uint4 u4; } pk;
    for (int c = 0; c < 6; c++) pk.h[c] = (__bf16)o[c];
    pk.h[6] = (__bf16)0.f; pk.h[7] = (__bf16)0.f;
    return pk.u4;
  } else {
    union { __half h[8]; uint4 u4; } pk;
    for (int c = 0; c < 6; c++) pk.h[c] = __float2half(o[c]);
    pk.h[6] = __float2half(0.f); pk.h[7] = __float2half(0.f);
    return pk.u4;
  }
}
__device__ __forceinline__ size_t s2d_index(size_t n, int y, int x) {
  constexpr int P = CROP / 2 + 2 * FP_NN_IN_BORDER;
  return ((n * P + (size_t)((y >> 1) + FP_NN_IN_BORDER)) * P + (size_t)((x >> 1) + FP_NN_IN_BORDER)) * 4 +
         (size_t)((y & 1) * 2 + (x & 1));
}

// ---------------------------------------------------------------------------------------------
// pose setup
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ void mat4_mul(const float *A, const float *B, float *C) {
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++) {
      float s = A[i + 0] * B[j * 4 + 0];
      s = s + A[i + 4] * B[j * 4 + 1];
      s = s + A[i + 8] * B[j * 4 + 2];
      s = s + A[i + 12] * B[j * 4 + 3];
      C[j * 4 + i] = s;
    }
}

// the per-hypothesis record (crop window, bounding box, projection): one function for the stand-alone kernel and for the vertex
// kernel that computes it in place (one launch less per render)
__device__ __forceinline__ void make_pose_rec(const float *__restrict__ poses, int i, const K9 &K, int img_h, int img_w, float crop_ratio,
                                              float diameter, PoseRec &rec) {
  for (int k = 0; k < 16; k++) rec.pose[k] = poses[(size_t)i * 16 + k];
  // crop window
  float r = diameter * crop_ratio / 2;
  float tx = rec.pose[12], ty = rec.pose[13], tz = rec.pose[14];
  float u0 = 0, v0 = 0, mx = 0;
  for (int k = 0; k < 5; k++) {
    float ox = (k == 1) ? r : (k == 2 ? -r : 0.0f);
    float oy = (k == 3) ? r : (k == 4 ? -r : 0.0f);
    float px = tx + ox, py = ty + oy, pz = tz + 0.0f;
    float q[3];
    for (int rr = 0; rr < 3; rr++) {
      float s = K.k[rr * 3] * px;
      s = s + K.k[rr * 3 + 1] * py;
      s = s + K.k[rr * 3 + 2] * pz;
      q[rr] = s;
    }
    float u = q[0] / q[2], v = q[1] / q[2];
    if (k == 0) { u0 = u; v0 = v; mx = v - v0; }
    else { float d = v - v0; if (d > mx) mx = d; }
  }
  float radius = fabsf(mx);
  float left = roundf(u0 - radius), right = roundf(u0 + radius);
  float top = roundf(v0 - radius), bottom = roundf(v0 + radius);
  float sx = (float)CROP / (right - left), sy = (float)CROP / (bottom - top);
  rec.tf[0] = sx; rec.tf[1] = 0; rec.tf[2] = sx * (-left);
  rec.tf[3] = 0; rec.tf[4] = sy; rec.tf[5] = sy * (-top);
  rec.tf[6] = 0; rec.tf[7] = 0; rec.tf[8] = 1;
  // bbox2d = tf^-1 {(0,0),(159,159)}
  float i00 = 1.0f / rec.tf[0], i11 = 1.0f / rec.tf[4];
  float i02 = -rec.tf[2] / rec.tf[0], i12 = -rec.tf[5] / rec.tf[4];
  float x1 = (float)(CROP - 1), y1 = (float)(CROP - 1);
  rec.bbox[0] = (i00 * 0.0f + 0.0f * 0.0f) + i02;
  rec.bbox[1] = (0.0f * 0.0f + i11 * 0.0f) + i12;
  rec.bbox[2] = (i00 * x1 + 0.0f * y1) + i02;
  rec.bbox[3] = (0.0f * x1 + i11 * y1) + i12;
  float l = rec.bbox[0], t = img_h - rec.bbox[1], rr2 = rec.bbox[2], b = img_h - rec.bbox[3];
  rec.a00 = img_w / (rr2 - l);
  rec.a11 = img_h / (t - b);
  rec.a30 = (img_w - rr2 - l) / (rr2 - l);
  rec.a31 = (img_h - t - b) / (t - b);
  // inverse crop transform for the warp (double, cast to float)
  {
    double a = rec.tf[0], c = rec.tf[2], e = rec.tf[4], f = rec.tf[5];
    rec.m0 = (float)(1.0 / a); rec.m2 = (float)(-c / a); rec.m4 = (float)(1.0 / e); rec.m5 = (float)(-f / e);
  }
  // projection (y_down), znear 0.1, zfar 100
  float P[16];
  {
    int w = img_w, h = img_h;
    float nc = 0.1f, fc = 100.0f;
    float depth = fc - nc, q = -(fc + nc) / depth, qn = -2 * (fc * nc) / depth;
    float rm[16] = {2 * K.k[0] / w, -2 * K.k[1] / w, (-2 * K.k[2] + w + 2 * 0) / w, 0,
                    0, 2 * K.k[4] / h, (2 * K.k[5] - h + 2 * 0) / h, 0,
                    0, 0, q, qn,
                    0, 0, -1, 0};
    for (int rr = 0; rr < 4; rr++) for (int c = 0; c < 4; c++) P[c * 4 + rr] = rm[rr * 4 + c];
  }
  const float GL[16] = {1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1};
  float GP[16];
  mat4_mul(GL, rec.pose, GP);
  mat4_mul(P, GP, rec.M);
}

__global__ void pose_setup_kernel(const float *__restrict__ poses, int N, K9 K, int img_h, int img_w,
                                  float crop_ratio, float diameter, PoseRec *__restrict__ recs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  PoseRec rec;
  make_pose_rec(poses, i, K, img_h, img_w, crop_ratio, diameter, rec);
  recs[i] = rec;
}

void launch_pose_setup(hipStream_t s, const float *poses_dev, int N, const float *K9_host, int img_h, int img_w,
                       float crop_ratio, float diameter, PoseRec *recs) {
  K9 K;
  for (int i = 0; i < 9; i++) K.k[i] = K9_host[i];
  hipLaunchKernelGGL(pose_setup_kernel, dim3((N + 63) / 64), dim3(64), 0, s, poses_dev, N, K, img_h, img_w,
                     crop_ratio, diameter, recs);
}

// ---------------------------------------------------------------------------------------------
// vertex stage: clip position, camera-space point, per-vertex Lambert term
// ---------------------------------------------------------------------------------------------

// The per-hypothesis record is read through uniform (scalar) loads -- or, SETUP, computed here by every thread from the pose (the
// same function as pose_setup_kernel: identical values) and written out once per hypothesis for the rasteriser and the crop kernel.
struct PoseSetupArgs {
  const float *poses;
  K9 K;
  int img_h, img_w;
  float crop_ratio, diameter;
};
// clip-space position of one model vertex: the ONE place this arithmetic is written (the vertex stage stores it, the fused
// row-range blocks of vertex_crop_kernel recompute it for the three corners of their triangle -- same expressions, no
// contraction in this translation unit, hence the same bits)
template <bool FMAD>
__device__ __forceinline__ float4 clip_of(const PoseRec &rec, float x, float y, float z) {
  const float *M = rec.M;
  float tx = dot3<FMAD>(M[0], x, M[4], y, M[8], z) + M[12];
  float ty = dot3<FMAD>(M[1], x, M[5], y, M[9], z) + M[13];
  float tz = dot3<FMAD>(M[2], x, M[6], y, M[10], z) + M[14];
  float tw = dot3<FMAD>(M[3], x, M[7], y, M[11], z) + M[15];
  float4 c;
  c.x = mad<FMAD>(tw, rec.a30, tx * rec.a00);
  c.y = mad<FMAD>(tw, rec.a31, ty * rec.a11);
  c.z = tz;
  c.w = tw;
  return c;
}

// one vertex of hypothesis n under the record `rec`
template <bool FMAD>
__device__ __forceinline__ void vertex_body(const float *__restrict__ verts, const float *__restrict__ normals, int V, int v, int n, const PoseRec &rec,
                                            float4 *__restrict__ clip, float4 *__restrict__ attr, float4 *__restrict__ dbg, unsigned long long t_begin) {
  const float *pose = rec.pose;
  float x = verts[v * 3], y = verts[v * 3 + 1], z = verts[v * 3 + 2];
  const float4 c = clip_of<FMAD>(rec, x, y, z);
  float4 a;
  a.x = dot3<FMAD>(pose[0], x, pose[4], y, pose[8], z) + pose[12];
  a.y = dot3<FMAD>(pose[1], x, pose[5], y, pose[9], z) + pose[13];
  a.z = dot3<FMAD>(pose[2], x, pose[6], y, pose[10], z) + pose[14];
  float nx = normals[v * 3], ny = normals[v * 3 + 1], nz = normals[v * 3 + 2];
  float ux = dot3<FMAD>(pose[0], nx, pose[4], ny, pose[8], nz);
  float uy = dot3<FMAD>(pose[1], nx, pose[5], ny, pose[9], nz);
  float uz = dot3<FMAD>(pose[2], nx, pose[6], ny, pose[10], nz);
  float l2 = sqrtf(dot3<FMAD>(ux, ux, uy, uy, uz, uz));
  float val = l2 == 0 ? 0 : -uz / l2;
  a.w = clampf(val, 0, 1);
  clip[(size_t)n * V + v] = c;
  attr[(size_t)n * V + v] = a;
#ifdef FP_TEST_HOOKS
  if (dbg) {  // tools/dbg_concurrent3.py: per-vertex intermediates, wave duration and placement
    const unsigned long long t_end = wall_clock64();
    unsigned hwid = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    dbg[((size_t)n * V + v) * 3] = make_float4(nx, ny, nz, l2);
    dbg[((size_t)n * V + v) * 3 + 1] = make_float4(ux, uy, uz, val);
    dbg[((size_t)n * V + v) * 3 + 2] = make_float4((float)(t_end - t_begin), __uint_as_float(hwid), __uint_as_float((unsigned)(t_begin & 0xffffffffu)), 0.f);
  }
#else
  (void)dbg; (void)t_begin;
#endif
}

template <bool FMAD, bool SETUP>
__global__ void vertex_kernel(const float *__restrict__ verts, const float *__restrict__ normals, int V,
                              PoseRec *__restrict__ recs, float4 *__restrict__ clip, float4 *__restrict__ attr,
                              float4 *__restrict__ dbg, const PoseSetupArgs sa) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
#ifdef FP_TEST_HOOKS
  const unsigned long long t_begin = dbg ? wall_clock64() : 0ull;
#else
  const unsigned long long t_begin = 0ull;
#endif
  if (v >= V) return;
  PoseRec own;
  if constexpr (SETUP) {
    make_pose_rec(sa.poses, n, sa.K, sa.img_h, sa.img_w, sa.crop_ratio, sa.diameter, own);
    if (v == 0) recs[n] = own;
  }
  const PoseRec &rec = SETUP ? own : recs[n];
  vertex_body<FMAD>(verts, normals, V, v, n, rec, clip, attr, dbg, t_begin);
}

#ifdef FP_TEST_HOOKS
float4 *g_vertex_dbg = nullptr;  // race hunt (tools/dbg_concurrent3.py): launches with N == 64 fill it
#endif
void launch_vertex(hipStream_t s, const DeviceMesh &m, const PoseRec *recs, int N, float4 *clip, float4 *attr, bool fmad) {
  float4 *dbg = nullptr;
#ifdef FP_TEST_HOOKS
  if (g_vertex_dbg && N == 64) dbg = g_vertex_dbg;
#endif
  const PoseSetupArgs none{};
  PoseRec *r = const_cast<PoseRec *>(recs);
  if (fmad) hipLaunchKernelGGL((vertex_kernel<true, false>), dim3((m.V + 255) / 256, N), dim3(256), 0, s, m.verts, m.normals, m.V, r, clip, attr, dbg, none);
  else hipLaunchKernelGGL((vertex_kernel<false, false>), dim3((m.V + 255) / 256, N), dim3(256), 0, s, m.verts, m.normals, m.V, r, clip, attr, dbg, none);
}

// pose set-up + vertex stage in one launch: recs[0..N) are WRITTEN here (by the thread of vertex 0 of each hypothesis)
void launch_setup_vertex(hipStream_t s, const DeviceMesh &m, const float *poses_dev, int N, const float *K9_host, int img_h, int img_w,
                         float crop_ratio, float diameter, PoseRec *recs, float4 *clip, float4 *attr, bool fmad) {
  PoseSetupArgs sa;
  sa.poses = poses_dev;
  for (int i = 0; i < 9; i++) sa.K.k[i] = K9_host[i];
  sa.img_h = img_h; sa.img_w = img_w; sa.crop_ratio = crop_ratio; sa.diameter = diameter;
  float4 *dbg = nullptr;
#ifdef FP_TEST_HOOKS
  if (g_vertex_dbg && N == 64) dbg = g_vertex_dbg;
#endif
  if (fmad) hipLaunchKernelGGL((vertex_kernel<true, true>), dim3((m.V + 255) / 256, N), dim3(256), 0, s, m.verts, m.normals, m.V, recs, clip, attr, dbg, sa);
  else hipLaunchKernelGGL((vertex_kernel<false, true>), dim3((m.V + 255) / 256, N), dim3(256), 0, s, m.verts, m.normals, m.V, recs, clip, attr, dbg, sa);
}

// ---------------------------------------------------------------------------------------------
// rasteriser (CudaRaster semantics) + shading, fused
// ---------------------------------------------------------------------------------------------

struct I2 { int x, y; };

__device__ __forceinline__ int f32_to_s32_sat(float a) {  // cvt.rni.sat.s32.f32
  if (a != a) return 0;
  float r = rintf(a);
  if (r >= 2147483648.0f) return 2147483647;
  if (r <= -2147483648.0f) return (int)(-2147483647 - 1);
  return (int)r;
}
__device__ __forceinline__ unsigned f32_to_u32_trunc(float a) {  // (U32)float as CUDA: cvt.rzi.u32.f32 saturating
  if (a != a) return 0u;
  if (a >= 4294967296.0f) return 0xFFFFFFFFu;
  if (a <= 0.0f) return 0u;
  return (unsigned)a;
}
__device__ __forceinline__ int imin3(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(max(a, b), c); }

__device__ __forceinline__ bool edge_covers(int ox, int oy, int dx, int dy) {
  int e = ox * dy - oy * dx;
  if (dy > 0 || (dy == 0 && dx <= 0)) e--;  // exclusive edges: top-left fill rule (CR/Util.inl:304-309)
  return e >= 0;
}

// 32-bit fixed-point depth plane of a triangle, depth(px, py) = gx*px + gy*py + g0 (mod 2^32), with CudaRaster's
// arithmetic (setupPleq, CR/impl/Util.inl:184-210: the results must agree bit for bit, so the integer operations and
// their order are the reference's; the decomposition into steps is this file's):
//   1. the three depths share one block exponent: `drop` low bits are shifted out so that the largest keeps 23 bits;
//   2. gradients = (depth differences x edge vectors) / area, the division carried out as a multiplication by the 24-bit
//      mantissa of 1/area and one final shift;
//   3. the constant term is anchored at the centre of the triangle's bounding box to keep the products small.
struct DepthPlane { unsigned gx, gy, g0; };
__device__ __forceinline__ DepthPlane make_depth_plane(float zA, float zB, float zC, I2 A /* vertex A, window sub-pixels minus half a pixel */,
                                                       I2 e1, I2 e2, int area) {
  const float inv_area = 1.0f / (float)area;
  const float zmax = fmaxf(fmaxf(zA, zB), zC);
  const int drop = min(max((__float_as_int(zmax) >> 23) - (127 + 22), 0), 8);
  const int qA = (int)(f32_to_u32_trunc(zA) >> drop);
  const int dB = (int)((f32_to_u32_trunc(zB) >> drop) - (unsigned)qA);
  const int dC = (int)((f32_to_u32_trunc(zC) >> drop) - (unsigned)qA);
  const unsigned mant = ((unsigned)__float_as_int(inv_area) & 0x007FFFFFu) | 0x00800000u;
  const int expo = (23 + 127) - (__float_as_int(inv_area) >> 23);
  const long long nx = ((long long)dB * e2.y - (long long)dC * e1.y) * (long long)mant;
  const long long ny = ((long long)dC * e1.x - (long long)dB * e2.x) * (long long)mant;
  DepthPlane pl;
  pl.gx = (unsigned)(nx >> (expo - (drop + CR_SUBPIXEL_LOG2)));
  pl.gy = (unsigned)(ny >> (expo - (drop + CR_SUBPIXEL_LOG2)));
  const int cx = (A.x * 2 + imin3(e1.x, e2.x, 0) + imax3(e1.x, e2.x, 0)) >> (CR_SUBPIXEL_LOG2 + 1);
  const int cy = (A.y * 2 + imin3(e1.y, e2.y, 0) + imax3(e1.y, e2.y, 0)) >> (CR_SUBPIXEL_LOG2 + 1);
  const int ox = A.x - (int)((unsigned)cx << CR_SUBPIXEL_LOG2);
  const int oy = A.y - (int)((unsigned)cy << CR_SUBPIXEL_LOG2);
  pl.g0 = (unsigned)qA << drop;
  pl.g0 -= (unsigned)(((nx >> 13) * ox + (ny >> 13) * oy) >> (expo - (drop + 13)));
  pl.g0 -= pl.gx * (unsigned)cx + pl.gy * (unsigned)cy;
  return pl;
}

// snap + setup + rasterise one (sub)triangle into the strip's LDS z-buffer
template <int STRIP_ROWS>
__device__ __forceinline__ void raster_one(float4 v0, float4 v1, float4 v2, unsigned color, int row0,
                                           unsigned long long *zbuf) {
  const float vs = (float)(CROP << (CR_SUBPIXEL_LOG2 - 1));
  float rw0 = 1.0f / v0.w, rw1 = 1.0f / v1.w, rw2 = 1.0f / v2.w;
  I2 p0 = {f32_to_s32_sat(v0.x * rw0 * vs), f32_to_s32_sat(v0.y * rw0 * vs)};
  I2 p1 = {f32_to_s32_sat(v1.x * rw1 * vs), f32_to_s32_sat(v1.y * rw1 * vs)};
  I2 p2 = {f32_to_s32_sat(v2.x * rw2 * vs), f32_to_s32_sat(v2.y * rw2 * vs)};
  I2 d1 = {p1.x - p0.x, p1.y - p0.y}, d2 = {p2.x - p0.x, p2.y - p0.y};
  int area = d1.x * d2.y - d1.y * d2.x;
  if (area == 0) return;
  float z0 = v0.z, z1 = v1.z, z2 = v2.z;
  if (area < 0) {
    I2 t = d1; d1 = d2; d2 = t; t = p1; p1 = p2; p2 = t;
    float f = z1; z1 = z2; z2 = f; f = rw1; rw1 = rw2; rw2 = f;
    area = -area;
  }
  // pixel bounding box clipped to this strip
  const int bx = (CROP - 1) << (CR_SUBPIXEL_LOG2 - 1);
  int minx = imin3(p0.x, p1.x, p2.x), maxx = imax3(p0.x, p1.x, p2.x);
  int miny = imin3(p0.y, p1.y, p2.y), maxy = imax3(p0.y, p1.y, p2.y);
  int px0 = max((minx + bx + 15) >> 4, 0), px1 = min((maxx + bx) >> 4, CROP - 1);
  int py0 = max((miny + bx + 15) >> 4, row0), py1 = min((maxy + bx) >> 4, row0 + STRIP_ROWS - 1);
  if (px0 > px1 || py0 > py1) return;
  // window-space depths in the 32-bit fixed-point range (the one fused multiply-add the reference's source spells out)
  const float zcoef = (float)(CR_DEPTH_MAX - CR_DEPTH_MIN) * 0.5f;
  const float zbias = (float)(unsigned)(CR_DEPTH_MAX + CR_DEPTH_MIN) * 0.5f;
  const float zv0 = fmaf(z0 * zcoef, rw0, zbias), zv1 = fmaf(z1 * zcoef, rw1, zbias), zv2 = fmaf(z2 * zcoef, rw2, zbias);
  const I2 anchor = {p0.x + (CROP << (CR_SUBPIXEL_LOG2 - 1)) - (1 << (CR_SUBPIXEL_LOG2 - 1)),
                     p0.y + (CROP << (CR_SUBPIXEL_LOG2 - 1)) - (1 << (CR_SUBPIXEL_LOG2 - 1))};
  const DepthPlane pl = make_depth_plane(zv0, zv1, zv2, anchor, d1, d2, area);
  int d01x = p1.x - p0.x, d01y = p1.y - p0.y;
  int d12x = p2.x - p1.x, d12y = p2.y - p1.y;
  int d20x = p0.x - p2.x, d20y = p0.y - p2.y;
  unsigned long long lowkey = (unsigned long long)(~color);
  for (int py = py0; py <= py1; py++) {
    int sy = py * 16 - bx;
    int o0y = p0.y - sy;
    for (int px = px0; px <= px1; px++) {
      int sx = px * 16 - bx;
      int o0x = p0.x - sx;
      if (!edge_covers(o0x, o0y, d01x, d01y)) continue;
      if (!edge_covers(o0x + d01x, o0y + d01y, d12x, d12y)) continue;
      if (!edge_covers(o0x, o0y, d20x, d20y)) continue;
      unsigned depth = pl.gx * (unsigned)px + pl.gy * (unsigned)py + pl.g0;
      unsigned long long key = ((unsigned long long)depth << 32) | lowkey;
      atomicMin(&zbuf[(py - row0) * CROP + px], key);
    }
  }
}

// One Sutherland-Hodgman pass.  The polygon lives in the triangle's barycentric plane (points (u, v)); the half-space
// kept is f(u, v) = c0 + c1*u + c2*v >= 0 -- for a frustum plane w +- x_axis >= 0 that is (w0 +- a0) + (dw1 +- da1)*u +
// (dw2 +- da2)*v.  An edge whose end points have values of opposite sign contributes its intersection, interpolated with
// the weights fp / (fp - fc) like the reference's clipper (CR/impl/Util.inl:101-132), so that the fan fed to the
// rasteriser is identical; FMAD applies the float model above to the two sums.
template <bool FMAD>
__device__ __forceinline__ int sh_clip_pass(const float (*src)[2], int n, float (*dst)[2], float c0, float c1, float c2) {
  if (n < 3) return 0;
  int m = 0;
  float pu = src[n - 1][0], pv = src[n - 1][1];
  float pf = FMAD ? fmaf(c2, pv, fmaf(c1, pu, c0)) : c0 + c1 * pu + c2 * pv;
  for (int i = 0; i < n; i++) {
    const float cu = src[i][0], cv = src[i][1];
    const float cf = FMAD ? fmaf(c2, cv, fmaf(c1, cu, c0)) : c0 + c1 * cu + c2 * cv;
    if (pf * cf < 0.0f) {  // the edge previous -> current crosses the plane
      const float wc = pf / (pf - cf), wp = 1.0f - wc;
      dst[m][0] = mad<FMAD>(cu, wc, pu * wp);
      dst[m][1] = mad<FMAD>(cv, wc, pv * wp);
      m++;
    }
    if (cf >= 0.0f) { dst[m][0] = cu; dst[m][1] = cv; m++; }
    pu = cu; pv = cv; pf = cf;
  }
  return m;
}

// rare path: triangle crosses the depth range or leaves the S16 snap range -> clip against the frustum and fan
template <int STRIP_ROWS, bool FMAD>
__device__ __noinline__ void raster_clipped(float4 v0, float4 v1, float4 v2, unsigned color, int row0,
                                            unsigned long long *zbuf) {
  float poly[2][9][2];  // ping-pong polygon buffers: at most 3 + 6 vertices
  int cur = 0, num = 3;
  poly[0][0][0] = 0.0f; poly[0][0][1] = 0.0f; poly[0][1][0] = 1.0f; poly[0][1][1] = 0.0f; poly[0][2][0] = 0.0f; poly[0][2][1] = 1.0f;
  const float a0[4] = {v0.x, v0.y, v0.z, v0.w}, a1[4] = {v1.x, v1.y, v1.z, v1.w}, a2[4] = {v2.x, v2.y, v2.z, v2.w};
  const float e1[4] = {v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, v1.w - v0.w};
  const float e2[4] = {v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, v2.w - v0.w};
  for (int ax = 0; ax < 3; ax++) {
    const bool outside = (a0[3] < fabsf(a0[ax])) | (a1[3] < fabsf(a1[ax])) | (a2[3] < fabsf(a2[ax]));
    if (!outside) continue;
    num = sh_clip_pass<FMAD>(poly[cur], num, poly[cur ^ 1], a0[3] + a0[ax], e1[3] + e1[ax], e2[3] + e2[ax]);  // w + axis >= 0
    num = sh_clip_pass<FMAD>(poly[cur ^ 1], num, poly[cur], a0[3] - a0[ax], e1[3] - e1[ax], e2[3] - e2[ax]);  // w - axis >= 0
  }
  if (num < 3) return;
  auto point = [&](int i) {
    const float u = poly[cur][i][0], v = poly[cur][i][1];
    float4 q;
    q.x = mad<FMAD>(e2[0], v, mad<FMAD>(e1[0], u, a0[0]));
    q.y = mad<FMAD>(e2[1], v, mad<FMAD>(e1[1], u, a0[1]));
    q.z = mad<FMAD>(e2[2], v, mad<FMAD>(e1[2], u, a0[2]));
    q.w = mad<FMAD>(e2[3], v, mad<FMAD>(e1[3], u, a0[3]));
    return q;
  };
  const float4 hub = point(0);
  float4 prev = point(1);
  for (int i = 2; i < num; i++) {  // triangle fan around the first vertex
    const float4 next = point(i);
    raster_one<STRIP_ROWS>(hub, prev, next, color, row0, zbuf);
    prev = next;
  }
}

// [r4] Row range of every triangle, once per hypothesis: a strip of the rasteriser below then only sets up the triangles whose rows
// meet it (each strip used to walk ALL triangles: 40 strips x F set-ups for one Track crop; 58 us at 20 k triangles, 186 us at 82 k).
// The range is what raster_one would compute -- the same snapped vertices, the same bounding-box arithmetic -- so a triangle that is
// skipped is one raster_one would have left at its bounding-box test; triangles on the clipping path get the whole crop, culled or
// degenerate ones the empty range.  Packed lo | hi << 16; empty = 1 | 0 << 16.
constexpr unsigned TRI_ROWS_EMPTY = 1u, TRI_ROWS_ALL = (unsigned)(CROP - 1) << 16;
__device__ __forceinline__ unsigned tri_rows_of(const float4 v0, const float4 v1, const float4 v2) {
  unsigned out = TRI_ROWS_EMPTY;
  {
    bool culled = false;
    if ((v0.w < fabsf(v0.x)) | (v0.w < fabsf(v0.y)) | (v0.w < fabsf(v0.z))) {
      culled = ((v0.w < +v0.x) & (v1.w < +v1.x) & (v2.w < +v2.x)) | ((v0.w < -v0.x) & (v1.w < -v1.x) & (v2.w < -v2.x)) |
               ((v0.w < +v0.y) & (v1.w < +v1.y) & (v2.w < +v2.y)) | ((v0.w < -v0.y) & (v1.w < -v1.y) & (v2.w < -v2.y)) |
               ((v0.w < +v0.z) & (v1.w < +v1.z) & (v2.w < +v2.z)) | ((v0.w < -v0.z) & (v1.w < -v1.z) & (v2.w < -v2.z));
    }
    if (!culled) {
      out = TRI_ROWS_ALL;   // the clipping path: every strip looks at it
      if ((v0.w >= fabsf(v0.z)) & (v1.w >= fabsf(v1.z)) & (v2.w >= fabsf(v2.z))) {
        const float vs = (float)(CROP << (CR_SUBPIXEL_LOG2 - 1));
        const float rw0 = 1.0f / v0.w, rw1 = 1.0f / v1.w, rw2 = 1.0f / v2.w;
        const int ax = f32_to_s32_sat(v0.x * rw0 * vs), ay = f32_to_s32_sat(v0.y * rw0 * vs);
        const int bx_ = f32_to_s32_sat(v1.x * rw1 * vs), by_ = f32_to_s32_sat(v1.y * rw1 * vs);
        const int cx = f32_to_s32_sat(v2.x * rw2 * vs), cy = f32_to_s32_sat(v2.y * rw2 * vs);
        const int loxy = min(imin3(ax, bx_, cx), imin3(ay, by_, cy));
        const int hixy = max(imax3(ax, bx_, cx), imax3(ay, by_, cy));
        const int aabbLimit = (1 << (CR_MAXVIEWPORT_LOG2 + CR_SUBPIXEL_LOG2)) - 1;
        if (loxy >= -32768 && hixy <= 32767 && hixy - loxy <= aabbLimit) {   // the fast path: raster_one's own tests
          const int area = (bx_ - ax) * (cy - ay) - (by_ - ay) * (cx - ax);
          const int bx = (CROP - 1) << (CR_SUBPIXEL_LOG2 - 1);
          const int px0 = max((imin3(ax, bx_, cx) + bx + 15) >> 4, 0), px1 = min((imax3(ax, bx_, cx) + bx) >> 4, CROP - 1);
          const int py0 = max((imin3(ay, by_, cy) + bx + 15) >> 4, 0), py1 = min((imax3(ay, by_, cy) + bx) >> 4, CROP - 1);
          out = (area == 0 || px0 > px1 || py0 > py1) ? TRI_ROWS_EMPTY : ((unsigned)py0 | ((unsigned)py1 << 16));
        }
      }
    }
  }
  return out;
}
__global__ __launch_bounds__(256) void tri_rows_kernel(const int32_t *__restrict__ faces, int F, int V, const float4 *__restrict__ clip_all,
                                                       unsigned *__restrict__ rows_all) {
  const int f = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (f >= F) return;
  const float4 *clip = clip_all + (size_t)n * V;
  unsigned out = TRI_ROWS_EMPTY;
  const int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
  if ((unsigned)i0 < (unsigned)V && (unsigned)i1 < (unsigned)V && (unsigned)i2 < (unsigned)V) out = tri_rows_of(clip[i0], clip[i1], clip[i2]);
  rows_all[(size_t)n * F + f] = out;
}
void launch_tri_rows(hipStream_t s, const DeviceMesh &m, int N, const float4 *clip, unsigned *rows) {
  if (m.F <= 0 || N <= 0) return;
  hipLaunchKernelGGL(tri_rows_kernel, dim3((m.F + 255) / 256, N), dim3(256), 0, s, m.faces, m.F, m.V, clip, rows);
}
constexpr int TRI_LIST = 8192;   // capacity of the rasteriser's list of triangles that meet its strip (32 KB of LDS)

// NT threads per workgroup: 256 normally; 1024 for tiny batches (Track), where the kernel is bound by the latency of the
// F/NT dependent triangle iterations of each strip rather than by throughput
template <int MODE, int STRIP_ROWS, int NT, bool FMAD>
__global__ __launch_bounds__(NT) void raster_shade_kernel(
    const int32_t *__restrict__ faces, int F, int V, const float *__restrict__ uvs, const uint8_t *__restrict__ tex,
    int TH, int TW, float downscale, const PoseRec *__restrict__ recs, const float4 *__restrict__ clip_all,
    const float4 *__restrict__ attr_all, void *__restrict__ out_all, int32_t *__restrict__ tri_id_dbg,
    float *__restrict__ rast_dbg, const unsigned *__restrict__ tri_rows_all) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long *zbuf = reinterpret_cast<unsigned long long *>(smem);
  const int strip = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int row0 = strip * STRIP_ROWS;
  const float4 *clip = clip_all + (size_t)n * V;
  const float4 *attr = attr_all + (size_t)n * V;

  const unsigned long long clear_key = ((unsigned long long)CR_DEPTH_MAX << 32) | 0xFFFFFFFFull;
  for (int i = tid; i < STRIP_ROWS * CROP; i += NT) zbuf[i] = clear_key;
  __syncthreads();

  // triangle loop, software-pipelined two deep: a thread's iteration is a chain index load -> three vertex gathers -> setup, and
  // with F / NT = 5 iterations per thread at N = 1 (Track) those latencies added up; now the vertices of triangle k+1 and the
  // indices of triangle k+2 are in flight while triangle k is set up and rasterised
  auto process = [&](int f, const float4 &v0, const float4 &v1, const float4 &v2) {
    if ((v0.w < fabsf(v0.x)) | (v0.w < fabsf(v0.y)) | (v0.w < fabsf(v0.z))) {
      if (((v0.w < +v0.x) & (v1.w < +v1.x) & (v2.w < +v2.x)) | ((v0.w < -v0.x) & (v1.w < -v1.x) & (v2.w < -v2.x)) |
          ((v0.w < +v0.y) & (v1.w < +v1.y) & (v2.w < +v2.y)) | ((v0.w < -v0.y) & (v1.w < -v1.y) & (v2.w < -v2.y)) |
          ((v0.w < +v0.z) & (v1.w < +v1.z) & (v2.w < +v2.z)) | ((v0.w < -v0.z) & (v1.w < -v1.z) & (v2.w < -v2.z)))
        return;
    }
    bool fast = false;
    if ((v0.w >= fabsf(v0.z)) & (v1.w >= fabsf(v1.z)) & (v2.w >= fabsf(v2.z))) {
      const float vs = (float)(CROP << (CR_SUBPIXEL_LOG2 - 1));
      float rw0 = 1.0f / v0.w, rw1 = 1.0f / v1.w, rw2 = 1.0f / v2.w;
      int ax = f32_to_s32_sat(v0.x * rw0 * vs), ay = f32_to_s32_sat(v0.y * rw0 * vs);
      int bx_ = f32_to_s32_sat(v1.x * rw1 * vs), by_ = f32_to_s32_sat(v1.y * rw1 * vs);
      int cx = f32_to_s32_sat(v2.x * rw2 * vs), cy = f32_to_s32_sat(v2.y * rw2 * vs);
      int loxy = min(imin3(ax, bx_, cx), imin3(ay, by_, cy));
      int hixy = max(imax3(ax, bx_, cx), imax3(ay, by_, cy));
      const int aabbLimit = (1 << (CR_MAXVIEWPORT_LOG2 + CR_SUBPIXEL_LOG2)) - 1;
      fast = (loxy >= -32768 && hixy <= 32767 && hixy - loxy <= aabbLimit);
    }
    if (fast) raster_one<STRIP_ROWS>(v0, v1, v2, (unsigned)(f + 1), row0, zbuf);
    else raster_clipped<STRIP_ROWS, FMAD>(v0, v1, v2, (unsigned)(f + 1), row0, zbuf);
  };
  struct Tri { int i0, i1, i2; };
  auto load_idx = [&](int f) {
    Tri t{-1, -1, -1};
    if (f < F) { t.i0 = faces[f * 3]; t.i1 = faces[f * 3 + 1]; t.i2 = faces[f * 3 + 2]; }
    return t;
  };
  auto valid = [&](const Tri &t) { return (unsigned)t.i0 < (unsigned)V && (unsigned)t.i1 < (unsigned)V && (unsigned)t.i2 < (unsigned)V; };
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tri_rows_all) {
    // [r4] compacted: every thread tests the row ranges of 4 triangles per slab (4 coalesced bytes each, the loads independent) and
    // appends the ones that meet this strip to an LDS list; the list is set up and rasterised when another slab might not fit, and at
    // the end -- for a 4-row strip of a 20 k-triangle mesh that is ONE pass over ~1 000 survivors instead of 20 dependent
    // index -> vertex -> set-up iterations per thread.  The list order varies from run to run; the z-buffer keys (depth, triangle id)
    // make the result independent of it.
    static_assert(TRI_LIST >= 8 * NT, "a slab of 4 NT triangles must fit behind the flush threshold");
    unsigned *list = reinterpret_cast<unsigned *>(zbuf + STRIP_ROWS * CROP);
    unsigned *count = list + TRI_LIST;
    const unsigned *tri_rows = tri_rows_all + (size_t)n * F;
    const unsigned row_last = (unsigned)(row0 + STRIP_ROWS - 1);
    if (tid == 0) *count = 0u;
    __syncthreads();
    auto flush = [&]() {   // (called by all threads, after a barrier that follows the last append)
      const int nl = (int)*count;
      // software-pipelined like the full walk below: vertices of entry k+1 and indices of entry k+2 in flight
      auto entry = [&](int k) { return k < nl ? load_idx((int)list[k]) : Tri{-1, -1, -1}; };
      Tri cur = entry(tid), nxt = entry(tid + NT);
      bool ok = valid(cur);
      float4 v0 = ok ? clip[cur.i0] : zero4, v1 = ok ? clip[cur.i1] : zero4, v2 = ok ? clip[cur.i2] : zero4;
      for (int k = tid; k < nl; k += NT) {
        const bool okn = valid(nxt);
        const float4 w0 = okn ? clip[nxt.i0] : zero4, w1 = okn ? clip[nxt.i1] : zero4, w2 = okn ? clip[nxt.i2] : zero4;
        const Tri nn = entry(k + 2 * NT);
        if (ok) process((int)list[k], v0, v1, v2);
        v0 = w0; v1 = w1; v2 = w2; ok = okn; nxt = nn;
      }
    };
    auto load_slab = [&](int base, unsigned (&r)[4]) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int f = base + u * NT + tid;
        r[u] = f < F ? tri_rows[f] : TRI_ROWS_EMPTY;
      }
    };
    unsigned r[4], rn[4];
    load_slab(0, r);
    for (int base = 0; base < F; base += 4 * NT) {
      load_slab(base + 4 * NT, rn);   // the next slab's ranges travel under this slab's appends and barriers
#pragma unroll
      for (int u = 0; u < 4; u++)
        if ((r[u] & 0xffffu) <= row_last && (r[u] >> 16) >= (unsigned)row0) list[atomicAdd(count, 1u)] = (unsigned)(base + u * NT + tid);
      __syncthreads();
      // every thread reads the same count between two barriers (the second keeps the next slab's appends behind the reads): a uniform
      // decision without a workgroup reduction
      const bool full = *count > (unsigned)(TRI_LIST - 4 * NT) && base + 4 * NT < F;
      __syncthreads();
      if (full) {
        flush();
        __syncthreads();
        if (tid == 0) *count = 0u;
        __syncthreads();
      }
#pragma unroll
      for (int u = 0; u < 4; u++) r[u] = rn[u];
    }
    flush();
  } else {
    Tri cur = load_idx(tid), nxt = load_idx(tid + NT);
    bool ok = valid(cur);
    float4 v0 = ok ? clip[cur.i0] : zero4, v1 = ok ? clip[cur.i1] : zero4, v2 = ok ? clip[cur.i2] : zero4;
    for (int f = tid; f < F; f += NT) {
      const bool okn = valid(nxt);
      const float4 w0 = okn ? clip[nxt.i0] : zero4, w1 = okn ? clip[nxt.i1] : zero4, w2 = okn ? clip[nxt.i2] : zero4;
      const Tri nn = load_idx(f + 2 * NT);
      if (ok) process(f, v0, v1, v2);
      v0 = w0; v1 = w1; v2 = w2; ok = okn; nxt = nn;
    }
  }
  __syncthreads();

  // shading pass: one output pixel per lane-iteration, coalesced along x
  const PoseRec &rec = recs[n];
  const float tpx = rec.pose[12], tpy = rec.pose[13], tpz = rec.pose[14];
  const float xs = 2.f / (float)CROP, xo = 1.f / (float)CROP - 1.f;
  for (int i = tid; i < STRIP_ROWS * CROP; i += NT) {
    int ly = i / CROP, px = i - ly * CROP, py = row0 + ly;
    unsigned color = ~(unsigned)(zbuf[i] & 0xFFFFFFFFull);
    int triIdx = (int)color - 1;
    float b0 = 0, b1 = 0, zw = 0, idf = 0;
    float xyz0 = 0, xyz1 = 0, xyz2 = 0, uu = 0, vv = 0, dif = 0;
    if (triIdx >= 0 && triIdx < F) {
      int vi0 = faces[triIdx * 3], vi1 = faces[triIdx * 3 + 1], vi2 = faces[triIdx * 3 + 2];
      float4 p0 = clip[vi0], p1 = clip[vi1], p2 = clip[vi2];
      float fx = mad<FMAD>(xs, (float)px, xo), fy = mad<FMAD>(xs, (float)py, xo);
      float p0x = mad<FMAD>(-fx, p0.w, p0.x), p0y = mad<FMAD>(-fy, p0.w, p0.y);
      float p1x = mad<FMAD>(-fx, p1.w, p1.x), p1y = mad<FMAD>(-fy, p1.w, p1.y);
      float p2x = mad<FMAD>(-fx, p2.w, p2.x), p2y = mad<FMAD>(-fy, p2.w, p2.y);
      float a0 = diffprod<FMAD>(p1x, p2y, p1y, p2x), a1 = diffprod<FMAD>(p2x, p0y, p2y, p0x), a2 = diffprod<FMAD>(p0x, p1y, p0y, p1x);
      float iw = 1.f / (a0 + a1 + a2);
      b0 = a0 * iw; b1 = a1 * iw;
      float z = dot3<FMAD>(p0.z, a0, p1.z, a1, p2.z, a2), w = dot3<FMAD>(p0.w, a0, p1.w, a1, p2.w, a2);
      zw = z / w;
      b0 = clampf(b0, 0.f, 1.f); b1 = clampf(b1, 0.f, 1.f);
      if (b0 != b0) b0 = 0.f;
      if (b1 != b1) b1 = 0.f;
      zw = fmaxf(fminf(zw, 1.f), -1.f);
      idf = (float)(triIdx + 1);
      float b2 = 1.f - b0 - b1;
      float4 q0 = attr[vi0], q1 = attr[vi1], q2 = attr[vi2];
      xyz0 = dot3<FMAD>(b0, q0.x, b1, q1.x, b2, q2.x);
      xyz1 = dot3<FMAD>(b0, q0.y, b1, q1.y, b2, q2.y);
      xyz2 = dot3<FMAD>(b0, q0.z, b1, q1.z, b2, q2.z);
      dif = dot3<FMAD>(b0, q0.w, b1, q1.w, b2, q2.w);
      uu = dot3<FMAD>(b0, uvs[vi0 * 2], b1, uvs[vi1 * 2], b2, uvs[vi2 * 2]);
      vv = dot3<FMAD>(b0, uvs[vi0 * 2 + 1], b1, uvs[vi1 * 2 + 1], b2, uvs[vi2 * 2 + 1]);
    }
    if (tri_id_dbg) tri_id_dbg[((size_t)n * CROP + py) * CROP + px] = (int)color;
    if (rast_dbg) {
      float *ro = rast_dbg + (((size_t)n * CROP + py) * CROP + px) * 4;
      ro[0] = b0; ro[1] = b1; ro[2] = zw; ro[3] = idf;
    }
    float o[6];
    float fg = clampf(idf, 0, 1);
    if (fg > 0.0f) {
      // bilinear texture fetch, wrap addressing, texel centre u*w - 0.5, texture value = u8 * (1/255)
      float u = uu - floorf(uu), v = vv - floorf(vv);
      u = mad<FMAD>(u, (float)TW, -0.5f); v = mad<FMAD>(v, (float)TH, -0.5f);
      int iu0 = (int)floorf(u), iv0 = (int)floorf(v);
      int iu1 = iu0 + 1, iv1 = iv0 + 1;
      u -= (float)iu0; v -= (float)iv0;
      if (iu0 < 0) iu0 += TW;
      if (iv0 < 0) iv0 += TH;
      if (iu1 >= TW) iu1 -= TW;
      if (iv1 >= TH) iv1 -= TH;
      const float sc = 1.0f / 255.0f;
      const uint8_t *t00 = tex + (iu0 + TW * iv0) * 3, *t10 = tex + (iu1 + TW * iv0) * 3;
      const uint8_t *t01 = tex + (iu0 + TW * iv1) * 3, *t11 = tex + (iu1 + TW * iv1) * 3;
      float shade = mad<FMAD>(dif, 0.5f, 0.8f);
      for (int c = 0; c < 3; c++) {
        float a00 = (float)t00[c] * sc, a10 = (float)t10[c] * sc, a01 = (float)t01[c] * sc, a11 = (float)t11[c] * sc;
        float top = mad<FMAD>(u, a10 - a00, a00), bot = mad<FMAD>(u, a11 - a01, a01);
        float rgb = mad<FMAD>(v, bot - top, top);
        float q = rgb * shade * fg;
        o[c] = clampf(clampf(q, 0, 1), 0.0f, 1.0f);
      }
    } else {
      o[0] = o[1] = o[2] = 0.0f;
    }
    {
      bool invalid = xyz2 < FP_MIN_DEPTH;
      float q0 = (xyz0 - tpx) / downscale, q1 = (xyz1 - tpy) / downscale, q2 = (xyz2 - tpz) / downscale;
      o[3] = (fabsf(q0) > FP_MAX_DEPTH || invalid) ? 0.0f : q0;
      o[4] = (fabsf(q1) > FP_MAX_DEPTH || invalid) ? 0.0f : q1;
      o[5] = (fabsf(q2) > FP_MAX_DEPTH || invalid) ? 0.0f : q2;
    }
    size_t opix = ((size_t)n * CROP + (CROP - 1 - py)) * CROP + px;  // vertical flip
    if (MODE == OUT_F32X6) {
      float *d = reinterpret_cast<float *>(out_all) + opix * 6;
      float2 *d2 = reinterpret_cast<float2 *>(d);
      d2[0] = make_float2(o[0], o[1]); d2[1] = make_float2(o[2], o[3]); d2[2] = make_float2(o[4], o[5]);
    } else {
      reinterpret_cast<uint4 *>(out_all)[s2d_index((size_t)n, CROP - 1 - py, px)] = pack6<MODE>(o);
    }
  }
}

// the row ranges of the launch in progress (launch_raster_shade sets it; the launchers below are plain host code of the same call)
static thread_local const unsigned *t_tri_rows = nullptr;
static size_t tri_list_lds() { return t_tri_rows ? (size_t)TRI_LIST * 4 + 16 : 0; }

#ifdef FP_TEST_HOOKS
static int g_tri_rows_tall = 0;  // A/B (test build): row ranges for the tall strips of large batches too
void set_tri_rows_tall(int v) { g_tri_rows_tall = v; }
#else
static constexpr int g_tri_rows_tall = 0;
#endif
#ifdef FP_TEST_HOOKS
static int g_strip_threads = 0;  // A/B (test build): threads per 8-row strip workgroup for small batches, 0 = by batch size
void set_raster_strip_threads(int t) { g_strip_threads = t; }
#else
static constexpr int g_strip_threads = 0;
#endif

template <int MODE, int STRIP_ROWS, bool FMAD>
static void launch_raster_shade_t(hipStream_t s, const DeviceMesh &m, const PoseRec *recs, int N, const float4 *clip,
                                  const float4 *attr, void *out, int32_t *tri_id_dbg, float *rast_dbg) {
  size_t lds = (size_t)STRIP_ROWS * CROP * sizeof(unsigned long long) + tri_list_lds();
  dim3 grid(CROP / STRIP_ROWS, N), block(256);
  float downscale = m.diameter / 2;
  // small batches (a few objects, a 32-hypothesis shard of a strong-scaled Register): the launch is a latency chain of F / NT dependent
  // triangle iterations per strip, not throughput -- 16 waves per strip while two such workgroups per CU hold the whole grid
  // (20 strips x N <= 512), 8 waves up to where 8-row strips are used at all (N < 48).  [r4] tools/profile_shard.py: N = 12: 55 -> 29 us per
  // launch; from N ~ 32 on the launch is bound by the gather rate of the 20x redundant triangle set-up instead (64 us at any width)
  if (STRIP_ROWS == 8 && MODE != OUT_F32X6 && !tri_id_dbg && !rast_dbg && g_strip_threads != 256) {
    if (g_strip_threads == 1024 || (g_strip_threads == 0 && N <= 25)) {
      hipLaunchKernelGGL((raster_shade_kernel<MODE, STRIP_ROWS, 1024, FMAD>), grid, dim3(1024), lds, s, m.faces, m.F, m.V, m.uvs,
                         m.tex, m.TH, m.TW, downscale, recs, clip, attr, out, tri_id_dbg, rast_dbg, t_tri_rows);
      return;
    }
    if (g_strip_threads == 512 || (g_strip_threads == 0 && N < 48)) {
      hipLaunchKernelGGL((raster_shade_kernel<MODE, STRIP_ROWS, 512, FMAD>), grid, dim3(512), lds, s, m.faces, m.F, m.V, m.uvs,
                         m.tex, m.TH, m.TW, downscale, recs, clip, attr, out, tri_id_dbg, rast_dbg, t_tri_rows);
      return;
    }
  }
  if constexpr (STRIP_ROWS >= 40) {   // (test-build strip heights: z-buffer + list exceed the 64 KB a kernel gets without asking)
    static PerDeviceOnce attr_once;
    attr_once.run([] {
      (void)hipFuncSetAttribute((const void *)raster_shade_kernel<MODE, STRIP_ROWS, 256, FMAD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((size_t)STRIP_ROWS * CROP * sizeof(unsigned long long) + (size_t)TRI_LIST * 4 + 16));
    });
  }
  hipLaunchKernelGGL((raster_shade_kernel<MODE, STRIP_ROWS, 256, FMAD>), grid, block, lds, s, m.faces, m.F, m.V, m.uvs, m.tex,
                     m.TH, m.TW, downscale, recs, clip, attr, out, tri_id_dbg, rast_dbg, t_tri_rows);
}

#ifdef FP_TEST_HOOKS
static int g_strip_rows_override = 0;  // A/B: tools/ab_raster_strips.py (test build only)
void set_raster_strip_rows(int r) { g_strip_rows_override = r; }
#else
static constexpr int g_strip_rows_override = 0;
#endif

// tall strips with 1024-thread workgroups: every strip walks ALL triangles (setup + cull), so 2 strips of 80 rows do a
// quarter of the redundant setup of 8 strips of 20 (0.40 -> 0.21 ms per Register at N = 252); 102 KB of LDS = one
// workgroup per CU, hence 16 waves per workgroup.  A/B codes for set_raster_strip_rows: 1080 / 1040 / 1020
template <int MODE, int STRIP_ROWS, bool FMAD>
static void launch_raster_tall(hipStream_t s, const DeviceMesh &m, const PoseRec *recs, int N, const float4 *clip,
                               const float4 *attr, void *out) {
  const size_t lds_max = (size_t)STRIP_ROWS * CROP * sizeof(unsigned long long) + (size_t)TRI_LIST * 4 + 16;
  size_t lds = (size_t)STRIP_ROWS * CROP * sizeof(unsigned long long) + tri_list_lds();
  // once per instantiation and device: opt in to > 64 KB of dynamic LDS
  static PerDeviceOnce attr_once;
  attr_once.run([lds_max] {
    (void)hipFuncSetAttribute((const void *)raster_shade_kernel<MODE, STRIP_ROWS, 1024, FMAD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_max);
  });
  hipLaunchKernelGGL((raster_shade_kernel<MODE, STRIP_ROWS, 1024, FMAD>), dim3(CROP / STRIP_ROWS, N), dim3(1024), lds, s, m.faces, m.F,
                     m.V, m.uvs, m.tex, m.TH, m.TW, m.diameter / 2, recs, clip, attr, out, nullptr, nullptr, t_tri_rows);
}

// Row ranges pay where a crop is cut into many short strips (Track: 40, small batches: 20 or 8); the two 80-row strips of a full
// Register batch meet half of the triangles each, and the extra launch + list passes cost what the skipped set-ups save
// (tools/mesh_size_sweep.py: N = 252, 20 k triangles, 412 -> 396 us of rasteriser + the range kernel).
static int strip_rows_for(int N) { return g_strip_rows_override ? g_strip_rows_override : (N >= 100 ? 1080 : (N >= 48 ? 20 : 8)); }
bool raster_wants_tri_rows(int N) { return strip_rows_for(N) < 1000 || g_tri_rows_tall; }

template <int MODE, bool FMAD>
static void launch_raster_mode(hipStream_t s, const DeviceMesh &m, const PoseRec *recs, int N, const float4 *clip,
                               const float4 *attr, void *out, int32_t *tri_id_dbg, float *rast_dbg) {
  int rows = strip_rows_for(N);  // (tools/ab_raster_strips.py: 8-row strips win up to ~40 hypotheses)
  if (rows > 1000 && MODE != OUT_F32X6 && !tri_id_dbg && !rast_dbg) {
    if (rows == 1080) { launch_raster_tall<MODE, 80, FMAD>(s, m, recs, N, clip, attr, out); return; }
#ifdef FP_TEST_HOOKS
    if (rows == 1040) launch_raster_tall<MODE, 40, FMAD>(s, m, recs, N, clip, attr, out);
    else launch_raster_tall<MODE, 20, FMAD>(s, m, recs, N, clip, attr, out);
    return;
#endif
  }
  if (rows > 1000) rows = 20;
#ifdef FP_TEST_HOOKS
  if (rows == 40) { launch_raster_shade_t<MODE, 40, FMAD>(s, m, recs, N, clip, attr, out, tri_id_dbg, rast_dbg); return; }
#endif
  if (rows == 20) { launch_raster_shade_t<MODE, 20, FMAD>(s, m, recs, N, clip, attr, out, tri_id_dbg, rast_dbg); return; }
  // one or two hypotheses (Track): 4-row strips with 1024 threads -- the shading pass covers the strip in ONE iteration
  // (640 pixels; 8 rows = 1280 pixels took two, the second a quarter full) and a strip meets half as many triangles
  if ((rows == 4 || (rows == 8 && N <= 2 && !g_strip_rows_override)) && MODE != OUT_F32X6 && !tri_id_dbg && !rast_dbg) {
    const size_t lds = (size_t)4 * CROP * sizeof(unsigned long long) + tri_list_lds();
    hipLaunchKernelGGL((raster_shade_kernel<MODE, 4, 1024, FMAD>), dim3(CROP / 4, N), dim3(1024), lds, s, m.faces, m.F, m.V, m.uvs,
                       m.tex, m.TH, m.TW, m.diameter / 2, recs, clip, attr, out, tri_id_dbg, rast_dbg, t_tri_rows);
    return;
  }
  launch_raster_shade_t<MODE, 8, FMAD>(s, m, recs, N, clip, attr, out, tri_id_dbg, rast_dbg);
}

void launch_raster_shade(hipStream_t s, const DeviceMesh &m, const PoseRec *recs, int N, const float4 *clip,
                         const float4 *attr, OutMode mode, void *out, int32_t *tri_id_dbg, float *rast_dbg, bool fmad,
                         const unsigned *tri_rows) {
  t_tri_rows = tri_rows;
#define FP_RASTER_MODE(MODE)                                                                                      \
  do {                                                                                                            \
    if (fmad) launch_raster_mode<MODE, true>(s, m, recs, N, clip, attr, out, tri_id_dbg, rast_dbg);               \
    else launch_raster_mode<MODE, false>(s, m, recs, N, clip, attr, out, tri_id_dbg, rast_dbg);                   \
  } while (0)
  if (mode == OUT_F32X6) FP_RASTER_MODE(OUT_F32X6);
  else if (mode == OUT_BF16X8) FP_RASTER_MODE(OUT_BF16X8);
  else FP_RASTER_MODE(OUT_F16X8);
#undef FP_RASTER_MODE
}

// ---------------------------------------------------------------------------------------------
// GuessTranslation on the device (foundationpose_sampling.cpp:250-298): no 1.2 MB read-back, no host median, no
// synchronisation in the middle of Register.
//   sampler_scan_kernel : bounding box of mask > 0 and a compacted list of the filtered depths with mask > 0 and
//                         depth >= min_depth (order arbitrary: only order statistics are taken from it)
//   sampler_pose_kernel : one workgroup; exact median by a 4-pass radix select on the float bit patterns (positive
//                         floats order like their bits; even counts average the two middle values in double like the
//                         reference), centre = K^-1 (uc, vc, 1) zc in the host code's float expression order, then
//                         poses[i] = grid[first + i] with that translation.  state[6] = 0 ok / 1 empty mask / 2 no depth.
// state layout (ints): 0 umin, 1 umax, 2 vmin, 3 vmax, 4 count, 5 (unused), 6 status
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sampler_scan_kernel(const float *__restrict__ depth, const uint8_t *__restrict__ mask,
                                                           int H, int W, float min_depth, int *__restrict__ state,
                                                           float *__restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool in = i < H * W;
  const bool m = in && mask[i] > 0;
  const int v = in ? i / W : 0, u = in ? i - v * W : 0;
  if (__any(m)) {
    int umin = m ? u : 0x7fffffff, umax = m ? u : -1, vmin = m ? v : 0x7fffffff, vmax = m ? v : -1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      umin = min(umin, __shfl_xor(umin, o)); umax = max(umax, __shfl_xor(umax, o));
      vmin = min(vmin, __shfl_xor(vmin, o)); vmax = max(vmax, __shfl_xor(vmax, o));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&state[0], umin); atomicMax(&state[1], umax);
      atomicMin(&state[2], vmin); atomicMax(&state[3], vmax);
    }
  }
  const float d = m ? depth[i] : 0.f;
  const bool valid = m && d >= min_depth;
  const unsigned long long ball = __ballot(valid);
  if (ball) {
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(&state[4], __popcll(ball));
    base = __shfl(base, 0);
    if (valid) vals[base + __popcll(ball & ((1ull << lane) - 1ull))] = d;
  }
}

// k-th smallest (0-based) of keys[0..n): 4 passes over 8-bit digits, most significant first
__device__ unsigned radix_select(const unsigned *keys, int n, int k, unsigned *hist /* LDS [256] */, unsigned *sh /* LDS [2] */) {
  unsigned prefix = 0, pmask = 0;
  for (int pass = 3; pass >= 0; pass--) {
    for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const int shift = pass * 8;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      unsigned key = keys[i];
      if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned acc = 0, b = 0;
      for (; b < 256; b++) {
        if (acc + hist[b] > (unsigned)k) break;
        acc += hist[b];
      }
      sh[0] = b; sh[1] = acc;
    }
    __syncthreads();
    prefix |= sh[0] << shift;
    pmask |= 255u << shift;
    k -= (int)sh[1];
    __syncthreads();
  }
  return prefix;
}

__global__ __launch_bounds__(1024) void sampler_pose_kernel(int *__restrict__ state, const float *__restrict__ vals, K9 K,
                                                            const float *__restrict__ grid, int first, int N,
                                                            float *__restrict__ poses) {
  __shared__ unsigned hist[256];
  __shared__ unsigned sh[2];
  __shared__ float center[3];
  const int n = state[4];
  const int umin = state[0], umax = state[1], vmin = state[2], vmax = state[3];
  const bool bad = umax < 0 || n <= 0;  // uniform over the workgroup
  const unsigned *keys = reinterpret_cast<const unsigned *>(vals);
  unsigned hi_bits = 0, lo_bits = 0;
  if (!bad) {
    hi_bits = radix_select(keys, n, n / 2, hist, sh);
    lo_bits = hi_bits;
    if ((n & 1) == 0) lo_bits = radix_select(keys, n, n / 2 - 1, hist, sh);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // leave the scan state ready for the next frame (the status word is read back by the host after this launch)
    state[0] = 0x7fffffff; state[1] = -1; state[2] = 0x7fffffff; state[3] = -1; state[4] = 0;
    state[6] = bad ? (umax < 0 ? 1 : 2) : 0;
  }
  if (bad) return;
  if (threadIdx.x == 0) {
    const float hi = __uint_as_float(hi_bits), lo = __uint_as_float(lo_bits);
    const float zc = (n & 1) ? hi : (float)(((double)lo + (double)hi) / 2.0);
    const float uc = (float)((umin + umax) / 2.0), vc = (float)((vmin + vmax) / 2.0);
    const float *m = K.k;
    float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    float det = m[0] * c00 + m[1] * c01 + m[2] * c02, id = 1.0f / det;
    float Ki[9] = {c00 * id, (m[2] * m[7] - m[1] * m[8]) * id, (m[1] * m[5] - m[2] * m[4]) * id,
                   c01 * id, (m[0] * m[8] - m[2] * m[6]) * id, (m[2] * m[3] - m[0] * m[5]) * id,
                   c02 * id, (m[1] * m[6] - m[0] * m[7]) * id, (m[0] * m[4] - m[1] * m[3]) * id};
    for (int r = 0; r < 3; r++) {
      float sacc = Ki[r * 3] * uc;
      sacc = sacc + Ki[r * 3 + 1] * vc;
      sacc = sacc + Ki[r * 3 + 2] * 1.0f;
      center[r] = sacc * zc;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N * 16; i += blockDim.x) {
    const int e = i & 15;
    poses[i] = (e >= 12 && e < 15) ? center[e - 12] : grid[(size_t)first * 16 + i];
  }
}

void launch_sampler(hipStream_t s, const float *filtered_depth, const uint8_t *mask_dev, int H, int W, float min_depth,
                    const float *K9_host, const float *grid_dev, int first, int N, int *state, float *vals, float *poses) {
  // `state` was initialised when it was allocated and every sampler_pose_kernel launch resets it for the next frame
  hipLaunchKernelGGL(sampler_scan_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, filtered_depth, mask_dev, H, W, min_depth,
                     state, vals);
  K9 K;
  for (int i = 0; i < 9; i++) K.k[i] = K9_host[i];
  hipLaunchKernelGGL(sampler_pose_kernel, dim3(1), dim3(1024), 0, s, state, vals, K, grid_dev, first, N, poses);
}

// ---------------------------------------------------------------------------------------------
// crop / warp of the observed RGB-D frame
// ---------------------------------------------------------------------------------------------

// one output pixel i (0 .. 160*160) of the observed crop of hypothesis n under the record `rec`
template <int MODE>
__device__ __forceinline__ void crop_body(const FrameRef *__restrict__ frame, int H, int W, float fx, float fy, float cx, float cy,
                                          const PoseRec &rec, float downscale, void *__restrict__ out_all, int n, int i) {
  const uint8_t *__restrict__ rgb = frame->rgb;
  const float *__restrict__ depth = frame->depth;
  // (whole frames: P = W and the window tests are always true; a packed window never lacks a pixel this function reads -- the
  // host-side window estimate has a margin and tests/test_nn_gpu.py compares against whole frames -- the tests only make an
  // estimate that were wrong read a zero instead of foreign memory)
  const int P = frame->pitch > 0 ? frame->pitch : W;
  const int wx0 = frame->wx0, wy0 = frame->wy0, wx1 = frame->wx1, wy1 = frame->wy1;
  const int y = i / CROP, x = i - y * CROP;
  float sxf = rec.m0 * (float)x + rec.m2, syf = rec.m4 * (float)y + rec.m5;
  // degenerate hypotheses (tz ~ 0: a crop window of 1e13 pixels) give source coordinates far outside any image, or NaN; clamp them
  // to +-2^24 BEFORE the float->int conversions so that x0 + 1 / the pointer arithmetic below cannot overflow (values inside
  // the image are untouched, NaN becomes -2^24 = outside).  A 1e-12 m hypothesis faulted the GPU here before (round 3).
  sxf = fminf(fmaxf(sxf, -16777216.0f), 16777216.0f);
  syf = fminf(fmaxf(syf, -16777216.0f), 16777216.0f);
  float o[6];
  {
    int x0 = (int)floorf(sxf), y0 = (int)floorf(syf);
    float ax = sxf - (float)x0, ay = syf - (float)y0;
    bool vx0 = x0 >= 0 && x0 < W && x0 >= wx0 && x0 < wx1, vx1 = x0 + 1 >= 0 && x0 + 1 < W && x0 + 1 >= wx0 && x0 + 1 < wx1;
    bool vy0 = y0 >= 0 && y0 < H && y0 >= wy0 && y0 < wy1, vy1 = y0 + 1 >= 0 && y0 + 1 < H && y0 + 1 >= wy0 && y0 + 1 < wy1;
    const uint8_t *r0 = rgb + ((long long)y0 * P + x0) * 3, *r1 = rgb + ((long long)(y0 + 1) * P + x0) * 3;
    float w00 = (1.0f - ax) * (1.0f - ay), w10 = ax * (1.0f - ay), w01 = (1.0f - ax) * ay, w11 = ax * ay;
    for (int ch = 0; ch < 3; ch++) {
      float p00 = (vy0 && vx0) ? (float)r0[ch] : 0.0f, p10 = (vy0 && vx1) ? (float)r0[3 + ch] : 0.0f;
      float p01 = (vy1 && vx0) ? (float)r1[ch] : 0.0f, p11 = (vy1 && vx1) ? (float)r1[3 + ch] : 0.0f;
      float val = p00 * w00 + p10 * w10 + p01 * w01 + p11 * w11;
      float q = rintf(val);
      q = q < 0 ? 0 : (q > 255 ? 255 : q);
      o[ch] = q * (1.0f / 255.0f);
    }
  }
  {
    int xn = (int)floorf(sxf + 0.5f), yn = (int)floorf(syf + 0.5f);
    float p0 = 0, p1 = 0, p2 = 0;
    if (xn >= 0 && xn < W && yn >= 0 && yn < H && xn >= wx0 && xn < wx1 && yn >= wy0 && yn < wy1) {
      float d = depth[(long long)yn * P + xn];
      if (!(d < 0.001f)) { p0 = ((float)xn - cx) * d / fx; p1 = ((float)yn - cy) * d / fy; p2 = d; }
    }
    bool invalid = p2 < FP_MIN_DEPTH;
    float q0 = (p0 - rec.pose[12]) / downscale, q1 = (p1 - rec.pose[13]) / downscale, q2 = (p2 - rec.pose[14]) / downscale;
    o[3] = (fabsf(q0) > FP_MAX_DEPTH || invalid) ? 0.0f : q0;
    o[4] = (fabsf(q1) > FP_MAX_DEPTH || invalid) ? 0.0f : q1;
    o[5] = (fabsf(q2) > FP_MAX_DEPTH || invalid) ? 0.0f : q2;
  }
  size_t opix = (size_t)n * CROP * CROP + i;
  if (MODE == OUT_F32X6) {
    float2 *d2 = reinterpret_cast<float2 *>(reinterpret_cast<float *>(out_all) + opix * 6);
    d2[0] = make_float2(o[0], o[1]); d2[1] = make_float2(o[2], o[3]); d2[2] = make_float2(o[4], o[5]);
  } else {
    reinterpret_cast<uint4 *>(out_all)[s2d_index((size_t)n, y, x)] = pack6<MODE>(o);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void crop_kernel(const FrameRef *__restrict__ frame,
                                                   int H, int W, float fx, float fy, float cx, float cy,
                                                   const PoseRec *__restrict__ recs, float downscale,
                                                   void *__restrict__ out_all) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= CROP * CROP) return;
  crop_body<MODE>(frame, H, W, fx, fy, cx, cy, recs[n], downscale, out_all, n, i);
}

// Tiny batches (Track): pose set-up + vertex stage + observed-crop warp in ONE launch.  Blocks [0, nvb) of a hypothesis are the
// vertex kernel's; blocks [nvb, nvb + 100) warp the crop -- they cannot wait for the record a vertex block writes, so they compute
// it themselves (make_pose_rec: the same function on the same inputs, identical values).  One launch-floor kernel less per Track.
template <bool FMAD, int MODE>
__global__ __launch_bounds__(256) void vertex_crop_kernel(const float *__restrict__ verts, const float *__restrict__ normals, int V, int nvb,
                                                          PoseRec *__restrict__ recs, float4 *__restrict__ clip, float4 *__restrict__ attr,
                                                          const PoseSetupArgs sa, const FrameRef *__restrict__ frame, int n_crop,
                                                          void *__restrict__ out_b, const int32_t *__restrict__ faces, int F,
                                                          unsigned *__restrict__ tri_rows) {
  const int n = blockIdx.y;
  PoseRec own;
  constexpr int NCB = (CROP * CROP + 255) / 256;
  if ((int)blockIdx.x >= nvb + NCB) {
    // [r4] row-range blocks (present when tri_rows != null): the range of a triangle needs its three clip positions, which the
    // vertex blocks of THIS launch are still writing -- so they are recomputed here from the model vertices (clip_of: the
    // same expressions on the same record, identical values), and the separate tri_rows_kernel launch disappears
    const int f = ((int)blockIdx.x - nvb - NCB) * 256 + threadIdx.x;
    if (f >= F) return;
    unsigned out = TRI_ROWS_EMPTY;
    const int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
    if ((unsigned)i0 < (unsigned)V && (unsigned)i1 < (unsigned)V && (unsigned)i2 < (unsigned)V) {
      make_pose_rec(sa.poses, n, sa.K, sa.img_h, sa.img_w, sa.crop_ratio, sa.diameter, own);
      const float4 c0 = clip_of<FMAD>(own, verts[i0 * 3], verts[i0 * 3 + 1], verts[i0 * 3 + 2]);
      const float4 c1 = clip_of<FMAD>(own, verts[i1 * 3], verts[i1 * 3 + 1], verts[i1 * 3 + 2]);
      const float4 c2 = clip_of<FMAD>(own, verts[i2 * 3], verts[i2 * 3 + 1], verts[i2 * 3 + 2]);
      out = tri_rows_of(c0, c1, c2);
    }
    tri_rows[(size_t)n * F + f] = out;
    return;
  }
  if ((int)blockIdx.x < nvb) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    make_pose_rec(sa.poses, n, sa.K, sa.img_h, sa.img_w, sa.crop_ratio, sa.diameter, own);
    if (v == 0) recs[n] = own;
    vertex_body<FMAD>(verts, normals, V, v, n, own, clip, attr, nullptr, 0ull);
    return;
  }
  const int i = ((int)blockIdx.x - nvb) * 256 + threadIdx.x;
  if (n >= n_crop || i >= CROP * CROP) return;
  make_pose_rec(sa.poses, n, sa.K, sa.img_h, sa.img_w, sa.crop_ratio, sa.diameter, own);
  crop_body<MODE>(frame, sa.img_h, sa.img_w, sa.K.k[0], sa.K.k[4], sa.K.k[2], sa.K.k[5], own, sa.diameter / 2, out_b, n, i);
}

// returns false when the combination is not instantiated (fp32 blobs): the caller then launches the two kernels
bool launch_setup_vertex_crop(hipStream_t s, const DeviceMesh &m, const float *poses_dev, int N, const float *K9_host, int img_h, int img_w,
                              float crop_ratio, float diameter, PoseRec *recs, float4 *clip, float4 *attr, bool fmad, const FrameRef *frame,
                              int n_crop, OutMode mode, void *out_b, unsigned *tri_rows) {
  if (mode == OUT_F32X6) return false;
  PoseSetupArgs sa;
  sa.poses = poses_dev;
  for (int i = 0; i < 9; i++) sa.K.k[i] = K9_host[i];
  sa.img_h = img_h; sa.img_w = img_w; sa.crop_ratio = crop_ratio; sa.diameter = diameter;
  const int nvb = (m.V + 255) / 256;
  const int ntb = tri_rows && m.F > 0 ? (m.F + 255) / 256 : 0;   // row-range blocks (the rasteriser's tri_rows of these poses)
  const dim3 grid(nvb + (CROP * CROP + 255) / 256 + ntb, N);
#define FP_VC(FM_, MD_) hipLaunchKernelGGL((vertex_crop_kernel<FM_, MD_>), grid, dim3(256), 0, s, m.verts, m.normals, m.V, nvb, recs, clip, attr, sa, frame, n_crop, out_b, m.faces, m.F, tri_rows)
  if (mode == OUT_BF16X8) { if (fmad) FP_VC(true, OUT_BF16X8); else FP_VC(false, OUT_BF16X8); }
  else { if (fmad) FP_VC(true, OUT_F16X8); else FP_VC(false, OUT_F16X8); }
#undef FP_VC
  return true;
}

void launch_crop(hipStream_t s, const FrameRef *frame, int H, int W, const float *K, const PoseRec *recs,
                 int N, float diameter, OutMode mode, void *out) {
  dim3 grid((CROP * CROP + 255) / 256, N), block(256);
  float downscale = diameter / 2;
  if (mode == OUT_F32X6)
    hipLaunchKernelGGL(crop_kernel<OUT_F32X6>, grid, block, 0, s, frame, H, W, K[0], K[4], K[2], K[5], recs,
                       downscale, out);
  else if (mode == OUT_BF16X8)
    hipLaunchKernelGGL(crop_kernel<OUT_BF16X8>, grid, block, 0, s, frame, H, W, K[0], K[4], K[2], K[5], recs,
                       downscale, out);
  else
    hipLaunchKernelGGL(crop_kernel<OUT_F16X8>, grid, block, 0, s, frame, H, W, K[0], K[4], K[2], K[5], recs,
                       downscale, out);
}

// ---------------------------------------------------------------------------------------------
// depth -> xyz, erode, bilateral
// ---------------------------------------------------------------------------------------------

__global__ void depth_to_xyz_kernel(const float *__restrict__ depth, int H, int W, float fx, float fy, float cx,
                                    float cy, float *__restrict__ xyz) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= H * W) return;
  int r = p / W, c = p - r * W;
  float d = depth[p];
  float x = 0, y = 0, z = 0;
  if (!(d < 0.001f)) { x = ((float)c - cx) * d / fx; y = ((float)r - cy) * d / fy; z = d; }
  xyz[(size_t)p * 3] = x; xyz[(size_t)p * 3 + 1] = y; xyz[(size_t)p * 3 + 2] = z;
}

// [r4] Track's packed crop window (frame record + rgb rows + depth rows, 8-330 KB) from the model's host-pinned block into its device
// block: 16 bytes per lane, 1 KB per wave instruction straight over PCIe.  A copy COMMAND of this size costs ~27 us before the graph
// behind it can start, whatever its size above ~16 KB (tools/track_window_sweep.py); a kernel dispatch costs what a kernel costs.
__global__ __launch_bounds__(256) void window_fetch_kernel(const uint4 *__restrict__ src_host, uint4 *__restrict__ dst, unsigned n16) {
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src_host[i];
}
void launch_window_fetch(hipStream_t s, const void *src_host_mapped, void *dst, size_t bytes) {
  const unsigned n16 = (unsigned)((bytes + 15) / 16);
  const unsigned blocks = std::min(256u, (n16 + 255) / 256);
  hipLaunchKernelGGL(window_fetch_kernel, dim3(blocks), dim3(256), 0, s, (const uint4 *)src_host_mapped, (uint4 *)dst, n16);
}

void launch_depth_to_xyz(hipStream_t s, const float *depth, int H, int W, const float *K, float *xyz) {
  hipLaunchKernelGGL(depth_to_xyz_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, depth, H, W, K[0], K[4], K[2],
                     K[5], xyz);
}

__global__ void erode_kernel(const float *__restrict__ depth, float *__restrict__ out, int H, int W) {
  const int radius = 2;
  const float depth_diff_thres = 0.001f, ratio_thres = 0.8f, zfar = 100.0f;
  int w = blockIdx.x * blockDim.x + threadIdx.x, h = blockIdx.y * blockDim.y + threadIdx.y;
  if (w >= W || h >= H) return;
  float d_ori = depth[h * W + w];
  if (d_ori < 0.1f || d_ori >= zfar) { out[h * W + w] = 0.0f; return; }
  float bad = 0.0f, total = 0.0f;
  for (int u = w - radius; u <= w + radius; u++) {
    if (u < 0 || u >= W) continue;
    for (int v = h - radius; v <= h + radius; v++) {
      if (v < 0 || v >= H) continue;
      float cur = depth[v * W + u];
      total += 1.0f;
      if (cur < 0.1f || cur >= zfar || fabsf(cur - d_ori) > depth_diff_thres) bad += 1.0f;
    }
  }
  out[h * W + w] = ((bad / total) > ratio_thres) ? 0.0f : d_ori;
}

__global__ void bilateral_kernel(const float *__restrict__ depth, float *__restrict__ out, int H, int W) {
  const int radius = 2;
  const float zfar = 100.0f, sigmaD = 2.0f, sigmaR = 100000.0f;
  int w = blockIdx.x * blockDim.x + threadIdx.x, h = blockIdx.y * blockDim.y + threadIdx.y;
  if (w >= W || h >= H) return;
  float mean = 0.0f;
  int nvalid = 0;
  for (int u = w - radius; u <= w + radius; u++) {
    if (u < 0 || u >= W) continue;
    for (int v = h - radius; v <= h + radius; v++) {
      if (v < 0 || v >= H) continue;
      float cur = depth[v * W + u];
      if (cur >= 0.1f && cur < zfar) { nvalid++; mean += cur; }
    }
  }
  if (nvalid == 0) { out[h * W + w] = 0.0f; return; }
  mean /= (float)nvalid;
  float dc = depth[h * W + w], sw = 0.0f, sum = 0.0f;
  for (int u = w - radius; u <= w + radius; u++) {
    if (u < 0 || u >= W) continue;
    for (int v = h - radius; v <= h + radius; v++) {
      if (v < 0 || v >= H) continue;
      float cur = depth[v * W + u];
      if (cur >= 0.1f && cur < zfar && fabsf(cur - mean) < 0.01f) {
        float wgt = expf(-((float)((u - w) * (u - w) + (v - h) * (v - h))) / (2.0f * sigmaD * sigmaD) -
                         (dc - cur) * (dc - cur) / (2.0f * sigmaR * sigmaR));
        sw += wgt;
        sum += wgt * cur;
      }
    }
  }
  out[h * W + w] = (sw > 0.0f && nvalid > 0) ? sum / sw : 0.0f;
}

void launch_erode(hipStream_t s, const float *depth, float *out, int H, int W) {
  hipLaunchKernelGGL(erode_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(64, 4), 0, s, depth, out, H, W);
}
void launch_bilateral(hipStream_t s, const float *depth, float *out, int H, int W) {
  hipLaunchKernelGGL(bilateral_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(64, 4), 0, s, depth, out, H, W);
}

// ---------------------------------------------------------------------------------------------
// pose update, arg-max, packing
// ---------------------------------------------------------------------------------------------

__global__ void pose_update_kernel(float *poses, const float *__restrict__ trans, const float *__restrict__ rot, int N, float diameter,
                                   const float *poses_in, float *extra_out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  pose_update_one(poses, trans, rot, i, diameter, poses_in, extra_out);
}

void launch_pose_update(hipStream_t s, float *poses, const float *trans, const float *rot, int N, float diameter, const float *poses_in,
                        float *extra_out) {
  hipLaunchKernelGGL(pose_update_kernel, dim3((N + 63) / 64), dim3(64), 0, s, poses, trans, rot, N, diameter, poses_in ? poses_in : poses,
                     extra_out);
}

// index[0] = first maximum; when `poses` is given the winner's 4x4 is copied to best_pose so the host needs ONE read-back
__global__ void argmax_kernel(const float *__restrict__ scores, int N, int *__restrict__ index, const float *__restrict__ poses,
                              float *__restrict__ best_pose) {
  __shared__ float sv[256];
  __shared__ int si[256];
  int tid = threadIdx.x;
  __shared__ int any_nan;
  float best = -INFINITY;
  int bi = 0x7FFFFFFF;
  if (tid == 0) any_nan = 0;
  __syncthreads();
  bool bad = false;
  for (int i = tid; i < N; i += 256) {
    float v = scores[i];
    bad |= v != v;
    if (v > best || bi == 0x7FFFFFFF) { best = v; bi = i; }
  }
  if (bad) any_nan = 1;   // same value from every writer
  sv[tid] = best; si[tid] = bi;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) {
      float ov = sv[tid + st]; int oi = si[tid + st];
      if (oi != 0x7FFFFFFF && (si[tid] == 0x7FFFFFFF || ov > sv[tid] || (ov == sv[tid] && oi < si[tid]))) { sv[tid] = ov; si[tid] = oi; }
    }
    __syncthreads();
  }
  const int win = si[0] == 0x7FFFFFFF ? 0 : si[0];
  // NaN scores (a rank of a sharded Register poisoned its rows because its half failed): index -2, reported by the host
  if (tid == 0) *index = any_nan ? -2 : win;
  if (poses && tid < 16) best_pose[tid] = poses[(size_t)win * 16 + tid];
}

void launch_argmax(hipStream_t s, const float *scores, int N, int *index_dev, const float *poses, float *best_pose_dev) {
  hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(256), 0, s, scores, N, index_dev, poses, best_pose_dev);
}

template <int MODE>
__global__ void pack_f32x6_kernel(const float *__restrict__ in, uint4 *__restrict__ out, size_t pixels) {
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pixels) return;
  const float2 *s2 = reinterpret_cast<const float2 *>(in + p * 6);
  float2 a = s2[0], b = s2[1], c = s2[2];
  const float o[6] = {a.x, a.y, b.x, b.y, c.x, c.y};
  size_t n = p / (CROP * CROP);
  int rem = (int)(p - n * (CROP * CROP));
  out[s2d_index(n, rem / CROP, rem % CROP)] = pack6<MODE>(o);
}

void launch_pack_f32x6(hipStream_t s, const float *in, void *out, size_t pixels, OutMode mode) {
  const dim3 grid((unsigned)((pixels + 255) / 256));
  if (mode == OUT_BF16X8) hipLaunchKernelGGL(pack_f32x6_kernel<OUT_BF16X8>, grid, dim3(256), 0, s, in, reinterpret_cast<uint4 *>(out), pixels);
  else hipLaunchKernelGGL(pack_f32x6_kernel<OUT_F16X8>, grid, dim3(256), 0, s, in, reinterpret_cast<uint4 *>(out), pixels);
}

}  // namespace fp
