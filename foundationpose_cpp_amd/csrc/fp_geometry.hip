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

__global__ void pose_setup_kernel(const float *__restrict__ poses, int N, K9 K, int img_h, int img_w,
                                  float crop_ratio, float diameter, PoseRec *__restrict__ recs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  PoseRec rec;
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

// The per-hypothesis record is read through uniform (scalar) loads.
__global__ void vertex_kernel(const float *__restrict__ verts, const float *__restrict__ normals, int V,
                              const PoseRec *__restrict__ recs, float4 *__restrict__ clip, float4 *__restrict__ attr,
                              float4 *__restrict__ dbg) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
#ifdef FP_TEST_HOOKS
  const unsigned long long t_begin = dbg ? wall_clock64() : 0ull;
#endif
  if (v >= V) return;
  const PoseRec &rec = recs[n];
  const float *M = rec.M, *pose = rec.pose;
  float x = verts[v * 3], y = verts[v * 3 + 1], z = verts[v * 3 + 2];
  float tx = M[0] * x + M[4] * y + M[8] * z + M[12];
  float ty = M[1] * x + M[5] * y + M[9] * z + M[13];
  float tz = M[2] * x + M[6] * y + M[10] * z + M[14];
  float tw = M[3] * x + M[7] * y + M[11] * z + M[15];
  float4 c;
  c.x = tx * rec.a00 + tw * rec.a30;
  c.y = ty * rec.a11 + tw * rec.a31;
  c.z = tz;
  c.w = tw;
  float4 a;
  a.x = pose[0] * x + pose[4] * y + pose[8] * z + pose[12];
  a.y = pose[1] * x + pose[5] * y + pose[9] * z + pose[13];
  a.z = pose[2] * x + pose[6] * y + pose[10] * z + pose[14];
  float nx = normals[v * 3], ny = normals[v * 3 + 1], nz = normals[v * 3 + 2];
  float ux = pose[0] * nx + pose[4] * ny + pose[8] * nz;
  float uy = pose[1] * nx + pose[5] * ny + pose[9] * nz;
  float uz = pose[2] * nx + pose[6] * ny + pose[10] * nz;
  float l2 = sqrtf(ux * ux + uy * uy + uz * uz);
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
#endif
}

#ifdef FP_TEST_HOOKS
float4 *g_vertex_dbg = nullptr;  // race hunt (tools/dbg_concurrent3.py): launches with N == 64 fill it
#endif
void launch_vertex(hipStream_t s, const DeviceMesh &m, const PoseRec *recs, int N, float4 *clip, float4 *attr) {
  float4 *dbg = nullptr;
#ifdef FP_TEST_HOOKS
  if (g_vertex_dbg && N == 64) dbg = g_vertex_dbg;
#endif
  hipLaunchKernelGGL(vertex_kernel, dim3((m.V + 255) / 256, N), dim3(256), 0, s, m.verts, m.normals, m.V, recs, clip, attr, dbg);
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
  // fixed-point depth plane (setupTriangle + setupPleq)
  const float zcoef = (float)(CR_DEPTH_MAX - CR_DEPTH_MIN) * 0.5f;
  const float zbias = (float)(unsigned)(CR_DEPTH_MAX + CR_DEPTH_MIN) * 0.5f;
  float zv0 = fmaf(z0 * zcoef, rw0, zbias), zv1 = fmaf(z1 * zcoef, rw1, zbias), zv2 = fmaf(z2 * zcoef, rw2, zbias);
  unsigned plx, ply, plz;
  {
    I2 q0 = {p0.x + (CROP << (CR_SUBPIXEL_LOG2 - 1)) - (1 << (CR_SUBPIXEL_LOG2 - 1)),
             p0.y + (CROP << (CR_SUBPIXEL_LOG2 - 1)) - (1 << (CR_SUBPIXEL_LOG2 - 1))};
    float areaRcp = 1.0f / (float)area;
    float mxz = fmaxf(fmaxf(zv0, zv1), zv2);
    int sh = (__float_as_int(mxz) >> 23) - (127 + 22);
    sh = min(max(sh, 0), 8);
    int t0 = (int)(f32_to_u32_trunc(zv0) >> sh);
    int t1 = (int)((f32_to_u32_trunc(zv1) >> sh) - (unsigned)t0);
    int t2 = (int)((f32_to_u32_trunc(zv2) >> sh) - (unsigned)t0);
    unsigned rcpMant = ((unsigned)__float_as_int(areaRcp) & 0x007FFFFFu) | 0x00800000u;
    int rcpShift = (23 + 127) - (__float_as_int(areaRcp) >> 23);
    long long xc = ((long long)t1 * d2.y - (long long)t2 * d1.y) * (long long)rcpMant;
    long long yc = ((long long)t2 * d1.x - (long long)t1 * d2.x) * (long long)rcpMant;
    plx = (unsigned)(xc >> (rcpShift - (sh + CR_SUBPIXEL_LOG2)));
    ply = (unsigned)(yc >> (rcpShift - (sh + CR_SUBPIXEL_LOG2)));
    int centerX = (q0.x * 2 + imin3(d1.x, d2.x, 0) + imax3(d1.x, d2.x, 0)) >> (CR_SUBPIXEL_LOG2 + 1);
    int centerY = (q0.y * 2 + imin3(d1.y, d2.y, 0) + imax3(d1.y, d2.y, 0)) >> (CR_SUBPIXEL_LOG2 + 1);
    int vcx = q0.x - (int)((unsigned)centerX << CR_SUBPIXEL_LOG2);
    int vcy = q0.y - (int)((unsigned)centerY << CR_SUBPIXEL_LOG2);
    plz = (unsigned)t0 << sh;
    plz -= (unsigned)(((xc >> 13) * vcx + (yc >> 13) * vcy) >> (rcpShift - (sh + 13)));
    plz -= plx * (unsigned)centerX + ply * (unsigned)centerY;
  }
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
      unsigned depth = plx * (unsigned)px + ply * (unsigned)py + plz;
      unsigned long long key = ((unsigned long long)depth << 32) | lowkey;
      atomicMin(&zbuf[(py - row0) * CROP + px], key);
    }
  }
}

__device__ __forceinline__ int clip_poly_plane(float *out, const float *in, int numIn, float v0, float v1, float v2) {
  int numOut = 0;
  if (numIn >= 3) {
    int ai = (numIn - 1) * 2;
    float av = v0 + v1 * in[ai + 0] + v2 * in[ai + 1];
    for (int bi = 0; bi < numIn * 2; bi += 2) {
      float bv = v0 + v1 * in[bi + 0] + v2 * in[bi + 1];
      if (av * bv < 0.0f) {
        float bc = av / (av - bv), ac = 1.0f - bc;
        out[numOut + 0] = in[ai + 0] * ac + in[bi + 0] * bc;
        out[numOut + 1] = in[ai + 1] * ac + in[bi + 1] * bc;
        numOut += 2;
      }
      if (bv >= 0.0f) { out[numOut + 0] = in[bi + 0]; out[numOut + 1] = in[bi + 1]; numOut += 2; }
      ai = bi; av = bv;
    }
  }
  return numOut >> 1;
}

// rare path: triangle crosses the depth range or leaves the S16 snap range -> clip against the frustum and fan
template <int STRIP_ROWS>
__device__ __noinline__ void raster_clipped(float4 v0, float4 v1, float4 v2, unsigned color, int row0,
                                            unsigned long long *zbuf) {
  float bary[18], temp[18];
  int num = 3;
  bary[0] = 0.0f; bary[1] = 0.0f; bary[2] = 1.0f; bary[3] = 0.0f; bary[4] = 0.0f; bary[5] = 1.0f;
  const float a0[4] = {v0.x, v0.y, v0.z, v0.w}, a1[4] = {v1.x, v1.y, v1.z, v1.w}, a2[4] = {v2.x, v2.y, v2.z, v2.w};
  const float d1[4] = {v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, v1.w - v0.w};
  const float d2[4] = {v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, v2.w - v0.w};
  for (int ax = 0; ax < 3; ax++) {
    if ((a0[3] < fabsf(a0[ax])) | (a1[3] < fabsf(a1[ax])) | (a2[3] < fabsf(a2[ax]))) {
      num = clip_poly_plane(temp, bary, num, a0[3] + a0[ax], d1[3] + d1[ax], d2[3] + d2[ax]);
      num = clip_poly_plane(bary, temp, num, a0[3] - a0[ax], d1[3] - d1[ax], d2[3] - d2[ax]);
    }
  }
  if (num < 3) return;
  float4 c0, c1, c2;
#define FP_BARY_PT(dst, i)                                             \
  do {                                                                 \
    (dst).x = a0[0] + d1[0] * bary[(i)*2] + d2[0] * bary[(i)*2 + 1];   \
    (dst).y = a0[1] + d1[1] * bary[(i)*2] + d2[1] * bary[(i)*2 + 1];   \
    (dst).z = a0[2] + d1[2] * bary[(i)*2] + d2[2] * bary[(i)*2 + 1];   \
    (dst).w = a0[3] + d1[3] * bary[(i)*2] + d2[3] * bary[(i)*2 + 1];   \
  } while (0)
  FP_BARY_PT(c0, 0);
  FP_BARY_PT(c1, 1);
  for (int i = 2; i < num; i++) {
    FP_BARY_PT(c2, i);
    raster_one<STRIP_ROWS>(c0, c1, c2, color, row0, zbuf);
    c1 = c2;
  }
#undef FP_BARY_PT
}

// NT threads per workgroup: 256 normally; 1024 for tiny batches (Track), where the kernel is bound by the latency of the
// F/NT dependent triangle iterations of each strip rather than by throughput
template <int MODE, int STRIP_ROWS, int NT = 256>
__global__ __launch_bounds__(NT) void raster_shade_kernel(
    const int32_t *__restrict__ faces, int F, int V, const float *__restrict__ uvs, const uint8_t *__restrict__ tex,
    int TH, int TW, float downscale, const PoseRec *__restrict__ recs, const float4 *__restrict__ clip_all,
    const float4 *__restrict__ attr_all, void *__restrict__ out_all, int32_t *__restrict__ tri_id_dbg,
    float *__restrict__ rast_dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long *zbuf = reinterpret_cast<unsigned long long *>(smem);
  const int strip = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int row0 = strip * STRIP_ROWS;
  const float4 *clip = clip_all + (size_t)n * V;
  const float4 *attr = attr_all + (size_t)n * V;

  const unsigned long long clear_key = ((unsigned long long)CR_DEPTH_MAX << 32) | 0xFFFFFFFFull;
  for (int i = tid; i < STRIP_ROWS * CROP; i += NT) zbuf[i] = clear_key;
  __syncthreads();

  for (int f = tid; f < F; f += NT) {
    int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
    if ((unsigned)i0 >= (unsigned)V || (unsigned)i1 >= (unsigned)V || (unsigned)i2 >= (unsigned)V) continue;
    float4 v0 = clip[i0], v1 = clip[i1], v2 = clip[i2];
    if ((v0.w < fabsf(v0.x)) | (v0.w < fabsf(v0.y)) | (v0.w < fabsf(v0.z))) {
      if (((v0.w < +v0.x) & (v1.w < +v1.x) & (v2.w < +v2.x)) | ((v0.w < -v0.x) & (v1.w < -v1.x) & (v2.w < -v2.x)) |
          ((v0.w < +v0.y) & (v1.w < +v1.y) & (v2.w < +v2.y)) | ((v0.w < -v0.y) & (v1.w < -v1.y) & (v2.w < -v2.y)) |
          ((v0.w < +v0.z) & (v1.w < +v1.z) & (v2.w < +v2.z)) | ((v0.w < -v0.z) & (v1.w < -v1.z) & (v2.w < -v2.z)))
        continue;
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
    else raster_clipped<STRIP_ROWS>(v0, v1, v2, (unsigned)(f + 1), row0, zbuf);
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
      float fx = xs * (float)px + xo, fy = xs * (float)py + xo;
      float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w;
      float p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w;
      float p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
      float a0 = p1x * p2y - p1y * p2x, a1 = p2x * p0y - p2y * p0x, a2 = p0x * p1y - p0y * p1x;
      float iw = 1.f / (a0 + a1 + a2);
      b0 = a0 * iw; b1 = a1 * iw;
      float z = p0.z * a0 + p1.z * a1 + p2.z * a2, w = p0.w * a0 + p1.w * a1 + p2.w * a2;
      zw = z / w;
      b0 = clampf(b0, 0.f, 1.f); b1 = clampf(b1, 0.f, 1.f);
      if (b0 != b0) b0 = 0.f;
      if (b1 != b1) b1 = 0.f;
      zw = fmaxf(fminf(zw, 1.f), -1.f);
      idf = (float)(triIdx + 1);
      float b2 = 1.f - b0 - b1;
      float4 q0 = attr[vi0], q1 = attr[vi1], q2 = attr[vi2];
      xyz0 = b0 * q0.x + b1 * q1.x + b2 * q2.x;
      xyz1 = b0 * q0.y + b1 * q1.y + b2 * q2.y;
      xyz2 = b0 * q0.z + b1 * q1.z + b2 * q2.z;
      dif = b0 * q0.w + b1 * q1.w + b2 * q2.w;
      uu = b0 * uvs[vi0 * 2] + b1 * uvs[vi1 * 2] + b2 * uvs[vi2 * 2];
      vv = b0 * uvs[vi0 * 2 + 1] + b1 * uvs[vi1 * 2 + 1] + b2 * uvs[vi2 * 2 + 1];
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
      u = u * (float)TW - 0.5f; v = v * (float)TH - 0.5f;
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
      float shade = 0.8f + dif * 0.5f;
      for (int c = 0; c < 3; c++) {
        float a00 = (float)t00[c] * sc, a10 = (float)t10[c] * sc, a01 = (float)t01[c] * sc, a11 = (float)t11[c] * sc;
        float top = a00 + u * (a10 - a00), bot = a01 + u * (a11 - a01);
        float rgb = top + v * (bot - top);
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

template <int MODE, int STRIP_ROWS>
static void launch_raster_shade_t(hipStream_t s, const DeviceMesh &m, const PoseRec *recs, int N, const float4 *clip,
                                  const float4 *attr, void *out, int32_t *tri_id_dbg, float *rast_dbg) {
  size_t lds = (size_t)STRIP_ROWS * CROP * sizeof(unsigned long long);
  dim3 grid(CROP / STRIP_ROWS, N), block(256);
  float downscale = m.diameter / 2;
  if (STRIP_ROWS == 8 && N <= 4 && MODE != OUT_F32X6 && !tri_id_dbg && !rast_dbg) {
    hipLaunchKernelGGL((raster_shade_kernel<MODE, STRIP_ROWS, 1024>), grid, dim3(1024), lds, s, m.faces, m.F, m.V, m.uvs,
                       m.tex, m.TH, m.TW, downscale, recs, clip, attr, out, tri_id_dbg, rast_dbg);
    return;
  }
  hipLaunchKernelGGL((raster_shade_kernel<MODE, STRIP_ROWS>), grid, block, lds, s, m.faces, m.F, m.V, m.uvs, m.tex,
                     m.TH, m.TW, downscale, recs, clip, attr, out, tri_id_dbg, rast_dbg);
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
template <int MODE, int STRIP_ROWS>
static void launch_raster_tall(hipStream_t s, const DeviceMesh &m, const PoseRec *recs, int N, const float4 *clip,
                               const float4 *attr, void *out) {
  size_t lds = (size_t)STRIP_ROWS * CROP * sizeof(unsigned long long);
  // once per instantiation, thread-safe (function-local static): opt in to > 64 KB of dynamic LDS
  static const hipError_t attr_rc = hipFuncSetAttribute((const void *)raster_shade_kernel<MODE, STRIP_ROWS, 1024>,
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr_rc;
  hipLaunchKernelGGL((raster_shade_kernel<MODE, STRIP_ROWS, 1024>), dim3(CROP / STRIP_ROWS, N), dim3(1024), lds, s, m.faces, m.F,
                     m.V, m.uvs, m.tex, m.TH, m.TW, m.diameter / 2, recs, clip, attr, out, nullptr, nullptr);
}

template <int MODE>
static void launch_raster_mode(hipStream_t s, const DeviceMesh &m, const PoseRec *recs, int N, const float4 *clip,
                               const float4 *attr, void *out, int32_t *tri_id_dbg, float *rast_dbg) {
  int rows = g_strip_rows_override ? g_strip_rows_override : (N >= 100 ? 1080 : (N >= 64 ? 20 : 8));
  if (rows > 1000 && MODE != OUT_F32X6 && !tri_id_dbg && !rast_dbg) {
    if (rows == 1080) { launch_raster_tall<MODE, 80>(s, m, recs, N, clip, attr, out); return; }
#ifdef FP_TEST_HOOKS
    if (rows == 1040) launch_raster_tall<MODE, 40>(s, m, recs, N, clip, attr, out);
    else launch_raster_tall<MODE, 20>(s, m, recs, N, clip, attr, out);
    return;
#endif
  }
  if (rows > 1000) rows = 20;
#ifdef FP_TEST_HOOKS
  if (rows == 40) { launch_raster_shade_t<MODE, 40>(s, m, recs, N, clip, attr, out, tri_id_dbg, rast_dbg); return; }
#endif
  if (rows == 20) launch_raster_shade_t<MODE, 20>(s, m, recs, N, clip, attr, out, tri_id_dbg, rast_dbg);
  else launch_raster_shade_t<MODE, 8>(s, m, recs, N, clip, attr, out, tri_id_dbg, rast_dbg);
}

void launch_raster_shade(hipStream_t s, const DeviceMesh &m, const PoseRec *recs, int N, const float4 *clip,
                         const float4 *attr, OutMode mode, void *out, int32_t *tri_id_dbg, float *rast_dbg) {
  if (mode == OUT_F32X6) launch_raster_mode<OUT_F32X6>(s, m, recs, N, clip, attr, out, tri_id_dbg, rast_dbg);
  else if (mode == OUT_BF16X8) launch_raster_mode<OUT_BF16X8>(s, m, recs, N, clip, attr, out, tri_id_dbg, rast_dbg);
  else launch_raster_mode<OUT_F16X8>(s, m, recs, N, clip, attr, out, tri_id_dbg, rast_dbg);
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

template <int MODE>
__global__ __launch_bounds__(256) void crop_kernel(const uint8_t *__restrict__ rgb, const float *__restrict__ depth,
                                                   int H, int W, float fx, float fy, float cx, float cy,
                                                   const PoseRec *__restrict__ recs, float downscale,
                                                   void *__restrict__ out_all) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= CROP * CROP) return;
  const int y = i / CROP, x = i - y * CROP;
  const PoseRec &rec = recs[n];
  float sxf = rec.m0 * (float)x + rec.m2, syf = rec.m4 * (float)y + rec.m5;
  float o[6];
  {
    int x0 = (int)floorf(sxf), y0 = (int)floorf(syf);
    float ax = sxf - (float)x0, ay = syf - (float)y0;
    bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W;
    bool vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
    const uint8_t *r0 = rgb + ((size_t)y0 * W + x0) * 3, *r1 = rgb + ((size_t)(y0 + 1) * W + x0) * 3;
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
    if (xn >= 0 && xn < W && yn >= 0 && yn < H) {
      float d = depth[(size_t)yn * W + xn];
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

void launch_crop(hipStream_t s, const uint8_t *rgb, const float *depth, int H, int W, const float *K, const PoseRec *recs,
                 int N, float diameter, OutMode mode, void *out) {
  dim3 grid((CROP * CROP + 255) / 256, N), block(256);
  float downscale = diameter / 2;
  if (mode == OUT_F32X6)
    hipLaunchKernelGGL(crop_kernel<OUT_F32X6>, grid, block, 0, s, rgb, depth, H, W, K[0], K[4], K[2], K[5], recs,
                       downscale, out);
  else if (mode == OUT_BF16X8)
    hipLaunchKernelGGL(crop_kernel<OUT_BF16X8>, grid, block, 0, s, rgb, depth, H, W, K[0], K[4], K[2], K[5], recs,
                       downscale, out);
  else
    hipLaunchKernelGGL(crop_kernel<OUT_F16X8>, grid, block, 0, s, rgb, depth, H, W, K[0], K[4], K[2], K[5], recs,
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

__global__ void pose_update_kernel(float *__restrict__ poses, const float *__restrict__ trans,
                                   const float *__restrict__ rot, int N, float diameter) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float NORM = 0.349065850398865f;
  float P[16];
  for (int k = 0; k < 16; k++) P[k] = poses[(size_t)i * 16 + k];
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
}

void launch_pose_update(hipStream_t s, float *poses, const float *trans, const float *rot, int N, float diameter) {
  hipLaunchKernelGGL(pose_update_kernel, dim3((N + 63) / 64), dim3(64), 0, s, poses, trans, rot, N, diameter);
}

// index[0] = first maximum; when `poses` is given the winner's 4x4 is copied to best_pose so the host needs ONE read-back
__global__ void argmax_kernel(const float *__restrict__ scores, int N, int *__restrict__ index, const float *__restrict__ poses,
                              float *__restrict__ best_pose) {
  __shared__ float sv[256];
  __shared__ int si[256];
  int tid = threadIdx.x;
  float best = -INFINITY;
  int bi = 0x7FFFFFFF;
  for (int i = tid; i < N; i += 256) {
    float v = scores[i];
    if (v > best || bi == 0x7FFFFFFF) { best = v; bi = i; }
  }
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
  if (tid == 0) *index = win;
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
