/*
 * fp_oracle.c -- CPU restatement (ORACLE) of the FoundationPose Register/Track geometry path.
 *
 * TEST INFRASTRUCTURE ONLY -- see fp_oracle.h.  PARITY UNPINNED (no golden vectors in the reference).
 * Citations are path:line under /root/reference (D6F = detection_6d_foundationpose,
 * CR = D6F/src/nvdiffrast/common/cudaraster/impl, NVDR = D6F/src/nvdiffrast/common).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp (see Makefile).  Float expressions are written in the
 * operand order of the reference and evaluated without FMA contraction unless fmaf() is spelled out.
 */
#include "fp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------ */
/* small helpers                                                                                     */
/* ------------------------------------------------------------------------------------------------ */

static inline int32_t f2i_bits(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); } /* foundationpose_render.cu:25-28 */

/* ---- float model of the rendering stage ---------------------------------------------------------------------------
 * The reference's kernels are built by nvcc with its default -fmad=true (D6F/CMakeLists.txt:5 sets only -O3): its binary
 * contracts multiply-adds.  This file is compiled with -ffp-contract=off; g_fmad = 1 restates the contraction with
 * explicit fmaf under one documented rule [EXT: ptxas' actual choices are not published]: in a sum of products evaluated
 * left to right every `acc + a*b` becomes fmaf(a, b, acc), the first product of a chain stays a plain multiply,
 * `x - a*b` is fmaf(-a, b, x), `a*b - c*d` is fmaf(a, b, -(c*d)).  g_fmad = 0: every operation separately rounded
 * (round 1's model).  Applies to K4 / K5 / K13 (foundationpose_render.cu:321-443), the nvdiffrast shader, interpolator
 * and texture unit, and CudaRaster's clipper -- exactly the expressions fp_geometry.hip marks with mad<> / dot3<>. */
static int g_fmad = 0;
void fpo_set_fmad(int on) { g_fmad = on != 0; }
int fpo_get_fmad(void) { return g_fmad; }
static inline float MAD(float a, float b, float c) { return g_fmad ? fmaf(a, b, c) : a * b + c; }
static inline float DOT3(float a0, float b0, float a1, float b1, float a2, float b2) { return MAD(a2, b2, MAD(a1, b1, a0 * b0)); }
static inline float DIFFPROD(float a, float b, float c, float d) { return g_fmad ? fmaf(a, b, -(c * d)) : a * b - c * d; }

/* cvt.rni.sat.s32.f32 (CR/Util.inl:37): round-to-nearest-even, saturating, NaN -> 0 */
static inline int32_t f32_to_s32_sat(float a) {
  if (a != a) return 0;
  float r = rintf(a);
  if (r >= 2147483648.0f) return 2147483647;
  if (r <= -2147483648.0f) return (int32_t)(-2147483647 - 1);
  return (int32_t)r;
}
/* cvt.rni.sat.u32.f32 (CR/Util.inl:38) */
static inline uint32_t f32_to_u32_sat(float a) {
  if (a != a) return 0u;
  float r = rintf(a);
  if (r >= 4294967296.0f) return 0xFFFFFFFFu;
  if (r <= 0.0f) return 0u;
  return (uint32_t)r;
}
/* (U32)float cast as CUDA does it: cvt.rzi.u32.f32 (truncate, saturating) -- used by setupPleq */
static inline uint32_t f32_to_u32_trunc(float a) {
  if (a != a) return 0u;
  if (a >= 4294967296.0f) return 0xFFFFFFFFu;
  if (a <= 0.0f) return 0u;
  return (uint32_t)a;
}

int fpo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* 4x4 column-major product C = A*B, k-ordered accumulation (Eigen 4x4 product; no FMA) */
static void mat4_mul(const float *A, const float *B, float *C) {
  float T[16];
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++) {
      float s = A[i + 0] * B[j * 4 + 0];
      s = s + A[i + 4] * B[j * 4 + 1];
      s = s + A[i + 8] * B[j * 4 + 2];
      s = s + A[i + 12] * B[j * 4 + 3];
      T[j * 4 + i] = s;
    }
  memcpy(C, T, sizeof(T));
}

/* ------------------------------------------------------------------------------------------------ */
/* a7: rotation grid  (D6F/src/foundationpose_sampling.cpp:15-121,178-237)                           */
/* ------------------------------------------------------------------------------------------------ */

typedef struct { float x, y, z; } v3;

static v3 v3_normalized(v3 p) { /* Eigen normalized(): p / sqrt(squaredNorm) */
  float n2 = p.x * p.x + p.y * p.y + p.z * p.z;
  if (n2 > 0.0f) { float n = sqrtf(n2); p.x /= n; p.y /= n; p.z /= n; }
  return p;
}

int fpo_icosphere(int min_views, float *out_verts, int max_out) {
  /* foundationpose_sampling.cpp:56-121 */
  int cap = 12, nfaces = 20;
  { int v = 12, f = 20; while (v < min_views) { v = v + (f * 3) / 2; f *= 4; } cap = v; nfaces = f; }
  v3 *verts = (v3 *)malloc(sizeof(v3) * (size_t)cap);
  int (*faces)[3] = (int (*)[3])malloc(sizeof(int[3]) * (size_t)nfaces);
  int (*nfacesbuf)[3] = (int (*)[3])malloc(sizeof(int[3]) * (size_t)nfaces);
  int64_t *ckey = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
  int *cval = (int *)malloc(sizeof(int) * (size_t)cap);
  int nv = 0, nf = 0, nc = 0;
  float t = (float)((1.0 + sqrt(5.0)) / 2.0); /* :63 */
  const float init[12][3] = {{-1, t, 0}, {1, t, 0}, {-1, -t, 0}, {1, -t, 0}, {0, -1, t}, {0, 1, t},
                             {0, -1, -t}, {0, 1, -t}, {t, 0, -1}, {t, 0, 1}, {-t, 0, -1}, {-t, 0, 1}}; /* :64-75 */
  for (int i = 0; i < 12; i++) { v3 p = {init[i][0], init[i][1], init[i][2]}; verts[nv++] = v3_normalized(p); }
  const int f0[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4},
                         {11, 10, 2}, {10, 7, 6}, {7, 1, 8}, {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8},
                         {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}}; /* :78-97 */
  for (int i = 0; i < 20; i++) { faces[nf][0] = f0[i][0]; faces[nf][1] = f0[i][1]; faces[nf][2] = f0[i][2]; nf++; }
  while (nv < min_views) { /* :100 */
    int nnf = 0;
    for (int fi = 0; fi < nf; fi++) {
      int abc[3] = {faces[fi][0], faces[fi][1], faces[fi][2]};
      int mid[3];
      for (int e = 0; e < 3; e++) { /* GetMiddlePoint(a,b) (b,c) (c,a)  :28-52,109-111 */
        int i = abc[e], j = abc[(e + 1) % 3];
        int64_t sm = i < j ? i : j, gr = i < j ? j : i, key = (sm << 32) + gr;
        int found = -1;
        for (int c = 0; c < nc; c++) if (ckey[c] == key) { found = cval[c]; break; }
        if (found < 0) {
          v3 p1 = verts[i], p2 = verts[j];
          v3 pm = {(p1.x + p2.x) / 2.0f, (p1.y + p2.y) / 2.0f, (p1.z + p2.z) / 2.0f};
          verts[nv] = v3_normalized(pm);
          found = nv++;
          ckey[nc] = key; cval[nc] = found; nc++;
        }
        mid[e] = found;
      }
      int a = abc[0], b = abc[1], c = abc[2], ab = mid[0], bc = mid[1], ca = mid[2];
      const int nf4[4][3] = {{a, ab, ca}, {b, bc, ab}, {c, ca, bc}, {ab, bc, ca}}; /* :113-116 */
      for (int q = 0; q < 4; q++) { nfacesbuf[nnf][0] = nf4[q][0]; nfacesbuf[nnf][1] = nf4[q][1]; nfacesbuf[nnf][2] = nf4[q][2]; nnf++; }
    }
    /* grow for the next round */
    nf = nnf;
    int (*tmp)[3] = faces; faces = nfacesbuf; nfacesbuf = tmp;
    nfacesbuf = (int (*)[3])realloc(nfacesbuf, sizeof(int[3]) * (size_t)nf * 4);
    faces = (int (*)[3])realloc(faces, sizeof(int[3]) * (size_t)nf * 4);
  }
  int n = nv < max_out ? nv : max_out;
  for (int i = 0; i < n; i++) { out_verts[i * 3] = verts[i].x; out_verts[i * 3 + 1] = verts[i].y; out_verts[i * 3 + 2] = verts[i].z; }
  free(verts); free(faces); free(nfacesbuf); free(ckey); free(cval);
  return nv;
}

int fpo_rotation_grid(int min_views, int inplane_step_deg, float *out_poses, int max_out) {
  int cap = 12; { int v = 12, f = 20; while (v < min_views) { v = v + (f * 3) / 2; f *= 4; } cap = v; }
  float *vv = (float *)malloc(sizeof(float) * 3 * (size_t)cap);
  int nv = fpo_icosphere(min_views, vv, cap);
  int count = 0;
  for (int i = 0; i < nv; i++) {
    /* SampleViewsIcosphere :178-203 */
    v3 pos = {vv[i * 3], vv[i * 3 + 1], vv[i * 3 + 2]};
    v3 z = {-pos.x, -pos.y, -pos.z}; z = v3_normalized(z);
    v3 up = {0, 0, 1};
    v3 x = {up.y * z.z - up.z * z.y, up.z * z.x - up.x * z.z, up.x * z.y - up.y * z.x};
    if (x.x == 0.0f && x.y == 0.0f && x.z == 0.0f) { x.x = 1; x.y = 0; x.z = 0; }
    x = v3_normalized(x);
    v3 y = {z.y * x.z - z.z * x.y, z.z * x.x - z.x * x.z, z.x * x.y - z.y * x.x};
    y = v3_normalized(y);
    float cam_in_ob[16] = {x.x, x.y, x.z, 0, y.x, y.y, y.z, 0, z.x, z.y, z.z, 0, pos.x, pos.y, pos.z, 1};
    /* MakeRotationGrid :212-237 */
    for (double inplane = 0; inplane < 360; inplane += inplane_step_deg) {
      float a = (float)(inplane * M_PI / 180.0f);
      float s = sinf(a), c = cosf(a);
      /* Eigen AngleAxisf(a, UnitZ).toRotationMatrix(): R = [[c,-s,0],[s,c,0],[0,0,(1-c)*1*1+c]] */
      float Rz[16] = {c, s, 0, 0, -s, c, 0, 0, 0, 0, (1.0f - c) + c, 0, 0, 0, 0, 1};
      float M[16];
      mat4_mul(cam_in_ob, Rz, M);
      /* ob_in_cam = M.inverse(): rigid inverse evaluated in double (Eigen's 4x4 float inverse agrees to ~1e-7) */
      double R[9], tt[3] = {M[12], M[13], M[14]};
      for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) R[r * 3 + cc] = M[cc * 4 + r]; /* R[r][c] */
      float *o = out_poses + (size_t)count * 16;
      if (count < max_out) {
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) o[cc * 4 + r] = (float)R[cc * 3 + r]; /* R^T */
        for (int r = 0; r < 3; r++) o[12 + r] = (float)(-(R[0 * 3 + r] * tt[0] + R[1 * 3 + r] * tt[1] + R[2 * 3 + r] * tt[2]));
        o[3] = o[7] = o[11] = 0; o[15] = 1;
      }
      count++;
    }
  }
  free(vv);
  /* ClusterPoses (:130-176) is called at :235 but its result is discarded -> no effect. */
  return count;
}

/* ------------------------------------------------------------------------------------------------ */
/* a5: depth -> xyz  (D6F/src/foundationpose_utils.cu:3-32)                                          */
/* ------------------------------------------------------------------------------------------------ */

void fpo_depth_to_xyz(const float *depth, int H, int W, float fx, float fy, float cx, float cy,
                      float min_depth, float *xyz) {
#pragma omp parallel for schedule(static)
  for (int r = 0; r < H; r++)
    for (int c = 0; c < W; c++) {
      int p = r * W + c;
      float d = depth[p];
      float *o = xyz + (size_t)p * 3;
      if (d < min_depth) { o[0] = o[1] = o[2] = 0.0f; continue; } /* reference leaves these unwritten (:21-22) */
      o[0] = ((float)c - cx) * d / fx;
      o[1] = ((float)r - cy) * d / fy;
      o[2] = d;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* a8: erode / bilateral / GuessTranslation                                                          */
/* ------------------------------------------------------------------------------------------------ */

void fpo_erode_depth(const float *depth, float *out, int H, int W, int radius, float depth_diff_thres,
                     float ratio_thres, float zfar) {
  /* foundationpose_sampling.cu:21-82 */
#pragma omp parallel for schedule(static)
  for (int h = 0; h < H; h++)
    for (int w = 0; w < W; w++) {
      float d_ori = depth[h * W + w];
      if (d_ori < 0.1f || d_ori >= zfar) { out[h * W + w] = 0.0f; continue; }
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
}

void fpo_bilateral_filter_depth(const float *depth, float *out, int H, int W, float zfar, int radius,
                                float sigmaD, float sigmaR) {
  /* foundationpose_sampling.cu:84-164 */
#pragma omp parallel for schedule(static)
  for (int h = 0; h < H; h++)
    for (int w = 0; w < W; w++) {
      out[h * W + w] = 0.0f;
      float mean = 0.0f; int nvalid = 0;
      for (int u = w - radius; u <= w + radius; u++) {
        if (u < 0 || u >= W) continue;
        for (int v = h - radius; v <= h + radius; v++) {
          if (v < 0 || v >= H) continue;
          float cur = depth[v * W + u];
          if (cur >= 0.1f && cur < zfar) { nvalid++; mean += cur; }
        }
      }
      if (nvalid == 0) continue;
      mean /= (float)nvalid;
      float dc = depth[h * W + w], sw = 0.0f, s = 0.0f;
      for (int u = w - radius; u <= w + radius; u++) {
        if (u < 0 || u >= W) continue;
        for (int v = h - radius; v <= h + radius; v++) {
          if (v < 0 || v >= H) continue;
          float cur = depth[v * W + u];
          if (cur >= 0.1f && cur < zfar && fabsf(cur - mean) < 0.01f) {
            float wgt = expf(-((float)((u - w) * (u - w) + (v - h) * (v - h))) / (2.0f * sigmaD * sigmaD) -
                             (dc - cur) * (dc - cur) / (2.0f * sigmaR * sigmaR));
            sw += wgt; s += wgt * cur;
          }
        }
      }
      if (sw > 0.0f && nvalid > 0) out[h * W + w] = s / sw;
    }
}

static int cmp_float(const void *a, const void *b) {
  float x = *(const float *)a, y = *(const float *)b;
  return (x > y) - (x < y);
}

/* 3x3 inverse, row-major, cofactor / determinant in float (Eigen compute_inverse_size3) */
static void mat3_inverse(const float *m, float *inv) {
  float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  float id = 1.0f / det;
  inv[0] = c00 * id; inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  inv[3] = c01 * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  inv[6] = c02 * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

int fpo_guess_translation(const float *depth, const uint8_t *mask, int H, int W, const float K[9],
                          float min_depth, float center[3]) {
  /* foundationpose_sampling.cpp:250-298 */
  int umin = W, umax = -1, vmin = H, vmax = -1;
  size_t nvalid = 0;
  for (int i = 0; i < H; i++)
    for (int j = 0; j < W; j++)
      if (mask[i * W + j] > 0) {
        if (j < umin) umin = j; if (j > umax) umax = j;
        if (i < vmin) vmin = i; if (i > vmax) vmax = i;
        if (depth[i * W + j] >= min_depth) nvalid++;
      }
  if (umax < 0) return 0;   /* "Mask is all zero" :269 */
  if (nvalid == 0) return 0; /* "No valid value in mask" :278 */
  float uc = (float)((umin + umax) / 2.0), vc = (float)((vmin + vmax) / 2.0);
  float *vals = (float *)malloc(sizeof(float) * nvalid);
  size_t n = 0;
  for (int i = 0; i < H; i++)
    for (int j = 0; j < W; j++)
      if (mask[i * W + j] > 0 && depth[i * W + j] >= min_depth) vals[n++] = depth[i * W + j];
  qsort(vals, n, sizeof(float), cmp_float);
  float zc = (n % 2 == 0) ? (float)((vals[n / 2 - 1] + vals[n / 2]) / 2.0) : vals[n / 2];
  free(vals);
  float Ki[9];
  mat3_inverse(K, Ki);
  /* center = K.inverse() * (uc,vc,1) * zc  (:296) */
  for (int r = 0; r < 3; r++) {
    float s = Ki[r * 3 + 0] * uc; s = s + Ki[r * 3 + 1] * vc; s = s + Ki[r * 3 + 2] * 1.0f;
    center[r] = s * zc;
  }
  return 1;
}

int fpo_get_hyp_poses(const float *depth, const uint8_t *mask, int H, int W, const float K[9],
                      int inplane_step_deg, float *out_poses, int max_out) {
  /* foundationpose_sampling.cpp:344-394; defaults foundationpose_sampling.cu.hpp:27-44, min_depth foundationpose.cpp:36 */
  int n = fpo_rotation_grid(40, inplane_step_deg, out_poses, max_out);
  if (n > max_out) n = max_out;
  float *er = (float *)malloc(sizeof(float) * (size_t)H * W), *bi = (float *)malloc(sizeof(float) * (size_t)H * W);
  fpo_erode_depth(depth, er, H, W, 2, 0.001f, 0.8f, 100.0f);
  fpo_bilateral_filter_depth(er, bi, H, W, 100.0f, 2, 2.0f, 100000.0f);
  float c[3];
  int ok = fpo_guess_translation(bi, mask, H, W, K, 0.001f, c);
  free(er); free(bi);
  if (!ok) return 0;
  for (int i = 0; i < n; i++) { out_poses[i * 16 + 12] = c[0]; out_poses[i * 16 + 13] = c[1]; out_poses[i * 16 + 14] = c[2]; }
  return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* a10-a12: crop window maths  (D6F/src/foundationpose_render.cpp:25-186)                            */
/* ------------------------------------------------------------------------------------------------ */

void fpo_compute_crop_window_tf(const float *poses, int N, const float K[9], int out_h, int out_w,
                                float crop_ratio, float mesh_diameter, float *tfs) {
  /* ComputeCropWindowTF :44-75, ComputeTF :25-42.  out_size = {crop_H, crop_W} (:828) and
   * new_tf(0,0) = out_size(0)/(right-left), new_tf(1,1) = out_size(1)/(bottom-top) (:37-38). */
  float r = mesh_diameter * crop_ratio / 2;
  const float off[5][3] = {{0, 0, 0}, {r, 0, 0}, {-r, 0, 0}, {0, r, 0}, {0, -r, 0}};
  for (int i = 0; i < N; i++) {
    const float *t = poses + (size_t)i * 16 + 12;
    float u[5], v[5];
    for (int k = 0; k < 5; k++) {
      float px = t[0] + off[k][0], py = t[1] + off[k][1], pz = t[2] + off[k][2];
      float q[3];
      for (int rr = 0; rr < 3; rr++) { float s = K[rr * 3] * px; s = s + K[rr * 3 + 1] * py; s = s + K[rr * 3 + 2] * pz; q[rr] = s; }
      u[k] = q[0] / q[2]; v[k] = q[1] / q[2];
    }
    float mx = v[0] - v[0];
    for (int k = 1; k < 5; k++) { float d = v[k] - v[0]; if (d > mx) mx = d; } /* rightCols(1).maxCoeff() :66 */
    float radius = fabsf(mx);
    float left = roundf(u[0] - radius), right = roundf(u[0] + radius);
    float top = roundf(v[0] - radius), bottom = roundf(v[0] + radius);
    float sx = (float)out_h / (right - left), sy = (float)out_w / (bottom - top);
    float *tf = tfs + (size_t)i * 9;
    tf[0] = sx; tf[1] = 0; tf[2] = sx * (-left);
    tf[3] = 0; tf[4] = sy; tf[5] = sy * (-top);
    tf[6] = 0; tf[7] = 0; tf[8] = 1;
  }
}

void fpo_construct_bbox2d(const float *tfs, int N, int out_h, int out_w, float *bbox2d) {
  /* ConstructBBox2D :123-149 + TransformPts :86-121.  tf is upper triangular, so the LU inverse Eigen
   * computes (dynamic-size MatrixXf::inverse) reduces to back substitution: inv = [[1/a,0,-c/a],[0,1/b,-d/b],[0,0,1]]. */
  for (int i = 0; i < N; i++) {
    const float *tf = tfs + (size_t)i * 9;
    float i00 = 1.0f / tf[0], i11 = 1.0f / tf[4];
    float i02 = -tf[2] / tf[0], i12 = -tf[5] / tf[4];
    float x1 = (float)(out_w - 1), y1 = (float)(out_h - 1);
    float *b = bbox2d + (size_t)i * 4;
    b[0] = (i00 * 0.0f + 0.0f * 0.0f) + i02;
    b[1] = (0.0f * 0.0f + i11 * 0.0f) + i12;
    b[2] = (i00 * x1 + 0.0f * y1) + i02;
    b[3] = (0.0f * x1 + i11 * y1) + i12;
  }
}

void fpo_projection_matrix(const float K[9], int height, int width, float znear, float zfar, float P[16]) {
  /* ProjectMatrixFromIntrinsics :151-186, "y_down" branch :175-178; P is column-major here */
  int x0 = 0, y0 = 0, w = width, h = height;
  float nc = znear, fc = zfar;
  float depth = fc - nc, q = -(fc + nc) / depth, qn = -2 * (fc * nc) / depth;
  float rm[16] = {2 * K[0] / w, -2 * K[1] / w, (-2 * K[2] + w + 2 * x0) / w, 0,
                  0, 2 * K[4] / h, (2 * K[5] - h + 2 * y0) / h, 0,
                  0, 0, q, qn,
                  0, 0, -1, 0};
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) P[c * 4 + r] = rm[r * 4 + c];
}

/* ------------------------------------------------------------------------------------------------ */
/* a14: CudaRaster semantics  (CR/TriangleSetup.inl, CR/Util.inl, CR/FineRaster.inl)                 */
/* ------------------------------------------------------------------------------------------------ */

#define CR_SUBPIXEL_LOG2 4                          /* CR/Constants.hpp:23 */
#define CR_MAXVIEWPORT_LOG2 11                      /* CR/Constants.hpp:22 */
#define CR_LERP_ERROR(s) (2200u << (s))             /* CR/Constants.hpp:78 */
#define CR_DEPTH_MIN CR_LERP_ERROR(3)               /* :79 */
#define CR_DEPTH_MAX (0xFFFFFFFFu - CR_LERP_ERROR(3)) /* :80 */

typedef struct { float x, y, z, w; } f4;
typedef struct { int32_t x, y; } i2;

static inline int imin3(int a, int b, int c) { int m = a < b ? a : b; return m < c ? m : c; }
static inline int imax3(int a, int b, int c) { int m = a > b ? a : b; return m > c ? m : c; }

/* CR/Util.inl:101-132 */
static int clip_polygon_with_plane(float *out, const float *in, int numIn, float v0, float v1, float v2) {
  int numOut = 0;
  if (numIn >= 3) {
    int ai = (numIn - 1) * 2;
    float av = MAD(v2, in[ai + 1], MAD(v1, in[ai + 0], v0));
    for (int bi = 0; bi < numIn * 2; bi += 2) {
      float bv = MAD(v2, in[bi + 1], MAD(v1, in[bi + 0], v0));
      if (av * bv < 0.0f) {
        float bc = av / (av - bv), ac = 1.0f - bc;
        out[numOut + 0] = MAD(in[bi + 0], bc, in[ai + 0] * ac);
        out[numOut + 1] = MAD(in[bi + 1], bc, in[ai + 1] * ac);
        numOut += 2;
      }
      if (bv >= 0.0f) { out[numOut + 0] = in[bi + 0]; out[numOut + 1] = in[bi + 1]; numOut += 2; }
      ai = bi; av = bv;
    }
  }
  return numOut >> 1;
}

/* CR/Util.inl:136-160 */
static int clip_triangle_with_frustum(float *bary, const float *v0, const float *v1, const float *v2,
                                      const float *d1, const float *d2) {
  int num = 3;
  bary[0] = 0.0f; bary[1] = 0.0f; bary[2] = 1.0f; bary[3] = 0.0f; bary[4] = 0.0f; bary[5] = 1.0f;
  for (int ax = 0; ax < 3; ax++) {
    if ((v0[3] < fabsf(v0[ax])) | (v1[3] < fabsf(v1[ax])) | (v2[3] < fabsf(v2[ax]))) {
      float temp[18];
      num = clip_polygon_with_plane(temp, bary, num, v0[3] + v0[ax], d1[3] + d1[ax], d2[3] + d2[ax]);
      num = clip_polygon_with_plane(bary, temp, num, v0[3] - v0[ax], d1[3] - d1[ax], d2[3] - d2[ax]);
    }
  }
  return num;
}

typedef struct {
  i2 p0, p1, p2;        /* snapped vertices (sub-pixel units, viewport-centred) after the area<0 swap */
  uint32_t zx, zy, zb;  /* fixed-point depth plane (CRTriangleData, CR/PrivateDefs.hpp) */
} cr_tri;

/* snapTriangle CR/TriangleSetup.inl:11-24 */
static void snap_triangle(int vw, int vh, f4 v0, f4 v1, f4 v2, i2 *p0, i2 *p1, i2 *p2, float rcpW[3]) {
  float sx = (float)(vw << (CR_SUBPIXEL_LOG2 - 1)), sy = (float)(vh << (CR_SUBPIXEL_LOG2 - 1));
  rcpW[0] = 1.0f / v0.w; rcpW[1] = 1.0f / v1.w; rcpW[2] = 1.0f / v2.w;
  p0->x = f32_to_s32_sat(v0.x * rcpW[0] * sx); p0->y = f32_to_s32_sat(v0.y * rcpW[0] * sy);
  p1->x = f32_to_s32_sat(v1.x * rcpW[1] * sx); p1->y = f32_to_s32_sat(v1.y * rcpW[1] * sy);
  p2->x = f32_to_s32_sat(v2.x * rcpW[2] * sx); p2->y = f32_to_s32_sat(v2.y * rcpW[2] * sy);
}

/* setupPleq CR/Util.inl:184-210 */
static void setup_pleq(const float zv[3], i2 v0, i2 d1, i2 d2, float areaRcp, uint32_t *px, uint32_t *py, uint32_t *pz) {
  float mx = fmaxf(fmaxf(zv[0], zv[1]), zv[2]);
  int sh = (f2i_bits(mx) >> 23) - (127 + 22);
  sh = sh < 0 ? 0 : sh; sh = sh > 8 ? 8 : sh;
  int32_t t0 = (int32_t)(f32_to_u32_trunc(zv[0]) >> sh);
  int32_t t1 = (int32_t)((f32_to_u32_trunc(zv[1]) >> sh) - (uint32_t)t0);
  int32_t t2 = (int32_t)((f32_to_u32_trunc(zv[2]) >> sh) - (uint32_t)t0);
  uint32_t rcpMant = ((uint32_t)f2i_bits(areaRcp) & 0x007FFFFFu) | 0x00800000u;
  int rcpShift = (23 + 127) - (f2i_bits(areaRcp) >> 23);
  int64_t xc = ((int64_t)t1 * d2.y - (int64_t)t2 * d1.y) * (int64_t)rcpMant;
  int64_t yc = ((int64_t)t2 * d1.x - (int64_t)t1 * d2.x) * (int64_t)rcpMant;
  uint32_t plx = (uint32_t)(xc >> (rcpShift - (sh + CR_SUBPIXEL_LOG2)));
  uint32_t ply = (uint32_t)(yc >> (rcpShift - (sh + CR_SUBPIXEL_LOG2)));
  int32_t centerX = (v0.x * 2 + imin3(d1.x, d2.x, 0) + imax3(d1.x, d2.x, 0)) >> (CR_SUBPIXEL_LOG2 + 1);
  int32_t centerY = (v0.y * 2 + imin3(d1.y, d2.y, 0) + imax3(d1.y, d2.y, 0)) >> (CR_SUBPIXEL_LOG2 + 1);
  int32_t vcx = v0.x - (int32_t)((uint32_t)centerX << CR_SUBPIXEL_LOG2);
  int32_t vcy = v0.y - (int32_t)((uint32_t)centerY << CR_SUBPIXEL_LOG2);
  uint32_t plz = (uint32_t)t0 << sh;
  plz -= (uint32_t)(((xc >> 13) * vcx + (yc >> 13) * vcy) >> (rcpShift - (sh + 13)));
  plz -= plx * (uint32_t)centerX + ply * (uint32_t)centerY;
  *px = plx; *py = ply; *pz = plz;
}

/* prepareTriangle (area / degenerate part; the between-sample culls at :59-113 only drop triangles that cover no
 * sample and are therefore omitted) + setupTriangle CR/TriangleSetup.inl:42-58,120-177 */
static int setup_triangle(int vw, int vh, f4 v0, f4 v1, f4 v2, cr_tri *out) {
  i2 p0, p1, p2; float rcpW[3];
  snap_triangle(vw, vh, v0, v1, v2, &p0, &p1, &p2, rcpW);
  i2 d1 = {p1.x - p0.x, p1.y - p0.y}, d2 = {p2.x - p0.x, p2.y - p0.y};
  int32_t area = d1.x * d2.y - d1.y * d2.x;
  if (area == 0) return 0; /* degenerate :52-53; backface culling is off (foundationpose_render.cu:201) */
  float z0 = v0.z, z1 = v1.z, z2 = v2.z;
  if (area < 0) { /* :131-138 */
    i2 t = d1; d1 = d2; d2 = t; t = p1; p1 = p2; p2 = t;
    float f = z1; z1 = z2; z2 = f; f = rcpW[1]; rcpW[1] = rcpW[2]; rcpW[2] = f;
    area = -area;
  }
  i2 wv0 = {p0.x + (vw << (CR_SUBPIXEL_LOG2 - 1)), p0.y + (vh << (CR_SUBPIXEL_LOG2 - 1))};
  float zcoef = (float)(CR_DEPTH_MAX - CR_DEPTH_MIN) * 0.5f;
  float zbias = (float)(uint32_t)(CR_DEPTH_MAX + CR_DEPTH_MIN) * 0.5f;
  /* (v0z * zcoef) * rcpW.x + zbias  (:147-151); nvcc contracts the final mul+add -> spelled as fmaf here */
  float zv[3] = {fmaf(z0 * zcoef, rcpW[0], zbias), fmaf(z1 * zcoef, rcpW[1], zbias), fmaf(z2 * zcoef, rcpW[2], zbias)};
  i2 zv0 = {wv0.x - (1 << (CR_SUBPIXEL_LOG2 - 1)), wv0.y - (1 << (CR_SUBPIXEL_LOG2 - 1))};
  setup_pleq(zv, zv0, d1, d2, 1.0f / (float)area, &out->zx, &out->zy, &out->zb);
  out->p0 = p0; out->p1 = p1; out->p2 = p2;
  return 1;
}

/* Exact coverage rule cover8x8_exact_noLUT CR/Util.inl:304-309 evaluated per pixel:
 * sample covered by edge (origin o relative to the sample, direction d) iff
 *   o.x*d.y - o.y*d.x - (d.y > 0 || (d.y == 0 && d.x <= 0) ? 1 : 0) >= 0 */
static inline int edge_covers(int32_t ox, int32_t oy, int32_t dx, int32_t dy) {
  int32_t e = ox * dy - oy * dx;
  if (dy > 0 || (dy == 0 && dx <= 0)) e--;
  return e >= 0;
}

/* Fine raster: per fragment depth = zx*px + zy*py + zb (u32 wrap) CR/FineRaster.inl:338; kill if depth > old
 * (:343); otherwise the fragment replaces the pixel.  Triangles are consumed in index order, and inside one
 * 32-fragment batch executeROP (:152-172) keeps the highest lane (= latest triangle) among equal depths, so
 * the rule is "nearest wins, ties -> later triangle".  Clear depth = CR_DEPTH_MAX (CR/RasterImpl.cpp:284). */
static void raster_triangle(const cr_tri *t, uint32_t color, int vw, int vh, uint32_t *cbuf, uint32_t *dbuf) {
  int minx = imin3(t->p0.x, t->p1.x, t->p2.x), maxx = imax3(t->p0.x, t->p1.x, t->p2.x);
  int miny = imin3(t->p0.y, t->p1.y, t->p2.y), maxy = imax3(t->p0.y, t->p1.y, t->p2.y);
  /* sample of pixel px sits at px*16 - (vw-1)*8 (CR/FineRaster.inl:77-78) */
  int bx = (vw - 1) << (CR_SUBPIXEL_LOG2 - 1), by = (vh - 1) << (CR_SUBPIXEL_LOG2 - 1);
  int px0 = (minx + bx + 15) >> 4, px1 = (maxx + bx) >> 4; /* ceil / floor (arithmetic shift) */
  int py0 = (miny + by + 15) >> 4, py1 = (maxy + by) >> 4;
  if (px0 < 0) px0 = 0; if (py0 < 0) py0 = 0;
  if (px1 > vw - 1) px1 = vw - 1; if (py1 > vh - 1) py1 = vh - 1;
  int32_t d01x = t->p1.x - t->p0.x, d01y = t->p1.y - t->p0.y;
  int32_t d12x = t->p2.x - t->p1.x, d12y = t->p2.y - t->p1.y;
  int32_t d20x = t->p0.x - t->p2.x, d20y = t->p0.y - t->p2.y;
  for (int py = py0; py <= py1; py++)
    for (int px = px0; px <= px1; px++) {
      int32_t sx = px * 16 - bx, sy = py * 16 - by;
      int32_t o0x = t->p0.x - sx, o0y = t->p0.y - sy;
      if (!edge_covers(o0x, o0y, d01x, d01y)) continue;                        /* c01 :94 */
      if (!edge_covers(o0x + d01x, o0y + d01y, d12x, d12y)) continue;          /* c12 :95 */
      if (!edge_covers(o0x, o0y, d20x, d20y)) continue;                        /* c20 :96 */
      uint32_t depth = t->zx * (uint32_t)px + t->zy * (uint32_t)py + t->zb;
      size_t pi = (size_t)py * vw + px;
      if (depth > dbuf[pi]) continue;
      dbuf[pi] = depth; cbuf[pi] = color;
    }
}

/* triangleSetupImpl CR/TriangleSetup.inl:181-391 for one triangle, instance mode */
static void cr_draw_triangle(int vw, int vh, f4 v0, f4 v1, f4 v2, uint32_t color, uint32_t *cbuf, uint32_t *dbuf) {
  /* outside view frustum => cull (:262-274) */
  if ((v0.w < fabsf(v0.x)) | (v0.w < fabsf(v0.y)) | (v0.w < fabsf(v0.z))) {
    if (((v0.w < +v0.x) & (v1.w < +v1.x) & (v2.w < +v2.x)) | ((v0.w < -v0.x) & (v1.w < -v1.x) & (v2.w < -v2.x)) |
        ((v0.w < +v0.y) & (v1.w < +v1.y) & (v2.w < +v2.y)) | ((v0.w < -v0.y) & (v1.w < -v1.y) & (v2.w < -v2.y)) |
        ((v0.w < +v0.z) & (v1.w < +v1.z) & (v2.w < +v2.z)) | ((v0.w < -v0.z) & (v1.w < -v1.z) & (v2.w < -v2.z)))
      return;
  }
  /* inside depth range => try the fast path (:278-307) */
  if ((v0.w >= fabsf(v0.z)) & (v1.w >= fabsf(v1.z)) & (v2.w >= fabsf(v2.z))) {
    i2 p0, p1, p2; float rcpW[3];
    snap_triangle(vw, vh, v0, v1, v2, &p0, &p1, &p2, rcpW);
    int lox = imin3(p0.x, p1.x, p2.x), loy = imin3(p0.y, p1.y, p2.y);
    int hix = imax3(p0.x, p1.x, p2.x), hiy = imax3(p0.y, p1.y, p2.y);
    int loxy = lox < loy ? lox : loy, hixy = hix > hiy ? hix : hiy;
    int aabbLimit = (1 << (CR_MAXVIEWPORT_LOG2 + CR_SUBPIXEL_LOG2)) - 1;
    if (loxy >= -32768 && hixy <= 32767 && hixy - loxy <= aabbLimit) {
      cr_tri t;
      if (setup_triangle(vw, vh, v0, v1, v2, &t)) raster_triangle(&t, color, vw, vh, cbuf, dbuf);
      return;
    }
  }
  /* clip to the view frustum and fan the polygon (:311-390) */
  float bary[18];
  float ov0[4] = {v0.x, v0.y, v0.z, v0.w}, a1[4] = {v1.x, v1.y, v1.z, v1.w}, a2[4] = {v2.x, v2.y, v2.z, v2.w};
  float od1[4] = {v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, v1.w - v0.w};
  float od2[4] = {v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, v2.w - v0.w};
  int numVerts = clip_triangle_with_frustum(bary, ov0, a1, a2, od1, od2);
  if (numVerts < 3) return;
  f4 c0, c1, c2;
#define BARY_PT(dst, i)                                       \
  do {                                                        \
    (dst).x = MAD(od2[0], bary[(i)*2 + 1], MAD(od1[0], bary[(i)*2], ov0[0])); \
    (dst).y = MAD(od2[1], bary[(i)*2 + 1], MAD(od1[1], bary[(i)*2], ov0[1])); \
    (dst).z = MAD(od2[2], bary[(i)*2 + 1], MAD(od1[2], bary[(i)*2], ov0[2])); \
    (dst).w = MAD(od2[3], bary[(i)*2 + 1], MAD(od1[3], bary[(i)*2], ov0[3])); \
  } while (0)
  BARY_PT(c0, 0); BARY_PT(c1, 1);
  for (int i = 2; i < numVerts; i++) {
    BARY_PT(c2, i);
    cr_tri t;
    if (setup_triangle(vw, vh, c0, c1, c2, &t)) raster_triangle(&t, color, vw, vh, cbuf, dbuf);
    c1 = c2;
  }
#undef BARY_PT
}

/* ------------------------------------------------------------------------------------------------ */
/* a13-a16: render branch                                                                            */
/* ------------------------------------------------------------------------------------------------ */

static inline float lerpf(float a, float b, float c) { return MAD(c, b - a, a); } /* a + c * (b - a), NVDR/texture.cu:14 */

void fpo_render(const fpo_mesh *m, const float *poses, int N, const float K[9], int img_h, int img_w,
                int out_h, int out_w, float crop_ratio, float min_depth, float max_depth,
                float *render_input, int32_t *tri_id, float *rast_out) {
  const int V = m->V, F = m->F, HW = out_h * out_w;
  float *tfs = (float *)malloc(sizeof(float) * 9 * (size_t)N), *bbox = (float *)malloc(sizeof(float) * 4 * (size_t)N);
  fpo_compute_crop_window_tf(poses, N, K, out_h, out_w, crop_ratio, m->diameter, tfs);
  fpo_construct_bbox2d(tfs, N, out_h, out_w, bbox);
  float P[16];
  fpo_projection_matrix(K, img_h, img_w, 0.1f, 100.0f, P);
  const float GL[16] = {1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1}; /* kGLCamInCVCam foundationpose_render.cpp:22-23 */
  const float downscale = m->diameter / 2; /* foundationpose_render.cpp:719 */

#pragma omp parallel for schedule(dynamic, 1)
  for (int n = 0; n < N; n++) {
    const float *pose = poses + (size_t)n * 16;
    float GP[16], M[16];
    mat4_mul(GL, pose, GP);
    mat4_mul(P, GP, M); /* projection_mat * (kGLCamInCVCam * poses[i]) foundationpose_render.cpp:590 */
    f4 *clip = (f4 *)malloc(sizeof(f4) * (size_t)V);
    float *pts_cam = (float *)malloc(sizeof(float) * 3 * (size_t)V), *diffuse = (float *)malloc(sizeof(float) * (size_t)V);
    const float *bb = bbox + (size_t)n * 4;
    /* generate_pose_clip_kernel foundationpose_render.cu:363-398 */
    float l = bb[0], t = img_h - bb[1], r = bb[2], b = img_h - bb[3];
    float a00 = img_w / (r - l), a11 = img_h / (t - b), a30 = (img_w - r - l) / (r - l), a31 = (img_h - t - b) / (t - b);
    for (int v = 0; v < V; v++) {
      float x = m->verts[v * 3], y = m->verts[v * 3 + 1], z = m->verts[v * 3 + 2];
      float tx = DOT3(M[0], x, M[4], y, M[8], z) + M[12];   /* M[0]*x + M[4]*y + M[8]*z + M[12] */
      float ty = DOT3(M[1], x, M[5], y, M[9], z) + M[13];
      float tz = DOT3(M[2], x, M[6], y, M[10], z) + M[14];
      float tw = DOT3(M[3], x, M[7], y, M[11], z) + M[15];
      clip[v].x = MAD(tw, a30, tx * a00); clip[v].y = MAD(tw, a31, ty * a11); clip[v].z = tz; clip[v].w = tw;
      /* transform_points_kernel :321-341 */
      pts_cam[v * 3 + 0] = DOT3(pose[0], x, pose[4], y, pose[8], z) + pose[12];
      pts_cam[v * 3 + 1] = DOT3(pose[1], x, pose[5], y, pose[9], z) + pose[13];
      pts_cam[v * 3 + 2] = DOT3(pose[2], x, pose[6], y, pose[10], z) + pose[14];
      /* transform_normals_kernel :418-443 */
      float nx = m->normals[v * 3], ny = m->normals[v * 3 + 1], nz = m->normals[v * 3 + 2];
      float ux = DOT3(pose[0], nx, pose[4], ny, pose[8], nz);
      float uy = DOT3(pose[1], nx, pose[5], ny, pose[9], nz);
      float uz = DOT3(pose[2], nx, pose[6], ny, pose[10], nz);
      float l2 = sqrtf(DOT3(ux, ux, uy, uy, uz, uz));
      float val = l2 == 0 ? 0 : -uz / l2;
      diffuse[v] = clampf(val, 0, 1);
    }
    /* CudaRaster draw */
    uint32_t *cbuf = (uint32_t *)calloc((size_t)HW, 4), *dbuf = (uint32_t *)malloc((size_t)HW * 4);
    for (int i = 0; i < HW; i++) dbuf[i] = CR_DEPTH_MAX;
    for (int f = 0; f < F; f++) {
      int32_t i0 = m->faces[f * 3], i1 = m->faces[f * 3 + 1], i2_ = m->faces[f * 3 + 2];
      if ((uint32_t)i0 >= (uint32_t)V || (uint32_t)i1 >= (uint32_t)V || (uint32_t)i2_ >= (uint32_t)V) continue; /* :237-244 */
      cr_draw_triangle(out_w, out_h, clip[i0], clip[i1], clip[i2_], (uint32_t)(f + 1), cbuf, dbuf);
    }
    if (tri_id) for (int i = 0; i < HW; i++) tri_id[(size_t)n * HW + i] = (int32_t)cbuf[i];

    /* shading: RasterizeCudaFwdShaderKernel NVDR/rasterize.cu:15-90, InterpolateFwdKernel NVDR/interpolate.cu:16-84,
     * TextureFwdKernelLinear1 NVDR/texture.cu:20-96,132-179, renfine_color_kernel foundationpose_render.cu:459-501,
     * clamp :30-39, Flip(0) foundationpose_render.cpp:676-680, threshold_and_downscale :61-118, concat :121-140 */
    float xs = 2.f / (float)out_w, xo = 1.f / (float)out_w - 1.f, ys = 2.f / (float)out_h, yo = 1.f / (float)out_h - 1.f;
    float *dst = render_input + (size_t)n * HW * 6;
    for (int py = 0; py < out_h; py++)
      for (int px = 0; px < out_w; px++) {
        int pidx = px + out_w * py;
        int triIdx = (int)cbuf[pidx] - 1;
        float b0 = 0, b1 = 0, zw = 0, idf = 0;
        float xyz[3] = {0, 0, 0}, uv[2] = {0, 0}, dif = 0;
        int valid = (triIdx >= 0 && triIdx < F);
        if (valid) {
          int vi0 = m->faces[triIdx * 3], vi1 = m->faces[triIdx * 3 + 1], vi2 = m->faces[triIdx * 3 + 2];
          f4 p0 = clip[vi0], p1 = clip[vi1], p2 = clip[vi2];
          float fx = MAD(xs, (float)px, xo), fy = MAD(ys, (float)py, yo);
          float p0x = MAD(-fx, p0.w, p0.x), p0y = MAD(-fy, p0.w, p0.y);   /* p0.x - fx * p0.w */
          float p1x = MAD(-fx, p1.w, p1.x), p1y = MAD(-fy, p1.w, p1.y);
          float p2x = MAD(-fx, p2.w, p2.x), p2y = MAD(-fy, p2.w, p2.y);
          float a0 = DIFFPROD(p1x, p2y, p1y, p2x), a1 = DIFFPROD(p2x, p0y, p2y, p0x), a2 = DIFFPROD(p0x, p1y, p0y, p1x);
          float iw = 1.f / (a0 + a1 + a2);
          b0 = a0 * iw; b1 = a1 * iw;
          float z = DOT3(p0.z, a0, p1.z, a1, p2.z, a2), w = DOT3(p0.w, a0, p1.w, a1, p2.w, a2);
          zw = z / w;
          b0 = clampf(b0, 0.f, 1.f); b1 = clampf(b1, 0.f, 1.f); /* __saturatef */
          if (b0 != b0) b0 = 0.f; if (b1 != b1) b1 = 0.f;       /* __saturatef(NaN) = +0 */
          zw = fmaxf(fminf(zw, 1.f), -1.f);
          idf = (float)(triIdx + 1);
          float b2 = 1.f - b0 - b1;
          for (int i = 0; i < 3; i++) xyz[i] = DOT3(b0, pts_cam[vi0 * 3 + i], b1, pts_cam[vi1 * 3 + i], b2, pts_cam[vi2 * 3 + i]);
          for (int i = 0; i < 2; i++) uv[i] = DOT3(b0, m->uvs[vi0 * 2 + i], b1, m->uvs[vi1 * 2 + i], b2, m->uvs[vi2 * 2 + i]);
          dif = DOT3(b0, diffuse[vi0], b1, diffuse[vi1], b2, diffuse[vi2]);
        }
        if (rast_out) { float *ro = rast_out + ((size_t)n * HW + pidx) * 4; ro[0] = b0; ro[1] = b1; ro[2] = zw; ro[3] = idf; }
        /* texture: bilinear, wrap, texel centre u*w-0.5; texture = u8 * (1/255) (foundationpose_render.cpp:503-506) */
        float rgb[3];
        {
          int w = m->TW, h = m->TH;
          float u = uv[0], v = uv[1];
          u = u - floorf(u); v = v - floorf(v);
          u = MAD(u, (float)w, -0.5f); v = MAD(v, (float)h, -0.5f);
          int iu0 = (int)floorf(u), iv0 = (int)floorf(v);
          int iu1 = iu0 + 1, iv1 = iv0 + 1;
          u -= (float)iu0; v -= (float)iv0;
          if (iu0 < 0) iu0 += w; if (iv0 < 0) iv0 += h;
          if (iu1 >= w) iu1 -= w; if (iv1 >= h) iv1 -= h;
          const float sc = 1.0f / 255.0f;
          for (int c = 0; c < 3; c++) {
            float a00_ = (float)m->tex[(iu0 + w * iv0) * 3 + c] * sc, a10_ = (float)m->tex[(iu1 + w * iv0) * 3 + c] * sc;
            float a01_ = (float)m->tex[(iu0 + w * iv1) * 3 + c] * sc, a11_ = (float)m->tex[(iu1 + w * iv1) * 3 + c] * sc;
            rgb[c] = lerpf(lerpf(a00_, a10_, u), lerpf(a01_, a11_, u), v);
          }
        }
        float fg = clampf(idf, 0, 1);
        float shade = MAD(dif, 0.5f, 0.8f);
        float o[6];
        for (int c = 0; c < 3; c++) { float q = rgb[c] * shade * fg; q = clampf(q, 0, 1); o[c] = clampf(q, 0.0f, 1.0f); }
        /* threshold_and_downscale on the rendered xyz */
        {
          int invalid = xyz[2] < min_depth;
          float q[3] = {xyz[0] - pose[12], xyz[1] - pose[13], xyz[2] - pose[14]};
          for (int c = 0; c < 3; c++) { q[c] = q[c] / downscale; if (fabsf(q[c]) > max_depth || invalid) q[c] = 0.0f; o[3 + c] = q[c]; }
        }
        int fy_ = out_h - 1 - py; /* vertical flip */
        float *d = dst + ((size_t)fy_ * out_w + px) * 6;
        for (int c = 0; c < 6; c++) d[c] = o[c];
      }
    free(cbuf); free(dbuf); free(clip); free(pts_cam); free(diffuse);
  }
  free(tfs); free(bbox);
}

/* ------------------------------------------------------------------------------------------------ */
/* a17: crop branch  (D6F/src/foundationpose_render.cpp:731-812)                                     */
/* ------------------------------------------------------------------------------------------------ */

void fpo_crop(const uint8_t *rgb, const float *depth, int img_h, int img_w, const float K[9],
              const float *poses, int N, int out_h, int out_w, float crop_ratio, float mesh_diameter,
              float min_depth, float max_depth, float *transf_input) {
  /* cvcuda::WarpPerspective is called WITHOUT WARP_INVERSE_MAP (:751-753,785), so it inverts tf and samples
   * src at tf^-1 * (x,y,1) for dst pixel (x,y); integer coordinates are pixel centres [EXT: CV-CUDA/OpenCV
   * convention, unverifiable from the reference tree].  rgb: bilinear on u8, constant-0 border, result rounded
   * to u8 (round-half-even) then * 1/255 (ConvertTo :790).  xyz: nearest (floor(c+0.5)), constant-0 border. */
  const int HW = out_h * out_w;
  float *tfs = (float *)malloc(sizeof(float) * 9 * (size_t)N);
  fpo_compute_crop_window_tf(poses, N, K, out_h, out_w, crop_ratio, mesh_diameter, tfs);
  const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  const float downscale = mesh_diameter / 2;
#pragma omp parallel for schedule(dynamic, 1)
  for (int n = 0; n < N; n++) {
    const float *tf = tfs + (size_t)n * 9;
    const float *pose = poses + (size_t)n * 16;
    /* inverse in double, coefficients cast to float */
    double a = tf[0], c = tf[2], e = tf[4], f = tf[5];
    float m0 = (float)(1.0 / a), m2 = (float)(-c / a), m4 = (float)(1.0 / e), m5 = (float)(-f / e);
    float *dst = transf_input + (size_t)n * HW * 6;
    for (int y = 0; y < out_h; y++)
      for (int x = 0; x < out_w; x++) {
        float sxf = m0 * (float)x + m2, syf = m4 * (float)y + m5;
        float o[6];
        /* bilinear rgb */
        {
          int x0 = (int)floorf(sxf), y0 = (int)floorf(syf);
          float ax = sxf - (float)x0, ay = syf - (float)y0;
          for (int ch = 0; ch < 3; ch++) {
            float p00 = 0, p10 = 0, p01 = 0, p11 = 0;
            if (y0 >= 0 && y0 < img_h) {
              if (x0 >= 0 && x0 < img_w) p00 = (float)rgb[((size_t)y0 * img_w + x0) * 3 + ch];
              if (x0 + 1 >= 0 && x0 + 1 < img_w) p10 = (float)rgb[((size_t)y0 * img_w + x0 + 1) * 3 + ch];
            }
            if (y0 + 1 >= 0 && y0 + 1 < img_h) {
              if (x0 >= 0 && x0 < img_w) p01 = (float)rgb[((size_t)(y0 + 1) * img_w + x0) * 3 + ch];
              if (x0 + 1 >= 0 && x0 + 1 < img_w) p11 = (float)rgb[((size_t)(y0 + 1) * img_w + x0 + 1) * 3 + ch];
            }
            float val = p00 * ((1.0f - ax) * (1.0f - ay)) + p10 * (ax * (1.0f - ay)) + p01 * ((1.0f - ax) * ay) + p11 * (ax * ay);
            float q = rintf(val); q = q < 0 ? 0 : (q > 255 ? 255 : q);
            o[ch] = q * (1.0f / 255.0f);
          }
        }
        /* nearest xyz, recomputed from depth (identical to sampling the K1 xyz map) */
        {
          int xn = (int)floorf(sxf + 0.5f), yn = (int)floorf(syf + 0.5f);
          float p[3] = {0, 0, 0};
          if (xn >= 0 && xn < img_w && yn >= 0 && yn < img_h) {
            float d = depth[(size_t)yn * img_w + xn];
            if (!(d < 0.001f)) { p[0] = ((float)xn - cx) * d / fx; p[1] = ((float)yn - cy) * d / fy; p[2] = d; }
          }
          int invalid = p[2] < min_depth;
          float q[3] = {p[0] - pose[12], p[1] - pose[13], p[2] - pose[14]};
          for (int ch = 0; ch < 3; ch++) { q[ch] = q[ch] / downscale; if (fabsf(q[ch]) > max_depth || invalid) q[ch] = 0.0f; o[3 + ch] = q[ch]; }
        }
        float *d = dst + ((size_t)y * out_w + x) * 6;
        for (int ch = 0; ch < 6; ch++) d[ch] = o[ch];
      }
  }
  free(tfs);
}

/* ------------------------------------------------------------------------------------------------ */
/* a20 / a22 / a23                                                                                   */
/* ------------------------------------------------------------------------------------------------ */

void fpo_refine_post_process(const float *poses, const float *trans, const float *rot, int N,
                             float mesh_diameter, float *out_poses) {
  /* foundationpose.cpp:360-406 */
  const float NORM = 0.349065850398865f; /* REFINE_ROT_NORMALIZER :82 */
  for (int i = 0; i < N; i++) {
    const float *P = poses + (size_t)i * 16;
    float *O = out_poses + (size_t)i * 16;
    float td[3], v[3];
    for (int k = 0; k < 3; k++) { td[k] = trans[i * 3 + k] * (mesh_diameter / 2); v[k] = tanhf(rot[i * 3 + k]) * NORM; }
    float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float ang = sqrtf(n2);
    float ax[3] = {v[0], v[1], v[2]};
    if (n2 > 0.0f) { ax[0] /= ang; ax[1] /= ang; ax[2] /= ang; }
    /* Eigen AngleAxis::toRotationMatrix */
    float s = sinf(ang), c = cosf(ang);
    float sa[3] = {s * ax[0], s * ax[1], s * ax[2]}, ca[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
    float R[9]; /* row-major */
    float tmp;
    tmp = ca[0] * ax[1]; R[1] = tmp - sa[2]; R[3] = tmp + sa[2];
    tmp = ca[0] * ax[2]; R[2] = tmp + sa[1]; R[6] = tmp - sa[1];
    tmp = ca[1] * ax[2]; R[5] = tmp - sa[0]; R[7] = tmp + sa[0];
    R[0] = ca[0] * ax[0] + c; R[4] = ca[1] * ax[1] + c; R[8] = ca[2] * ax[2] + c;
    /* rot_mat_delta = R^T (:390); result = rot_mat_delta * top_left_3x3 (:400) */
    memcpy(O, P, sizeof(float) * 16);
    O[12] = P[12] + td[0]; O[13] = P[13] + td[1]; O[14] = P[14] + td[2];
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++) {
        float sacc = R[0 * 3 + r] * P[cc * 4 + 0];
        sacc = sacc + R[1 * 3 + r] * P[cc * 4 + 1];
        sacc = sacc + R[2 * 3 + r] * P[cc * 4 + 2];
        O[cc * 4 + r] = sacc;
      }
  }
}

int fpo_argmax(const float *scores, int N) {
  int best = 0;
  for (int i = 1; i < N; i++) if (scores[i] > scores[best]) best = i; /* thrust::max_element: first max */
  return best;
}

float fpo_mesh_diameter(const float *verts, int V) {
  /* assimp_mesh_loader.cpp:47-60: O(V^2) max pairwise distance */
  float best = 0.0f;
#pragma omp parallel for schedule(dynamic, 64) reduction(max : best)
  for (int i = 0; i < V; i++)
    for (int j = i + 1; j < V; j++) {
      float dx = verts[i * 3] - verts[j * 3], dy = verts[i * 3 + 1] - verts[j * 3 + 1], dz = verts[i * 3 + 2] - verts[j * 3 + 2];
      float d = sqrtf(dx * dx + dy * dy + dz * dz);
      if (d > best) best = d;
    }
  return best;
}

void fpo_mesh_center(const float *verts, int V, float center[3]) {
  /* FindMinMaxVertex assimp_mesh_loader.cpp:16-45; center = (max+min)/2.0 (:179-180) */
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  if (V > 0) for (int k = 0; k < 3; k++) mn[k] = mx[k] = verts[k];
  for (int i = 0; i < V; i++)
    for (int k = 0; k < 3; k++) { float q = verts[i * 3 + k]; if (q < mn[k]) mn[k] = q; if (q > mx[k]) mx[k] = q; }
  for (int k = 0; k < 3; k++) center[k] = (float)((mx[k] + mn[k]) / 2.0);
}
