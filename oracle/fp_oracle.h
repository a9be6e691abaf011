/*
 * fp_oracle.h -- CPU restatement (ORACLE) of the FoundationPose Register/Track geometry path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under foundationpose_cpp_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: the reference (zz990099/foundationpose_cpp @ 2025-05-23) ships no golden vectors or
 * numeric assertions for this path (simple_tests/src/test_foundationpose.cpp:48-155 only checks the bool
 * return) and cannot be built here (CUDA + TensorRT + CV-CUDA + Eigen + OpenCV, PTX inline asm).  This file
 * restates the algorithm from the reference sources, each function citing the file:line it follows, and
 * is pinned only by hand-derived known-answer tests (tests/test_oracle_kat.py).
 *
 * All 4x4 matrices are COLUMN-MAJOR float[16] (Eigen default; translation at 12..14), as handed to the
 * reference kernels (foundationpose_render.cu:337-340).  K is a ROW-MAJOR 3x3 float[9].
 * Images are row-major, rgb u8 [H,W,3] (RGB order), depth f32 [H,W] metres, mask u8 [H,W] (>0 = object).
 */
#ifndef FP_ORACLE_H
#define FP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int V, F;
  const float *verts;    /* [V,3] already centred: vertex - model_center (foundationpose_render.cpp:396-398) */
  const float *normals;  /* [V,3] */
  const float *uvs;      /* [V,2] = (u, 1-v) as the renderer stores them (foundationpose_render.cpp:405-406) */
  const int32_t *faces;  /* [F,3] */
  const uint8_t *tex;    /* [TH,TW,3] RGB */
  int TH, TW;
  float diameter;        /* max pairwise vertex distance (assimp_mesh_loader.cpp:47-60) */
} fpo_mesh;

/* float model of the rendering stage: 1 = multiply-adds contracted like the reference's nvcc -fmad=true build (one documented
 * rule, see fp_oracle.c), 0 = every operation separately rounded.  Process-wide; default 0. */
void fpo_set_fmad(int on);
int fpo_get_fmad(void);

/* foundationpose_sampling.cpp:56-121,178-237.  Returns number of poses written (42*360/step). */
int fpo_rotation_grid(int min_views, int inplane_step_deg, float *out_poses, int max_out);
int fpo_icosphere(int min_views, float *out_verts /*[n,3]*/, int max_out);

/* foundationpose_utils.cu:3-32 (invalid pixels are DEFINED as 0 here; the reference leaves them unwritten). */
void fpo_depth_to_xyz(const float *depth, int H, int W, float fx, float fy, float cx, float cy,
                      float min_depth, float *xyz);

/* foundationpose_sampling.cu:21-82 / :84-164 with the defaults of foundationpose_sampling.cu.hpp:27-44. */
void fpo_erode_depth(const float *depth, float *out, int H, int W, int radius, float depth_diff_thres,
                     float ratio_thres, float zfar);
void fpo_bilateral_filter_depth(const float *depth, float *out, int H, int W, float zfar, int radius,
                                float sigmaD, float sigmaR);

/* foundationpose_sampling.cpp:250-298.  Returns 1 ok, 0 on empty mask / no valid depth. */
int fpo_guess_translation(const float *depth, const uint8_t *mask, int H, int W, const float K[9],
                          float min_depth, float center[3]);

/* foundationpose_sampling.cpp:344-394: erode -> bilateral -> guess translation -> n poses sharing it. */
int fpo_get_hyp_poses(const float *depth, const uint8_t *mask, int H, int W, const float K[9],
                      int inplane_step_deg, float *out_poses, int max_out);

/* foundationpose_render.cpp:25-75.  tfs: [N,9] row-major 3x3. */
void fpo_compute_crop_window_tf(const float *poses, int N, const float K[9], int out_h, int out_w,
                                float crop_ratio, float mesh_diameter, float *tfs);
/* foundationpose_render.cpp:123-149.  bbox2d: [N,4] = tf^-1 * {(0,0),(W-1,H-1)}. */
void fpo_construct_bbox2d(const float *tfs, int N, int out_h, int out_w, float *bbox2d);
/* foundationpose_render.cpp:151-186 (y_down), column-major out. */
void fpo_projection_matrix(const float K[9], int height, int width, float znear, float zfar, float P[16]);

/* Render branch of RenderAndTransform (foundationpose_render.cpp:611-729).
 * render_input [N,oh,ow,6] f32; optional debug outputs (may be NULL):
 *   tri_id [N,oh,ow] i32 = CudaRaster colour buffer (triangle index + 1, 0 = background), raster (y-up) order;
 *   rast_out [N,oh,ow,4] f32 = (b0,b1,z/w,triId+1), raster order (before the vertical flip). */
void fpo_render(const fpo_mesh *mesh, const float *poses, int N, const float K[9], int img_h, int img_w,
                int out_h, int out_w, float crop_ratio, float min_depth, float max_depth,
                float *render_input, int32_t *tri_id, float *rast_out);

/* Crop branch (foundationpose_render.cpp:731-812). transf_input [N,oh,ow,6] f32. */
void fpo_crop(const uint8_t *rgb, const float *depth, int img_h, int img_w, const float K[9],
              const float *poses, int N, int out_h, int out_w, float crop_ratio, float mesh_diameter,
              float min_depth, float max_depth, float *transf_input);

/* foundationpose.cpp:360-406. */
void fpo_refine_post_process(const float *poses, const float *trans, const float *rot, int N,
                             float mesh_diameter, float *out_poses);
/* foundationpose_decoder.cu:24-35 (first max wins). */
int fpo_argmax(const float *scores, int N);

/* assimp_mesh_loader.cpp:47-60 / :16-45,179-180. */
float fpo_mesh_diameter(const float *verts, int V);
void fpo_mesh_center(const float *verts, int V, float center[3]);

int fpo_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
