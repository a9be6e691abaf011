/*
 * foundationpose_amd.h -- C ABI of the MI355X-native FoundationPose Register/Track hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / OpenCV / Eigen types.
 * Every entry point names the reference interface it replaces (path:line under zz990099/foundationpose_cpp,
 * D6F = detection_6d_foundationpose).  INTEGRATION.md shows the C++ shim that puts the reference's own
 * `detection_6d::Base6DofDetectionModel` header on top of these calls.
 *
 * Conventions
 *   - 4x4 poses are COLUMN-MAJOR float[16] (Eigen::Matrix4f::data()), translation at [12..14].
 *   - K is ROW-MAJOR float[9].
 *   - rgb: u8 [H,W,3] RGB order; depth: f32 [H,W] metres; mask: u8 [H,W], >0 = object.
 *   - returned poses are CENTRED-mesh -> camera, exactly like the reference (D6F/src/foundationpose_render.cpp:396-398);
 *     use the mesh centre to convert (D6F/include/detection_6d_foundationpose/mesh_loader.hpp:75-81).
 *   - every function returning int returns 0 on success, non-zero on failure; fp_last_error() describes it
 *     (the reference returns bool + a glog line, D6F/src/foundationpose_utils.hpp:76-84).
 *   - not re-entrant per model (like the reference: one renderer / scratch set per target,
 *     D6F/src/foundationpose.cpp:103-105); DIFFERENT models may be driven from different threads concurrently, each runs
 *     on its own non-blocking stream and their kernels overlap on the GPU.  fp_create / fp_destroy (fp_net_create / fp_net_destroy) and the
 *     entry points that load or rebuild networks (fp_set_precision, fp_set_calibration*, fp_set_float_model) are
 *     exclusive against every other call of the process: they wait for calls in progress and hold new ones back while they run.
 *   - memspace arguments: FP_HOST pointers are ordinary host memory, FP_DEVICE pointers are HIP device memory on
 *     the model's device (lets callers keep frames resident in HBM).
 */
#ifndef FOUNDATIONPOSE_AMD_H
#define FOUNDATIONPOSE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FP_HOST 0
#define FP_DEVICE 1

#define FP_CROP 160            /* crop_window_H/W, D6F/src/foundationpose.cpp:34-35 */
#define FP_NUM_HYP_DEFAULT 252 /* score_mode_poses_num_, D6F/src/foundationpose.cpp:85 */

typedef struct fp_model fp_model;

/* What detection_6d::BaseMeshLoader exposes (D6F/include/detection_6d_foundationpose/mesh_loader.hpp:25-61). */
typedef struct fp_mesh {
  const char *name;          /* GetName() */
  int num_vertices;          /* GetMeshNumVertices() */
  int num_faces;             /* GetMeshNumFaces() */
  const float *vertices;     /* GetMeshVertices()       [V,3], mesh frame (NOT centred) */
  const float *normals;      /* GetMeshVertexNormals()  [V,3] */
  const float *texcoords;    /* GetMeshTextureCoords()  [V,2] = (u,v) as loaded (v is flipped internally) */
  const uint32_t *faces;     /* GetMeshTriangleFaces()  [F,3] */
  const uint8_t *texture;    /* GetTextureMap()         [tex_height,tex_width,3] RGB u8 */
  int tex_height, tex_width;
  float diameter;            /* GetMeshDiameter() */
  float center[3];           /* GetMeshModelCenter() */
} fp_mesh;

/* ---- mesh loading: CreateAssimpMeshLoader(name, mesh_file_path) (mesh_loader.hpp:92-93, src/mesh_loader/assimp_mesh_loader.cpp:159-228)
 * without assimp/OpenCV.  Mesh files: Wavefront OBJ (+ MTL map_Kd) and Stanford PLY (ascii / binary, per-vertex or per-wedge UVs,
 * `comment TextureFile`: the form of the BOP / YCB-V object models); the name is historical.  Textures: PNG (any bit depth, interlaced
 * or not), baseline / progressive JPEG (decoded like libjpeg-turbo), BMP, PNM, TGA -> RGB u8 as cv::imread + BGR2RGB delivers it.  Returns NULL
 * (+ fp_last_error) where the reference throws: empty path, unreadable file, no texture coordinates; also for a texture file that
 * exists but cannot be decoded (never a silent grey texture).  Missing texture -> 2x2 (100,100,100) like the reference. */
typedef struct fp_loaded_mesh fp_loaded_mesh;
fp_loaded_mesh *fp_mesh_load_obj(const char *name, const char *mesh_file_path);
void fp_mesh_free(fp_loaded_mesh *mesh);
/* all BaseMeshLoader getters as one fp_mesh view (valid until fp_mesh_free) */
const fp_mesh *fp_mesh_view(const fp_loaded_mesh *mesh);
/* GetOrientBounds() (column-major 4x4: PCA axes | vertex mean) and GetObjectDimension() */
int fp_mesh_orient_bounds(const fp_loaded_mesh *mesh, float orient_bounds[16], float dimension[3]);

/* ---- construction: CreateFoundationPoseModel (D6F/include/.../foundationpose.hpp:99-105, src/foundationpose.cpp:108-153,448-458).
 * refiner_weights / scorer_weights: paths of packed weight files (tools/pack_weights.py; replaces the TensorRT
 * engines of simple_tests/src/test_foundationpose.cpp:13-14).  NULL = geometry-only model (NN entry points fail). */
fp_model *fp_create(const fp_mesh *meshes, int n_meshes, const float K[9], const char *refiner_weights,
                    const char *scorer_weights, int max_input_image_height, int max_input_image_width);
/* The same on HIP device `device` (-1 = the calling thread's current device, which is what fp_create uses).  A model remembers its
 * device: every entry point makes it current for the duration of the call and restores the caller's afterwards, so a process that
 * drives several GPUs -- one host thread per GPU, the natural C++ shape of SURVEY.md section 8e, or one thread walking over the
 * models -- cannot launch a model's work on the wrong device.  FP_DEVICE pointers handed to a model must live on ITS device. */
fp_model *fp_create_on(int device, const fp_mesh *meshes, int n_meshes, const float K[9], const char *refiner_weights,
                       const char *scorer_weights, int max_input_image_height, int max_input_image_width);
int fp_device(const fp_model *m);
void fp_destroy(fp_model *m);
const char *fp_last_error(void);
/* number of in-plane rotations per icosphere view: 6 -> 252 hypotheses (reference), 24 -> 1008 (SURVEY.md §8a note). */
int fp_set_inplane_steps(fp_model *m, int steps);
int fp_num_hypotheses(const fp_model *m);

/* ---- Base6DofDetectionModel::Register / Track (D6F/include/.../foundationpose.hpp:36-41,59-64; src/foundationpose.cpp:181-265). */
int fp_register(fp_model *m, const uint8_t *rgb, const float *depth, const uint8_t *mask, int H, int W,
                const char *target_name, int refine_itr, float out_pose[16]);
/* Track with refine_itr == 1 reads the frame only inside the observed-crop window of the hypothesis: from a host frame only that
 * window is uploaded -- packed into a pinned block of the model and fetched from there when it is at most half the frame wide,
 * whole rows otherwise (the stage operators below refuse the resulting partial copy until the next fp_upload_frame / Register). */
int fp_track(fp_model *m, const uint8_t *rgb, const float *depth, int H, int W, const float hyp_pose[16],
             const char *target_name, int refine_itr, float out_pose[16]);
/* Track in two halves for pipelined serving (one host thread, several models / objects in flight; the reference's un-vendored
 * async_pipeline plays this role, D6F/src/foundationpose_utils.hpp:33-37): fp_track_submit ENQUEUES the frame upload and the whole
 * refinement on the model's stream and returns; fp_track_wait returns the pose once the refinement has finished.  (It polls a flag the
 * last kernel raises in host-pinned memory right after storing the pose there -- ~13 us sooner per Track than a stream wait -- and
 * falls back to hipStreamSynchronize, which also reports device errors; later calls on the model are stream-ordered behind it.)
 * A host frame must stay valid until the wait; one submission per model at a time. */
int fp_track_submit(fp_model *m, const void *rgb, const void *depth, int memspace, int H, int W, const float hyp_pose[16],
                    const char *target_name, int refine_itr);
int fp_track_wait(fp_model *m, float out_pose[16]);
/* Track of K (1..64) objects of one frame as ONE batch: hyp_poses / out_poses are K column-major 4x4 matrices, target_names K names
 * (objects with the same mesh should be adjacent).  The rendering runs per object, the refine-net once over all K crops; at this
 * size the step is launch-latency-bound, so K objects cost little more than one. */
int fp_track_multi(fp_model *m, const void *rgb, const void *depth, int memspace, int H, int W, int K, const float *hyp_poses,
                   const char *const *target_names, int refine_itr, float *out_poses);
/* same, frame already resident in HBM (memspace FP_DEVICE for rgb/depth/mask) */
int fp_register_ex(fp_model *m, const void *rgb, const void *depth, const void *mask, int memspace, int H, int W,
                   const char *target_name, int refine_itr, float out_pose[16]);
int fp_track_ex(fp_model *m, const void *rgb, const void *depth, int memspace, int H, int W,
                const float hyp_pose[16], const char *target_name, int refine_itr, float out_pose[16]);

/* ---- stage-level operators (what the reference's orchestrator calls; used by the parity tests) ---- */

/* UploadDataToDevice + convert_depth_to_xyz_map (src/foundationpose.cpp:267-315, src/foundationpose_utils.cu:3-32). */
int fp_upload_frame(fp_model *m, const void *rgb, const void *depth, int memspace, int H, int W);
/* optional readback of the xyz map [H,W,3] f32 (pixels with depth < 0.001 are 0). */
int fp_get_xyz_map(fp_model *m, float *xyz_host);

/* FoundationPoseSampler::GetHypPoses (src/foundationpose_sampling.hpp:18-22, .cpp:344-394) on the uploaded frame.
 * poses_out: host [fp_num_hypotheses()*16]. Fails on empty mask / no valid depth like the reference (.cpp:269,278). */
int fp_get_hyp_poses(fp_model *m, const void *mask, int memspace, float *poses_out, int *n_out);
/* erode_depth / bilateral_filter_depth (src/foundationpose_sampling.cu:172-204) on the uploaded frame; host outputs [H,W]. */
int fp_filter_depth(fp_model *m, float *eroded_out, float *bilateral_out);

/* FoundationPoseRenderer::RenderAndTransform (src/foundationpose_render.hpp:29-37, .cpp:814-857):
 * poses host [N*16]; render_out / transf_out: [N,160,160,6] f32 NHWC in `out_memspace` (either may be NULL). */
int fp_render_and_transform(fp_model *m, const char *target_name, const float *poses, int N, float crop_ratio,
                            float *render_out, float *transf_out, int out_memspace);
/* debug view of the rasteriser (CR::CudaRaster colour buffer + nvdiffrast rast_out, src/foundationpose_render.cu:184-232):
 * tri_id host [N,160,160] i32 (triangle index+1, raster/y-up order), rast_out host [N,160,160,4] f32; either may be NULL. */
int fp_debug_rasterize(fp_model *m, const char *target_name, const float *poses, int N, float crop_ratio,
                       int32_t *tri_id, float *rast_out);

/* refiner_core_->SyncInfer (src/foundationpose.cpp:206-208; blobs :78-81): inputs [N,160,160,6] f32 NHWC,
 * outputs host trans[N,3], rot[N,3]. */
int fp_refiner_infer(fp_model *m, const float *render_input, const float *transf_input, int memspace, int N,
                     float *trans_out, float *rot_out);
/* scorer_core_->SyncInfer (src/foundationpose.cpp:218-220; blob :83): outputs host scores[N]. */
int fp_scorer_infer(fp_model *m, const float *render_input, const float *transf_input, int memspace, int N,
                    float *scores_out);
/* RefinePostProcess (src/foundationpose.cpp:360-406): host in/out. */
int fp_refine_post_process(fp_model *m, const char *target_name, const float *poses, const float *trans,
                           const float *rot, int N, float *poses_out);
/* getMaxScoreIndex (src/foundationpose_decoder.cu:24-35): first maximum wins. scores host [N].  A NaN among the scores is an error
 * (the reference always returns an index in [0, N); thrust::max_element's answer on NaNs is unspecified). */
int fp_argmax(fp_model *m, const float *scores, int N, int *index_out);

/* ---- hypothesis sharding over GPUs (new).  One process per GPU calls:
 *   fp_register_shard_begin : sampler (redundant on every rank) + refine + score trunk for hypotheses
 *                             [shard_begin, shard_begin+shard_count) of the fp_num_hypotheses() grid;
 *                             leaves pooled score features [shard_count,512] f32 and refined poses [shard_count,16]
 *                             in device buffers returned through feat_dev / poses_dev;
 *   (caller all-gathers both over RCCL);
 *   fp_register_shard_finish: cross-hypothesis attention + Linear + arg-max over all N gathered rows. */
int fp_register_shard_begin(fp_model *m, const void *rgb, const void *depth, const void *mask, int memspace, int H,
                            int W, const char *target_name, int refine_itr, int shard_begin, int shard_count,
                            float **feat_dev, float **poses_dev);
int fp_register_shard_finish(fp_model *m, const float *all_feat_dev, const float *all_poses_dev, int N_total,
                             float out_pose[16], int *best_index, float *scores_host /* may be NULL */);
/* Read a device buffer the library handed out (feat_dev / poses_dev above) back to the host: a copy ordered on the model's own
 * stream + one synchronisation.  For callers without a HIP runtime of their own (the ctypes host side, tests); never uses the
 * legacy stream (a plain hipMemcpy fails with hipErrorStreamCaptureImplicit while any thread of the process captures a graph). */
int fp_download(fp_model *m, void *dst_host, const void *src_dev, size_t bytes);
/* The same exchange without host stalls (what foundationpose_cpp_amd/distributed.py and bench.py --gpus N use):
 *   fp_register_shard_begin_packed : as above, but only ENQUEUES on the model's stream (fp_stream) and leaves one row
 *                                    [feature 512 | pose 16] per hypothesis in the CALLER's device buffer packed_dev
 *                                    [rows_per_rank, 528] f32 (rows >= shard_count zeroed; shard_count may be 0).  No
 *                                    synchronisation, no allocation: order the collective after it with an event on fp_stream.
 *   (caller: ONE all_gather of packed_dev into gathered [world * rows_per_rank, 528]; with contiguous shards of
 *    rows_per_rank hypotheses the gathered rows are already in global hypothesis order)
 *   fp_register_shard_finish_packed: make fp_stream wait for the collective, then call; the cross-hypothesis head and the
 *                                    arg-max run over rows [0, n_total); one synchronisation at the end. */
int fp_register_shard_begin_packed(fp_model *m, const void *rgb, const void *depth, const void *mask, int memspace, int H,
                                   int W, const char *target_name, int refine_itr, int shard_begin, int shard_count,
                                   float *packed_dev, int rows_per_rank);
int fp_register_shard_finish_packed(fp_model *m, const float *gathered_dev, int n_total, float out_pose[16], int *best_index);

/* The whole sharded Register as ONE call, natively ("one ncclAllGather (RCCL over xGMI)"): every rank (one
 * model per GPU; ranks may be processes or threads of one process) passes its RCCL communicator (ncclComm_t as void*; rank and
 * world size are read from it).  Rank r refines and scores hypotheses [r * ceil(N / world), ...) of the fp_num_hypotheses() grid,
 * the rows [feature 512 | pose 16] are exchanged by one ncclAllGather enqueued on the model's stream (no host synchronisation,
 * no staging copies, persistent buffers), and every rank evaluates the cross-hypothesis head and the arg-max redundantly: all
 * ranks return the same pose / index without a second collective.  A rank whose own half fails still joins the collective (with
 * NaN rows, which every other rank reports) so that nobody hangs.  RCCL is bound at first use (dlopen: a copy already in the
 * process, else librccl.so.1): the library has no link-time RCCL dependency.  world == 1 works without RCCL traffic.
 * A rank that cannot even allocate its exchange buffers cannot join: it calls ncclCommAbort (the other ranks' collective returns an
 * error instead of hanging; the communicator must be re-created) and reports that.  With more than one rank the call does not take
 * the FP_SERIALIZE_MODELS lock (one thread per rank would deadlock on it; ranks live on different devices). */
int fp_register_sharded(fp_model *m, void *nccl_comm, const void *rgb, const void *depth, const void *mask, int memspace, int H, int W,
                        const char *target_name, int refine_itr, float out_pose[16], int *best_index /* may be NULL */);

/* ---- frame / dataset I/O of the acceptance harness (simple_tests/include/tests/help_func.hpp) without OpenCV ----
 * dataset layout test_data/download.md:6-15: <dir>/cam_K.txt, rgb/<id>.png, depth/<id>.png (u16 mm), masks/<id>.png, mesh/ */
/* generic PNG decode (8/16-bit, grey/RGB/palette/alpha, non-interlaced): samples widened to u16, `channels` interleaved.
 * out may be NULL to query the size. */
int fp_image_read_png(const char *path, int *H, int *W, int *channels, int *bit_depth, uint16_t *out, size_t out_capacity);
int fp_frame_size(const char *rgb_path, int *H, int *W);
/* ReadRgbDepthMask / ReadRgbDepth (help_func.hpp:10-53): rgb u8 [H,W,3] RGB; depth f32 [H,W] = u16 / 1000; mask u8 [H,W]
 * (first channel).  Any of the three outputs (with its path) may be NULL. */
int fp_read_rgb_depth_mask(const char *rgb_path, const char *depth_path, const char *mask_path, int H, int W,
                           uint8_t *rgb, float *depth, uint8_t *mask);
/* ReadCamK (help_func.hpp:108-129): nine whitespace-separated numbers, row-major. */
int fp_read_cam_k(const char *cam_K_path, float K[9]);
/* cv::imwrite stand-in: 8-bit RGB PNG. */
int fp_image_write_png_rgb(const char *path, const uint8_t *rgb, int H, int W);
/* draw3DBoundingBox (help_func.hpp:55-106): green 12-edge box of `dimension` under the column-major bbox->camera `pose`
 * (= ConvertPoseMesh2BBox(pose_in_mesh, loader), mesh_loader.hpp:75-81), drawn into rgb in place. */
int fp_draw_bbox3d(uint8_t *rgb, int H, int W, const float K[9], const float pose[16], const float dimension[3]);

/* ---- a network on its own: the infer-core contract ----------------------------------------------------------------
 * The reference hands two deploy_core `BaseInferCore`s to the model and drives them through blobs
 * (GetBuffer / GetTensor(name) / SetBufferLocation / RawPtr / SetShape / SyncInfer: D6F/src/foundationpose.cpp:126-139,
 * 331-354,410-436; factory call shape simple_tests/src/test_foundationpose.cpp:24-35).  fp_net is that contract in C;
 * include/infer_core_amd.hpp puts the C++ names on top.  Blobs: "render_input", "transf_input" f32 [max_batch,160,160,6]
 * (foundationpose.cpp:78-83); refiner outputs "trans", "rot" [max_batch,3]; scorer output "scores" [max_batch]. */
typedef struct fp_net fp_net;
fp_net *fp_net_create(const char *packed_weights_path, int is_scorer, int max_batch);
void fp_net_destroy(fp_net *net);
int fp_net_max_batch(const fp_net *net);
/* pointer of a blob's FP_HOST (pinned) or FP_DEVICE copy; NULL + fp_last_error for an unknown name */
void *fp_net_blob(fp_net *net, const char *name, int memspace);
/* SyncInfer over the first `batch` entries; *_loc say which copy of each input holds the data and whether the outputs are
 * also wanted on the host */
int fp_net_infer(fp_net *net, int batch, int render_loc, int transf_loc, int out_loc);

/* ---- network precision ----------------------------------------------------------------------------------------------
 * The reference runs TensorRT engines built with --fp16 (tools/cvt_onnx2trt.bash:3-15): FP_PREC_F16 is the default and
 * the parity baseline.  FP_PREC_BF16: every tensor and MFMA operand in bf16 (BASELINE configs[1]).
 * FP_PREC_FP8 / FP_PREC_INT8 -- EXPERIMENTAL (round 6: BASELINE configs[4] runs in f16; no subset of trunk stages holds its parity bar,
 * >= 95 % of the refined poses within 1 mm / 1 deg of the 16-bit path on every unseen frame with < 0.3 mm of common mode, on 8-bit operands:
 * DESIGN.md section 4.4) -- the 8-bit MFMA conv path: the 13 3x3 trunk convolutions from encodeA.2 on
 * (91 % of the FLOPs) on 8-bit operands -- OCP e4m3 (v_mfma_f32_16x16x128_f8f6f4) or signed / unsigned 8-bit integers
 * (v_mfma_i32_16x16x64_i8, the same matrix-pipe rate class) -- with per-output-channel weight scales, per-input-channel
 * activation scales folded into the weights, an f16 residual stream (a skip connection is never re-quantised) and a data-driven
 * bias correction; everything else f16.  Each needs ITS OWN fp_calibrate / fp_set_calibration_blob first (the record holds the
 * statistics and the corrections solved against them; fp_set_calibration, the round-2 per-tensor form, serves both).  INT8 is the closer
 * of the two (85-90 % of the poses inside the bar on average, e4m3 under 1 %: uniform 8-bit steps over a ReLU output's range round 5-7x
 * finer than e4m3's 3 mantissa bits; DESIGN.md section 4.4 has the measurements).  Networks of a precision are built from the weight files given
 * to fp_create the first time the precision is selected. */
#define FP_PREC_F16 0
#define FP_PREC_BF16 1
#define FP_PREC_FP8 2
#define FP_PREC_INT8 3
int fp_set_precision(fp_model *m, int precision);
int fp_get_precision(const fp_model *m);
/* Post-training static quantisation for `precision` (FP_PREC_FP8 / FP_PREC_INT8) over the frames of a calibration session:
 *   fp_calibrate_begin(m, precision); fp_calibrate_add_frame(m, frame ...) x K; fp_calibrate_finish(m);
 * finish runs one f16 Register per frame to collect per-channel |max| and mean of the 15 trunk activations of both networks OVER ALL
 * FRAMES, quantises the 8-bit networks from them, then solves the bias / token / output correction layer by layer (~30 Registers per
 * frame).  Use K >= 8 frames of the deployment's scene family (object distances, orientations, sensor noise): the common-mode part of
 * the correction solved on ONE frame is partly that frame's own.  Frames are copied (host) when added; any
 * target / size the model accepts.  The precision's record is replaced only if every step succeeds -- on failure the previous record
 * (or "uncalibrated") stays; the pose is discarded; the model's precision is unchanged.  fp_calibrate_abort drops an open session;
 * fp_calibrate_frames = number of frames added so far (-1: no session).  Only the loading of networks not yet in memory is
 * exclusive against other calls of the process; the Registers of the calibration run like any Register.
 * fp_calibrate(..., precision) = begin + one add_frame + finish; fp_calibrate_fp8 = fp_calibrate(..., FP_PREC_FP8) (the round-2 name). */
int fp_calibrate_begin(fp_model *m, int precision);
int fp_calibrate_add_frame(fp_model *m, const void *rgb, const void *depth, const void *mask, int memspace, int H, int W,
                           const char *target_name);
int fp_calibrate_finish(fp_model *m);
int fp_calibrate_abort(fp_model *m);
int fp_calibrate_frames(const fp_model *m);
int fp_calibrate(fp_model *m, const void *rgb, const void *depth, const void *mask, int memspace, int H, int W,
                 const char *target_name, int precision);
int fp_calibrate_fp8(fp_model *m, const void *rgb, const void *depth, const void *mask, int memspace, int H, int W,
                     const char *target_name);
/* The calibration record of a precision (scales + corrections + the calibration frames' channel means the INT8 weights were rounded
 * against; fp_calibration_size() bytes, versioned: records written by the previous version are still accepted) so that a deployment
 * calibrates once and reuses it; fp_set_calibration_blob reads the precision from the record. */
size_t fp_calibration_size(void);
int fp_get_calibration_blob(const fp_model *m, int precision, void *out, size_t capacity);
int fp_set_calibration_blob(fp_model *m, const void *blob, size_t bytes);
/* Per-tensor view of the same (the round-2 interface): |max| per trunk activation, [refiner 16 | scorer 16], 15 used each.
 * fp_set_calibration gives every channel of a tensor the tensor's scale and NO corrections -- the round-3 behaviour, kept for
 * records made then; prefer the blob. */
int fp_get_calibration(const fp_model *m, float amax_out[32]);
int fp_set_calibration(fp_model *m, const float amax[32]);

/* ---- float model of the rendering stage --------------------------------------------------------------------------
 * The reference's CUDA kernels are compiled by nvcc with its default -fmad=true (D6F/CMakeLists.txt:5 sets only -O3), so
 * the vertex transforms (foundationpose_render.cu:321-443), the nvdiffrast shader / interpolator / texture unit and
 * CudaRaster's clipper run with multiply-adds contracted.  1 (default): the same contraction, spelled with explicit fmaf
 * under one documented rule; 0: every operation separately rounded.  The CPU oracle implements both. */
#define FP_FLOAT_SEPARATE 0
#define FP_FLOAT_FMAD 1
int fp_set_float_model(fp_model *m, int float_model);
int fp_get_float_model(const fp_model *m);

/* ---- measurement hooks ---- */
/* When enabled, every kernel launch is bracketed with HIP events on the model's stream and accumulated per kernel
 * family; fp_profile_report writes "name calls total_ms flops bytes" lines. */
int fp_profile_enable(fp_model *m, int on);
int fp_profile_reset(fp_model *m);
int fp_profile_report(fp_model *m, char *buf, int buf_len);
/* stream used for all launches (hipStream_t as void*), so callers can order their own work / events. */
void *fp_stream(fp_model *m);
int fp_synchronize(fp_model *m);

#ifdef __cplusplus
}
#endif
#endif
