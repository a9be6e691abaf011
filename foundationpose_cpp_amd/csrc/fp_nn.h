// Refine-net / score-net on MFMA (fp_nn.hip): replaces the two opaque TensorRT engines of the reference
// (refiner_core_->SyncInfer foundationpose.cpp:206-208,254-256; scorer_core_->SyncInfer :218-220).
#pragma once

#include "fp_internal.h"

namespace fp {

// element types of device tensors / MFMA operands
// DT_FP8 = OCP e4m3 (v_mfma_f32_16x16x128_f8f6f4), DT_I8 = signed 8-bit integers (v_mfma_i32_16x16x64_i8; activations are unsigned
// 8-bit values stored with an offset of -128).  Output-type codes of the convolution kernels additionally use DT_DUAL_FP8 /
// DT_DUAL_I8: an f16 tensor (the residual stream) AND its 8-bit copy (the next convolution's operand) written by one epilogue.
// DT_QS_*: the 8-bit copy ALONE, scaled in the epilogue (a layer with a residual whose f16 output nobody reads).
// DT_QSR_I8 / DT_F16RQ_I8: the residual operand is the 8-bit stream copy itself (INT8 networks: no f16 stream at all); output = the
// scaled 8-bit copy alone / an f16 tensor (the token tensor).
enum { DT_F16 = 0, DT_BF16 = 1, DT_FP8 = 2, DT_I8 = 3, DT_DUAL_FP8 = 4, DT_DUAL_I8 = 5, DT_QS_FP8 = 6, DT_QS_I8 = 7, DT_QSR_I8 = 8, DT_F16RQ_I8 = 9 };
// network precision (include/foundationpose_amd.h FP_PREC_*): F16 = the reference's TensorRT --fp16 engines; BF16 = every
// tensor and MFMA operand in bf16 (BASELINE configs[1]); FP8 = the 3x3 trunk convolutions from encodeA.2 on in OCP e4m3
// with per-channel weight / activation scales, everything else f16 (BASELINE configs[4]); INT8 = the same layers on 8-bit
// integers (same matrix-pipe rate class, 5-7x lower rounding noise on the activations: DESIGN.md section 4.4)
enum { PREC_F16 = 0, PREC_BF16 = 1, PREC_FP8 = 2, PREC_INT8 = 3, N_PREC = 4 };

struct Net;        // packed weights of one network, resident in HBM
struct NNScratch;  // activation buffers, grown on demand (one object per precision: border positions depend on element size)

// Reads a packed weight file ("FPW1" format: fp32 tensors in PyTorch layout, BatchNorm already folded into conv
// weight+bias), checks every tensor against the fixed architecture and re-lays it out for the MFMA kernels.
Net *net_load(const char *path, bool is_scorer, int prec, std::string *err);   // = net_prepare + net_commit
// two-phase loading: host work (file, layouts) without any HIP call, then one device allocation + upload
Net *net_prepare(const char *path, bool is_scorer, int prec, std::string *err);
int net_commit(Net *net, std::string *err);
void net_free(Net *);
int net_precision(const Net *);
int net_input_dt(const Net *);   // element type the network expects for nn_in (DT_F16 or DT_BF16)
bool net_q8_ready(const Net *);  // false only for an 8-bit network (PREC_FP8 / PREC_INT8) that has not been given a calibration

// Calibration of the 8-bit networks (fp_api.hip: fp_calibrate).  While a network runs between begin / end its trunk records per-channel
// statistics of its 15 activations ([15][512] floats each): mode 1 (the f16 network) |max| and sum of the values, mode 2 (an 8-bit
// network) the sum of the de-quantised values.  net_apply_q8 turns amax (+ the solved bias / token corrections, null = zero) into
// per-channel activation scales folded into re-quantised weights, corrected biases and the corrected positional table;
// weights = false rebuilds only the biases / the table (the correction sweeps).
void net_calib_begin(Net *net, hipStream_t s, int mode, int only_act = -1);   // only_act >= 0: record that activation only
int net_calib_end(Net *net, hipStream_t s, float *amax_out, float *mean_out);   // [15][512] each; means over the interior pixels
int net_apply_q8(Net *q8_net, const float *amax /*[15][512]*/, const float *bias_fix /*[13][512]*/, const float *tok_fix /*[512]*/, bool weights,
                 const float *frame_means = nullptr /*[n_frames][15][512]: INT8 weights rounded with error feedback against them*/, int n_frames = 0);
int net_q8_set_out_fix(Net *q8_net, const float *fix /* refiner: [6] trans | rot biases; scorer: [512] pooled-feature bias */);
void net_calib_abort(Net *net);
void net_q8_unready(Net *net);
void net_q8_get_fix(const Net *q8_net, float *bias_fix /*[13][512]*/, float *tok_fix /*[512]*/);
int net_q8_bias_channels(int layer);   // output channels of 8-bit layer 0..12 (layer i writes trunk activation i + 2)
bool net_q8_layer_on(const Net *q8_net, int layer);   // [r6] trunk layer 0..12 runs on 8-bit operands in this network (its stage is in the mask)
int net_q8_blocks(const Net *q8_net);                 // the stage mask the network was loaded with (0 for the 2-byte networks)

NNScratch *nn_scratch_create(int prec);
void nn_scratch_free(NNScratch *);
// debug (cross-model corruption checks): the activation arena and the f32 side buffer
void nn_scratch_debug_info(const NNScratch *, const void **buf, size_t *bytes, const void **f32, size_t *f32_bytes);

// Network input: nn_in = [2N,84,84,32] (net_input_dt elements): space-to-depth(2x2) view of the NHWC [2N,160,160,8] tensor
// (channels r,g,b,x,y,z,0,0) with a zero border of 2, rendered crops A in images [0,N), observed crops B in [N,2N).
// Outputs are device pointers.
// shared_b != 0: all N hypotheses share ONE observed crop, stored as image N of nn_in (Register's first refine iteration)
// fuse (N == 1 only, may be null): the head kernel also applies RefinePostProcess to the pose -- one launch less per Track; the caller then
// skips its pose_update launch when *fused_out is set
struct PoseUpdateFuse {
  float *poses;            // [16] updated in place (or from poses_in)
  float diameter;
  const float *poses_in;   // null = poses
  float *extra_out;        // optional second copy of the result
  unsigned *done_flag;     // optional (host-pinned): set to 1, system scope, AFTER the result is stored -- fp_track_wait polls it
};
bool refiner_fuses_pose(const Net *net);   // N == 1: the head kernel applies RefinePostProcess itself (and raises done_flag)
int refiner_forward(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const void *nn_in, int N,
                    float *trans_dev /*[N,3]*/, float *rot_dev /*[N,3]*/, int shared_b = 0, const PoseUpdateFuse *fuse = nullptr,
                    bool *fused_out = nullptr);
int scorer_features(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const void *nn_in, int N,
                    float *feat_dev /*[N,512]*/);
int scorer_head(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const float *feats_dev, int n_total,
                float *scores_dev /*[n_total]*/);

}  // namespace fp
