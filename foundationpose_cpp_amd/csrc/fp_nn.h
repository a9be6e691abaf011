// Refine-net / score-net on MFMA (fp_nn.hip): replaces the two opaque TensorRT engines of the reference
// (refiner_core_->SyncInfer foundationpose.cpp:206-208,254-256; scorer_core_->SyncInfer :218-220).
#pragma once

#include "fp_internal.h"

namespace fp {

// element types of device tensors / MFMA operands
enum { DT_F16 = 0, DT_BF16 = 1, DT_FP8 = 2 };
// network precision (include/foundationpose_amd.h FP_PREC_*): F16 = the reference's TensorRT --fp16 engines; BF16 = every
// tensor and MFMA operand in bf16 (BASELINE configs[1]); FP8 = the 3x3 trunk convolutions from encodeA.2 on in OCP e4m3
// with per-channel weight / per-tensor activation scales, everything else f16 (BASELINE configs[4])
enum { PREC_F16 = 0, PREC_BF16 = 1, PREC_FP8 = 2 };

struct Net;        // packed weights of one network, resident in HBM
struct NNScratch;  // activation buffers, grown on demand (one object per precision: border positions depend on element size)

// Reads a packed weight file ("FPW1" format: fp32 tensors in PyTorch layout, BatchNorm already folded into conv
// weight+bias), checks every tensor against the fixed architecture and re-lays it out for the MFMA kernels.
Net *net_load(const char *path, bool is_scorer, int prec, std::string *err);
void net_free(Net *);
int net_precision(const Net *);
int net_input_dt(const Net *);   // element type the network expects for nn_in (DT_F16 or DT_BF16)
bool net_fp8_ready(const Net *); // false only for an FP8 network whose activation scales have not been set

// FP8 calibration: run a 2-byte network between begin / end to collect |max| of its 15 trunk activations, then hand the
// result to the FP8 network of the same weights.
void net_calib_begin(Net *net, hipStream_t s);
int net_calib_end(Net *net, hipStream_t s, float amax_out[16]);
int net_set_fp8_scales(Net *fp8_net, const float amax[16]);

NNScratch *nn_scratch_create();
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
};
int refiner_forward(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const void *nn_in, int N,
                    float *trans_dev /*[N,3]*/, float *rot_dev /*[N,3]*/, int shared_b = 0, const PoseUpdateFuse *fuse = nullptr,
                    bool *fused_out = nullptr);
int scorer_features(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const void *nn_in, int N,
                    float *feat_dev /*[N,512]*/);
int scorer_head(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const float *feats_dev, int n_total,
                float *scores_dev /*[n_total]*/);

}  // namespace fp
