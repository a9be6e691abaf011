// Refine-net / score-net on MFMA (fp_nn.hip): replaces the two opaque TensorRT engines of the reference
// (refiner_core_->SyncInfer foundationpose.cpp:206-208,254-256; scorer_core_->SyncInfer :218-220).
#pragma once

#include "fp_internal.h"

namespace fp {

struct Net;        // packed weights of one network, resident in HBM
struct NNScratch;  // activation buffers, grown on demand

// Reads a packed weight file (tools/pack_weights.py, "FPW1" format: fp32 tensors in PyTorch layout, BatchNorm
// already folded into conv weight+bias) and re-lays it out for the MFMA kernels (fp16, [Cout][KH][KW][Cin]).
Net *net_load(const char *path, bool is_scorer, std::string *err);
void net_free(Net *);

NNScratch *nn_scratch_create();
void nn_scratch_free(NNScratch *);
// debug (cross-model corruption checks): the activation arena and the f32 side buffer
void nn_scratch_debug_info(const NNScratch *, const void **buf, size_t *bytes, const void **f32, size_t *f32_bytes);

// Network input: nn_in = f16 [2N,80,80,32]: space-to-depth(2x2) view of the NHWC [2N,160,160,8] tensor
// (channels r,g,b,x,y,z,0,0), rendered crops A in images [0,N), observed crops B in [N,2N).
// Outputs are device pointers.
// shared_b != 0: all N hypotheses share ONE observed crop, stored as image N of nn_in (Register's first refine iteration)
int refiner_forward(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const __half *nn_in, int N,
                    float *trans_dev /*[N,3]*/, float *rot_dev /*[N,3]*/, int shared_b = 0);
int scorer_features(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const __half *nn_in, int N,
                    float *feat_dev /*[N,512]*/);
int scorer_head(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const float *feats_dev, int n_total,
                float *scores_dev /*[n_total]*/);

}  // namespace fp
