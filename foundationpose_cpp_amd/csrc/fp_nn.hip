// fp_nn.hip -- refine-net and score-net as hand-written CDNA4 (gfx950) kernels: fp16 storage, fp32 accumulate.
//
// Replaces the two opaque TensorRT fp16 engines of the reference (refiner_core_->SyncInfer / scorer_core_->SyncInfer,
// detection_6d_foundationpose/src/foundationpose.cpp:206-208,218-220,254-256; I/O blobs :78-83; shapes
// simple_tests/src/test_foundationpose.cpp:24-35).  The architecture is the published NVlabs FoundationPose one
// (SURVEY.md Appendix B [EXT]); the arithmetic oracle is oracle/nets_torch.py.
//
// Source layout [r4]: one translation unit (kernel templates must be visible where they are launched), four textual parts --
//   fp_nn_conv_kernels.inc       the convolution / Linear schedules        fp_nn_attention_kernels.inc   attention
//   fp_nn_small_kernels.inc      LayerNorm, token mean, small Linear ...   fp_nn_test_hooks.inc          fpt_* hooks (test build only)
// and this file: element types and helpers, weights (loading, layouts, 8-bit quantisation), scratch, schedule selection and launch
// (run_conv_dt), the two forward passes.
//
// Kernels (DESIGN.md section 4.2 has the why and the measurements)
//   All convolution / Linear schedules are the same contraction D^T[channel][pixel] = W * X^T on
//   v_mfma_f32_16x16x32_f16 over NHWC activations that carry a physical zero border: operand tiles are staged with
//   global_load_lds_dwordx4 (16 B/lane LDS-DMA, no VGPR round trip), LDS stays lane-linear and the XOR swizzle is applied
//   to the per-lane SOURCE chunk and to the ds_read_b128 fragment address (conflict-free); im2col never touches HBM.
//   Weight rows are permuted on the host so a lane's accumulators are 8 consecutive channels: bias + residual + ReLU (+
//   the a|b channel concat as an addressing mode) fuse into an epilogue of 16-byte stores (conv_epilogue_px).
//     conv_halo_kernel<40>      3x3/s1 on 40x40 maps: the (8+2)x(40+2) input tile of a 64-channel chunk resident in LDS,
//                               the 9 taps are shifted LDS windows, only weights stream; 2 workgroups per CU.
//     conv_stem_halo_kernel     the 7x7/s2 stem as a 4x4/s1 conv over the space-to-depth input, same resident-halo scheme.
//     conv_big_pp_kernel        256x256 implicit-GEMM tile, 8 waves in two ping-pong groups, hand-counted s_waitcnt /
//                               raw s_barrier; conv_512, the stride-2 convs, Linear layers (full rounds of the 256 CUs).
//     conv_s2_halo_kernel       the 3x3/s2 conv 64->128 on the 80x80 stem output: the input tile staged one column-parity
//                               plane at a time (a stride-2 tap is a stride-1 window of one plane).  HBM-bound layer.
//     conv_pp32_kernel<512,128> 32-wide K-steps, 4-stage ring; the same stride-2 conv below ~30 hypotheses.
//     conv_igemm_kernel<BN>     128 x BN tile, 2 workgroups per CU, optional split-K (+ conv_splitk_reduce_kernel) and
//                               weight groups along M: left-over rows, small batches (Track).
//     conv_smallx_kernel [r3]   small problems (Track, a few objects): one launch per layer, K split over the waves of a workgroup;
//                               weights global -> registers from a copy in MFMA-fragment order, pixels through a per-wave LDS-DMA
//                               ring -- both in the one address shape the vector L1 serves at full rate (tools/bench_tcp.hip).
//     conv_pp / conv_smallm kernels: earlier schedules kept behind fpt_set_conv_variant / fpt_set_smallm for A/B.
//   Weight layouts [r3]: besides the row-major [Cout][K] copy every layer carries the copies its schedules stream from -- fragment
//   order (conv_smallx), LDS-stage order for gemm_k32 / conv_halo / conv_halo8 / conv_big_pp / conv_deep + conv_pp (pack_stage_w,
//   pack_stage_w128): a wave's 2-4 LDS-DMA pieces of a stage are ONE contiguous run, so one address and one M0 serve them (the
//   instruction's immediate offset moves the global AND the LDS address) and every fetched cache line is used whole.
//   attention_kernel       softmax(QK^T/sqrt(d))V for 4 heads x 128, any sequence length (400 tokens per hypothesis, or
//                          the N hypotheses of the score-net's cross attention): S^T = K Q^T on MFMA so a softmax row is
//                          lane-local, P feeds the PV MFMA straight from registers (k-slot permutation shared with V^T).
//   layernorm / add_pos_embed / token_mean / small_linear / cast: bandwidth-trivial helpers.
#include "fp_nn.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <atomic>
#include <memory>
#include <thread>

namespace fp {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));  // 16 raw bytes: one ds_read_b128 / one LDS-DMA lane
typedef int i8 __attribute__((ext_vector_type(8)));  // 32 raw bytes: one fp8 operand of v_mfma_f32_16x16x128_f8f6f4

static constexpr int EMBED = 512, HEADS = 4, HDIM = 128;

// ---- element types ------------------------------------------------------------------------------
// DT_F16 / DT_BF16: 2-byte storage, v_mfma_f32_16x16x32_{f16,bf16}; a 128-byte LDS row holds 64 elements = two MFMA k-steps.
// DT_FP8 (OCP e4m3, BASELINE configs[4]): 1-byte storage, v_mfma_f32_16x16x128_f8f6f4 (unscaled form: both block
//   scales are the literal 0, which selects the instruction without the scale prefix); a 128-byte row holds 128 elements = ONE
//   MFMA k-step whose 32-byte operand is the concatenation of the two 16-byte reads the 2-byte types feed to their two
//   MFMAs (lane group kg owns chunks kg and 4+kg of the row: W and X use the same assignment, so the contraction is
//   unchanged and the LDS swizzles / bank-conflict analysis carry over).  Quantisation: per-output-channel weight scale,
//   per-tensor activation scale (static, from fp_calibrate_fp8), folded into the epilogue.
template <int DT> struct ElemT { using t = _Float16; using v8 = h8; using v4 = h4; };
template <> struct ElemT<DT_BF16> { using t = __bf16; using v8 = b8; using v4 = b4; };

template <int DT>
__device__ __forceinline__ f4 mfma32(i4 a, i4 b, f4 c) {  // one 16x16x32 step on 2-byte operands
  static_assert(DT == DT_F16 || DT == DT_BF16, "32-wide MFMA steps exist for the 2-byte types only");
  if constexpr (DT == DT_BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f4 mfma128_fp8(i8 a, i8 b, f4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /*A: fp8 e4m3*/, 0 /*B: fp8 e4m3*/, 0, 0, 0, 0);
}
// 8-bit operand types: DT_FP8 and DT_I8 share the byte geometry (a 128-byte LDS row = 128 channels = one 32-byte operand per lane
// group: chunks kg and 4+kg) and differ only in the matrix instruction.  DT_I8: the same 32 bytes feed TWO
// v_mfma_i32_16x16x64_i8 (16 cycles each = the 32 cycles of the one 128-wide FP8 instruction); the int32 accumulators live in
// the same f4 registers as raw bits (float 0.0 == int 0), the epilogue converts.
__host__ __device__ constexpr bool is_q8(int dt) { return dt == DT_FP8 || dt == DT_I8; }
template <int DT>
__device__ __forceinline__ f4 mfma8(i8 a, i8 b, f4 c) {
  static_assert(is_q8(DT), "8-bit operand types");
  if constexpr (DT == DT_FP8) return mfma128_fp8(a, b, c);
  else {
    i4 ci = __builtin_bit_cast(i4, c);
    ci = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_shufflevector(a, a, 0, 1, 2, 3), __builtin_shufflevector(b, b, 0, 1, 2, 3), ci, 0, 0, 0);
    ci = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_shufflevector(a, a, 4, 5, 6, 7), __builtin_shufflevector(b, b, 4, 5, 6, 7), ci, 0, 0, 0);
    return __builtin_bit_cast(f4, ci);
  }
}
// output-type codes (fp_nn.h): which tensors an epilogue writes
__host__ __device__ constexpr bool odt_dual(int odt) { return odt == DT_DUAL_FP8 || odt == DT_DUAL_I8; }
__host__ __device__ constexpr bool odt_qs(int odt) { return odt == DT_QS_FP8 || odt == DT_QS_I8 || odt == DT_QSR_I8; }
__host__ __device__ constexpr bool odt_rq(int odt) { return odt == DT_QSR_I8 || odt == DT_F16RQ_I8; }   // the residual operand is an 8-bit tensor
__host__ __device__ constexpr int odt_q(int odt) { return (odt == DT_DUAL_FP8 || odt == DT_QS_FP8) ? DT_FP8 : (odt == DT_DUAL_I8 || odt == DT_QS_I8 || odt == DT_QSR_I8) ? DT_I8 : is_q8(odt) ? odt : -1; }   // 8-bit type written, or -1
__host__ __device__ constexpr int odt_16(int odt) { return (odt_dual(odt) || odt == DT_F16RQ_I8) ? DT_F16 : (is_q8(odt) || odt_qs(odt)) ? -1 : odt; }   // 2-byte type written, or -1
// all MFMAs of one 128-byte K-step: w[ks][ni] / x[ks][mi] are the two 16-byte fragment reads of each row
template <int DT, int NI, int MI>
__device__ __forceinline__ void mma_kstep(f4 (&acc)[NI][MI], const i4 (&w)[2][NI], const i4 (&x)[2][MI]) {
  if constexpr (is_q8(DT)) {
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
      const i8 wv = __builtin_shufflevector(w[0][ni], w[1][ni], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int mi = 0; mi < MI; mi++)
        acc[ni][mi] = mfma8<DT>(wv, __builtin_shufflevector(x[0][mi], x[1][mi], 0, 1, 2, 3, 4, 5, 6, 7), acc[ni][mi]);
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int mi = 0; mi < MI; mi++) acc[ni][mi] = mfma32<DT>(w[ks][ni], x[ks][mi], acc[ni][mi]);
  }
}

// 8 consecutive channels <-> float; FP8 values are stored SCALED (real = stored * scale).  The raw form (16 bytes, or 8
// for FP8) is what stays in registers between the load and its use.
__device__ __forceinline__ i4 load8_raw(const unsigned char *ptr, int dt) {
  if (dt == DT_FP8 || dt == DT_I8) {
    const i2 v = *reinterpret_cast<const i2 *>(ptr);
    return (i4){v[0], v[1], 0, 0};
  }
  return *reinterpret_cast<const i4 *>(ptr);
}
__device__ __forceinline__ void decode8(i4 raw, int dt, float (&f)[8]) {
  if (dt == DT_FP8) {
    f[0] = __builtin_amdgcn_cvt_f32_fp8(raw[0], 0); f[1] = __builtin_amdgcn_cvt_f32_fp8(raw[0], 1);
    f[2] = __builtin_amdgcn_cvt_f32_fp8(raw[0], 2); f[3] = __builtin_amdgcn_cvt_f32_fp8(raw[0], 3);
    f[4] = __builtin_amdgcn_cvt_f32_fp8(raw[1], 0); f[5] = __builtin_amdgcn_cvt_f32_fp8(raw[1], 1);
    f[6] = __builtin_amdgcn_cvt_f32_fp8(raw[1], 2); f[7] = __builtin_amdgcn_cvt_f32_fp8(raw[1], 3);
  } else if (dt == DT_BF16) {
    const b8 v = __builtin_bit_cast(b8, raw);
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = (float)v[e];
  } else {
    const h8 v = __builtin_bit_cast(h8, raw);
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = (float)v[e];
  }
}
// element e (0..7) of a raw 8-channel group, as float (e is a constant once the caller's loop is unrolled)
template <int DT>
__device__ __forceinline__ float raw_elem(const i4 &raw, int e) {
  if constexpr (DT == DT_FP8) {
    const int w = raw[e >> 2];
    switch (e & 3) {
      case 0: return __builtin_amdgcn_cvt_f32_fp8(w, 0);
      case 1: return __builtin_amdgcn_cvt_f32_fp8(w, 1);
      case 2: return __builtin_amdgcn_cvt_f32_fp8(w, 2);
      default: return __builtin_amdgcn_cvt_f32_fp8(w, 3);
    }
  } else if constexpr (DT == DT_BF16) return (float)__builtin_bit_cast(b8, raw)[e];
  else return (float)__builtin_bit_cast(h8, raw)[e];
}
__device__ __forceinline__ float sat_fp8(float v) { return __builtin_amdgcn_fmed3f(v, -448.f, 448.f); }  // e4m3 finite range
__device__ __forceinline__ void store8(unsigned char *ptr, int dt, const float (&f)[8]) {
  if (dt == DT_FP8) {
    i2 v = {0, 0};
    v[0] = __builtin_amdgcn_cvt_pk_fp8_f32(sat_fp8(f[0]), sat_fp8(f[1]), v[0], false);
    v[0] = __builtin_amdgcn_cvt_pk_fp8_f32(sat_fp8(f[2]), sat_fp8(f[3]), v[0], true);
    v[1] = __builtin_amdgcn_cvt_pk_fp8_f32(sat_fp8(f[4]), sat_fp8(f[5]), v[1], false);
    v[1] = __builtin_amdgcn_cvt_pk_fp8_f32(sat_fp8(f[6]), sat_fp8(f[7]), v[1], true);
    *reinterpret_cast<i2 *>(ptr) = v;
  } else if (dt == DT_BF16) {
    b8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (__bf16)f[e];
    *reinterpret_cast<b8 *>(ptr) = v;
  } else {
    h8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (_Float16)f[e];
    *reinterpret_cast<h8 *>(ptr) = v;
  }
}
__host__ __device__ constexpr int elem_bytes(int dt) { return (dt == DT_FP8 || dt == DT_I8) ? 1 : 2; }

#include "fp_nn_conv_kernels.inc"
#include "fp_nn_attention_kernels.inc"
#include "fp_nn_small_kernels.inc"
#include "fp_nn_enc_kernels.inc"

// =================================================================================================
// weights
// =================================================================================================

struct HostTensor {
  std::vector<int> shape;
  std::vector<float> data;
};

static bool read_fpw(const char *path, std::map<std::string, HostTensor> &out, std::string *err) {
  FILE *f = std::fopen(path, "rb");
  if (!f) { *err = std::string("cannot open ") + path; return false; }
  char magic[4];
  uint32_t n = 0;
  bool ok = std::fread(magic, 1, 4, f) == 4 && std::memcmp(magic, "FPW1", 4) == 0 && std::fread(&n, 4, 1, f) == 1 && n <= 4096;
  for (uint32_t i = 0; ok && i < n; i++) {
    uint32_t ln = 0, nd = 0;
    ok = std::fread(&ln, 4, 1, f) == 1 && ln < 4096;
    std::string name(ok ? ln : 0, '\0');
    ok = ok && std::fread(&name[0], 1, ln, f) == ln && std::fread(&nd, 4, 1, f) == 1 && nd <= 8;
    HostTensor t;
    size_t cnt = 1;
    for (uint32_t d = 0; ok && d < nd; d++) {
      uint32_t s = 0;
      ok = std::fread(&s, 4, 1, f) == 1 && s <= (1u << 24);
      t.shape.push_back((int)s);
      cnt *= s;
      ok = ok && cnt <= ((size_t)1 << 28);  // no tensor of these networks exceeds 2^28 elements: a corrupt header must not drive a huge allocation
    }
    uint64_t nbytes = 0;
    ok = ok && std::fread(&nbytes, 8, 1, f) == 1 && nbytes == cnt * 4;
    if (ok) {
      t.data.resize(cnt);
      ok = std::fread(t.data.data(), 4, cnt, f) == cnt;
      out[name] = std::move(t);
    }
  }
  std::fclose(f);
  if (!ok) *err = std::string("malformed FPW1 file ") + path;
  return ok;
}

struct ConvLayer {
  unsigned char *w = nullptr;  // kernel layout, element type dt
  unsigned char *wfrag = nullptr;  // 2-byte types: a second copy in MFMA-fragment order for conv_smallm_kernel (fragment_order)
  unsigned char *wstep = nullptr;  // Linear layers: the fragment-order copy with the K-step OUTER (fragment_order_step_major; enc_tail_kernel, qkv_tile_kernel)
  unsigned char *wpack = nullptr;  // a copy in the LDS-stage order of gemm_k32_kernel (Linear layers) / conv_halo_kernel (3x3 layers) (pack_stage_w)
  unsigned char *wpack128 = nullptr;  // 3x3 layers with Cout % 256 == 0: a copy in conv_big_pp_kernel's stage order (pack_stage_w128)
  unsigned char *wdeep = nullptr;     // Cout % 128 == 0, 128-byte K-steps: a copy in conv_deep_kernel's stage order (FP8 3x3 layers: = wpack, the same order)
  float *bias = nullptr;
  float *cscale = nullptr;     // 8-bit layers: [Cout] dequantisation scale of the accumulator (weight-row scale, times the consumer's
                               // inverse activation scale when the output is 8-bit only: net_apply_q8)
  int dt = DT_F16;
  // 8-bit layers keep what net_apply_q8 needs to (re-)quantise them at calibration time: f32 rows [Cout][tap][Cin] and bias
  std::vector<float> rows_f32, bias_f32;
  std::vector<float> q_sw;     // per-row weight scale of the current quantisation
  std::vector<double> q_sum;   // DT_I8: per-row sum of the quantised weights
  float *tmat_t = nullptr;     // DT_I8 [r5]: [Cin][Cout] tap sums of the weights' rounding errors, T[co][c] = sum_taps (q - v), transposed (q8_img_bias_kernel)
  int ntaps = 0;
  int Cin = 0, Cout = 0, KH = 0, KW = 0, stride = 1, pad = 0;
  int algo_K = 0;  // algorithmic reduction length for FLOP accounting (the s2d stem pads 7x7x6=294 to 512)
};
struct LinearF32 {
  float *w = nullptr, *b = nullptr;
  int out = 0, in = 0;
};
struct LNParams {
  float *g = nullptr, *b = nullptr;
};
struct MHA {
  ConvLayer in_proj, out_proj;
  LinearF32 out_proj_f32;  // same weights in f32 for the "mean first" shortcut
};
struct EncLayer {  // transformer encoder layer (refiner heads)
  MHA att;
  ConvLayer lin1, lin2;
  LNParams ln1, ln2;
  LinearF32 head;
};

// trunk activation ids (FP8 scales / calibration): 0 stem, 1 a1, 2..5 encodeA blocks, 6..9 encodeAB 256 blocks, 10 b2,
// 11..14 encodeAB 512 blocks (14 = the token tensor, never quantised)
static constexpr int N_TRUNK_ACT = 15;
// trunk layer i (0..12, q8_layers order) -> its stage bit
static constexpr int q8_block_of(int i) { return i < 4 ? 1 : i < 8 ? 2 : i == 8 ? 4 : 8; }

struct Net {
  bool scorer = false;
  int prec = PREC_F16;
  int act_dt = DT_F16;               // element type of nn_in, of the token path and (non-FP8) of the trunk activations
  ConvLayer a0, a1, ra[2][2];        // encodeA
  ConvLayer rb[2][2], b2, rc[2][2];  // encodeAB
  EncLayer trans, rot;               // refiner
  // the two heads' Linear layers stored back to back ([2][Cout][K], [2][Cout]) for the one-launch small-batch path
  ConvLayer g_in_proj, g_out_proj, g_lin1, g_lin2;
  MHA att, att_cross;                // scorer
  LinearF32 score_lin;
  unsigned char *pe = nullptr;       // [400,512], act_dt
  // 8-bit networks (PREC_FP8 / PREC_INT8), set by net_apply_q8: per-CHANNEL activation scales of the trunk tensors that feed an
  // 8-bit convolution (real = stored * scale, DT_I8: real = (stored + 128) * scale); act_oinv = 1 / scale on the device for the
  // producers that write an f16 stream tensor together with its 8-bit copy (DT_DUAL_*); bias_fix / tok_fix = the bias correction
  // solved by the calibration sweeps (fp_api.hip: fp_calibrate)
  bool q8_ready = false;
  int qdt = DT_FP8;                                  // element type of the 8-bit layers
  int q8_blocks = 0;                                 // [r6] stages on 8-bit operands (bits: q8_block_of); 0 for the 2-byte networks
  // [r6] INT8: the per-image first-order compensation (run_trunk_q8: IB) belongs to records whose corrections were SOLVED with it on,
  // i.e. records that carry frame means (version 2, fp_calibrate_*).  A round-4 record or fp_set_calibration's per-tensor scales
  // (no frame means) were solved without it: compensating on top would subtract the same term twice.
  bool q8_img_comp = false;
  bool q8_on(int layer) const { return (q8_blocks & q8_block_of(layer)) != 0; }
  std::vector<float> act_scale[N_TRUNK_ACT];         // host, [channels of the activation]
  float *act_oinv[N_TRUNK_ACT] = {nullptr};          // device
  float *act_scale_dev[N_TRUNK_ACT] = {nullptr};     // device copy of act_scale (calibration statistics of 8-bit tensors)
  std::vector<float> bias_fix[13], tok_fix;          // host
  std::vector<float> out_bias0, out_fix;             // output-layer biases as loaded (refiner: trans 3 | rot 3; scorer: att.out_proj 512) and their correction
  std::vector<float> pe_host;                        // the positional table in f32 (tok_fix is added to the device copy)
  // calibration: per-channel |max| and sum of the 15 trunk activations collected on the device while the trunk runs
  // ([N_TRUNK_ACT][512] each); calib_mode 1 = |max| + sum (2-byte networks), 2 = sum only (8-bit networks, dequantised)
  float *calib_amax = nullptr;
  long long *calib_sum = nullptr;   // fixed point (2^-20): integer atomics are order-independent, so a calibration is reproducible bit for bit
  int calib_mode = 0;
  int calib_only = -1;                               // >= 0: record this activation only (the sequential correction sweeps)
  mutable double calib_count[N_TRUNK_ACT] = {0};     // interior pixels summed per activation (host side)
  std::vector<void *> allocs;
  // Two-phase loading [r5]: net_prepare reads the file and builds every layout on the HOST (no HIP call: it runs outside the
  // process-wide exclusive section); each device buffer it wants is recorded here as (address of the pointer field, bytes) and the
  // field holds a placeholder until net_commit carves one arena, uploads, and patches the fields.
  struct Pending {
    void **slot;
    std::vector<unsigned char> bytes;
  };
  std::vector<Pending> pending;
  bool deferred = false;
  const std::vector<unsigned char> *pending_bytes(const void *slot) const {
    for (const Pending &q : pending)
      if ((const void *)q.slot == slot) return &q.bytes;
    return nullptr;
  }
  ~Net() {
    for (void *p : allocs) (void)hipFree(p);
  }
};
static unsigned char g_placeholder;   // what a deferred pointer field holds between net_prepare and net_commit (never dereferenced)

template <typename T>
static T *upload(Net *net, const std::vector<T> &h) {
  if (net->deferred) return nullptr;   // (every load-time buffer goes through put(): the field's address is what net_commit patches)
  T *d = nullptr;
  if (hipMalloc((void **)&d, std::max<size_t>(h.size(), 1) * sizeof(T)) != hipSuccess) return nullptr;
  net->allocs.push_back(d);
  if (fp::memcpy_sync(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

// device copy of a host vector: allocated the first time, overwritten in place afterwards (re-calibration of an 8-bit network)
template <typename T>
static bool put(Net *net, T *&dst, const std::vector<T> &h) {
  if (net->deferred) {
    const unsigned char *b = reinterpret_cast<const unsigned char *>(h.data());
    for (Net::Pending &q : net->pending)
      if (q.slot == (void **)&dst) { q.bytes.assign(b, b + h.size() * sizeof(T)); return true; }
    net->pending.push_back({(void **)&dst, std::vector<unsigned char>(b, b + h.size() * sizeof(T))});
    dst = reinterpret_cast<T *>(&g_placeholder);
    return true;
  }
  if (!dst) { dst = upload(net, h); return dst != nullptr; }
  return fp::memcpy_sync(dst, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
}
static bool put_bytes(Net *net, unsigned char *&dst, const std::vector<unsigned char> &h) { return put<unsigned char>(net, dst, h); }

// ---- host-side element conversion (round to nearest even) ----
static uint16_t f32_to_bf16_bits(float x) {
  uint32_t u;
  std::memcpy(&u, &x, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// OCP FP8 e4m3fn: 1-4-3, bias 7, max 448, no infinities; saturating
static uint8_t f32_to_e4m3_bits(float x) {
  const uint8_t sign = std::signbit(x) ? 0x80 : 0;
  const float a = std::fabs(x);
  if (!(a == a)) return (uint8_t)(sign | 0x7f);
  if (a >= 448.f) return (uint8_t)(sign | 0x7e);
  if (a < 0x1p-10f) return sign;  // below half of the smallest subnormal (2^-9)
  int e;
  (void)std::frexp(a, &e);
  int E = e - 1;  // a in [2^E, 2^(E+1))
  if (E < -6) {   // subnormal: quantum 2^-9
    const int r = (int)std::nearbyint(a * 512.f);
    return (uint8_t)(sign | (r >= 8 ? 0x08 : r));
  }
  int r = (int)std::nearbyint(std::ldexp(a, 3 - E));  // 8..16
  if (r == 16) { r = 8; E++; }
  if (E > 8 || (E == 8 && r > 14)) return (uint8_t)(sign | 0x7e);
  return (uint8_t)(sign | ((E + 7) << 3) | (r - 8));
}
// rows of `K` floats -> kernel element bytes (2-byte types; the 8-bit layers go through quantise_q8)
static std::vector<unsigned char> to_elems(const std::vector<float> &w, int rows, int K, int dt, std::vector<float> *row_scale) {
  std::vector<unsigned char> o((size_t)rows * K * 2);
  if (row_scale) row_scale->assign(rows, 1.f);
  for (int r = 0; r < rows; r++) {
    const float *src = &w[(size_t)r * K];
    uint16_t *dst = reinterpret_cast<uint16_t *>(&o[(size_t)r * K * 2]);
    for (int k = 0; k < K; k++) {
      if (dt == DT_BF16) dst[k] = f32_to_bf16_bits(src[k]);
      else { __half h = __float2half(src[k]); std::memcpy(&dst[k], &h, 2); }
    }
  }
  return o;
}

// [a | b] copy of two equally shaped Linear layers (weights already in kernel row order), concatenated on the host from the
// layers' pending uploads (net_prepare)
static bool make_grouped(Net *net, const ConvLayer &a, const ConvLayer &b, ConvLayer *g) {
  if (a.Cin != b.Cin || a.Cout != b.Cout || a.dt != b.dt || !net->deferred) return false;
  auto cat = [&](const void *sa, const void *sb, std::vector<unsigned char> *out) {
    const std::vector<unsigned char> *pa = net->pending_bytes(sa), *pb = net->pending_bytes(sb);
    if (!pa || !pb || pa->size() != pb->size()) return false;
    *out = *pa;
    out->insert(out->end(), pb->begin(), pb->end());
    return true;
  };
  *g = a;
  g->w = nullptr; g->bias = nullptr; g->wfrag = nullptr; g->wpack = nullptr; g->wpack128 = nullptr; g->wdeep = nullptr;   // (the grouped launch never runs on gemm_k32_kernel)
  g->wstep = nullptr; g->tmat_t = nullptr; g->cscale = nullptr;   // (copied as placeholders from `a`: a host address no launch may ever see)
  std::vector<unsigned char> buf;
  if (!cat(&a.w, &b.w, &buf) || !put_bytes(net, g->w, buf)) return false;
  {
    std::vector<unsigned char> bb;
    if (!cat(&a.bias, &b.bias, &bb)) return false;
    std::vector<float> bf(bb.size() / 4);
    std::memcpy(bf.data(), bb.data(), bb.size());
    if (!put(net, g->bias, bf)) return false;
  }
  // (a group's fragment-order / deep-ring copy has the size of its row-major copy: the same group stride serves them)
  if (a.wfrag && b.wfrag && (!cat(&a.wfrag, &b.wfrag, &buf) || !put_bytes(net, g->wfrag, buf))) return false;
  if (a.wdeep && b.wdeep && (!cat(&a.wdeep, &b.wdeep, &buf) || !put_bytes(net, g->wdeep, buf))) return false;
  return true;
}

static bool get(const std::map<std::string, HostTensor> &m, const std::string &name, const HostTensor **t, std::string *err) {
  auto it = m.find(name);
  if (it == m.end()) { *err = "missing tensor " + name; return false; }
  *t = &it->second;
  return true;
}
static bool shape_is(const HostTensor *t, std::initializer_list<int> want) { return t->shape == std::vector<int>(want); }

// [Cout][tap][Cin] -> kernel K order: for Cin >= 64 [Cout][Cin/CH][tap][CH] with CH = the channels of a 128-byte chunk (64
// for the 2-byte types, 128 for FP8), otherwise unchanged.  `es` = element bytes.
static std::vector<unsigned char> relayout_k(const std::vector<unsigned char> &w, int Cout, int ntaps, int Cin, int es) {
  const int CH = 128 / es;
  if (Cin < 64 || ntaps == 1) return w;
  std::vector<unsigned char> o(w.size());
  const int nch = Cin / CH;
  for (int co = 0; co < Cout; co++)
    for (int t = 0; t < ntaps; t++)
      for (int c0 = 0; c0 < nch; c0++)
        std::memcpy(&o[((((size_t)co * nch + c0) * ntaps + t) * CH) * es], &w[(((size_t)co * ntaps + t) * Cin + (size_t)c0 * CH) * es], (size_t)CH * es);
  return o;
}

// Row permutation matching conv_epilogue: inside every block of 16*NI output channels (NI = 4, or 2 when Cout == 64)
// kernel row ni*16 + 4g + j holds the weights of channel 32*(j>>1) + 8g + 4*(j&1) + ni (NI=4) / 8g + 2j + ni (NI=2).
static std::vector<unsigned char> permute_rows(const std::vector<unsigned char> &w, int Cout, size_t row_bytes) {
  const int NI = (Cout % 128 == 0) ? 4 : 2, blk = 16 * NI;
  std::vector<unsigned char> o(w.size());
  for (int b0 = 0; b0 < Cout; b0 += blk)
    for (int ni = 0; ni < NI; ni++)
      for (int g = 0; g < 4; g++)
        for (int j = 0; j < 4; j++) {
          int ch = (NI == 4) ? 32 * (j >> 1) + 8 * g + 4 * (j & 1) + ni : 8 * g + 2 * j + ni;
          std::memcpy(&o[(size_t)(b0 + ni * 16 + 4 * g + j) * row_bytes], &w[(size_t)(b0 + ch) * row_bytes], row_bytes);
        }
  return o;
}

// Fragment order for conv_smallm_kernel, which loads its weight operands global -> registers: the vector L1 serves a wave's
// 16-byte loads four lanes at a time and takes one clock per DISTINCT cache line in each group of four (tools/bench_tcp.hip: a load in
// MFMA-operand shape -- lane -> row lane & 15 -- costs 64 clocks per wave instruction at any row pitch, a load of 1 KB of
// consecutive bytes 18).  So the bytes lane l needs for (16-row tile t, 128-byte K-step kt, half ks) are stored at
//     ((t * KT + kt) * 2 + ks) * 1024 + l * 16       <-  row t*16 + (l & 15), bytes kt*128 + ks*64 + (l >> 4)*16 .. +15
// Same size as the row-major copy; `w` is that copy ([Cout][row_bytes], rows already permuted).
static std::vector<unsigned char> fragment_order(const std::vector<unsigned char> &w, int Cout, size_t row_bytes) {
  const size_t KT = row_bytes / 128;
  std::vector<unsigned char> o(w.size());
  for (size_t t = 0; t < (size_t)Cout / 16; t++)
    for (size_t kt = 0; kt < KT; kt++)
      for (int ks = 0; ks < 2; ks++)
        for (int l = 0; l < 64; l++)
          std::memcpy(&o[((t * KT + kt) * 2 + ks) * 1024 + (size_t)l * 16], &w[(t * 16 + (l & 15)) * row_bytes + kt * 128 + ks * 64 + (size_t)(l >> 4) * 16], 16);
  return o;
}

// The same 1 KB fragments with the 64-byte K-step OUTER: fragment (16-row tile t, K-step s) at (s * Cout / 16 + t) KB.  The tile kernels of the
// Linear layers (enc_tail_kernel, qkv_tile_kernel: eight waves x four row tiles per K-step) read 32 CONSECUTIVE KB per step this way; in
// the tile-outer order the same 32 fragments lie 16 KB apart and fall on a few L2 channels, which every workgroup of an XCD -- they start
// together and stream the same bytes in the same order -- then hammers at the same time.
static std::vector<unsigned char> fragment_order_step_major(const std::vector<unsigned char> &w, int Cout, size_t row_bytes) {
  const size_t S = row_bytes / 64, T = (size_t)Cout / 16;
  std::vector<unsigned char> o(w.size());
  for (size_t t = 0; t < T; t++)
    for (size_t s = 0; s < S; s++)
      for (int l = 0; l < 64; l++)
        std::memcpy(&o[(s * T + t) * 1024 + (size_t)l * 16], &w[(t * 16 + (l & 15)) * row_bytes + s * 64 + (size_t)(l >> 4) * 16], 16);
  return o;
}

// Stage order for the kernels that stream a [TILE rows][64 B] weight stage per K-step through LDS-DMA (gemm_k32_kernel: TILE = 256,
// conv_halo_kernel: TILE = 128): the PW pieces (16 rows x 64 B each; lane -> row = lane >> 2, swizzled 16-byte chunk) wave `w` of a
// workgroup stages for (row tile nt, 64-byte K-step st) are stored as ONE run at ((nt * S + st) * 4 + w) * PW KB -- the DMA then
// needs one address and one M0 per PW pieces (glds16x4_asm / glds16x2_asm), and every fetched cache line is used whole.
static std::vector<unsigned char> pack_stage_w(const std::vector<unsigned char> &w, int Cout, size_t row_bytes, int TILE) {
  const size_t S = row_bytes / 64;
  const int PW = TILE / 64;
  std::vector<unsigned char> o(w.size());
  for (size_t nt = 0; nt < (size_t)Cout / TILE; nt++)
    for (size_t st = 0; st < S; st++)
      for (int wv = 0; wv < 4; wv++)
        for (int i = 0; i < PW; i++)
          for (int l = 0; l < 64; l++) {
            const int prow = l >> 2, gch = (l & 3) ^ ((0x78 >> (((prow >> 2) & 3) * 2)) & 3);
            std::memcpy(&o[(((nt * S + st) * 4 + wv) * PW + i) * 1024 + (size_t)l * 16],
                        &w[(nt * TILE + (size_t)(wv * PW + i) * 16 + prow) * row_bytes + st * 64 + (size_t)gch * 16], 16);
          }
  return o;
}

// The same for the kernels with 128-byte K-steps: conv_big_pp_kernel (TILE = 256 rows, NW = 8 waves) and the FP8 conv_halo8_kernel
// (TILE = 128, NW = 4); 4 pieces of 8 rows x 128 B per wave, lane -> row = lane >> 3, 16-byte chunk (lane & 7) ^ row: wave `w`'s 4 KB
// of (row tile nt, K-step kt) at ((nt * KT + kt) * NW + w) * 4096.  Byte-level, so it serves every element type.
static std::vector<unsigned char> pack_stage_w128(const std::vector<unsigned char> &w, int Cout, size_t row_bytes, int TILE, int NW) {
  const size_t KT = row_bytes / 128;
  std::vector<unsigned char> o(w.size());
  for (size_t nt = 0; nt < (size_t)Cout / TILE; nt++)
    for (size_t kt = 0; kt < KT; kt++)
      for (int wv = 0; wv < NW; wv++)
        for (int i = 0; i < 4; i++)
          for (int l = 0; l < 64; l++) {
            const int srow = l >> 3, g = (l & 7) ^ srow;
            std::memcpy(&o[(((nt * KT + kt) * NW + wv) * 4 + i) * 1024 + (size_t)l * 16],
                        &w[(nt * TILE + (size_t)(wv * 4 + i) * 8 + srow) * row_bytes + kt * 128 + (size_t)g * 16], 16);
          }
  return o;
}

// element bytes [Cout][tap][Cin] -> every device layout the layer's schedules stream from
static bool upload_layouts(Net *net, const std::vector<unsigned char> &elems, int Cout, int ntaps, int Cin, int dt, ConvLayer *L) {
  const int K = ntaps * Cin, es = elem_bytes(dt);
  const auto rows = permute_rows(relayout_k(elems, Cout, ntaps, Cin, es), Cout, (size_t)K * es);
  if (!put_bytes(net, L->w, rows)) return false;
  if (((size_t)K * es) % 128 == 0 && Cout % 16 == 0) {   // (byte-level: every element type)
    if (!put_bytes(net, L->wfrag, fragment_order(rows, Cout, (size_t)K * es))) return false;
    if (ntaps == 1 && es == 2 && !put_bytes(net, L->wstep, fragment_order_step_major(rows, Cout, (size_t)K * es))) return false;
  }
  if (!is_q8(dt) && ntaps == 1 && ((size_t)K * es) % 64 == 0 && Cout % 256 == 0) {
    if (!put_bytes(net, L->wpack, pack_stage_w(rows, Cout, (size_t)K * es, 256))) return false;      // gemm_k32_kernel
  } else if (!is_q8(dt) && ntaps == 9 && Cin % 64 == 0 && Cout % 128 == 0) {
    if (!put_bytes(net, L->wpack, pack_stage_w(rows, Cout, (size_t)K * es, 128))) return false;      // conv_halo_kernel
  }
  if (ntaps == 9 && ((size_t)K * es) % 128 == 0 && Cout % 256 == 0) {
    if (!put_bytes(net, L->wpack128, pack_stage_w128(rows, Cout, (size_t)K * es, 256, 8))) return false;   // conv_big_pp_kernel
  }
  if (is_q8(dt) && ntaps == 9 && ((size_t)K * es) % 128 == 0 && Cout % 128 == 0) {
    if (!put_bytes(net, L->wpack, pack_stage_w128(rows, Cout, (size_t)K * es, 128, 4))) return false;      // conv_halo8_kernel
    L->wdeep = L->wpack;                                                                                   // conv_deep_kernel: the same order
  } else if (((size_t)K * es) % 128 == 0 && Cout % 128 == 0) {
    if (!put_bytes(net, L->wdeep, pack_stage_w128(rows, Cout, (size_t)K * es, 128, 4))) return false;      // conv_deep_kernel
  }
  return true;
}

// 8-bit quantisation of a layer's rows with the per-INPUT-channel activation scales folded in first (w'[co][tap][ci] = w * s_in[ci];
// s_in = null: all 1): per output row a scale sw (FP8: amax / 448, OCP e4m3, RNE, saturating; I8: amax / 127) and, for I8, the
// row sum of the quantised integers (the -128 offset of the unsigned activations contributes 128 * sum to every accumulator).
//
// I8 rounding [r5]: ERROR-FEEDBACK rounding against the calibration frames' channel means (m_int: [J][Cin], the mean INTEGER
// activation of every input channel in each of J calibration frames; null = round to nearest).  Why: with round-to-nearest the
// weight errors d_k of a row are independent, so the row's mean output error sum_k d_k * mean(x_k) is a random walk over K = 1152-4608
// terms -- a common-mode shift of every hypothesis' features that the discriminating heads amplify, and that differs from scene to
// scene with the channel means (tools/q8_sim_cross.py: ALL of the cross-scene common-mode error of the INT8 trunk comes from the
// weights, none from the 8-bit activations; a bias correction removes it only for the calibration frames' own means).  Here every
// weight whose fraction lies within TAU of .5 (either neighbour costs almost the same rounding error) is rounded up or down so that
// the running sums r_j = sum_k d_k m_j[c(k)] stay near zero for EVERY frame j: the mean error cancels by construction, also for
// scenes whose channel means are (near) combinations of the calibration frames'.  Weights further from .5 round to nearest, which
// keeps the per-pixel (de-meaned) error where it was.
#ifdef FP_TEST_HOOKS
// [r5] A/B (fpt_set_rem_fork): the left-over rows of conv_512 / conv_b2 on a side stream next to the 256x256 rounds.  OFF: measured
// SLOWER (tools/ab_wall.py fpt_set_rem_fork 0 1 0 1: Register 10.85 -> 11.20 ms) -- a deep-ring workgroup needs a whole CU (147 KB of
// LDS), so it waits for a 256x256 tile to finish, then holds that CU out of the next round: the rounds lose their lock-step and end in
// a tail longer than the 35 us the lone launch took.
static int g_rem_fork = 0;
#else
static constexpr int g_rem_fork = 0;
#endif
#ifdef FP_TEST_HOOKS
static float g_q8_headroom = 1.25f;   // INT8 activation scale = |max| * headroom / 255 (tools/q8_multi.py --headroom)
// imgbias: the per-image first-order compensation (q8_img_bias_kernel); wclip / efr: the row-step search / rounding form of quantise_q8
static int g_q8_wclip = 1, g_q8_efr = 2, g_q8_imgbias = 1;   // A/B (tools/q8_multi.py --wq): the row-step search / the error-feedback rounding of quantise_q8
#else
static constexpr float g_q8_headroom = 1.25f;
static constexpr int g_q8_wclip = 1, g_q8_efr = 2, g_q8_imgbias = 1;
#endif
// [r6] which residual stages of the trunk an 8-bit network runs on 8-bit operands (the others keep their f16 weights and kernels):
// bit 0 = encodeA.2-3 (4 convs, 128 ch), bit 1 = encodeAB.0-1 (4 convs, 256 ch), bit 2 = encodeAB.2 (3x3 / s2, 256 -> 512),
// bit 3 = encodeAB.3-4 (4 convs, 512 ch).  A network keeps the mask it was loaded with (Net::q8_blocks).
static constexpr int Q8_BLOCKS_ALL = 15;
#ifdef FP_TEST_HOOKS
static int env_int(const char *name, int dflt) { const char *e = std::getenv(name); return e && *e ? std::atoi(e) : dflt; }
static int g_q8_blocks = env_int("FP_Q8_BLOCKS", Q8_BLOCKS_ALL);   // A/B (fpt_set_q8_blocks; tools/q8_blocks.py): networks loaded AFTER a change carry the new mask
#else
static constexpr int g_q8_blocks = Q8_BLOCKS_ALL;
#endif
#ifdef FP_TEST_HOOKS
static float env_float(const char *name, float dflt) { const char *e = std::getenv(name); return e && *e ? (float)std::atof(e) : dflt; }
static const float kEfrTau = env_float("FP_Q8_TAU", 0.2f), kEfrLam = env_float("FP_Q8_LAM", 0.05f);   // A/B of the rounding knobs (tools/q8_multi.py)
#else
static constexpr float kEfrTau = 0.2f, kEfrLam = 0.05f;
#endif
static std::vector<unsigned char> quantise_q8(const ConvLayer &L, int dt, const float *s_in, std::vector<float> *sw, std::vector<double> *qsum,
                                              const float *m_int = nullptr, int J = 0, std::vector<float> *tmat_t = nullptr) {
  const int K = L.ntaps * L.Cin;
  std::vector<unsigned char> o((size_t)L.Cout * K);
  if (tmat_t) tmat_t->assign((size_t)L.Cin * L.Cout, 0.f);
  sw->assign(L.Cout, 1.f);
  qsum->assign(L.Cout, 0.0);
  std::vector<float> wf(K);
  std::vector<double> r(std::max(J, 1));
  for (int co = 0; co < L.Cout; co++) {
    const float *src = &L.rows_f32[(size_t)co * K];
    float amax = 0.f;
    for (int k = 0; k < K; k++) {
      wf[k] = s_in ? src[k] * s_in[k % L.Cin] : src[k];
      amax = std::max(amax, std::fabs(wf[k]));
    }
    float sc = amax > 0.f ? amax / (dt == DT_FP8 ? 448.f : 127.f) : 1.f;
    if (dt == DT_I8 && m_int && J > 0 && amax > 0.f && g_q8_wclip) {
      // the row's step [r5]: the range that minimises the activation-weighted squared rounding + clipping error,
      // sum_k (q_k sc - w_k)^2 E_j[m_j,c(k)^2], over amax * {1, .95, ..., .5} / 127 -- a row's few largest folded weights (large weight on a
      // channel with a large activation scale) otherwise set the step of the thousands of ordinary ones
      std::vector<double> m2(L.Cin, 0.0);
      for (int j = 0; j < J; j++)
        for (int c = 0; c < L.Cin; c++) m2[c] += (double)m_int[(size_t)j * L.Cin + c] * m_int[(size_t)j * L.Cin + c] / J;
      double best = -1;
      float best_sc = sc;
      for (int step = 0; step <= 10; step++) {
        const float cand = amax * (1.f - 0.05f * step) / 127.f;
        double err = 0;
        for (int k = 0; k < K; k++) {
          const float q = std::max(-127.f, std::min(127.f, std::nearbyint(wf[k] / cand)));
          const double d = (double)q * cand - wf[k];
          err += d * d * (m2[k % L.Cin] + 1.0);
        }
        if (best < 0 || err < best) { best = err; best_sc = cand; }
      }
      sc = best_sc;
    }
    (*sw)[co] = sc;
    double qs = 0;
    std::fill(r.begin(), r.end(), 0.0);
    if (dt == DT_I8 && m_int && J > 0 && g_q8_efr == 2) {
      // ---- per input channel: the TAP SUM of the rounding errors [r5, second form] ---------------------------------------------
      // The mean output error of the row is sum_c T_c mean(x_c) with T_c = sum over the taps of channel c of its weights' rounding
      // errors (every tap of a channel sees the same mean).  Cancelling it against the calibration frames' means (first form) leaves
      // what a NEW scene's means add: mean(x_c) = m_c (1 + eps_c) with eps_c different per scene and channel, i.e. an error
      // sum_c T_c m_c eps_c that no finite set of calibration frames spans.  It is small for EVERY scene when every |T_c| is small:
      // round to nearest, then flip the weights closest to .5 (|fraction - .5| < TAU) while that brings |T_c| towards <= .5 (standard
      // deviation 0.86 -> 0.42 steps: not every channel has an eligible weight on the right side), and use the last free choice per channel (T_c or T_c -+ 1 when |T_c| is near .5) to keep the
      // running sums r_j = sum_c T_c m_jc of the calibration frames near zero.
      const int nt = L.ntaps, Cin = L.Cin;
      std::vector<int> qv(nt);
      std::vector<float> fr(nt);
      std::vector<char> can(nt);
      for (int c = 0; c < Cin; c++) {
        double T = 0;
        for (int t = 0; t < nt; t++) {
          const float v = wf[(size_t)t * Cin + c] / sc;
          const float base = std::floor(v);
          fr[t] = v - base;
          qv[t] = std::max(-127, std::min(127, (int)base + (fr[t] >= 0.5f ? 1 : 0)));
          can[t] = std::fabs(fr[t] - 0.5f) < kEfrTau && std::fabs(v) < 126.f;
          T += (double)qv[t] - (double)v;
        }
        auto flip_toward = [&](int dir) -> bool {   // dir = +1: raise T by one (flip a rounded-down weight up), -1: lower it; cheapest eligible weight
          int best = -1;
          float best_cost = 1e9f;
          for (int t = 0; t < nt; t++) {
            if (!can[t]) continue;
            const bool is_up = fr[t] >= 0.5f ? qv[t] == (int)std::floor(wf[(size_t)t * Cin + c] / sc) + 1 : false;
            const int cur = qv[t] - (int)std::floor(wf[(size_t)t * Cin + c] / sc);   // 0 = down, 1 = up
            (void)is_up;
            if (dir > 0 && cur != 0) continue;
            if (dir < 0 && cur != 1) continue;
            const float cost = std::fabs(fr[t] - 0.5f);
            if (cost < best_cost) { best_cost = cost; best = t; }
          }
          if (best < 0) return false;
          qv[best] += dir;
          can[best] = 0;   // a weight is moved at most once
          T += dir;
          return true;
        };
        while (T > 0.5 && flip_toward(-1)) {}
        while (T < -0.5 && flip_toward(+1)) {}
        // the free choice: T or T - sign(T) (|.| = 1 - |T|): take it when it serves the calibration frames' running sums more than it costs
        {
          const int dir = T > 0 ? -1 : +1;
          double now = 0, alt = 0, mbar2 = 0;
          for (int j = 0; j < J; j++) {
            const double m = m_int[(size_t)j * Cin + c];
            const double a = r[j] + T * m, b2 = r[j] + (T + dir) * m;
            now += a * a; alt += b2 * b2; mbar2 += m * m;
          }
          const double lam = (double)kEfrLam * mbar2;   // what a unit of T_c^2 costs on unseen scenes: (scene-to-scene variation of a channel mean ~ 20 %)^2 x m^2
          if (alt + lam * (T + dir) * (T + dir) < now + lam * T * T) {
            const double keep = T;
            if (!flip_toward(dir)) T = keep;
          }
        }
        for (int j = 0; j < J; j++) r[j] += T * m_int[(size_t)j * Cin + c];
        if (tmat_t) (*tmat_t)[(size_t)c * L.Cout + co] = (float)T;
        for (int t = 0; t < nt; t++) {
          o[(size_t)co * K + (size_t)t * Cin + c] = (unsigned char)(signed char)qv[t];
          qs += qv[t];
        }
      }
      (*qsum)[co] = qs;
      continue;
    }
    for (int k = 0; k < K; k++) {
      if (dt == DT_FP8) { o[(size_t)co * K + k] = f32_to_e4m3_bits(wf[k] / sc); continue; }
      const float v = wf[k] / sc;
      int q;
      if (m_int && J > 0) {
        const float base = std::floor(v), frac = v - base;
        bool up = frac >= 0.5f;
        const int c = k % L.Cin;
        const double e_dn = -(double)frac, e_up = 1.0 - (double)frac;
        if (std::fabs(frac - 0.5f) < kEfrTau && std::fabs(v) < 126.f) {
          double c_dn = 0, c_up = 0;
          for (int j = 0; j < J; j++) {
            const double m = m_int[(size_t)j * L.Cin + c];
            const double a = r[j] + e_dn * m, b = r[j] + e_up * m;
            c_dn += a * a; c_up += b * b;
          }
          up = c_up < c_dn;
        }
        q = std::max(-127, std::min(127, (int)base + (up ? 1 : 0)));
        const double e = (double)q - (double)v;   // (the error actually made: a clipped weight's is larger than one step)
        for (int j = 0; j < J; j++) r[j] += e * m_int[(size_t)j * L.Cin + c];
      } else {
        q = (int)std::max(-127.f, std::min(127.f, std::nearbyint(v)));
      }
      o[(size_t)co * K + k] = (unsigned char)(signed char)q;
      qs += q;
      if (tmat_t && dt == DT_I8) (*tmat_t)[(size_t)(k % L.Cin) * L.Cout + co] += (float)((double)q - (double)v);
    }
    (*qsum)[co] = qs;
  }
  return o;
}

// [Cout][KH][KW][Cin] f32 rows -> device layer of element type dt.  8-bit layers are quantised with unit activation scales here
// (so that every buffer exists) and again, with the calibrated scales, by net_apply_q8.
static bool finish_layer(Net *net, const std::vector<float> &rows_f32, const std::vector<float> &bias, int Cout, int ntaps, int Cin,
                         int dt, ConvLayer *L) {
  const int K = ntaps * Cin;
  L->dt = dt;
  L->ntaps = ntaps;
  if (is_q8(dt)) {
    L->rows_f32 = rows_f32;
    L->bias_f32 = bias;
    std::vector<float> sw;
    std::vector<double> qsum;
    const auto elems = quantise_q8(*L, dt, nullptr, &sw, &qsum);
    if (!upload_layouts(net, elems, Cout, ntaps, Cin, dt, L)) return false;
    return put(net, L->bias, bias) && put(net, L->cscale, sw);
  }
  if (!upload_layouts(net, to_elems(rows_f32, Cout, K, dt, nullptr), Cout, ntaps, Cin, dt, L)) return false;
  return put(net, L->bias, bias);
}

// PyTorch conv weight [Cout,Cin,KH,KW] -> kernel layout; the layer must have exactly the expected shape (the kernels and
// the arena carve hard-code the architecture: a foreign file is a load error, not an out-of-bounds launch)
static bool make_conv(Net *net, const std::map<std::string, HostTensor> &m, const std::string &prefix, int stride, int Cout,
                      int Cin, int k, int dt, ConvLayer *L, std::string *err) {
  const HostTensor *w, *b;
  if (!get(m, prefix + ".weight", &w, err) || !get(m, prefix + ".bias", &b, err)) return false;
  if (!shape_is(w, {Cout, Cin, k, k}) || !shape_is(b, {Cout})) { *err = prefix + ": unexpected weight / bias shape"; return false; }
  std::vector<float> hw((size_t)Cout * k * k * Cin);
  for (int co = 0; co < Cout; co++)
    for (int ci = 0; ci < Cin; ci++)
      for (int kh = 0; kh < k; kh++)
        for (int kw = 0; kw < k; kw++)
          hw[(((size_t)co * k + kh) * k + kw) * Cin + ci] = w->data[(((size_t)co * Cin + ci) * k + kh) * k + kw];
  L->Cin = Cin; L->Cout = Cout; L->KH = k; L->KW = k; L->stride = stride; L->pad = (k - 1) / 2;
  return finish_layer(net, hw, b->data, Cout, k * k, Cin, dt, L);
}

// 7x7 stride-2 pad-3 stem on [.,160,160,6] == 4x4 stride-1 pad-2 conv on the space-to-depth input [.,80,80,32]:
// w_s2d[co][a][b][(dy*2+dx)*8 + c] = w[co][c][2a+dy-1][2b+dx-1] (zero outside the 7x7 support / for c >= 6)
static bool make_stem(Net *net, const std::map<std::string, HostTensor> &m, const std::string &prefix, int dt, ConvLayer *L,
                      std::string *err) {
  const HostTensor *w, *b;
  if (!get(m, prefix + ".weight", &w, err) || !get(m, prefix + ".bias", &b, err)) return false;
  if (!shape_is(w, {64, 6, 7, 7}) || !shape_is(b, {64})) { *err = prefix + ": expected [64,6,7,7] stem weight"; return false; }
  std::vector<float> hw((size_t)64 * 16 * 32, 0.f);
  for (int co = 0; co < 64; co++)
    for (int a = 0; a < 4; a++)
      for (int bb = 0; bb < 4; bb++)
        for (int dy = 0; dy < 2; dy++)
          for (int dx = 0; dx < 2; dx++) {
            int kh = 2 * a + dy - 1, kw = 2 * bb + dx - 1;
            if (kh < 0 || kw < 0) continue;
            for (int c = 0; c < 6; c++)
              hw[(((size_t)co * 4 + a) * 4 + bb) * 32 + (dy * 2 + dx) * 8 + c] = w->data[(((size_t)co * 6 + c) * 7 + kh) * 7 + kw];
          }
  L->Cin = 32; L->Cout = 64; L->KH = 4; L->KW = 4; L->stride = 1; L->pad = 2; L->algo_K = 7 * 7 * 6;
  return finish_layer(net, hw, b->data, 64, 16, 32, dt, L);
}

// Linear [out,in] as a 1x1 conv
static bool make_linear_conv(Net *net, const std::map<std::string, HostTensor> &m, const std::string &wname,
                             const std::string &bname, int out, int in, int dt, ConvLayer *L, std::string *err) {
  const HostTensor *w, *b;
  if (!get(m, wname, &w, err) || !get(m, bname, &b, err)) return false;
  if (!shape_is(w, {out, in}) || !shape_is(b, {out})) { *err = wname + ": unexpected Linear shape"; return false; }
  L->Cout = out; L->Cin = in; L->KH = L->KW = 1; L->stride = 1; L->pad = 0;
  return finish_layer(net, w->data, b->data, out, 1, in, dt, L);
}

static bool make_linear_f32(Net *net, const std::map<std::string, HostTensor> &m, const std::string &wname,
                            const std::string &bname, int out, int in, LinearF32 *L, std::string *err) {
  const HostTensor *w, *b;
  if (!get(m, wname, &w, err) || !get(m, bname, &b, err)) return false;
  if (!shape_is(w, {out, in}) || !shape_is(b, {out})) { *err = wname + ": unexpected Linear shape"; return false; }
  L->out = out; L->in = in;
  return put(net, L->w, w->data) && put(net, L->b, b->data);
}

static bool make_ln(Net *net, const std::map<std::string, HostTensor> &m, const std::string &prefix, LNParams *L,
                    std::string *err) {
  const HostTensor *w, *b;
  if (!get(m, prefix + ".weight", &w, err) || !get(m, prefix + ".bias", &b, err)) return false;
  if (!shape_is(w, {EMBED}) || !shape_is(b, {EMBED})) { *err = prefix + ": unexpected LayerNorm shape"; return false; }
  return put(net, L->g, w->data) && put(net, L->b, b->data);
}

static bool make_mha(Net *net, const std::map<std::string, HostTensor> &m, const std::string &prefix, int dt, MHA *a,
                     std::string *err) {
  return make_linear_conv(net, m, prefix + ".in_proj_weight", prefix + ".in_proj_bias", 3 * EMBED, EMBED, dt, &a->in_proj, err) &&
         make_linear_conv(net, m, prefix + ".out_proj.weight", prefix + ".out_proj.bias", EMBED, EMBED, dt, &a->out_proj, err) &&
         make_linear_f32(net, m, prefix + ".out_proj.weight", prefix + ".out_proj.bias", EMBED, EMBED, &a->out_proj_f32, err);
}

static Net *net_load_impl(const char *path, bool is_scorer, int prec, std::string *err) {
  std::map<std::string, HostTensor> m;
  if (!read_fpw(path, m, err)) return nullptr;
  std::unique_ptr<Net> net(new Net());
  net->scorer = is_scorer;
  net->prec = prec;
  net->deferred = true;   // host phase: every put() records a pending upload (net_commit makes the device copies)
  // PREC_FP8 / PREC_INT8: the 3x3 trunk convolutions from encodeA.2 on (91 % of the FLOPs) run on 8-bit operands; the stem and
  // encodeA.1 (bandwidth-bound, K = 294 / 576) and the transformer part stay in f16
  const int adt = prec == PREC_BF16 ? DT_BF16 : DT_F16;
  const int tdt = prec == PREC_FP8 ? DT_FP8 : prec == PREC_INT8 ? DT_I8 : adt;
  net->act_dt = adt;
  net->qdt = tdt;
  net->q8_blocks = is_q8(tdt) ? (g_q8_blocks & Q8_BLOCKS_ALL) : 0;
  const auto ldt = [&](int layer) { return net->q8_on(layer) ? tdt : adt; };   // a stage outside the mask keeps 2-byte weights
  bool ok = make_stem(net.get(), m, "encodeA.0", adt, &net->a0, err) && make_conv(net.get(), m, "encodeA.1", 2, 128, 64, 3, adt, &net->a1, err);
  for (int i = 0; ok && i < 2; i++)
    for (int j = 0; ok && j < 2; j++) {
      std::string cj = ".conv" + std::to_string(j + 1);
      ok = make_conv(net.get(), m, "encodeA." + std::to_string(2 + i) + cj, 1, 128, 128, 3, ldt(0), &net->ra[i][j], err) &&
           make_conv(net.get(), m, "encodeAB." + std::to_string(i) + cj, 1, 256, 256, 3, ldt(4), &net->rb[i][j], err) &&
           make_conv(net.get(), m, "encodeAB." + std::to_string(3 + i) + cj, 1, 512, 512, 3, ldt(9), &net->rc[i][j], err);
    }
  ok = ok && make_conv(net.get(), m, "encodeAB.2", 2, 512, 256, 3, ldt(8), &net->b2, err);
  if (ok && !is_scorer) {
    EncLayer *heads[2] = {&net->trans, &net->rot};
    const char *names[2] = {"trans_head", "rot_head"};
    for (int i = 0; ok && i < 2; i++) {
      std::string p0 = std::string(names[i]) + ".0", p1 = std::string(names[i]) + ".1";
      ok = make_mha(net.get(), m, p0 + ".self_attn", adt, &heads[i]->att, err) &&
           make_linear_conv(net.get(), m, p0 + ".linear1.weight", p0 + ".linear1.bias", EMBED, EMBED, adt, &heads[i]->lin1, err) &&
           make_linear_conv(net.get(), m, p0 + ".linear2.weight", p0 + ".linear2.bias", EMBED, EMBED, adt, &heads[i]->lin2, err) &&
           make_ln(net.get(), m, p0 + ".norm1", &heads[i]->ln1, err) && make_ln(net.get(), m, p0 + ".norm2", &heads[i]->ln2, err) &&
           make_linear_f32(net.get(), m, p1 + ".weight", p1 + ".bias", 3, EMBED, &heads[i]->head, err);
    }
    if (ok) {
      ok = make_grouped(net.get(), net->trans.att.in_proj, net->rot.att.in_proj, &net->g_in_proj) &&
           make_grouped(net.get(), net->trans.att.out_proj, net->rot.att.out_proj, &net->g_out_proj) &&
           make_grouped(net.get(), net->trans.lin1, net->rot.lin1, &net->g_lin1) &&
           make_grouped(net.get(), net->trans.lin2, net->rot.lin2, &net->g_lin2);
      if (!ok) *err = "could not build the grouped head weights";
    }
  } else if (ok) {
    ok = make_mha(net.get(), m, "att", adt, &net->att, err) && make_mha(net.get(), m, "att_cross", adt, &net->att_cross, err) &&
         make_linear_f32(net.get(), m, "linear.weight", "linear.bias", 1, EMBED, &net->score_lin, err);
  }
  if (ok) {
    // PositionalEmbedding(d_model=512, max_len=400): pe[t,2i]=sin(t*w_i), pe[t,2i+1]=cos(t*w_i), w_i=exp(-2i*ln(1e4)/512)
    std::vector<float> pe((size_t)400 * EMBED);
    for (int t = 0; t < 400; t++)
      for (int i = 0; i < EMBED / 2; i++) {
        float div = std::exp((float)(2 * i) * -(std::log(10000.0f) / (float)EMBED));
        pe[(size_t)t * EMBED + 2 * i] = std::sin((float)t * div);
        pe[(size_t)t * EMBED + 2 * i + 1] = std::cos((float)t * div);
      }
    ok = put_bytes(net.get(), net->pe, to_elems(pe, 400, EMBED, adt, nullptr));
    net->pe_host = pe;
    if (!ok) *err = "device allocation failed";
  }
  if (!ok) return nullptr;
  return net.release();
}

static void q8_layers(Net *n, ConvLayer *(&L)[13]);
// Host phase of loading: the weight file is read and every device layout is built in host memory.  No HIP call, no device needed:
// callers run it OUTSIDE the process-wide exclusive section (fp_create, fp_set_precision, fp_calibrate: fp_api.hip).
Net *net_prepare(const char *path, bool is_scorer, int prec, std::string *err) {
  try {  // a malformed file must not take the process down through the C ABI
    return net_load_impl(path, is_scorer, prec, err);
  } catch (const std::exception &e) {
    *err = std::string("loading ") + path + ": " + e.what();
    return nullptr;
  }
}
// Device phase: ONE allocation (every buffer a 256-byte-aligned slice of it), ONE staged upload, the pointer fields patched.  On
// failure the network is unusable and the caller frees it.
int net_commit(Net *net, std::string *err) {
  if (!net->deferred) return 0;
  auto slice = [](const Net::Pending &q) { return (std::max<size_t>(q.bytes.size(), 1) + 255) & ~(size_t)255; };
  size_t total = 0;
  for (const Net::Pending &q : net->pending) total += slice(q);
  const size_t calib_off = total;
  total += (size_t)N_TRUNK_ACT * 512 * (sizeof(float) + sizeof(long long));
  unsigned char *arena = nullptr;
  if (hipMalloc((void **)&arena, total) != hipSuccess) { *err = "device allocation failed"; return 1; }
  net->allocs.push_back(arena);
  std::vector<unsigned char> image(total, 0);   // host image of the arena (the calibration statistics start zeroed)
  size_t off = 0;
  for (const Net::Pending &q : net->pending) {
    if (!q.bytes.empty()) std::memcpy(&image[off], q.bytes.data(), q.bytes.size());
    off += slice(q);
  }
  if (fp::memcpy_sync(arena, image.data(), total, hipMemcpyHostToDevice) != hipSuccess) { *err = "device upload failed"; return 1; }
  off = 0;
  for (Net::Pending &q : net->pending) {
    *q.slot = arena + off;
    off += slice(q);
  }
  net->calib_sum = reinterpret_cast<long long *>(arena + calib_off);
  net->calib_amax = reinterpret_cast<float *>(arena + calib_off + (size_t)N_TRUNK_ACT * 512 * sizeof(long long));
  // fields that ALIAS another field's buffer were copied while both held the placeholder: ConvLayer::wdeep = wpack (8-bit 3x3 layers)
  ConvLayer *q8l[13];
  q8_layers(net, q8l);
  for (ConvLayer *l : q8l)
    if (is_q8(l->dt) && l->wdeep == &g_placeholder) l->wdeep = l->wpack;
  net->pending.clear();
  net->pending.shrink_to_fit();
  net->deferred = false;
  return 0;
}

Net *net_load(const char *path, bool is_scorer, int prec, std::string *err) {
  Net *n = net_prepare(path, is_scorer, prec, err);
  if (n && net_commit(n, err)) { delete n; n = nullptr; }
  return n;
}

void net_free(Net *n) { delete n; }
int net_precision(const Net *n) { return n->prec; }
int net_input_dt(const Net *n) { return n->act_dt; }
bool net_q8_ready(const Net *n) { return !(n->prec == PREC_FP8 || n->prec == PREC_INT8) || n->q8_ready; }

// ---- calibration of the 8-bit networks ------------------------------------------------------------------
// Statistics: while calib_mode != 0 the trunk records, per channel of each of its 15 activations, |max| (mode 1) and the sum of
// the stored values (8-bit tensors: de-quantised) -- fp_api.hip turns the sums into means.
void net_calib_begin(Net *net, hipStream_t s, int mode, int only_act) {
  (void)hipMemsetAsync(net->calib_sum, 0, (size_t)N_TRUNK_ACT * 512 * (sizeof(float) + sizeof(long long)), s);
  net->calib_mode = mode;
  net->calib_only = only_act;
  for (int i = 0; i < N_TRUNK_ACT; i++) net->calib_count[i] = 0;
}
// sum_out: per-channel MEANS over the interior pixels of the recorded activations (sums / pixel count)
int net_calib_end(Net *net, hipStream_t s, float *amax_out /*[15][512] or null*/, float *sum_out /*[15][512]*/) {
  net->calib_mode = 0;
  net->calib_only = -1;
  if (amax_out) FP_HIP_OK(hipMemcpyAsync(amax_out, net->calib_amax, (size_t)N_TRUNK_ACT * 512 * sizeof(float), hipMemcpyDeviceToHost, s));
  std::vector<long long> sums(sum_out ? (size_t)N_TRUNK_ACT * 512 : 0);
  if (sum_out) FP_HIP_OK(hipMemcpyAsync(sums.data(), net->calib_sum, sums.size() * sizeof(long long), hipMemcpyDeviceToHost, s));
  FP_HIP_OK(hipStreamSynchronize(s));
  if (sum_out)
    for (int a = 0; a < N_TRUNK_ACT; a++)
      for (int c = 0; c < 512; c++)
        sum_out[a * 512 + c] = net->calib_count[a] > 0 ? (float)((double)sums[a * 512 + c] * (1.0 / 1048576.0) / net->calib_count[a]) : 0.f;
  return 0;
}
void net_calib_abort(Net *net) { net->calib_mode = 0; net->calib_only = -1; }   // a calibration pass that failed half-way
void net_q8_unready(Net *net) { net->q8_ready = false; }                          // the precision's record was dropped: quantise again before use
// the 13 8-bit layers in trunk order; layer i reads activation i + 1 and writes activation i + 2
static void q8_layers(Net *n, ConvLayer *(&L)[13]) {
  ConvLayer *l[13] = {&n->ra[0][0], &n->ra[0][1], &n->ra[1][0], &n->ra[1][1], &n->rb[0][0], &n->rb[0][1], &n->rb[1][0],
                      &n->rb[1][1], &n->b2, &n->rc[0][0], &n->rc[0][1], &n->rc[1][0], &n->rc[1][1]};
  for (int i = 0; i < 13; i++) L[i] = l[i];
}
// channels of trunk activation a (a = 1..14)
static int act_channels(int a) { return a <= 4 ? 128 : a <= 9 ? 256 : 512; }
// layer i's output goes ONLY to the next 8-bit convolution (a block's first conv): the consumer's scales fold into cscale / bias
static bool q8_folded_out(int i) { return i == 0 || i == 2 || i == 4 || i == 6 || i == 9 || i == 11; }
int net_q8_bias_channels(int layer) { return layer < 4 ? 128 : layer < 8 ? 256 : 512; }
bool net_q8_layer_on(const Net *n, int layer) { return n->q8_on(layer); }
int net_q8_blocks(const Net *n) { return n->q8_blocks; }

// One 8-bit layer: (weights) quantise its rows with the input-channel scales s_in folded in and upload every layout; then the epilogue
// tables.  accumulator -> real value: acc * sw (+ DT_I8: 128 * sw * sum_k q, the offset of the unsigned activations) + bias + fix;
// s_out_fold != null (the output is 8-bit ONLY): the consumer's inverse scales multiply both (legal: no residual, ReLU).
static int apply_q8_layer(Net *net, ConvLayer &l, int dt, const float *s_in, const float *bias_fix, const float *s_out_fold, bool weights,
                          const float *m_int = nullptr, int J = 0) {
  const int Cout = l.Cout;
  if (weights) {
    std::vector<float> sw;
    std::vector<double> qsum;
    std::vector<float> tmat;
    const auto elems = quantise_q8(l, dt, s_in, &sw, &qsum, dt == DT_I8 ? m_int : nullptr, J, dt == DT_I8 ? &tmat : nullptr);
    if (dt == DT_I8 && !put(net, l.tmat_t, tmat)) { set_error("net_apply_q8: device upload failed"); return 1; }
    if (!upload_layouts(net, elems, Cout, l.ntaps, l.Cin, dt, &l)) { set_error("net_apply_q8: device upload failed"); return 1; }
    l.q_sw = sw; l.q_sum = qsum;
  }
  std::vector<float> cs(Cout), bs(Cout);
  for (int co = 0; co < Cout; co++) {
    double b = (double)l.bias_f32[co] + (bias_fix ? bias_fix[co] : 0.f);
    if (dt == DT_I8) b += 128.0 * l.q_sw[co] * l.q_sum[co];
    double c = l.q_sw[co];
    if (s_out_fold) { const double inv = 1.0 / s_out_fold[co]; b *= inv; c *= inv; }
    cs[co] = (float)c; bs[co] = (float)b;
  }
  if (!put(net, l.cscale, cs) || !put(net, l.bias, bs)) { set_error("net_apply_q8: device upload failed"); return 1; }
  return 0;
}

// Applies a calibration to an 8-bit network: amax [15][512] = per-channel |max| of the trunk activations of the f16 network on
// the calibration frame; bias_fix [13][512] (may be null = zeros) and tok_fix [512] (may be null) = the solved bias correction.
// weights = false: only the biases / the positional table are rebuilt (the correction sweeps).
//   scale of activation a, channel c:  FP8: amax / 224 (one binade of headroom below the e4m3 maximum 448)
//                                      I8 : amax * 1.25 / 255 (unsigned 8-bit: every tensor here is a ReLU output)
//   (amax floored at 1/1024 of the tensor's largest channel)
//   the concat tensor (activation 5) gets the SAME scale for channel c of its a-half and its b-half, so that its producer (128
//   output channels, two image groups) indexes one table.
// frame_means [n_frames][15][512] (weights = true only; null / 0 = round to nearest): per-channel means of the f16 network's trunk
// activations in each calibration frame -- what the INT8 weights' error-feedback rounding cancels against (quantise_q8).
int net_apply_q8(Net *net, const float *amax, const float *bias_fix, const float *tok_fix, bool weights, const float *frame_means, int n_frames) {
  FP_CHECK(net->prec == PREC_FP8 || net->prec == PREC_INT8, "net_apply_q8: not an 8-bit network");
  const int dt = net->qdt;
  ConvLayer *L[13];
  q8_layers(net, L);
  if (weights) {
    for (int a = 1; a <= 13; a++) {
      const int C = act_channels(a);
      std::vector<float> sc(C);
      // a channel that is (almost) dead on the calibration frame gets the floor tensor-|max| / 1024, NOT a scale of 1: the scales are
      // folded into the consumer's weights, and one large scale would take the whole range of every weight row
      float tmax = 0.f;
      for (int c = 0; c < C; c++) tmax = std::max(tmax, amax[a * 512 + c]);
      if (!(tmax > 0.f)) tmax = 1.f;
      for (int c = 0; c < C; c++) {
        float m = amax[a * 512 + c];
        if (a == 5) m = std::max(amax[a * 512 + (c & 127)], amax[a * 512 + (c & 127) + 128]);
        m = std::max(m, tmax * (1.f / 1024.f));
        sc[c] = dt == DT_FP8 ? m / 224.f : m * g_q8_headroom / 255.f;
      }
      net->act_scale[a] = sc;
      std::vector<float> inv(C);
      for (int c = 0; c < C; c++) inv[c] = 1.f / sc[c];
      if (!put(net, net->act_oinv[a], inv) || !put(net, net->act_scale_dev[a], sc)) { set_error("net_apply_q8: device upload failed"); return 1; }
    }
  }
  FP_CHECK(!net->act_scale[1].empty(), "net_apply_q8: bias update before the scales were set");
  if (weights) net->q8_img_comp = dt == DT_I8 && frame_means != nullptr && n_frames > 0;
  for (int i = 0; i < 13; i++) {
    ConvLayer &l = *L[i];
    const int Cout = l.Cout;
    if (bias_fix) net->bias_fix[i].assign(bias_fix + (size_t)i * 512, bias_fix + (size_t)i * 512 + Cout);
    else if (net->bias_fix[i].empty()) net->bias_fix[i].assign(Cout, 0.f);
  }
  std::vector<float> m_int;
  for (int i = 0; i < 13; i++) {
    const int a = i + 1, Cin = L[i]->Cin;   // layer i reads activation i + 1
    if (!net->q8_on(i)) continue;           // [r6] a 2-byte layer of a partly 8-bit trunk: nothing to quantise
    const bool efr = weights && dt == DT_I8 && frame_means && n_frames > 0 && g_q8_efr;
    if (efr) {   // mean integer activation of every input channel per frame: f16 mean / scale
      m_int.assign((size_t)n_frames * Cin, 0.f);
      for (int j = 0; j < n_frames; j++)
        for (int c = 0; c < Cin; c++) m_int[(size_t)j * Cin + c] = frame_means[((size_t)j * N_TRUNK_ACT + a) * 512 + c] / net->act_scale[a][c];
    }
    if (apply_q8_layer(net, *L[i], dt, net->act_scale[i + 1].data(), net->bias_fix[i].data(),
                       q8_folded_out(i) ? net->act_scale[i + 2].data() : nullptr, weights, efr ? m_int.data() : nullptr, efr ? n_frames : 0)) return 1;
  }
  if (tok_fix) net->tok_fix.assign(tok_fix, tok_fix + EMBED);
  else if (net->tok_fix.empty()) net->tok_fix.assign(EMBED, 0.f);
  {  // positional table + token correction (the trunk's last epilogue adds the table to the rounded token)
    std::vector<float> pe(net->pe_host);
    for (int t = 0; t < 400; t++)
      for (int c = 0; c < EMBED; c++) pe[(size_t)t * EMBED + c] += net->tok_fix[c];
    const auto e = to_elems(pe, 400, EMBED, net->act_dt, nullptr);
    if (fp::memcpy_sync(net->pe, e.data(), e.size(), hipMemcpyHostToDevice) != hipSuccess) { set_error("net_apply_q8: device upload failed"); return 1; }
  }
  net->q8_ready = true;
  return 0;
}
// Output-layer correction of an 8-bit network: fix (refiner: [trans 3 | rot 3], scorer: [512] on the pooled score feature) is added to
// the biases of the f32 output layers (Linear(512,3) x 2 / att.out_proj) -- the last stage of the calibration's bias correction.
int net_q8_set_out_fix(Net *net, const float *fix) {
  const int n = net->scorer ? EMBED : 6;
  if (net->out_bias0.empty()) {
    net->out_bias0.resize(n);
    if (net->scorer) FP_HIP_OK(fp::memcpy_sync(net->out_bias0.data(), net->att.out_proj_f32.b, EMBED * 4, hipMemcpyDeviceToHost));
    else {
      FP_HIP_OK(fp::memcpy_sync(net->out_bias0.data(), net->trans.head.b, 12, hipMemcpyDeviceToHost));
      FP_HIP_OK(fp::memcpy_sync(net->out_bias0.data() + 3, net->rot.head.b, 12, hipMemcpyDeviceToHost));
    }
  }
  net->out_fix.assign(fix, fix + n);
  std::vector<float> b(n);
  for (int i = 0; i < n; i++) b[i] = net->out_bias0[i] + fix[i];
  if (net->scorer) FP_HIP_OK(fp::memcpy_sync(net->att.out_proj_f32.b, b.data(), EMBED * 4, hipMemcpyHostToDevice));
  else {
    FP_HIP_OK(fp::memcpy_sync(net->trans.head.b, b.data(), 12, hipMemcpyHostToDevice));
    FP_HIP_OK(fp::memcpy_sync(net->rot.head.b, b.data() + 3, 12, hipMemcpyHostToDevice));
  }
  return 0;
}
void net_q8_get_fix(const Net *net, float *bias_fix /*[13][512]*/, float *tok_fix /*[512]*/) {
  std::memset(bias_fix, 0, sizeof(float) * 13 * 512);
  std::memset(tok_fix, 0, sizeof(float) * 512);
  for (int i = 0; i < 13; i++)
    for (size_t c = 0; c < net->bias_fix[i].size(); c++) bias_fix[(size_t)i * 512 + c] = net->bias_fix[i][c];
  for (size_t c = 0; c < net->tok_fix.size(); c++) tok_fix[c] = net->tok_fix[c];
}

// =================================================================================================
// scratch
// =================================================================================================

struct NNScratch {
  int cap = 0;
  int q8 = 0;       // 0: 2-byte network; DT_FP8 / DT_I8: the arena also carries 1-byte slots for the 8-bit operand copies, whose zero
                    // border is the byte 0x00 (FP8) / 0x80 (I8: unsigned 0 stored with the offset of -128)
  unsigned char *buf = nullptr;
  float *f32 = nullptr;
  // cross-attention head over all gathered hypotheses (sized by n_total, independent of the local shard)
  int head_cap = 0;
  unsigned char *head_buf = nullptr;
  float *head_f32 = nullptr;
  // fp32 partial slabs of split-K convolutions (small batches); per model so that models on different streams /
  // threads never share it
  float *splitk = nullptr;
  size_t splitk_cap = 0;
  hipStream_t side = nullptr;                       // (Ctx::s2)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int *img_sum = nullptr;      // INT8 networks: [2 * cap][512] per-image channel sums (integer atomics; zero between uses) ...
  float *img_bias = nullptr;   // ... and the per-image bias made from them (q8_img_*_kernel)
  ~NNScratch() {
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (side) (void)hipStreamDestroy(side);
    if (img_sum) (void)hipFree(img_sum);
    if (img_bias) (void)hipFree(img_bias);
    if (splitk) (void)hipFree(splitk);
    if (buf) (void)hipFree(buf);
    if (f32) (void)hipFree(f32);
    if (head_buf) (void)hipFree(head_buf);
    if (head_f32) (void)hipFree(head_f32);
  }
};
NNScratch *nn_scratch_create(int prec) {
  NNScratch *w = new NNScratch();
  w->q8 = prec == PREC_FP8 ? DT_FP8 : prec == PREC_INT8 ? DT_I8 : 0;
  return w;
}

void nn_scratch_free(NNScratch *w) { delete w; }

// per-hypothesis activation sizes (BYTES at 2 bytes per element: one arena layout for every precision; an FP8 tensor uses
// the first half of its slot); conv inputs carry their physical zero border.  A scratch object serves ONE precision:
// the border positions of a tensor depend on its element size.
static constexpr size_t SZ_STEM = 2ull * 82 * 82 * 64 * 2;   // stem out, read by the 3x3/s2 conv (border 1)
static constexpr size_t SZ_128 = 2ull * 42 * 42 * 128 * 2;
static constexpr size_t SZ_256 = 42ull * 42 * 256 * 2;
static constexpr size_t SZ_512 = 22ull * 22 * 512 * 2;
static constexpr size_t SZ_TOK = 400ull * 512 * 2;            // token buffers (no border)
static constexpr size_t SZ_QKV = 400ull * 1536 * 2;
static constexpr size_t PER_HYP = SZ_STEM + 3 * SZ_128 + 3 * SZ_256 + 3 * SZ_512 + SZ_QKV + 4 * SZ_TOK;
static constexpr size_t PER_HYP_Q8 = (3 * SZ_128 + 3 * SZ_256 + 3 * SZ_512) / 2;   // the 1-byte copies (8-bit networks only)
static size_t per_hyp_bytes(const NNScratch *w) { return PER_HYP + (w->q8 ? PER_HYP_Q8 : 0); }
void nn_scratch_debug_info(const NNScratch *w, const void **buf, size_t *bytes, const void **f32, size_t *f32_bytes) {
  *buf = w->buf; *bytes = (size_t)w->cap * per_hyp_bytes(w);
  *f32 = w->f32; *f32_bytes = (size_t)w->cap * EMBED * sizeof(float);
}

static int ensure_scratch(NNScratch *ws, int N, hipStream_t s) {
  if (!ws->side && g_rem_fork) {   // (first call of a model is eager, never inside a capture)
    FP_HIP_OK(hipStreamCreateWithFlags(&ws->side, hipStreamNonBlocking));
    FP_HIP_OK(hipEventCreateWithFlags(&ws->ev_fork, hipEventDisableTiming));
    FP_HIP_OK(hipEventCreateWithFlags(&ws->ev_join, hipEventDisableTiming));
  }
  if (N <= ws->cap) return 0;
  if (ws->buf) (void)hipFree(ws->buf);
  if (ws->f32) (void)hipFree(ws->f32);
  ws->buf = nullptr; ws->f32 = nullptr; ws->cap = 0;
  int cap = std::max(N, 8);
  g_alloc_epoch++;
  FP_HIP_OK(hipMalloc((void **)&ws->buf, (size_t)cap * per_hyp_bytes(ws)));
  // [cap][512] pooled features + the arrival counter of token_mean_pose_kernel + [2][16][512] partial sums of layernorm_pmean_kernel (Track)
  FP_HIP_OK(hipMalloc((void **)&ws->f32, ((size_t)cap * EMBED + 16 + 2 * 16 * EMBED) * sizeof(float)));
  FP_HIP_OK(hipMemsetAsync(ws->f32 + (size_t)cap * EMBED, 0, 16 * sizeof(float), s));
  // the zero borders are written here once and never again: every producer stores interiors only, and the arena is
  // carved by CAPACITY (not by the current N), so an image slot's border never moves
  FP_HIP_OK(hipMemsetAsync(ws->buf, 0, (size_t)cap * PER_HYP, s));
  if (ws->q8) FP_HIP_OK(hipMemsetAsync(ws->buf + (size_t)cap * PER_HYP, ws->q8 == DT_I8 ? 0x80 : 0, (size_t)cap * PER_HYP_Q8, s));
  if (ws->q8 == DT_I8) {
    if (ws->img_sum) (void)hipFree(ws->img_sum);
    if (ws->img_bias) (void)hipFree(ws->img_bias);
    ws->img_sum = nullptr; ws->img_bias = nullptr;
    FP_HIP_OK(hipMalloc((void **)&ws->img_sum, (size_t)2 * cap * 512 * sizeof(int)));
    FP_HIP_OK(hipMemsetAsync(ws->img_sum, 0, (size_t)2 * cap * 512 * sizeof(int), s));
    FP_HIP_OK(hipMalloc((void **)&ws->img_bias, (size_t)2 * cap * 512 * sizeof(float)));
  }
  ws->cap = cap;
  return 0;
}

static int ensure_head_scratch(NNScratch *ws, int n_total) {
  if (n_total <= ws->head_cap) return 0;
  if (ws->head_buf) (void)hipFree(ws->head_buf);
  if (ws->head_f32) (void)hipFree(ws->head_f32);
  ws->head_buf = nullptr; ws->head_f32 = nullptr; ws->head_cap = 0;
  int cap = std::max(n_total, 256);
  g_alloc_epoch++;
  FP_HIP_OK(hipMalloc((void **)&ws->head_buf, (size_t)cap * 5 * EMBED * 2));
  FP_HIP_OK(hipMalloc((void **)&ws->head_f32, (size_t)cap * EMBED * sizeof(float)));
  ws->head_cap = cap;
  return 0;
}

// =================================================================================================
// launch helpers
// =================================================================================================

struct Ctx {
  hipStream_t s;
  Profiler *prof;
  const Net *net;
  NNScratch *ws = nullptr;  // owner of the split-K slab (null only in the single-threaded test hooks)
  // [r5] side stream for the left-over rows of a long-K layer (fork before the 256x256 rounds, join behind them): nullptr = off
  hipStream_t s2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

// A/B and ablation switches exist only in the test build (libfoundationpose_amd_test.so, -DFP_TEST_HOOKS); in the product
// they are compile-time constants, so the alternative branches and their kernel instantiations are not in the library
// and nothing can flip a schedule under a running model.
#ifdef FP_TEST_HOOKS
#define FP_HOOK static int
static unsigned long long *g_clk_probe = nullptr;
static NNScratch g_hook_ws;  // split-K slab of the fpt_* test hooks
#else
#define FP_HOOK [[maybe_unused]] static constexpr int
static constexpr unsigned long long *g_clk_probe = nullptr;
#endif
FP_HOOK g_conv_variant = 0;    // 0 default; 7 force / 8 disable the resident-halo kernels; 3 = 256x128 ping-pong everywhere; 5 = 256x256 rounds without
                               // the halo kernels; 10..25 = conv_igemm_kernel<128> fragment-read / ablation variants
FP_HOOK g_conv_ablate = 0;     // timing-only ablations (wrong results) of conv_big_pp_kernel / conv_halo_kernel
FP_HOOK g_rem_kernel = 3;      // rows the full 256x256 rounds of a long-K layer leave over: 3 = conv_deep_kernel<64>, 4 = <128>, 1 = 256x128 ping-pong, 0 = 128x128 2-stage
FP_HOOK g_rem_small = 1;       // long-K layers too small for one full 256x256 round take the left-over path as a whole
FP_HOOK g_gemm_kernel = 1;     // Linear layers on gemm_k32_kernel (0 = the 256x256 ping-pong tile + left-overs; 11 / 12 / 14 ablations)
FP_HOOK g_grouped_heads = 1;   // the refiner's two heads as one launch per layer when N == 1 (Track)
FP_HOOK g_rem_splitk = 0;      // split-K for left-over rows.  Measured -0.1 ms per Register, but OFF: a row's fp32 summation order would then
                               // depend on where it falls in the batch, and sharded and unsharded Register must pick the same near-tied winner
FP_HOOK g_splitk_target = 128; // workgroups a split-K launch aims for (tools/ab_track.py: 96-128 best, 256 is 6 % slower)
FP_HOOK g_splitk_min_kt = 9;    // layers with fewer 128-byte K-steps never split
FP_HOOK g_splitk_deep = 1;      // split-K slices of at least 4 K-steps on conv_deep_kernel<128> (0 = conv_igemm_kernel<128>)
FP_HOOK g_att_skv = 1;          // small attention grids on attention32_skv_kernel (keys split over the waves of a workgroup)
FP_HOOK g_gemm_deep = 1;         // short-K layers of small problems on conv_deep_kernel<128> instead of the two-stage 128x128 tile
FP_HOOK g_splitk_mid = 1;        // two split-K slices for long-K layers with 97..128 tiles (batches of ~8 objects)
FP_HOOK g_small_deep = 18;     // small problems (Track): conv_deep_kernel<64> over ALL K-steps instead of split-K + reduce when K has at most this many 128-byte steps
FP_HOOK g_smallm_maxkt = 80;   // the small-problem kernel takes layers with fewer 128-byte K-steps than this: every layer of both networks (72 for conv_512); 40 was the limit of its first version
FP_HOOK g_conv_lds_store = 0;  // conv_big_pp_kernel: epilogue stores staged through LDS (whole 128-byte lines per store instruction).  OFF: measured [r3] conv_512 3.205 -> 3.227 / 3.184 -> 3.180 ms, i.e. nothing -- the 256x256 tile's store burst is not bound by the store shape (unlike gemm_k32_kernel's, -6 %)
FP_HOOK g_gemm_lds_store = 1;  // gemm_k32_kernel: output rows leave through LDS as whole 256-byte runs instead of 64-byte pieces per store instruction
FP_HOOK g_smallm_maxt16 = 1024; // ... and with at most this many 16-pixel x 64-channel tiles (the grouped QKV of Track has 1200)
FP_HOOK g_fuse_pose = 1;       // Track: 1 = both Linear(512,3) heads + RefinePostProcess in one kernel (small_linear2_pose_kernel), 0 = two kernels, 2 = A/B: the token mean in that kernel too (token_mean_pose_kernel, last-arriver; not faster)
FP_HOOK g_gemm_wpack = 1;      // gemm_k32_kernel streams its weights from the stage-order copy (one address + one M0 per four LDS-DMA pieces)
FP_HOOK g_deep_wpack = 1;      // conv_deep_kernel streams its weights from the stage-order copy
FP_HOOK g_big_wpack = 1;       // conv_big_pp_kernel streams its weights from the stage-order copy
FP_HOOK g_halo_wpack = 1;      // conv_halo_kernel streams its weights from the stage-order copy
FP_HOOK g_att_tail = 0;        // [r5] A/B, OFF (measured slower: attention 0.555 -> 0.625 ms per Register): the 16-row tail of a 400-token sequence on attention32_skv_kernel instead of a 4th 128-row block
FP_HOOK g_ln_pmean = 1;        // [r5] Track: LayerNorm 2 + partial token sums in one launch (layernorm_pmean_kernel) instead of layernorm + token_mean
FP_HOOK g_qkv_ablate = 0;      // timing-only ablations of qkv_tile_kernel (test build, wrong results)
FP_HOOK g_qkv_tile = 1;        // [r5] QKV projections of Register (N > 1) on qkv_tile_kernel (80-token tiles resident in LDS) instead of gemm_k32_kernel
FP_HOOK g_enc_tail = 1;        // [r5] Register (N > 1): out_proj + LayerNorm 1 + FFN + LayerNorm 2 + token sums of BOTH heads as one launch (enc_tail_kernel) instead of five per head
FP_HOOK g_halo_wreg = 0;       // [r5] A/B, OFF (conv_256 -3 % in the stage profile, nothing on the wall clock: tools/ab_wall.py, EXPERIMENTS.md): 1 = 3x3 / 40x40 layers with >= 256 input channels on conv_halo_wreg_kernel: weights global -> registers (fragment-order copy), no weight ring, 2 barriers per chunk (2 = every such layer incl. the 128-channel ones, where it measures even)
FP_HOOK g_i8_stream = 0;       // test build A/B: 1 = INT8 networks with an 8-bit residual stream (run_trunk_i8; faster, but its common-mode error is frame-specific: DESIGN.md section 4.4)
FP_HOOK g_smallx_pf = 0;       // A/B (test build): prefetch depth of conv_smallx_kernel<2,4> (4 or 2; 0 = the default 3)
FP_HOOK g_smallm = 1;          // small problems (Track, a few objects) on conv_smallm_kernel: K split over the waves of a workgroup, no split-K slabs / reduce launch
FP_HOOK g_att_variant = 1;     // 1 = attention32_kernel (8 = without the XCD remap); round-1 kernel: 2 remap + 16-B stores, 3 no XCD remap, 5 remap + 2-B stores, 7 neither

// One launch = (once per launch site, element type and DEVICE) dynamic-LDS opt-in + the launch itself.
#define FP_LAUNCH(KERN, grid, block, lds_bytes, stream, ...)                                                                        \
  do {                                                                                                                              \
    static fp::PerDeviceOnce fp_attr_once_;                                                                                         \
    fp_attr_once_.run([] { (void)hipFuncSetAttribute((const void *)(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
    hipLaunchKernelGGL((KERN), grid, block, lds_bytes, stream, __VA_ARGS__);                                                        \
  } while (0)

// a tensor as run_conv sees it: element type + (FP8) per-tensor scale, real = stored * scale
struct Act {
  void *p = nullptr;
  int dt = DT_F16;
  float scale = 1.f;
};

// in: [NB, H+2*ipad, W+2*ipad, Cin]; out: [.., OH+2*opad, OW+2*opad, ..]; res: border rpad
struct ConvGroup {  // two weight groups along M (see ConvParams::grp_rows); L holds [2][Cout][K] weights and [2][Cout] biases
  int rows = 0;     // rows per group (multiple of 128); the launch covers 2 * rows
  bool in_shared = false, res_shared = false;
};

// DT = element type of the layer's operands; ODT = of its output.  ODT != DT only for the two layers at the FP8 boundary
// (encodeA.1: f16 -> FP8, the last encodeAB conv: FP8 -> f16 tokens); those instantiate only the schedules they can reach.
template <int DT, int ODT>
static int run_conv_dt(const Ctx &c, const char *tag, const ConvLayer &L, ConvParams &p, int NB, int H, int W, int ipad,
                       bool has_res, int split_imgs, const ConvGroup *grp) {
  constexpr bool B2 = !is_q8(DT);    // 2-byte element type: the 64-byte-row kernels exist
  constexpr bool SAME = DT == ODT;   // kernels without an ODT parameter write their operand type
  constexpr bool QOUT = odt_q(ODT) >= 0;   // an 8-bit tensor is written (alone or next to the f16 stream tensor)
  const int KT = p.krow_b / 128;
  bool post_main = false;  // ConvParams::post handled by the 256x256 rounds + deep-ring left-over (see below)
  bool fork_rem = false;   // the left-over launch goes to the side stream (forked before the 256x256 rounds)
  double flops = 2.0 * (double)p.M * p.Cout * (L.algo_K > 0 ? L.algo_K : p.Ktot);
  double bytes = ((double)NB * H * W * L.Cin + (double)p.M * p.Cout * (has_res ? 2 : 1) + (double)p.Cout * p.Ktot) * elem_bytes(DT);
  constexpr int LDS_IG128 = 2 * (128 * 128 + 128 * 128), LDS_IG64 = 2 * (128 * 128 + 64 * 128);
  constexpr int LDS3_128 = 3 * (256 * 128 + 128 * 128);
  constexpr int LDS_BIG = 2 * (256 * 128 + 256 * 128);
  constexpr int LDS_HALO40 = ((10 * 42 + 7) / 8) * 1024 + 3 * 128 * 64;
  [[maybe_unused]] constexpr int LDS_HALO40W = ((10 * 42 + 7) / 8) * 1024;
  constexpr int LDS_HALO8 = ((10 * 42 + 7) / 8) * 1024 + 128 * 128;
  constexpr int LDS_STEM_HALO = ((11 * 84 + 15) / 16) * 1024 + 3 * 64 * 64;
  constexpr int LDS_DEEP64 = 6 * (64 + 128) * 128, LDS_DEEP128 = 4 * (128 + 128) * 128;
  constexpr int LDS_GEMM_K32 = 3 * (128 + 256) * 64;
  constexpr int LDS_S2_HALO = ((9 * 41 + 7) / 8) * 1024 + 3 * 128 * 64;
  constexpr int LDS_PP32 = 4 * (512 + 128) * 64;
  int mtiles = (p.M + 127) / 128;
  // split-K for small problems (Track, N <= ~8): a 128x128 tile count far below the 512 workgroup slots of the chip
  // would leave most CUs idle while a few walk up to 72 K-steps; give every CU a slice instead
  // (also used for the rows a 256x256 / 512x128 launch leaves over: `target` workgroups on an otherwise idle chip)
  auto plan_splitk = [&](int rows, int target) -> int {
    const int mt = (rows + 127) / 128;
    const int tiles = mt * (L.Cout % 128 == 0 ? L.Cout / 128 : L.Cout / 64);
    if (KT < g_splitk_min_kt || is_q8(DT)) return 0;
    if (tiles > 96) {
      // a little above the split-K range (a batch of ~8 objects): long-K layers still leave half the chip idle for 72 K-steps;
      // two slices per tile while the grid fits one round of the 256 CUs
      if (!(g_splitk_mid && tiles <= 128 && KT >= 36)) return 0;
      target = 2 * tiles;
    }
    int S = std::min(std::max(target / tiles, 1), KT / 2);
    if (S <= 1) return 0;
    p.kt_per = (KT + S - 1) / S;
    p.ksplit = (KT + p.kt_per - 1) / p.kt_per;
    size_t need = (size_t)p.ksplit * rows * p.Cout;
#ifdef FP_TEST_HOOKS
    NNScratch *sk = c.ws ? c.ws : &g_hook_ws;
#else
    NNScratch *sk = c.ws;
#endif
    if (need > sk->splitk_cap) {
      if (sk->splitk) (void)hipFree(sk->splitk);
      sk->splitk = nullptr; sk->splitk_cap = 0;
      g_alloc_epoch++;
      FP_HIP_OK(hipMalloc((void **)&sk->splitk, need * sizeof(float)));
      sk->splitk_cap = need;
    }
    p.partial = sk->splitk;
    return 0;
  };
  // ---- small problems: one launch per layer, the K-steps split over the four waves of a workgroup (conv_smallx_kernel)
  if (B2 || L.Cout % 128 == 0) {                         // (FP8 layers all have Cout % 128 == 0: 64-channel tiles only)
    const int cw = (L.Cout % 128 == 0) ? 64 : 32;         // channels per workgroup = the block of the host-side row permutation
    const int t16 = ((p.M + 15) / 16) * (L.Cout / cw);
    // Measured on Track (tools/profile_track.sh, profiles/r03g_track_timeline.txt): 5.6-7.2 us for the layers of up to 18 K-steps, 8.5-9.1
    // for the 36-step ones, 10.2-10.6 for conv_512 (72 steps; 12.8 + 5.5 us as split-K + reduce).  The first version
    // (conv_smallm_kernel: both operands global -> registers in MFMA-operand shape, 64 clocks of the vector L1 per instruction) lost on
    // the long-K layers (21 us); g_smallm = 3 selects it for A/B, g_smallm = 2 forces the small-problem kernel for every size.
    if (g_smallm && L.wfrag && (KT < g_smallm_maxkt || g_smallm >= 2) && g_conv_variant == 0 && g_conv_ablate == 0 && L.Cout % cw == 0 && t16 <= g_smallm_maxt16 * (64 / cw) &&
        (!grp || grp->rows % 32 == 0) && ((L.Cout / cw) % 8 == 0 || 8 % (L.Cout / cw) == 0)) {
      const int t32 = ((p.M + 31) / 32) * (L.Cout / cw);
      const bool two = t32 >= 160;                        // 32-pixel tiles halve the weight stream once they still fill the chip
      bool post = p.post != nullptr;
      if constexpr (QOUT) post = false;
      if (!post) p.post = nullptr;
      ProfScope ps(c.prof, c.s, (std::string(tag) + "/conv_smallm_kernel").c_str(), flops, bytes);
      // grid = 8 XCD lanes x ceil(workgroups / 8), see the kernel's placement rule
      const int ntl = L.Cout / cw, mtl = two ? (p.M + 31) / 32 : (p.M + 15) / 16;
      const int per_xcd = ntl >= 8 ? mtl * (ntl / 8) : (mtl + 8 / ntl - 1) / (8 / ntl);
      const dim3 grid(8 * per_xcd);
      const bool deep = KT >= 32;                          // every wave has >= 8 K-steps >= 2 * PF (PF = 4 / 3)
#ifdef FP_TEST_HOOKS   // A/B (g_smallm == 3): the first version, input fragments global -> registers in operand shape
#define FP_SMALLM_DIRECT(MI_, NI_, POST_)                                                                                     \
    if (g_smallx_pf && MI_ == 2 && NI_ == 4 && !POST_ && DT == DT_F16 && ODT == DT_F16) {                                     \
      if (g_smallx_pf == 4) FP_LAUNCH((conv_smallx_kernel<2, 4, DT_F16, DT_F16, false, 4>), grid, dim3(256), 4 * 4 * 2 * 2048 + 3 * 4 * 2 * 1024, c.s, p); \
      else FP_LAUNCH((conv_smallx_kernel<2, 4, DT_F16, DT_F16, false, 2>), grid, dim3(256), 4 * 2 * 2 * 2048 + 3 * 4 * 2 * 1024, c.s, p); \
      break;                                                                                                                  \
    }                                                                                                                         \
    if (g_smallm == 3) {                                                                                                      \
      if (deep) FP_LAUNCH((conv_smallm_kernel<MI_, NI_, DT, ODT, POST_, true>), grid, dim3(256), 3 * NI_ * MI_ * 1024, c.s, p); \
      else FP_LAUNCH((conv_smallm_kernel<MI_, NI_, DT, ODT, POST_, false>), grid, dim3(256), 3 * NI_ * MI_ * 1024, c.s, p);    \
      break;                                                                                                                  \
    }
#else
#define FP_SMALLM_DIRECT(MI_, NI_, POST_)
#endif
#define FP_SMALLM(MI_, NI_, POST_)                                                                                          \
  do {                                                                                                                      \
    if constexpr (B2) { FP_SMALLM_DIRECT(MI_, NI_, POST_) }                                                                 \
    FP_LAUNCH((conv_smallx_kernel<MI_, NI_, DT, ODT, POST_>), grid, dim3(256),                                              \
              4 * ((MI_) == 1 ? 4 : 3) * (MI_) * 2048 + 3 * NI_ * MI_ * 1024, c.s, p);                                      \
  } while (0)
      if constexpr (!QOUT) {
        if (post && cw == 64) { if (two) FP_SMALLM(2, 4, true); else FP_SMALLM(1, 4, true); return 0; }
        if (post) p.post = nullptr;                      // (no 32-channel layer carries a positional table)
      }
      if (cw == 64) { if (two) FP_SMALLM(2, 4, false); else FP_SMALLM(1, 4, false); }
      else if constexpr (B2) { if (two) FP_SMALLM(2, 2, false); else FP_SMALLM(1, 2, false); }
#undef FP_SMALLM
#undef FP_SMALLM_DIRECT
      return 0;
    }
  }
  const bool small_deep = !grp && g_small_deep > 0 && KT >= 4 && KT <= g_small_deep && L.Cout % 128 == 0 && ((p.M + 63) / 64) * (L.Cout / 128) >= 40 &&
                          ((p.M + 63) / 64) * (L.Cout / 128) <= 256;
  if (!small_deep && plan_splitk(p.M, g_splitk_target)) return 1;
  const std::string tg(tag);
  const bool halo_ok = g_conv_variant == 0 || g_conv_variant == 7;
  const bool force = g_conv_variant == 7;
  if constexpr (B2 && SAME) {
    if (!grp && halo_ok && L.Cin == 32 && L.KH == 4 && L.KW == 4 && L.Cout == 64 && ipad == 2 && W == 80 && H == 80 && p.ksplit == 1 &&
        !has_res && split_imgs == 0 && (force || NB * 10 >= 300)) {
      ProfScope ps(c.prof, c.s, (tg + "/conv_stem_halo_kernel").c_str(), flops, bytes);
      FP_LAUNCH((conv_stem_halo_kernel<DT>), dim3(NB * 10), dim3(256), LDS_STEM_HALO, c.s, p);
      return 0;
    }
    if (!grp && g_gemm_kernel && g_conv_variant == 0 && L.KH == 1 && L.KW == 1 && L.stride == 1 && L.pad == 0 && ipad == 0 && L.Cout % 256 == 0 &&
        p.Ktot % 32 == 0 && p.ksplit == 1 && split_imgs == 0 && ((p.M + 127) / 128) * (L.Cout / 256) >= 512) {
      ProfScope ps(c.prof, c.s, (tg + "/gemm_k32_kernel").c_str(), flops, bytes);
      const dim3 grid(((p.M + 127) / 128) * (L.Cout / 256));
#ifdef FP_TEST_HOOKS
      if (DT == DT_F16 && g_gemm_kernel == 11) { FP_LAUNCH((gemm_k32_kernel<1, DT_F16>), grid, dim3(256), LDS_GEMM_K32, c.s, p); return 0; }
      if (DT == DT_F16 && g_gemm_kernel == 12) { FP_LAUNCH((gemm_k32_kernel<2, DT_F16>), grid, dim3(256), LDS_GEMM_K32, c.s, p); return 0; }
      if (DT == DT_F16 && g_gemm_kernel == 14) { FP_LAUNCH((gemm_k32_kernel<4, DT_F16>), grid, dim3(256), LDS_GEMM_K32, c.s, p); return 0; }
#endif
      if (g_gemm_lds_store && g_gemm_wpack && L.wpack) FP_LAUNCH((gemm_k32_kernel<0, DT, true, true>), grid, dim3(256), LDS_GEMM_K32, c.s, p);
      else if (g_gemm_lds_store) FP_LAUNCH((gemm_k32_kernel<0, DT, true>), grid, dim3(256), LDS_GEMM_K32, c.s, p);
      else FP_LAUNCH((gemm_k32_kernel<0, DT>), grid, dim3(256), LDS_GEMM_K32, c.s, p);
      return 0;
    }
  }
  if constexpr (B2) {
    if (!grp && halo_ok && L.KH == 3 && L.KW == 3 && L.stride == 2 && L.pad == 1 && ipad == 1 && W == 80 && H == 80 && L.Cin == 64 &&
        L.Cout == 128 && p.ksplit == 1 && !has_res && split_imgs == 0 && (force || NB * 10 >= 300)) {
      ProfScope ps(c.prof, c.s, (tg + "/conv_s2_halo_kernel").c_str(), flops, bytes);
      FP_LAUNCH((conv_s2_halo_kernel<DT, ODT>), dim3(NB * 10), dim3(256), LDS_S2_HALO, c.s, p);
      return 0;
    }
  }
  // 3x3 / stride 1 on 40x40 maps with the input tile resident in LDS; measured crossover vs the implicit-GEMM tiles: ~32 hypotheses
  if ((SAME || (!B2 && odt_q(ODT) == DT)) && !grp && halo_ok && L.KH == 3 && L.KW == 3 && L.stride == 1 && L.pad == 1 && ipad == 1 && W == 40 && H % 8 == 0 &&
      p.cin_b % 128 == 0 && L.Cout % 128 == 0 && p.ksplit == 1 && (force || NB * (H / 8) * (L.Cout / 128) >= 300)) {
    const dim3 grid(NB * (H / 8) * (L.Cout / 128));
    if constexpr (B2 && !SAME) {
    } else if constexpr (B2) {
      ProfScope ps(c.prof, c.s, (tg + "/conv_halo_kernel").c_str(), flops, bytes);
#ifdef FP_TEST_HOOKS
      if (DT == DT_F16 && g_conv_ablate == 1) { FP_LAUNCH((conv_halo_kernel<40, 1, DT_F16>), grid, dim3(256), LDS_HALO40, c.s, p); return 0; }
      if (DT == DT_F16 && g_conv_ablate == 2) { FP_LAUNCH((conv_halo_kernel<40, 2, DT_F16>), grid, dim3(256), LDS_HALO40, c.s, p); return 0; }
      if (DT == DT_F16 && g_conv_ablate == 8) { FP_LAUNCH((conv_halo_kernel<40, 8, DT_F16>), grid, dim3(256), LDS_HALO40, c.s, p); return 0; }
      if (DT == DT_F16 && g_conv_ablate == 16) { FP_LAUNCH((conv_halo_kernel<40, 16, DT_F16>), grid, dim3(256), LDS_HALO40, c.s, p); return 0; }
      if (DT == DT_F16 && g_conv_ablate == 32) { FP_LAUNCH((conv_halo_kernel<40, 32, DT_F16>), grid, dim3(256), LDS_HALO40, c.s, p); return 0; }
#endif
      if (!g_halo_wpack) p.wpack = nullptr;
#ifdef FP_TEST_HOOKS
      if (g_halo_wreg && p.wfrag && g_conv_ablate == 0 && (p.cin_b >= 512 || g_halo_wreg == 2)) { FP_LAUNCH((conv_halo_wreg_kernel<DT>), grid, dim3(256), LDS_HALO40W, c.s, p); return 0; }
#endif
      FP_LAUNCH((conv_halo_kernel<40, 0, DT>), grid, dim3(256), LDS_HALO40, c.s, p);
    } else if constexpr (odt_q(ODT) == DT) {
      ProfScope ps(c.prof, c.s, (tg + "/conv_halo8_kernel").c_str(), flops, bytes);
      if (!g_halo_wpack) p.wpack = nullptr;
      FP_LAUNCH((conv_halo8_kernel<DT, ODT>), grid, dim3(256), LDS_HALO8, c.s, p);
    }
    return 0;
  }
  if constexpr (B2) {
    if (!grp && (g_conv_variant == 0 || g_conv_variant == 8) && L.Cout == 128 && p.ksplit == 1 && KT >= 4 && KT <= 80) {
      // 512x128 ping-pong tiles (conv_pp32_kernel) for as many FULL rounds of the 256 CUs as the problem has; the remaining
      // rows go to the kernels below
      const int mt_all = p.M / 512;
      const int mt_big = (mt_all / 256) * 256;
      if (mt_big > 0) {
        ConvParams pb = p;
        pb.M = mt_big * 512;
        const double frac = (double)pb.M / (double)p.M;
        {
          ProfScope ps(c.prof, c.s, (tg + "/conv_pp32_kernel<512,128>").c_str(), flops * frac, bytes * frac);
          FP_LAUNCH((conv_pp32_kernel<512, 128, DT, ODT>), dim3(mt_big), dim3(512), LDS_PP32, c.s, pb);
        }
        flops *= (1.0 - frac); bytes *= (1.0 - frac);
        p.m_begin = mt_big * 512;
        if (p.m_begin >= p.M) return 0;
        mtiles = (p.M - p.m_begin + 127) / 128;
        if (g_rem_splitk && plan_splitk(p.M - p.m_begin, 384)) return 1;
      }
    }
  }
  if constexpr (!(DT == DT_F16 && QOUT))  // (encodeA.1 has 128 output channels: no 256-wide instantiation of the f16 -> 8-bit boundary)
  if (!grp && (g_conv_variant == 5 || g_conv_variant == 0 || g_conv_variant == 8) && L.Cout % 256 == 0 && p.ksplit == 1 && KT >= 2) {
    // 256x256 tiles for as many FULL rounds of the 256 CUs as the problem has, the remaining rows on smaller tiles
    const int nt2 = L.Cout / 256;
    const int mt_all = p.M / 256;                         // whole 256-row m-tiles
    const int mt_big = (mt_all * nt2 / 256) * 256 / nt2;  // m-tiles covered by full rounds
    if (p.post) {
      // the positional table is fused only when every row takes a schedule that implements it: the 256x256 rounds + the deep-ring
      // kernel for the left-over (N = 252); otherwise nobody adds it here and the caller launches add_pos_embed_kernel
      const int rows_left = p.M - mt_big * 256;
      post_main = mt_big > 0 && g_conv_variant == 0 && g_conv_ablate == 0 && g_rem_kernel == 3 && KT >= 16 &&
                  (rows_left == 0 || ((rows_left + 63) / 64) * (L.Cout / 128) <= 256);
      if (!post_main) p.post = nullptr;
    }
    // [r5] the left-over rows run NEXT TO the full rounds, on the side stream: a lone deep-ring launch leaves 100 of the 256 CUs idle for
    // 35-40 us behind every conv_512 / conv_b2; forked, its workgroups take CUs as the 256x256 tiles release them (disjoint rows, same inputs)
    if (mt_big > 0) {
      const int rows_left = p.M - mt_big * 256;
      fork_rem = g_rem_fork && c.s2 && !(c.prof && c.prof->on) && rows_left > 0 && g_rem_kernel == 3 && KT >= 16 && L.Cout % 128 == 0 &&
                 ((rows_left + 63) / 64) * (L.Cout / 128) <= 256 && g_conv_variant == 0 && g_conv_ablate == 0 && !g_rem_splitk;
      if (fork_rem) {
        FP_HIP_OK(hipEventRecord(c.ev_fork, c.s));
        FP_HIP_OK(hipStreamWaitEvent(c.s2, c.ev_fork, 0));
      }
      ConvParams pb = p;
      pb.M = mt_big * 256;                                // rows [0, mt_big*256)
      const double frac = (double)pb.M / (double)p.M;
      {
        ProfScope ps(c.prof, c.s, (tg + "/conv_big_pp_kernel").c_str(), flops * frac, bytes * frac);
        const dim3 grid(mt_big * nt2);
        bool done = false;
#ifdef FP_TEST_HOOKS
        if (DT == DT_F16 && g_conv_ablate) {
          done = true;
          switch (g_conv_ablate) {
            case 1: FP_LAUNCH((conv_big_pp_kernel<1, DT_F16>), grid, dim3(512), LDS_BIG, c.s, pb); break;
            case 2: FP_LAUNCH((conv_big_pp_kernel<2, DT_F16>), grid, dim3(512), LDS_BIG, c.s, pb); break;
            case 3: FP_LAUNCH((conv_big_pp_kernel<3, DT_F16>), grid, dim3(512), LDS_BIG, c.s, pb); break;
            case 4: FP_LAUNCH((conv_big_pp_kernel<4, DT_F16>), grid, dim3(512), LDS_BIG, c.s, pb); break;
            case 8: FP_LAUNCH((conv_big_pp_kernel<8, DT_F16>), grid, dim3(512), LDS_BIG, c.s, pb); break;
            case 16: FP_LAUNCH((conv_big_pp_kernel<16, DT_F16>), grid, dim3(512), LDS_BIG, c.s, pb); break;
            default: done = false;
          }
        }
#endif
        if constexpr (!QOUT) {
          if (!done && post_main) { FP_LAUNCH((conv_big_pp_kernel<0, DT, ODT, true>), grid, dim3(512), LDS_BIG, c.s, pb); done = true; }
        }
        if constexpr (!QOUT && B2) {
          if (!done && g_conv_lds_store) { FP_LAUNCH((conv_big_pp_kernel<0, DT, ODT, false, true>), grid, dim3(512), LDS_BIG, c.s, pb); done = true; }
        }
        if (!done) FP_LAUNCH((conv_big_pp_kernel<0, DT, ODT>), grid, dim3(512), LDS_BIG, c.s, pb);
      }
      flops *= (1.0 - frac); bytes *= (1.0 - frac);
      p.m_begin = mt_big * 256;
      if (p.m_begin >= p.M) return 0;
      mtiles = (p.M - p.m_begin + 127) / 128;
      if (g_rem_splitk && plan_splitk(p.M - p.m_begin, 384)) return 1;
    }
  }
  if (p.post && !post_main && p.ksplit == 1) p.post = nullptr;  // only the split-K reduce implements it on the remaining paths
  if (!grp && (p.m_begin > 0 || (g_rem_small && p.M >= 8192) || small_deep) && g_rem_kernel && (KT >= 16 || small_deep) && p.ksplit == 1 && L.Cout % 128 == 0) {
    // left-over rows on an otherwise idle chip: a lone workgroup per CU walks all K-steps, so per-step latency is what
    // counts: conv_512 left-overs 61 us per launch on the 2-stage 128x128 tile, 55 us on the 256x128 ping-pong, 39 us on
    // conv_deep_kernel<64> (neutral-to-slower for the 8-step Linear layers, hence KT >= 16)
    const int n128 = L.Cout / 128;
    if (g_rem_kernel == 4) {
      const dim3 grid(((p.M - p.m_begin + 127) / 128) * n128);
      ProfScope ps(c.prof, c.s, (tg + "/conv_deep_kernel").c_str(), flops, bytes);
      FP_LAUNCH((conv_deep_kernel<128, DT, ODT>), grid, dim3(256), LDS_DEEP128, c.s, p);
      return 0;
    }
    if (g_rem_kernel == 3) {
      // conv_deep_kernel owns its CU, so it only pays while its grid is a single round (<= 256 workgroups).  A larger
      // left-over (mid-sized batches) first takes full rounds of the 256x128 ping-pong kernel, then the deep kernel.
      int rows = p.M - p.m_begin;
      if (((rows + 63) / 64) * n128 > 256) {
        const int mt_pp = (((rows + 255) / 256) * n128 / 256) * 256 / n128;  // 256-row m-tiles in full rounds
        const int rows_pp = std::min(mt_pp * 256, rows);
        const int rest = rows - rows_pp;
        const bool cascade = mt_pp > 0 && ((rest + 63) / 64) * n128 <= 256;
        ConvParams pb = p;
        if (cascade) pb.M = p.m_begin + rows_pp;
        const double frac = cascade ? (double)rows_pp / rows : 1.0;
        {
          ProfScope ps(c.prof, c.s, (tg + "/conv_pp_kernel(rem)").c_str(), flops * frac, bytes * frac);
          FP_LAUNCH((conv_pp_kernel<128, DT, ODT>), dim3(((pb.M - pb.m_begin + 255) / 256) * n128), dim3(512), LDS3_128, c.s, pb);
        }
        if (!cascade || rest == 0) return 0;
        flops *= (1.0 - frac); bytes *= (1.0 - frac);
        p.m_begin += rows_pp;
        rows = rest;
      }
      ProfScope ps(c.prof, c.s, (tg + "/conv_deep_kernel").c_str(), flops, bytes);
      const hipStream_t ls = fork_rem ? c.s2 : c.s;
      bool launched = false;
      if constexpr (!QOUT) {
        if (post_main) {
          FP_LAUNCH((conv_deep_kernel<64, DT, ODT, true>), dim3(((rows + 63) / 64) * n128), dim3(256), LDS_DEEP64, ls, p);
          launched = true;
        }
      }
      if (!launched) FP_LAUNCH((conv_deep_kernel<64, DT, ODT>), dim3(((rows + 63) / 64) * n128), dim3(256), LDS_DEEP64, ls, p);
      if (fork_rem) {   // join: the layer's consumers wait for both parts
        FP_HIP_OK(hipEventRecord(c.ev_join, c.s2));
        FP_HIP_OK(hipStreamWaitEvent(c.s, c.ev_join, 0));
      }
      return 0;
    }
    if (g_rem_kernel == 1) {
      const int mt2 = (p.M - p.m_begin + 255) / 256;
      ProfScope ps(c.prof, c.s, (tg + "/conv_pp_kernel(rem)").c_str(), flops, bytes);
      FP_LAUNCH((conv_pp_kernel<128, DT, ODT>), dim3(mt2 * n128), dim3(512), LDS3_128, c.s, p);
      return 0;
    }
  }
  if (!grp && g_conv_variant == 3 && KT >= 3 && p.ksplit == 1 && L.Cout % 128 == 0) {
    ProfScope ps(c.prof, c.s, (tg + "/conv_pp_kernel").c_str(), flops, bytes);
    FP_LAUNCH((conv_pp_kernel<128, DT, ODT>), dim3(((p.M + 255) / 256) * (L.Cout / 128)), dim3(512), LDS3_128, c.s, p);
    return 0;
  }
  if (L.Cout % 128 == 0 && p.ksplit == 1 && g_gemm_deep && p.m_begin == 0 && KT >= 4 && KT <= 8 && mtiles * (L.Cout / 128) <= 128 &&
      (!grp || grp->rows % 128 == 0)) {
    // short-K layers of small problems (the Linear layers of Track): a workgroup is a chain of <= 8 K-steps whose load latency the
    // two-stage tile below exposes every step; the deep ring keeps three steps in flight
    ProfScope ps(c.prof, c.s, (tg + "/conv_deep_kernel(short-K)").c_str(), flops, bytes);
    FP_LAUNCH((conv_deep_kernel<128, DT, ODT>), dim3(mtiles * (L.Cout / 128)), dim3(256), LDS_DEEP128, c.s, p);
  } else if (L.Cout % 128 == 0 && !grp && p.ksplit > 1 && g_splitk_deep && p.kt_per >= 4) {
    // split-K slices on the deep-ring kernel (three K-steps in flight instead of one: a slice is a latency chain)
    ProfScope ps(c.prof, c.s, (tg + "/conv_deep_kernel(split-K)").c_str(), flops, bytes);
    FP_LAUNCH((conv_deep_kernel<128, DT, ODT>), dim3(mtiles * (L.Cout / 128) * p.ksplit), dim3(256), LDS_DEEP128, c.s, p);
  } else if (L.Cout % 128 == 0) {
    ProfScope ps(c.prof, c.s, (tg + "/conv_igemm_kernel<128>").c_str(), flops, bytes);
    const dim3 grid(mtiles * (L.Cout / 128) * p.ksplit);
    bool done = false;
#ifdef FP_TEST_HOOKS
    if (DT == DT_F16 && g_conv_variant >= 10) {
      done = true;
      switch (g_conv_variant) {
        case 10: FP_LAUNCH((conv_igemm_kernel<128, 0, DT_F16>), grid, dim3(256), LDS_IG128, c.s, p); break;
        case 11: FP_LAUNCH((conv_igemm_kernel<128, 1, DT_F16>), grid, dim3(256), LDS_IG128, c.s, p); break;
        case 12: FP_LAUNCH((conv_igemm_kernel<128, 2, DT_F16>), grid, dim3(256), LDS_IG128, c.s, p); break;
        case 17: FP_LAUNCH((conv_igemm_kernel<128, 7, DT_F16>), grid, dim3(256), LDS_IG128, c.s, p); break;   // no loads
        case 21: FP_LAUNCH((conv_igemm_kernel<128, 11, DT_F16>), grid, dim3(256), LDS_IG128, c.s, p); break;  // no MFMAs
        case 25: FP_LAUNCH((conv_igemm_kernel<128, 15, DT_F16>), grid, dim3(256), LDS_IG128, c.s, p); break;  // neither
        default: done = false;
      }
    }
#endif
    if (!done) FP_LAUNCH((conv_igemm_kernel<128, 3, DT, ODT>), grid, dim3(256), LDS_IG128, c.s, p);
  } else {
    ProfScope ps(c.prof, c.s, (tg + "/conv_igemm_kernel<64>").c_str(), flops, bytes);
    FP_LAUNCH((conv_igemm_kernel<64, 3, DT, ODT>), dim3(mtiles * (L.Cout / 64) * p.ksplit), dim3(256), LDS_IG64, c.s, p);
  }
  if (p.ksplit > 1) {
    ProfScope ps(c.prof, c.s, (tg + "/conv_splitk_reduce_kernel").c_str(), 0, 0);
    size_t octs = (size_t)(p.M - p.m_begin) * (p.Cout / 8);
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((octs + 255) / 256)), dim3(256), 0, c.s, p);
  }
  return 0;
}

// post / post_fused: a positional table the layer may add to its output (see ConvParams::post); *post_fused tells the
// caller whether the schedule that ran did (otherwise the caller launches add_pos_embed_kernel)
// out2 / oinv (8-bit networks): the layer also writes the 8-bit copy of its f16 output, value * oinv[channel] (DT_DUAL_*)
static int run_conv(const Ctx &c, const char *tag, const ConvLayer &L, const Act &in, int NB, int H, int W, int ipad,
                    const Act &out, int opad, bool relu, const Act *res = nullptr, int rpad = 0, int split_imgs = 0,
                    const ConvGroup *grp = nullptr, const void *post = nullptr, bool *post_fused = nullptr, const Act *out2 = nullptr,
                    const float *oinv = nullptr, const float *rscale = nullptr, const float *bias_img = nullptr) {
  ConvParams p;
  p.bias_img = bias_img;
  FP_CHECK(!bias_img || (L.dt == DT_I8 && !grp), "run_conv: a per-image bias belongs to an INT8 layer");
  p.out2 = out2 ? (unsigned char *)out2->p : nullptr;
  p.oinv = oinv;
  p.rscale = rscale;
  FP_CHECK(!rscale || (res && res->dt == DT_I8 && L.dt == DT_I8 && !out2 && !grp), "run_conv: unsupported 8-bit residual");
  FP_CHECK(!out2 || (out.dt == DT_F16 && is_q8(out2->dt) && oinv && !grp && !post), "run_conv: unsupported dual output");
  FP_CHECK(out2 || !oinv || (is_q8(out.dt) && (out.dt == L.dt || (L.dt == DT_F16 && out.dt == DT_I8)) && !grp && !post), "run_conv: unsupported scaled 8-bit output");
  p.post = (const unsigned char *)post;
  if (post_fused) *post_fused = false;
  FP_CHECK(in.dt == L.dt, "run_conv: input element type does not match the layer's weights");
  FP_CHECK(!is_q8(L.dt) || L.cscale, "run_conv: 8-bit layer without scales");
  const int es = elem_bytes(L.dt);
  p.clk = g_clk_probe;
  p.grp_rows = grp ? grp->rows : 0;
  p.in_shared = grp && grp->in_shared;
  p.res_shared = grp && grp->res_shared;
  p.grp_w_bytes = grp ? (unsigned)((size_t)L.Cout * L.KH * L.KW * L.Cin * es) : 0;
  FP_CHECK(!grp || (grp->rows % 128 == 0 && L.KH == 1 && L.KW == 1 && NB == 2 * grp->rows && !is_q8(L.dt)), "grouped launch: unsupported shape");
  p.in = (const unsigned char *)in.p; p.w = L.w; p.wfrag = L.wfrag; p.wpack = L.wpack; p.wpack128 = g_big_wpack ? L.wpack128 : nullptr; p.wdeep = g_deep_wpack ? L.wdeep : nullptr; p.bias = L.bias; p.cscale = is_q8(L.dt) ? L.cscale : nullptr;
  p.res = res ? (const unsigned char *)res->p : nullptr; p.out = (unsigned char *)out.p;
  p.out_dt = out.dt; p.res_dt = res ? res->dt : out.dt;
  p.NB = NB; p.H = H; p.W = W; p.Cin = L.Cin;
  p.KH = L.KH; p.KW = L.KW; p.stride = L.stride; p.pad = L.pad;
  p.ipad = ipad; p.opad = opad; p.rpad = rpad;
  p.OH = (H + 2 * L.pad - L.KH) / L.stride + 1;
  p.OW = (W + 2 * L.pad - L.KW) / L.stride + 1;
  if (L.KH == 4 && L.pad == 2 && L.stride == 1) { p.OH = H; p.OW = W; }  // s2d stem: asymmetric padding (2 before, 1 after)
  p.Cout = L.Cout;
  p.M = NB * p.OH * p.OW;
  p.Ktot = L.KH * L.KW * L.Cin;
  p.cin_b = L.Cin * es;
  p.krow_b = p.Ktot * es;
  p.ntaps = L.KH * L.KW;
  p.relu = relu ? 1 : 0;
  p.split_imgs = split_imgs;
  p.out_ld = split_imgs > 0 ? 2 * L.Cout : L.Cout;
  p.res_ld = L.Cout;
  p.ksplit = 1; p.kt_per = p.krow_b / 128; p.partial = nullptr;
  p.m_begin = 0;
  FP_CHECK(ipad >= L.pad && (p.cin_b == 64 || p.cin_b % 128 == 0) && p.krow_b % 128 == 0 && (L.Cout % 64) == 0 &&
               (p.cin_b != 64 || L.KW % 2 == 0) && p.krow_b / 128 <= 80,
           "conv shape not supported by the MFMA kernel");
  {
    // K order: 128-byte channel chunks -> (chunk outer, tap inner); 64-byte pixels (s2d stem) -> two horizontally adjacent
    // taps (128 contiguous bytes) per K-step.  koff = byte offset of the K-step's X slab from the row's (tap 0, ch 0) address.
    const int IWp = W + 2 * ipad;
    for (int kt = 0; kt < p.krow_b / 128; kt++) {
      int kh, kw, ch;
      if (p.cin_b >= 128) { ch = kt / p.ntaps; int tap = kt % p.ntaps; kh = tap / L.KW; kw = tap % L.KW; }
      else { ch = 0; int tap = 2 * kt; kh = tap / L.KW; kw = tap % L.KW; }
      p.koff[kt] = (unsigned)((kh * IWp + kw) * p.cin_b + ch * 128);
      if (2 * kt + 1 < 160) { p.koff32[2 * kt] = p.koff[kt]; p.koff32[2 * kt + 1] = p.koff[kt] + 64; }
    }
  }
  const bool hr = res != nullptr;
  FP_CHECK(!res || res->dt == (rscale ? DT_I8 : is_q8(L.dt) ? DT_F16 : L.dt), "run_conv: the residual must have the layer's operand type (f16 for the 8-bit layers)");
  FP_CHECK(!post || (opad == 0 && split_imgs == 0 && !is_q8(out.dt) && post_fused), "run_conv: positional table on an unsupported layer");
  struct PostReport {  // the split-K decision is taken inside run_conv_dt (p.ksplit)
    ConvParams &p; bool *flag;
    ~PostReport() { if (flag) *flag = p.post != nullptr; }  // (run_conv_dt clears p.post when no schedule that ran implements it)
  } post_report{p, post_fused};
  if (out2) {
    if (L.dt == DT_FP8 && out2->dt == DT_FP8) return run_conv_dt<DT_FP8, DT_DUAL_FP8>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);
    if (L.dt == DT_I8 && out2->dt == DT_I8) return run_conv_dt<DT_I8, DT_DUAL_I8>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);
    if (L.dt == DT_F16 && out2->dt == DT_FP8) return run_conv_dt<DT_F16, DT_DUAL_FP8>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);
    if (L.dt == DT_F16 && out2->dt == DT_I8) return run_conv_dt<DT_F16, DT_DUAL_I8>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);
    FP_CHECK(false, "run_conv: unsupported combination of operand / dual-output element types");
  }
#ifdef FP_TEST_HOOKS
  if (rscale) {   // (run_trunk_i8) the residual is the 8-bit stream copy
    if (oinv) return run_conv_dt<DT_I8, DT_QSR_I8>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);
    FP_CHECK(out.dt == DT_F16, "run_conv: 8-bit residual with an unsupported output type");
    return run_conv_dt<DT_I8, DT_F16RQ_I8>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);
  }
#else
  FP_CHECK(!rscale, "run_conv: 8-bit residual operands exist in the test build only");
#endif
  if (oinv) {   // 8-bit output alone, scaled in the epilogue
    if (L.dt == DT_FP8) return run_conv_dt<DT_FP8, DT_QS_FP8>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);
#ifdef FP_TEST_HOOKS
    if (L.dt == DT_F16) return run_conv_dt<DT_F16, DT_QS_I8>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);   // (encodeA.1 in run_trunk_i8)
#else
    FP_CHECK(L.dt != DT_F16, "run_conv: f16 -> scaled 8-bit alone exists in the test build only");
#endif
    return run_conv_dt<DT_I8, DT_QS_I8>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);
  }
  if (L.dt == DT_FP8 && out.dt == DT_FP8) return run_conv_dt<DT_FP8, DT_FP8>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);
  if (L.dt == DT_FP8 && out.dt == DT_F16) return run_conv_dt<DT_FP8, DT_F16>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);
  if (L.dt == DT_I8 && out.dt == DT_I8) return run_conv_dt<DT_I8, DT_I8>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);
  if (L.dt == DT_I8 && out.dt == DT_F16) return run_conv_dt<DT_I8, DT_F16>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);
  if (L.dt == DT_BF16 && out.dt == DT_BF16) return run_conv_dt<DT_BF16, DT_BF16>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);
  if (L.dt == DT_F16 && out.dt == DT_F16) return run_conv_dt<DT_F16, DT_F16>(c, tag, L, p, NB, H, W, ipad, hr, split_imgs, grp);
  FP_CHECK(false, "run_conv: unsupported combination of operand / output element types");
}

// plain GEMM rows x Cin -> rows x Cout (Linear layer) on unpadded buffers
static int run_gemm(const Ctx &c, const char *tag, const ConvLayer &L, const void *in, int rows, void *out, bool relu,
                    const void *res = nullptr, const ConvGroup *grp = nullptr) {
  Act ai{const_cast<void *>(in), L.dt, 1.f}, ao{out, L.dt, 1.f}, ar{const_cast<void *>(res), L.dt, 1.f};
  return run_conv(c, tag, L, ai, rows, 1, 1, 0, ao, 0, relu, res ? &ar : nullptr, 0, 0, grp);
}

// the QKV projection of `rows` tokens ([rows][512] -> [rows][1536]); rows % 80 == 0 and N > 1: qkv_tile_kernel, else the Linear schedule
static int run_qkv(const Ctx &c, const ConvLayer &L, const void *x, int rows, void *qkv) {
  if (g_qkv_tile && rows % 80 == 0 && rows >= 800 && L.wstep && L.Cin == EMBED && L.Cout == 3 * EMBED && (L.dt == DT_F16 || L.dt == DT_BF16)) {
    ProfScope ps(c.prof, c.s, "gemm_qkv/qkv_tile_kernel", 2.0 * rows * EMBED * 3.0 * EMBED, (double)rows * EMBED * 2.0 * 4.0);
    QkvTileParams q{(const unsigned char *)x, L.wstep, L.bias, (unsigned char *)qkv, rows / 80};
#ifdef FP_TEST_HOOKS
    if (g_qkv_ablate == 1) { FP_LAUNCH((qkv_tile_kernel<DT_F16, 1>), dim3((unsigned)q.tiles), dim3(512), 16 * 80 * 64, c.s, q); return 0; }
    if (g_qkv_ablate == 2) { FP_LAUNCH((qkv_tile_kernel<DT_F16, 2>), dim3((unsigned)q.tiles), dim3(512), 16 * 80 * 64, c.s, q); return 0; }
    if (g_qkv_ablate == 3) { FP_LAUNCH((qkv_tile_kernel<DT_F16, 3>), dim3((unsigned)q.tiles), dim3(512), 16 * 80 * 64, c.s, q); return 0; }
    if (g_qkv_ablate == 4) { FP_LAUNCH((qkv_tile_kernel<DT_F16, 4>), dim3((unsigned)q.tiles), dim3(512), 16 * 80 * 64, c.s, q); return 0; }
    if (g_qkv_ablate == 7) { FP_LAUNCH((qkv_tile_kernel<DT_F16, 7>), dim3((unsigned)q.tiles), dim3(512), 16 * 80 * 64, c.s, q); return 0; }
#endif
    if (L.dt == DT_BF16) FP_LAUNCH((qkv_tile_kernel<DT_BF16>), dim3((unsigned)q.tiles), dim3(512), 16 * 80 * 64, c.s, q);
    else FP_LAUNCH((qkv_tile_kernel<DT_F16>), dim3((unsigned)q.tiles), dim3(512), 16 * 80 * 64, c.s, q);
    return 0;
  }
  return run_gemm(c, "gemm_qkv", L, x, rows, qkv, false);
}

template <int DT>
static void launch_attention(const Ctx &c, const void *qkv, void *out, int B, int T, int tstride, int ld) {
  using E = typename ElemT<DT>::t;
  const E *q = (const E *)qkv;
  E *o = (E *)out;
#ifdef FP_TEST_HOOKS
  if (g_att_variant != 1) {  // the round-1 kernel (64 query rows per workgroup), kept in the test build for A/B
    const int nq = (T + 63) / 64;
    dim3 grid((unsigned)(nq * HEADS * B)), blk(256);
    if (g_att_variant == 3) hipLaunchKernelGGL((attention_kernel<64, false, true, DT>), grid, blk, 0, c.s, q, o, T, nq, tstride);
    else if (g_att_variant == 5) hipLaunchKernelGGL((attention_kernel<64, true, false, DT>), grid, blk, 0, c.s, q, o, T, nq, tstride);
    else if (g_att_variant == 7) hipLaunchKernelGGL((attention_kernel<64, false, false, DT>), grid, blk, 0, c.s, q, o, T, nq, tstride);
    else if (g_att_variant == 9)  // 8 waves = 256 query rows per workgroup (K/V staged half as often; 9 % slower: the two waves of a SIMD run in lockstep)
      hipLaunchKernelGGL((attention32_kernel<true, DT, 0, 8>), dim3((unsigned)(((T + 255) / 256) * HEADS * B)), dim3(512), 0, c.s, q, o, T, (T + 255) / 256, tstride, ld);
    else if (g_att_variant == 10) hipLaunchKernelGGL((attention32_kernel<true, DT, 64>), dim3((unsigned)(((T + 127) / 128) * HEADS * B)), blk, 0, c.s, q, o, T, (T + 127) / 128, tstride, ld);
    else if (g_att_variant == 8) hipLaunchKernelGGL((attention32_kernel<false, DT>), dim3((unsigned)(((T + 127) / 128) * HEADS * B)), blk, 0, c.s, q, o, T, (T + 127) / 128, tstride, ld);
    else if (g_att_variant >= 16 && g_att_variant < 32) {
      const dim3 g32((unsigned)(((T + 127) / 128) * HEADS * B));
      const int nq32 = (T + 127) / 128;
      switch (g_att_variant - 16) {
        case 1: hipLaunchKernelGGL((attention32_kernel<true, DT, 1>), g32, blk, 0, c.s, q, o, T, nq32, tstride, ld); break;
        case 2: hipLaunchKernelGGL((attention32_kernel<true, DT, 2>), g32, blk, 0, c.s, q, o, T, nq32, tstride, ld); break;
        case 4: hipLaunchKernelGGL((attention32_kernel<true, DT, 4>), g32, blk, 0, c.s, q, o, T, nq32, tstride, ld); break;
        case 8: hipLaunchKernelGGL((attention32_kernel<true, DT, 8>), g32, blk, 0, c.s, q, o, T, nq32, tstride, ld); break;
        case 6: hipLaunchKernelGGL((attention32_kernel<true, DT, 6>), g32, blk, 0, c.s, q, o, T, nq32, tstride, ld); break;
        case 14: hipLaunchKernelGGL((attention32_kernel<true, DT, 14>), g32, blk, 0, c.s, q, o, T, nq32, tstride, ld); break;
        case 15: hipLaunchKernelGGL((attention32_kernel<true, DT, 15>), g32, blk, 0, c.s, q, o, T, nq32, tstride, ld); break;
        case 0: hipLaunchKernelGGL((attention32_kernel<true, DT, 16>), g32, blk, 0, c.s, q, o, T, nq32, tstride, ld); break;
        case 3: hipLaunchKernelGGL((attention32_kernel<true, DT, 32>), g32, blk, 0, c.s, q, o, T, nq32, tstride, ld); break;
        default: hipLaunchKernelGGL((attention32_kernel<true, DT, 0>), g32, blk, 0, c.s, q, o, T, nq32, tstride, ld); break;
      }
    }
    else hipLaunchKernelGGL((attention_kernel<64, true, true, DT>), grid, blk, 0, c.s, q, o, T, nq, tstride);
    return;
  }
#endif
  const int nq = (T + 127) / 128;
  if (g_att_skv && nq * HEADS * B <= 64 && T > 32) {  // a small grid of long latency chains: split the keys over the waves instead
    const int nq32 = (T + 31) / 32;
    FP_LAUNCH((attention32_skv_kernel<true, DT>), dim3((unsigned)(nq32 * HEADS * B)), dim3(256), 4 * 2 * (32 * 256 + 8 * 1056), c.s, q, o, T, nq32, tstride, ld);
    return;
  }
  // [r5] a sequence of 400 tokens is 3 query blocks of 128 rows + 16 rows: the 4th block's workgroup stages every key tile for one half-
  // empty wave.  That tail goes to the split-KV kernel instead (its four waves share the 13 key blocks): one more launch, 25 % fewer
  // workgroups in the main one.  (Small grids take the split-KV kernel for ALL rows, above: the tail rows come out the same either way.)
  const int tail = T % 128;
  if (g_att_tail && tail > 0 && tail <= 32 && T > 128) {
    const int nq_main = T / 128;
    hipLaunchKernelGGL((attention32_kernel<true, DT>), dim3((unsigned)(nq_main * HEADS * B)), dim3(256), 0, c.s, q, o, T, nq_main, tstride, ld);
    FP_LAUNCH((attention32_skv_kernel<true, DT>), dim3((unsigned)(HEADS * B)), dim3(256), 4 * 2 * (32 * 256 + 8 * 1056), c.s, q, o, T, 1, tstride, ld, nq_main * 4);
    return;
  }
  hipLaunchKernelGGL((attention32_kernel<true, DT>), dim3((unsigned)(nq * HEADS * B)), dim3(256), 0, c.s, q, o, T, nq, tstride, ld);
}
static int run_attention(const Ctx &c, int dt, const void *qkv, void *out, int B, int T, int tstride = 0, int ld = 3 * EMBED) {
  if (tstride == 0) tstride = T;
  double flops = 4.0 * (double)B * HEADS * (double)T * T * HDIM;
  ProfScope ps(c.prof, c.s, "attention", flops, (double)B * T * (1536 + 512) * 2.0);
  if (dt == DT_BF16) launch_attention<DT_BF16>(c, qkv, out, B, T, tstride, ld);
  else launch_attention<DT_F16>(c, qkv, out, B, T, tstride, ld);
  return 0;
}

static void run_layernorm(const Ctx &c, int dt, const void *x, const LNParams &ln, void *y, size_t rows, const LNParams *ln1 = nullptr,
                          size_t split_row = 0) {
  ProfScope ps(c.prof, c.s, "layernorm", 0, (double)rows * EMBED * 4.0);
  const dim3 grid((unsigned)((rows + 3) / 4));
  const float *g1 = ln1 ? ln1->g : ln.g, *b1 = ln1 ? ln1->b : ln.b;
  const size_t sr = ln1 ? split_row : rows;
  if (dt == DT_BF16) hipLaunchKernelGGL(layernorm_kernel<DT_BF16>, grid, dim3(256), 0, c.s, (const __bf16 *)x, ln.g, ln.b, (__bf16 *)y, rows, g1, b1, sr);
  else hipLaunchKernelGGL(layernorm_kernel<DT_F16>, grid, dim3(256), 0, c.s, (const _Float16 *)x, ln.g, ln.b, (_Float16 *)y, rows, g1, b1, sr);
}

static void run_layernorm_mean(const Ctx &c, int dt, const void *x, const LNParams &ln, float *out, int B, int T, int tstride = 0,
                               const LNParams *ln1 = nullptr, int split_b = 0) {
  ProfScope ps(c.prof, c.s, "layernorm_mean", 0, (double)B * T * EMBED * 2.0);
  const float *g1 = ln1 ? ln1->g : ln.g, *b1 = ln1 ? ln1->b : ln.b;
  const int sb = ln1 ? split_b : B;
  if (tstride == 0) tstride = T;
  if (dt == DT_BF16) hipLaunchKernelGGL(layernorm_mean_kernel<DT_BF16>, dim3(B), dim3(1024), 0, c.s, (const __bf16 *)x, ln.g, ln.b, g1, b1, sb, out, T, tstride);
  else hipLaunchKernelGGL(layernorm_mean_kernel<DT_F16>, dim3(B), dim3(1024), 0, c.s, (const _Float16 *)x, ln.g, ln.b, g1, b1, sb, out, T, tstride);
}

static void run_small_linear(const Ctx &c, const float *x, const LinearF32 &L, float *y, int B) {
  ProfScope ps(c.prof, c.s, "small_linear", 2.0 * B * L.out * L.in, 0);
  size_t waves = (size_t)((B + 7) / 8) * L.out;
  hipLaunchKernelGGL(small_linear_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, c.s, x, L.w, L.b, y, B, L.out, L.in);
}

static void run_token_mean(const Ctx &c, int dt, const void *x, float *out, int B, int T, int tstride = 0) {
  ProfScope ps(c.prof, c.s, "token_mean", 0, (double)B * T * EMBED * 2.0);
  if (dt == DT_BF16) hipLaunchKernelGGL(token_mean_kernel<DT_BF16>, dim3(B, EMBED / 64), dim3(256), 0, c.s, (const __bf16 *)x, out, T, tstride ? tstride : T);
  else hipLaunchKernelGGL(token_mean_kernel<DT_F16>, dim3(B, EMBED / 64), dim3(256), 0, c.s, (const _Float16 *)x, out, T, tstride ? tstride : T);
}

// calibration statistics of a trunk activation [pixels incl. the zero border][C]: per channel |max| (optional) and the sum of the
// values -- 8-bit tensors de-quantised with their per-channel scale (DT_I8: (stored ^ 0x80) * scale).  Border pixels hold 0 and
// add nothing.  thread = (8 channels, a strided set of pixels): a thread's partial sum has a fixed order, the partial sums are
// combined as 2^-20 fixed-point INTEGER atomics -- order-independent, so the statistics (and every table derived from them) are
// reproducible bit for bit.
template <int DT>
__global__ __launch_bounds__(256) void chan_stats_kernel(const unsigned char *__restrict__ x, size_t pixels, int C, const float *__restrict__ scale,
                                                         float *__restrict__ amax, long long *__restrict__ sum) {
  const int groups = C / 8;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
  const int cg = (int)(t % groups);
  const size_t p0 = t / groups, pstride = nthreads / groups;
  if (p0 >= pstride) return;   // (threads beyond the last whole group of `groups`)
  float m[8], sacc[8], sc[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { m[e] = 0.f; sacc[e] = 0.f; sc[e] = (is_q8(DT) && scale) ? scale[cg * 8 + e] : 1.f; }
  for (size_t px = p0; px < pixels; px += pstride) {
    float f[8];
    if constexpr (DT == DT_I8) {
      const i2 v = *reinterpret_cast<const i2 *>(x + px * C + cg * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) f[e] = (float)(((unsigned)v[e >> 2] >> ((e & 3) * 8) & 0xffu) ^ 0x80u) * sc[e];
    } else {
      decode8(load8_raw(x + (px * C + cg * 8) * elem_bytes(DT), DT), DT, f);
      if constexpr (DT == DT_FP8) {
#pragma unroll
        for (int e = 0; e < 8; e++) f[e] *= sc[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) { m[e] = fmaxf(m[e], fabsf(f[e])); sacc[e] += f[e]; }
  }
#pragma unroll
  for (int e = 0; e < 8; e++) {
    if (amax && m[e] > 0.f) atomicMax(reinterpret_cast<int *>(amax + cg * 8 + e), __float_as_int(m[e]));   // (bits of non-negative floats order like ints)
    if (sacc[e] != 0.f) atomicAdd(reinterpret_cast<unsigned long long *>(sum + cg * 8 + e), (unsigned long long)__double2ll_rn((double)sacc[e] * 1048576.0));
  }
}
// act_id: trunk activation (0..14); dt / scale describe the tensor at `buf`
static void calib_record(const Ctx &c, int act_id, const void *buf, size_t pixels, int C, int dt, const float *scale = nullptr) {
  if (!c.net->calib_mode || (c.net->calib_only >= 0 && c.net->calib_only != act_id)) return;
  // interior pixels: the padded tensors are [images][h+2][w+2] with h = w (40x40 / 20x20 maps), the token tensor has no border
  {
    double interior = (double)pixels;
    if (act_id <= 9) interior = (double)pixels / (42.0 * 42.0) * 1600.0;
    else if (act_id <= 13) interior = (double)pixels / (22.0 * 22.0) * 400.0;
    c.net->calib_count[act_id] += interior;
  }
  float *amax = c.net->calib_mode == 1 ? c.net->calib_amax + act_id * 512 : nullptr;
  long long *sum = c.net->calib_sum + act_id * 512;
  const unsigned char *x = (const unsigned char *)buf;
  const int groups = C / 8;
  const dim3 grid((unsigned)((size_t)1024 * groups / 256)), blk(256);   // 1024 pixel lanes per channel group
  if (dt == DT_BF16) hipLaunchKernelGGL(chan_stats_kernel<DT_BF16>, grid, blk, 0, c.s, x, pixels, C, scale, amax, sum);
  else if (dt == DT_FP8) hipLaunchKernelGGL(chan_stats_kernel<DT_FP8>, grid, blk, 0, c.s, x, pixels, C, scale, amax, sum);
  else if (dt == DT_I8) hipLaunchKernelGGL(chan_stats_kernel<DT_I8>, grid, blk, 0, c.s, x, pixels, C, scale, amax, sum);
  else hipLaunchKernelGGL(chan_stats_kernel<DT_F16>, grid, blk, 0, c.s, x, pixels, C, scale, amax, sum);
}

// arena carve (by capacity, see ensure_scratch)
struct Arena {
  unsigned char *stem, *x128[3], *x256[3], *x512[3], *tokens, *qkv, *att, *y1, *y2;
  unsigned char *q128[3], *q256[3], *q512[3];   // 8-bit networks: 1-byte operand copies
};
static Arena carve(NNScratch *ws) {
  Arena a;
  const size_t cap = (size_t)ws->cap;
  unsigned char *p = ws->buf;
  a.stem = p; p += cap * SZ_STEM;
  for (int i = 0; i < 3; i++) { a.x128[i] = p; p += cap * SZ_128; }
  for (int i = 0; i < 3; i++) { a.x256[i] = p; p += cap * SZ_256; }
  for (int i = 0; i < 3; i++) { a.x512[i] = p; p += cap * SZ_512; }
  a.tokens = p; p += cap * SZ_TOK;
  a.qkv = p; p += cap * SZ_QKV;
  a.att = p; p += cap * SZ_TOK;
  a.y1 = p; p += cap * SZ_TOK;
  a.y2 = p; p += cap * SZ_TOK;
  for (int i = 0; i < 3; i++) { a.q128[i] = p; p += cap * SZ_128 / 2; }
  for (int i = 0; i < 3; i++) { a.q256[i] = p; p += cap * SZ_256 / 2; }
  for (int i = 0; i < 3; i++) { a.q512[i] = p; p += cap * SZ_512 / 2; }   // (only allocated for 8-bit networks; never touched otherwise)
  return a;
}

// the last trunk convolution writes the un-bordered token tensor; whoever did not fuse the positional table adds it here
static void add_pos_embed(const Ctx &c, const Arena &a, int N) {
  const Net *net = c.net;
  size_t rows = (size_t)N * 400;
  ProfScope ps(c.prof, c.s, "add_pos_embed", 0, (double)rows * EMBED * 4.0);
  size_t chunks = rows * (EMBED / 8);
  const dim3 grid((unsigned)((chunks + 255) / 256));
  if (net->act_dt == DT_BF16) hipLaunchKernelGGL(add_pos_embed_kernel<DT_BF16>, grid, dim3(256), 0, c.s, (__bf16 *)a.tokens, (const __bf16 *)net->pe, 400, rows);
  else hipLaunchKernelGGL(add_pos_embed_kernel<DT_F16>, grid, dim3(256), 0, c.s, (_Float16 *)a.tokens, (const _Float16 *)net->pe, 400, rows);
}
static void broadcast_b(const Ctx &c, unsigned char *cat, int N, int cb /* bytes of the b-half of a pixel */) {
  ProfScope ps(c.prof, c.s, "broadcast_b", 0, (double)N * 1600 * 2 * cb);
  size_t total = (size_t)(N - 1) * 1600 * (cb / 16);
  hipLaunchKernelGGL(broadcast_b_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.s, cat, N, 42, 42, 40, 40, 1, cb);
}

// 8-bit trunk (PREC_FP8 / PREC_INT8) [r4].  The 13 3x3 convolutions from encodeA.2 on read 8-bit operands; what differs from the
// round-3 FP8 trunk: (1) the RESIDUAL STREAM stays f16 -- a block's second conv (and encodeA.1 / encodeAB.2, whose outputs start a
// stream) writes the f16 tensor and, in the same epilogue, its 8-bit copy for the next conv (DT_DUAL_*), so the skip path is never
// re-quantised; (2) activation scales are per CHANNEL and folded into the consumer's weights before those are quantised; (3) biases
// carry the calibration's bias correction, the positional table the token correction (net_apply_q8).
#ifdef FP_TEST_HOOKS
// INT8 WITHOUT an f16 stream [r4, an experiment kept in the test build: tools/q8_cross.py].  The sums of a residual block are
// re-quantised to the unsigned 8-bit copy the next conv reads anyway; here the block's second conv reads its skip operand from the
// PREVIOUS 8-bit copy (1 byte instead of 2 per element, scaled per channel in the epilogue: DT_QSR_I8) and writes only the new 8-bit
// copy: 3 bytes less per output element than the dual-output form, Register 720p 7.64 -> 7.08 ms, same-frame accuracy 98.8-100 %
// within 1 mm / 1 deg.  NOT shipped: calibrated on one frame and run on another, the common-mode error of the refined poses is
// 0.6-2.2 mm depending on the frame (f16 stream: 0.6-0.7 mm) -- the rounding bias of a coarsely quantised stream is a property of the
// frame's activation distribution, which the one-frame bias correction cannot carry over.
static int run_trunk_i8(const Ctx &c, const Arena &a, const void *nn_in, int N, int n_b) {
  const Net *net = c.net;
  const int NB2 = N + n_b, q = DT_I8;
  auto F = [&](void *p) { return Act{p, DT_F16, 1.f}; };
  auto Q = [&](void *p) { return Act{p, q, 1.f}; };
  const size_t P1 = (size_t)NB2 * 42 * 42, P2 = (size_t)N * 42 * 42, P5 = (size_t)N * 22 * 22;
  float *const *sc = net->act_scale_dev, *const *oi = net->act_oinv;
  const Act in = F(const_cast<void *>(nn_in)), stem = F(a.stem);
  if (run_conv(c, "conv_stem", net->a0, in, NB2, 80, 80, 2, stem, 1, true)) return 1;
  const Act x0q = Q(a.q128[0]), x1q = Q(a.q128[1]), x2q = Q(a.q128[2]);
  if (run_conv(c, "conv_a1", net->a1, stem, NB2, 80, 80, 1, x0q, 1, true, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, oi[1])) return 1;
  calib_record(c, 1, a.q128[0], P1, 128, q, sc[1]);
  if (run_conv(c, "conv_128", net->ra[0][0], x0q, NB2, 40, 40, 1, x1q, 1, true)) return 1;
  calib_record(c, 2, a.q128[1], P1, 128, q, sc[2]);
  if (run_conv(c, "conv_128", net->ra[0][1], x1q, NB2, 40, 40, 1, x2q, 1, true, &x0q, 1, 0, nullptr, nullptr, nullptr, nullptr, oi[3], sc[1])) return 1;
  calib_record(c, 3, a.q128[2], P1, 128, q, sc[3]);
  if (run_conv(c, "conv_128", net->ra[1][0], x2q, NB2, 40, 40, 1, x1q, 1, true)) return 1;
  calib_record(c, 4, a.q128[1], P1, 128, q, sc[4]);
  const Act catq = Q(a.q256[0]);   // the a|b channel concat
  if (run_conv(c, "conv_128", net->ra[1][1], x1q, NB2, 40, 40, 1, catq, 1, true, &x2q, 1, N, nullptr, nullptr, nullptr, nullptr, oi[5], sc[3])) return 1;
  if (n_b == 1 && N > 1) broadcast_b(c, a.q256[0], N, 128);
  calib_record(c, 5, a.q256[0], P2, 256, q, sc[5]);
  const Act y1q = Q(a.q256[1]), y2q = Q(a.q256[2]), y0q = Q(a.q256[0]);
  if (run_conv(c, "conv_256", net->rb[0][0], catq, N, 40, 40, 1, y1q, 1, true)) return 1;
  calib_record(c, 6, a.q256[1], P2, 256, q, sc[6]);
  if (run_conv(c, "conv_256", net->rb[0][1], y1q, N, 40, 40, 1, y2q, 1, true, &catq, 1, 0, nullptr, nullptr, nullptr, nullptr, oi[7], sc[5])) return 1;
  calib_record(c, 7, a.q256[2], P2, 256, q, sc[7]);
  if (run_conv(c, "conv_256", net->rb[1][0], y2q, N, 40, 40, 1, y1q, 1, true)) return 1;
  calib_record(c, 8, a.q256[1], P2, 256, q, sc[8]);
  if (run_conv(c, "conv_256", net->rb[1][1], y1q, N, 40, 40, 1, y0q, 1, true, &y2q, 1, 0, nullptr, nullptr, nullptr, nullptr, oi[9], sc[7])) return 1;
  calib_record(c, 9, a.q256[0], P2, 256, q, sc[9]);
  const Act z0q = Q(a.q512[0]), z1q = Q(a.q512[1]), z2q = Q(a.q512[2]);
  if (run_conv(c, "conv_b2", net->b2, y0q, N, 40, 40, 1, z0q, 1, true, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, oi[10])) return 1;
  calib_record(c, 10, a.q512[0], P5, 512, q, sc[10]);
  if (run_conv(c, "conv_512", net->rc[0][0], z0q, N, 20, 20, 1, z1q, 1, true)) return 1;
  calib_record(c, 11, a.q512[1], P5, 512, q, sc[11]);
  if (run_conv(c, "conv_512", net->rc[0][1], z1q, N, 20, 20, 1, z2q, 1, true, &z0q, 1, 0, nullptr, nullptr, nullptr, nullptr, oi[12], sc[10])) return 1;
  calib_record(c, 12, a.q512[2], P5, 512, q, sc[12]);
  if (run_conv(c, "conv_512", net->rc[1][0], z2q, N, 20, 20, 1, z1q, 1, true)) return 1;
  calib_record(c, 13, a.q512[1], P5, 512, q, sc[13]);
  const Act tok = F(a.tokens);
  bool pe_done = false;
  if (run_conv(c, "conv_512", net->rc[1][1], z1q, N, 20, 20, 1, tok, 0, true, &z2q, 1, 0, nullptr, net->pe, &pe_done, nullptr, nullptr, sc[12])) return 1;
  if (!pe_done) add_pos_embed(c, a, N);
  calib_record(c, 14, a.tokens, (size_t)N * 400, 512, DT_F16);
  return 0;
}
#endif
static int run_trunk_q8(const Ctx &c, const Arena &a, const void *nn_in, int N, int n_b) {
  const Net *net = c.net;
#ifdef FP_TEST_HOOKS
  if (net->qdt == DT_I8 && g_i8_stream) return run_trunk_i8(c, a, nn_in, N, n_b);
#endif
  const int NB2 = N + n_b, q = net->qdt;
  auto F = [&](void *p) { return Act{p, DT_F16, 1.f}; };
  auto Q = [&](void *p) { return Act{p, q, 1.f}; };
  // INT8 [r5]: per-image bias of layer L for the 8-bit input tensor xq ([NBi, HW+2, HW+2, Cin] bytes): bias minus the first-order
  // compensation of the weights' rounding error for that image's channel means (q8_img_sum_kernel / q8_img_bias_kernel); null when off
  auto IB = [&](const ConvLayer &L, const void *xq, int NBi, int HW) -> const float * {
    // (passes of fewer than 16 hypotheses -- Track, small shards -- run without it: the error-feedback rounding has already cancelled this
    // term for the calibration frames' means, what the compensation adds is the image's deviation from them, and 26 more launches would
    // double a Track.  The corrections of a record are solved on full Registers WITH it: a small pass therefore carries the image's own
    // first-order term uncorrected -- measured on Track 0.24 deg / 0.28 mm against f16)
    // [r6] the decision is per CALL (N, the hypotheses of this pass), not per layer: with NBi the 128-channel layers of a 15-hypothesis
    // pass (NB2 = 16 or 30 images) compensated while its 256- / 512-channel layers (N = 15 images) did not
    if (q != DT_I8 || !net->q8_img_comp || !L.tmat_t || !c.ws || !c.ws->img_sum || !g_q8_imgbias || N < 16) return nullptr;
    ProfScope ps(c.prof, c.s, "q8_img_bias", 0, (double)NBi * (HW + 2) * (HW + 2) * L.Cin);
#ifdef FP_TEST_HOOKS
    if (g_q8_imgbias == 2) {   // A/B (test build): the three-launch form (sliced integer-atomic sums, 64-channel bias blocks, clear)
      hipLaunchKernelGGL(q8_img_sum_kernel, dim3(NBi, HW == 40 ? 6 : 2), dim3(256), 0, c.s, (const unsigned char *)xq, (HW + 2) * (HW + 2), L.Cin, c.ws->img_sum);
      hipLaunchKernelGGL(q8_img_bias_kernel, dim3(NBi, L.Cout / 64), dim3(256), 0, c.s, c.ws->img_sum, L.tmat_t, L.cscale, L.bias, 1.f / (float)(HW * HW), L.Cin, L.Cout, c.ws->img_bias);
      (void)hipMemsetAsync(c.ws->img_sum, 0, (size_t)NBi * L.Cin * sizeof(int), c.s);
    } else
#endif
    {
      // even rows x even columns of the padded image (HW + 2 is even): (HW/2 + 1)^2 lattice points, (HW/2)^2 of them interior
      const int wp2 = g_q8_imgbias == 3 ? 0 : (HW + 2) / 2;
      hipLaunchKernelGGL(q8_img_bias_fused_kernel, dim3((NBi + Q8_BIAS_IMGS - 1) / Q8_BIAS_IMGS), dim3(1024), 0, c.s, (const unsigned char *)xq, wp2 ? wp2 * wp2 : (HW + 2) * (HW + 2), L.tmat_t, L.cscale,
                         L.bias, wp2 ? 1.f / (float)((HW / 2) * (HW / 2)) : 1.f / (float)(HW * HW), L.Cin, L.Cout, c.ws->img_bias, wp2, (HW + 2) * (HW + 2), NBi);
    }
    return c.ws->img_bias;
  };
  const size_t P1 = (size_t)NB2 * 42 * 42, P2 = (size_t)N * 42 * 42, P5 = (size_t)N * 22 * 22;
  // [r6] every stage (encodeA.2-3 / encodeAB.0-1 / encodeAB.2 / encodeAB.3-4) is either on 8-bit operands or on its 2-byte weights and
  // kernels (Net::q8_blocks).  The residual stream is f16 either way, so the tensor a stage hands over always exists in f16; an
  // 8-bit stage's producer also leaves the 8-bit operand copy -- from its own epilogue (DT_DUAL_*) when it is an 8-bit layer or
  // encodeA.1, through q8_copy_kernel when it is a 2-byte layer (those keep their SAME-type kernels: the resident-halo and 256x256
  // schedules have no f16 -> dual-output instantiation).
  const bool A = net->q8_on(0), B = net->q8_on(4), S = net->q8_on(8), C = net->q8_on(9);
  const float *const *oinv = net->act_oinv;
  float *const *scd = net->act_scale_dev;
  auto qcopy = [&](const void *x16, void *xq, int imgs, int HW, int Cc, int act) {
    ProfScope ps(c.prof, c.s, "q8_copy", 0, (double)imgs * HW * HW * Cc * 3.0);
    const size_t octs = (size_t)imgs * HW * HW * (Cc / 8);
    const dim3 grid((unsigned)((octs + 255) / 256));
    if (q == DT_FP8) hipLaunchKernelGGL(q8_copy_kernel<DT_FP8>, grid, dim3(256), 0, c.s, (const _Float16 *)x16, (unsigned char *)xq, oinv[act], HW + 2, HW + 2, 1, Cc, octs);
    else hipLaunchKernelGGL(q8_copy_kernel<DT_I8>, grid, dim3(256), 0, c.s, (const _Float16 *)x16, (unsigned char *)xq, oinv[act], HW + 2, HW + 2, 1, Cc, octs);
  };
  // first conv of a residual block (no skip operand): 8-bit -> 8-bit with the consumer's scales folded in, or plain f16
  auto first = [&](const char *tag, const ConvLayer &L, bool on, const Act &x16, const Act &xq, int NB, int HW, const Act &y16, const Act &yq, int act, size_t P, int Cc) -> int {
    if (on) {
      if (run_conv(c, tag, L, xq, NB, HW, HW, 1, yq, 1, true, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, IB(L, xq.p, NB, HW))) return 1;
      calib_record(c, act, yq.p, P, Cc, q, scd[act]);
    } else {
      if (run_conv(c, tag, L, x16, NB, HW, HW, 1, y16, 1, true)) return 1;
      calib_record(c, act, y16.p, P, Cc, DT_F16);
    }
    return 0;
  };
  // second conv of a block (skip operand res16, always f16): writes the f16 stream tensor y16 and / or the 8-bit copy yq for an 8-bit consumer
  auto second = [&](const char *tag, const ConvLayer &L, bool on, const Act &x16, const Act &xq, int NB, int HW, const Act &res16, const Act *y16, const Act *yq,
                    int split, int act, size_t P, int Cc) -> int {
    const Act &x = on ? xq : x16;
    const float *bi = on ? IB(L, xq.p, NB, HW) : nullptr;
    if (on && y16 && yq) { if (run_conv(c, tag, L, x, NB, HW, HW, 1, *y16, 1, true, &res16, 1, split, nullptr, nullptr, nullptr, yq, oinv[act], nullptr, bi)) return 1; }
    else if (on && yq) { if (run_conv(c, tag, L, x, NB, HW, HW, 1, *yq, 1, true, &res16, 1, split, nullptr, nullptr, nullptr, nullptr, oinv[act], nullptr, bi)) return 1; }
    else {
      if (run_conv(c, tag, L, x, NB, HW, HW, 1, *y16, 1, true, &res16, 1, split, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, bi)) return 1;
      if (yq && !split) qcopy(y16->p, yq->p, NB, HW, Cc, act);   // (the concat layer's copy is made by its caller, behind broadcast_b)
    }
    if (split) return 0;   // (the concat tensor is complete -- and recorded -- only behind broadcast_b: its caller does both)
    if (y16) calib_record(c, act, y16->p, P, Cc, DT_F16);
    else calib_record(c, act, yq->p, P, Cc, q, scd[act]);
    return 0;
  };
  const Act in = F(const_cast<void *>(nn_in)), stem = F(a.stem);
  if (run_conv(c, "conv_stem", net->a0, in, NB2, 80, 80, 2, stem, 1, true)) return 1;
  // encodeA.1 (f16 operands) starts the 128-channel stream: f16 x0 (+ its 8-bit copy)
  const Act x0 = F(a.x128[0]), x1 = F(a.x128[1]), x2 = F(a.x128[2]), x0q = Q(a.q128[0]), x1q = Q(a.q128[1]), x2q = Q(a.q128[2]);
  if (A) { if (run_conv(c, "conv_a1", net->a1, stem, NB2, 80, 80, 1, x0, 1, true, nullptr, 0, 0, nullptr, nullptr, nullptr, &x0q, oinv[1])) return 1; }
  else if (run_conv(c, "conv_a1", net->a1, stem, NB2, 80, 80, 1, x0, 1, true)) return 1;
  if (first("conv_128", net->ra[0][0], A, x0, x0q, NB2, 40, x1, x1q, 2, P1, 128)) return 1;
  if (second("conv_128", net->ra[0][1], A, x1, x1q, NB2, 40, x0, &x2, A ? &x2q : nullptr, 0, 3, P1, 128)) return 1;
  if (first("conv_128", net->ra[1][0], A, x2, x2q, NB2, 40, x1, x1q, 4, P1, 128)) return 1;
  // the last encodeA conv writes the a|b channel concat
  const Act cat = F(a.x256[0]), catq = Q(a.q256[0]);
  if (second("conv_128", net->ra[1][1], A, x1, x1q, NB2, 40, x2, &cat, (A && B) ? &catq : nullptr, N, 5, P2, 256)) return 1;
  if (n_b == 1 && N > 1) {  // image N landed in cat[0][..,128:256]; replicate it for the other hypotheses (both copies)
    broadcast_b(c, a.x256[0], N, 256);
    if (A && B) broadcast_b(c, a.q256[0], N, 128);
  }
  if (B && !A) qcopy(cat.p, catq.p, N, 40, 256, 5);
  calib_record(c, 5, a.x256[0], P2, 256, DT_F16);
  const Act y1 = F(a.x256[1]), y2 = F(a.x256[2]), y0 = F(a.x256[0]), y1q = Q(a.q256[1]), y2q = Q(a.q256[2]), y0q = Q(a.q256[0]);
  if (first("conv_256", net->rb[0][0], B, cat, catq, N, 40, y1, y1q, 6, P2, 256)) return 1;
  if (second("conv_256", net->rb[0][1], B, y1, y1q, N, 40, cat, &y2, B ? &y2q : nullptr, 0, 7, P2, 256)) return 1;
  if (first("conv_256", net->rb[1][0], B, y2, y2q, N, 40, y1, y1q, 8, P2, 256)) return 1;
  // y0 feeds only encodeAB.2: no f16 copy when that layer reads 8-bit operands from an 8-bit producer (0.2 GB per launch less; with a
  // residual the consumer's scales cannot be folded into the tables, so the epilogue scales: DT_QS_*)
  if (second("conv_256", net->rb[1][1], B, y1, y1q, N, 40, y2, (B && S) ? nullptr : &y0, S ? &y0q : nullptr, 0, 9, P2, 256)) return 1;
  const Act z0 = F(a.x512[0]), z1 = F(a.x512[1]), z2 = F(a.x512[2]), z0q = Q(a.q512[0]), z1q = Q(a.q512[1]), z2q = Q(a.q512[2]);
  if (S) {
    const float *bi = IB(net->b2, y0q.p, N, 40);
    if (C) { if (run_conv(c, "conv_b2", net->b2, y0q, N, 40, 40, 1, z0, 1, true, nullptr, 0, 0, nullptr, nullptr, nullptr, &z0q, oinv[10], nullptr, bi)) return 1; }
    else if (run_conv(c, "conv_b2", net->b2, y0q, N, 40, 40, 1, z0, 1, true, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, bi)) return 1;
  } else {
    if (run_conv(c, "conv_b2", net->b2, y0, N, 40, 40, 1, z0, 1, true)) return 1;
    if (C) qcopy(z0.p, z0q.p, N, 20, 512, 10);
  }
  calib_record(c, 10, a.x512[0], P5, 512, DT_F16);
  if (first("conv_512", net->rc[0][0], C, z0, z0q, N, 20, z1, z1q, 11, P5, 512)) return 1;
  if (second("conv_512", net->rc[0][1], C, z1, z1q, N, 20, z0, &z2, C ? &z2q : nullptr, 0, 12, P5, 512)) return 1;
  if (first("conv_512", net->rc[1][0], C, z2, z2q, N, 20, z1, z1q, 13, P5, 512)) return 1;
  const Act tok = F(a.tokens);
  bool pe_done = false;
  if (run_conv(c, "conv_512", net->rc[1][1], C ? z1q : z1, N, 20, 20, 1, tok, 0, true, &z2, 1, 0, nullptr, net->pe, &pe_done, nullptr, nullptr, nullptr, C ? IB(net->rc[1][1], a.q512[1], N, 20) : nullptr)) return 1;
  if (!pe_done) add_pos_embed(c, a, N);
  calib_record(c, 14, a.tokens, (size_t)N * 400, 512, DT_F16);
  return 0;
}

// shared CNN trunk: nn_in [2N,84,84,32] (s2d, border 2) -> tokens [N,400,512] + positional embedding
// n_b = number of observed-crop (B) images following the N rendered (A) images: N, or 1 when all hypotheses share it
static int run_trunk(const Ctx &c, const Arena &a, const void *nn_in, int N, int n_b) {
  const Net *net = c.net;
  if (net->prec == PREC_FP8 || net->prec == PREC_INT8) return run_trunk_q8(c, a, nn_in, N, n_b);
  const int NB2 = N + n_b;
  const int adt = net->act_dt;
  auto T = [&](void *p) { return Act{p, adt, 1.f}; };
  const size_t P1 = (size_t)NB2 * 42 * 42, P2 = (size_t)N * 42 * 42, P5 = (size_t)N * 22 * 22;
  const Act in = T(const_cast<void *>(nn_in)), stem = T(a.stem);
  const Act x0 = T(a.x128[0]), x1 = T(a.x128[1]), x2 = T(a.x128[2]), cat = T(a.x256[0]);
  // ([r5] stem + encodeA.1 in chunks of 63 / 84 / 126 images, so that the stem's output would be read back from the 256 MB memory-side
  // cache instead of HBM: 10.93 -> 11.24 / 11.13 / 11.00 ms per Register, slower with every extra launch -- EXPERIMENTS.md R5.4)
  if (run_conv(c, "conv_stem", net->a0, in, NB2, 80, 80, 2, stem, 1, true)) return 1;
  if (run_conv(c, "conv_a1", net->a1, stem, NB2, 80, 80, 1, x0, 1, true)) return 1;
  calib_record(c, 1, a.x128[0], P1, 128, adt);
  // encodeA residual blocks @40x40x128; the last conv writes the a|b channel concat directly
  if (run_conv(c, "conv_128", net->ra[0][0], x0, NB2, 40, 40, 1, x1, 1, true)) return 1;
  calib_record(c, 2, a.x128[1], P1, 128, adt);
  if (run_conv(c, "conv_128", net->ra[0][1], x1, NB2, 40, 40, 1, x2, 1, true, &x0, 1)) return 1;
  calib_record(c, 3, a.x128[2], P1, 128, adt);
  if (run_conv(c, "conv_128", net->ra[1][0], x2, NB2, 40, 40, 1, x1, 1, true)) return 1;
  calib_record(c, 4, a.x128[1], P1, 128, adt);
  if (run_conv(c, "conv_128", net->ra[1][1], x1, NB2, 40, 40, 1, cat, 1, true, &x2, 1, N)) return 1;
  if (n_b == 1 && N > 1) broadcast_b(c, a.x256[0], N, 256);  // image N landed in cat[0][..,128:256]; replicate it for the other hypotheses
  calib_record(c, 5, a.x256[0], P2, 256, adt);
  // encodeAB
  const Act y1 = T(a.x256[1]), y2 = T(a.x256[2]), y0 = T(a.x256[0]);
  if (run_conv(c, "conv_256", net->rb[0][0], cat, N, 40, 40, 1, y1, 1, true)) return 1;
  calib_record(c, 6, a.x256[1], P2, 256, adt);
  if (run_conv(c, "conv_256", net->rb[0][1], y1, N, 40, 40, 1, y2, 1, true, &cat, 1)) return 1;
  calib_record(c, 7, a.x256[2], P2, 256, adt);
  if (run_conv(c, "conv_256", net->rb[1][0], y2, N, 40, 40, 1, y1, 1, true)) return 1;
  calib_record(c, 8, a.x256[1], P2, 256, adt);
  if (run_conv(c, "conv_256", net->rb[1][1], y1, N, 40, 40, 1, y0, 1, true, &y2, 1)) return 1;
  calib_record(c, 9, a.x256[0], P2, 256, adt);
  const Act z0 = T(a.x512[0]), z1 = T(a.x512[1]), z2 = T(a.x512[2]);
  if (run_conv(c, "conv_b2", net->b2, y0, N, 40, 40, 1, z0, 1, true)) return 1;
  calib_record(c, 10, a.x512[0], P5, 512, adt);
  if (run_conv(c, "conv_512", net->rc[0][0], z0, N, 20, 20, 1, z1, 1, true)) return 1;
  calib_record(c, 11, a.x512[1], P5, 512, adt);
  if (run_conv(c, "conv_512", net->rc[0][1], z1, N, 20, 20, 1, z2, 1, true, &z0, 1)) return 1;
  calib_record(c, 12, a.x512[2], P5, 512, adt);
  if (run_conv(c, "conv_512", net->rc[1][0], z2, N, 20, 20, 1, z1, 1, true)) return 1;
  calib_record(c, 13, a.x512[1], P5, 512, adt);
  // last conv writes the un-bordered token tensor [N,400,512] (2-byte type in every precision)
  const Act tok = T(a.tokens);
  bool pe_done = false;
  if (run_conv(c, "conv_512", net->rc[1][1], z1, N, 20, 20, 1, tok, 0, true, &z2, 1, 0, nullptr, net->pe, &pe_done)) return 1;
  if (!pe_done) add_pos_embed(c, a, N);
  calib_record(c, 14, a.tokens, (size_t)N * 400, 512, adt);
  return 0;
}

bool refiner_fuses_pose(const Net *net) {
  return net && !net->scorer && g_grouped_heads && g_fuse_pose && net->trans.head.out == 3 && net->rot.head.out == 3;
}
int refiner_forward(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const void *nn_in, int N,
                    float *trans_dev, float *rot_dev, int shared_b, const PoseUpdateFuse *fuse, bool *fused_out) {
  if (fused_out) *fused_out = false;
  FP_CHECK(net && !net->scorer, "refiner_forward: wrong network");
  FP_CHECK(net_q8_ready(net), "[FoundationPose] the 8-bit precisions need a calibration: call fp_calibrate (fp_calibrate_fp8) first");
  if (ensure_scratch(ws, N, s)) return 1;
  Ctx c{s, prof, net, ws, ws->side, ws->ev_fork, ws->ev_join};
  const Arena a = carve(ws);
  if (run_trunk(c, a, nn_in, N, shared_b ? 1 : N)) return 1;
  const int dt = net->act_dt;
  const void *x = a.tokens;
  const size_t rows = (size_t)N * 400;
  const EncLayer *heads[2] = {&net->trans, &net->rot};
  float *outs[2] = {trans_dev, rot_dev};
  const auto tail_ok = [&](const EncLayer &L) {
    return L.att.out_proj.wstep && L.lin1.wstep && L.lin2.wstep && L.head.in == EMBED && L.head.out <= 3 && L.att.out_proj.Cin == EMBED && L.lin1.Cin == EMBED &&
           L.lin1.Cout == EMBED && L.lin2.Cin == EMBED && L.lin2.Cout == EMBED && L.att.out_proj.dt == dt && L.lin1.dt == dt && L.lin2.dt == dt;
  };
  if (N == 1 && g_grouped_heads) {
    // Track: both heads in ONE launch per layer (Track is bound by its ~65 dependent launches, not by work).  Rows
    // [0,400) = translation head, [512,912) = rotation head (groups padded to the 128-row tile; the rows in between
    // carry don't-care values that no valid row ever reads: every op here is row-wise, attention is per sequence).
    const int G = 512;
    ConvGroup g_x{G, true, true}, g_in{G, false, true}, g_own{G, false, false};
    const EncLayer &T0 = net->trans, &R0 = net->rot;
    if (run_gemm(c, "gemm_qkv", net->g_in_proj, x, 2 * G, a.qkv, false, nullptr, &g_x)) return 1;
    if (run_attention(c, dt, a.qkv, a.att, 2, 400, G)) return 1;
    if (g_enc_tail && (dt == DT_F16 || dt == DT_BF16) && tail_ok(T0) && tail_ok(R0) && T0.head.out == 3 && R0.head.out == 3
#ifdef FP_TEST_HOOKS
        && g_fuse_pose != 2
#endif
    ) {
      // [r5] everything row-wise behind the attention as ONE launch of 2 x 25 sixteen-token tiles (enc_tail_kernel<., 1>) instead of
      // out_proj, LayerNorm 1, FFN1, FFN2, LayerNorm 2 + partial sums (five dependent launches, 33 us of the 202 us graph)
      float *const pdot = reinterpret_cast<float *>(a.y2);   // [2][25][4]
      {
        ProfScope ps(c.prof, c.s, "enc_tail", 2.0 * 3.0 * 2.0 * 400.0 * EMBED * EMBED, 2.0 * 2.0 * 400.0 * EMBED * 2.0);
        EncTailParams q{};
        q.x = (const unsigned char *)x;
        q.pdot = pdot;
        q.tiles = 25;
        q.head_out = 3;
        const EncLayer *hl[2] = {&T0, &R0};
        for (int i = 0; i < 2; i++) {
          const EncLayer &L = *hl[i];
          q.att[i] = (const unsigned char *)a.att + (size_t)i * G * EMBED * 2;
          q.w[i][0] = L.att.out_proj.wstep; q.w[i][1] = L.lin1.wstep; q.w[i][2] = L.lin2.wstep;
          q.bias[i][0] = L.att.out_proj.bias; q.bias[i][1] = L.lin1.bias; q.bias[i][2] = L.lin2.bias;
          q.ln_g[i][0] = L.ln1.g; q.ln_b[i][0] = L.ln1.b; q.ln_g[i][1] = L.ln2.g; q.ln_b[i][1] = L.ln2.b;
          q.head_w[i] = L.head.w;
        }
        constexpr unsigned kLds = 16 * 16 * 64 + 2 * 8 * 16 * 4;
        if (dt == DT_BF16) FP_LAUNCH((enc_tail_kernel<DT_BF16, 1>), dim3(50), dim3(512), kLds, c.s, q);
        else FP_LAUNCH((enc_tail_kernel<DT_F16, 1>), dim3(50), dim3(512), kLds, c.s, q);
      }
      {
        ProfScope ps(c.prof, c.s, "small_linear", 2.0 * 2 * T0.head.out * T0.head.in, 0);
        EncHeadsParams hp{pdot, {T0.head.b, R0.head.b}, {trans_dev, rot_dev}, 1, 25, 3, 400.f};
        const bool do_fuse = fuse && g_fuse_pose;
        hipLaunchKernelGGL(enc_heads_kernel, dim3(1), dim3(64), 0, c.s, hp, do_fuse ? *fuse : PoseUpdateFuse{}, do_fuse ? 1 : 0);
        if (do_fuse && fused_out) *fused_out = true;
      }
      FP_HIP_OK(hipGetLastError());
      return 0;
    }
    if (run_gemm(c, "gemm_512", net->g_out_proj, a.att, 2 * G, a.y1, false, x, &g_in)) return 1;   // + residual x (shared)
    run_layernorm(c, dt, a.y1, T0.ln1, a.y2, 2 * G, &R0.ln1, G);
    if (run_gemm(c, "gemm_512", net->g_lin1, a.y2, 2 * G, a.y1, true, nullptr, &g_own)) return 1;
    if (run_gemm(c, "gemm_512", net->g_lin2, a.y1, 2 * G, a.att, false, a.y2, &g_own)) return 1;   // + residual x1
    // LayerNorm 2 feeds nothing but the token mean.  [r5] ONE launch: 2 x 16 workgroups normalise 25 rows each and leave partial column sums,
    // which the heads kernel adds up (layernorm_pmean_kernel).  Before: LayerNorm over 800 workgroup-rows + a 16-workgroup mean, two
    // dependent launches (11 us; the one-workgroup-per-sequence fused kernel of Register is a 20 us serial chain at two sequences)
    constexpr int kParts = 16, kRowsPerPart = 25;
    float *const psums = ws->f32 + (size_t)ws->cap * EMBED + 16;   // [2][kParts][512]
    const bool pmean = g_ln_pmean != 0
#ifdef FP_TEST_HOOKS
                       && g_fuse_pose != 2
#endif
        ;
    if (pmean) {
      ProfScope ps(c.prof, c.s, "layernorm_pmean", 0, 2.0 * 400 * EMBED * 2.0);
      if (dt == DT_BF16) hipLaunchKernelGGL(layernorm_pmean_kernel<DT_BF16>, dim3(2, kParts), dim3(256), 0, c.s, (const __bf16 *)a.att, T0.ln2.g, T0.ln2.b, R0.ln2.g, R0.ln2.b, 1, psums, 400, G, kRowsPerPart);
      else hipLaunchKernelGGL(layernorm_pmean_kernel<DT_F16>, dim3(2, kParts), dim3(256), 0, c.s, (const _Float16 *)a.att, T0.ln2.g, T0.ln2.b, R0.ln2.g, R0.ln2.b, 1, psums, 400, G, kRowsPerPart);
    } else run_layernorm(c, dt, a.att, T0.ln2, a.y1, 2 * G, &R0.ln2, G);
#ifdef FP_TEST_HOOKS
    if (fuse && g_fuse_pose == 2 && T0.head.out == 3 && R0.head.out == 3 && T0.head.in == EMBED) {
      // A/B (test build): token mean + both heads + RefinePostProcess as one launch whose last workgroup runs the heads -- measured
      // no faster than the two launches (14.0 us against 6.0 + 6.6 in the replayed graph: the release / acquire pair and the 16
      // same-address atomics across XCDs cost what the launch did), DESIGN.md section 8
      ProfScope ps(c.prof, c.s, "token_mean", 0, 2.0 * 400 * EMBED * 2.0);
      SmallLinear2 sl{{ws->f32, ws->f32 + EMBED}, {T0.head.w, R0.head.w}, {T0.head.b, R0.head.b}, {trans_dev, rot_dev}};
      unsigned *arrivals = reinterpret_cast<unsigned *>(ws->f32 + (size_t)ws->cap * EMBED);
      if (dt == DT_BF16) hipLaunchKernelGGL(token_mean_pose_kernel<DT_BF16>, dim3(2, EMBED / 64), dim3(384), 0, c.s, (const __bf16 *)a.y1, ws->f32, 400, G, arrivals, sl, T0.head.in, *fuse);
      else hipLaunchKernelGGL(token_mean_pose_kernel<DT_F16>, dim3(2, EMBED / 64), dim3(384), 0, c.s, (const _Float16 *)a.y1, ws->f32, 400, G, arrivals, sl, T0.head.in, *fuse);
      if (fused_out) *fused_out = true;
      FP_HIP_OK(hipGetLastError());
      return 0;
    }
#endif
    if (!pmean) run_token_mean(c, dt, a.y1, ws->f32, 2, 400, G);
    {
      ProfScope ps(c.prof, c.s, "small_linear", 2.0 * 2 * T0.head.out * T0.head.in, 0);
      SmallLinear2 a{{pmean ? psums : ws->f32, pmean ? psums + kParts * EMBED : ws->f32 + EMBED}, {T0.head.w, R0.head.w}, {T0.head.b, R0.head.b}, {trans_dev, rot_dev}};
      if (pmean) { a.parts = kParts; a.tokens = 400.f; }
      if (fuse && g_fuse_pose && T0.head.out == 3 && R0.head.out == 3) {
        hipLaunchKernelGGL(small_linear2_pose_kernel, dim3(1), dim3(1024), 0, c.s, a, T0.head.in, *fuse);
        if (fused_out) *fused_out = true;
      } else
        hipLaunchKernelGGL(small_linear2_kernel, dim3((unsigned)((T0.head.out + 3) / 4), 2), dim3(256), 0, c.s, a, 1, T0.head.out, T0.head.in);
    }
    FP_HIP_OK(hipGetLastError());
    return 0;
  }
  if (g_enc_tail && N > 1 && (dt == DT_F16 || dt == DT_BF16) && tail_ok(net->trans) && tail_ok(net->rot) && net->trans.head.out == net->rot.head.out) {
    // [r5] both heads: QKV projection + self-attention per head (the attention outputs in a.att / a.y1), then ONE launch for everything
    // row-wise behind them (enc_tail_kernel, fp_nn_enc_kernels.inc) and one for the token mean + Linear(512,3) of both heads
    void *att_out[2] = {a.att, a.y1};
    for (int i = 0; i < 2; i++) {
      if (run_qkv(c, heads[i]->att.in_proj, x, (int)rows, a.qkv)) return 1;
      if (run_attention(c, dt, a.qkv, att_out[i], N, 400)) return 1;
    }
    float *const pdot = reinterpret_cast<float *>(a.y2);   // [2][N * 5][4] f32
    {
      ProfScope ps(c.prof, c.s, "enc_tail", 2.0 * 3.0 * 2.0 * (double)rows * EMBED * EMBED, 2.0 * 2.0 * (double)rows * EMBED * 2.0);
      EncTailParams q{};
      q.x = (const unsigned char *)x;
      q.pdot = pdot;
      q.tiles = N * 5;
      q.head_out = net->trans.head.out;
      for (int i = 0; i < 2; i++) {
        const EncLayer &L = *heads[i];
        q.att[i] = (const unsigned char *)att_out[i];
        q.w[i][0] = L.att.out_proj.wstep; q.w[i][1] = L.lin1.wstep; q.w[i][2] = L.lin2.wstep;
        q.bias[i][0] = L.att.out_proj.bias; q.bias[i][1] = L.lin1.bias; q.bias[i][2] = L.lin2.bias;
        q.ln_g[i][0] = L.ln1.g; q.ln_b[i][0] = L.ln1.b; q.ln_g[i][1] = L.ln2.g; q.ln_b[i][1] = L.ln2.b;
        q.head_w[i] = L.head.w;
      }
      constexpr unsigned kLds = 16 * 80 * 64 + 2 * 8 * 80 * 4 + 4 * 8192;   // tile + LayerNorm exchanges + the two parked x1 fragments
      if (dt == DT_BF16) FP_LAUNCH((enc_tail_kernel<DT_BF16, 5>), dim3((unsigned)(2 * q.tiles)), dim3(512), kLds, c.s, q);
      else FP_LAUNCH((enc_tail_kernel<DT_F16, 5>), dim3((unsigned)(2 * q.tiles)), dim3(512), kLds, c.s, q);
    }
    {
      ProfScope ps(c.prof, c.s, "small_linear", 2.0 * 2 * N * net->trans.head.out * EMBED, 0);
      EncHeadsParams hp{pdot, {net->trans.head.b, net->rot.head.b}, {trans_dev, rot_dev}, N, 5, net->trans.head.out, 400.f};
      hipLaunchKernelGGL(enc_heads_kernel, dim3((unsigned)N), dim3(64), 0, c.s, hp, PoseUpdateFuse{}, 0);
    }
    FP_HIP_OK(hipGetLastError());
    return 0;
  }
  for (int i = 0; i < 2; i++) {
    const EncLayer &L = *heads[i];
    // post-norm TransformerEncoderLayer: x1 = LN1(x + SA(x)); x2 = LN2(x1 + W2 relu(W1 x1))
    if (run_gemm(c, "gemm_qkv", L.att.in_proj, x, (int)rows, a.qkv, false)) return 1;
    if (run_attention(c, dt, a.qkv, a.att, N, 400)) return 1;
    if (run_gemm(c, "gemm_512", L.att.out_proj, a.att, (int)rows, a.y1, false, x)) return 1;  // + residual x
    run_layernorm(c, dt, a.y1, L.ln1, a.y2, rows);                                           // x1 = y2
    if (run_gemm(c, "gemm_512", L.lin1, a.y2, (int)rows, a.y1, true)) return 1;
    if (run_gemm(c, "gemm_512", L.lin2, a.y1, (int)rows, a.att, false, a.y2)) return 1;       // + residual x1
    if (N >= 96) run_layernorm_mean(c, dt, a.att, L.ln2, ws->f32, N, 400);
    else {  // few sequences: one workgroup per sequence is a serial chain (26 us at N = 32 against 9 + 9 for the two-kernel form)
      run_layernorm(c, dt, a.att, L.ln2, a.y1, rows);
      run_token_mean(c, dt, a.y1, ws->f32, N, 400);
    }
    run_small_linear(c, ws->f32, L.head, outs[i], N);  // Linear(512,3) commutes with the token mean
  }
  FP_HIP_OK(hipGetLastError());
  return 0;
}

int scorer_features(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const void *nn_in, int N, float *feat_dev) {
  FP_CHECK(net && net->scorer, "scorer_features: wrong network");
  FP_CHECK(net_q8_ready(net), "[FoundationPose] the 8-bit precisions need a calibration: call fp_calibrate (fp_calibrate_fp8) first");
  if (ensure_scratch(ws, N, s)) return 1;
  Ctx c{s, prof, net, ws, ws->side, ws->ev_fork, ws->ev_join};
  const Arena a = carve(ws);
  if (run_trunk(c, a, nn_in, N, N)) return 1;
  const size_t rows = (size_t)N * 400;
  if (run_qkv(c, net->att.in_proj, a.tokens, (int)rows, a.qkv)) return 1;
  if (run_attention(c, net->act_dt, a.qkv, a.att, N, 400)) return 1;
  // feature = mean_t(out_proj(att)) = out_proj(mean_t(att))  (out_proj is affine) -> 512x512 GEMV per hypothesis
  run_token_mean(c, net->act_dt, a.att, ws->f32, N, 400);
  run_small_linear(c, ws->f32, net->att.out_proj_f32, feat_dev, N);
  FP_HIP_OK(hipGetLastError());
  return 0;
}

int scorer_head(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const float *feats_dev, int n_total, float *scores_dev) {
  FP_CHECK(net && net->scorer, "scorer_head: wrong network");
  if (ensure_head_scratch(ws, n_total)) return 1;
  Ctx c{s, prof, net, ws, ws->side, ws->ev_fork, ws->ev_join};
  const int N = n_total, dt = net->act_dt;
  unsigned char *p = ws->head_buf;
  unsigned char *xf = p; p += (size_t)N * EMBED * 2;
  unsigned char *qkv = p; p += (size_t)N * 3 * EMBED * 2;
  unsigned char *att = p; p += (size_t)N * EMBED * 2;
  float *o32 = ws->head_f32;                  // [N,512]
  {
    ProfScope ps(c.prof, c.s, "cast", 0, (double)N * EMBED * 6.0);
    size_t n = (size_t)N * EMBED;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dt == DT_BF16) hipLaunchKernelGGL(cast_f32_kernel<DT_BF16>, grid, dim3(256), 0, c.s, feats_dev, (__bf16 *)xf, n);
    else hipLaunchKernelGGL(cast_f32_kernel<DT_F16>, grid, dim3(256), 0, c.s, feats_dev, (_Float16 *)xf, n);
  }
  // att_cross: sequence = the N hypotheses, batch 1
  if (run_gemm(c, "gemm_cross", net->att_cross.in_proj, xf, N, qkv, false)) return 1;
  if (run_attention(c, dt, qkv, att, 1, N)) return 1;
  // out_proj through the same MFMA GEMM (M = N rows), then Linear(512,1) in f32
  if (run_gemm(c, "gemm_cross", net->att_cross.out_proj, att, N, xf, false)) return 1;
  {
    // Linear(512,1) on 2-byte rows: widen to f32 first (token_mean with T = 1 is a plain copy of each row)
    ProfScope ps(c.prof, c.s, "score_linear", 2.0 * N * EMBED, 0);
    if (dt == DT_BF16) hipLaunchKernelGGL(token_mean_kernel<DT_BF16>, dim3(N, EMBED / 64), dim3(256), 0, c.s, (const __bf16 *)xf, o32, 1, 1);
    else hipLaunchKernelGGL(token_mean_kernel<DT_F16>, dim3(N, EMBED / 64), dim3(256), 0, c.s, (const _Float16 *)xf, o32, 1, 1);
  }
  run_small_linear(c, o32, net->score_lin, scores_dev, N);
  FP_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace fp

#ifdef FP_TEST_HOOKS
#include "fp_nn_test_hooks.inc"
#endif  // FP_TEST_HOOKS
