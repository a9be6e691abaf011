// fp_nn.hip -- refine-net and score-net as hand-written CDNA4 (gfx950) kernels: fp16 storage, fp32 accumulate.
//
// Replaces the two opaque TensorRT fp16 engines of the reference (refiner_core_->SyncInfer / scorer_core_->SyncInfer,
// detection_6d_foundationpose/src/foundationpose.cpp:206-208,218-220,254-256; I/O blobs :78-83; shapes
// simple_tests/src/test_foundationpose.cpp:24-35).  The architecture is the published NVlabs FoundationPose one
// (SURVEY.md Appendix B [EXT]); the arithmetic oracle is oracle/nets_torch.py.
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
//     conv_igemm3 / conv_pp / conv_big kernels: earlier schedules kept behind fpt_set_conv_variant for A/B.
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

// =================================================================================================
// implicit-GEMM convolution
// =================================================================================================

// LDS-DMA issued from inline asm: hipcc neither counts these loads nor inserts its conservative `s_waitcnt vmcnt(0)`
// in front of a new LDS-DMA while an older one is in flight (it cannot tell the LDS stages apart), so the counted
// waits in the hand-scheduled kernels are authoritative.  M0 (LDS destination base) is saved and restored inside the statement
// (cdna_hip_programming.md §5.7).  lds_addr must be wave-uniform; the 16 bytes land at lds_addr + lane*16.
__device__ __forceinline__ void glds16_asm(const void *gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
}
// FOUR consecutive 1 KB pieces with ONE M0 set-up: the instruction's immediate offset is added to the global address AND to the
// LDS address, so a source that is stored in LDS order (pack_gemm_w) needs no per-piece address arithmetic and no per-piece
// M0 save / set / restore.  The 4 KB land at lds_addr + lane*16 + {0, 1024, 2048, 3072}.
__device__ __forceinline__ void glds16x2_asm(const void *gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
}
__device__ __forceinline__ void glds16x4_asm(const void *gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
               "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
}
// the same with the non-temporal cache policy, for the input tiles of the resident-halo kernels (each tile is staged once per
// 64-channel chunk by at most two workgroups).  A/B build switch FP_X_NT (tools/ab_xnt.sh), OFF: measured [r3] the halo layers move
// by -4...+1 %, inside the run-to-run noise of a box (+-3 %); the same policy on the X tiles of conv_big_pp / gemm_k32, which other
// n-tiles re-read from L2, costs +4 % / +17 %.
#ifndef FP_X_NT
#define FP_X_NT 0
#endif
__device__ __forceinline__ void glds16_asm_x(const void *gsrc, unsigned lds_addr) {
#if FP_X_NT
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
#else
  glds16_asm(gsrc, lds_addr);
#endif
}

struct ConvParams {
  const unsigned char *in;   // [NB, H+2*ipad, W+2*ipad, Cin]  (zero border of width ipad >= pad is physically present)
  const unsigned char *w;    // [Cout][K] in kernel K order (relayout_k), element type = the kernel's DT
  const unsigned char *wfrag;  // the same weights in MFMA-fragment order (fragment_order; conv_smallm_kernel only), or null
  const unsigned char *wpack;  // the same weights in the LDS-stage order of gemm_k32_kernel (1x1 layers) / conv_halo_kernel (3x3) (pack_stage_w), or null
  const unsigned char *wpack128;  // ... in conv_big_pp_kernel's stage order (pack_stage_w128), or null
  const unsigned char *wdeep;     // ... in conv_deep_kernel's stage order (pack_stage_w128 with 128-row tiles, 4 waves), or null
  const float *bias;         // [Cout]
  const float *cscale;       // FP8 input: [Cout] activation scale * weight scale of the channel; null otherwise
  const unsigned char *res;  // optional residual [NB, OH+2*rpad, OW+2*rpad, res_ld], element type res_dt
  unsigned char *out;        // [NB', OH+2*opad, OW+2*opad, out_ld], element type out_dt
  unsigned char *out2;       // DT_DUAL_* outputs: the 8-bit copy (same shape, 1 byte per element); null otherwise
  const float *rscale;       // DT_QSR_I8 / DT_F16RQ_I8: [res_ld] per-channel scale of the 8-bit residual tensor (value = (byte ^ 0x80) * rscale[c])
  const float *oinv;         // DT_DUAL_* outputs: [out_ld channels... indexed by the layer's OUTPUT channel] 1 / scale of the 8-bit copy
  int NB, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
  int ipad, opad, rpad;
  int M, Ktot, relu, out_ld, res_ld, split_imgs;
  int cin_b;      // bytes per input pixel  (Cin  * element size)
  int krow_b;     // bytes per weight row   (Ktot * element size); K-steps of 128 bytes: krow_b >> 7, of 64 bytes: krow_b >> 6
  int out_dt, res_dt;          // DT_* of the output / residual tensors
  int ntaps;  // KH*KW; for Cin >= 64 the K order is (128-byte channel chunk outer, tap inner) so the taps of a chunk
              // are consecutive K-steps and their overlapping input pixels are re-read while still L2-resident
  // byte offset of K-step kt's X slab relative to a row's (tap 0, channel 0) address; host-filled, read with s_load
  unsigned koff[80];
  unsigned koff32[160];  // same per 64-byte K-step (conv_pp32_kernel)
  int m_begin;            // first output row handled by this launch (hybrid 256^2 + 128^2 launches)
  int ksplit, kt_per;     // split-K (small problems): K-steps [split*kt_per, ...) per workgroup, fp32 partial slabs
  float *partial;         // [ksplit][M - m_begin][Cout]
  // weight groups along M (the refiner's two heads in ONE launch at small N, conv_igemm_kernel only): rows
  // [g*grp_rows, (g+1)*grp_rows) use weights w + g*grp_w_bytes and bias + g*Cout; grp_rows % 128 == 0, 0 = off.
  // in_shared / res_shared: the input / residual tensor has only the first group's rows and is read by every group.
  int grp_rows, in_shared, res_shared;
  unsigned grp_w_bytes;
  unsigned long long *clk;  // optional clock probe: per block {cycles0, realtime0, cycles1, realtime1}
  // positional table [OH*OW][Cout] (element type out_dt) added to the ROUNDED output (exactly what add_pos_embed_kernel
  // computes on the stored tensor); only conv_splitk_reduce_kernel implements it (Track: one launch less)
  const unsigned char *post;
};

// Epilogue shared by every conv schedule.
// Weight rows are PERMUTED on the host inside each block of 16*NI output channels (permute_rows): MFMA tile ni, row
// i = 4g + j (g = lane>>4, j = accumulator register) computes channel
//     NI == 4:  32*(j>>1) + 8*g + 4*(j&1) + ni        NI == 2:  8*g + 2*j + ni
// so a lane's accumulators hold 8 CONSECUTIVE channels per store and the four lane groups of a store
// instruction cover 64 contiguous channels of one pixel (the natural D^T layout gives 4 channels per store and
// twice the store instructions; the 256x256 kernel spent 13 us of a 70 us workgroup in its store burst).
// All bias and residual loads are issued BEFORE the first store: on CDNA4 stores also count in vmcnt, so a load issued
// behind a store cannot be waited for without draining the store.
// DT = the kernel's operand type.  8-bit kernels (DT_FP8 / DT_I8) [r4]: acc (float, or int32 bits for DT_I8) * cscale[c] + bias[c]
// -- cscale = the weight row's scale; the per-INPUT-channel activation scales are folded into the weights before they are quantised
// (net_apply_q8) -- and the residual is ALWAYS f16 (the skip path of the 8-bit networks is never re-quantised).
// ODT = what is written (fp_nn.h):
//   2-byte type            p.out, as before;
//   DT_FP8 / DT_I8         p.out only, no scaling here: the consumer's per-channel scales are folded into this layer's cscale / bias
//                          on the host (legal because these layers have no residual and end in a ReLU);
//   DT_DUAL_FP8 / _I8      an f16 tensor at p.out (the residual stream) AND its 8-bit copy at p.out2 = value * oinv[c] (the next
//                          convolution's operand); both tensors have the same shape / border, so one element offset serves both.
//   DT_QS_FP8 / _I8        the 8-bit copy alone, value * oinv[c], at p.out (a layer WITH a residual whose f16 output nobody reads:
//                          the consumer's scales cannot be folded into the tables then)
// The epilogue covers channel tiles [NI0, NI0 + NI) of an accumulator array of NIT tiles (the stem: two 32-channel passes
// over its 4 tiles; the array is passed whole so that it stays in registers).
__device__ __forceinline__ int pack4_fp8(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(sat_fp8(a), sat_fp8(b), 0, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(sat_fp8(c), sat_fp8(d), w, true);
}
// unsigned 8-bit activations (round to nearest even, saturating to [0, 255]) stored with an offset of -128 (x ^ 0x80) so that the
// signed-integer MFMA can consume them; the offset's contribution, 128 * sum_k w, is folded into the consumer's bias on the host
__device__ __forceinline__ int pack4_u8(float a, float b, float c, float d) {
  unsigned w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(a), 0, 0u);
  w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(b), 1, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(c), 2, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(d), 3, w);
  return (int)(w ^ 0x80808080u);
}
template <int MI, int NI, int DT, int ODT, int EABL = 0, int NIT = NI, int NI0 = 0, class PixFn>  // EABL: 1 = no stores, 2 = no residual loads (timing ablations / layers without residual), 4 = add ConvParams::post
__device__ __forceinline__ void conv_epilogue_px(const ConvParams &p, f4 (&acc)[NIT][MI], int n_base, int lane, PixFn pix,
                                                 int bias_off = 0, int res_img_off = 0) {
  static_assert(NI == 4 || NI == 2, "wave covers 64 or 32 channels");
  constexpr int NS = NI / 2;  // 8-channel stores per pixel per lane
  constexpr int O16 = odt_16(ODT), OQ = odt_q(ODT);
  constexpr bool DUAL = odt_dual(ODT), SCALED = odt_dual(ODT) || odt_qs(ODT), RQ = odt_rq(ODT);
  constexpr int RDT = RQ ? DT_I8 : is_q8(DT) ? DT_F16 : DT;
  const int OHp = p.OH + 2 * p.opad, OWp = p.OW + 2 * p.opad;
  const int RHp = p.OH + 2 * p.rpad, RWp = p.OW + 2 * p.rpad;
  const int g = lane >> 4;
  const int nl = n_base + 8 * g;  // store k covers channels nl + 32*k .. +7
  constexpr int res_es = elem_bytes(RDT);
  // pass 1, in place: acc = acc [* dequantisation scale] + bias (8 channels of bias / scale live at a time)
#pragma unroll
  for (int k = 0; k < NS; k++) {
    float bv[8], sc[8];
    float4 b0 = *reinterpret_cast<const float4 *>(p.bias + bias_off + nl + 32 * k), b1 = *reinterpret_cast<const float4 *>(p.bias + bias_off + nl + 32 * k + 4);
    bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    if constexpr (is_q8(DT)) {
      float4 s0 = *reinterpret_cast<const float4 *>(p.cscale + bias_off + nl + 32 * k), s1 = *reinterpret_cast<const float4 *>(p.cscale + bias_off + nl + 32 * k + 4);
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int jj = (NI == 4) ? 2 * k + (e >> 2) : (e >> 1);
      const int ni = (NI == 4) ? (e & 3) : (e & 1);
#pragma unroll
      for (int mi = 0; mi < MI; mi++) {
        // (__float_as_int takes the element BY VALUE: __builtin_bit_cast on a vector-element lvalue reads element 0 of the vector)
        if constexpr (DT == DT_I8) acc[NI0 + ni][mi][jj] = __builtin_fmaf((float)__float_as_int(acc[NI0 + ni][mi][jj]), sc[e], bv[e]);
        else if constexpr (DT == DT_FP8) acc[NI0 + ni][mi][jj] = __builtin_fmaf(acc[NI0 + ni][mi][jj], sc[e], bv[e]);
        else acc[NI0 + ni][mi][jj] += bv[e];
      }
    }
  }
  float oi[SCALED ? NS : 1][8];  // 1 / (scale of output channel c in the 8-bit copy)
  if constexpr (SCALED) {
#pragma unroll
    for (int k = 0; k < NS; k++) {
      float4 s0 = *reinterpret_cast<const float4 *>(p.oinv + nl + 32 * k), s1 = *reinterpret_cast<const float4 *>(p.oinv + nl + 32 * k + 4);
      oi[k][0] = s0.x; oi[k][1] = s0.y; oi[k][2] = s0.z; oi[k][3] = s0.w; oi[k][4] = s1.x; oi[k][5] = s1.y; oi[k][6] = s1.z; oi[k][7] = s1.w;
    }
  }
  float rs[RQ ? NS : 1][8];    // RQ: scale of channel c of the 8-bit residual tensor
  if constexpr (RQ) {
#pragma unroll
    for (int k = 0; k < NS; k++) {
      float4 s0 = *reinterpret_cast<const float4 *>(p.rscale + nl + 32 * k), s1 = *reinterpret_cast<const float4 *>(p.rscale + nl + 32 * k + 4);
      rs[k][0] = s0.x; rs[k][1] = s0.y; rs[k][2] = s0.z; rs[k][3] = s0.w; rs[k][4] = s1.x; rs[k][5] = s1.y; rs[k][6] = s1.z; rs[k][7] = s1.w;
    }
  }
  // pixels in groups of at most 8 fragments (4 with a positional table): a group's residual / table values stay in registers
  constexpr bool POST = (EABL & 4) != 0;
  static_assert(!POST || (O16 >= 0 && !DUAL), "the positional table is added to plain 2-byte outputs");
  // (8-bit kernels: groups of 4 -- their f16 residual values are twice the bytes of the operands and the FP8 instantiations sit at the register limit)
  constexpr int GB = (POST || is_q8(DT)) ? (MI > 4 ? 4 : MI) : (MI > 8 ? (MI + 1) / 2 : MI);
#pragma unroll
  for (int g0 = 0; g0 < MI; g0 += GB) {
  size_t oofs[GB];
  bool ok[GB];
  i4 rv[NS][GB], pv[NS][POST ? GB : 1];
#pragma unroll
  for (int gi = 0; gi < GB; gi++) {
    const int mi = g0 + gi;
    if (mi >= MI) break;
    int img, oh, ow;
    ok[gi] = pix(mi, img, oh, ow);  // (img, oh, ow) must be a valid address even when !ok
    if constexpr (POST) {
      const size_t tok = (size_t)oh * p.OW + ow;
#pragma unroll
      for (int k = 0; k < NS; k++) pv[k][gi] = *reinterpret_cast<const i4 *>(p.post + (tok * p.Cout + nl + 32 * k) * 2);
    }
    int choff = 0, oimg = img;
    if (p.split_imgs > 0 && img >= p.split_imgs) { oimg = img - p.split_imgs; choff = p.Cout; }
    oofs[gi] = (((size_t)oimg * OHp + oh + p.opad) * OWp + ow + p.opad) * p.out_ld + choff;
    if (p.res && !(EABL & 2)) {
      size_t rpix = ((size_t)(img - res_img_off) * RHp + oh + p.rpad) * RWp + ow + p.rpad;
#pragma unroll
      for (int k = 0; k < NS; k++)
        rv[k][gi] = ok[gi] ? load8_raw(p.res + (rpix * p.res_ld + nl + 32 * k) * res_es, RDT) : (i4){0, 0, 0, 0};
    }
  }
#pragma unroll
  for (int gi = 0; gi < GB; gi++) {
    const int mi = g0 + gi;
    if (mi >= MI) break;
    if (!ok[gi]) continue;
#pragma unroll
    for (int k = 0; k < NS; k++) {
      i4 ov = {0, 0, 0, 0};  // packed 2-byte output: 8 values
      i2 oq = {0, 0};        // packed 8-bit output: 8 bytes
#pragma unroll
      for (int e4 = 0; e4 < 8; e4 += 4) {
        float v4[4];
#pragma unroll
        for (int h = 0; h < 4; h++) {
          const int e = e4 + h;
          const int jj = (NI == 4) ? 2 * k + (e >> 2) : (e >> 1);
          const int ni = (NI == 4) ? (e & 3) : (e & 1);
          float v = acc[NI0 + ni][mi][jj];
          if (p.res && !(EABL & 2)) {
            if constexpr (RQ) v = __builtin_fmaf((float)((((unsigned)rv[k][gi][e >> 2] >> ((e & 3) * 8)) & 0xffu) ^ 0x80u), rs[k][e], v);
            else v += raw_elem<RDT>(rv[k][gi], e);
          }
          if (p.relu) v = fmaxf(v, 0.f);
          if constexpr (POST) v = (float)(typename ElemT<O16 < 0 ? DT_F16 : O16>::t)v + raw_elem<O16 < 0 ? DT_F16 : O16>(pv[k][gi], e);  // = add_pos_embed_kernel on the stored value
          v4[h] = v;
        }
        if constexpr (O16 >= 0) {
          typedef typename ElemT<O16 < 0 ? DT_F16 : O16>::t OE;
          typedef OE oe2 __attribute__((ext_vector_type(2)));
          const oe2 p0 = {(OE)v4[0], (OE)v4[1]}, p1 = {(OE)v4[2], (OE)v4[3]};
          ov[e4 >> 1] = __builtin_bit_cast(int, p0);
          ov[(e4 >> 1) + 1] = __builtin_bit_cast(int, p1);
        }
        if constexpr (OQ >= 0) {
          if constexpr (SCALED) {
#pragma unroll
            for (int h = 0; h < 4; h++) v4[h] *= oi[k][e4 + h];
          }
          if constexpr (OQ == DT_FP8) oq[e4 >> 2] = pack4_fp8(v4[0], v4[1], v4[2], v4[3]);
          else oq[e4 >> 2] = pack4_u8(v4[0], v4[1], v4[2], v4[3]);
        }
      }
      if (EABL & 1) { asm volatile("" ::"v"(ov)); asm volatile("" ::"v"(oq)); continue; }
      if constexpr (O16 >= 0) *reinterpret_cast<i4 *>(p.out + (oofs[gi] + nl + 32 * k) * 2) = ov;
      if constexpr (OQ >= 0) *reinterpret_cast<i2 *>((DUAL ? p.out2 : p.out) + (oofs[gi] + nl + 32 * k)) = oq;
    }
  }
  }
}

// the implicit-GEMM schedules: output row m = m_base + mi*16 + (lane&15) in (image, oh, ow) raster order
template <int MI, int NI, int DT, int ODT, int EABL = 0>
__device__ __forceinline__ void conv_epilogue(const ConvParams &p, f4 (&acc)[NI][MI], int m_base, int n_base, int lane) {
  const int ohw = p.OH * p.OW;
  const int grp = p.grp_rows ? m_base / p.grp_rows : 0;  // tile-uniform (grp_rows is a multiple of the tile height)
  conv_epilogue_px<MI, NI, DT, ODT, EABL>(p, acc, n_base, lane, [&](int mi, int &img, int &oh, int &ow) {
    int m = m_base + mi * 16 + (lane & 15);
    const bool ok = m < p.M;
    int mm = ok ? m : 0;
    img = mm / ohw;
    int rem = mm - img * ohw;
    oh = rem / p.OW;
    ow = rem - oh * p.OW;
    return ok;
  }, grp * p.Cout, p.res_shared ? grp * p.grp_rows : 0);
}

// [r3] The same epilogue with the stores staged through LDS.  conv_epilogue's store instruction covers 16 pixels x 64 bytes: 16 cache
// lines, half of each, and the vector memory path retires lines, not bytes.  Here a wave writes its finished
// 2-byte outputs to a private LDS tile ([rows][NI*32 bytes], 16-byte chunk c of row r at c ^ (r & 7)) together with each row's
// output offset, and copies them out as whole NI*32-byte runs, 8 rows (NI = 4: 8 full lines) per instruction.  Values are
// identical to conv_epilogue's (same operations in the same order).  `tile`: wave-private, HM*16*(NI*32) + HM*64 bytes; the MI
// fragments go through it in groups of HM.  2-byte output types, no positional table.
template <int MI, int NI, int DT, int ODT, int HM>
__device__ __forceinline__ void conv_epilogue_lds(const ConvParams &p, f4 (&acc)[NI][MI], int m_base, int n_base, int lane, unsigned char *tile) {
  static_assert(NI == 4 && !is_q8(DT) && odt_q(ODT) < 0 && MI % HM == 0, "64-channel wave tiles, 2-byte operands and outputs");
  constexpr int RB = NI * 32;                 // bytes per pixel row of the tile (64 channels x 2 B)
  constexpr int res_es = elem_bytes(DT);
  const int OHp = p.OH + 2 * p.opad, OWp = p.OW + 2 * p.opad;
  const int RHp = p.OH + 2 * p.rpad, RWp = p.OW + 2 * p.rpad;
  const int g = lane >> 4, li = lane & 15;
  const int nl = n_base + 8 * g;
  const int ohw = p.OH * p.OW;
  unsigned *otab = reinterpret_cast<unsigned *>(tile + HM * 16 * RB);
  float bv[2][8];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + nl + 32 * k), b1 = *reinterpret_cast<const float4 *>(p.bias + nl + 32 * k + 4);
    bv[k][0] = b0.x; bv[k][1] = b0.y; bv[k][2] = b0.z; bv[k][3] = b0.w; bv[k][4] = b1.x; bv[k][5] = b1.y; bv[k][6] = b1.z; bv[k][7] = b1.w;
  }
#pragma unroll
  for (int h0 = 0; h0 < MI; h0 += HM) {
    // residual loads of the group first (stores count in vmcnt on CDNA4: nothing is stored before they are all issued)
    i4 rv[HM][2];
    bool ok[HM];
    unsigned oo[HM];
#pragma unroll
    for (int q = 0; q < HM; q++) {
      const int m = m_base + (h0 + q) * 16 + li;
      ok[q] = m < p.M;
      const int mm = ok[q] ? m : 0;
      const int img = mm / ohw, rem = mm - img * ohw;
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      int choff = 0, oimg = img;
      if (p.split_imgs > 0 && img >= p.split_imgs) { oimg = img - p.split_imgs; choff = p.Cout; }
      oo[q] = ok[q] ? (unsigned)((((size_t)oimg * OHp + oh + p.opad) * OWp + ow + p.opad) * p.out_ld + choff) : 0xFFFFFFFFu;
      if (p.res) {
        const size_t rpix = ((size_t)img * RHp + oh + p.rpad) * RWp + ow + p.rpad;
#pragma unroll
        for (int k = 0; k < 2; k++) rv[q][k] = ok[q] ? load8_raw(p.res + (rpix * p.res_ld + nl + 32 * k) * res_es, DT) : (i4){0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int q = 0; q < HM; q++) {
      const int mi = h0 + q, row = q * 16 + li;
      if (g == 0) otab[row] = oo[q];
#pragma unroll
      for (int k = 0; k < 2; k++) {
        i4 ov = {0, 0, 0, 0};
#pragma unroll
        for (int e2 = 0; e2 < 8; e2 += 2) {
          float v2[2];
#pragma unroll
          for (int hh = 0; hh < 2; hh++) {
            const int e = e2 + hh, jj = 2 * k + (e >> 2), ni = e & 3;
            float v = acc[ni][mi][jj];
            v += bv[k][e];
            if (p.res) v += raw_elem<DT>(rv[q][k], e);
            if (p.relu) v = fmaxf(v, 0.f);
            v2[hh] = v;
          }
          typedef typename ElemT<ODT>::t OE;
          typedef OE oe2 __attribute__((ext_vector_type(2)));
          const oe2 pr = {(OE)v2[0], (OE)v2[1]};
          ov[e2 >> 1] = __builtin_bit_cast(int, pr);
        }
        const int c = k * 4 + g;
        *reinterpret_cast<i4 *>(tile + row * RB + ((c ^ (row & 7)) << 4)) = ov;
      }
    }
    // copy-out (same wave wrote the tile: program order suffices): lane -> (row = it*8 + lane/8, chunk = lane%8)
#pragma unroll
    for (int it = 0; it < HM * 2; it++) {
      const int row = it * 8 + (lane >> 3), c = lane & 7;
      const unsigned off = otab[row];
      const i4 v = *reinterpret_cast<const i4 *>(tile + row * RB + ((c ^ (row & 7)) << 4));
      if (off != 0xFFFFFFFFu) *reinterpret_cast<i4 *>(p.out + ((size_t)off + n_base + c * 8) * 2) = v;
    }
  }
}

// split-K partial slab (true channel order): per accumulator register j a lane owns NI consecutive channels
template <int MI, int NI>
__device__ __forceinline__ void conv_store_partial(const ConvParams &p, f4 (&acc)[NI][MI], int split, int m_base, int n_base, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < MI; mi++) {
    int m = m_base + mi * 16 + (lane & 15);
    if (m >= p.M) continue;
    float *dst = p.partial + ((size_t)split * (p.M - p.m_begin) + (m - p.m_begin)) * p.Cout + n_base;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
      if (NI == 4) {
        f4 v = {acc[0][mi][jj], acc[1][mi][jj], acc[2][mi][jj], acc[3][mi][jj]};
        *reinterpret_cast<f4 *>(dst + 32 * (jj >> 1) + 8 * g + 4 * (jj & 1)) = v;
      } else {
        float2 v = make_float2(acc[0][mi][jj], acc[1][mi][jj]);
        *reinterpret_cast<float2 *>(dst + 8 * g + 2 * jj) = v;
      }
    }
  }
}

// Activations carry a physical zero border, so the K loop has no bounds checks, no selects and no divergent
// branches: a tap's operand address is (wave-uniform tap/chunk offset in SGPRs) + (per-lane row offset fixed for the
// whole kernel), which is exactly the saddr + voffset form of global_load_lds.
template <int BN, int VAR, int DT, int ODT = DT>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = 128;
  constexpr int XB = BM * 128;  // bytes per X stage (128 rows x 64 halfs)
  constexpr int WB = BN * 128;
  constexpr int STAGE = XB + WB;
  constexpr int NREP = BN / 32;     // 16-channel tiles per wave (wave tile = 64 pixels x BN/2 channels)
  constexpr int WPIECES = BN / 32;  // 8-row pieces of the W tile per wave

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  if (p.clk && tid == 0) { p.clk[blockIdx.x * 4] = __builtin_readcyclecounter(); p.clk[blockIdx.x * 4 + 1] = wall_clock64(); }
  const int n_tiles = p.Cout / BN;
  // XCD-aware tile order: hardware places workgroup b on XCD b%8 (speed-only assumption).  Remap so each XCD walks a
  // contiguous range of logical tiles: the n-tiles of one m-tile (same X rows) and neighbouring m-tiles (shared halo
  // rows) hit the same private L2.  Bijective for any grid size.
  int logical;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, within = b >> 3, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int split = logical % p.ksplit;
  logical /= p.ksplit;
  const int mt = logical / n_tiles, nt = logical - mt * n_tiles;
  const int m0 = p.m_begin + mt * BM, n0 = nt * BN;
  const int ohw = p.OH * p.OW;
  const int IHp = p.H + 2 * p.ipad, IWp = p.W + 2 * p.ipad;

  // ---- staging roles: piece = 8 rows x 128 B; lane -> row (lane>>3), slot (lane&7); source chunk is swizzled
  const int srow = lane >> 3;
  const int g = (lane & 7) ^ srow;  // source 16-B chunk within the 64-wide K step
  unsigned xoff[4];                 // byte offset of (row's input pixel at tap (0,0), chunk 0) + g*16
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int m = min(m0 + (wave * 4 + i) * 8 + srow, p.M - 1);  // rows past M re-read the last pixel (never stored)
    if (p.in_shared) m -= (m0 / p.grp_rows) * p.grp_rows;   // every weight group reads the first group's rows
    int img = m / ohw;
    int rem = m - img * ohw;
    int oh = rem / p.OW, ow = rem - oh * p.OW;
    int ih0 = oh * p.stride - p.pad + p.ipad, iw0 = ow * p.stride - p.pad + p.ipad;
    xoff[i] = (unsigned)(((img * IHp + ih0) * IWp + iw0) * p.cin_b + g * 16);
  }
  unsigned woffv[WPIECES];
#pragma unroll
  for (int i = 0; i < WPIECES; i++) {
    int row = (wave * WPIECES + i) * 8 + srow;
    woffv[i] = (unsigned)((n0 + row) * p.krow_b + g * 16);
  }
  const unsigned char *in_b = p.in;
  const unsigned char *w_b = p.w + (p.grp_rows ? (size_t)(m0 / p.grp_rows) * p.grp_w_bytes : 0);

  auto stage = [&](int kt, int buf) {
    unsigned char *xs = smem + buf * STAGE;
    unsigned char *ws = xs + XB;
    const unsigned char *xb = in_b + p.koff[kt];
    const unsigned char *wb = w_b + (size_t)kt * 128;
#pragma unroll
    for (int i = 0; i < 4; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xb + xoff[i]),
                                       (__attribute__((address_space(3))) void *)(xs + (wave * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < WPIECES; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wb + woffv[i]),
                                       (__attribute__((address_space(3))) void *)(ws + (wave * WPIECES + i) * 1024), 16, 0, 0);
  };

  f4 acc[NREP][4];
#pragma unroll
  for (int a = 0; a < NREP; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes) inside a stage; slot = chunk ^ (row&7), row&7 == lane&7
  const int frow = lane & 15, fk = lane >> 4;
  int xfo[2], wfo[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    int slot = (ks * 4 + fk) ^ (lane & 7);
    xfo[ks] = (wm * 64 + frow) * 128 + slot * 16;
    wfo[ks] = XB + (wn * (BN / 2) + frow) * 128 + slot * 16;
  }

  auto compute = [&](int buf) {
    const unsigned char *sb = smem + buf * STAGE;
    if (VAR & 2) {
      // all 16 fragment reads of the K-step are issued up front: only the first LDS round trip is exposed
      i4 xf[2][4], wf[2][NREP];
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
#pragma unroll
        for (int mi = 0; mi < 4; mi++) xf[ks][mi] = *reinterpret_cast<const i4 *>(sb + xfo[ks] + mi * 16 * 128);
#pragma unroll
        for (int ni = 0; ni < NREP; ni++) wf[ks][ni] = *reinterpret_cast<const i4 *>(sb + wfo[ks] + ni * 16 * 128);
      }
      if (VAR & 8) {  // ablation: no MFMAs, fragments kept live
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
#pragma unroll
          for (int mi = 0; mi < 4; mi++) asm volatile("" ::"v"(xf[ks][mi]));
#pragma unroll
          for (int ni = 0; ni < NREP; ni++) asm volatile("" ::"v"(wf[ks][ni]));
        }
        return;
      }
      if (VAR & 1) __builtin_amdgcn_s_setprio(1);
      mma_kstep<DT, NREP, 4>(acc, wf, xf);
      if (VAR & 1) __builtin_amdgcn_s_setprio(0);
      return;
    }
    if constexpr (!is_q8(DT)) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      i4 xf[4], wf[NREP];
#pragma unroll
      for (int mi = 0; mi < 4; mi++) xf[mi] = *reinterpret_cast<const i4 *>(sb + xfo[ks] + mi * 16 * 128);
#pragma unroll
      for (int ni = 0; ni < NREP; ni++) wf[ni] = *reinterpret_cast<const i4 *>(sb + wfo[ks] + ni * 16 * 128);
      if (VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ni = 0; ni < NREP; ni++)
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
          acc[ni][mi] = mfma32<DT>(wf[ni], xf[mi], acc[ni][mi]);
      if (VAR & 1) __builtin_amdgcn_s_setprio(0);
    }
    }
  };

  const int k_begin = split * p.kt_per, k_end = min(p.krow_b >> 7, k_begin + p.kt_per);
  stage(k_begin, 0);
  __syncthreads();
  // steady state is branch-free (stage next tile, compute current tile, one barrier); the last tile is peeled
  int buf = 0;
  for (int kt = k_begin; kt < k_end - 1; kt++) {
    if (!(VAR & 4)) stage(kt + 1, buf ^ 1);  // VAR&4: ablation (stale LDS, timing only)
    compute(buf);
    __syncthreads();
    buf ^= 1;
  }
  compute(buf);

  if (p.clk && tid == 0) { p.clk[blockIdx.x * 4 + 2] = __builtin_readcyclecounter(); p.clk[blockIdx.x * 4 + 3] = wall_clock64(); }
  if (p.ksplit > 1) {  // split-K: raw fp32 partial slab, reduced (+ bias / residual / ReLU) by conv_splitk_reduce_kernel
    conv_store_partial<4, NREP>(p, acc, split, m0 + wm * 64, n0 + wn * (BN / 2), lane);
    return;
  }
  // ---- epilogue: lane owns channels cb..cb+3 (cb = 4*(lane>>4)) of pixel (lane&15) in each 16x16 tile
  conv_epilogue<4, NREP, DT, ODT>(p, acc, m0 + wm * 64, n0 + wn * (BN / 2), lane);
}

// -------------------------------------------------------------------------------------------------
// conv_smallm_kernel [r3]: SMALL problems (Track: 1-2 images, a handful of objects) where every other schedule is a latency
// chain.  At N = 1 a layer is a weight-streaming problem -- conv_512 is 1.9 GFLOP on 4.7 MB of weights, 400 output pixels -- and
// the split-K schedules paid for their parallelism with an fp32 slab round trip and a SECOND launch per layer (9 reduce launches
// = 47 us of a 335 us Track).  Here a workgroup owns 16*MI pixels x 64 channels and its four waves split the K-STEPS (wave w
// takes steps w, w+4, ...): operands go global -> registers directly (no LDS staging, no barriers in the loop; three steps in
// flight per wave), the four partial accumulators are summed through LDS in a fixed order (w0+w1+w2+w3) and wave 0 runs the usual
// epilogue (bias, residual, ReLU, channel-concat addressing, positional table).  One launch per layer, every tensor written once.
// Generic over the conv shapes of both networks (3x3 s1 / s2, the 4x4 space-to-depth stem, 1x1 Linear layers incl. the two weight
// groups of the refiner heads): addressing is the implicit-GEMM one (ConvParams::koff + a per-lane row offset).
// -------------------------------------------------------------------------------------------------
// DEEP: the layer has at least 8 * PF K-steps, i.e. every wave at least 2 * PF: the prologue loads are then UNCONDITIONAL, which
// is what lets the compiler prove how many loads are in flight at the head of the steady-state loop (s_waitcnt vmcnt(10 * (PF-1) + 1)
// instead of vmcnt(1): with conditional prologue loads the in-order counter has to assume the shortest path).
template <int MI, int NI, int DT, int ODT, bool POST, bool DEEP>  // NI = 4: 64 channels per workgroup (Cout % 128 == 0); NI = 2: 32 (the row permutation of Cout == 64 layers)
__global__ __launch_bounds__(256, 2) void conv_smallm_kernel(const ConvParams p) {
  static_assert(!is_q8(DT), "2-byte operand types");
  constexpr int PF = MI == 1 ? 4 : 3;         // K-steps in flight per wave (register budget: (2*NI + 2*MI) * 4 VGPRs per step)
  extern __shared__ __attribute__((aligned(16))) unsigned char red[];  // 3 * NI * MI KB (dynamic, like every kernel launched through FP_LAUNCH)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_tiles = p.Cout / (16 * NI);
  // XCD placement for a WEIGHT-bound layer: hardware puts workgroup b on XCD b % 8, every XCD has its own L2, and at this size
  // the weights (4.7 MB for conv_512) are the traffic.  All workgroups of one channel tile go to ONE XCD (or to 8 / n_tiles XCDs
  // when the layer has fewer than 8 channel tiles), so each L2 pulls its 1/8 of the weights once instead of every L2 pulling all
  // of them (the m-tile-major order of the big schedules would cost 8 x 4.7 MB of fabric traffic per layer here).
  int mt, nt;
  {
    const int b = blockIdx.x, xcd = b & 7, within = b >> 3;
    if (n_tiles >= 8) {              // n_tiles % 8 == 0 (checked by the launcher)
      const int per = n_tiles >> 3;
      nt = xcd + 8 * (within % per);
      mt = within / per;
    } else {                         // 1, 2 or 4 channel tiles: 8 / n_tiles XCDs share one
      const int share = 8 / n_tiles;
      nt = xcd / share;
      mt = within * share + (xcd % share);
    }
  }
  if (mt * (16 * MI) >= p.M) return;   // (grid padded to a multiple of 8)
  const int m0 = mt * (16 * MI), n0 = nt * (16 * NI);
  const int ohw = p.OH * p.OW;
  const int IHp = p.H + 2 * p.ipad, IWp = p.W + 2 * p.ipad;
  const int frow = lane & 15, fk = lane >> 4;
  const int grp = p.grp_rows ? m0 / p.grp_rows : 0;
  // per-lane operand addresses: X row of pixel fragment mi (rows past M re-read the last pixel, never stored), W row of tile ni
  const unsigned char *xrow[MI];
#pragma unroll
  for (int mi = 0; mi < MI; mi++) {
    int m = min(m0 + mi * 16 + frow, p.M - 1);
    if (p.in_shared) m -= grp * p.grp_rows;
    const int img = m / ohw, rem = m - img * ohw;
    const int oh = rem / p.OW, ow = rem - oh * p.OW;
    const int ih0 = oh * p.stride - p.pad + p.ipad, iw0 = ow * p.stride - p.pad + p.ipad;
    xrow[mi] = p.in + ((size_t)(img * IHp + ih0) * IWp + iw0) * p.cin_b + fk * 16;
  }
  // weights in fragment order (fragment_order): one wave instruction = 1 KB of consecutive bytes = 8 whole cache lines
  const unsigned char *wrow = p.wfrag + (p.grp_rows ? (size_t)grp * p.grp_w_bytes : 0) + (size_t)(n0 >> 4) * ((size_t)16 * p.krow_b) + lane * 16;
  const size_t wtile = (size_t)16 * p.krow_b;

  f4 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; a++)
#pragma unroll
    for (int b = 0; b < MI; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

  const int KT = p.krow_b >> 7;
  const int n_my = (KT - wave + 3) >> 2;      // steps wave, wave+4, ...
  i4 wf[PF][2][NI], xf[PF][2][MI];
  auto load = [&](i4 (&w)[2][NI], i4 (&x)[2][MI], int kt) {
    const unsigned ko = p.koff[kt];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
      for (int ni = 0; ni < NI; ni++) w[ks][ni] = *reinterpret_cast<const i4 *>(wrow + ni * wtile + (size_t)kt * 2048 + ks * 1024);
#pragma unroll
      for (int mi = 0; mi < MI; mi++) x[ks][mi] = *reinterpret_cast<const i4 *>(xrow[mi] + ko + ks * 64);
    }
  };
#pragma unroll
  for (int s = 0; s < PF; s++)
    if (DEEP || s < n_my) load(wf[s], xf[s], wave + 4 * s);
  // steady state WITHOUT conditions (every step has a successor PF steps ahead): the compiler's vmcnt bookkeeping stays exact, so
  // PF - 1 steps of loads really are in flight under each step's MFMAs (with the conditions inside, every step waited for
  // vmcnt(0): 21 us per conv_512 launch instead of the ~6 the L1 rate allows)
  int i = 0;
  for (; i + 2 * PF <= n_my; i += PF) {
#pragma unroll
    for (int s = 0; s < PF; s++) {
      // (scheduling fences: hipcc otherwise gathers the loads of all slots at the end of the iteration, and the in-order vmcnt
      // at the loop head then has to wait for nearly all of them -- s_waitcnt vmcnt(1) instead of vmcnt(10 * (PF - 1)))
      __builtin_amdgcn_sched_barrier(0);
      mma_kstep<DT, NI, MI>(acc, wf[s], xf[s]);
      __builtin_amdgcn_sched_barrier(0);
      load(wf[s], xf[s], wave + 4 * (i + s + PF));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // tail: at most 2*PF - 1 steps
#pragma unroll
  for (int s = 0; s < PF; s++) {
    if (i + s < n_my) {
      mma_kstep<DT, NI, MI>(acc, wf[s], xf[s]);
      if (i + s + PF < n_my) load(wf[s], xf[s], wave + 4 * (i + s + PF));
    }
  }
#pragma unroll
  for (int s = 0; s < PF; s++)
    if (i + PF + s < n_my) mma_kstep<DT, NI, MI>(acc, wf[s], xf[s]);
  // ---- fixed-order reduction of the four partial sums: waves 1..3 publish, wave 0 adds them in order and stores
  if (wave > 0) {
    f4 *dst = reinterpret_cast<f4 *>(red + (wave - 1) * (NI * MI * 1024)) + lane;
#pragma unroll
    for (int a = 0; a < NI; a++)
#pragma unroll
      for (int b = 0; b < MI; b++) dst[(a * MI + b) * 64] = acc[a][b];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; w++) {
    const f4 *src = reinterpret_cast<const f4 *>(red + w * (NI * MI * 1024)) + lane;
#pragma unroll
    for (int a = 0; a < NI; a++)
#pragma unroll
      for (int b = 0; b < MI; b++) {
        const f4 v = src[(a * MI + b) * 64];
        // component-wise on purpose: a float4 add is legalised to v_pk_add_f32, which this library must not contain
#pragma unroll
        for (int r = 0; r < 4; r++) acc[a][b][r] = acc[a][b][r] + v[r];
      }
  }
  conv_epilogue<MI, NI, DT, ODT, POST ? 4 : 0>(p, acc, m0, n0, lane);
}

// -------------------------------------------------------------------------------------------------
// conv_smallx_kernel [r3]: conv_smallm_kernel with BOTH operands in the shape the vector L1 likes (tools/bench_tcp.hip: the L1
// serves a wave's 16-byte loads four lanes at a time, one clock per distinct cache line in the four -- 64 clocks for a load in
// MFMA-operand shape, 18 for 1 KB of consecutive bytes or for 8 pixel rows of 128 bytes):
//   * weights: global -> registers from the fragment-order copy (fragment_order), 1 KB of consecutive bytes per instruction;
//   * input pixels: LDS-DMA into a ring PRIVATE to the wave (8 pixels x 128 bytes per instruction, the 16-byte chunks XOR-swizzled
//     by the pixel on the source side), fragments read back with ds_read_b128 -- no workgroup barrier, the wave waits for its own
//     DMA with the in-order vmcnt.
// Every vector-memory instruction in the loop is an asm statement and the waits are counted by hand: L = 2 MI + 2 NI instructions
// per K-step, PF steps in flight, step s is complete when at most (steps issued after s) * L are outstanding.  Same tiles, same
// K split over the four waves, same fixed-order reduction and epilogue as conv_smallm_kernel: results are bit-identical.
// -------------------------------------------------------------------------------------------------
template <int MI, int NI, int DT, int ODT, bool POST, int PFO = 0>
__global__ __launch_bounds__(256, 2) void conv_smallx_kernel(const ConvParams p) {
  // (every element type: operands travel as raw 16-byte pieces, a 128-byte K-step is 64 two-byte or 128 one-byte channels; mma_kstep
  // picks the MFMA.  FP8 layers: Track / a few objects in FP8 mode no longer fall back to the split-K schedules)
  constexpr int PF = PFO ? PFO : (MI == 1 ? 4 : 3);        // K-steps in flight per wave
  constexpr int L = 2 * MI + 2 * NI;         // vector-memory instructions per K-step
  constexpr int STAGE = MI * 2048, RING = PF * STAGE;
  static_assert((PF - 1) * L <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [4 waves][PF stages][MI][16 pixels][128 B], then 3 * NI * MI KB for the reduction
  unsigned char *red = smem + 4 * RING;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_tiles = p.Cout / (16 * NI);
  int mt, nt;  // XCD placement by channel tile, as in conv_smallm_kernel
  {
    const int b = blockIdx.x, xcd = b & 7, within = b >> 3;
    if (n_tiles >= 8) {
      const int per = n_tiles >> 3;
      nt = xcd + 8 * (within % per);
      mt = within / per;
    } else {
      const int share = 8 / n_tiles;
      nt = xcd / share;
      mt = within * share + (xcd % share);
    }
  }
  if (mt * (16 * MI) >= p.M) return;
  const int m0 = mt * (16 * MI), n0 = nt * (16 * NI);
  const int ohw = p.OH * p.OW;
  const int IHp = p.H + 2 * p.ipad, IWp = p.W + 2 * p.ipad;
  const int grp = p.grp_rows ? m0 / p.grp_rows : 0;
  // DMA sources: instruction (mi, j) stages pixels j*8 .. j*8+7 of fragment mi, lane -> (pixel = lane >> 3, chunk = (lane & 7) ^ pixel)
  const unsigned char *xsrc[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      int m = min(m0 + mi * 16 + j * 8 + (lane >> 3), p.M - 1);   // rows past M re-read the last pixel, never stored
      if (p.in_shared) m -= grp * p.grp_rows;
      const int img = m / ohw, rem = m - img * ohw;
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      const int ih0 = oh * p.stride - p.pad + p.ipad, iw0 = ow * p.stride - p.pad + p.ipad;
      xsrc[mi][j] = p.in + ((size_t)(img * IHp + ih0) * IWp + iw0) * p.cin_b + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);
    }
  const unsigned char *wrow = p.wfrag + (p.grp_rows ? (size_t)grp * p.grp_w_bytes : 0) + (size_t)(n0 >> 4) * ((size_t)16 * p.krow_b) + lane * 16;
  const size_t wtile = (size_t)16 * p.krow_b;
  const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + wave * RING;
  const int li = lane & 15, g = lane >> 4;
  const unsigned char *xrd = smem + wave * RING + li * 128;
  const int xsl[2] = {(g ^ (li & 7)) << 4, ((4 + g) ^ (li & 7)) << 4};

  f4 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; a++)
#pragma unroll
    for (int b = 0; b < MI; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

  const int KT = p.krow_b >> 7;
  const int n_my = (KT - wave + 3) >> 2;      // steps wave, wave+4, ...
  i4 wf[PF][2][NI];
  auto issue = [&](const int slot, i4 (&w)[2][NI], int kt) {
    const unsigned ko = p.koff[kt];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
      for (int j = 0; j < 2; j++) glds16_asm(xsrc[mi][j] + ko, ring_lds + slot * STAGE + mi * 2048 + j * 1024);
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
      const unsigned char *wp = wrow + ni * wtile + (size_t)kt * 2048;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[0][ni]) : "v"(wp) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(w[1][ni]) : "v"(wp) : "memory");
    }
  };
  auto wait_steps = [&](int later) {  // wave-uniform: the number of K-steps issued after the one about to be consumed
    if (later >= PF - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * L) : "memory");
    else if (later == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * L <= 63 ? 3 * L : 63) : "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L) : "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto consume = [&](const int slot, i4 (&w)[2][NI]) {
    // (the weight registers are asm outputs: tie them to this point so that no MFMA is scheduled above the wait)
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int ni = 0; ni < NI; ni++) asm volatile("" : "+v"(w[ks][ni]));
    __builtin_amdgcn_sched_barrier(0);
    i4 x[2][MI];
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int mi = 0; mi < MI; mi++) x[ks][mi] = *reinterpret_cast<const i4 *>(xrd + slot * STAGE + mi * 2048 + xsl[ks]);
    mma_kstep<DT, NI, MI>(acc, w, x);
    __builtin_amdgcn_sched_barrier(0);
  };
  int issued = 0;
#pragma unroll
  for (int s = 0; s < PF; s++)
    if (s < n_my) { issue(s, wf[s], wave + 4 * s); issued++; }
  for (int base = 0; base < n_my; base += PF) {
#pragma unroll
    for (int s = 0; s < PF; s++) {
      const int step = base + s;
      if (step < n_my) {
        wait_steps(issued - step - 1);
        consume(s, wf[s]);
        if (issued < n_my) { issue(s, wf[s], wave + 4 * issued); issued++; }   // (issued == step + PF here)
      }
    }
  }
  // ---- fixed-order reduction of the four partial sums: waves 1..3 publish, wave 0 adds them in order and stores
  if (wave > 0) {
    f4 *dst = reinterpret_cast<f4 *>(red + (wave - 1) * (NI * MI * 1024)) + lane;
#pragma unroll
    for (int a = 0; a < NI; a++)
#pragma unroll
      for (int b = 0; b < MI; b++) dst[(a * MI + b) * 64] = acc[a][b];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; w++) {
    const f4 *src = reinterpret_cast<const f4 *>(red + w * (NI * MI * 1024)) + lane;
#pragma unroll
    for (int a = 0; a < NI; a++)
#pragma unroll
      for (int b = 0; b < MI; b++) {
        const f4 v = src[(a * MI + b) * 64];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if constexpr (DT == DT_I8) acc[a][b][r] = __int_as_float(__float_as_int(acc[a][b][r]) + __float_as_int(v[r]));   // int32 partial sums
          else acc[a][b][r] = acc[a][b][r] + v[r];   // (component-wise: no v_pk_add_f32)
        }
      }
  }
  conv_epilogue<MI, NI, DT, ODT, POST ? 4 : 0>(p, acc, m0, n0, lane);
}

// -------------------------------------------------------------------------------------------------
// Ping-pong variant: 256 pixels x BN channels per workgroup, 8 waves = two groups of 4 (each group owns 128 pixel
// rows with the usual 2x2 arrangement of 64 x BN/2 wave tiles, the W tile is shared).  The groups run in strict
// anti-phase: while group 0 pulls its 16 operand fragments of K-step kt from LDS into registers and issues the LDS-DMA
// for K-step kt+2, group 1 issues the 32 MFMAs of K-step kt-1 from registers only, and vice versa.  Each SIMD hosts one
// wave of either group, so its matrix pipe always has a pure-MFMA wave to run.  Three LDS stages; loads stay in
// flight across the barriers (counted vmcnt, raw s_barrier); two barriers per K-step.
// -------------------------------------------------------------------------------------------------
template <int BN, int DT, int ODT = DT>
__global__ __launch_bounds__(512, 2) void conv_pp_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = 256;
  constexpr int XB = BM * 128;
  constexpr int WB = BN * 128;
  constexpr int STAGE = XB + WB;
  constexpr int NREP = BN / 32;
  constexpr int WPIECES = BN / 64;
  constexpr int G = 4 + WPIECES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                 // 0 / 1: pixel rows [grp*128, +128)
  const int wq = wave & 3, wm = wq & 1, wn = wq >> 1;
  const int n_tiles = p.Cout / BN;
  int logical;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, within = b >> 3, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int mt = logical / n_tiles, nt = logical - mt * n_tiles;
  const int m0 = p.m_begin + mt * BM, n0 = nt * BN;
  const int ohw = p.OH * p.OW;
  const int IHp = p.H + 2 * p.ipad, IWp = p.W + 2 * p.ipad;

  const int srow = lane >> 3;
  const int g = (lane & 7) ^ srow;
  unsigned xoff[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int m = min(m0 + (wave * 4 + i) * 8 + srow, p.M - 1);
    int img = m / ohw;
    int rem = m - img * ohw;
    int oh = rem / p.OW, ow = rem - oh * p.OW;
    int ih0 = oh * p.stride - p.pad + p.ipad, iw0 = ow * p.stride - p.pad + p.ipad;
    xoff[i] = (unsigned)(((img * IHp + ih0) * IWp + iw0) * p.cin_b + g * 16);
  }
  unsigned woffv[WPIECES];
#pragma unroll
  for (int i = 0; i < WPIECES; i++) {
    int row = (wave * WPIECES + i) * 8 + srow;
    woffv[i] = (unsigned)((n0 + row) * p.krow_b + g * 16);
  }
  const unsigned char *in_b = p.in;
  const unsigned char *w_b = p.w;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

  // (ConvParams::wdeep: the 16 pieces of a 128-row x 128-byte stage in piece order -- this wave's two are one 2 KB run)
  const bool packed = BN == 128 && p.wdeep != nullptr;   // (wave-uniform: a scalar branch)
  const unsigned char *wpk = p.wdeep + ((size_t)nt * (p.krow_b >> 7) * 16 + wave * 2) * 1024 + lane * 16;
  auto stage = [&](int kt, int buf) {
    const unsigned xs = lds_base + buf * STAGE;
    const unsigned ws = xs + XB;
    const unsigned char *xb = in_b + p.koff[kt];
    const unsigned char *wb = w_b + (size_t)kt * 128;
#pragma unroll
    for (int i = 0; i < 4; i++) glds16_asm(xb + xoff[i], xs + (wave * 4 + i) * 1024);
    if (packed) { glds16x2_asm(wpk + (size_t)kt * 16384, ws + wave * 2048); return; }
#pragma unroll
    for (int i = 0; i < WPIECES; i++) glds16_asm(wb + woffv[i], ws + (wave * WPIECES + i) * 1024);
  };

  f4 acc[NREP][4];
#pragma unroll
  for (int a = 0; a < NREP; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fk = lane >> 4;
  int xfo[2], wfo[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    int slot = (ks * 4 + fk) ^ (lane & 7);
    xfo[ks] = (grp * 128 + wm * 64 + frow) * 128 + slot * 16;
    wfo[ks] = XB + (wn * (BN / 2) + frow) * 128 + slot * 16;
  }

  i4 xf[2][4], wf[2][NREP];
  auto ldfrags = [&](int buf) {
    const unsigned char *sb = smem + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
      for (int mi = 0; mi < 4; mi++) xf[ks][mi] = *reinterpret_cast<const i4 *>(sb + xfo[ks] + mi * 16 * 128);
#pragma unroll
      for (int ni = 0; ni < NREP; ni++) wf[ks][ni] = *reinterpret_cast<const i4 *>(sb + wfo[ks] + ni * 16 * 128);
    }
  };
  auto mfmas = [&]() {
    __builtin_amdgcn_s_setprio(1);
    mma_kstep<DT, NREP, 4>(acc, wf, xf);
    __builtin_amdgcn_s_setprio(0);
  };
#define FP_PP_BARRIER()                  \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    asm volatile("" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

  const int KT = p.krow_b >> 7;  // >= 8
  stage(0, 0);
  stage(1, 1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");  // tile 0 landed (this wave's pieces)
  FP_PP_BARRIER();
  int rb = 0, wb3 = 2;
  if (grp == 0) {
    for (int kt = 0; kt < KT; kt++) {
      // even slot: LDS -> registers for K-step kt, prefetch K-step kt+2
      ldfrags(rb);
      if (kt + 2 < KT) stage(kt + 2, wb3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      FP_PP_BARRIER();
      // odd slot: MFMAs from registers
      mfmas();
      if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");  // tile kt+1 landed, kt+2 in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      FP_PP_BARRIER();
      rb = (rb == 2) ? 0 : rb + 1;
      wb3 = (wb3 == 2) ? 0 : wb3 + 1;
    }
  } else {
    for (int kt = 0; kt < KT; kt++) {
      // even slot: MFMAs of K-step kt-1
      if (kt > 0) mfmas();
      FP_PP_BARRIER();
      // odd slot: LDS -> registers for K-step kt, prefetch K-step kt+2
      ldfrags(rb);
      if (kt + 2 < KT) {
        stage(kt + 2, wb3);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      FP_PP_BARRIER();
      rb = (rb == 2) ? 0 : rb + 1;
      wb3 = (wb3 == 2) ? 0 : wb3 + 1;
    }
    mfmas();
  }
#undef FP_PP_BARRIER

  conv_epilogue<4, NREP, DT, ODT>(p, acc, m0 + grp * 128 + wm * 64, n0 + wn * (BN / 2), lane);
}

// -------------------------------------------------------------------------------------------------
// 256 x 256 tile.  Ablation of the 128 x 128 kernel (tools/bench_conv.py variants 17/21/25) shows its K-step is bound
// by the global -> LDS path, not by the matrix pipe: with the MFMAs removed a K-step still takes ~1360 cycles against
// ~1050 with the loads removed, i.e. the two co-resident workgroups pull 64 KB per K-step at ~47 B/clk/CU, the L2 ->
// CU ceiling.  A 256 x 256 tile moves the same 64 KB per K-step for TWICE the MFMA work, which puts the K-step back
// under the matrix pipe.  8 waves (2 along pixels x 4 along channels), wave tile 128 x 64 (32 accumulators), BK = 64,
// two 64-KB LDS stages, one workgroup per CU.  Needs Cout % 256 == 0.
// -------------------------------------------------------------------------------------------------
// Ping-pong schedule of the 256 x 256 tile (see the slot comment inside).
template <int ABL, int DT, int ODT = DT, bool POST = false, bool LSTORE = false>  // LSTORE: epilogue stores staged through LDS (conv_epilogue_lds)
__global__ __launch_bounds__(512, 2) void conv_big_pp_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = 256, BN = 256;
  constexpr int XB = BM * 128, WB = BN * 128, STAGE = XB + WB;
  constexpr int MI = 8, NI = 4;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wm doubles as the ping-pong group: waves w and w+4 share a SIMD
  const int n_tiles = p.Cout / BN;
  int logical;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, within = b >> 3, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int mt = logical / n_tiles, nt = logical - mt * n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int ohw = p.OH * p.OW;
  const int IHp = p.H + 2 * p.ipad, IWp = p.W + 2 * p.ipad;

  const int srow = lane >> 3;
  const int g = (lane & 7) ^ srow;
  unsigned xoff[4], woffv[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int m = min(m0 + (wave * 4 + i) * 8 + srow, p.M - 1);
    int img = m / ohw;
    int rem = m - img * ohw;
    int oh = rem / p.OW, ow = rem - oh * p.OW;
    int ih0 = oh * p.stride - p.pad + p.ipad, iw0 = ow * p.stride - p.pad + p.ipad;
    xoff[i] = (unsigned)(((img * IHp + ih0) * IWp + iw0) * p.cin_b + g * 16);
    int row = (wave * 4 + i) * 8 + srow;
    woffv[i] = (unsigned)((n0 + row) * p.krow_b + g * 16);
  }
  const unsigned char *in_b = p.in;
  const unsigned char *w_b = p.w;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

  // the 8 LDS-DMA instructions of a K-step are issued in two halves (X pieces in the wave's L0 slot, W pieces at the
  // head of its M0 slot) so the address path (64 B/clk/CU) sees them spread over three slots instead of bunched in two
  auto stage_x = [&](int kt, int buf) {
    const unsigned xs = lds_base + buf * STAGE;
    const unsigned char *xb = in_b + p.koff[kt];
#pragma unroll
    for (int i = 0; i < 4; i++) glds16_asm(xb + xoff[i], xs + (wave * 4 + i) * 1024);
  };
  // (ConvParams::wpack128, pack_stage_w128) this wave's 4 KB of every K-step as one run: one address + one M0 per stage
  const bool packed = p.wpack128 != nullptr;   // (wave-uniform: a scalar branch)
  const unsigned char *wpk = p.wpack128 + ((size_t)nt * (p.krow_b >> 7) * 8 + wave) * 4096 + lane * 16;
  auto stage_w = [&](int kt, int buf) {
    const unsigned ws = lds_base + buf * STAGE + XB;
    if (packed) { glds16x4_asm(wpk + (size_t)kt * 32768, ws + wave * 4096); return; }
    const unsigned char *wb = w_b + (size_t)kt * 128;
#pragma unroll
    for (int i = 0; i < 4; i++) glds16_asm(wb + woffv[i], ws + (wave * 4 + i) * 1024);
  };

  f4 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; a++)
#pragma unroll
    for (int b = 0; b < MI; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fk = lane >> 4;
  int xfo[2], wfo[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    int slot = (ks * 4 + fk) ^ (lane & 7);
    xfo[ks] = (wm * 128 + frow) * 128 + slot * 16;
    wfo[ks] = XB + (wn * 64 + frow) * 128 + slot * 16;
  }

#define FP_BAR()                         \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    asm volatile("" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
#define FP_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define FP_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
  // Slots of ~32 MFMAs (16 for FP8, each twice as long): group 0 runs L0 M0 L1 M1 per K-step, group 1 the same one slot later,
  // so on every SIMD one wave issues MFMAs from registers while its partner refills fragments from LDS / issues the next
  // tile's LDS-DMA.  The slot sequence is the same for every element type (FP_PP_LOOP); what a half-slot reads and
  // multiplies differs:
  //   2-byte types: half h = the 32-wide k-step h over all 8 pixel fragments (12 fragment reads, 32 MFMAs);
  //   FP8: one 128-wide MFMA consumes the whole 128-byte row, so the halves split the PIXEL fragments instead: half 0 reads
  //        the 4 weight operands (kept for both halves) + pixel fragments 0..3, half 1 pixel fragments 4..7.
#define FP_PP_LOOP(LD0, LD1, MF0, MF1)                                   \
  do {                                                                   \
    const int KT = p.krow_b >> 7;                                        \
    stage_x(0, 0);                                                       \
    stage_w(0, 0);                                                       \
    FP_VM0();                                                            \
    FP_BAR();                                                            \
    if (p.clk && tid == 0) { p.clk[blockIdx.x * 4] = __builtin_readcyclecounter(); p.clk[blockIdx.x * 4 + 1] = wall_clock64(); } \
    int buf = 0;                                                         \
    if (wm == 0) {                                                       \
      for (int kt = 0; kt < KT; kt++) {                                  \
        LD0(buf);                                                        \
        if (kt + 1 < KT && !(ABL & 1)) stage_x(kt + 1, buf ^ 1);         \
        FP_LGKM0(); FP_BAR();                                            \
        if (kt + 1 < KT && !(ABL & 1)) stage_w(kt + 1, buf ^ 1);         \
        MF0(); FP_BAR();                                                 \
        LD1(buf); FP_LGKM0(); FP_BAR();                                  \
        MF1(); FP_VM0(); FP_BAR();                                       \
        buf ^= 1;                                                        \
      }                                                                  \
      FP_BAR();                                                          \
    } else {                                                             \
      FP_BAR();                                                          \
      for (int kt = 0; kt < KT; kt++) {                                  \
        LD0(buf);                                                        \
        if (kt + 1 < KT && !(ABL & 1)) stage_x(kt + 1, buf ^ 1);         \
        FP_LGKM0(); FP_BAR();                                            \
        if (kt + 1 < KT && !(ABL & 1)) stage_w(kt + 1, buf ^ 1);         \
        MF0(); FP_BAR();                                                 \
        LD1(buf); FP_LGKM0(); FP_VM0(); FP_BAR();                        \
        MF1(); FP_BAR();                                                 \
        buf ^= 1;                                                        \
      }                                                                  \
    }                                                                    \
  } while (0)

  if constexpr (is_q8(DT)) {
    i8 xv[MI / 2], wv[NI];
    auto ld0 = [&](int buf) {
      const unsigned char *sb = smem + buf * STAGE;
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
        wv[ni] = __builtin_shufflevector(*reinterpret_cast<const i4 *>(sb + wfo[0] + ni * 16 * 128),
                                         *reinterpret_cast<const i4 *>(sb + wfo[1] + ni * 16 * 128), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int j = 0; j < MI / 2; j++)
        xv[j] = __builtin_shufflevector(*reinterpret_cast<const i4 *>(sb + xfo[0] + j * 16 * 128),
                                        *reinterpret_cast<const i4 *>(sb + xfo[1] + j * 16 * 128), 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto ld1 = [&](int buf) {
      const unsigned char *sb = smem + buf * STAGE;
#pragma unroll
      for (int j = 0; j < MI / 2; j++)
        xv[j] = __builtin_shufflevector(*reinterpret_cast<const i4 *>(sb + xfo[0] + (MI / 2 + j) * 16 * 128),
                                        *reinterpret_cast<const i4 *>(sb + xfo[1] + (MI / 2 + j) * 16 * 128), 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto mf0 = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int j = 0; j < MI / 2; j++) acc[ni][j] = mfma8<DT>(wv[ni], xv[j], acc[ni][j]);
      __builtin_amdgcn_s_setprio(0);
    };
    auto mf1 = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int j = 0; j < MI / 2; j++) acc[ni][MI / 2 + j] = mfma8<DT>(wv[ni], xv[j], acc[ni][MI / 2 + j]);
      __builtin_amdgcn_s_setprio(0);
    };
    FP_PP_LOOP(ld0, ld1, mf0, mf1);
  } else {
    i4 xf[MI], wf[NI];
    auto ldk = [&](int buf, int ks) {
      const unsigned char *sb = smem + buf * STAGE;
#pragma unroll
      for (int mi = 0; mi < MI; mi++) xf[mi] = *reinterpret_cast<const i4 *>(sb + xfo[ks] + mi * 16 * 128);
#pragma unroll
      for (int ni = 0; ni < NI; ni++) wf[ni] = *reinterpret_cast<const i4 *>(sb + wfo[ks] + ni * 16 * 128);
    };
    auto ld0 = [&](int buf) { ldk(buf, 0); };
    auto ld1 = [&](int buf) { ldk(buf, 1); };
    auto mf = [&]() {
      if (ABL & 2) {  // ablation: keep fragments live, no MFMAs
#pragma unroll
        for (int mi = 0; mi < MI; mi++) asm volatile("" ::"v"(xf[mi]));
#pragma unroll
        for (int ni = 0; ni < NI; ni++) asm volatile("" ::"v"(wf[ni]));
        return;
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
          acc[ni][mi] = mfma32<DT>(wf[ni], xf[mi], acc[ni][mi]);
      __builtin_amdgcn_s_setprio(0);
    };
    FP_PP_LOOP(ld0, ld1, mf, mf);
  }
#undef FP_PP_LOOP
#undef FP_BAR
#undef FP_LGKM0
#undef FP_VM0
  if (p.clk && tid == 0) { p.clk[blockIdx.x * 4 + 2] = __builtin_readcyclecounter(); p.clk[blockIdx.x * 4 + 3] = wall_clock64(); }

  if (ABL & 4) {  // ablation: no epilogue (keep the accumulators live)
#pragma unroll
    for (int a = 0; a < NI; a++)
#pragma unroll
      for (int b = 0; b < MI; b++) asm volatile("" ::"v"(acc[a][b]));
    return;
  }
  if constexpr (LSTORE && ABL == 0 && !POST && !is_q8(DT) && odt_q(ODT) < 0) {
    __syncthreads();   // both ping-pong groups are done with the ring: 8 x (8 KB tile + 256 B offsets) of it become the staging area
    conv_epilogue_lds<MI, NI, DT, ODT, 4>(p, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * (4 * 16 * 128 + 4 * 64));
    return;
  }
  conv_epilogue<MI, NI, DT, ODT, ((ABL >> 3) & 3) | (POST ? 4 : 0)>(p, acc, m0 + wm * 128, n0 + wn * 64, lane);
}

// -------------------------------------------------------------------------------------------------
// conv_pp32_kernel<BM,BN>: the ping-pong schedule on 32-wide K-steps with a ring of FOUR LDS stages.
// Ablation of conv_big_pp_kernel (64-wide K-steps, 2 stages) shows it is bound by the latency of the global -> LDS
// path: with the MFMAs removed a K-step still takes ~2190 cycles (64 KB in flight per CU), against ~1970 with the loads
// removed.  Halving the K-step and doubling the ring keeps the same 128 KB of LDS but lets three K-steps (96 KB) be in
// flight, with ~5 slots between issue and first use instead of ~3.
//   tile BM x BN, 8 waves = (BM/128) x (BN/64), wave tile 128 px x 64 ch (32 accumulators); (256,256) and (512,128).
//   stage = [BM + BN rows][64 B]; a DMA piece is 16 rows x 64 B; slot = chunk ^ G[(row>>2)&3], G = {0,2,3,1}, is
//   conflict-free for the four 16-lane groups of ds_read_b128 with 64-byte rows (checked by enumeration).
// -------------------------------------------------------------------------------------------------
template <int BM, int BN, int DT, int ODT = DT>
__global__ __launch_bounds__(512, 2) void conv_pp32_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WM = BM / 128, WN = BN / 64;
  static_assert(WM * WN == 8, "8 waves");
  constexpr int XP = BM / 16, WP = BN / 16, PER = (XP + WP) / 8;  // DMA pieces per stage / per wave
  constexpr int XB = XP * 1024, STAGE = (XP + WP) * 1024;
  constexpr int MI = 8, NI = 4;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;  // waves w and w+4 share a SIMD: the two ping-pong groups
  const int wm = wave / WN, wn = wave - wm * WN;
  const int n_tiles = p.Cout / BN;
  int logical;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, within = b >> 3, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int mt = logical / n_tiles, nt = logical - mt * n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int ohw = p.OH * p.OW;
  const int IHp = p.H + 2 * p.ipad, IWp = p.W + 2 * p.ipad;

  // DMA roles: piece = wave + 8*i; lane -> (row = lane>>2, slot = lane&3); source chunk = slot ^ G[(row>>2)&3]
  const int prow = lane >> 2;
  const int gsel = (prow >> 2) & 3;
  const int gch = (lane & 3) ^ ((0x78 >> (gsel * 2)) & 3);  // G = {0,2,3,1} packed two bits each = 0x78
  unsigned poff[PER];
#pragma unroll
  for (int i = 0; i < PER; i++) {
    const int piece = wave + 8 * i;
    if (piece < XP) {
      int m = min(m0 + piece * 16 + prow, p.M - 1);
      int img = m / ohw;
      int rem = m - img * ohw;
      int oh = rem / p.OW, ow = rem - oh * p.OW;
      int ih0 = oh * p.stride - p.pad + p.ipad, iw0 = ow * p.stride - p.pad + p.ipad;
      poff[i] = (unsigned)(((img * IHp + ih0) * IWp + iw0) * p.cin_b + gch * 16);
    } else {
      int row = (piece - XP) * 16 + prow;
      poff[i] = (unsigned)((n0 + row) * p.krow_b + gch * 16);
    }
  }
  const unsigned char *in_b = p.in;
  const unsigned char *w_b = p.w;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

  auto stage = [&](int st, int buf) {
    const unsigned sb = lds_base + buf * STAGE;
    const unsigned char *xb = in_b + p.koff32[st];
    const unsigned char *wb = w_b + (size_t)st * 64;
#pragma unroll
    for (int i = 0; i < PER; i++) {
      const int piece = wave + 8 * i;
      glds16_asm((piece < XP ? xb : wb) + poff[i], sb + piece * 1024);
    }
  };

  f4 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; a++)
#pragma unroll
    for (int b = 0; b < MI; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

  // fragment addresses: row-in-tile r, chunk kg = lane>>4, slot = kg ^ G[(r>>2)&3]; (r>>2)&3 == ((lane&15)>>2)
  const int fsel = (lane & 15) >> 2;
  const int fslot = (lane >> 4) ^ ((0x78 >> (fsel * 2)) & 3);
  const int xfo = (wm * 128 + (lane & 15)) * 64 + fslot * 16;
  const int wfo = XB + (wn * 64 + (lane & 15)) * 64 + fslot * 16;

  i4 xf[MI], wf[NI];
  auto ld = [&](int buf) {
    const unsigned char *sb = smem + buf * STAGE;
#pragma unroll
    for (int mi = 0; mi < MI; mi++) xf[mi] = *reinterpret_cast<const i4 *>(sb + xfo + mi * 16 * 64);
#pragma unroll
    for (int ni = 0; ni < NI; ni++) wf[ni] = *reinterpret_cast<const i4 *>(sb + wfo + ni * 16 * 64);
  };
  auto mfmas = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int mi = 0; mi < MI; mi++)
        acc[ni][mi] = mfma32<DT>(wf[ni], xf[mi], acc[ni][mi]);
    __builtin_amdgcn_s_setprio(0);
  };
#define FP_BAR()                         \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    asm volatile("" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
#define FP_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define FP_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  // after the (optional) issue for step s+3, make sure this wave's pieces of step s+1 have landed
#define FP_WAIT_NEXT(s)                                  \
  do {                                                   \
    if ((s) + 3 < S) FP_VM(2 * PER);                     \
    else if ((s) + 2 < S) FP_VM(PER);                    \
    else FP_VM(0);                                       \
  } while (0)

  const int S = p.krow_b >> 6;  // 32-wide K-steps, >= 16 for every layer
  stage(0, 0);
  stage(1, 1);
  stage(2, 2);
  FP_VM(2 * PER);
  FP_BAR();
  int rb = 0, wb = 3;
  if (grp == 0) {
    for (int s = 0; s < S; s++) {
      ld(rb);
      if (s + 3 < S) stage(s + 3, wb);
      FP_LGKM0(); FP_BAR();
      mfmas();
      FP_WAIT_NEXT(s);
      FP_BAR();
      rb = (rb + 1) & 3; wb = (wb + 1) & 3;
    }
    FP_BAR();
  } else {
    FP_BAR();
    for (int s = 0; s < S; s++) {
      ld(rb);
      if (s + 3 < S) stage(s + 3, wb);
      FP_LGKM0();
      FP_WAIT_NEXT(s);
      FP_BAR();
      mfmas();
      FP_BAR();
      rb = (rb + 1) & 3; wb = (wb + 1) & 3;
    }
  }
#undef FP_BAR
#undef FP_LGKM0
#undef FP_VM
#undef FP_WAIT_NEXT
  conv_epilogue<MI, NI, DT, ODT>(p, acc, m0 + wm * 128, n0 + wn * 64, lane);
}

// -------------------------------------------------------------------------------------------------
// conv_halo_kernel<TW>: 3x3 / stride 1 convolution with the INPUT TILE + HALO resident in LDS.
// The implicit-GEMM schedules above re-fetch every input pixel once per tap (9x) through the global -> LDS path, which
// is what bounds them (~47 B/clk/CU).  Here a workgroup owns an 8-row x TW-column output tile of one image and stages
// the (8+2) x (TW+2) halo tile of a 64-channel chunk ONCE (the padded image rows are contiguous in memory, so the halo
// tile is one linear run of pixels); the 9 taps are just 9 shifted LDS windows.  Only the weights stream per K-step
// (8 KB per 32-wide step, 3-stage ring).  Global -> LDS traffic per flop is ~2x below the 256x256 tile's, small enough
// that TWO workgroups (4 waves, 160 accumulators each) share a CU: one's prologue / halo reload / epilogue store burst
// overlaps the other's MFMAs, which the one-workgroup-per-CU 256x256 tile cannot do.
//   M fragment = a 4x4 pixel block (lane&15 -> dy = >>2, dx = &3); wave (wm, wn) owns block-row wm (TW/4 blocks) x 64 ch.
//   halo LDS layout: pixel-major 128-byte rows, 16-byte slot = chunk ^ g, g = ((x>>1)&1) | ((y&3)<<1): conflict-free for
//   every tap shift (checked by enumeration); the swizzle is applied by the DMA on the SOURCE chunk.
// -------------------------------------------------------------------------------------------------
template <int TW, int ABL, int DT>  // ABL (timing ablations, wrong results): 1 no per-step barrier, 2 no MFMAs, 4 X fragments read once
__global__ __launch_bounds__(256, 2) void conv_halo_kernel(const ConvParams p) {
  static_assert(!is_q8(DT), "2-byte element types (64-channel chunks); the 8-bit sibling is conv_halo8_kernel");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TH = 8, HC = TW + 2, HPX = (TH + 2) * HC;
  constexpr int HPIECES = (HPX + 7) / 8, HALO_B = HPIECES * 1024;
  constexpr int HPER = (HPIECES + 3) / 4;  // halo DMA pieces per wave
  constexpr int WST = 128 * 64, NWST = 3;  // weight ring: 128 rows x 64 B per 32-wide K-step
  constexpr int MI = TW / 4, NI = 4;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int n_tiles = p.Cout / 128;
  int logical;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, within = b >> 3, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int mt = logical / n_tiles, nt = logical - mt * n_tiles;
  const int tiles_per_img = p.H / TH;
  const int img = mt / tiles_per_img, ty0 = (mt - img * tiles_per_img) * TH;
  const int n0 = nt * 128;
  const int IHp = p.H + 2;

  const unsigned char *in_b = p.in + ((size_t)(img * IHp + ty0) * HC) * p.cin_b;
  const unsigned char *w_b = p.w;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
  const unsigned w_lds = lds_base + HALO_B;

  // halo DMA: piece = wave + 4*i covers halo pixels piece*8 .. +7; lane -> (pixel = lane>>3, slot = lane&7)
  auto issue_halo = [&](int chunk) {
    const unsigned char *src = in_b + chunk * 128;
    int lane8;  // opaque copy of lane>>3: keeps the 14 per-lane offsets from being hoisted out of the chunk loop (VGPRs)
    asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(lane8) : "v"(lane));
#pragma unroll
    for (int i = 0; i < HPER; i++) {
      const int piece = wave + 4 * i;
      if (piece < HPIECES) {
        int q = min(piece * 8 + lane8, HPX - 1);
        int hy = q / HC, hx = q - hy * HC;
        int g = ((hx >> 1) & 1) | ((hy & 3) << 1);
        unsigned off = (unsigned)(q * p.cin_b + (((lane & 7) ^ g) << 4));
        glds16_asm_x(src + off, lds_base + piece * 1024);
        __builtin_amdgcn_sched_barrier(0);  // one address at a time: 14 hoisted 64-bit addresses would spill accumulators
      }
    }
  };
  // weight DMA: piece = wave*2 + i (16 rows x 64 B); lane -> (row = lane>>2, slot = lane&3), source chunk = slot ^ G
  const int prow = lane >> 2;
  const int gch = (lane & 3) ^ ((0x78 >> (((prow >> 2) & 3) * 2)) & 3);
  unsigned woff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) woff[i] = (unsigned)((n0 + (wave * 2 + i) * 16 + prow) * p.krow_b + gch * 16);
  // (ConvParams::wpack, pack_stage_w) this wave's 2 KB of every K-step as one run: one address + one M0 per stage
  const bool packed = p.wpack != nullptr;   // (wave-uniform: a scalar branch)
  const unsigned char *wpk = p.wpack + ((size_t)nt * (p.krow_b >> 6) * 4 + wave) * 2048 + lane * 16;
  auto issue_w = [&](int st) {
    const unsigned dst = w_lds + (st % NWST) * WST;
    if (packed) { glds16x2_asm(wpk + (size_t)st * 8192, dst + wave * 2048); return; }
    const unsigned char *wb = w_b + (size_t)st * 64;
#pragma unroll
    for (int i = 0; i < 2; i++) glds16_asm(wb + woff[i], dst + (wave * 2 + i) * 1024);
  };

  f4 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; a++)
#pragma unroll
    for (int b = 0; b < MI; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, dy = li >> 2, dx = li & 3, kg = lane >> 4;
  const int fslot = kg ^ ((0x78 >> ((li >> 2) * 2)) & 3);
  const int wfo = HALO_B + (wn * 64 + li) * 64 + fslot * 16;

  const int S = p.krow_b >> 6;   // 32-wide K-steps: 18 per 64-channel chunk (9 taps x 2)
  const int nch = p.cin_b >> 7;
  issue_halo(0);
  issue_w(0);
  issue_w(1);
  if (p.clk && tid == 0) { p.clk[blockIdx.x * 4] = __builtin_readcyclecounter(); p.clk[blockIdx.x * 4 + 1] = wall_clock64(); }
  int s = 0;
  for (int ch = 0; ch < nch; ch++) {
    for (int tap = 0; tap < 9; tap++) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int ty = wm * 4 + dy + ky, tx = dx + kx;
      const int g = ((tx >> 1) & 1) | ((ty & 3) << 1);
      const int pix_off = (ty * HC + tx) * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ks++, s++) {
        // W(s) (and, on a chunk's first step, the halo tile) landed; everyone finished reading step s-1
        if (((tap == 0 && ks == 0) && (!(ABL & 16) || s == 0)) || s == S - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (!(ABL & 1) || (tap == 0 && ks == 0)) __builtin_amdgcn_s_barrier();
        // s_barrier is IntrNoMem for the compiler: without this fence the fragment loads below may be placed ABOVE the
        // barrier on the steps that issue no DMA (the last two), reading weight pieces other waves have not landed yet
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < S) issue_w(s + 2);
        const unsigned char *xs = smem + pix_off + (((ks * 4 + kg) ^ g) << 4);
        const unsigned char *ws = smem + wfo + (s % NWST) * WST;
        if constexpr ((ABL & ~16) == 0) {  // (any ablation flag but 16, e.g. 32: the round-1 schedule -- two halves of 5 fragments; 16: the halo tile is never reloaded)
          // X fragments in four groups (3,2,3,2 of MI = 10) through two small register sets: the LDS reads of group q+1 are in
          // flight while the MFMAs of group q issue, so only the first group of a K-step waits for the LDS with the matrix pipe idle
          static_assert(MI == 10, "group split written for 10 pixel fragments");
          i4 wf[NI], xa[3], xb[2];
#pragma unroll
          for (int ni = 0; ni < NI; ni++) wf[ni] = *reinterpret_cast<const i4 *>(ws + ni * 16 * 64);
#pragma unroll
          for (int i = 0; i < 3; i++) xa[i] = *reinterpret_cast<const i4 *>(xs + i * 512);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 2; i++) xb[i] = *reinterpret_cast<const i4 *>(xs + (3 + i) * 512);
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int i = 0; i < 3; i++) acc[ni][i] = mfma32<DT>(wf[ni], xa[i], acc[ni][i]);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 3; i++) xa[i] = *reinterpret_cast<const i4 *>(xs + (5 + i) * 512);
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int i = 0; i < 2; i++) acc[ni][3 + i] = mfma32<DT>(wf[ni], xb[i], acc[ni][3 + i]);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 2; i++) xb[i] = *reinterpret_cast<const i4 *>(xs + (8 + i) * 512);
          __builtin_amdgcn_sched_barrier(0);
          if (tap == 8 && ks == 1 && ch + 1 < nch && !(ABL & 16)) {  // last reads of this chunk's halo tile are issued: refill it under the MFMAs
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            issue_halo(ch + 1);
          }
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int i = 0; i < 3; i++) acc[ni][5 + i] = mfma32<DT>(wf[ni], xa[i], acc[ni][5 + i]);
#pragma unroll
          for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int i = 0; i < 2; i++) acc[ni][8 + i] = mfma32<DT>(wf[ni], xb[i], acc[ni][8 + i]);
          __builtin_amdgcn_s_setprio(0);
        } else {
          // X fragments in two halves of MI/2 (register budget: 160 accumulators + 20 + 16 fragment registers); the
          // second half's LDS reads are issued behind the first half's MFMAs
          constexpr int HM = MI / 2;
          i4 xf[HM], wf[NI];
#pragma unroll
          for (int ni = 0; ni < NI; ni++) wf[ni] = *reinterpret_cast<const i4 *>(ws + ni * 16 * 64);
#pragma unroll
          for (int mi = 0; mi < HM; mi++) xf[mi] = *reinterpret_cast<const i4 *>(xs + mi * 512);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (!(ABL & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int mi = 0; mi < HM; mi++) {
              if (ABL & 2) asm volatile("" ::"v"(wf[ni]), "v"(xf[mi]));
              else acc[ni][mi] = mfma32<DT>(wf[ni], xf[mi], acc[ni][mi]);
            }
          if (!(ABL & 8)) __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mi = 0; mi < HM; mi++) xf[mi] = *reinterpret_cast<const i4 *>(xs + (HM + mi) * 512);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (tap == 8 && ks == 1 && ch + 1 < nch) {  // last read of this chunk's halo tile: refill it under the MFMAs
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            issue_halo(ch + 1);
          }
          if (!(ABL & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int mi = 0; mi < HM; mi++) {
              if (ABL & 2) asm volatile("" ::"v"(wf[ni]), "v"(xf[mi]));
              else acc[ni][HM + mi] = mfma32<DT>(wf[ni], xf[mi], acc[ni][HM + mi]);
            }
          if (!(ABL & 8)) __builtin_amdgcn_s_setprio(0);
        }
      }
    }
  }
  if (p.clk && tid == 0) { p.clk[blockIdx.x * 4 + 2] = __builtin_readcyclecounter(); p.clk[blockIdx.x * 4 + 3] = wall_clock64(); }
  conv_epilogue_px<MI, NI, DT, DT>(p, acc, n0 + wn * 64, lane, [&](int mi, int &oimg, int &oh, int &ow) {
    oimg = img;
    oh = ty0 + wm * 4 + dy;
    ow = mi * 4 + dx;
    return true;
  });
}

// -------------------------------------------------------------------------------------------------
// conv_halo8_kernel: the FP8 (e4m3) sibling of conv_halo_kernel<40>: 3x3 / stride 1 on 40x40 maps, input tile + halo of a
// 128-CHANNEL chunk (again 128 bytes per pixel, so the halo layout, its DMA and its swizzle are byte-identical) resident
// in LDS.  One K-step = one tap of the chunk = one v_mfma_f32_16x16x128_f8f6f4 per (pixel block, channel tile): 9
// steps per chunk of 40 MFMAs x 32 cycles per wave -- the same matrix-pipe time per step PAIR as the 2-byte kernel for
// twice the contraction length.
//   Weights: 128 rows x 128 B = 16 KB per step.  Two such stages next to the 53 KB halo would not let two workgroups
//   share a CU (2 x 85 KB > 160 KB), so the weight tile is single-buffered in LDS and double-buffered through
//   REGISTERS: every wave pulls its four 32-byte weight operands first, a second barrier frees the stage, the DMA for
//   the next tap is issued, and the 40 MFMAs (1280 cycles) run from registers while it lands.  LDS: 53 + 16 = 69 KB.
//   X operands are read two pixel blocks at a time under the MFMAs (160 accumulators + 32 + 16 operand registers).
//   Weight stage layout: 128-byte rows, 16-byte slot = chunk ^ (row & 7) (the implicit-GEMM kernels' swizzle).
// -------------------------------------------------------------------------------------------------
template <int DT, int ODT>   // DT = DT_FP8 / DT_I8; ODT = DT (a block's first conv) or DT_DUAL_* (its second: f16 stream + 8-bit copy)
__global__ __launch_bounds__(256, 2) void conv_halo8_kernel(const ConvParams p) {
  static_assert(is_q8(DT), "8-bit operand types (128-channel chunks); the 2-byte sibling is conv_halo_kernel");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TW = 40, TH = 8, HC = TW + 2, HPX = (TH + 2) * HC;
  constexpr int HPIECES = (HPX + 7) / 8, HALO_B = HPIECES * 1024;
  constexpr int HPER = (HPIECES + 3) / 4;  // halo DMA pieces per wave
  constexpr int MI = TW / 4, NI = 4;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int n_tiles = p.Cout / 128;
  int logical;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, within = b >> 3, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int mt = logical / n_tiles, nt = logical - mt * n_tiles;
  const int tiles_per_img = p.H / TH;
  const int img = mt / tiles_per_img, ty0 = (mt - img * tiles_per_img) * TH;
  const int n0 = nt * 128;
  const int IHp = p.H + 2;

  const unsigned char *in_b = p.in + ((size_t)(img * IHp + ty0) * HC) * p.cin_b;
  const unsigned char *w_b = p.w;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
  const unsigned w_lds = lds_base + HALO_B;

  // halo DMA: piece = wave + 4*i covers halo pixels piece*8 .. +7; lane -> (pixel = lane>>3, slot = lane&7)
  auto issue_halo = [&](int chunk) {
    const unsigned char *src = in_b + chunk * 128;
    int lane8;  // opaque copy of lane>>3: keeps the 14 per-lane offsets from being hoisted out of the chunk loop (VGPRs)
    asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(lane8) : "v"(lane));
#pragma unroll
    for (int i = 0; i < HPER; i++) {
      const int piece = wave + 4 * i;
      if (piece < HPIECES) {
        int q = min(piece * 8 + lane8, HPX - 1);
        int hy = q / HC, hx = q - hy * HC;
        int g = ((hx >> 1) & 1) | ((hy & 3) << 1);
        unsigned off = (unsigned)(q * p.cin_b + (((lane & 7) ^ g) << 4));
        glds16_asm_x(src + off, lds_base + piece * 1024);
        __builtin_amdgcn_sched_barrier(0);  // one address at a time: hoisted 64-bit addresses would spill accumulators
      }
    }
  };
  // weight DMA: piece = wave*4 + i (8 rows x 128 B); lane -> (row = lane>>3, slot = lane&7), source chunk = slot ^ row
  const int srow = lane >> 3;
  unsigned woff[4];
#pragma unroll
  for (int i = 0; i < 4; i++) woff[i] = (unsigned)((n0 + (wave * 4 + i) * 8 + srow) * p.krow_b + (((lane & 7) ^ srow) << 4));
  // (ConvParams::wpack, pack_stage_w128) this wave's 4 KB of every K-step as one run: one address + one M0 per stage
  const bool packed = p.wpack != nullptr;   // (wave-uniform: a scalar branch)
  const unsigned char *wpk = p.wpack + ((size_t)nt * (p.krow_b >> 7) * 4 + wave) * 4096 + lane * 16;
  auto issue_w = [&](int st) {
    if (packed) { glds16x4_asm(wpk + (size_t)st * 16384, w_lds + wave * 4096); return; }
    const unsigned char *wb = w_b + (size_t)st * 128;
#pragma unroll
    for (int i = 0; i < 4; i++) glds16_asm(wb + woff[i], w_lds + (wave * 4 + i) * 1024);
  };

  f4 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; a++)
#pragma unroll
    for (int b = 0; b < MI; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, dy = li >> 2, dx = li & 3, kg = lane >> 4;
  // weight operand of channel tile ni: row wn*64 + ni*16 + li, chunks kg and 4 + kg
  const int wfo0 = HALO_B + (wn * 64 + li) * 128 + ((kg ^ (li & 7)) << 4);
  const int wfo1 = HALO_B + (wn * 64 + li) * 128 + (((4 + kg) ^ (li & 7)) << 4);

  const int S = p.krow_b >> 7;   // 9 taps per 128-channel chunk
  const int nch = p.cin_b >> 7;
  issue_halo(0);
  issue_w(0);
  if (p.clk && tid == 0) { p.clk[blockIdx.x * 4] = __builtin_readcyclecounter(); p.clk[blockIdx.x * 4 + 1] = wall_clock64(); }
  int s = 0;
  for (int ch = 0; ch < nch; ch++) {
    for (int tap = 0; tap < 9; tap++, s++) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int ty = wm * 4 + dy + ky, tx = dx + kx;
      const int g = ((tx >> 1) & 1) | ((ty & 3) << 1);
      const unsigned char *xs0 = smem + (ty * HC + tx) * 128 + ((kg ^ g) << 4);
      const unsigned char *xs1 = smem + (ty * HC + tx) * 128 + (((4 + kg) ^ g) << 4);
      // W(s) (and, on a chunk's first tap, the halo tile) landed
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      i8 wv[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
        wv[ni] = __builtin_shufflevector(*reinterpret_cast<const i4 *>(smem + wfo0 + ni * 16 * 128),
                                         *reinterpret_cast<const i4 *>(smem + wfo1 + ni * 16 * 128), 0, 1, 2, 3, 4, 5, 6, 7);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave holds W(s) in registers: the stage is free
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < S) issue_w(s + 1);
      // pixel fragments one at a time through two register sets: the reads of fragment m+1 are in flight while the four MFMAs of
      // fragment m issue (the round-1 schedule read two fragments, waited, issued eight MFMAs, five times per tap)
      auto read_x = [&](int m) {
        return __builtin_shufflevector(*reinterpret_cast<const i4 *>(xs0 + m * 512), *reinterpret_cast<const i4 *>(xs1 + m * 512), 0, 1, 2, 3, 4, 5, 6, 7);
      };
      i8 xv[2];
      xv[0] = read_x(0);
#pragma unroll
      for (int m = 0; m < MI; m++) {
        if (m + 1 < MI) {
          xv[(m + 1) & 1] = read_x(m + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (m == MI - 2 && tap == 8 && ch + 1 < nch) {  // the last read of this chunk's halo tile is issued: refill it under the MFMAs
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          issue_halo(ch + 1);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < NI; ni++) acc[ni][m] = mfma8<DT>(wv[ni], xv[m & 1], acc[ni][m]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (p.clk && tid == 0) { p.clk[blockIdx.x * 4 + 2] = __builtin_readcyclecounter(); p.clk[blockIdx.x * 4 + 3] = wall_clock64(); }
  conv_epilogue_px<MI, NI, DT, ODT>(p, acc, n0 + wn * 64, lane, [&](int mi, int &oimg, int &oh, int &ow) {
    oimg = img;
    oh = ty0 + wm * 4 + dy;
    ow = mi * 4 + dx;
    return true;
  });
}

// -------------------------------------------------------------------------------------------------
// conv_stem_halo_kernel: the space-to-depth stem (4x4 taps, stride 1, 32 -> 64 channels, 84x84 padded input -> 80x80)
// with the same resident-halo scheme as conv_halo_kernel.  The implicit-GEMM stem re-fetched every 64-byte input pixel
// once per tap (16x) for a 128x64 tile and ran at ~365 TFLOP/s of padded work, bound by the global -> LDS path.
//   tile = 8 rows x 80 cols (640 px) x all 64 channels; 4 waves split the pixels (block-row x column half), 160
//   accumulators each; halo tile = (8+3) padded rows x 84 px x 64 B = one linear run of 59 KB staged once; 16 K-steps
//   (one tap each, 32 channels); weights 4 KB per step through a 3-stage ring; two workgroups per CU.
//   64-byte pixel rows: 16-byte slot = chunk ^ (y & 3); conflict-free because the pitch (84) is a multiple of 4.
// -------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256, 2) void conv_stem_halo_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TW = 80, TH = 8, HC = 84, HPX = (TH + 3) * HC;
  constexpr int HPIECES = (HPX + 15) / 16, HALO_B = HPIECES * 1024;
  constexpr int HPER = (HPIECES + 3) / 4;
  constexpr int WST = 64 * 64, NWST = 3;
  constexpr int MI = TW / 8, NI = 4, HM = MI / 2;  // 10 fragments (4x4 pixel blocks) per wave, in two halves
  constexpr int S = 16;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int brow = wave >> 1, chalf = wave & 1;
  int logical;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, within = b >> 3, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int tiles_per_img = p.H / TH;
  const int img = logical / tiles_per_img, ty0 = (logical - img * tiles_per_img) * TH;

  const unsigned char *in_b = p.in + ((size_t)(img * HC + ty0) * HC) * 64;
  const unsigned char *w_b = p.w;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
  const unsigned w_lds = lds_base + HALO_B;

  {  // halo DMA: piece = wave + 4*i covers 16 halo pixels; lane -> (pixel = lane>>2, slot = lane&3)
#pragma unroll
    for (int i = 0; i < HPER; i++) {
      const int piece = wave + 4 * i;
      if (piece < HPIECES) {
        int q = min(piece * 16 + (lane >> 2), HPX - 1);
        int hy = q / HC;
        unsigned off = (unsigned)(q * 64 + (((lane & 3) ^ (hy & 3)) << 4));
        glds16_asm(in_b + off, lds_base + piece * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const int prow = lane >> 2;
  const int gch = (lane & 3) ^ ((0x78 >> (((prow >> 2) & 3) * 2)) & 3);
  const unsigned woff = (unsigned)((wave * 16 + prow) * p.krow_b + gch * 16);
  auto issue_w = [&](int st) { glds16_asm(w_b + (size_t)st * 64 + woff, w_lds + (st % NWST) * WST + wave * 1024); };

  f4 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; a++)
#pragma unroll
    for (int b = 0; b < MI; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, dy = li >> 2, dx = li & 3, kg = lane >> 4;
  const int fslot = kg ^ ((0x78 >> ((li >> 2) * 2)) & 3);
  const int wfo = HALO_B + li * 64 + fslot * 16;

  issue_w(0);
  issue_w(1);
#pragma unroll 1
  for (int s = 0; s < S; s++) {
    const int ky = s >> 2, kx = s & 3;
    const int ty = brow * 4 + dy + ky, tx = chalf * (TW / 2) + dx + kx;
    if (s == 0 || s == S - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < S) issue_w(s + 2);
    const unsigned char *xs = smem + (ty * HC + tx) * 64 + ((kg ^ (ty & 3)) << 4);
    const unsigned char *ws = smem + wfo + (s % NWST) * WST;
    i4 xf[HM], wf[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) wf[ni] = *reinterpret_cast<const i4 *>(ws + ni * 16 * 64);
#pragma unroll
    for (int mi = 0; mi < HM; mi++) xf[mi] = *reinterpret_cast<const i4 *>(xs + mi * 256);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int mi = 0; mi < HM; mi++)
        acc[ni][mi] = mfma32<DT>(wf[ni], xf[mi], acc[ni][mi]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < HM; mi++) xf[mi] = *reinterpret_cast<const i4 *>(xs + (HM + mi) * 256);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int mi = 0; mi < HM; mi++)
        acc[ni][HM + mi] = mfma32<DT>(wf[ni], xf[mi], acc[ni][HM + mi]);
  }
  // Cout = 64: weight rows are permuted in 32-channel blocks (NI = 2 form), so the epilogue runs once per block
  auto pix = [&](int mi, int &oimg, int &oh, int &ow) {
    oimg = img;
    oh = ty0 + brow * 4 + dy;
    ow = chalf * (TW / 2) + mi * 4 + dx;
    return true;
  };
  // EABL = 2: this layer never has a residual (the dispatcher requires it), so the residual path is compiled out
  conv_epilogue_px<MI, 2, DT, DT, 2, 4, 0>(p, acc, 0, lane, pix);
  conv_epilogue_px<MI, 2, DT, DT, 2, 4, 2>(p, acc, 32, lane, pix);
}

// -------------------------------------------------------------------------------------------------
// conv_s2_halo_kernel: the 3x3 / stride-2 convolution 64 -> 128 channels on the 80x80 stem output (encodeA.1), again
// with the input tile resident in LDS.
//   tile = 4 output rows x 40 cols (160 px) x 128 channels; 4 waves = 2 (column halves) x 2 (64 channels), 80
//   accumulators each; 2 workgroups per CU.
//   Stride 2: a tap (ky,kx) reads input columns 2*ox + kx, i.e. one column-PARITY plane at consecutive positions --
//   exactly like a stride-1 tap.  The tile's 9 input rows are therefore staged one parity plane at a time (9 x 41 px x
//   128 B = 47 KB; every fetched 128-byte line is used whole): plane 0 serves the six taps with kx in {0,2} (12
//   32-channel K-steps), plane 1 the three taps with kx = 1 (6 K-steps).  Weights: 8 KB per step, 3-stage ring.
//   16-byte slot of an LDS pixel = chunk ^ (((row>>1)&3) | (((x>>1)&1)<<2)): conflict-free fragment reads.
// -------------------------------------------------------------------------------------------------
template <int DT, int ODT = DT>
__global__ __launch_bounds__(256, 2) void conv_s2_halo_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int IC = 82, HR = 9, PW = 41, HPX = HR * PW;  // 4 output rows need 9 input rows; 41 columns per parity plane
  constexpr int HPIECES = (HPX + 7) / 8, HALO_B = HPIECES * 1024;
  constexpr int HPER = (HPIECES + 3) / 4;
  constexpr int WST = 128 * 64, NWST = 3;
  constexpr int MI = 5, NI = 4, S = 18, SA = 12;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chalf = wave >> 1, wn = wave & 1;
  int logical;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, within = b >> 3, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int img = logical / 10, oy0 = (logical - img * 10) * 4;

  // input image: [82][82][64] halfs (border 1); the tile's first input row is 2*oy0 (padded coordinates)
  const unsigned char *in_b = p.in + ((size_t)(img * IC + 2 * oy0) * IC) * 128;
  const unsigned char *w_b = p.w;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
  const unsigned w_lds = lds_base + HALO_B;

  auto issue_halo = [&](int plane) {  // columns of one parity, all 64 channels
    int lane8;
    asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(lane8) : "v"(lane));  // opaque: keeps the offsets out of long live ranges
#pragma unroll
    for (int i = 0; i < HPER; i++) {
      const int piece = wave + 4 * i;
      if (piece < HPIECES) {
        int q = min(piece * 8 + lane8, HPX - 1);
        int r = q / PW, xp = q - r * PW;
        int c = min(2 * xp + plane, IC - 1);
        int g = ((r >> 1) & 3) | (((xp >> 1) & 1) << 2);
        unsigned off = (unsigned)((r * IC + c) * 128 + (((lane & 7) ^ g) << 4));
        glds16_asm(in_b + off, __builtin_amdgcn_readfirstlane(lds_base + piece * 1024));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  const int prow = lane >> 2;
  const int gch = (lane & 3) ^ ((0x78 >> (((prow >> 2) & 3) * 2)) & 3);
  unsigned woff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) woff[i] = (unsigned)(((wave * 2 + i) * 16 + prow) * p.krow_b + gch * 16);
  // step -> (tap, 32-channel half): steps 0..11 walk the kx in {0,2} taps, 12..17 the kx = 1 taps
  auto step_tap = [&](int st, int &ky, int &kx) {
    if (st < SA) { const int t = st >> 1; ky = t >> 1; kx = (t & 1) * 2; }
    else { ky = (st - SA) >> 1; kx = 1; }
  };
  auto issue_w = [&](int st) {
    int ky, kx;
    step_tap(st, ky, kx);
    const unsigned char *wb = w_b + (size_t)((ky * 3 + kx) * 128 + (st & 1) * 64);
    const unsigned dst = __builtin_amdgcn_readfirstlane(w_lds + (st % NWST) * WST);
#pragma unroll
    for (int i = 0; i < 2; i++) glds16_asm(wb + woff[i], dst + (wave * 2 + i) * 1024);
  };

  f4 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; a++)
#pragma unroll
    for (int b = 0; b < MI; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, dy = li >> 2, dx = li & 3, kg = lane >> 4;
  const int fslot = kg ^ ((0x78 >> ((li >> 2) * 2)) & 3);
  const int wfo = HALO_B + (wn * 64 + li) * 64 + fslot * 16;

  issue_halo(0);
  issue_w(0);
  issue_w(1);
#pragma unroll 1
  for (int s = 0; s < S; s++) {
    int ky, kx;
    step_tap(s, ky, kx);
    if (s == 0 || s == SA || s == S - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < S) issue_w(s + 2);
    const int r = 2 * dy + ky;                         // input row inside the tile
    const int xp = chalf * 20 + dx + (kx >> 1);        // + mi*4 per fragment (keeps (xp>>1)&1 of the lane)
    const int g = ((r >> 1) & 3) | (((xp >> 1) & 1) << 2);
    const unsigned char *xs = smem + (r * PW + xp) * 128 + (((((s & 1) << 2) | kg) ^ g) << 4);
    const unsigned char *ws = smem + wfo + (s % NWST) * WST;
    i4 xf[MI], wf[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) wf[ni] = *reinterpret_cast<const i4 *>(ws + ni * 16 * 64);
#pragma unroll
    for (int mi = 0; mi < MI; mi++) xf[mi] = *reinterpret_cast<const i4 *>(xs + mi * 512);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (s == SA - 1) {  // last read of plane 0: stage plane 1 under this step's MFMAs
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      issue_halo(1);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int mi = 0; mi < MI; mi++)
        acc[ni][mi] = mfma32<DT>(wf[ni], xf[mi], acc[ni][mi]);
  }
  conv_epilogue_px<MI, NI, DT, ODT, 2>(p, acc, wn * 64, lane, [&](int mi, int &oimg, int &oh, int &ow) {  // 2: no residual in this layer
    oimg = img;
    oh = oy0 + dy;
    ow = chalf * 20 + mi * 4 + dx;
    return true;
  });
}

// -------------------------------------------------------------------------------------------------
// gemm_k32_kernel: the Linear layers (QKV, out_proj, FFN; K = 512, 100 800 rows at N = 252).  With only 8 64-wide
// K-steps a 256x256 tile that owns its CU spends as long in its prologue (two stages of loads with nothing to overlap)
// and in its 128 KB store burst as in the K loop (~720 TFLOP/s).  Here: 128 rows x 256 channels, 4 waves = 2 (64 rows)
// x 2 (128 channels), 128 accumulators, 32-wide K-steps of 24 KB through a 3-stage LDS-DMA ring (72 KB), TWO
// workgroups per CU so one's prologue / epilogue runs under the other's MFMAs.  LDS rows are 64 bytes; the 16-byte
// slot of (row, chunk) is chunk ^ f((row>>2)&3), f = {0,2,3,1} (conflict-free ds_read_b128 fragments).
// -------------------------------------------------------------------------------------------------
template <int ABL, int DT, bool LSTORE = false, bool WPACK = false>  // timing ablations: 1 = no MFMAs, 2 = no loads after the first stage, 4 = no stores; LSTORE: epilogue through LDS; WPACK: weights from the stage-order copy (ConvParams::wpack)
__global__ __launch_bounds__(256, 2) void gemm_k32_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = 128, BN = 256;
  constexpr int XST = BM * 64, WST = BN * 64, STAGE = XST + WST, NST = 3;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int n_tiles = p.Cout / BN;
  int logical;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, within = b >> 3, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int mt = logical / n_tiles, nt = logical - mt * n_tiles;
  const int m0 = p.m_begin + mt * BM, n0 = nt * BN;
  const int S = p.krow_b >> 6;

  const unsigned char *in_b = p.in;
  const unsigned char *w_b = p.w;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

  // staging: a wave-instruction moves 16 rows x 64 B; lane -> (row = lane>>2, slot = lane&3), source chunk swizzled
  const int prow = lane >> 2;
  const int gch = (lane & 3) ^ ((0x78 >> (((prow >> 2) & 3) * 2)) & 3);
  unsigned xoff[2], woff[4];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int m = min(m0 + (wave * 2 + i) * 16 + prow, p.M - 1);  // rows past M re-read the last row (never stored)
    xoff[i] = (unsigned)(m * p.krow_b + gch * 16);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) woff[i] = (unsigned)((n0 + (wave * 4 + i) * 16 + prow) * p.krow_b + gch * 16);
  // (WPACK) this wave's 4 KB of K-step 0; a K-step is 4 waves x 4 KB further
  const unsigned char *wpk = WPACK ? p.wpack + ((size_t)nt * S * 4 + wave) * 4096 + lane * 16 : nullptr;
  auto issue = [&](int st) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (st % NST) * STAGE);
    const unsigned char *xb = in_b + (size_t)st * 64, *wb = w_b + (size_t)st * 64;
#pragma unroll
    for (int i = 0; i < 2; i++) glds16_asm(xb + xoff[i], dst + (wave * 2 + i) * 1024);
    if constexpr (WPACK) {
      glds16x4_asm(wpk + (size_t)st * 16384, dst + XST + wave * 4096);
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) glds16_asm(wb + woff[i], dst + XST + (wave * 4 + i) * 1024);
    }
  };

  f4 acc[2][4][4];  // [64-channel block][16-channel tile][16-row tile]
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) acc[h][a][b] = (f4){0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, kg = lane >> 4;
  const int fslot = kg ^ ((0x78 >> ((li >> 2) * 2)) & 3);
  const int xfo = (wm * 64 + li) * 64 + fslot * 16;
  const int wfo = XST + (wn * 128 + li) * 64 + fslot * 16;

  issue(0);
  if (S > 1 && !(ABL & 2)) issue(1);
#pragma unroll 1
  for (int s = 0; s < S; s++) {
    if (s == S - 1 || (ABL & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < S && !(ABL & 2)) issue(s + 2);
    const unsigned char *sb = smem + ((ABL & 2) ? 0 : (s % NST)) * STAGE;
    i4 xf[4], wf[8];
#pragma unroll
    for (int mi = 0; mi < 4; mi++) xf[mi] = *reinterpret_cast<const i4 *>(sb + xfo + mi * 1024);
#pragma unroll
    for (int ni = 0; ni < 8; ni++) wf[ni] = *reinterpret_cast<const i4 *>(sb + wfo + ni * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ABL & 1) {
#pragma unroll
      for (int ni = 0; ni < 8; ni++) asm volatile("" ::"v"(wf[ni]));
#pragma unroll
      for (int mi = 0; mi < 4; mi++) asm volatile("" ::"v"(xf[mi]));
      continue;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ni = 0; ni < 8; ni++)
#pragma unroll
      for (int mi = 0; mi < 4; mi++)
        acc[ni >> 2][ni & 3][mi] = mfma32<DT>(wf[ni], xf[mi], acc[ni >> 2][ni & 3][mi]);
    __builtin_amdgcn_s_setprio(0);
  }
  if constexpr (ABL == 0 && LSTORE) {
    // [r3] epilogue through LDS: the accumulator layout gives every store instruction 16 rows x 64 bytes -- 16 cache lines, half
    // of each -- and the vector memory path retires lines, not bytes (section 4.5 of DESIGN.md).  The wave's 64 x 128 tile is
    // written to its own 16 KB of the (now idle) ring as finished output rows and leaves as 4 rows x 256 contiguous bytes per
    // instruction.  Values are identical to conv_epilogue's (same operations in the same order).
    __syncthreads();                                   // every wave is done reading the ring
    unsigned char *tile = smem + wave * 16384;         // [64 rows][256 B], 16-byte chunk c stored at c ^ (row & 15)
    const int g = lane >> 4, li = lane & 15;
    const int mb = m0 + wm * 64, nb = n0 + wn * 128;
    float bv[2][2][8];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + nb + h * 64 + 8 * g + 32 * k), b1 = *reinterpret_cast<const float4 *>(p.bias + nb + h * 64 + 8 * g + 32 * k + 4);
        bv[h][k][0] = b0.x; bv[h][k][1] = b0.y; bv[h][k][2] = b0.z; bv[h][k][3] = b0.w; bv[h][k][4] = b1.x; bv[h][k][5] = b1.y; bv[h][k][6] = b1.z; bv[h][k][7] = b1.w;
      }
#pragma unroll
    for (int mi = 0; mi < 4; mi++) {
      const int row = mi * 16 + li, m = mb + row;
      const bool ok = m < p.M;
      i4 rv[2][2];
      if (p.res) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
          for (int k = 0; k < 2; k++)
            rv[h][k] = ok ? *reinterpret_cast<const i4 *>(p.res + ((size_t)m * p.res_ld + nb + h * 64 + 8 * g + 32 * k) * 2) : (i4){0, 0, 0, 0};
      }
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int k = 0; k < 2; k++) {
          typename ElemT<DT>::v8 ov;
#pragma unroll
          for (int e = 0; e < 8; e++) {
            float v = acc[h][e & 3][mi][2 * k + (e >> 2)] + bv[h][k][e];
            if (p.res) v += raw_elem<DT>(rv[h][k], e);
            if (p.relu) v = fmaxf(v, 0.f);
            ov[e] = (typename ElemT<DT>::t)v;
          }
          const int c = h * 8 + k * 4 + g;
          *reinterpret_cast<typename ElemT<DT>::v8 *>(tile + row * 256 + ((c ^ (row & 15)) << 4)) = ov;
        }
    }
    // (same-wave LDS write -> read: program order suffices, no barrier)
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const int row = it * 4 + (lane >> 4), c = lane & 15, m = mb + row;
      const i4 v = *reinterpret_cast<const i4 *>(tile + row * 256 + ((c ^ (row & 15)) << 4));
      if (m < p.M) *reinterpret_cast<i4 *>(p.out + ((size_t)m * p.out_ld + nb + c * 8) * 2) = v;
    }
    return;
  }
  conv_epilogue<4, 4, DT, DT, (ABL >> 2) & 1>(p, acc[0], m0 + wm * 64, n0 + wn * 128, lane);
  conv_epilogue<4, 4, DT, DT, (ABL >> 2) & 1>(p, acc[1], m0 + wm * 64, n0 + wn * 128 + 64, lane);
}

// -------------------------------------------------------------------------------------------------
// conv_deep_kernel: the rows the full 256x256 rounds of a long-K layer leave over (2.5 % of conv_512 at N = 252) run on
// an otherwise idle chip, one workgroup per CU walking all 72 K-steps: that chain is latency-bound, not bandwidth- or
// MFMA-bound.  So: small tiles (BM x 128, more CUs in use), the whole LDS as a deep LDS-DMA ring (6 x 24 KB stages at
// BM = 64: five K-steps in flight, counted vmcnt, one barrier per step) and the next step's fragments read under the
// current step's MFMAs (two register sets).  Same K order / accumulation order as every
// other schedule, so a row's value does not depend on which kernel computed it.
// -------------------------------------------------------------------------------------------------
template <int BM, int DT, int ODT = DT, bool POST = false>
__global__ __launch_bounds__(256, 2) void conv_deep_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BN = 128;
  constexpr int XB = BM * 128, WB = BN * 128, STAGE = XB + WB;
  constexpr int NST = BM == 64 ? 6 : 4, D = NST - 1;  // D stages in flight
  constexpr int XP = BM / 32, WP = 4, LPS = XP + WP;  // 1-KB pieces (8 rows x 128 B) per wave per stage
  constexpr int MI = BM / 32;                         // wave tile = BM/2 rows x 64 channels

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int n_tiles = p.Cout / BN;
  int logical;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, within = b >> 3, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int split = logical % p.ksplit;  // split-K (small problems): this workgroup walks K-steps [K0, KT) into an fp32 partial slab
  logical /= p.ksplit;
  const int mt = logical / n_tiles, nt = logical - mt * n_tiles;
  const int m0 = p.m_begin + mt * BM, n0 = nt * BN;
  const int ohw = p.OH * p.OW;
  const int IHp = p.H + 2 * p.ipad, IWp = p.W + 2 * p.ipad;
  const int K0 = split * p.kt_per, KT = min(p.krow_b >> 7, K0 + p.kt_per);

  const int srow = lane >> 3;
  const int g = (lane & 7) ^ srow;
  unsigned xoff[XP], woff[WP];
#pragma unroll
  for (int i = 0; i < XP; i++) {
    int m = min(m0 + (wave * XP + i) * 8 + srow, p.M - 1);  // rows past M re-read the last pixel (never stored)
    if (p.in_shared) m -= (m0 / p.grp_rows) * p.grp_rows;   // weight groups along M: every group reads the first group's rows
    int img = m / ohw;
    int rem = m - img * ohw;
    int oh = rem / p.OW, ow = rem - oh * p.OW;
    int ih0 = oh * p.stride - p.pad + p.ipad, iw0 = ow * p.stride - p.pad + p.ipad;
    xoff[i] = (unsigned)(((img * IHp + ih0) * IWp + iw0) * p.cin_b + g * 16);
  }
#pragma unroll
  for (int i = 0; i < WP; i++) woff[i] = (unsigned)((n0 + (wave * WP + i) * 8 + srow) * p.krow_b + g * 16);
  const unsigned char *in_b = p.in;
  const unsigned char *w_b = p.w + (p.grp_rows ? (size_t)(m0 / p.grp_rows) * p.grp_w_bytes : 0);
  // (ConvParams::wdeep, pack_stage_w128) this wave's 4 KB of every K-step as one run: one address + one M0 per stage -- in this
  // kernel (8 MFMAs per wave and K-step at BM = 64) the LDS-DMA issue IS the K-step time
  const bool packed = p.wdeep != nullptr;   // (wave-uniform: a scalar branch)
  const unsigned char *wpk = p.wdeep + (p.grp_rows ? (size_t)(m0 / p.grp_rows) * p.grp_w_bytes : 0) + ((size_t)nt * (p.krow_b >> 7) * 4 + wave) * 4096 + lane * 16;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
  auto issue = [&](int kt) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (kt % NST) * STAGE);
    const unsigned char *xb = in_b + p.koff[kt];
    const unsigned char *wb = w_b + (size_t)kt * 128;
#pragma unroll
    for (int i = 0; i < XP; i++) glds16_asm(xb + xoff[i], dst + (wave * XP + i) * 1024);
    if (packed) { glds16x4_asm(wpk + (size_t)kt * 16384, dst + XB + wave * 4096); return; }
#pragma unroll
    for (int i = 0; i < WP; i++) glds16_asm(wb + woff[i], dst + XB + (wave * WP + i) * 1024);
  };

  f4 acc[4][MI];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < MI; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fk = lane >> 4;
  int xfo[2], wfo[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    int slot = (ks * 4 + fk) ^ (lane & 7);
    xfo[ks] = (wm * (BM / 2) + frow) * 128 + slot * 16;
    wfo[ks] = XB + (wn * 64 + frow) * 128 + slot * 16;
  }

  // wait until stage `st` has landed: at most min(max_younger, KT-1-st) younger stages may still be in flight
  auto wait_stage = [&](int st, int max_younger) {
    const int younger = min(max_younger, KT - 1 - st);
    if (younger >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPS) : "memory");
    else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPS) : "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  struct Frags { i4 x[2][MI], w[2][4]; };
  auto read_frags = [&](int kt, Frags &f) {
    const unsigned char *sb = smem + (kt % NST) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
      for (int mi = 0; mi < MI; mi++) f.x[ks][mi] = *reinterpret_cast<const i4 *>(sb + xfo[ks] + mi * 16 * 128);
#pragma unroll
      for (int ni = 0; ni < 4; ni++) f.w[ks][ni] = *reinterpret_cast<const i4 *>(sb + wfo[ks] + ni * 16 * 128);
    }
  };
  // one K-step: the fragments of stage kt are already in `cur`; stage kt+1's are read into `nxt` under this step's MFMAs
  auto step = [&](int kt, const Frags &cur, Frags &nxt) {
    if (kt + 1 < KT) {
      wait_stage(kt + 1, D - 2);
      __builtin_amdgcn_s_barrier();  // every wave has finished reading stages <= kt (read one step ahead)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (kt + D < KT) issue(kt + D);  // reuses the buffer of stage kt-1
      read_frags(kt + 1, nxt);
    }
    mma_kstep<DT, 4, MI>(acc, cur.w, cur.x);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

#pragma unroll
  for (int s = 0; s < D; s++)
    if (K0 + s < KT) issue(K0 + s);
  Frags fa, fb;
  wait_stage(K0, D - 1);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(K0, fa);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int kt = K0;
#pragma unroll 1
  for (; kt + 1 < KT; kt += 2) {
    step(kt, fa, fb);
    step(kt + 1, fb, fa);
  }
  if (kt < KT) step(kt, fa, fb);
  if (p.ksplit > 1) {
    conv_store_partial<MI, 4>(p, acc, split, m0 + wm * (BM / 2), n0 + wn * 64, lane);
    return;
  }
  conv_epilogue<MI, 4, DT, ODT, POST ? 4 : 0>(p, acc, m0 + wm * (BM / 2), n0 + wn * 64, lane);
}

// split-K reduction + the conv epilogue: out = relu(sum_s partial[s] + bias + res); thread = (pixel, 8 channels).  2-byte networks
// only: the 8-bit layers never split K across workgroups (run_conv_dt: plan_splitk).
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvParams p) {
  const int nq = p.Cout / 8;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int rows = p.M - p.m_begin;
  if (i >= (size_t)rows * nq) return;
  const int mr = (int)(i / nq), n = (int)(i - (size_t)mr * nq) * 8;
  const int m = p.m_begin + mr;
  float v[8];
  {
    const f4 a0 = *reinterpret_cast<const f4 *>(p.partial + (size_t)mr * p.Cout + n), a1 = *reinterpret_cast<const f4 *>(p.partial + (size_t)mr * p.Cout + n + 4);
#pragma unroll
    for (int e = 0; e < 4; e++) { v[e] = a0[e]; v[4 + e] = a1[e]; }
  }
  for (int sp = 1; sp < p.ksplit; sp++) {  // component-wise: no packed-f32 VALU ops in this library
    const float *src = p.partial + ((size_t)sp * rows + mr) * p.Cout + n;
    const f4 b0 = *reinterpret_cast<const f4 *>(src), b1 = *reinterpret_cast<const f4 *>(src + 4);
#pragma unroll
    for (int e = 0; e < 4; e++) { v[e] += b0[e]; v[4 + e] += b1[e]; }
  }
  const int grp = p.grp_rows ? m / p.grp_rows : 0;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    v[e] += p.bias[grp * p.Cout + n + e];
  }
  const int ohw = p.OH * p.OW;
  int img = m / ohw;
  int rem = m - img * ohw;
  int oh = rem / p.OW, ow = rem - oh * p.OW;
  const int OHp = p.OH + 2 * p.opad, OWp = p.OW + 2 * p.opad;
  const int RHp = p.OH + 2 * p.rpad, RWp = p.OW + 2 * p.rpad;
  if (p.res) {
    size_t rpix = ((size_t)(img - (p.res_shared ? grp * p.grp_rows : 0)) * RHp + oh + p.rpad) * RWp + ow + p.rpad;
    float r[8];
    decode8(load8_raw(p.res + (rpix * p.res_ld + n) * elem_bytes(p.res_dt), p.res_dt), p.res_dt, r);
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] += r[e];
  }
#pragma unroll
  for (int e = 0; e < 8; e++) {
    if (p.relu) v[e] = fmaxf(v[e], 0.f);
  }
  if (p.post) {  // (2-byte output types only)
    float pe[8];
    decode8(load8_raw(p.post + ((size_t)rem * p.Cout + n) * 2, p.out_dt), p.out_dt, pe);
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (p.out_dt == DT_BF16 ? (float)(__bf16)v[e] : (float)(_Float16)v[e]) + pe[e];
  }
  int choff = 0, oimg = img;
  if (p.split_imgs > 0 && img >= p.split_imgs) { oimg = img - p.split_imgs; choff = p.Cout; }
  size_t opix = ((size_t)oimg * OHp + oh + p.opad) * OWp + ow + p.opad;
  store8(p.out + (opix * p.out_ld + choff + n) * elem_bytes(p.out_dt), p.out_dt, v);
}

// =================================================================================================
// attention: out[b,t,h*128+d] = softmax_k(q.k/sqrt(128)) v,  qkv = [B,T,1536] (q|k|v, heads contiguous inside each)
// =================================================================================================

// (a 64-key-block variant of this kernel measured 6 % slower inside Register and was dropped)
// ATT_QROWS query rows per workgroup (64 = 4 waves, one per SIMD; 80-row / 5-wave tiles cover 400 tokens exactly but
// measured 12 % slower: two waves of a workgroup share a SIMD).  The 1-D grid is remapped so the query
// tiles of one (image, head) run on the SAME XCD and share its L2 copy of K/V (a (qt,h,b) grid spread them over all 8
// XCDs: rocprofv3 FETCH_SIZE showed 1.16 GB fetched per launch for 0.31 GB of QKV).
template <int ATT_QROWS, bool REMAP, bool PERM, int DT>
__global__ __launch_bounds__(ATT_QROWS * 4, 2) void attention_kernel(const typename ElemT<DT>::t *__restrict__ qkv, typename ElemT<DT>::t *__restrict__ out, int T, int nq,
                                                                 int tstride /* rows between the first tokens of consecutive sequences */) {
  constexpr int KS = 136;  // K tile row stride (halfs): 128 + 8 pad
  constexpr int VS = 40;   // V^T tile row stride (halfs): 32 keys + 8 pad
  using E = typename ElemT<DT>::t;
  using E8 = typename ElemT<DT>::v8;
  using E4 = typename ElemT<DT>::v4;
  __shared__ __attribute__((aligned(16))) E Ks[32 * KS];
  __shared__ __attribute__((aligned(16))) E Vt[HDIM * VS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int logical;
  {
    const int nblk = gridDim.x, bi = blockIdx.x;
    const int xcd = bi & 7, within = bi >> 3, q = nblk >> 3, r = nblk & 7;
    logical = REMAP ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within : bi;
  }
  const int qt = logical % nq, h = (logical / nq) % HEADS, b = logical / (nq * HEADS);
  const int g = lane >> 4, li = lane & 15;
  const size_t rowstride = 3 * EMBED;
  const E *base = qkv + (size_t)b * tstride * rowstride;
  const int q_row = qt * ATT_QROWS + wave * 16 + li;
  const int q_ld = min(q_row, T - 1);
  E8 qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ds++)
    qf[ds] = *reinterpret_cast<const E8 *>(base + (size_t)q_ld * rowstride + h * HDIM + ds * 32 + g * 8);

  f4 o[8];
#pragma unroll
  for (int dt = 0; dt < 8; dt++) o[dt] = (f4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float sl2e = 0.08838834764831845f * 1.4426950408889634f;  // 1/sqrt(128) * log2(e)

  const int nkb = (T + 31) / 32;
  // staging roles.  K: thread -> (key = idx>>4, 16-B chunk = idx&15): coalesced 256-B rows.  V: thread -> (key = idx&31,
  // chunk = idx>>5) so the 2-byte transposed LDS writes of one instruction cover 32 consecutive keys of one d row
  // (bank-conflict free; the previous key-major mapping was a 16-way conflict on every ds_write_b16).
  // (the first 4 waves stage; wave 4 only computes)
  E8 kreg[2], vreg[2];
  auto load_tile = [&](int kb) {
    if (ATT_QROWS > 64 && tid >= 256) return;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      int idx = tid + j * 256;
      int krow = min(kb * 32 + (idx >> 4), T - 1);
      kreg[j] = *reinterpret_cast<const E8 *>(base + (size_t)krow * rowstride + EMBED + h * HDIM + (idx & 15) * 8);
      int vrow = min(kb * 32 + (idx & 31), T - 1);
      vreg[j] = *reinterpret_cast<const E8 *>(base + (size_t)vrow * rowstride + 2 * EMBED + h * HDIM + (idx >> 5) * 8);
    }
  };
  load_tile(0);
  for (int kb = 0; kb < nkb; kb++) {
    if (ATT_QROWS <= 64 || tid < 256) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        int idx = tid + j * 256;
        *reinterpret_cast<E8 *>(&Ks[(idx >> 4) * KS + (idx & 15) * 8]) = kreg[j];
        int key = idx & 31, chunk = idx >> 5;
        // V^T row e*16 + chunk holds d = chunk*8 + e, so MFMA column li of tile dt is d = li*8 + dt and a lane ends up
        // owning 8 consecutive d (one 16-byte output store per query row)
#pragma unroll
        for (int e = 0; e < 8; e++) Vt[(PERM ? e * 16 + chunk : chunk * 8 + e) * VS + key] = vreg[j][e];
      }
    }
    __syncthreads();
    if (kb + 1 < nkb) load_tile(kb + 1);  // next tile's global loads fly under this tile's MFMAs
    // S^T tiles: st[kt][r] = S[key = kt*16 + g*4 + r][q = li]
    f4 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; kt++) {
      st[kt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < 4; ds++) {
        E8 kf = *reinterpret_cast<const E8 *>(&Ks[(kt * 16 + li) * KS + ds * 32 + g * 8]);
        st[kt] = mfma32<DT>(__builtin_bit_cast(i4, kf), __builtin_bit_cast(i4, qf[ds]), st[kt]);
      }
    }
    // softmax in base 2 on the RAW scores: p = exp2(s*c - m*c), c = scale*log2(e) -- one fma + one v_exp per score; the
    // running maximum m is kept unscaled.  Keys past T exist only in the last block (wave-uniform branch).
    if (kb == nkb - 1 && (T & 31)) {
#pragma unroll
      for (int kt = 0; kt < 2; kt++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (kb * 32 + kt * 16 + g * 4 + r >= T) st[kt][r] = -INFINITY;
    }
    float mx = fmaxf(fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3])),
                     fmaxf(fmaxf(st[1][0], st[1][1]), fmaxf(st[1][2], st[1][3])));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float mc = m_new * sl2e;
    const float alpha = __builtin_amdgcn_exp2f(m_run * sl2e - mc);  // m_run = -inf on the first block -> 0
    float psum = 0.f;
    E8 pf;
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r], sl2e, -mc));
        psum += pv;
        pf[kt * 4 + r] = (E)pv;
      }
    psum += __shfl_xor(psum, 16);
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // rescale O rows (row q' = g*4 + r lives in lanes with li == q') -- only when some row's running maximum moved:
    // after the first key blocks alpha is exactly 1 for every row most of the time, and the 32 multiplies + 4 shuffles
    // per block made this kernel VALU-bound (x * 1.0f is exact, so skipping it changes nothing)
    const bool rescale = __any(alpha != 1.0f);
    if (rescale) {
      float ar[4];
#pragma unroll
      for (int r = 0; r < 4; r++) ar[r] = __shfl(alpha, g * 4 + r);
#pragma unroll
      for (int dt = 0; dt < 8; dt++)
#pragma unroll
        for (int r = 0; r < 4; r++) o[dt][r] *= ar[r];
    }
#pragma unroll
    for (int dt = 0; dt < 8; dt++) {
      // V^T fragment: col li of tile dt is d = li*8 + dt; k-slots 0..3 -> keys g*4.., 4..7 -> keys 16+g*4..
      E4 v0 = *reinterpret_cast<const E4 *>(&Vt[(dt * 16 + li) * VS + g * 4]);
      E4 v1 = *reinterpret_cast<const E4 *>(&Vt[(dt * 16 + li) * VS + 16 + g * 4]);
      E8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      o[dt] = mfma32<DT>(__builtin_bit_cast(i4, pf), __builtin_bit_cast(i4, vf), o[dt]);
    }
    __syncthreads();
  }
  float lr[4];
#pragma unroll
  for (int r = 0; r < 4; r++) lr[r] = 1.0f / __shfl(l_run, g * 4 + r);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    int row = qt * ATT_QROWS + wave * 16 + g * 4 + r;
    if (row >= T) continue;
    if (PERM) {
      E8 ov;
#pragma unroll
      for (int dt = 0; dt < 8; dt++) ov[dt] = (E)(o[dt][r] * lr[r]);
      *reinterpret_cast<E8 *>(out + ((size_t)b * tstride + row) * EMBED + h * HDIM + li * 8) = ov;
    } else {
      E *dst = out + ((size_t)b * tstride + row) * EMBED + h * HDIM + li;
#pragma unroll
      for (int dt = 0; dt < 8; dt++) dst[dt * 16] = (E)(o[dt][r] * lr[r]);
    }
  }
}

// -------------------------------------------------------------------------------------------------
// attention32_kernel [r2]: the kernel above is LDS-bandwidth-bound -- a wave owns ONE 16-row query tile, so every 32-key
// block costs it the whole K and V tile (16 KB of fragment reads) for 16 MFMAs, 2.4x what the LDS delivers at the MFMA
// rate, plus sixteen 2-byte transposing LDS writes per thread.  Here:
//   * a wave owns 32 query rows (two 16-row tiles): every K / V fragment feeds two MFMAs (LDS bytes per MFMA halved),
//     a workgroup = 4 waves = 128 query rows (4 workgroups per 400-token sequence and head instead of 7: K/V staged 4x);
//   * O is accumulated TRANSPOSED, O^T[d][q] = V^T P^T: a lane owns one query in S^T and in O^T alike, so the online-softmax
//     rescale is a lane-local multiply (no shuffles) and P^T feeds the MFMA's B operand straight from the softmax registers;
//   * V stays row-major in LDS (16-byte staging writes) and the V^T operand comes from ds_read_b64_tr_b16 (hardware
//     transpose read): [32 keys][16 d] sub-tiles of 1 KB (+32 B so the staging writes of one instruction spread over all
//     banks), a 16-lane group reads one [4 keys][16 d] block = 128 contiguous bytes;
//   * K rows (256 B) are XOR-swizzled by 16-byte slot (slot ^= key & 15): conflict-free staging writes and fragment reads;
//   * two LDS buffers, ONE barrier per key block: tile kb+1 is written (from registers loaded three iterations earlier [r4]) while
//     tile kb is consumed, the global loads of tiles kb+2 / kb+3 are in flight (two register sets);
//   * all K fragments of a block are requested at once, all V^T fragments right after the QK MFMAs so their latency runs under the
//     softmax (asm reads + one explicit wait: hipcc otherwise sinks each read to its first use); the cross-row max / sum use
//     v_permlane16/32_swap instead of four LDS round trips (ds_bpermute) per softmax;
//   * O leaves through LDS as whole 256-byte rows.
// -------------------------------------------------------------------------------------------------
typedef short s4 __attribute__((ext_vector_type(4)));
// reductions over the four 16-lane rows of a wave (lanes li, li+16, li+32, li+48) on the VALU: v_permlane16_swap exchanges rows
// 0<->1 and 2<->3, v_permlane32_swap the two halves (ds_bpermute shuffles put four LDS round trips per softmax on the
// critical path of every key block)
__device__ __forceinline__ float vmax_f32(float a, float b) {  // (fmaxf would be preceded by two canonicalising v_max)
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float rows_max(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float m = vmax_f32(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return vmax_f32(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
template <bool REMAP, int DT, int ABL = 0, int NW = 4>  // NW waves = NW * 32 query rows per workgroup; ABL (timing ablations, wrong results): 1 no staging after the first tile, 2 no softmax, 4 no PV, 8 no QK
__global__ __launch_bounds__(NW * 64, 2) void attention32_kernel(const typename ElemT<DT>::t *__restrict__ qkv, typename ElemT<DT>::t *__restrict__ out, int T, int nq,
                                                             int tstride /* rows between the first tokens of consecutive sequences */,
                                                             int qkv_ld /* elements between consecutive qkv rows (>= 1536) */) {
  using E = typename ElemT<DT>::t;
  using E8 = typename ElemT<DT>::v8;
  constexpr int KB = 32 * 256;   // K tile: 32 keys x 128 d
  constexpr int VSUB = 1056;     // V sub-tile [32 keys][16 d] + 32 B
  constexpr int BUF = KB + 8 * VSUB;
  constexpr int NJ = 512 / (NW * 64);  // 16-byte chunks of K (and of V) a thread stages per tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF > NW * 8192 ? 2 * BUF : NW * 8192];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int logical;
  {
    const int nblk = gridDim.x, bi = blockIdx.x;
    const int xcd = bi & 7, within = bi >> 3, q = nblk >> 3, r = nblk & 7;
    logical = REMAP ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within : bi;
  }
  const int qt = logical % nq, h = (logical / nq) % HEADS, b = logical / (nq * HEADS);
  const int g = lane >> 4, li = lane & 15;
  const size_t rowstride = (size_t)qkv_ld;
  const E *base = qkv + (size_t)b * tstride * rowstride + h * HDIM;
  const int q0 = qt * (NW * 32) + wave * 32;
  const bool active = q0 < T;  // (wave-uniform) waves past the sequence only help staging

  i4 qf[2][4];
#pragma unroll
  for (int qi = 0; qi < 2; qi++) {
    const int q_ld = min(q0 + qi * 16 + li, T - 1);
#pragma unroll
    for (int ds = 0; ds < 4; ds++) qf[qi][ds] = *reinterpret_cast<const i4 *>(base + (size_t)q_ld * rowstride + ds * 32 + g * 8);
  }
  f4 o[8][2];
#pragma unroll
  for (int dt = 0; dt < 8; dt++)
#pragma unroll
    for (int qi = 0; qi < 2; qi++) o[dt][qi] = (f4){0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  const float sl2e = 0.08838834764831845f * 1.4426950408889634f;  // 1/sqrt(128) * log2(e)

  // staging: thread -> (key = idx >> 4, 16-byte chunk = idx & 15), idx = tid + 256 j: coalesced 256-byte rows
  const int nkb = (T + 31) / 32;
  // three register sets: tile kb+1 (written to LDS during block kb), tiles kb+2 and kb+3 in flight -- a tile's loads have three blocks
  // to arrive (with two sets the staging-only pipeline and the compute-only pipeline added up: tools/bench_attn.py V=17,30)
  i4 kreg[3][NJ], vreg[3][NJ];
  auto load_tile = [&](int kb, i4 (&kr)[NJ], i4 (&vr)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int idx = tid + j * (NW * 64);
      const int row = min(kb * 32 + (idx >> 4), T - 1);
      const E *src = base + (size_t)row * rowstride + (idx & 15) * 8;
      // asm loads: the compiler's own vmcnt bookkeeping collapses to vmcnt(0) at the joins of this loop (every store_tile then waited
      // for the tile requested one block earlier); the waits are counted by hand in wait_tile
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(kr[j]) : "v"(src), "n"(EMBED * (int)sizeof(E)));
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(vr[j]) : "v"(src), "n"(2 * EMBED * (int)sizeof(E)));
    }
  };
  // the set's 2 NJ loads have landed when at most `newer` younger loads are outstanding (vmcnt retires in order)
  auto wait_tile = [&](int newer_tiles, i4 (&kr)[NJ], i4 (&vr)[NJ]) {
    if (newer_tiles >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NJ) : "memory");
    else if (newer_tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < NJ; j++) asm volatile("" : "+v"(kr[j]), "+v"(vr[j]));
  };
  auto store_tile = [&](int buf, const i4 (&kr)[NJ], const i4 (&vr)[NJ]) {
    unsigned char *sb = smem + buf * BUF;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int idx = tid + j * (NW * 64);
      const int key = idx >> 4, chunk = idx & 15;
      *reinterpret_cast<i4 *>(sb + key * 256 + ((chunk ^ (key & 15)) << 4)) = kr[j];
      *reinterpret_cast<i4 *>(sb + KB + (chunk >> 1) * VSUB + key * 32 + (chunk & 1) * 16) = vr[j];
    }
  };
  load_tile(0, kreg[0], vreg[0]);
  wait_tile(0, kreg[0], vreg[0]);
#pragma unroll
  for (int qi = 0; qi < 2; qi++)  // a use of Q the compiler sees: its own wait for these loads happens here, not in front of the MFMAs of every block
#pragma unroll
    for (int ds = 0; ds < 4; ds++) asm volatile("" : "+v"(qf[qi][ds]));
  store_tile(0, kreg[0], vreg[0]);
  if (!(ABL & 1)) {  // (loads nobody consumes would land in registers the compiler has already handed out again)
    if (nkb > 1) load_tile(1, kreg[1], vreg[1]);   // tile t travels in set t % 3
    if (nkb > 2) load_tile(2, kreg[2], vreg[2]);
    if (nkb > 3) load_tile(3, kreg[0], vreg[0]);
  }
  __syncthreads();

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
  unsigned koff[4];
#pragma unroll
  for (int ds = 0; ds < 4; ds++) koff[ds] = (unsigned)(li * 256 + (((ds * 4 + g) ^ li) << 4));
  const unsigned voff = (unsigned)((g * 4 + (li >> 2)) * 32 + (li & 3) * 8);

  auto block = [&](const int kb, i4 (&kr)[NJ], i4 (&vr)[NJ]) {   // kr / vr hold tile kb+1 on entry, tile kb+4 on exit
    f4 st[2][2];  // [query tile][key tile]: st[qi][kt][r] = S[key = kt*16 + g*4 + r][q = qi*16 + li]
    if (ABL & 8) {
#pragma unroll
      for (int qi = 0; qi < 2; qi++)
#pragma unroll
        for (int kt = 0; kt < 2; kt++) st[qi][kt] = __builtin_bit_cast(f4, qf[qi][kt]);
    } else if (active) {
#pragma unroll
      for (int qi = 0; qi < 2; qi++)
#pragma unroll
        for (int kt = 0; kt < 2; kt++) st[qi][kt] = (f4){0.f, 0.f, 0.f, 0.f};
      // all K fragments of the block in one go (one LDS latency, then 16 MFMAs back to back).  The reads are asm statements: hipcc
      // otherwise sinks every read to its first use and waits after each small group
      i4 kf[2][4];
      const unsigned kbase = lds0 + (unsigned)((kb & 1) * BUF);
#pragma unroll
      for (int ds = 0; ds < 4; ds++) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[0][ds]) : "v"(kbase + koff[ds]));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(kf[1][ds]) : "v"(kbase + koff[ds]));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kt = 0; kt < 2; kt++)
#pragma unroll
        for (int ds = 0; ds < 4; ds++)
#pragma unroll
          for (int qi = 0; qi < 2; qi++) st[qi][kt] = mfma32<DT>(kf[kt][ds], qf[qi][ds], st[qi][kt]);
    }
    // V^T fragments of this block: requested now, consumed after the softmax (their latency runs under it)
    i2 vlo[8], vhi[8];
    if (active && !(ABL & 4)) {
      const unsigned vbase = lds0 + (unsigned)((kb & 1) * BUF + KB) + voff;
#pragma unroll
      for (int dt = 0; dt < 8; dt++) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vlo[dt]) : "v"(vbase), "n"(dt * VSUB));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vhi[dt]) : "v"(vbase), "n"(dt * VSUB + 512));
      }
    }
    // the next tile goes into the other buffer (its last readers passed the barrier that ended the previous iteration)
    if (kb + 1 < nkb && !(ABL & 1)) {
      if (ABL & 32) {
#pragma unroll
        for (int j = 0; j < NJ; j++) asm volatile("" ::"v"(kr[j]), "v"(vr[j]));
      } else {
        wait_tile(ABL & 16 ? 0 : min(2, nkb - 2 - kb), kr, vr);  // tiles kb+2, kb+3 (where they exist) were requested after this one
        store_tile((kb + 1) & 1, kr, vr);
      }
      if (kb + 4 < nkb && !(ABL & 16)) load_tile(kb + 4, kr, vr);
    }
    if (active) {
      i4 pf[2];
      if (ABL & 2) {
#pragma unroll
        for (int qi = 0; qi < 2; qi++) pf[qi] = __builtin_bit_cast(i4, st[qi][0] + st[qi][1]);
      } else
#pragma unroll
      for (int qi = 0; qi < 2; qi++) {
        // softmax in base 2 on the RAW scores: p = exp2(s*c - m*c), c = scale*log2(e); keys past T exist only in the last block
        if (kb == nkb - 1 && (T & 31)) {
#pragma unroll
          for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int r = 0; r < 4; r++)
              if (kb * 32 + kt * 16 + g * 4 + r >= T) st[qi][kt][r] = -INFINITY;
        }
        float mx = vmax_f32(vmax_f32(vmax_f32(st[qi][0][0], st[qi][0][1]), vmax_f32(st[qi][0][2], st[qi][0][3])),
                            vmax_f32(vmax_f32(st[qi][1][0], st[qi][1][1]), vmax_f32(st[qi][1][2], st[qi][1][3])));
        mx = rows_max(mx);
        // (ABL 64, test build: a LAZY reference -- it only moves when the block maximum exceeds it by more than 2^8, which skips most of
        // the 64-multiply rescales of O^T: 213 -> 204 us per launch.  Not shipped: mathematically the same, but a row whose threshold
        // decision flips under a 1e-3 perturbation of its inputs gets a different f16 rounding of ALL its P values, and the score-net's
        // pooled feature then moves by up to 1.3x the between-hypothesis spread between a shard of 32 and the full batch
        // (tools/ab_shard_att.py); with the exact running maximum the rounding pattern is shared and the two agree to 0.12x.)
        const float m_new = (ABL & 64) ? ((mx * sl2e > m_run[qi] * sl2e + 8.0f) ? mx : m_run[qi]) : vmax_f32(m_run[qi], mx);
        const float mc = m_new * sl2e;
        const float alpha = __builtin_amdgcn_exp2f(m_run[qi] * sl2e - mc);  // m_run = -inf on the first block -> 0
        float psum = 0.f;
        E8 pv8;
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qi][kt][r], sl2e, -mc));
            psum += pv;
            pv8[kt * 4 + r] = (E)pv;
          }
        pf[qi] = __builtin_bit_cast(i4, pv8);
        psum = rows_sum(psum);
        l_run[qi] = l_run[qi] * alpha + psum;
        m_run[qi] = m_new;
        // O^T columns of this lane all belong to query li: the rescale is lane-local; x * 1.0f is exact, so it is skipped
        // when no row's running maximum moved (most blocks after the first few)
        if (__any(alpha != 1.0f)) {
#pragma unroll
          for (int dt = 0; dt < 8; dt++)
#pragma unroll
            for (int r = 0; r < 4; r++) o[dt][qi][r] *= alpha;
        }
      }
      // O^T[d][q] += V^T[d][key] P^T[key][q]; k-slot j of lane group g is key (j>>2)*16 + g*4 + (j&3) in both operands
      if (ABL & 4) {
#pragma unroll
        for (int qi = 0; qi < 2; qi++) o[0][qi] += __builtin_bit_cast(f4, pf[qi]);
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
        for (int dt = 0; dt < 8; dt++) {
          const i4 vf = (i4){vlo[dt][0], vlo[dt][1], vhi[dt][0], vhi[dt][1]};
#pragma unroll
          for (int qi = 0; qi < 2; qi++) o[dt][qi] = mfma32<DT>(vf, pf[qi], o[dt][qi]);
        }
        }
    }
    __syncthreads();
  };
  for (int kb = 0; kb < nkb; kb += 3) {
    block(kb, kreg[1], vreg[1]);
    if (kb + 1 < nkb) block(kb + 1, kreg[2], vreg[2]);
    if (kb + 2 < nkb) block(kb + 2, kreg[0], vreg[0]);
  }
  if (!active) return;
  // O tile of the wave through LDS (both buffers are free now; 8 KB per wave) -> whole 256-byte rows
  unsigned char *ob = smem + wave * 8192;
#pragma unroll
  for (int qi = 0; qi < 2; qi++) {
    const float inv = 1.0f / l_run[qi];
    const int row = qi * 16 + li;
#pragma unroll
    for (int dt = 0; dt < 8; dt++) {
      typename ElemT<DT>::v4 ov;
#pragma unroll
      for (int r = 0; r < 4; r++) ov[r] = (E)(o[dt][qi][r] * inv);
      // d = dt*16 + g*4 + r: 16-byte slot dt*2 + (g>>1), swizzled by the row
      *reinterpret_cast<typename ElemT<DT>::v4 *>(ob + row * 256 + (((dt * 2 + (g >> 1)) ^ (row & 15)) << 4) + (g & 1) * 8) = ov;
    }
  }
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int id = it * 64 + lane, rr = id >> 4, c = id & 15;
    const int row = q0 + rr;
    const i4 v = *reinterpret_cast<const i4 *>(ob + rr * 256 + ((c ^ (rr & 15)) << 4));
    if (row < T) *reinterpret_cast<i4 *>(out + ((size_t)b * tstride + row) * EMBED + h * HDIM + c * 8) = v;
  }
}

// -------------------------------------------------------------------------------------------------
// attention32_skv_kernel [r2]: the same arithmetic for SMALL grids (Track: 2 sequences; the score-net's cross attention: one),
// where a workgroup of attention32_kernel is a latency chain of T/32 key blocks on a mostly idle chip.  One workgroup per 32
// query rows; its four waves split the KEY blocks (wave w takes blocks w, w+4, ...) and run independently -- private LDS
// double buffers filled by LDS-DMA (K rows swizzled on the source side, V in the transpose-read sub-tile layout), no workgroup
// barrier in the loop -- then merge their partial (max, sum, O) through LDS: chain length T/128 blocks, 3.25x more workgroups.
// -------------------------------------------------------------------------------------------------
template <bool REMAP, int DT>
__global__ __launch_bounds__(256, 2) void attention32_skv_kernel(const typename ElemT<DT>::t *__restrict__ qkv, typename ElemT<DT>::t *__restrict__ out, int T, int nq,
                                                                 int tstride, int qkv_ld) {
  using E = typename ElemT<DT>::t;
  using E8 = typename ElemT<DT>::v8;
  constexpr int KB = 32 * 256, VSUB = 1056, BUF = KB + 8 * VSUB, WREG = 2 * BUF;  // per wave: two tile buffers (33 280 B)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int logical;
  {
    const int nblk = gridDim.x, bi = blockIdx.x;
    const int xcd = bi & 7, within = bi >> 3, q = nblk >> 3, r = nblk & 7;
    logical = REMAP ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within : bi;
  }
  const int qt = logical % nq, h = (logical / nq) % HEADS, b = logical / (nq * HEADS);
  const int g = lane >> 4, li = lane & 15;
  const size_t rowstride = (size_t)qkv_ld;
  const E *base = qkv + (size_t)b * tstride * rowstride + h * HDIM;
  const int q0 = qt * 32;
  unsigned char *wbuf = smem_dyn + wave * WREG;
  const unsigned wlds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)wbuf;

  i4 qf[2][4];
#pragma unroll
  for (int qi = 0; qi < 2; qi++) {
    const int q_ld = min(q0 + qi * 16 + li, T - 1);
#pragma unroll
    for (int ds = 0; ds < 4; ds++) qf[qi][ds] = *reinterpret_cast<const i4 *>(base + (size_t)q_ld * rowstride + ds * 32 + g * 8);
  }
  f4 o[8][2];
#pragma unroll
  for (int dt = 0; dt < 8; dt++)
#pragma unroll
    for (int qi = 0; qi < 2; qi++) o[dt][qi] = (f4){0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  const float sl2e = 0.08838834764831845f * 1.4426950408889634f;

  const int nkb = (T + 31) / 32;
  // one wave stages a whole tile: 8 LDS-DMA instructions for K (4 keys x 256 B each; the lane at LDS slot s of key k fetches
  // chunk s ^ (k & 15)) and 8 for V (one [32 keys][16 d] sub-tile each: lane -> key lane>>1, 8-d half lane&1)
  auto issue_tile = [&](int kb, int bufi) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(wlds + bufi * BUF);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int key = i * 4 + (lane >> 4);
      const int row = min(kb * 32 + key, T - 1);
      const E *src = base + (size_t)row * rowstride + EMBED + (((lane & 15) ^ (key & 15)) * 8);
      glds16_asm(src, dst + i * 1024);
    }
#pragma unroll
    for (int dt = 0; dt < 8; dt++) {
      const int row = min(kb * 32 + (lane >> 1), T - 1);
      const E *src = base + (size_t)row * rowstride + 2 * EMBED + dt * 16 + (lane & 1) * 8;
      glds16_asm(src, dst + KB + dt * VSUB);
    }
  };
  unsigned koff[4];
#pragma unroll
  for (int ds = 0; ds < 4; ds++) koff[ds] = (unsigned)(li * 256 + (((ds * 4 + g) ^ li) << 4));
  const unsigned voff = (unsigned)((g * 4 + (li >> 2)) * 32 + (li & 3) * 8);

  int bufi = 0;
  if (wave < nkb) issue_tile(wave, 0);
  for (int kb = wave; kb < nkb; kb += 4, bufi ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's tile kb has landed (only this wave reads it)
    if (kb + 4 < nkb) issue_tile(kb + 4, bufi ^ 1);    // (its last reads finished an iteration ago: lgkmcnt(0) below)
    f4 st[2][2];
#pragma unroll
    for (int qi = 0; qi < 2; qi++)
#pragma unroll
      for (int kt = 0; kt < 2; kt++) st[qi][kt] = (f4){0.f, 0.f, 0.f, 0.f};
    i4 kf[2][4];
    const unsigned kbase = wlds + (unsigned)(bufi * BUF);
#pragma unroll
    for (int ds = 0; ds < 4; ds++) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(kf[0][ds]) : "v"(kbase + koff[ds]));
      asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(kf[1][ds]) : "v"(kbase + koff[ds]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
      for (int ds = 0; ds < 4; ds++)
#pragma unroll
        for (int qi = 0; qi < 2; qi++) st[qi][kt] = mfma32<DT>(kf[kt][ds], qf[qi][ds], st[qi][kt]);
    i2 vlo[8], vhi[8];
    {
      const unsigned vbase = kbase + KB + voff;
#pragma unroll
      for (int dt = 0; dt < 8; dt++) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vlo[dt]) : "v"(vbase), "n"(dt * VSUB));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vhi[dt]) : "v"(vbase), "n"(dt * VSUB + 512));
      }
    }
    i4 pf[2];
#pragma unroll
    for (int qi = 0; qi < 2; qi++) {
      if (kb == nkb - 1 && (T & 31)) {
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (kb * 32 + kt * 16 + g * 4 + r >= T) st[qi][kt][r] = -INFINITY;
      }
      float mx = vmax_f32(vmax_f32(vmax_f32(st[qi][0][0], st[qi][0][1]), vmax_f32(st[qi][0][2], st[qi][0][3])),
                          vmax_f32(vmax_f32(st[qi][1][0], st[qi][1][1]), vmax_f32(st[qi][1][2], st[qi][1][3])));
      mx = rows_max(mx);
      const float m_new = vmax_f32(m_run[qi], mx);
      const float mc = m_new * sl2e;
      const float alpha = __builtin_amdgcn_exp2f(m_run[qi] * sl2e - mc);
      float psum = 0.f;
      E8 pv8;
#pragma unroll
      for (int kt = 0; kt < 2; kt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qi][kt][r], sl2e, -mc));
          psum += pv;
          pv8[kt * 4 + r] = (E)pv;
        }
      pf[qi] = __builtin_bit_cast(i4, pv8);
      psum = rows_sum(psum);
      l_run[qi] = l_run[qi] * alpha + psum;
      m_run[qi] = m_new;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int dt = 0; dt < 8; dt++)
#pragma unroll
          for (int r = 0; r < 4; r++) o[dt][qi][r] *= alpha;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dt = 0; dt < 8; dt++) {
      const i4 vf = (i4){vlo[dt][0], vlo[dt][1], vhi[dt][0], vhi[dt][1]};
#pragma unroll
      for (int qi = 0; qi < 2; qi++) o[dt][qi] = mfma32<DT>(vf, pf[qi], o[dt][qi]);
    }
  }
  // ---- merge the four partial results: O = sum_v O_v 2^((m_v - M) c) / sum_v l_v 2^((m_v - M) c)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float *po = reinterpret_cast<float *>(wbuf);                  // [32 q][128 d] f32, 16-byte slots XOR-swizzled by the row
  float *pml = reinterpret_cast<float *>(wbuf + 32 * 512);      // [2][32]: m, l
#pragma unroll
  for (int qi = 0; qi < 2; qi++) {
    const int q = qi * 16 + li;
#pragma unroll
    for (int dt = 0; dt < 8; dt++)
      *reinterpret_cast<f4 *>(reinterpret_cast<unsigned char *>(po) + q * 512 + (((dt * 4 + g) ^ (q & 31)) << 4)) = o[dt][qi];
    if (g == 0) { pml[q] = m_run[qi]; pml[32 + q] = l_run[qi]; }
  }
  __syncthreads();
  {
    const int r8 = lane >> 3, c = lane & 7;     // this wave merges query rows wave*8 .. wave*8+7; a lane: one row, 16 d
    const int q = wave * 8 + r8;
    float mv[4], lv[4], M = -INFINITY;
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const float *ml = reinterpret_cast<const float *>(smem_dyn + v * WREG + 32 * 512);
      mv[v] = ml[q]; lv[v] = ml[32 + q];
      M = fmaxf(M, mv[v]);
    }
    float L = 0.f, acc[16];
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const float w = __builtin_amdgcn_exp2f((mv[v] - M) * sl2e);   // a wave without key blocks: m = -inf -> weight 0
      L += lv[v] * w;
      const unsigned char *pv = smem_dyn + v * WREG + q * 512;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const f4 x = *reinterpret_cast<const f4 *>(pv + (((c * 4 + j) ^ (q & 31)) << 4));
#pragma unroll
        for (int e = 0; e < 4; e++) acc[j * 4 + e] += x[e] * w;
      }
    }
    const float inv = 1.0f / L;
    const int row = q0 + q;
    if (row < T) {
      E8 o0, o1;
#pragma unroll
      for (int e = 0; e < 8; e++) { o0[e] = (E)(acc[e] * inv); o1[e] = (E)(acc[8 + e] * inv); }
      E *dst = out + ((size_t)b * tstride + row) * EMBED + h * HDIM + c * 16;
      *reinterpret_cast<E8 *>(dst) = o0;
      *reinterpret_cast<E8 *>(dst + 8) = o1;
    }
  }
}

// =================================================================================================
// small kernels
// =================================================================================================

// x[b,t,:] += pe[t,:]  (rows = B*T, 512 channels, 8 halfs per thread)
template <int DT>
__global__ void add_pos_embed_kernel(typename ElemT<DT>::t *__restrict__ x, const typename ElemT<DT>::t *__restrict__ pe, int T, size_t rows) {
  using E8 = typename ElemT<DT>::v8;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // 16-B chunk index
  if (i >= rows * (EMBED / 8)) return;
  size_t row = i / (EMBED / 8);
  int c = (int)(i - row * (EMBED / 8));
  int t = (int)(row % T);
  E8 a = reinterpret_cast<const E8 *>(x)[i];
  E8 pv = reinterpret_cast<const E8 *>(pe)[(size_t)t * (EMBED / 8) + c];
  E8 r;
#pragma unroll
  for (int e = 0; e < 8; e++) r[e] = (typename ElemT<DT>::t)((float)a[e] + (float)pv[e]);
  reinterpret_cast<E8 *>(x)[i] = r;
}

// y = LayerNorm(x) over 512 channels, eps 1e-5; one wave per row
// rows >= split_row use (gamma1, beta1): the refiner's two heads normalised in one launch
template <int DT>
__global__ __launch_bounds__(256) void layernorm_kernel(const typename ElemT<DT>::t *__restrict__ x, const float *__restrict__ gamma0,
                                                        const float *__restrict__ beta0, typename ElemT<DT>::t *__restrict__ y, size_t rows,
                                                        const float *__restrict__ gamma1, const float *__restrict__ beta1,
                                                        size_t split_row) {
  size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float *gamma = row >= split_row ? gamma1 : gamma0, *beta = row >= split_row ? beta1 : beta0;
  using E8 = typename ElemT<DT>::v8;
  E8 v = reinterpret_cast<const E8 *>(x + row * EMBED)[lane];
  float f[8], s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; e++) { f[e] = (float)v[e]; s += f[e]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  float mean = s * (1.0f / EMBED), q = 0.f;
#pragma unroll
  for (int e = 0; e < 8; e++) { f[e] -= mean; q += f[e] * f[e]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  float rstd = rsqrtf(q * (1.0f / EMBED) + 1e-5f);
  E8 r;
#pragma unroll
  for (int e = 0; e < 8; e++) r[e] = (typename ElemT<DT>::t)(f[e] * rstd * gamma[lane * 8 + e] + beta[lane * 8 + e]);
  reinterpret_cast<E8 *>(y + row * EMBED)[lane] = r;
}

// out[b,c] = mean_t x[b,t,c]  (f32 out).  Deterministic (no atomics: the arg-max over near-tied scores must not depend
// on summation order).  block = (b, 64-channel group); a lane loads 8 channels (16 B) of one token, so a wave covers 8
// tokens per load and walks the sequence in strides of 32 tokens (13 dependent steps for T = 400 instead of 100: at
// N = 1 this kernel was 30 us of a 570 us Track); the 8 token slots combine through shfl_xor, the 4 waves through LDS,
// both in a fixed order.
template <int DT>
__global__ __launch_bounds__(256) void token_mean_kernel(const typename ElemT<DT>::t *__restrict__ x, float *__restrict__ out, int T, int tstride) {
  using E8 = typename ElemT<DT>::v8;
  __shared__ float part[4][64];
  const int b = blockIdx.x, cg = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane >> 3, c8 = (lane & 7) * 8;
  const typename ElemT<DT>::t *src = x + (size_t)b * tstride * EMBED + cg * 64 + c8;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; e++) s[e] = 0.f;
  for (int t = wave * 8 + slot; t < T; t += 32) {
    E8 v = *reinterpret_cast<const E8 *>(src + (size_t)t * EMBED);
#pragma unroll
    for (int e = 0; e < 8; e++) s[e] += (float)v[e];
  }
#pragma unroll
  for (int e = 0; e < 8; e++) {
    s[e] += __shfl_xor(s[e], 8);
    s[e] += __shfl_xor(s[e], 16);
    s[e] += __shfl_xor(s[e], 32);
  }
  if (slot == 0) {
#pragma unroll
    for (int e = 0; e < 8; e++) part[wave][c8 + e] = s[e];
  }
  __syncthreads();
  if (wave == 0) out[(size_t)b * EMBED + cg * 64 + lane] = (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) / (float)T;
}

// out[b,:] = mean_t LayerNorm(x[b,t,:]): the encoder's second LayerNorm feeds nothing but the token mean, so the normalised
// tensor never goes to memory (two 103 MB passes per head at N = 252).  One workgroup per sequence, 16 waves, a wave per row
// (the arithmetic of layernorm_kernel incl. the rounding to the element type), per-lane column sums, fixed-order reduction.
// Sequences b >= split_b use (gamma1, beta1): the refiner's two heads in one launch (Track).
template <int DT>
__global__ __launch_bounds__(1024) void layernorm_mean_kernel(const typename ElemT<DT>::t *__restrict__ x, const float *__restrict__ gamma0,
                                                              const float *__restrict__ beta0, const float *__restrict__ gamma1,
                                                              const float *__restrict__ beta1, int split_b, float *__restrict__ out, int T,
                                                              int tstride) {
  using E = typename ElemT<DT>::t;
  using E8 = typename ElemT<DT>::v8;
  __shared__ float part[16][EMBED];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float *gamma = b >= split_b ? gamma1 : gamma0, *beta = b >= split_b ? beta1 : beta0;
  float gm[8], bt[8], acc[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { gm[e] = gamma[lane * 8 + e]; bt[e] = beta[lane * 8 + e]; acc[e] = 0.f; }
  const E *src = x + (size_t)b * tstride * EMBED;
  auto fold = [&](const E8 &v) {
    float f[8], s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) { f[e] = (float)v[e]; s += f[e]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    float mean = s * (1.0f / EMBED), q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) { f[e] -= mean; q += f[e] * f[e]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    float rstd = rsqrtf(q * (1.0f / EMBED) + 1e-5f);
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] += (float)(E)(f[e] * rstd * gm[e] + bt[e]);
  };
  // four rows of a wave in flight (rows wave, wave+16, ...: ascending order inside a wave, so the sums do not depend on the unroll)
  for (int t = wave; t < T; t += 64) {
    E8 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (t + 16 * j < T) v[j] = reinterpret_cast<const E8 *>(src + (size_t)(t + 16 * j) * EMBED)[lane];
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (t + 16 * j < T) fold(v[j]);
  }
#pragma unroll
  for (int e = 0; e < 8; e++) part[wave][lane * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < EMBED) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; w++) s += part[w][threadIdx.x];
    out[(size_t)b * EMBED + threadIdx.x] = s / (float)T;
  }
}

// y[b,o] = bias[o] + sum_c x[b,c] W[o,c]   (f32).  One wave per (output o, block of 8 rows b): the weight row lives in registers
// (C = 512: 8 floats per lane) and is reused for the 8 rows, so W is read B/8 times instead of B times (the 512x512 out_proj of
// the score-net at N = 252: 130 MB -> 16 MB of L2 reads); the summation order of a (b, o) pair is the same as one wave per output.
__global__ __launch_bounds__(256) void small_linear_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                                           const float *__restrict__ bias, float *__restrict__ y, int B,
                                                           int O, int C) {
  const int nbb = (B + 7) / 8;
  size_t widx = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (widx >= (size_t)nbb * O) return;
  const int bb = (int)(widx / O), o = (int)(widx - (size_t)bb * O);
  const int b0 = bb * 8, nb = min(8, B - b0);
  if (C == 512) {
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = W[(size_t)o * C + lane + 64 * j];
    for (int i = 0; i < nb; i++) {
      const float *xr = x + (size_t)(b0 + i) * C;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; j++) s += xr[lane + 64 * j] * w[j];
#pragma unroll
      for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k);
      if (lane == 0) y[(size_t)(b0 + i) * O + o] = s + bias[o];
    }
    return;
  }
  for (int i = 0; i < nb; i++) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += x[(size_t)(b0 + i) * C + c] * W[(size_t)o * C + c];
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k);
    if (lane == 0) y[(size_t)(b0 + i) * O + o] = s + bias[o];
  }
}

// the same for two independent layers of equal shape in ONE launch (blockIdx.y picks the layer): the refiner's two heads at Track
struct SmallLinear2 {
  const float *x[2], *W[2], *bias[2];
  float *y[2];
};
__global__ __launch_bounds__(256) void small_linear2_kernel(const SmallLinear2 a, int B, int O, int C) {
  const int h = blockIdx.y;
  size_t widx = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (widx >= (size_t)B * O) return;
  int b = (int)(widx / O), o = (int)(widx - (size_t)b * O);
  const float *x = a.x[h], *W = a.W[h];
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += x[(size_t)b * C + c] * W[(size_t)o * C + c];
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k);
  if (lane == 0) a.y[h][widx] = s + a.bias[h][o];
}

// cat[i][:, :, C:2C] = cat[0][:, :, C:2C] for i in 1..N-1 (bordered [N,HP,WP,2C] tensor, interior pixels only); CB = bytes
// of C channels.  Used when every hypothesis shares one observed crop (Register's first refine iteration: the sampler
// gives all 252 poses the same translation, foundationpose_sampling.cpp:388-391, so transf_input is identical for all of them).
// Track's last kernel: both Linear(512,3) heads (waves 0..5: one output each, the same summation as small_linear2_kernel) and, once
// they are stored, RefinePostProcess of the one hypothesis (pose_update_one) -- small_linear2_kernel + pose_update_kernel in one launch
__global__ __launch_bounds__(384) void small_linear2_pose_kernel(const SmallLinear2 a, int C, const PoseUpdateFuse f) {
  __shared__ float out[6];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int h = wv / 3, o = wv - h * 3;
  const float *x = a.x[h], *W = a.W[h];
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += x[c] * W[(size_t)o * C + c];
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k);
  if (lane == 0) {
    const float y = s + a.bias[h][o];
    a.y[h][o] = y;
    out[wv] = y;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    pose_update_one(f.poses, out, out + 3, 0, f.diameter, f.poses_in ? f.poses_in : f.poses, f.extra_out);
    // Track's completion signal: the refined pose above went to host-pinned memory; release it to the host and raise the flag the
    // waiting thread polls (an end-of-graph hipStreamSynchronize costs the host ~10 us more than this store takes to arrive)
    if (f.done_flag) __hip_atomic_store(f.done_flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void broadcast_b_kernel(unsigned char *__restrict__ cat, int N, int HP, int WP, int H, int W, int pad, int CB) {
  const int chunks = CB / 16;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t per_img = (size_t)H * W * chunks;
  if (i >= per_img * (size_t)(N - 1)) return;
  int img = 1 + (int)(i / per_img);
  size_t r = i - (size_t)(img - 1) * per_img;
  int pix = (int)(r / chunks), ch = (int)(r - (size_t)pix * chunks);
  int y = pix / W, x = pix - y * W;
  size_t off = (((size_t)(y + pad)) * WP + (x + pad)) * (2 * CB) + CB + ch * 16;
  const size_t img_stride = (size_t)HP * WP * 2 * CB;
  *reinterpret_cast<i4 *>(cat + (size_t)img * img_stride + off) = *reinterpret_cast<const i4 *>(cat + off);
}

template <int DT>
__global__ void cast_f32_kernel(const float *__restrict__ in, typename ElemT<DT>::t *__restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (typename ElemT<DT>::t)in[i];
}

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
  void *pe = nullptr;                // [400,512], act_dt
  // 8-bit networks (PREC_FP8 / PREC_INT8), set by net_apply_q8: per-CHANNEL activation scales of the trunk tensors that feed an
  // 8-bit convolution (real = stored * scale, DT_I8: real = (stored + 128) * scale); act_oinv = 1 / scale on the device for the
  // producers that write an f16 stream tensor together with its 8-bit copy (DT_DUAL_*); bias_fix / tok_fix = the bias correction
  // solved by the calibration sweeps (fp_api.hip: fp_calibrate)
  bool q8_ready = false;
  int qdt = DT_FP8;                                  // element type of the 8-bit layers
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
  ~Net() {
    for (void *p : allocs) (void)hipFree(p);
  }
};

template <typename T>
static T *upload(Net *net, const std::vector<T> &h) {
  T *d = nullptr;
  if (hipMalloc((void **)&d, std::max<size_t>(h.size(), 1) * sizeof(T)) != hipSuccess) return nullptr;
  net->allocs.push_back(d);
  if (fp::memcpy_sync(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

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

// [a | b] device copy of two equally shaped Linear layers (weights already in kernel row order)
static bool make_grouped(Net *net, const ConvLayer &a, const ConvLayer &b, ConvLayer *g) {
  if (a.Cin != b.Cin || a.Cout != b.Cout || a.dt != b.dt) return false;
  const size_t nw = (size_t)a.Cout * a.Cin * elem_bytes(a.dt), nb = (size_t)a.Cout;
  unsigned char *w = nullptr;
  float *bias = nullptr;
  if (hipMalloc((void **)&w, 2 * nw) != hipSuccess) return false;
  net->allocs.push_back(w);
  if (hipMalloc((void **)&bias, 2 * nb * sizeof(float)) != hipSuccess) return false;
  net->allocs.push_back(bias);
  if (fp::memcpy_sync(w, a.w, nw, hipMemcpyDeviceToDevice) != hipSuccess || fp::memcpy_sync(w + nw, b.w, nw, hipMemcpyDeviceToDevice) != hipSuccess ||
      fp::memcpy_sync(bias, a.bias, nb * 4, hipMemcpyDeviceToDevice) != hipSuccess || fp::memcpy_sync(bias + nb, b.bias, nb * 4, hipMemcpyDeviceToDevice) != hipSuccess)
    return false;
  *g = a;
  g->w = w;
  g->bias = bias;
  g->wfrag = nullptr;
  g->wpack = nullptr;   // (the grouped launch never runs on gemm_k32_kernel)
  g->wpack128 = nullptr;
  if (a.wfrag && b.wfrag) {  // (a group's fragment-order copy has the size of its row-major copy: the same group stride serves both)
    unsigned char *wf = nullptr;
    if (hipMalloc((void **)&wf, 2 * nw) != hipSuccess) return false;
    net->allocs.push_back(wf);
    if (fp::memcpy_sync(wf, a.wfrag, nw, hipMemcpyDeviceToDevice) != hipSuccess || fp::memcpy_sync(wf + nw, b.wfrag, nw, hipMemcpyDeviceToDevice) != hipSuccess) return false;
    g->wfrag = wf;
  }
  g->wdeep = nullptr;
  if (a.wdeep && b.wdeep) {
    unsigned char *wd = nullptr;
    if (hipMalloc((void **)&wd, 2 * nw) != hipSuccess) return false;
    net->allocs.push_back(wd);
    if (fp::memcpy_sync(wd, a.wdeep, nw, hipMemcpyDeviceToDevice) != hipSuccess || fp::memcpy_sync(wd + nw, b.wdeep, nw, hipMemcpyDeviceToDevice) != hipSuccess) return false;
    g->wdeep = wd;
  }
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

// device copy of a host vector: allocated the first time, overwritten in place afterwards (re-calibration of an 8-bit network)
template <typename T>
static bool put(Net *net, T *&dst, const std::vector<T> &h) {
  if (!dst) { dst = upload(net, h); return dst != nullptr; }
  return fp::memcpy_sync(dst, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
}
static bool put_bytes(Net *net, unsigned char *&dst, const std::vector<unsigned char> &h) { return put<unsigned char>(net, dst, h); }

// element bytes [Cout][tap][Cin] -> every device layout the layer's schedules stream from
static bool upload_layouts(Net *net, const std::vector<unsigned char> &elems, int Cout, int ntaps, int Cin, int dt, ConvLayer *L) {
  const int K = ntaps * Cin, es = elem_bytes(dt);
  const auto rows = permute_rows(relayout_k(elems, Cout, ntaps, Cin, es), Cout, (size_t)K * es);
  if (!put_bytes(net, L->w, rows)) return false;
  if (((size_t)K * es) % 128 == 0 && Cout % 16 == 0) {   // (byte-level: every element type)
    if (!put_bytes(net, L->wfrag, fragment_order(rows, Cout, (size_t)K * es))) return false;
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
// s_in = null: all 1): per output row a scale sw (FP8: amax / 448, OCP e4m3, RNE, saturating; I8: amax / 127, RNE) and, for I8, the
// row sum of the quantised integers (the -128 offset of the unsigned activations contributes 128 * sum to every accumulator).
static std::vector<unsigned char> quantise_q8(const ConvLayer &L, int dt, const float *s_in, std::vector<float> *sw, std::vector<double> *qsum) {
  const int K = L.ntaps * L.Cin;
  std::vector<unsigned char> o((size_t)L.Cout * K);
  sw->assign(L.Cout, 1.f);
  qsum->assign(L.Cout, 0.0);
  std::vector<float> wf(K);
  for (int co = 0; co < L.Cout; co++) {
    const float *src = &L.rows_f32[(size_t)co * K];
    float amax = 0.f;
    for (int k = 0; k < K; k++) {
      wf[k] = s_in ? src[k] * s_in[k % L.Cin] : src[k];
      amax = std::max(amax, std::fabs(wf[k]));
    }
    const float sc = amax > 0.f ? amax / (dt == DT_FP8 ? 448.f : 127.f) : 1.f;
    (*sw)[co] = sc;
    double qs = 0;
    for (int k = 0; k < K; k++) {
      if (dt == DT_FP8) o[(size_t)co * K + k] = f32_to_e4m3_bits(wf[k] / sc);
      else {
        const int q = (int)std::max(-127.f, std::min(127.f, std::nearbyint(wf[k] / sc)));
        o[(size_t)co * K + k] = (unsigned char)(signed char)q;
        qs += q;
      }
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
  L->w = upload(net, w->data);
  L->b = upload(net, b->data);
  L->out = out; L->in = in;
  return L->w && L->b;
}

static bool make_ln(Net *net, const std::map<std::string, HostTensor> &m, const std::string &prefix, LNParams *L,
                    std::string *err) {
  const HostTensor *w, *b;
  if (!get(m, prefix + ".weight", &w, err) || !get(m, prefix + ".bias", &b, err)) return false;
  if (!shape_is(w, {EMBED}) || !shape_is(b, {EMBED})) { *err = prefix + ": unexpected LayerNorm shape"; return false; }
  L->g = upload(net, w->data);
  L->b = upload(net, b->data);
  return L->g && L->b;
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
  // PREC_FP8 / PREC_INT8: the 3x3 trunk convolutions from encodeA.2 on (91 % of the FLOPs) run on 8-bit operands; the stem and
  // encodeA.1 (bandwidth-bound, K = 294 / 576) and the transformer part stay in f16
  const int adt = prec == PREC_BF16 ? DT_BF16 : DT_F16;
  const int tdt = prec == PREC_FP8 ? DT_FP8 : prec == PREC_INT8 ? DT_I8 : adt;
  net->act_dt = adt;
  net->qdt = tdt;
  bool ok = make_stem(net.get(), m, "encodeA.0", adt, &net->a0, err) && make_conv(net.get(), m, "encodeA.1", 2, 128, 64, 3, adt, &net->a1, err);
  for (int i = 0; ok && i < 2; i++)
    for (int j = 0; ok && j < 2; j++) {
      std::string cj = ".conv" + std::to_string(j + 1);
      ok = make_conv(net.get(), m, "encodeA." + std::to_string(2 + i) + cj, 1, 128, 128, 3, tdt, &net->ra[i][j], err) &&
           make_conv(net.get(), m, "encodeAB." + std::to_string(i) + cj, 1, 256, 256, 3, tdt, &net->rb[i][j], err) &&
           make_conv(net.get(), m, "encodeAB." + std::to_string(3 + i) + cj, 1, 512, 512, 3, tdt, &net->rc[i][j], err);
    }
  ok = ok && make_conv(net.get(), m, "encodeAB.2", 2, 512, 256, 3, tdt, &net->b2, err);
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
    net->pe = upload(net.get(), to_elems(pe, 400, EMBED, adt, nullptr));
    net->pe_host = pe;
    ok = net->pe != nullptr;
    if (!ok) *err = "device allocation failed";
  }
  if (ok) {
    unsigned char *c = nullptr;
    const size_t nb = (size_t)N_TRUNK_ACT * 512 * (sizeof(float) + sizeof(long long));
    ok = hipMalloc((void **)&c, nb) == hipSuccess && fp::memset_sync(c, 0, nb) == hipSuccess;
    if (c) net->allocs.push_back(c);
    net->calib_sum = reinterpret_cast<long long *>(c);
    net->calib_amax = reinterpret_cast<float *>(c + (size_t)N_TRUNK_ACT * 512 * sizeof(long long));
    if (!ok) *err = "device allocation failed";
  }
  if (!ok) return nullptr;
  return net.release();
}

Net *net_load(const char *path, bool is_scorer, int prec, std::string *err) {
  try {  // a malformed file must not take the process down through the C ABI
    return net_load_impl(path, is_scorer, prec, err);
  } catch (const std::exception &e) {
    *err = std::string("loading ") + path + ": " + e.what();
    return nullptr;
  }
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

// One 8-bit layer: (weights) quantise its rows with the input-channel scales s_in folded in and upload every layout; then the epilogue
// tables.  accumulator -> real value: acc * sw (+ DT_I8: 128 * sw * sum_k q, the offset of the unsigned activations) + bias + fix;
// s_out_fold != null (the output is 8-bit ONLY): the consumer's inverse scales multiply both (legal: no residual, ReLU).
static int apply_q8_layer(Net *net, ConvLayer &l, int dt, const float *s_in, const float *bias_fix, const float *s_out_fold, bool weights) {
  const int Cout = l.Cout;
  if (weights) {
    std::vector<float> sw;
    std::vector<double> qsum;
    const auto elems = quantise_q8(l, dt, s_in, &sw, &qsum);
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
int net_apply_q8(Net *net, const float *amax, const float *bias_fix, const float *tok_fix, bool weights) {
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
        sc[c] = dt == DT_FP8 ? m / 224.f : m * 1.25f / 255.f;
      }
      net->act_scale[a] = sc;
      std::vector<float> inv(C);
      for (int c = 0; c < C; c++) inv[c] = 1.f / sc[c];
      if (!put(net, net->act_oinv[a], inv) || !put(net, net->act_scale_dev[a], sc)) { set_error("net_apply_q8: device upload failed"); return 1; }
    }
  }
  FP_CHECK(!net->act_scale[1].empty(), "net_apply_q8: bias update before the scales were set");
  for (int i = 0; i < 13; i++) {
    ConvLayer &l = *L[i];
    const int Cout = l.Cout;
    if (bias_fix) net->bias_fix[i].assign(bias_fix + (size_t)i * 512, bias_fix + (size_t)i * 512 + Cout);
    else if (net->bias_fix[i].empty()) net->bias_fix[i].assign(Cout, 0.f);
  }
  for (int i = 0; i < 13; i++)
    if (apply_q8_layer(net, *L[i], dt, net->act_scale[i + 1].data(), net->bias_fix[i].data(),
                       q8_folded_out(i) ? net->act_scale[i + 2].data() : nullptr, weights)) return 1;
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
  ~NNScratch() {
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
  if (N <= ws->cap) return 0;
  if (ws->buf) (void)hipFree(ws->buf);
  if (ws->f32) (void)hipFree(ws->f32);
  ws->buf = nullptr; ws->f32 = nullptr; ws->cap = 0;
  int cap = std::max(N, 8);
  g_alloc_epoch++;
  FP_HIP_OK(hipMalloc((void **)&ws->buf, (size_t)cap * per_hyp_bytes(ws)));
  FP_HIP_OK(hipMalloc((void **)&ws->f32, (size_t)cap * EMBED * sizeof(float)));
  // the zero borders are written here once and never again: every producer stores interiors only, and the arena is
  // carved by CAPACITY (not by the current N), so an image slot's border never moves
  FP_HIP_OK(hipMemsetAsync(ws->buf, 0, (size_t)cap * PER_HYP, s));
  if (ws->q8) FP_HIP_OK(hipMemsetAsync(ws->buf + (size_t)cap * PER_HYP, ws->q8 == DT_I8 ? 0x80 : 0, (size_t)cap * PER_HYP_Q8, s));
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
FP_HOOK g_fuse_pose = 1;       // Track: both Linear(512,3) heads + RefinePostProcess in one kernel (small_linear2_pose_kernel)
FP_HOOK g_gemm_wpack = 1;      // gemm_k32_kernel streams its weights from the stage-order copy (one address + one M0 per four LDS-DMA pieces)
FP_HOOK g_deep_wpack = 1;      // conv_deep_kernel streams its weights from the stage-order copy
FP_HOOK g_big_wpack = 1;       // conv_big_pp_kernel streams its weights from the stage-order copy
FP_HOOK g_halo_wpack = 1;      // conv_halo_kernel streams its weights from the stage-order copy
FP_HOOK g_i8_stream = 0;       // test build A/B: 1 = INT8 networks with an 8-bit residual stream (run_trunk_i8; faster, but its common-mode error is frame-specific: DESIGN.md section 4.4)
FP_HOOK g_smallx_pf = 0;       // A/B (test build): prefetch depth of conv_smallx_kernel<2,4> (4 or 2; 0 = the default 3)
FP_HOOK g_smallm = 1;          // small problems (Track, a few objects) on conv_smallm_kernel: K split over the waves of a workgroup, no split-K slabs / reduce launch
FP_HOOK g_att_variant = 1;     // 1 = attention32_kernel (8 = without the XCD remap); round-1 kernel: 2 remap + 16-B stores, 3 no XCD remap, 5 remap + 2-B stores, 7 neither

// One launch = (once per launch site, element type and DEVICE) dynamic-LDS opt-in + the launch itself.
#define FP_LAUNCH(KERN, grid, block, lds_bytes, stream, ...)                                                                        \
  do {                                                                                                                              \
    static fp::PerDeviceOnce fp_attr_once_;                                                                                         \
    if (fp_attr_once_.first()) (void)hipFuncSetAttribute((const void *)(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
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
  double flops = 2.0 * (double)p.M * p.Cout * (L.algo_K > 0 ? L.algo_K : p.Ktot);
  double bytes = ((double)NB * H * W * L.Cin + (double)p.M * p.Cout * (has_res ? 2 : 1) + (double)p.Cout * p.Ktot) * elem_bytes(DT);
  constexpr int LDS_IG128 = 2 * (128 * 128 + 128 * 128), LDS_IG64 = 2 * (128 * 128 + 64 * 128);
  constexpr int LDS3_128 = 3 * (256 * 128 + 128 * 128);
  constexpr int LDS_BIG = 2 * (256 * 128 + 256 * 128);
  constexpr int LDS_HALO40 = ((10 * 42 + 7) / 8) * 1024 + 3 * 128 * 64;
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
    if (mt_big > 0) {
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
      if constexpr (!QOUT) {
        if (post_main) {
          FP_LAUNCH((conv_deep_kernel<64, DT, ODT, true>), dim3(((rows + 63) / 64) * n128), dim3(256), LDS_DEEP64, c.s, p);
          return 0;
        }
      }
      FP_LAUNCH((conv_deep_kernel<64, DT, ODT>), dim3(((rows + 63) / 64) * n128), dim3(256), LDS_DEEP64, c.s, p);
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
                    const float *oinv = nullptr, const float *rscale = nullptr) {
  ConvParams p;
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
  const size_t P1 = (size_t)NB2 * 42 * 42, P2 = (size_t)N * 42 * 42, P5 = (size_t)N * 22 * 22;
  const Act in = F(const_cast<void *>(nn_in)), stem = F(a.stem);
  if (run_conv(c, "conv_stem", net->a0, in, NB2, 80, 80, 2, stem, 1, true)) return 1;
  // encodeA.1 (f16 operands) starts the 128-channel stream: f16 x0 + 8-bit copy
  const Act x0 = F(a.x128[0]), x0q = Q(a.q128[0]), x1q = Q(a.q128[1]), x2 = F(a.x128[2]), x2q = Q(a.q128[2]);
  if (run_conv(c, "conv_a1", net->a1, stem, NB2, 80, 80, 1, x0, 1, true, nullptr, 0, 0, nullptr, nullptr, nullptr, &x0q, net->act_oinv[1])) return 1;
  if (run_conv(c, "conv_128", net->ra[0][0], x0q, NB2, 40, 40, 1, x1q, 1, true)) return 1;
  calib_record(c, 2, a.q128[1], P1, 128, q, net->act_scale_dev[2]);
  if (run_conv(c, "conv_128", net->ra[0][1], x1q, NB2, 40, 40, 1, x2, 1, true, &x0, 1, 0, nullptr, nullptr, nullptr, &x2q, net->act_oinv[3])) return 1;
  calib_record(c, 3, a.x128[2], P1, 128, DT_F16);
  if (run_conv(c, "conv_128", net->ra[1][0], x2q, NB2, 40, 40, 1, x1q, 1, true)) return 1;
  calib_record(c, 4, a.q128[1], P1, 128, q, net->act_scale_dev[4]);
  // the last encodeA conv writes the a|b channel concat (f16 + 8-bit copy)
  const Act cat = F(a.x256[0]), catq = Q(a.q256[0]);
  if (run_conv(c, "conv_128", net->ra[1][1], x1q, NB2, 40, 40, 1, cat, 1, true, &x2, 1, N, nullptr, nullptr, nullptr, &catq, net->act_oinv[5])) return 1;
  if (n_b == 1 && N > 1) {  // image N landed in cat[0][..,128:256]; replicate it for the other hypotheses (both copies)
    broadcast_b(c, a.x256[0], N, 256);
    broadcast_b(c, a.q256[0], N, 128);
  }
  calib_record(c, 5, a.x256[0], P2, 256, DT_F16);
  const Act y1q = Q(a.q256[1]), y2 = F(a.x256[2]), y2q = Q(a.q256[2]), y0 = F(a.x256[1]), y0q = Q(a.q256[0]);
  if (run_conv(c, "conv_256", net->rb[0][0], catq, N, 40, 40, 1, y1q, 1, true)) return 1;
  calib_record(c, 6, a.q256[1], P2, 256, q, net->act_scale_dev[6]);
  if (run_conv(c, "conv_256", net->rb[0][1], y1q, N, 40, 40, 1, y2, 1, true, &cat, 1, 0, nullptr, nullptr, nullptr, &y2q, net->act_oinv[7])) return 1;
  calib_record(c, 7, a.x256[2], P2, 256, DT_F16);
  if (run_conv(c, "conv_256", net->rb[1][0], y2q, N, 40, 40, 1, y1q, 1, true)) return 1;
  calib_record(c, 8, a.q256[1], P2, 256, q, net->act_scale_dev[8]);
  // y0 feeds only encodeAB.2: no f16 copy (0.2 GB per launch less); with a residual the consumer's scales cannot be folded into the
  // tables, so the epilogue scales (DT_QS_*)
  (void)y0;
  if (run_conv(c, "conv_256", net->rb[1][1], y1q, N, 40, 40, 1, y0q, 1, true, &y2, 1, 0, nullptr, nullptr, nullptr, nullptr, net->act_oinv[9])) return 1;
  calib_record(c, 9, a.q256[0], P2, 256, q, net->act_scale_dev[9]);
  const Act z0 = F(a.x512[0]), z0q = Q(a.q512[0]), z1q = Q(a.q512[1]), z2 = F(a.x512[2]), z2q = Q(a.q512[2]);
  if (run_conv(c, "conv_b2", net->b2, y0q, N, 40, 40, 1, z0, 1, true, nullptr, 0, 0, nullptr, nullptr, nullptr, &z0q, net->act_oinv[10])) return 1;
  calib_record(c, 10, a.x512[0], P5, 512, DT_F16);
  if (run_conv(c, "conv_512", net->rc[0][0], z0q, N, 20, 20, 1, z1q, 1, true)) return 1;
  calib_record(c, 11, a.q512[1], P5, 512, q, net->act_scale_dev[11]);
  if (run_conv(c, "conv_512", net->rc[0][1], z1q, N, 20, 20, 1, z2, 1, true, &z0, 1, 0, nullptr, nullptr, nullptr, &z2q, net->act_oinv[12])) return 1;
  calib_record(c, 12, a.x512[2], P5, 512, DT_F16);
  if (run_conv(c, "conv_512", net->rc[1][0], z2q, N, 20, 20, 1, z1q, 1, true)) return 1;
  calib_record(c, 13, a.q512[1], P5, 512, q, net->act_scale_dev[13]);
  const Act tok = F(a.tokens);
  bool pe_done = false;
  if (run_conv(c, "conv_512", net->rc[1][1], z1q, N, 20, 20, 1, tok, 0, true, &z2, 1, 0, nullptr, net->pe, &pe_done)) return 1;
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
  if (run_conv(c, "conv_stem", net->a0, in, NB2, 80, 80, 2, stem, 1, true)) return 1;
  const Act x0 = T(a.x128[0]), x1 = T(a.x128[1]), x2 = T(a.x128[2]), cat = T(a.x256[0]);
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
  Ctx c{s, prof, net, ws};
  const Arena a = carve(ws);
  if (run_trunk(c, a, nn_in, N, shared_b ? 1 : N)) return 1;
  const int dt = net->act_dt;
  const void *x = a.tokens;
  const size_t rows = (size_t)N * 400;
  const EncLayer *heads[2] = {&net->trans, &net->rot};
  float *outs[2] = {trans_dev, rot_dev};
  if (N == 1 && g_grouped_heads) {
    // Track: both heads in ONE launch per layer (Track is bound by its ~65 dependent launches, not by work).  Rows
    // [0,400) = translation head, [512,912) = rotation head (groups padded to the 128-row tile; the rows in between
    // carry don't-care values that no valid row ever reads: every op here is row-wise, attention is per sequence).
    const int G = 512;
    ConvGroup g_x{G, true, true}, g_in{G, false, true}, g_own{G, false, false};
    const EncLayer &T0 = net->trans, &R0 = net->rot;
    if (run_gemm(c, "gemm_qkv", net->g_in_proj, x, 2 * G, a.qkv, false, nullptr, &g_x)) return 1;
    if (run_attention(c, dt, a.qkv, a.att, 2, 400, G)) return 1;
    if (run_gemm(c, "gemm_512", net->g_out_proj, a.att, 2 * G, a.y1, false, x, &g_in)) return 1;   // + residual x (shared)
    run_layernorm(c, dt, a.y1, T0.ln1, a.y2, 2 * G, &R0.ln1, G);
    if (run_gemm(c, "gemm_512", net->g_lin1, a.y2, 2 * G, a.y1, true, nullptr, &g_own)) return 1;
    if (run_gemm(c, "gemm_512", net->g_lin2, a.y1, 2 * G, a.att, false, a.y2, &g_own)) return 1;   // + residual x1
    // (two sequences only: LayerNorm over 800 workgroup-rows + a 16-workgroup mean beat the one-workgroup-per-sequence fused kernel, 11 vs 20 us)
    run_layernorm(c, dt, a.att, T0.ln2, a.y1, 2 * G, &R0.ln2, G);
    run_token_mean(c, dt, a.y1, ws->f32, 2, 400, G);
    {
      ProfScope ps(c.prof, c.s, "small_linear", 2.0 * 2 * T0.head.out * T0.head.in, 0);
      SmallLinear2 a{{ws->f32, ws->f32 + EMBED}, {T0.head.w, R0.head.w}, {T0.head.b, R0.head.b}, {trans_dev, rot_dev}};
      if (fuse && g_fuse_pose && T0.head.out == 3 && R0.head.out == 3) {
        hipLaunchKernelGGL(small_linear2_pose_kernel, dim3(1), dim3(384), 0, c.s, a, T0.head.in, *fuse);
        if (fused_out) *fused_out = true;
      } else
        hipLaunchKernelGGL(small_linear2_kernel, dim3((unsigned)((T0.head.out + 3) / 4), 2), dim3(256), 0, c.s, a, 1, T0.head.out, T0.head.in);
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
  Ctx c{s, prof, net, ws};
  const Arena a = carve(ws);
  if (run_trunk(c, a, nn_in, N, N)) return 1;
  const size_t rows = (size_t)N * 400;
  if (run_gemm(c, "gemm_qkv", net->att.in_proj, a.tokens, (int)rows, a.qkv, false)) return 1;
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
  Ctx c{s, prof, net, ws};
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
// =================================================================================================
// MFMA micro-benchmark (measurement only): every wave issues `iters` x 8 independent v_mfma_f32_16x16x32_f16 from
// registers -- the rate the matrix pipes sustain with all 256 CUs busy at whatever clock the power limit allows.
__global__ __launch_bounds__(256) void mfma_peak_kernel(float *out, int iters, int zero_operands, unsigned long long *clk) {
  using fp::f4;
  using fp::h8;
  const int lane = threadIdx.x & 63;
  h8 a, b;
  unsigned st = 2654435761u * (unsigned)(blockIdx.x * 256 + threadIdx.x + 1);
#pragma unroll
  for (int i = 0; i < 8; i++) {  // random operands in [-0.5, 0.5): data-dependent power draw like real activations
    st = st * 1664525u + 1013904223u;
    a[i] = (_Float16)(((st >> 8) & 0xffff) / 65536.0f - 0.5f);
    st = st * 1664525u + 1013904223u;
    b[i] = (_Float16)(((st >> 8) & 0xffff) / 65536.0f - 0.5f);
    if (zero_operands) { a[i] = 0; b[i] = 0; }
  }
  unsigned long long c0 = 0, w0 = 0;
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
  f4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; j++) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++)  // inline asm: the builtin made hipcc shuffle the accumulators through AGPRs every iteration
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // drain the MFMA pipe before the accumulators are read
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 8; j++) sum += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (sum == 12345.678f) out[lane] = sum;  // keeps the accumulators live; never true in practice
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; }
}

// kernel-level test / micro-benchmark hooks (not part of the public C ABI; used by tests/test_nn_gpu.py and
// tools/bench_conv.py).  Host f32 in / out, fp16 on the device exactly like the production path.
// =================================================================================================
namespace {
template <typename T>
struct DevBuf {
  T *p = nullptr;
  explicit DevBuf(size_t n) { if (hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) p = nullptr; }
  ~DevBuf() { if (p) (void)hipFree(p); }
};
std::vector<__half> to_half(const float *src, size_t n) {
  std::vector<__half> h(n);
  for (size_t i = 0; i < n; i++) h[i] = __float2half(src[i]);
  return h;
}
// float <-> element bytes of a 2-byte tensor
std::vector<unsigned char> encode(const float *src, size_t n, int dt, float scale) {
  (void)scale;
  std::vector<float> tmp(src, src + n);
  return fp::to_elems(tmp, 1, (int)n, dt, nullptr);
}
float e4m3_to_f32(unsigned char b) {
  const int e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
  return (b & 0x80) ? -v : v;
}
void decode(const unsigned char *src, size_t n, int dt, float scale, float *dst) {
  for (size_t i = 0; i < n; i++) {
    if (dt == fp::DT_FP8) dst[i] = e4m3_to_f32(src[i]) * scale;
    else if (dt == fp::DT_BF16) { uint32_t u = (uint32_t)reinterpret_cast<const uint16_t *>(src)[i] << 16; std::memcpy(&dst[i], &u, 4); }
    else dst[i] = __half2float(reinterpret_cast<const __half *>(src)[i]);
  }
}
}  // namespace

extern "C" {

void fpt_set_att_variant(int v) { fp::g_att_variant = v; }
void fpt_set_smallm(int v) { fp::g_smallm = v; }
void fpt_set_smallm_maxkt(int v) { fp::g_smallm_maxkt = v; }
void fpt_set_smallx_pf(int v) { fp::g_smallx_pf = v; }
void fpt_set_gemm_wpack(int v) { fp::g_gemm_wpack = v; }
void fpt_set_fuse_pose(int v) { fp::g_fuse_pose = v; }
void fpt_set_halo_wpack(int v) { fp::g_halo_wpack = v; }
void fpt_set_big_wpack(int v) { fp::g_big_wpack = v; }
void fpt_set_deep_wpack(int v) { fp::g_deep_wpack = v; }
void fpt_set_smallm_maxt16(int v) { fp::g_smallm_maxt16 = v; }
void fpt_set_gemm_lds_store(int v) { fp::g_gemm_lds_store = v; }
void fpt_set_conv_lds_store(int v) { fp::g_conv_lds_store = v; }
void fpt_set_conv_variant(int v) { fp::g_conv_variant = v; }
void fpt_set_i8_stream(int v) { fp::g_i8_stream = v; }
void fpt_set_conv_ablate(int v) { fp::g_conv_ablate = v; }
void fpt_set_splitk_target(int v) { fp::g_splitk_target = v; }
void fpt_set_rem_splitk(int v) { fp::g_rem_splitk = v; }
void fpt_set_grouped_heads(int v) { fp::g_grouped_heads = v; }
void fpt_set_gemm_kernel(int v) { fp::g_gemm_kernel = v; }
void fpt_set_rem_kernel(int v) { fp::g_rem_kernel = v; }
void fpt_set_rem_small(int v) { fp::g_rem_small = v; }
void fpt_set_small_deep(int v) { fp::g_small_deep = v; }
void fpt_set_splitk_mid(int v) { fp::g_splitk_mid = v; }
void fpt_set_gemm_deep(int v) { fp::g_gemm_deep = v; }
void fpt_set_att_skv(int v) { fp::g_att_skv = v; }
void fpt_set_splitk_deep(int v) { fp::g_splitk_deep = v; }
void fpt_set_splitk_min_kt(int v) { fp::g_splitk_min_kt = v; }
void fpt_set_raster_strip_rows(int r) { fp::set_raster_strip_rows(r); }
void fpt_set_raster_strip_threads(int t) { fp::set_raster_strip_threads(t); }

// clock probe: allocate room for `blocks` records, run convs, then read back mean shader MHz and mean main-loop cycles
int fpt_clk_probe(int blocks, double *mhz_out, double *loop_cycles_out) {
  using namespace fp;
  if (blocks > 0) {
    if (g_clk_probe) (void)hipFree(g_clk_probe);
    FP_HIP_OK(hipMalloc((void **)&g_clk_probe, (size_t)blocks * 32));
    FP_HIP_OK(fp::memset_sync(g_clk_probe, 0, (size_t)blocks * 32));
    return 0;
  }
  FP_CHECK(g_clk_probe, "no probe");
  int n = -blocks;
  std::vector<unsigned long long> h((size_t)n * 4);
  FP_HIP_OK(hipDeviceSynchronize());
  FP_HIP_OK(fp::memcpy_sync(h.data(), g_clk_probe, h.size() * 8, hipMemcpyDeviceToHost));
  double sc = 0, sr = 0; int cnt = 0;
  for (int i = 0; i < n; i++) {
    if (h[i * 4 + 3] > h[i * 4 + 1]) { sc += (double)(h[i * 4 + 2] - h[i * 4]); sr += (double)(h[i * 4 + 3] - h[i * 4 + 1]); cnt++; }
  }
  if (mhz_out) *mhz_out = sr > 0 ? sc / sr * 100.0 : 0;  // wall_clock64 ticks at 100 MHz
  if (loop_cycles_out) *loop_cycles_out = cnt ? sc / cnt : 0;
  (void)hipFree(g_clk_probe);
  g_clk_probe = nullptr;
  return 0;
}

// x [NB,H,W,Cin] NHWC f32, w [Cout,KH,KW,Cin] f32, bias [Cout], res (optional) [NB,OH,OW,Cout]
// -> out f32 [NB,OH,OW,Cout] (or, with split_imgs > 0, [NB-split,OH,OW,2*Cout]).  iters > 1: returns mean ms in *ms_out.
// dt = element type of x / w / res on the device (DT_F16 / DT_BF16), out_dt = element type of the output; the scale arguments are
// ignored (kept for the callers' signature).  8-bit layers: fpt_conv_q8.
int fpt_conv_dt(const float *x, const float *w, const float *bias, const float *res, int NB, int H, int W, int Cin, int Cout,
                int KH, int KW, int stride, int pad, int OH, int OW, int relu, int split_imgs, float *out, int iters,
                float *ms_out, int dt, int out_dt, float in_scale, float res_scale, float out_scale) {
  using namespace fp;
  const int ip = pad;  // physical input border
  const int Hp = H + 2 * ip, Wp = W + 2 * ip;
  const int es = elem_bytes(dt), oes = elem_bytes(out_dt);
  size_t nx = (size_t)NB * Hp * Wp * Cin, nw = (size_t)Cout * KH * KW * Cin;
  size_t M = (size_t)NB * OH * OW;
  size_t nout = M * Cout;
  DevBuf<unsigned char> dx(nx * es), dres(nout * es), dout(nout * 2 * oes);
  FP_CHECK(dx.p && dres.p && dout.p, "fpt_conv: allocation failed");
  std::vector<float> xp(nx, 0.f);
  for (int n = 0; n < NB; n++)
    for (int y = 0; y < H; y++)
      for (int xx = 0; xx < W; xx++)
        for (int c = 0; c < Cin; c++)
          xp[(((size_t)n * Hp + y + ip) * Wp + xx + ip) * Cin + c] = x[(((size_t)n * H + y) * W + xx) * Cin + c];
  auto hx = encode(xp.data(), nx, dt, in_scale);
  FP_HIP_OK(fp::memcpy_sync(dx.p, hx.data(), hx.size(), hipMemcpyHostToDevice));
  FP_HIP_OK(fp::memset_sync(dout.p, 0, nout * 2 * oes));
  if (res) {
    auto hr = encode(res, nout, dt, res_scale);
    FP_HIP_OK(fp::memcpy_sync(dres.p, hr.data(), hr.size(), hipMemcpyHostToDevice));
  }
  Net net;
  ConvLayer L;
  L.Cin = Cin; L.Cout = Cout; L.KH = KH; L.KW = KW; L.stride = stride; L.pad = pad;
  FP_CHECK(finish_layer(&net, std::vector<float>(w, w + nw), std::vector<float>(bias, bias + Cout), Cout, KH * KW, Cin, dt, &L), "fpt_conv: weight upload failed");
  FP_CHECK(!is_q8(dt) && !is_q8(out_dt), "fpt_conv_dt: 2-byte element types only (8-bit layers: fpt_conv_q8)");
  Ctx c{nullptr, nullptr, &net};
  (void)OH; (void)OW;
  const Act ain{dx.p, dt, in_scale}, aout{dout.p, out_dt, out_scale}, ares{dres.p, dt, res_scale};
  hipEvent_t e0, e1;
  FP_HIP_OK(hipEventCreate(&e0));
  FP_HIP_OK(hipEventCreate(&e1));
  if (run_conv(c, "t", L, ain, NB, H, W, ip, aout, 0, relu != 0, res ? &ares : nullptr, 0, split_imgs)) return 1;
  FP_HIP_OK(hipDeviceSynchronize());
  if (iters > 1) {
    FP_HIP_OK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; i++)
      if (run_conv(c, "t", L, ain, NB, H, W, ip, aout, 0, relu != 0, res ? &ares : nullptr, 0, split_imgs)) return 1;
    FP_HIP_OK(hipEventRecord(e1, nullptr));
    FP_HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    FP_HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / iters;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  std::vector<unsigned char> ho(nout * oes);
  FP_HIP_OK(fp::memcpy_sync(ho.data(), dout.p, ho.size(), hipMemcpyDeviceToHost));
  decode(ho.data(), nout, out_dt, out_scale, out);
  return 0;
}
// One 8-bit convolution exactly as the 8-bit trunk runs it (net_apply_q8 / run_trunk_q8).
//   x [NB,H,W,Cin] f32 (>= 0 for DT_I8), s_in [Cin] per-channel activation scales: the device input is e4m3(x / s) or
//   (clamp(rint(x / s), 0, 255) ^ 0x80) with a border of "zero" bytes; w [Cout,KH,KW,Cin] f32 is quantised per row AFTER s_in is folded
//   in; res (optional) [NB,OH,OW,Cout] is f16 on the device.
//   mode 0: 8-bit output only, the consumer's scales s_out [Cout] folded into the epilogue tables -> outq (de-quantised values)
//   mode 1: f16 output only -> out16;  mode 2: f16 output + its 8-bit copy (value / s_out[c]) -> out16, outq
//   mode 3: the 8-bit copy alone, scaled in the epilogue (a layer with a residual whose f16 output nobody reads) -> outq
//   mode 4 / 5 (DT_I8): the residual is an 8-bit tensor itself, clamp(rint(res / s_res), 0, 255) ^ 0x80 (run_trunk_i8); 4: scaled 8-bit
//   output -> outq, 5: f16 output -> out16
// split_imgs > 0: the a|b channel concat ([NB - split, OH, OW, 2 * Cout]; s_out still indexed by the layer's output channel).
// wq_out (optional) [Cout*KH*KW*Cin]: the de-quantised weights the device used (w' / s_in, i.e. comparable with w), sw_out [Cout].
int fpt_conv_q8(const float *x, const float *s_in, const float *w, const float *bias, const float *res, int NB, int H, int W, int Cin, int Cout,
                int KH, int KW, int stride, int pad, int OH, int OW, int relu, int split_imgs, int mode, const float *s_out, float *out16,
                float *outq, int iters, float *ms_out, int dt, float *wq_out, const float *s_res) {
  using namespace fp;
  FP_CHECK(is_q8(dt) && mode >= 0 && mode <= 5 && (mode < 4 || (dt == DT_I8 && res && s_res)), "fpt_conv_q8: invalid arguments");
  const int ip = pad, Hp = H + 2 * ip, Wp = W + 2 * ip;
  const size_t nx = (size_t)NB * Hp * Wp * Cin, nw = (size_t)Cout * KH * KW * Cin, M = (size_t)NB * OH * OW, nout = M * Cout;
  DevBuf<unsigned char> dx(nx), dres(nout * 2), d16(nout * 2), dq(nout);
  FP_CHECK(dx.p && dres.p && d16.p && dq.p, "fpt_conv_q8: allocation failed");
  std::vector<unsigned char> hx(nx, dt == DT_I8 ? 0x80 : 0x00);
  for (int n = 0; n < NB; n++)
    for (int y = 0; y < H; y++)
      for (int xx = 0; xx < W; xx++)
        for (int c = 0; c < Cin; c++) {
          const float v = x[(((size_t)n * H + y) * W + xx) * Cin + c] / s_in[c];
          unsigned char b;
          if (dt == DT_FP8) b = f32_to_e4m3_bits(v);
          else b = (unsigned char)((int)std::max(0.f, std::min(255.f, std::nearbyint(v))) ^ 0x80);
          hx[(((size_t)n * Hp + y + ip) * Wp + xx + ip) * Cin + c] = b;
        }
  FP_HIP_OK(fp::memcpy_sync(dx.p, hx.data(), nx, hipMemcpyHostToDevice));
  FP_HIP_OK(fp::memset_sync(d16.p, 0, nout * 2));
  FP_HIP_OK(fp::memset_sync(dq.p, dt == DT_I8 ? 0x80 : 0, nout));
  float *rscale_dev = nullptr;
  if (res && mode >= 4) {
    std::vector<unsigned char> hr(nout);
    for (size_t i = 0; i < nout; i++) hr[i] = (unsigned char)((int)std::max(0.f, std::min(255.f, std::nearbyint(res[i] / s_res[i % Cout]))) ^ 0x80);
    FP_HIP_OK(fp::memcpy_sync(dres.p, hr.data(), nout, hipMemcpyHostToDevice));
  } else if (res) {
    auto hr = encode(res, nout, DT_F16, 1.f);
    FP_HIP_OK(fp::memcpy_sync(dres.p, hr.data(), hr.size(), hipMemcpyHostToDevice));
  }
  Net net;
  net.prec = dt == DT_FP8 ? PREC_FP8 : PREC_INT8;
  net.qdt = dt;
  ConvLayer L;
  L.Cin = Cin; L.Cout = Cout; L.KH = KH; L.KW = KW; L.stride = stride; L.pad = pad;
  FP_CHECK(finish_layer(&net, std::vector<float>(w, w + nw), std::vector<float>(bias, bias + Cout), Cout, KH * KW, Cin, dt, &L), "fpt_conv_q8: weight upload failed");
  if (apply_q8_layer(&net, L, dt, s_in, nullptr, mode == 0 ? s_out : nullptr, true)) return 1;
  if (wq_out) {  // what the device multiplies with, de-quantised and with s_in divided out again
    std::vector<float> sw; std::vector<double> qs;
    const auto e = quantise_q8(L, dt, s_in, &sw, &qs);
    for (size_t i = 0; i < nw; i++) {
      const int co = (int)(i / ((size_t)KH * KW * Cin)), ci = (int)(i % Cin);
      const float q = dt == DT_FP8 ? e4m3_to_f32(e[i]) : (float)(signed char)e[i];
      wq_out[i] = q * sw[co] / s_in[ci];
    }
  }
  float *oinv_dev = nullptr;
  if (mode >= 4) {
    rscale_dev = upload(&net, std::vector<float>(s_res, s_res + Cout));
    FP_CHECK(rscale_dev, "fpt_conv_q8: allocation failed");
  }
  if (mode >= 2 && mode != 5) {
    std::vector<float> inv(Cout);
    for (int c2 = 0; c2 < Cout; c2++) inv[c2] = 1.f / s_out[c2];
    oinv_dev = upload(&net, inv);
    FP_CHECK(oinv_dev, "fpt_conv_q8: allocation failed");
  }
  Ctx c{nullptr, nullptr, &net};
  (void)OH; (void)OW;
  const Act ain{dx.p, dt, 1.f}, a16{d16.p, DT_F16, 1.f}, aq{dq.p, dt, 1.f}, ares{dres.p, mode >= 4 ? DT_I8 : DT_F16, 1.f};
  auto once = [&]() {
    if (mode == 4) return run_conv(c, "t", L, ain, NB, H, W, ip, aq, 0, relu != 0, &ares, 0, split_imgs, nullptr, nullptr, nullptr, nullptr, oinv_dev, rscale_dev);
    if (mode == 5) return run_conv(c, "t", L, ain, NB, H, W, ip, a16, 0, relu != 0, &ares, 0, split_imgs, nullptr, nullptr, nullptr, nullptr, nullptr, rscale_dev);
    if (mode == 0) return run_conv(c, "t", L, ain, NB, H, W, ip, aq, 0, relu != 0, res ? &ares : nullptr, 0, split_imgs);
    if (mode == 1) return run_conv(c, "t", L, ain, NB, H, W, ip, a16, 0, relu != 0, res ? &ares : nullptr, 0, split_imgs);
    if (mode == 3) return run_conv(c, "t", L, ain, NB, H, W, ip, aq, 0, relu != 0, res ? &ares : nullptr, 0, split_imgs, nullptr, nullptr, nullptr, nullptr, oinv_dev);
    return run_conv(c, "t", L, ain, NB, H, W, ip, a16, 0, relu != 0, res ? &ares : nullptr, 0, split_imgs, nullptr, nullptr, nullptr, &aq, oinv_dev);
  };
  if (once()) return 1;
  FP_HIP_OK(hipDeviceSynchronize());
  if (iters > 1) {
    hipEvent_t e0, e1;
    FP_HIP_OK(hipEventCreate(&e0));
    FP_HIP_OK(hipEventCreate(&e1));
    FP_HIP_OK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; i++)
      if (once()) return 1;
    FP_HIP_OK(hipEventRecord(e1, nullptr));
    FP_HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    FP_HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  if ((mode == 1 || mode == 2 || mode == 5) && out16) {
    std::vector<unsigned char> ho(nout * 2);
    FP_HIP_OK(fp::memcpy_sync(ho.data(), d16.p, ho.size(), hipMemcpyDeviceToHost));
    decode(ho.data(), nout, DT_F16, 1.f, out16);
  }
  if (mode != 1 && mode != 5 && outq) {
    std::vector<unsigned char> ho(nout);
    FP_HIP_OK(fp::memcpy_sync(ho.data(), dq.p, nout, hipMemcpyDeviceToHost));
    // output layout [.., 2*Cout] with the concat: channel index modulo Cout selects the scale
    for (size_t i = 0; i < nout; i++) {
      const int ch = (int)(i % (split_imgs > 0 ? 2 * (size_t)Cout : (size_t)Cout)) % Cout;
      const float q = dt == DT_FP8 ? e4m3_to_f32(ho[i]) : (float)(ho[i] ^ 0x80);
      outq[i] = q * s_out[ch];
    }
  }
  return 0;
}
// The f16 -> 8-bit boundary layer (encodeA.1 of the 8-bit networks): f16 operands, f16 stream output + 8-bit copy (value / s_out[c]);
// out16 == null: the 8-bit copy alone (the INT8 trunk).
int fpt_conv_f16_dual(const float *x, const float *w, const float *bias, int NB, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                      int pad, int OH, const float *s_out, float *out16, float *outq, int qdt) {
  using namespace fp;
  FP_CHECK(is_q8(qdt), "fpt_conv_f16_dual: qdt must be an 8-bit type");
  const int ip = pad, Hp = H + 2 * ip, Wp = W + 2 * ip, OW = OH;
  const size_t nx = (size_t)NB * Hp * Wp * Cin, nw = (size_t)Cout * KH * KW * Cin, nout = (size_t)NB * OH * OW * Cout;
  DevBuf<unsigned char> dx(nx * 2), d16(nout * 2), dq(nout);
  FP_CHECK(dx.p && d16.p && dq.p, "fpt_conv_f16_dual: allocation failed");
  std::vector<float> xp(nx, 0.f);
  for (int n = 0; n < NB; n++)
    for (int y = 0; y < H; y++)
      for (int xx = 0; xx < W; xx++)
        for (int c = 0; c < Cin; c++) xp[(((size_t)n * Hp + y + ip) * Wp + xx + ip) * Cin + c] = x[(((size_t)n * H + y) * W + xx) * Cin + c];
  auto hx = encode(xp.data(), nx, DT_F16, 1.f);
  FP_HIP_OK(fp::memcpy_sync(dx.p, hx.data(), hx.size(), hipMemcpyHostToDevice));
  Net net;
  ConvLayer L;
  L.Cin = Cin; L.Cout = Cout; L.KH = KH; L.KW = KW; L.stride = stride; L.pad = pad;
  FP_CHECK(finish_layer(&net, std::vector<float>(w, w + nw), std::vector<float>(bias, bias + Cout), Cout, KH * KW, Cin, DT_F16, &L), "fpt_conv_f16_dual: weight upload failed");
  std::vector<float> inv(Cout);
  for (int c2 = 0; c2 < Cout; c2++) inv[c2] = 1.f / s_out[c2];
  float *oinv_dev = upload(&net, inv);
  FP_CHECK(oinv_dev, "fpt_conv_f16_dual: allocation failed");
  Ctx c{nullptr, nullptr, &net};
  const Act ain{dx.p, DT_F16, 1.f}, a16{d16.p, DT_F16, 1.f}, aq{dq.p, qdt, 1.f};
  if (out16 ? run_conv(c, "t", L, ain, NB, H, W, ip, a16, 0, true, nullptr, 0, 0, nullptr, nullptr, nullptr, &aq, oinv_dev)
            : run_conv(c, "t", L, ain, NB, H, W, ip, aq, 0, true, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, oinv_dev)) return 1;
  FP_HIP_OK(hipDeviceSynchronize());
  std::vector<unsigned char> h16(nout * 2), hq(nout);
  FP_HIP_OK(fp::memcpy_sync(h16.data(), d16.p, h16.size(), hipMemcpyDeviceToHost));
  FP_HIP_OK(fp::memcpy_sync(hq.data(), dq.p, nout, hipMemcpyDeviceToHost));
  if (out16) decode(h16.data(), nout, DT_F16, 1.f, out16);
  for (size_t i = 0; i < nout; i++) outq[i] = (qdt == DT_FP8 ? e4m3_to_f32(hq[i]) : (float)(hq[i] ^ 0x80)) * s_out[i % Cout];
  return 0;
}
// Host-only check of the weight layouts (no GPU): every (tile, K-step, lane) address a kernel forms into fragment_order /
// pack_stage_w / pack_stage_w128 must hold the bytes the same lane would have fetched from the row-major copy.
// Returns the number of mismatching 16-byte pieces (0 = all layouts agree), -1 for unsupported sizes.
long long fpt_check_weight_layouts(int Cout, int row_bytes) {
  using namespace fp;
  if (Cout % 256 || row_bytes % 128) return -1;
  std::vector<unsigned char> rows((size_t)Cout * row_bytes);
  unsigned st = 12345u;
  for (auto &b : rows) { st = st * 1664525u + 1013904223u; b = (unsigned char)(st >> 24); }
  long long bad = 0;
  auto same = [&](const unsigned char *a, const unsigned char *b) { return std::memcmp(a, b, 16) == 0; };
  const size_t KT = row_bytes / 128, S = row_bytes / 64;
  {  // conv_smallx_kernel: wrow = frag + (n0 >> 4) * 16 * krow_b + lane * 16;  + ni * 16 * krow_b + kt * 2048 + ks * 1024
    const auto f = fragment_order(rows, Cout, row_bytes);
    for (size_t t = 0; t < (size_t)Cout / 16; t++)
      for (size_t kt = 0; kt < KT; kt++)
        for (int ks = 0; ks < 2; ks++)
          for (int l = 0; l < 64; l++)
            bad += !same(&f[t * 16 * row_bytes + kt * 2048 + (size_t)ks * 1024 + (size_t)l * 16],
                         &rows[(t * 16 + (l & 15)) * row_bytes + kt * 128 + (size_t)ks * 64 + (size_t)(l >> 4) * 16]);
  }
  for (int TILE : {256, 128}) {  // gemm_k32_kernel (4 pieces per wave) / conv_halo_kernel (2): row = lane >> 2, swizzled chunk
    const auto pk = pack_stage_w(rows, Cout, row_bytes, TILE);
    const int PW = TILE / 64;
    for (size_t nt = 0; nt < (size_t)Cout / TILE; nt++)
      for (size_t s2 = 0; s2 < S; s2++)
        for (int wv = 0; wv < 4; wv++)
          for (int i = 0; i < PW; i++)
            for (int l = 0; l < 64; l++) {
              const int prow = l >> 2, gch = (l & 3) ^ ((0x78 >> (((prow >> 2) & 3) * 2)) & 3);
              // the kernel: base of (nt, wave) + s2 * (4 waves * PW KB) + i * 1024 + lane * 16
              const size_t at = ((nt * S * 4 + wv) * PW) * 1024 + s2 * (size_t)(4 * PW * 1024) + (size_t)i * 1024 + (size_t)l * 16;
              bad += !same(&pk[at], &rows[(nt * TILE + (size_t)(wv * PW + i) * 16 + prow) * row_bytes + s2 * 64 + (size_t)gch * 16]);
            }
  }
  for (int cfg = 0; cfg < 2; cfg++) {  // conv_big_pp_kernel (256 rows, 8 waves) / conv_deep_kernel, conv_halo8_kernel (128 rows, 4 waves)
    const int TILE = cfg ? 128 : 256, NW = cfg ? 4 : 8;
    const auto pk = pack_stage_w128(rows, Cout, row_bytes, TILE, NW);
    for (size_t nt = 0; nt < (size_t)Cout / TILE; nt++)
      for (size_t kt = 0; kt < KT; kt++)
        for (int wv = 0; wv < NW; wv++)
          for (int i = 0; i < 4; i++)
            for (int l = 0; l < 64; l++) {
              const int srow = l >> 3, g = (l & 7) ^ srow;
              const size_t at = ((nt * KT * NW + wv) * 4096) + kt * (size_t)(NW * 4096) + (size_t)i * 1024 + (size_t)l * 16;
              bad += !same(&pk[at], &rows[(nt * TILE + (size_t)(wv * 4 + i) * 8 + srow) * row_bytes + kt * 128 + (size_t)g * 16]);
              // conv_pp_kernel<128> reads the 128-row form as 8 waves x 2 pieces: piece P = 2 * wave + i at P KB of the (nt, kt) block
              if (cfg && i < 2) {
                for (int w8 = wv * 2; w8 < wv * 2 + 2; w8++) {
                  const size_t at8 = (nt * KT * 16 + (size_t)w8 * 2) * 1024 + kt * 16384 + (size_t)i * 1024 + (size_t)l * 16;
                  bad += !same(&pk[at8], &rows[(nt * 128 + (size_t)(w8 * 2 + i) * 8 + srow) * row_bytes + kt * 128 + (size_t)g * 16]);
                }
              }
            }
  }
  return bad;
}

int fpt_conv(const float *x, const float *w, const float *bias, const float *res, int NB, int H, int W, int Cin, int Cout,
             int KH, int KW, int stride, int pad, int OH, int OW, int relu, int split_imgs, float *out, int iters,
             float *ms_out) {
  return fpt_conv_dt(x, w, bias, res, NB, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, relu, split_imgs, out, iters, ms_out,
                     fp::DT_F16, fp::DT_F16, 1.f, 1.f, 1.f);
}

// qkv f32 [B,T,1536] -> out f32 [B,T,512]; dt = DT_F16 / DT_BF16
int fpt_attention_dt(const float *qkv, int B, int T, float *out, int dt) {
  using namespace fp;
  size_t nq = (size_t)B * T * 1536, no = (size_t)B * T * 512;
  DevBuf<unsigned char> dq(nq * 2), dout(no * 2);
  FP_CHECK(dq.p && dout.p, "fpt_attention: allocation failed");
  auto hq = encode(qkv, nq, dt, 1.f);
  FP_HIP_OK(fp::memcpy_sync(dq.p, hq.data(), nq * 2, hipMemcpyHostToDevice));
  Ctx c{nullptr, nullptr, nullptr};
  if (run_attention(c, dt, dq.p, dout.p, B, T)) return 1;
  FP_HIP_OK(hipDeviceSynchronize());
  std::vector<unsigned char> ho(no * 2);
  FP_HIP_OK(fp::memcpy_sync(ho.data(), dout.p, no * 2, hipMemcpyDeviceToHost));
  decode(ho.data(), no, dt, 1.f, out);
  return 0;
}
int fpt_attention(const float *qkv, int B, int T, float *out) { return fpt_attention_dt(qkv, B, T, out, fp::DT_F16); }


// concurrency stress: `nthreads` host threads, each with its own stream and buffers, run the same convolution `iters`
// times and compare every result bit-for-bit with their first one (device-side).  Returns the number of mismatching
// elements summed over all threads (0 = deterministic under contention), negative on failure.
__global__ void fpt_count_diff_kernel(const uint32_t *a, const uint32_t *b, size_t n, unsigned long long *cnt) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned local = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) local += a[i] != b[i];
  if (local) atomicAdd(cnt, (unsigned long long)local);
}

long long fpt_conv_stress(int NB0, int H0, int Cin0, int Cout0, int with_res, int iters, int nthreads, int mix) {
  using namespace fp;
  std::vector<long long> bad(nthreads, -1);
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++)
    th.emplace_back([&, t]() {
      // mix: odd threads run a different layer (the 256x256-tile kernel on 20x20 maps) next to the first thread's
      int NB = NB0, H = H0, Cin = Cin0, Cout = Cout0;
      if (mix && (t & 1)) { NB = 2 * NB0; H = 20; Cin = 512; Cout = 512; }
      const int Hp = H + 2, Wp = H + 2;
      size_t nx = (size_t)NB * Hp * Wp * Cin, nw = (size_t)Cout * 9 * Cin, nout = (size_t)NB * Hp * Wp * Cout;
      DevBuf<__half> dx(nx), dw(nw), dres(nout), dout(nout), dref(nout);
      DevBuf<float> db(Cout);
      DevBuf<unsigned long long> dcnt(1);
      if (!dx.p || !dw.p || !dres.p || !dout.p || !dref.p || !db.p || !dcnt.p) return;
      uint32_t st = 1234567u + 977u * t;
      auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
      std::vector<__half> hx(nx, __float2half(0.f)), hw(nw), hr(nout, __float2half(0.f));
      for (int n = 0; n < NB; n++)
        for (int y = 1; y <= H; y++)
          for (int x = 1; x <= H; x++)
            for (int c = 0; c < Cin; c++) hx[(((size_t)n * Hp + y) * Wp + x) * Cin + c] = __float2half(rnd());
      for (auto &v : hw) v = __float2half(rnd() * 0.05f);
      for (auto &v : hr) v = __float2half(rnd());
      std::vector<float> hb(Cout);
      for (auto &v : hb) v = rnd();
      hipStream_t s;
      if (hipStreamCreate(&s) != hipSuccess) return;
      (void)fp::memcpy_sync(dx.p, hx.data(), nx * 2, hipMemcpyHostToDevice);
      (void)fp::memcpy_sync(dw.p, hw.data(), nw * 2, hipMemcpyHostToDevice);
      (void)fp::memcpy_sync(dres.p, hr.data(), nout * 2, hipMemcpyHostToDevice);
      (void)fp::memcpy_sync(db.p, hb.data(), (size_t)Cout * 4, hipMemcpyHostToDevice);
      (void)fp::memset_sync(dout.p, 0, nout * 2);
      (void)fp::memset_sync(dref.p, 0, nout * 2);
      (void)fp::memset_sync(dcnt.p, 0, 8);
      Net net;
      ConvLayer L;
      L.w = (unsigned char *)dw.p; L.bias = db.p; L.Cin = Cin; L.Cout = Cout; L.KH = 3; L.KW = 3; L.stride = 1; L.pad = 1;
      NNScratch ws;
      Ctx c{s, nullptr, &net, &ws};
      const Act ain{dx.p, DT_F16, 1.f}, aref{dref.p, DT_F16, 1.f}, aout{dout.p, DT_F16, 1.f}, ares{dres.p, DT_F16, 1.f};
      if (run_conv(c, "t", L, ain, NB, H, H, 1, aref, 1, true, with_res ? &ares : nullptr, 1, 0)) return;
      (void)hipStreamSynchronize(s);
      for (int i = 0; i < iters; i++) {
        if (run_conv(c, "t", L, ain, NB, H, H, 1, aout, 1, true, with_res ? &ares : nullptr, 1, 0)) return;
        hipLaunchKernelGGL(fpt_count_diff_kernel, dim3(1024), dim3(256), 0, s, (const uint32_t *)dout.p, (const uint32_t *)dref.p,
                           nout / 2, dcnt.p);
      }
      unsigned long long cnt = 0;
      (void)hipMemcpyAsync(&cnt, dcnt.p, 8, hipMemcpyDeviceToHost, s);
      (void)hipStreamSynchronize(s);
      (void)hipStreamDestroy(s);
      bad[t] = (long long)cnt;
    });
  for (auto &x : th) x.join();
  long long tot = 0;
  for (auto b : bad) {
    if (b < 0) return -1;
    tot += b;
  }
  return tot;
}

// LDS canary: workgroups that own `words` dwords of LDS each fill them with a pattern and keep re-checking it while a
// convolution runs on another stream; a non-zero return means some kernel wrote outside its own LDS allocation.
__global__ void fpt_lds_canary_kernel(int words, int spins, unsigned long long *bad) {
  extern __shared__ unsigned canary[];
  const unsigned pat = 0xC0FFEE00u ^ (blockIdx.x * 2654435761u);
  for (int i = threadIdx.x; i < words; i += blockDim.x) canary[i] = pat + i;
  __syncthreads();
  unsigned local = 0;
  for (int r = 0; r < spins; r++) {
    for (int i = threadIdx.x; i < words; i += blockDim.x) local += canary[i] != pat + i;
    __builtin_amdgcn_s_sleep(64);
  }
  if (local) atomicAdd(bad, (unsigned long long)local);
}

long long fpt_lds_canary(int NB, int H, int Cin, int Cout, int iters, int canary_bytes) {
  using namespace fp;
  unsigned long long *dbad = nullptr;
  if (hipMalloc((void **)&dbad, 8) != hipSuccess || fp::memset_sync(dbad, 0, 8) != hipSuccess) return -1;
  std::atomic<int> stop{0};
  std::thread canary([&]() {
    hipStream_t s;
    if (hipStreamCreate(&s) != hipSuccess) return;
    while (!stop.load()) {
      hipLaunchKernelGGL(fpt_lds_canary_kernel, dim3(2048), dim3(64), (size_t)canary_bytes, s, canary_bytes / 4, 200, dbad);
      (void)hipStreamSynchronize(s);
    }
    (void)hipStreamDestroy(s);
  });
  long long rc = fpt_conv_stress(NB, H, Cin, Cout, 1, iters, 1, 0);
  stop.store(1);
  canary.join();
  unsigned long long bad = 0;
  (void)fp::memcpy_sync(&bad, dbad, 8, hipMemcpyDeviceToHost);
  (void)hipFree(dbad);
  return rc < 0 ? rc : (long long)bad;
}

// Inter-kernel visibility under concurrency: every thread owns a stream and a buffer and alternates
//   writer (buf[i] = f(i, iteration))  ->  checker (counts buf[i] != f(i, iteration))
// on it.  Same-stream ordering makes any non-zero count a platform-level visibility failure (stale data from the
// previous iteration), independent of this library's kernels.
__global__ void fpt_vis_write_kernel(float4 *buf, size_t n, unsigned it) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { float v = (float)((i * 7u + it * 13u) & 0xffff); buf[i] = make_float4(v, v + 1.f, v + 2.f, v + 3.f); }
}
__global__ void fpt_vis_check_kernel(const float4 *buf, size_t n, unsigned it, unsigned long long *bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // gather-style read (like the rasteriser reading vertex attributes): a different workgroup than the writer's
  size_t j = (i * 2654435761ull) % n;
  float v = (float)((j * 7u + it * 13u) & 0xffff);
  float4 x = buf[j];
  if (x.x != v || x.y != v + 1.f || x.z != v + 2.f || x.w != v + 3.f) atomicAdd(bad, 1ull);
}
long long fpt_visibility_stress(int nthreads, int iters, int mbytes) {
  std::vector<long long> bad(nthreads, -1);
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++)
    th.emplace_back([&, t]() {
      const size_t n = (size_t)mbytes * (1 << 20) / 16;
      float4 *buf = nullptr;
      unsigned long long *dbad = nullptr;
      hipStream_t s;
      if (hipMalloc((void **)&buf, n * 16) != hipSuccess || hipMalloc((void **)&dbad, 8) != hipSuccess || hipStreamCreate(&s) != hipSuccess) return;
      (void)hipMemsetAsync(dbad, 0, 8, s);
      const unsigned grid = (unsigned)((n + 255) / 256);
      for (int it = 0; it < iters; it++) {
        hipLaunchKernelGGL(fpt_vis_write_kernel, dim3(grid), dim3(256), 0, s, buf, n, (unsigned)(it + 1000 * t));
        hipLaunchKernelGGL(fpt_vis_check_kernel, dim3(grid), dim3(256), 0, s, buf, n, (unsigned)(it + 1000 * t), dbad);
      }
      unsigned long long b = 0;
      (void)hipMemcpyAsync(&b, dbad, 8, hipMemcpyDeviceToHost, s);
      (void)hipStreamSynchronize(s);
      (void)hipStreamDestroy(s);
      (void)hipFree(buf);
      (void)hipFree(dbad);
      bad[t] = (long long)b;
    });
  for (auto &x : th) x.join();
  long long tot = 0;
  for (auto b : bad) {
    if (b < 0) return -1;
    tot += b;
  }
  return tot;
}

// timing hook: random QKV resident in HBM, `iters` launches, returns ms per launch (negative on failure)
float fpt_attention_bench(int B, int T, int iters, int variant) {
  using namespace fp;
  const char *pe = getenv("FPT_QKV_LD");   // row pitch experiment (attention32_kernel only)
  const int ld = pe ? atoi(pe) : 1536;
  size_t nq = (size_t)B * T * ld, no = (size_t)B * T * 512;
  DevBuf<__half> dq(nq), dout(no);
  if (!dq.p || !dout.p) return -1.f;
  std::vector<__half> hq(nq);
  uint32_t st = 12345u;
  for (size_t i = 0; i < nq; i++) { st = st * 1664525u + 1013904223u; hq[i] = __float2half(((st >> 8) & 0xffff) / 65536.0f - 0.5f); }
  if (fp::memcpy_sync(dq.p, hq.data(), nq * 2, hipMemcpyHostToDevice) != hipSuccess) return -1.f;
  Ctx c{nullptr, nullptr, nullptr};
  int saved = g_att_variant;
  g_att_variant = variant;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
  for (int i = 0; i < 3; i++) run_attention(c, DT_F16, dq.p, dout.p, B, T, 0, ld);
  (void)hipEventRecord(e0, nullptr);
  for (int i = 0; i < iters; i++) run_attention(c, DT_F16, dq.p, dout.p, B, T, 0, ld);
  (void)hipEventRecord(e1, nullptr);
  float ms = -1.f;
  if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) ms = -(float)iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  g_att_variant = saved;
  return ms / iters;
}

// sustained dense fp16 MFMA rate in TFLOP/s (whole chip, `waves_per_simd` waves on every SIMD); negative on failure
// the same with v_mfma_f32_32x32x16_f16 (4 independent 32x32 accumulators): half the instructions and half the operand reads per flop
__global__ __launch_bounds__(256) void mfma_peak32_kernel(float *out, int iters, int zero_operands, unsigned long long *clk) {
  using fp::h8;
  typedef float f16v __attribute__((ext_vector_type(16)));
  const int lane = threadIdx.x & 63;
  h8 a, b;
  unsigned st = 2654435761u * (unsigned)(blockIdx.x * 256 + threadIdx.x + 1);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    st = st * 1664525u + 1013904223u;
    a[i] = (_Float16)(((st >> 8) & 0xffff) / 65536.0f - 0.5f);
    st = st * 1664525u + 1013904223u;
    b[i] = (_Float16)(((st >> 8) & 0xffff) / 65536.0f - 0.5f);
    if (zero_operands) { a[i] = 0; b[i] = 0; }
  }
  unsigned long long c0 = 0, w0 = 0;
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
  f16v acc[4];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int e = 0; e < 16; e++) sum += acc[j][e];
  if (sum == 12345.678f) out[lane] = sum;
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; }
}

float fpt_mfma_peak(int iters, int waves_per_simd, int zero_operands, double *mhz) {
  hipStream_t s;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return -1.f;
  hipDeviceProp_t prop;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1.f;
  const int wgs = prop.multiProcessorCount * waves_per_simd;  // 256 threads = one wave per SIMD of a CU
  DevBuf<float> out(64);
  DevBuf<unsigned long long> clk(2);
  hipEvent_t e0, e1;
  if (!out.p || !clk.p || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
  const bool big = (zero_operands & 2) != 0;   // bit 1: the 32x32x16 form (same flops per iteration: 4 x 32768 = 8 x 16384)
  zero_operands &= 1;
  if (big) hipLaunchKernelGGL(mfma_peak32_kernel, dim3(wgs), dim3(256), 0, s, out.p, iters / 4, zero_operands, (unsigned long long *)nullptr);
  else hipLaunchKernelGGL(mfma_peak_kernel, dim3(wgs), dim3(256), 0, s, out.p, iters / 4, zero_operands, (unsigned long long *)nullptr);  // warm-up / clock ramp
  (void)hipEventRecord(e0, s);
  if (big) hipLaunchKernelGGL(mfma_peak32_kernel, dim3(wgs), dim3(256), 0, s, out.p, iters, zero_operands, clk.p);
  else hipLaunchKernelGGL(mfma_peak_kernel, dim3(wgs), dim3(256), 0, s, out.p, iters, zero_operands, clk.p);
  (void)hipEventRecord(e1, s);
  float ms = -1.f;
  const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(s);
  if (!ok || ms <= 0.f) return -1.f;
  if (mhz) {  // shader clock during the run: cycle counter against the 100 MHz wall clock
    unsigned long long h[2] = {0, 0};
    if (fp::memcpy_sync(h, clk.p, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -1.f;
    *mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
  }
  const double flops = (double)wgs * 4.0 * (double)iters * 8.0 * 16384.0;
  return (float)(flops / (ms * 1e-3) / 1e12);
}

}  // extern "C"
#endif  // FP_TEST_HOOKS
