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
//     conv_igemm3 / conv_pp / conv_big kernels: earlier schedules kept behind fpt_set_conv_variant for A/B.
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
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

static constexpr int EMBED = 512, HEADS = 4, HDIM = 128;

// =================================================================================================
// implicit-GEMM convolution
// =================================================================================================

// LDS-DMA issued from inline asm: hipcc neither counts these loads nor inserts its conservative `s_waitcnt vmcnt(0)`
// in front of a new LDS-DMA while an older one is in flight (it cannot tell the LDS stages apart), so the counted
// waits in conv_igemm3_kernel are authoritative.  M0 (LDS destination base) is saved and restored inside the statement
// (cdna_hip_programming.md §5.7).  lds_addr must be wave-uniform; the 16 bytes land at lds_addr + lane*16.
__device__ __forceinline__ void glds16_asm(const void *gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
}

struct ConvParams {
  const __half *in;     // [NB, H+2*ipad, W+2*ipad, Cin]  (zero border of width ipad >= pad is physically present)
  const __half *w;      // [Cout][K] in kernel K order (relayout_k)
  const float *bias;    // [Cout]
  const __half *res;    // optional residual [NB, OH+2*rpad, OW+2*rpad, res_ld]
  __half *out;          // [NB', OH+2*opad, OW+2*opad, out_ld]
  int NB, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
  int ipad, opad, rpad;
  int M, Ktot, relu, out_ld, res_ld, split_imgs;
  int ntaps;  // KH*KW; for Cin >= 64 the K order is (64-channel chunk outer, tap inner) so the taps of a chunk
              // are consecutive K-steps and their overlapping input pixels are re-read while still L2-resident
  // byte offset of K-step kt's X slab relative to a row's (tap 0, channel 0) address; host-filled, read with s_load
  unsigned koff[80];
  unsigned koff32[160];  // same per 32-wide K-step (conv_pp32_kernel)
  int m_begin;            // first output row handled by this launch (hybrid 256^2 + 128^2 launches)
  int ksplit, kt_per;     // split-K (small problems): K-steps [split*kt_per, ...) per workgroup, fp32 partial slabs
  float *partial;         // [ksplit][M - m_begin][Cout]
  // weight groups along M (the refiner's two heads in ONE launch at small N, conv_igemm_kernel only): rows
  // [g*grp_rows, (g+1)*grp_rows) use weights w + g*grp_w_halfs and bias + g*Cout; grp_rows % 128 == 0, 0 = off.
  // in_shared / res_shared: the input / residual tensor has only the first group's rows and is read by every group.
  int grp_rows, in_shared, res_shared;
  unsigned grp_w_halfs;
  unsigned long long *clk;  // optional clock probe: per block {cycles0, realtime0, cycles1, realtime1}
};

// Epilogue shared by every conv schedule.
// Weight rows are PERMUTED on the host inside each block of 16*NI output channels (permute_rows): MFMA tile ni, row
// i = 4g + j (g = lane>>4, j = accumulator register) computes channel
//     NI == 4:  32*(j>>1) + 8*g + 4*(j&1) + ni        NI == 2:  8*g + 2*j + ni
// so a lane's accumulators hold 8 CONSECUTIVE channels per 16-byte store and the four lane groups of a store
// instruction cover 64 contiguous bytes of one pixel (the natural D^T layout gives 4 channels / 8 bytes per store and
// twice the store instructions; the 256x256 kernel spent 13 us of a 70 us workgroup in its store burst).
// All bias and residual loads are issued BEFORE the first store: on CDNA4 stores also count in vmcnt, so a load issued
// behind a store cannot be waited for without draining the store.
template <int MI, int NI, int EABL = 0, class PixFn>  // EABL (timing ablations): 1 = no stores, 2 = no residual loads
__device__ __forceinline__ void conv_epilogue_px(const ConvParams &p, f4 (&acc)[NI][MI], int n_base, int lane, PixFn pix,
                                                 int bias_off = 0, int res_img_off = 0) {
  static_assert(NI == 4 || NI == 2, "wave covers 64 or 32 channels");
  constexpr int NS = NI / 2;  // 16-byte stores per pixel per lane
  const int OHp = p.OH + 2 * p.opad, OWp = p.OW + 2 * p.opad;
  const int RHp = p.OH + 2 * p.rpad, RWp = p.OW + 2 * p.rpad;
  const int g = lane >> 4;
  const int nl = n_base + 8 * g;  // store k covers channels nl + 32*k .. +7
  float bv[NS][8];
#pragma unroll
  for (int k = 0; k < NS; k++) {
    float4 b0 = *reinterpret_cast<const float4 *>(p.bias + bias_off + nl + 32 * k), b1 = *reinterpret_cast<const float4 *>(p.bias + bias_off + nl + 32 * k + 4);
    bv[k][0] = b0.x; bv[k][1] = b0.y; bv[k][2] = b0.z; bv[k][3] = b0.w; bv[k][4] = b1.x; bv[k][5] = b1.y; bv[k][6] = b1.z; bv[k][7] = b1.w;
  }
  // pixels in groups of at most 8 fragments: a group's residual values (4 VGPRs per fragment and store) stay in registers
  constexpr int GB = MI > 8 ? (MI + 1) / 2 : MI;
#pragma unroll
  for (int g0 = 0; g0 < MI; g0 += GB) {
  size_t oofs[GB];
  bool ok[GB];
  h8 rv[NS][GB];
#pragma unroll
  for (int gi = 0; gi < GB; gi++) {
    const int mi = g0 + gi;
    if (mi >= MI) break;
    int img, oh, ow;
    ok[gi] = pix(mi, img, oh, ow);  // (img, oh, ow) must be a valid address even when !ok
    int choff = 0, oimg = img;
    if (p.split_imgs > 0 && img >= p.split_imgs) { oimg = img - p.split_imgs; choff = p.Cout; }
    oofs[gi] = (((size_t)oimg * OHp + oh + p.opad) * OWp + ow + p.opad) * p.out_ld + choff;
    if (p.res && !(EABL & 2)) {
      size_t rpix = ((size_t)(img - res_img_off) * RHp + oh + p.rpad) * RWp + ow + p.rpad;
#pragma unroll
      for (int k = 0; k < NS; k++)
        rv[k][gi] = ok[gi] ? *reinterpret_cast<const h8 *>(p.res + rpix * p.res_ld + nl + 32 * k) : (h8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
#pragma unroll
  for (int gi = 0; gi < GB; gi++) {
    const int mi = g0 + gi;
    if (mi >= MI) break;
    if (!ok[gi]) continue;
#pragma unroll
    for (int k = 0; k < NS; k++) {
      h8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int jj = (NI == 4) ? 2 * k + (e >> 2) : (e >> 1);
        const int ni = (NI == 4) ? (e & 3) : (e & 1);
        float v = acc[ni][mi][jj] + bv[k][e];
        if (p.res && !(EABL & 2)) v += (float)rv[k][gi][e];
        if (p.relu) v = fmaxf(v, 0.f);
        o[e] = (_Float16)v;
      }
      if (EABL & 1) asm volatile("" ::"v"(o));
      else *reinterpret_cast<h8 *>(p.out + oofs[gi] + nl + 32 * k) = o;
    }
  }
  }
}

// the implicit-GEMM schedules: output row m = m_base + mi*16 + (lane&15) in (image, oh, ow) raster order
template <int MI, int NI, int EABL = 0>
__device__ __forceinline__ void conv_epilogue(const ConvParams &p, f4 (&acc)[NI][MI], int m_base, int n_base, int lane) {
  const int ohw = p.OH * p.OW;
  const int grp = p.grp_rows ? m_base / p.grp_rows : 0;  // tile-uniform (grp_rows is a multiple of the tile height)
  conv_epilogue_px<MI, NI, EABL>(p, acc, n_base, lane, [&](int mi, int &img, int &oh, int &ow) {
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
template <int BN, int VAR = 0>
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
    xoff[i] = (unsigned)(((img * IHp + ih0) * IWp + iw0) * p.Cin + g * 8) * 2u;
  }
  unsigned woffv[WPIECES];
#pragma unroll
  for (int i = 0; i < WPIECES; i++) {
    int row = (wave * WPIECES + i) * 8 + srow;
    woffv[i] = (unsigned)((n0 + row) * p.Ktot + g * 8) * 2u;
  }
  const unsigned char *in_b = reinterpret_cast<const unsigned char *>(p.in);
  const unsigned char *w_b = reinterpret_cast<const unsigned char *>(p.w) + (p.grp_rows ? (size_t)(m0 / p.grp_rows) * p.grp_w_halfs * 2 : 0);

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
      h8 xf[2][4], wf[2][NREP];
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
#pragma unroll
        for (int mi = 0; mi < 4; mi++) xf[ks][mi] = *reinterpret_cast<const h8 *>(sb + xfo[ks] + mi * 16 * 128);
#pragma unroll
        for (int ni = 0; ni < NREP; ni++) wf[ks][ni] = *reinterpret_cast<const h8 *>(sb + wfo[ks] + ni * 16 * 128);
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
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int ni = 0; ni < NREP; ni++)
#pragma unroll
          for (int mi = 0; mi < 4; mi++)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][ni], xf[ks][mi], acc[ni][mi], 0, 0, 0);
      if (VAR & 1) __builtin_amdgcn_s_setprio(0);
      return;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      h8 xf[4], wf[NREP];
#pragma unroll
      for (int mi = 0; mi < 4; mi++) xf[mi] = *reinterpret_cast<const h8 *>(sb + xfo[ks] + mi * 16 * 128);
#pragma unroll
      for (int ni = 0; ni < NREP; ni++) wf[ni] = *reinterpret_cast<const h8 *>(sb + wfo[ks] + ni * 16 * 128);
      if (VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ni = 0; ni < NREP; ni++)
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
      if (VAR & 1) __builtin_amdgcn_s_setprio(0);
    }
  };

  const int k_begin = split * p.kt_per, k_end = min(p.Ktot >> 6, k_begin + p.kt_per);
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
  conv_epilogue<4, NREP>(p, acc, m0 + wm * 64, n0 + wn * (BN / 2), lane);
}

// -------------------------------------------------------------------------------------------------
// Large-problem variant: 256 pixels x BN channels per workgroup, 8 waves (4 along pixels x 2 along channels, the
// same 64 x BN/2 wave tile as above), THREE LDS stages and a prefetch distance of two K-steps.  The LDS-DMA loads are
// kept in flight across the barrier: per K-step each wave issues G = 4 + BN/64 global_load_lds, so
// `s_waitcnt vmcnt(G)` before the (raw) barrier retires exactly the tile about to be consumed and leaves the next
// tile's loads outstanding (cdna_hip_programming.md "Pipelining across barriers").  One barrier per K-step orders
// both hazards: RAW (every wave waited for its own pieces of tile kt) and WAR (every wave finished reading tile
// kt-1 before anyone overwrites its buffer with tile kt+2).
// -------------------------------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(512, 2) void conv_igemm3_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = 256;
  constexpr int XB = BM * 128;
  constexpr int WB = BN * 128;
  constexpr int STAGE = XB + WB;
  constexpr int NREP = BN / 32;
  constexpr int WPIECES = BN / 64;  // 8-row pieces of the W tile per wave (8 waves)
  constexpr int G = 4 + WPIECES;    // LDS-DMA instructions per wave per K-step

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  if (p.clk && tid == 0) { p.clk[blockIdx.x * 4] = __builtin_readcyclecounter(); p.clk[blockIdx.x * 4 + 1] = wall_clock64(); }
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
    xoff[i] = (unsigned)(((img * IHp + ih0) * IWp + iw0) * p.Cin + g * 8) * 2u;
  }
  unsigned woffv[WPIECES];
#pragma unroll
  for (int i = 0; i < WPIECES; i++) {
    int row = (wave * WPIECES + i) * 8 + srow;
    woffv[i] = (unsigned)((n0 + row) * p.Ktot + g * 8) * 2u;
  }
  const unsigned char *in_b = reinterpret_cast<const unsigned char *>(p.in);
  const unsigned char *w_b = reinterpret_cast<const unsigned char *>(p.w);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

  auto stage = [&](int kt, int buf) {
    const unsigned xs = lds_base + buf * STAGE;
    const unsigned ws = xs + XB;
    const unsigned char *xb = in_b + p.koff[kt];
    const unsigned char *wb = w_b + (size_t)kt * 128;
#pragma unroll
    for (int i = 0; i < 4; i++) glds16_asm(xb + xoff[i], xs + (wave * 4 + i) * 1024);
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
    xfo[ks] = (wm * 64 + frow) * 128 + slot * 16;
    wfo[ks] = XB + (wn * (BN / 2) + frow) * 128 + slot * 16;
  }

  auto compute = [&](int buf) {
    const unsigned char *sb = smem + buf * STAGE;
    h8 xf[2][4], wf[2][NREP];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
      for (int mi = 0; mi < 4; mi++) xf[ks][mi] = *reinterpret_cast<const h8 *>(sb + xfo[ks] + mi * 16 * 128);
#pragma unroll
      for (int ni = 0; ni < NREP; ni++) wf[ks][ni] = *reinterpret_cast<const h8 *>(sb + wfo[ks] + ni * 16 * 128);
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int ni = 0; ni < NREP; ni++)
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][ni], xf[ks][mi], acc[ni][mi], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  const int KT = p.Ktot >> 6;  // >= 8 for every layer of the networks
  stage(0, 0);
  stage(1, 1);
  int rb = 0, wb3 = 2;  // stage being read / written
  for (int kt = 0; kt < KT - 2; kt++) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");  // tile kt landed (this wave's pieces); tile kt+1 in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");  // s_barrier is IntrNoMem: keep LDS accesses on their side of it
    stage(kt + 2, wb3);
    compute(rb);
    rb = (rb == 2) ? 0 : rb + 1;
    wb3 = (wb3 == 2) ? 0 : wb3 + 1;
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");  // s_barrier is IntrNoMem: keep LDS accesses on their side of it
  compute(rb);
  rb = (rb == 2) ? 0 : rb + 1;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");  // s_barrier is IntrNoMem: keep LDS accesses on their side of it
  compute(rb);
  if (p.clk && tid == 0) { p.clk[blockIdx.x * 4 + 2] = __builtin_readcyclecounter(); p.clk[blockIdx.x * 4 + 3] = wall_clock64(); }

  conv_epilogue<4, NREP>(p, acc, m0 + wm * 64, n0 + wn * (BN / 2), lane);
}

// -------------------------------------------------------------------------------------------------
// Ping-pong variant: 256 pixels x BN channels per workgroup, 8 waves = two groups of 4 (each group owns 128 pixel
// rows with the usual 2x2 arrangement of 64 x BN/2 wave tiles, the W tile is shared).  The groups run in strict
// anti-phase: while group 0 pulls its 16 operand fragments of K-step kt from LDS into registers and issues the LDS-DMA
// for K-step kt+2, group 1 issues the 32 MFMAs of K-step kt-1 from registers only, and vice versa.  Each SIMD hosts one
// wave of either group, so its matrix pipe always has a pure-MFMA wave to run.  Three LDS stages; loads stay in
// flight across the barriers (counted vmcnt, raw s_barrier); two barriers per K-step.
// -------------------------------------------------------------------------------------------------
template <int BN>
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
    xoff[i] = (unsigned)(((img * IHp + ih0) * IWp + iw0) * p.Cin + g * 8) * 2u;
  }
  unsigned woffv[WPIECES];
#pragma unroll
  for (int i = 0; i < WPIECES; i++) {
    int row = (wave * WPIECES + i) * 8 + srow;
    woffv[i] = (unsigned)((n0 + row) * p.Ktot + g * 8) * 2u;
  }
  const unsigned char *in_b = reinterpret_cast<const unsigned char *>(p.in);
  const unsigned char *w_b = reinterpret_cast<const unsigned char *>(p.w);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

  auto stage = [&](int kt, int buf) {
    const unsigned xs = lds_base + buf * STAGE;
    const unsigned ws = xs + XB;
    const unsigned char *xb = in_b + p.koff[kt];
    const unsigned char *wb = w_b + (size_t)kt * 128;
#pragma unroll
    for (int i = 0; i < 4; i++) glds16_asm(xb + xoff[i], xs + (wave * 4 + i) * 1024);
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

  h8 xf[2][4], wf[2][NREP];
  auto ldfrags = [&](int buf) {
    const unsigned char *sb = smem + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
      for (int mi = 0; mi < 4; mi++) xf[ks][mi] = *reinterpret_cast<const h8 *>(sb + xfo[ks] + mi * 16 * 128);
#pragma unroll
      for (int ni = 0; ni < NREP; ni++) wf[ks][ni] = *reinterpret_cast<const h8 *>(sb + wfo[ks] + ni * 16 * 128);
    }
  };
  auto mfmas = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int ni = 0; ni < NREP; ni++)
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][ni], xf[ks][mi], acc[ni][mi], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
#define FP_PP_BARRIER()                  \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    asm volatile("" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

  const int KT = p.Ktot >> 6;  // >= 8
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

  conv_epilogue<4, NREP>(p, acc, m0 + grp * 128 + wm * 64, n0 + wn * (BN / 2), lane);
}

// -------------------------------------------------------------------------------------------------
// 256 x 256 tile.  Ablation of the 128 x 128 kernel (tools/bench_conv.py variants 17/21/25) shows its K-step is bound
// by the global -> LDS path, not by the matrix pipe: with the MFMAs removed a K-step still takes ~1360 cycles against
// ~1050 with the loads removed, i.e. the two co-resident workgroups pull 64 KB per K-step at ~47 B/clk/CU, the L2 ->
// CU ceiling.  A 256 x 256 tile moves the same 64 KB per K-step for TWICE the MFMA work, which puts the K-step back
// under the matrix pipe.  8 waves (2 along pixels x 4 along channels), wave tile 128 x 64 (32 accumulators), BK = 64,
// two 64-KB LDS stages, one workgroup per CU.  Needs Cout % 256 == 0.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void conv_big_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = 256, BN = 256;
  constexpr int XB = BM * 128, WB = BN * 128, STAGE = XB + WB;
  constexpr int MI = 8, NI = 4;

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
    xoff[i] = (unsigned)(((img * IHp + ih0) * IWp + iw0) * p.Cin + g * 8) * 2u;
    int row = (wave * 4 + i) * 8 + srow;
    woffv[i] = (unsigned)((n0 + row) * p.Ktot + g * 8) * 2u;
  }
  const unsigned char *in_b = reinterpret_cast<const unsigned char *>(p.in);
  const unsigned char *w_b = reinterpret_cast<const unsigned char *>(p.w);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

  auto stage = [&](int kt, int buf) {
    const unsigned xs = lds_base + buf * STAGE;
    const unsigned ws = xs + XB;
    const unsigned char *xb = in_b + p.koff[kt];
    const unsigned char *wb = w_b + (size_t)kt * 128;
#pragma unroll
    for (int i = 0; i < 4; i++) glds16_asm(xb + xoff[i], xs + (wave * 4 + i) * 1024);
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

  auto compute = [&](int buf) {
    const unsigned char *sb = smem + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      h8 xf[MI], wf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; mi++) xf[mi] = *reinterpret_cast<const h8 *>(sb + xfo[ks] + mi * 16 * 128);
#pragma unroll
      for (int ni = 0; ni < NI; ni++) wf[ni] = *reinterpret_cast<const h8 *>(sb + wfo[ks] + ni * 16 * 128);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  };

  const int KT = p.Ktot >> 6;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");  // s_barrier is IntrNoMem: keep LDS accesses on their side of it
  int buf = 0;
  for (int kt = 0; kt < KT - 1; kt++) {
    stage(kt + 1, buf ^ 1);
    compute(buf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");  // s_barrier is IntrNoMem: keep LDS accesses on their side of it
    buf ^= 1;
  }
  compute(buf);

  conv_epilogue<MI, NI>(p, acc, m0 + wm * 128, n0 + wn * 64, lane);
}

// Ping-pong schedule of the 256 x 256 tile (see the slot comment inside).
template <int ABL>
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
    xoff[i] = (unsigned)(((img * IHp + ih0) * IWp + iw0) * p.Cin + g * 8) * 2u;
    int row = (wave * 4 + i) * 8 + srow;
    woffv[i] = (unsigned)((n0 + row) * p.Ktot + g * 8) * 2u;
  }
  const unsigned char *in_b = reinterpret_cast<const unsigned char *>(p.in);
  const unsigned char *w_b = reinterpret_cast<const unsigned char *>(p.w);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

  // the 8 LDS-DMA instructions of a K-step are issued in two halves (X pieces in the wave's L0 slot, W pieces at the
  // head of its M0 slot) so the address path (64 B/clk/CU) sees them spread over three slots instead of bunched in two
  auto stage_x = [&](int kt, int buf) {
    const unsigned xs = lds_base + buf * STAGE;
    const unsigned char *xb = in_b + p.koff[kt];
#pragma unroll
    for (int i = 0; i < 4; i++) glds16_asm(xb + xoff[i], xs + (wave * 4 + i) * 1024);
  };
  auto stage_w = [&](int kt, int buf) {
    const unsigned ws = lds_base + buf * STAGE + XB;
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

  h8 xf[MI], wf[NI];
  auto ld = [&](int buf, int ks) {
    const unsigned char *sb = smem + buf * STAGE;
#pragma unroll
    for (int mi = 0; mi < MI; mi++) xf[mi] = *reinterpret_cast<const h8 *>(sb + xfo[ks] + mi * 16 * 128);
#pragma unroll
    for (int ni = 0; ni < NI; ni++) wf[ni] = *reinterpret_cast<const h8 *>(sb + wfo[ks] + ni * 16 * 128);
  };
  auto mfmas = [&]() {
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
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
#define FP_BAR()                         \
  do {                                   \
    __builtin_amdgcn_s_barrier();        \
    asm volatile("" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
#define FP_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define FP_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

  // Slots of ~32 MFMAs: group 0 runs L0 M0 L1 M1 per K-step, group 1 the same one slot later, so on every SIMD one
  // wave issues MFMAs from registers while its partner refills fragments from LDS / issues the next tile's LDS-DMA.
  const int KT = p.Ktot >> 6;
  stage_x(0, 0);
  stage_w(0, 0);
  FP_VM0();
  FP_BAR();
  if (p.clk && tid == 0) { p.clk[blockIdx.x * 4] = __builtin_readcyclecounter(); p.clk[blockIdx.x * 4 + 1] = wall_clock64(); }
  int buf = 0;
  if (wm == 0) {
    for (int kt = 0; kt < KT; kt++) {
      ld(buf, 0);
      if (kt + 1 < KT && !(ABL & 1)) stage_x(kt + 1, buf ^ 1);
      FP_LGKM0(); FP_BAR();
      if (kt + 1 < KT && !(ABL & 1)) stage_w(kt + 1, buf ^ 1);
      mfmas(); FP_BAR();
      ld(buf, 1); FP_LGKM0(); FP_BAR();
      mfmas(); FP_VM0(); FP_BAR();
      buf ^= 1;
    }
    FP_BAR();
  } else {
    FP_BAR();
    for (int kt = 0; kt < KT; kt++) {
      ld(buf, 0);
      if (kt + 1 < KT && !(ABL & 1)) stage_x(kt + 1, buf ^ 1);
      FP_LGKM0(); FP_BAR();
      if (kt + 1 < KT && !(ABL & 1)) stage_w(kt + 1, buf ^ 1);
      mfmas(); FP_BAR();
      ld(buf, 1); FP_LGKM0(); FP_VM0(); FP_BAR();
      mfmas(); FP_BAR();
      buf ^= 1;
    }
  }
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
  conv_epilogue<MI, NI, (ABL >> 3) & 3>(p, acc, m0 + wm * 128, n0 + wn * 64, lane);
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
template <int BM, int BN>
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
      poff[i] = (unsigned)(((img * IHp + ih0) * IWp + iw0) * p.Cin + gch * 8) * 2u;
    } else {
      int row = (piece - XP) * 16 + prow;
      poff[i] = (unsigned)((n0 + row) * p.Ktot + gch * 8) * 2u;
    }
  }
  const unsigned char *in_b = reinterpret_cast<const unsigned char *>(p.in);
  const unsigned char *w_b = reinterpret_cast<const unsigned char *>(p.w);
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

  h8 xf[MI], wf[NI];
  auto ld = [&](int buf) {
    const unsigned char *sb = smem + buf * STAGE;
#pragma unroll
    for (int mi = 0; mi < MI; mi++) xf[mi] = *reinterpret_cast<const h8 *>(sb + xfo + mi * 16 * 64);
#pragma unroll
    for (int ni = 0; ni < NI; ni++) wf[ni] = *reinterpret_cast<const h8 *>(sb + wfo + ni * 16 * 64);
  };
  auto mfmas = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int mi = 0; mi < MI; mi++)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
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

  const int S = p.Ktot >> 5;  // 32-wide K-steps, >= 16 for every layer
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
  conv_epilogue<MI, NI>(p, acc, m0 + wm * 128, n0 + wn * 64, lane);
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
template <int TW, int ABL = 0>  // ABL (timing ablations, wrong results): 1 no per-step barrier, 2 no MFMAs, 4 X fragments read once
__global__ __launch_bounds__(256, 2) void conv_halo_kernel(const ConvParams p) {
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

  const unsigned char *in_b = reinterpret_cast<const unsigned char *>(p.in) + ((size_t)(img * IHp + ty0) * HC) * p.Cin * 2;
  const unsigned char *w_b = reinterpret_cast<const unsigned char *>(p.w);
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
        unsigned off = (unsigned)(q * p.Cin * 2 + (((lane & 7) ^ g) << 4));
        glds16_asm(src + off, lds_base + piece * 1024);
        __builtin_amdgcn_sched_barrier(0);  // one address at a time: 14 hoisted 64-bit addresses would spill accumulators
      }
    }
  };
  // weight DMA: piece = wave*2 + i (16 rows x 64 B); lane -> (row = lane>>2, slot = lane&3), source chunk = slot ^ G
  const int prow = lane >> 2;
  const int gch = (lane & 3) ^ ((0x78 >> (((prow >> 2) & 3) * 2)) & 3);
  unsigned woff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) woff[i] = (unsigned)((n0 + (wave * 2 + i) * 16 + prow) * p.Ktot + gch * 8) * 2u;
  auto issue_w = [&](int st) {
    const unsigned char *wb = w_b + (size_t)st * 64;
    const unsigned dst = w_lds + (st % NWST) * WST;
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

  const int S = p.Ktot >> 5;   // 32-wide K-steps: 18 per 64-channel chunk (9 taps x 2)
  const int nch = p.Cin >> 6;
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
        if ((tap == 0 && ks == 0) || s == S - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (!(ABL & 1) || (tap == 0 && ks == 0)) __builtin_amdgcn_s_barrier();
        // s_barrier is IntrNoMem for the compiler: without this fence the fragment loads below may be placed ABOVE the
        // barrier on the steps that issue no DMA (the last two), reading weight pieces other waves have not landed yet
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < S) issue_w(s + 2);
        const unsigned char *xs = smem + pix_off + (((ks * 4 + kg) ^ g) << 4);
        const unsigned char *ws = smem + wfo + (s % NWST) * WST;
        // X fragments in two halves of MI/2 (register budget: 160 accumulators + 20 + 16 fragment registers); the
        // second half's LDS reads are issued behind the first half's MFMAs
        constexpr int HM = MI / 2;
        h8 xf[HM], wf[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ni++) wf[ni] = *reinterpret_cast<const h8 *>(ws + ni * 16 * 64);
#pragma unroll
        for (int mi = 0; mi < HM; mi++) xf[mi] = *reinterpret_cast<const h8 *>(xs + mi * 512);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(ABL & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
          for (int mi = 0; mi < HM; mi++) {
            if (ABL & 2) asm volatile("" ::"v"(wf[ni]), "v"(xf[mi]));
            else acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
          }
        if (!(ABL & 8)) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < HM; mi++) xf[mi] = *reinterpret_cast<const h8 *>(xs + (HM + mi) * 512);
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
            else acc[ni][HM + mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][HM + mi], 0, 0, 0);
          }
        if (!(ABL & 8)) __builtin_amdgcn_s_setprio(0);
      }
    }
  }
  if (p.clk && tid == 0) { p.clk[blockIdx.x * 4 + 2] = __builtin_readcyclecounter(); p.clk[blockIdx.x * 4 + 3] = wall_clock64(); }
  conv_epilogue_px<MI, NI>(p, acc, n0 + wn * 64, lane, [&](int mi, int &oimg, int &oh, int &ow) {
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

  const unsigned char *in_b = reinterpret_cast<const unsigned char *>(p.in) + ((size_t)(img * HC + ty0) * HC) * 64;
  const unsigned char *w_b = reinterpret_cast<const unsigned char *>(p.w);
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
  const unsigned woff = (unsigned)((wave * 16 + prow) * p.Ktot + gch * 8) * 2u;
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
    h8 xf[HM], wf[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) wf[ni] = *reinterpret_cast<const h8 *>(ws + ni * 16 * 64);
#pragma unroll
    for (int mi = 0; mi < HM; mi++) xf[mi] = *reinterpret_cast<const h8 *>(xs + mi * 256);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int mi = 0; mi < HM; mi++)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < HM; mi++) xf[mi] = *reinterpret_cast<const h8 *>(xs + (HM + mi) * 256);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int mi = 0; mi < HM; mi++)
        acc[ni][HM + mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][HM + mi], 0, 0, 0);
  }
  // Cout = 64: weight rows are permuted in 32-channel blocks (NI = 2 form), so the epilogue runs once per block
  auto pix = [&](int mi, int &oimg, int &oh, int &ow) {
    oimg = img;
    oh = ty0 + brow * 4 + dy;
    ow = chalf * (TW / 2) + mi * 4 + dx;
    return true;
  };
  conv_epilogue_px<MI, 2>(p, reinterpret_cast<f4(&)[2][MI]>(acc[0]), 0, lane, pix);
  conv_epilogue_px<MI, 2>(p, reinterpret_cast<f4(&)[2][MI]>(acc[2]), 32, lane, pix);
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
  const unsigned char *in_b = reinterpret_cast<const unsigned char *>(p.in) + ((size_t)(img * IC + 2 * oy0) * IC) * 128;
  const unsigned char *w_b = reinterpret_cast<const unsigned char *>(p.w);
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
  for (int i = 0; i < 2; i++) woff[i] = (unsigned)(((wave * 2 + i) * 16 + prow) * p.Ktot + gch * 8) * 2u;
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
    h8 xf[MI], wf[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) wf[ni] = *reinterpret_cast<const h8 *>(ws + ni * 16 * 64);
#pragma unroll
    for (int mi = 0; mi < MI; mi++) xf[mi] = *reinterpret_cast<const h8 *>(xs + mi * 512);
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
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
  }
  conv_epilogue_px<MI, NI>(p, acc, wn * 64, lane, [&](int mi, int &oimg, int &oh, int &ow) {
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
template <int ABL = 0>  // timing ablations: 1 = no MFMAs, 2 = no loads after the first stage, 4 = no stores
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
  const int S = p.Ktot >> 5;

  const unsigned char *in_b = reinterpret_cast<const unsigned char *>(p.in);
  const unsigned char *w_b = reinterpret_cast<const unsigned char *>(p.w);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

  // staging: a wave-instruction moves 16 rows x 64 B; lane -> (row = lane>>2, slot = lane&3), source chunk swizzled
  const int prow = lane >> 2;
  const int gch = (lane & 3) ^ ((0x78 >> (((prow >> 2) & 3) * 2)) & 3);
  unsigned xoff[2], woff[4];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int m = min(m0 + (wave * 2 + i) * 16 + prow, p.M - 1);  // rows past M re-read the last row (never stored)
    xoff[i] = (unsigned)(m * p.Ktot + gch * 8) * 2u;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) woff[i] = (unsigned)((n0 + (wave * 4 + i) * 16 + prow) * p.Ktot + gch * 8) * 2u;
  auto issue = [&](int st) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (st % NST) * STAGE);
    const unsigned char *xb = in_b + (size_t)st * 64, *wb = w_b + (size_t)st * 64;
#pragma unroll
    for (int i = 0; i < 2; i++) glds16_asm(xb + xoff[i], dst + (wave * 2 + i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; i++) glds16_asm(wb + woff[i], dst + XST + (wave * 4 + i) * 1024);
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
    h8 xf[4], wf[8];
#pragma unroll
    for (int mi = 0; mi < 4; mi++) xf[mi] = *reinterpret_cast<const h8 *>(sb + xfo + mi * 1024);
#pragma unroll
    for (int ni = 0; ni < 8; ni++) wf[ni] = *reinterpret_cast<const h8 *>(sb + wfo + ni * 1024);
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
        acc[ni >> 2][ni & 3][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni >> 2][ni & 3][mi], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  }
  conv_epilogue<4, 4, (ABL >> 2) & 1>(p, acc[0], m0 + wm * 64, n0 + wn * 128, lane);
  conv_epilogue<4, 4, (ABL >> 2) & 1>(p, acc[1], m0 + wm * 64, n0 + wn * 128 + 64, lane);
}

// -------------------------------------------------------------------------------------------------
// conv_deep_kernel: the rows the full 256x256 rounds of a long-K layer leave over (2.5 % of conv_512 at N = 252) run on
// an otherwise idle chip, one workgroup per CU walking all 72 K-steps: that chain is latency-bound, not bandwidth- or
// MFMA-bound.  So: small tiles (BM x 128, more CUs in use), the whole LDS as a deep LDS-DMA ring (6 x 24 KB stages at
// BM = 64: five K-steps in flight, counted vmcnt, one barrier per step) and the next step's fragments read under the
// current step's MFMAs (two register sets).  Same K order / accumulation order as every
// other schedule, so a row's value does not depend on which kernel computed it.
// -------------------------------------------------------------------------------------------------
template <int BM>
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
  const int mt = logical / n_tiles, nt = logical - mt * n_tiles;
  const int m0 = p.m_begin + mt * BM, n0 = nt * BN;
  const int ohw = p.OH * p.OW;
  const int IHp = p.H + 2 * p.ipad, IWp = p.W + 2 * p.ipad;
  const int KT = p.Ktot >> 6;

  const int srow = lane >> 3;
  const int g = (lane & 7) ^ srow;
  unsigned xoff[XP], woff[WP];
#pragma unroll
  for (int i = 0; i < XP; i++) {
    int m = min(m0 + (wave * XP + i) * 8 + srow, p.M - 1);  // rows past M re-read the last pixel (never stored)
    int img = m / ohw;
    int rem = m - img * ohw;
    int oh = rem / p.OW, ow = rem - oh * p.OW;
    int ih0 = oh * p.stride - p.pad + p.ipad, iw0 = ow * p.stride - p.pad + p.ipad;
    xoff[i] = (unsigned)(((img * IHp + ih0) * IWp + iw0) * p.Cin + g * 8) * 2u;
  }
#pragma unroll
  for (int i = 0; i < WP; i++) woff[i] = (unsigned)((n0 + (wave * WP + i) * 8 + srow) * p.Ktot + g * 8) * 2u;
  const unsigned char *in_b = reinterpret_cast<const unsigned char *>(p.in);
  const unsigned char *w_b = reinterpret_cast<const unsigned char *>(p.w);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
  auto issue = [&](int kt) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (kt % NST) * STAGE);
    const unsigned char *xb = in_b + p.koff[kt];
    const unsigned char *wb = w_b + (size_t)kt * 128;
#pragma unroll
    for (int i = 0; i < XP; i++) glds16_asm(xb + xoff[i], dst + (wave * XP + i) * 1024);
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
  struct Frags { h8 x[2][MI], w[2][4]; };
  auto read_frags = [&](int kt, Frags &f) {
    const unsigned char *sb = smem + (kt % NST) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
      for (int mi = 0; mi < MI; mi++) f.x[ks][mi] = *reinterpret_cast<const h8 *>(sb + xfo[ks] + mi * 16 * 128);
#pragma unroll
      for (int ni = 0; ni < 4; ni++) f.w[ks][ni] = *reinterpret_cast<const h8 *>(sb + wfo[ks] + ni * 16 * 128);
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
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int ni = 0; ni < 4; ni++)
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.w[ks][ni], cur.x[ks][mi], acc[ni][mi], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

#pragma unroll
  for (int s = 0; s < D; s++)
    if (s < KT) issue(s);
  Frags fa, fb;
  wait_stage(0, D - 1);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(0, fa);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int kt = 0;
#pragma unroll 1
  for (; kt + 1 < KT; kt += 2) {
    step(kt, fa, fb);
    step(kt + 1, fb, fa);
  }
  if (kt < KT) step(kt, fa, fb);
  conv_epilogue<MI, 4>(p, acc, m0 + wm * (BM / 2), n0 + wn * 64, lane);
}

// split-K reduction + the conv epilogue: out = relu(sum_s partial[s] + bias + res); thread = (pixel, 4 channels)
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvParams p) {
  const int nq = p.Cout / 4;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int rows = p.M - p.m_begin;
  if (i >= (size_t)rows * nq) return;
  const int mr = (int)(i / nq), n = (int)(i - (size_t)mr * nq) * 4;
  const int m = p.m_begin + mr;
  f4 a = *reinterpret_cast<const f4 *>(p.partial + (size_t)mr * p.Cout + n);
  for (int sp = 1; sp < p.ksplit; sp++) {  // component-wise: no packed-f32 VALU ops in this library ("packed f32")
    const f4 b = *reinterpret_cast<const f4 *>(p.partial + ((size_t)sp * rows + mr) * p.Cout + n);
    a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
  }
  const int grp = p.grp_rows ? m / p.grp_rows : 0;
  float4 bv = *reinterpret_cast<const float4 *>(p.bias + grp * p.Cout + n);
  float v0 = a[0] + bv.x, v1 = a[1] + bv.y, v2 = a[2] + bv.z, v3 = a[3] + bv.w;
  const int ohw = p.OH * p.OW;
  int img = m / ohw;
  int rem = m - img * ohw;
  int oh = rem / p.OW, ow = rem - oh * p.OW;
  const int OHp = p.OH + 2 * p.opad, OWp = p.OW + 2 * p.opad;
  const int RHp = p.OH + 2 * p.rpad, RWp = p.OW + 2 * p.rpad;
  if (p.res) {
    size_t rpix = ((size_t)(img - (p.res_shared ? grp * p.grp_rows : 0)) * RHp + oh + p.rpad) * RWp + ow + p.rpad;
    h4 r = *reinterpret_cast<const h4 *>(p.res + rpix * p.res_ld + n);
    v0 += (float)r[0]; v1 += (float)r[1]; v2 += (float)r[2]; v3 += (float)r[3];
  }
  if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
  int choff = 0, oimg = img;
  if (p.split_imgs > 0 && img >= p.split_imgs) { oimg = img - p.split_imgs; choff = p.Cout; }
  size_t opix = ((size_t)oimg * OHp + oh + p.opad) * OWp + ow + p.opad;
  h4 o = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
  *reinterpret_cast<h4 *>(p.out + opix * p.out_ld + choff + n) = o;
}

// =================================================================================================
// attention: out[b,t,h*128+d] = softmax_k(q.k/sqrt(128)) v,  qkv = [B,T,1536] (q|k|v, heads contiguous inside each)
// =================================================================================================

// (a 64-key-block variant of this kernel measured 6 % slower inside Register and was dropped)
// ATT_QROWS query rows per workgroup (64 = 4 waves, one per SIMD; 80-row / 5-wave tiles cover 400 tokens exactly but
// measured 12 % slower: two waves of a workgroup share a SIMD).  The 1-D grid is remapped so the query
// tiles of one (image, head) run on the SAME XCD and share its L2 copy of K/V (a (qt,h,b) grid spread them over all 8
// XCDs: rocprofv3 FETCH_SIZE showed 1.16 GB fetched per launch for 0.31 GB of QKV).
template <int ATT_QROWS, bool REMAP, bool PERM = true>
__global__ __launch_bounds__(ATT_QROWS * 4, 2) void attention_kernel(const __half *__restrict__ qkv, __half *__restrict__ out, int T, int nq,
                                                                 int tstride /* rows between the first tokens of consecutive sequences */) {
  constexpr int KS = 136;  // K tile row stride (halfs): 128 + 8 pad
  constexpr int VS = 40;   // V^T tile row stride (halfs): 32 keys + 8 pad
  __shared__ __attribute__((aligned(16))) _Float16 Ks[32 * KS];
  __shared__ __attribute__((aligned(16))) _Float16 Vt[HDIM * VS];
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
  const __half *base = qkv + (size_t)b * tstride * rowstride;
  const int q_row = qt * ATT_QROWS + wave * 16 + li;
  const int q_ld = min(q_row, T - 1);
  h8 qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ds++)
    qf[ds] = *reinterpret_cast<const h8 *>(base + (size_t)q_ld * rowstride + h * HDIM + ds * 32 + g * 8);

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
  h8 kreg[2], vreg[2];
  auto load_tile = [&](int kb) {
    if (ATT_QROWS > 64 && tid >= 256) return;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      int idx = tid + j * 256;
      int krow = min(kb * 32 + (idx >> 4), T - 1);
      kreg[j] = *reinterpret_cast<const h8 *>(base + (size_t)krow * rowstride + EMBED + h * HDIM + (idx & 15) * 8);
      int vrow = min(kb * 32 + (idx & 31), T - 1);
      vreg[j] = *reinterpret_cast<const h8 *>(base + (size_t)vrow * rowstride + 2 * EMBED + h * HDIM + (idx >> 5) * 8);
    }
  };
  load_tile(0);
  for (int kb = 0; kb < nkb; kb++) {
    if (ATT_QROWS <= 64 || tid < 256) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        int idx = tid + j * 256;
        *reinterpret_cast<h8 *>(&Ks[(idx >> 4) * KS + (idx & 15) * 8]) = kreg[j];
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
        h8 kf = *reinterpret_cast<const h8 *>(&Ks[(kt * 16 + li) * KS + ds * 32 + g * 8]);
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ds], st[kt], 0, 0, 0);
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
    h8 pf;
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r], sl2e, -mc));
        psum += pv;
        pf[kt * 4 + r] = (_Float16)pv;
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
      h4 v0 = *reinterpret_cast<const h4 *>(&Vt[(dt * 16 + li) * VS + g * 4]);
      h4 v1 = *reinterpret_cast<const h4 *>(&Vt[(dt * 16 + li) * VS + 16 + g * 4]);
      h8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf, vf, o[dt], 0, 0, 0);
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
      h8 ov;
#pragma unroll
      for (int dt = 0; dt < 8; dt++) ov[dt] = (_Float16)(o[dt][r] * lr[r]);
      *reinterpret_cast<h8 *>(out + ((size_t)b * tstride + row) * EMBED + h * HDIM + li * 8) = ov;
    } else {
      __half *dst = out + ((size_t)b * tstride + row) * EMBED + h * HDIM + li;
#pragma unroll
      for (int dt = 0; dt < 8; dt++) dst[dt * 16] = __float2half(o[dt][r] * lr[r]);
    }
  }
}

// =================================================================================================
// small kernels
// =================================================================================================

// x[b,t,:] += pe[t,:]  (rows = B*T, 512 channels, 8 halfs per thread)
__global__ void add_pos_embed_kernel(__half *__restrict__ x, const __half *__restrict__ pe, int T, size_t rows) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // 16-B chunk index
  if (i >= rows * (EMBED / 8)) return;
  size_t row = i / (EMBED / 8);
  int c = (int)(i - row * (EMBED / 8));
  int t = (int)(row % T);
  h8 a = reinterpret_cast<const h8 *>(x)[i];
  h8 pv = reinterpret_cast<const h8 *>(pe)[(size_t)t * (EMBED / 8) + c];
  reinterpret_cast<h8 *>(x)[i] = a + pv;
}

// y = LayerNorm(x) over 512 channels, eps 1e-5; one wave per row
// rows >= split_row use (gamma1, beta1): the refiner's two heads normalised in one launch
__global__ __launch_bounds__(256) void layernorm_kernel(const __half *__restrict__ x, const float *__restrict__ gamma0,
                                                        const float *__restrict__ beta0, __half *__restrict__ y, size_t rows,
                                                        const float *__restrict__ gamma1, const float *__restrict__ beta1,
                                                        size_t split_row) {
  size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float *gamma = row >= split_row ? gamma1 : gamma0, *beta = row >= split_row ? beta1 : beta0;
  h8 v = reinterpret_cast<const h8 *>(x + row * EMBED)[lane];
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
  h8 r;
#pragma unroll
  for (int e = 0; e < 8; e++) r[e] = (_Float16)(f[e] * rstd * gamma[lane * 8 + e] + beta[lane * 8 + e]);
  reinterpret_cast<h8 *>(y + row * EMBED)[lane] = r;
}

// out[b,c] = mean_t x[b,t,c]  (f32 out).  Deterministic (no atomics: the arg-max over near-tied scores must not depend
// on summation order).  block = (b, 64-channel group); a lane loads 8 channels (16 B) of one token, so a wave covers 8
// tokens per load and walks the sequence in strides of 32 tokens (13 dependent steps for T = 400 instead of 100: at
// N = 1 this kernel was 30 us of a 570 us Track); the 8 token slots combine through shfl_xor, the 4 waves through LDS,
// both in a fixed order.
__global__ __launch_bounds__(256) void token_mean_kernel(const __half *__restrict__ x, float *__restrict__ out, int T, int tstride) {
  __shared__ float part[4][64];
  const int b = blockIdx.x, cg = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane >> 3, c8 = (lane & 7) * 8;
  const __half *src = x + (size_t)b * tstride * EMBED + cg * 64 + c8;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; e++) s[e] = 0.f;
  for (int t = wave * 8 + slot; t < T; t += 32) {
    h8 v = *reinterpret_cast<const h8 *>(src + (size_t)t * EMBED);
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

// y[b,o] = bias[o] + sum_c x[b,c] W[o,c]   (f32; one wave per output)
__global__ __launch_bounds__(256) void small_linear_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                                           const float *__restrict__ bias, float *__restrict__ y, int B,
                                                           int O, int C) {
  size_t widx = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (widx >= (size_t)B * O) return;
  int b = (int)(widx / O), o = (int)(widx - (size_t)b * O);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += x[(size_t)b * C + c] * W[(size_t)o * C + c];
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k);
  if (lane == 0) y[widx] = s + bias[o];
}

// cat[i][:, :, C:2C] = cat[0][:, :, C:2C] for i in 1..N-1 (bordered [N,HP,WP,2C] tensor, interior pixels only).
// Used when every hypothesis shares one observed crop (Register's first refine iteration: the sampler gives all 252
// poses the same translation, foundationpose_sampling.cpp:388-391, so transf_input is identical for all of them).
__global__ void broadcast_b_kernel(__half *__restrict__ cat, int N, int HP, int WP, int H, int W, int pad, int C) {
  const int chunks = C / 8;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t per_img = (size_t)H * W * chunks;
  if (i >= per_img * (size_t)(N - 1)) return;
  int img = 1 + (int)(i / per_img);
  size_t r = i - (size_t)(img - 1) * per_img;
  int pix = (int)(r / chunks), ch = (int)(r - (size_t)pix * chunks);
  int y = pix / W, x = pix - y * W;
  size_t off = (((size_t)(y + pad)) * WP + (x + pad)) * (2 * C) + C + ch * 8;
  const size_t img_stride = (size_t)HP * WP * 2 * C;
  *reinterpret_cast<h8 *>(cat + (size_t)img * img_stride + off) = *reinterpret_cast<const h8 *>(cat + off);
}

__global__ void cast_f32_f16_kernel(const float *__restrict__ in, __half *__restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = __float2half(in[i]);
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
  bool ok = std::fread(magic, 1, 4, f) == 4 && std::memcmp(magic, "FPW1", 4) == 0 && std::fread(&n, 4, 1, f) == 1;
  for (uint32_t i = 0; ok && i < n; i++) {
    uint32_t ln = 0, nd = 0;
    ok = std::fread(&ln, 4, 1, f) == 1 && ln < 4096;
    std::string name(ln, '\0');
    ok = ok && std::fread(&name[0], 1, ln, f) == ln && std::fread(&nd, 4, 1, f) == 1 && nd <= 8;
    HostTensor t;
    size_t cnt = 1;
    for (uint32_t d = 0; ok && d < nd; d++) {
      uint32_t s = 0;
      ok = std::fread(&s, 4, 1, f) == 1;
      t.shape.push_back((int)s);
      cnt *= s;
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
  __half *w = nullptr;
  float *bias = nullptr;
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

struct Net {
  bool scorer = false;
  ConvLayer a0, a1, ra[2][2];        // encodeA
  ConvLayer rb[2][2], b2, rc[2][2];  // encodeAB
  EncLayer trans, rot;               // refiner
  // the two heads' Linear layers stored back to back ([2][Cout][K], [2][Cout]) for the one-launch small-batch path
  ConvLayer g_in_proj, g_out_proj, g_lin1, g_lin2;
  MHA att, att_cross;                // scorer
  LinearF32 score_lin;
  __half *pe = nullptr;     // [400,512]
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
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

// [a | b] device copy of two equally shaped Linear layers (weights already in kernel row order)
static bool make_grouped(Net *net, const ConvLayer &a, const ConvLayer &b, ConvLayer *g) {
  if (a.Cin != b.Cin || a.Cout != b.Cout) return false;
  const size_t nw = (size_t)a.Cout * a.Cin, nb = (size_t)a.Cout;
  __half *w = nullptr;
  float *bias = nullptr;
  if (hipMalloc((void **)&w, 2 * nw * sizeof(__half)) != hipSuccess) return false;
  net->allocs.push_back(w);
  if (hipMalloc((void **)&bias, 2 * nb * sizeof(float)) != hipSuccess) return false;
  net->allocs.push_back(bias);
  if (hipMemcpy(w, a.w, nw * 2, hipMemcpyDeviceToDevice) != hipSuccess || hipMemcpy(w + nw, b.w, nw * 2, hipMemcpyDeviceToDevice) != hipSuccess ||
      hipMemcpy(bias, a.bias, nb * 4, hipMemcpyDeviceToDevice) != hipSuccess || hipMemcpy(bias + nb, b.bias, nb * 4, hipMemcpyDeviceToDevice) != hipSuccess)
    return false;
  *g = a;
  g->w = w;
  g->bias = bias;
  return true;
}

static bool get(const std::map<std::string, HostTensor> &m, const std::string &name, const HostTensor **t, std::string *err) {
  auto it = m.find(name);
  if (it == m.end()) { *err = "missing tensor " + name; return false; }
  *t = &it->second;
  return true;
}

// [Cout][tap][Cin] -> kernel K order: for Cin >= 64 [Cout][Cin/64][tap][64], otherwise unchanged
static std::vector<__half> relayout_k(const std::vector<__half> &w, int Cout, int ntaps, int Cin) {
  if (Cin < 64 || ntaps == 1) return w;
  std::vector<__half> o(w.size());
  const int nch = Cin / 64;
  for (int co = 0; co < Cout; co++)
    for (int t = 0; t < ntaps; t++)
      for (int ci = 0; ci < Cin; ci++)
        o[(((size_t)co * nch + ci / 64) * ntaps + t) * 64 + (ci % 64)] = w[((size_t)co * ntaps + t) * Cin + ci];
  return o;
}

// Row permutation matching conv_epilogue: inside every block of 16*NI output channels (NI = 4, or 2 when Cout == 64)
// kernel row ni*16 + 4g + j holds the weights of channel 32*(j>>1) + 8g + 4*(j&1) + ni (NI=4) / 8g + 2j + ni (NI=2).
static std::vector<__half> permute_rows(const std::vector<__half> &w, int Cout, int K) {
  const int NI = (Cout % 128 == 0) ? 4 : 2, blk = 16 * NI;
  std::vector<__half> o(w.size());
  for (int b0 = 0; b0 < Cout; b0 += blk)
    for (int ni = 0; ni < NI; ni++)
      for (int g = 0; g < 4; g++)
        for (int j = 0; j < 4; j++) {
          int ch = (NI == 4) ? 32 * (j >> 1) + 8 * g + 4 * (j & 1) + ni : 8 * g + 2 * j + ni;
          std::memcpy(&o[(size_t)(b0 + ni * 16 + 4 * g + j) * K], &w[(size_t)(b0 + ch) * K], (size_t)K * sizeof(__half));
        }
  return o;
}

// PyTorch conv weight [Cout,Cin,KH,KW] -> [Cout][KH][KW][Cin] fp16 -> kernel K order
static bool make_conv(Net *net, const std::map<std::string, HostTensor> &m, const std::string &prefix, int stride,
                      ConvLayer *L, std::string *err) {
  const HostTensor *w, *b;
  if (!get(m, prefix + ".weight", &w, err) || !get(m, prefix + ".bias", &b, err)) return false;
  if (w->shape.size() != 4) { *err = prefix + ": expected 4-d conv weight"; return false; }
  int Co = w->shape[0], Ci = w->shape[1], KH = w->shape[2], KW = w->shape[3];
  std::vector<__half> hw((size_t)Co * KH * KW * Ci);
  for (int co = 0; co < Co; co++)
    for (int ci = 0; ci < Ci; ci++)
      for (int kh = 0; kh < KH; kh++)
        for (int kw = 0; kw < KW; kw++)
          hw[(((size_t)co * KH + kh) * KW + kw) * Ci + ci] = __float2half(w->data[(((size_t)co * Ci + ci) * KH + kh) * KW + kw]);
  L->w = upload(net, permute_rows(relayout_k(hw, Co, KH * KW, Ci), Co, KH * KW * Ci));
  L->bias = upload(net, b->data);
  L->Cin = Ci; L->Cout = Co; L->KH = KH; L->KW = KW; L->stride = stride; L->pad = (KH - 1) / 2;
  return L->w && L->bias;
}

// 7x7 stride-2 pad-3 stem on [.,160,160,6] == 4x4 stride-1 pad-2 conv on the space-to-depth input [.,80,80,32]:
// w_s2d[co][a][b][(dy*2+dx)*8 + c] = w[co][c][2a+dy-1][2b+dx-1] (zero outside the 7x7 support / for c >= 6)
static bool make_stem(Net *net, const std::map<std::string, HostTensor> &m, const std::string &prefix, ConvLayer *L,
                      std::string *err) {
  const HostTensor *w, *b;
  if (!get(m, prefix + ".weight", &w, err) || !get(m, prefix + ".bias", &b, err)) return false;
  if (w->shape != std::vector<int>({64, 6, 7, 7})) { *err = prefix + ": expected [64,6,7,7] stem weight"; return false; }
  std::vector<__half> hw((size_t)64 * 16 * 32, __float2half(0.f));
  for (int co = 0; co < 64; co++)
    for (int a = 0; a < 4; a++)
      for (int bb = 0; bb < 4; bb++)
        for (int dy = 0; dy < 2; dy++)
          for (int dx = 0; dx < 2; dx++) {
            int kh = 2 * a + dy - 1, kw = 2 * bb + dx - 1;
            if (kh < 0 || kw < 0) continue;
            for (int c = 0; c < 6; c++)
              hw[(((size_t)co * 4 + a) * 4 + bb) * 32 + (dy * 2 + dx) * 8 + c] =
                  __float2half(w->data[(((size_t)co * 6 + c) * 7 + kh) * 7 + kw]);
          }
  L->w = upload(net, permute_rows(hw, 64, 16 * 32));
  L->bias = upload(net, b->data);
  L->Cin = 32; L->Cout = 64; L->KH = 4; L->KW = 4; L->stride = 1; L->pad = 2; L->algo_K = 7 * 7 * 6;
  return L->w && L->bias;
}

// Linear [out,in] as a 1x1 conv; rows [r0, r0+rows) of the weight
static bool make_linear_conv(Net *net, const std::map<std::string, HostTensor> &m, const std::string &wname,
                             const std::string &bname, ConvLayer *L, std::string *err) {
  const HostTensor *w, *b;
  if (!get(m, wname, &w, err) || !get(m, bname, &b, err)) return false;
  if (w->shape.size() != 2) { *err = wname + ": expected 2-d weight"; return false; }
  std::vector<__half> hw(w->data.size());
  for (size_t i = 0; i < hw.size(); i++) hw[i] = __float2half(w->data[i]);
  L->w = upload(net, permute_rows(hw, w->shape[0], w->shape[1]));
  L->bias = upload(net, b->data);
  L->Cout = w->shape[0]; L->Cin = w->shape[1]; L->KH = L->KW = 1; L->stride = 1; L->pad = 0;
  return L->w && L->bias;
}

static bool make_linear_f32(Net *net, const std::map<std::string, HostTensor> &m, const std::string &wname,
                            const std::string &bname, LinearF32 *L, std::string *err) {
  const HostTensor *w, *b;
  if (!get(m, wname, &w, err) || !get(m, bname, &b, err)) return false;
  L->w = upload(net, w->data);
  L->b = upload(net, b->data);
  L->out = w->shape[0]; L->in = w->shape[1];
  return L->w && L->b;
}

static bool make_ln(Net *net, const std::map<std::string, HostTensor> &m, const std::string &prefix, LNParams *L,
                    std::string *err) {
  const HostTensor *w, *b;
  if (!get(m, prefix + ".weight", &w, err) || !get(m, prefix + ".bias", &b, err)) return false;
  L->g = upload(net, w->data);
  L->b = upload(net, b->data);
  return L->g && L->b;
}

static bool make_mha(Net *net, const std::map<std::string, HostTensor> &m, const std::string &prefix, MHA *a,
                     std::string *err) {
  return make_linear_conv(net, m, prefix + ".in_proj_weight", prefix + ".in_proj_bias", &a->in_proj, err) &&
         make_linear_conv(net, m, prefix + ".out_proj.weight", prefix + ".out_proj.bias", &a->out_proj, err) &&
         make_linear_f32(net, m, prefix + ".out_proj.weight", prefix + ".out_proj.bias", &a->out_proj_f32, err);
}

Net *net_load(const char *path, bool is_scorer, std::string *err) {
  std::map<std::string, HostTensor> m;
  if (!read_fpw(path, m, err)) return nullptr;
  std::unique_ptr<Net> net(new Net());
  net->scorer = is_scorer;
  bool ok = make_stem(net.get(), m, "encodeA.0", &net->a0, err) && make_conv(net.get(), m, "encodeA.1", 2, &net->a1, err);
  for (int i = 0; ok && i < 2; i++)
    for (int j = 0; ok && j < 2; j++) {
      std::string cj = ".conv" + std::to_string(j + 1);
      ok = make_conv(net.get(), m, "encodeA." + std::to_string(2 + i) + cj, 1, &net->ra[i][j], err) &&
           make_conv(net.get(), m, "encodeAB." + std::to_string(i) + cj, 1, &net->rb[i][j], err) &&
           make_conv(net.get(), m, "encodeAB." + std::to_string(3 + i) + cj, 1, &net->rc[i][j], err);
    }
  ok = ok && make_conv(net.get(), m, "encodeAB.2", 2, &net->b2, err);
  if (ok && !is_scorer) {
    EncLayer *heads[2] = {&net->trans, &net->rot};
    const char *names[2] = {"trans_head", "rot_head"};
    for (int i = 0; ok && i < 2; i++) {
      std::string p0 = std::string(names[i]) + ".0", p1 = std::string(names[i]) + ".1";
      ok = make_mha(net.get(), m, p0 + ".self_attn", &heads[i]->att, err) &&
           make_linear_conv(net.get(), m, p0 + ".linear1.weight", p0 + ".linear1.bias", &heads[i]->lin1, err) &&
           make_linear_conv(net.get(), m, p0 + ".linear2.weight", p0 + ".linear2.bias", &heads[i]->lin2, err) &&
           make_ln(net.get(), m, p0 + ".norm1", &heads[i]->ln1, err) && make_ln(net.get(), m, p0 + ".norm2", &heads[i]->ln2, err) &&
           make_linear_f32(net.get(), m, p1 + ".weight", p1 + ".bias", &heads[i]->head, err);
    }
    if (ok) {
      ok = make_grouped(net.get(), net->trans.att.in_proj, net->rot.att.in_proj, &net->g_in_proj) &&
           make_grouped(net.get(), net->trans.att.out_proj, net->rot.att.out_proj, &net->g_out_proj) &&
           make_grouped(net.get(), net->trans.lin1, net->rot.lin1, &net->g_lin1) &&
           make_grouped(net.get(), net->trans.lin2, net->rot.lin2, &net->g_lin2);
      if (!ok) *err = "could not build the grouped head weights";
    }
  } else if (ok) {
    ok = make_mha(net.get(), m, "att", &net->att, err) && make_mha(net.get(), m, "att_cross", &net->att_cross, err) &&
         make_linear_f32(net.get(), m, "linear.weight", "linear.bias", &net->score_lin, err);
  }
  if (ok) {
    // PositionalEmbedding(d_model=512, max_len=400): pe[t,2i]=sin(t*w_i), pe[t,2i+1]=cos(t*w_i), w_i=exp(-2i*ln(1e4)/512)
    std::vector<__half> pe((size_t)400 * EMBED);
    for (int t = 0; t < 400; t++)
      for (int i = 0; i < EMBED / 2; i++) {
        float div = std::exp((float)(2 * i) * -(std::log(10000.0f) / (float)EMBED));
        pe[(size_t)t * EMBED + 2 * i] = __float2half(std::sin((float)t * div));
        pe[(size_t)t * EMBED + 2 * i + 1] = __float2half(std::cos((float)t * div));
      }
    net->pe = upload(net.get(), pe);
    ok = net->pe != nullptr;
    if (!ok) *err = "device allocation failed";
  }
  if (!ok) return nullptr;
  return net.release();
}

void net_free(Net *n) { delete n; }

// =================================================================================================
// scratch
// =================================================================================================

struct NNScratch {
  int cap = 0;
  __half *buf = nullptr;
  float *f32 = nullptr;
  // cross-attention head over all gathered hypotheses (sized by n_total, independent of the local shard)
  int head_cap = 0;
  __half *head_buf = nullptr;
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
NNScratch *nn_scratch_create() { return new NNScratch(); }

void nn_scratch_free(NNScratch *w) { delete w; }

// per-hypothesis activation sizes (halfs); conv inputs carry their physical zero border
static constexpr size_t SZ_STEM = 2ull * 82 * 82 * 64;   // stem out, read by the 3x3/s2 conv (border 1)
static constexpr size_t SZ_128 = 2ull * 42 * 42 * 128;
static constexpr size_t SZ_256 = 42ull * 42 * 256;
static constexpr size_t SZ_512 = 22ull * 22 * 512;
static constexpr size_t SZ_TOK = 400ull * 512;            // token buffers (no border)
static constexpr size_t SZ_QKV = 400ull * 1536;
static constexpr size_t PER_HYP = SZ_STEM + 3 * SZ_128 + 3 * SZ_256 + 3 * SZ_512 + SZ_QKV + 4 * SZ_TOK;
void nn_scratch_debug_info(const NNScratch *w, const void **buf, size_t *bytes, const void **f32, size_t *f32_bytes) {
  *buf = w->buf; *bytes = (size_t)w->cap * PER_HYP * sizeof(__half);
  *f32 = w->f32; *f32_bytes = (size_t)w->cap * EMBED * sizeof(float);
}

static int ensure_scratch(NNScratch *ws, int N, hipStream_t s) {
  if (N <= ws->cap) return 0;
  if (ws->buf) (void)hipFree(ws->buf);
  if (ws->f32) (void)hipFree(ws->f32);
  ws->buf = nullptr; ws->f32 = nullptr; ws->cap = 0;
  int cap = std::max(N, 8);
  g_alloc_epoch++;
  FP_HIP_OK(hipMalloc((void **)&ws->buf, (size_t)cap * PER_HYP * sizeof(__half)));
  FP_HIP_OK(hipMalloc((void **)&ws->f32, (size_t)cap * EMBED * sizeof(float)));
  // the zero borders are written here once and never again: every producer stores interiors only, and the arena is
  // carved by CAPACITY (not by the current N), so an image slot's border never moves
  FP_HIP_OK(hipMemsetAsync(ws->buf, 0, (size_t)cap * PER_HYP * sizeof(__half), s));
  ws->cap = cap;
  return 0;
}

static int ensure_head_scratch(NNScratch *ws, int n_total) {
  if (n_total <= ws->head_cap) return 0;
  if (ws->head_buf) (void)hipFree(ws->head_buf);
  if (ws->head_f32) (void)hipFree(ws->head_f32);
  ws->head_buf = nullptr; ws->head_f32 = nullptr; ws->head_cap = 0;
  int cap = std::max(n_total, 256);
  FP_HIP_OK(hipMalloc((void **)&ws->head_buf, (size_t)cap * 5 * EMBED * sizeof(__half)));
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

static bool g_conv_attr_done = false;
static unsigned long long *g_clk_probe = nullptr;
static NNScratch g_hook_ws;  // split-K slab of the fpt_* test hooks
static int g_rem_kernel = 3;  // A/B hook: kernel for the rows the full 256x256 rounds of a long-K layer leave over (3 = conv_deep_kernel<64>, 4 = <128>, 1 = 256x128 ping-pong, 2 = 256x128 3-stage, 0 = 128x128 2-stage)
static int g_rem_small = 1;   // A/B hook (0 = off): long-K layers too small for one full 256x256 round take the left-over path as a whole
static int g_gemm_kernel = 1;  // A/B hook: Linear layers on gemm_k32_kernel (0 = the 256x256 ping-pong tile + left-overs)
static int g_grouped_heads = 1;  // A/B hook: the refiner's two heads as one launch per layer when N == 1 (Track)
static int g_rem_splitk = 0;     // A/B hook: split-K for the rows a 256x256 / 512x128 launch leaves over.  Measured -0.1 ms per
                                 // Register, but OFF: a row's fp32 summation order would then depend on where it falls in the
                                 // batch, and sharded and unsharded Register must pick the same near-tied winner
static int g_splitk_target = 128;  // workgroups a split-K launch aims for (tools/ab_track.py: 96-128 best, 256 is 6 % slower)
static int g_conv_variant = 0;
static int g_conv_ablate = 0;  // timing-only ablations of conv_big_pp_kernel: 1 no loads, 2 no MFMAs, 3 neither  // 0 auto, 1 force the 128-pixel 2-stage kernel, 2 force the 256-pixel 3-stage kernel (A/B hook)

// in: [NB, H+2*ipad, W+2*ipad, Cin]; out: [.., OH+2*opad, OW+2*opad, ..]; res: border rpad
struct ConvGroup {  // two weight groups along M (see ConvParams::grp_rows); L holds [2][Cout][K] weights and [2][Cout] biases
  int rows = 0;     // rows per group (multiple of 128); the launch covers 2 * rows
  bool in_shared = false, res_shared = false;
};
static int run_conv(const Ctx &c, const char *tag, const ConvLayer &L, const __half *in, int NB, int H, int W, int ipad,
                    __half *out, int opad, bool relu, const __half *res = nullptr, int rpad = 0, int split_imgs = 0,
                    const ConvGroup *grp = nullptr) {
  ConvParams p;
  p.clk = g_clk_probe;
  p.grp_rows = grp ? grp->rows : 0;
  p.in_shared = grp && grp->in_shared;
  p.res_shared = grp && grp->res_shared;
  p.grp_w_halfs = grp ? (unsigned)((size_t)L.Cout * L.KH * L.KW * L.Cin) : 0;
  FP_CHECK(!grp || (grp->rows % 128 == 0 && L.KH == 1 && L.KW == 1 && NB == 2 * grp->rows), "grouped launch: unsupported shape");
  p.in = in; p.w = L.w; p.bias = L.bias; p.res = res; p.out = out;
  p.NB = NB; p.H = H; p.W = W; p.Cin = L.Cin;
  p.KH = L.KH; p.KW = L.KW; p.stride = L.stride; p.pad = L.pad;
  p.ipad = ipad; p.opad = opad; p.rpad = rpad;
  p.OH = (H + 2 * L.pad - L.KH) / L.stride + 1;
  p.OW = (W + 2 * L.pad - L.KW) / L.stride + 1;
  if (L.KH == 4 && L.pad == 2 && L.stride == 1) { p.OH = H; p.OW = W; }  // s2d stem: asymmetric padding (2 before, 1 after)
  p.Cout = L.Cout;
  p.M = NB * p.OH * p.OW;
  p.Ktot = L.KH * L.KW * L.Cin;
  p.ntaps = L.KH * L.KW;
  p.relu = relu ? 1 : 0;
  p.split_imgs = split_imgs;
  p.out_ld = split_imgs > 0 ? 2 * L.Cout : L.Cout;
  p.res_ld = L.Cout;
  FP_CHECK(ipad >= L.pad && (L.Cin == 32 || L.Cin % 64 == 0) && p.Ktot % 64 == 0 && (L.Cout % 64) == 0 &&
               (L.Cin != 32 || L.KW % 2 == 0) && p.Ktot / 64 <= 80,
           "conv shape not supported by the MFMA kernel");
  {
    // K order: Cin >= 64 -> (64-channel chunk outer, tap inner); Cin == 32 (s2d stem) -> two horizontally adjacent taps
    // (128 contiguous bytes) per K-step.  koff = byte offset of the K-step's X slab from the row's (tap 0, ch 0) address.
    const int IWp = W + 2 * ipad;
    for (int kt = 0; kt < p.Ktot / 64; kt++) {
      int kh, kw, ch;
      if (L.Cin >= 64) { ch = kt / p.ntaps; int tap = kt % p.ntaps; kh = tap / L.KW; kw = tap % L.KW; }
      else { ch = 0; int tap = 2 * kt; kh = tap / L.KW; kw = tap % L.KW; }
      p.koff[kt] = (unsigned)(((kh * IWp + kw) * L.Cin + ch * 64) * 2);
      if (2 * kt + 1 < 160) { p.koff32[2 * kt] = p.koff[kt]; p.koff32[2 * kt + 1] = p.koff[kt] + 64; }
    }
  }
  double flops = 2.0 * (double)p.M * p.Cout * (L.algo_K > 0 ? L.algo_K : p.Ktot);
  double bytes = ((double)NB * H * W * L.Cin + (double)p.M * p.Cout * (res ? 2 : 1) + (double)p.Cout * p.Ktot) * 2.0;
  constexpr int LDS3_128 = 3 * (256 * 128 + 128 * 128), LDS3_64 = 3 * (256 * 128 + 64 * 128);
  constexpr int LDS_HALO40 = ((10 * 42 + 7) / 8) * 1024 + 3 * 128 * 64;
  constexpr int LDS_STEM_HALO = ((11 * 84 + 15) / 16) * 1024 + 3 * 64 * 64;
  constexpr int LDS_DEEP64 = 6 * (64 + 128) * 128, LDS_DEEP128 = 4 * (128 + 128) * 128;
  constexpr int LDS_GEMM_K32 = 3 * (128 + 256) * 64;
  constexpr int LDS_S2_HALO = ((9 * 41 + 7) / 8) * 1024 + 3 * 128 * 64;
  if (!g_conv_attr_done) {
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_igemm_kernel<128, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 * 128 + 128 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_igemm_kernel<128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 * 128 + 128 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_igemm_kernel<128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 * 128 + 128 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_igemm_kernel<128, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 * 128 + 128 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_igemm_kernel<128, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 * 128 + 128 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_igemm_kernel<128, 11>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 * 128 + 128 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_igemm_kernel<128, 15>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 * 128 + 128 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_igemm_kernel<64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 * 128 + 64 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_igemm3_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS3_128));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_igemm3_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS3_64));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 * 128 + 256 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_pp32_kernel<256, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (256 + 256) * 64));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_pp32_kernel<512, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (512 + 128) * 64));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_big_pp_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 * 128 + 256 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_big_pp_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 * 128 + 256 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_big_pp_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 * 128 + 256 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_big_pp_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 * 128 + 256 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_big_pp_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 * 128 + 256 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_big_pp_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 * 128 + 256 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_big_pp_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 * 128 + 256 * 128)));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_stem_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_STEM_HALO));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_s2_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_S2_HALO));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_deep_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DEEP64));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_deep_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DEEP128));
    FP_HIP_OK(hipFuncSetAttribute((const void *)gemm_k32_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_GEMM_K32));
    FP_HIP_OK(hipFuncSetAttribute((const void *)gemm_k32_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_GEMM_K32));
    FP_HIP_OK(hipFuncSetAttribute((const void *)gemm_k32_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_GEMM_K32));
    FP_HIP_OK(hipFuncSetAttribute((const void *)gemm_k32_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_GEMM_K32));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_halo_kernel<40, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HALO40));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_halo_kernel<40, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HALO40));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_halo_kernel<40, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HALO40));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_halo_kernel<40, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HALO40));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_pp_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS3_128));
    FP_HIP_OK(hipFuncSetAttribute((const void *)conv_pp_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS3_64));
    g_conv_attr_done = true;
  }
  int mtiles = (p.M + 127) / 128;
  const int KT = p.Ktot / 64;
  // split-K for small problems (Track, N <= ~8): a 128x128 tile count far below the 512 workgroup slots of the chip
  // would leave most CUs idle while a few walk up to 72 K-steps; give every CU a slice instead
  p.ksplit = 1; p.kt_per = KT; p.partial = nullptr;
  p.m_begin = 0;
  // (also used for the rows a 256x256 / 512x128 launch leaves over: `target` workgroups on an otherwise idle chip)
  auto plan_splitk = [&](int rows, int target) -> int {
    const int mt = (rows + 127) / 128;
    const int tiles = mt * (L.Cout % 128 == 0 ? L.Cout / 128 : L.Cout / 64);
    if (tiles > 96 || KT < 8 || g_conv_variant == 2) return 0;
    int S = std::min(std::max(target / tiles, 1), KT / 2);
    if (S <= 1) return 0;
    p.kt_per = (KT + S - 1) / S;
    p.ksplit = (KT + p.kt_per - 1) / p.kt_per;
    size_t need = (size_t)p.ksplit * rows * p.Cout;
    NNScratch *sk = c.ws ? c.ws : &g_hook_ws;
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
  if (plan_splitk(p.M, g_splitk_target)) return 1;
  const std::string tg(tag);
  // large problems: 256-pixel tiles, 3-stage LDS-DMA pipeline (needs >= 3 K-steps and enough tiles to fill 256 CUs)
  const int big_tiles = ((p.M + 255) / 256) * (L.Cout % 128 == 0 ? L.Cout / 128 : L.Cout / 64);
  (void)big_tiles;
  // measured (tools/bench_conv.py, N=126): the 128-px 2-stage kernel beats the 256-px 3-stage one on every layer
  // (756-769 vs 703-709 TF/s weighted), so the latter is only reachable through the A/B hook
  if (!grp && (g_conv_variant == 7 || g_conv_variant == 0) && L.Cin == 32 && L.KH == 4 && L.KW == 4 && L.Cout == 64 && ipad == 2 && W == 80 &&
      H == 80 && p.ksplit == 1 && res == nullptr && split_imgs == 0 && (g_conv_variant == 7 || NB * 10 >= 300)) {
    ProfScope ps(c.prof, c.s, (tg + "/conv_stem_halo_kernel").c_str(), flops, bytes);
    hipLaunchKernelGGL(conv_stem_halo_kernel, dim3(NB * 10), dim3(256), LDS_STEM_HALO, c.s, p);
    return 0;
  }
  if (!grp && g_gemm_kernel && g_conv_variant == 0 && L.KH == 1 && L.KW == 1 && L.stride == 1 && L.pad == 0 && ipad == 0 && L.Cout % 256 == 0 &&
      p.Ktot % 32 == 0 && p.ksplit == 1 && split_imgs == 0 && ((p.M + 127) / 128) * (L.Cout / 256) >= 512) {
    ProfScope ps(c.prof, c.s, (tg + "/gemm_k32_kernel").c_str(), flops, bytes);
    const dim3 grid(((p.M + 127) / 128) * (L.Cout / 256));
    switch (g_gemm_kernel) {  // 1 = product; 11 / 12 / 14: timing ablations (wrong results)
      case 11: hipLaunchKernelGGL(gemm_k32_kernel<1>, grid, dim3(256), LDS_GEMM_K32, c.s, p); break;
      case 12: hipLaunchKernelGGL(gemm_k32_kernel<2>, grid, dim3(256), LDS_GEMM_K32, c.s, p); break;
      case 14: hipLaunchKernelGGL(gemm_k32_kernel<4>, grid, dim3(256), LDS_GEMM_K32, c.s, p); break;
      default: hipLaunchKernelGGL(gemm_k32_kernel<0>, grid, dim3(256), LDS_GEMM_K32, c.s, p); break;
    }
    return 0;
  }
  if (!grp && (g_conv_variant == 7 || g_conv_variant == 0) && L.KH == 3 && L.KW == 3 && L.stride == 2 && L.pad == 1 && ipad == 1 &&
      W == 80 && H == 80 && L.Cin == 64 && L.Cout == 128 && p.ksplit == 1 && res == nullptr && split_imgs == 0 &&
      (g_conv_variant == 7 || NB * 10 >= 300)) {
    ProfScope ps(c.prof, c.s, (tg + "/conv_s2_halo_kernel").c_str(), flops, bytes);
    hipLaunchKernelGGL(conv_s2_halo_kernel, dim3(NB * 10), dim3(256), LDS_S2_HALO, c.s, p);
    return 0;
  }
  if (!grp && (g_conv_variant == 7 || g_conv_variant == 0) && L.KH == 3 && L.KW == 3 && L.stride == 1 && L.pad == 1 && ipad == 1 && W == 40 && H % 8 == 0 &&
      L.Cin % 64 == 0 && L.Cout % 128 == 0 && p.ksplit == 1 &&
      (g_conv_variant == 7 || NB * (H / 8) * (L.Cout / 128) >= 300)) {  // measured crossover vs the implicit-GEMM tiles: ~32 hypotheses
    ProfScope ps(c.prof, c.s, (tg + "/conv_halo_kernel").c_str(), flops, bytes);
    const dim3 grid(NB * (H / 8) * (L.Cout / 128));
    if (g_conv_ablate == 1) hipLaunchKernelGGL((conv_halo_kernel<40, 1>), grid, dim3(256), LDS_HALO40, c.s, p);
    else if (g_conv_ablate == 2) hipLaunchKernelGGL((conv_halo_kernel<40, 2>), grid, dim3(256), LDS_HALO40, c.s, p);
    else if (g_conv_ablate == 8) hipLaunchKernelGGL((conv_halo_kernel<40, 8>), grid, dim3(256), LDS_HALO40, c.s, p);
    else hipLaunchKernelGGL((conv_halo_kernel<40, 0>), grid, dim3(256), LDS_HALO40, c.s, p);
    return 0;
  }
  if (!grp && (((g_conv_variant == 0 || g_conv_variant == 8) && L.Cout == 128) || (g_conv_variant == 6 && (L.Cout % 256 == 0 || L.Cout == 128))) && p.ksplit == 1 &&
      KT >= 4 && KT <= 80) {
    // ping-pong tiles (conv_pp32_kernel) for as many FULL rounds of the 256 CUs as the problem has; the remaining rows
    // go to the 128x128 kernel below
    const bool wide = L.Cout % 256 == 0;
    const int bm = wide ? 256 : 512, bn = wide ? 256 : 128;
    const int nt2 = L.Cout / bn;
    const int mt_all = p.M / bm;
    const int mt_big = (mt_all * nt2 / 256) * 256 / nt2;
    if (mt_big > 0) {
      ConvParams pb = p;
      pb.M = mt_big * bm;
      const double frac = (double)pb.M / (double)p.M;
      {
        ProfScope ps(c.prof, c.s, (tg + (wide ? "/conv_pp32_kernel<256,256>" : "/conv_pp32_kernel<512,128>")).c_str(), flops * frac, bytes * frac);
        if (wide) hipLaunchKernelGGL((conv_pp32_kernel<256, 256>), dim3(mt_big * nt2), dim3(512), 4 * (256 + 256) * 64, c.s, pb);
        else hipLaunchKernelGGL((conv_pp32_kernel<512, 128>), dim3(mt_big * nt2), dim3(512), 4 * (512 + 128) * 64, c.s, pb);
      }
      flops *= (1.0 - frac); bytes *= (1.0 - frac);
      p.m_begin = mt_big * bm;
      if (p.m_begin >= p.M) return 0;
      mtiles = (p.M - p.m_begin + 127) / 128;
      if (g_rem_splitk && plan_splitk(p.M - p.m_begin, 384)) return 1;
    }
  }
  if (!grp && (g_conv_variant == 4 || g_conv_variant == 5 || g_conv_variant == 0 || g_conv_variant == 8) && L.Cout % 256 == 0 && p.ksplit == 1 && KT >= 2) {
    // 256x256 tiles for as many FULL rounds of the 256 CUs as the problem has, the remaining rows on 128x128 tiles
    const int nt2 = L.Cout / 256;
    const int mt_all = p.M / 256;                         // whole 256-row m-tiles
    const int mt_big = (mt_all * nt2 / 256) * 256 / nt2;  // m-tiles covered by full rounds
    if (mt_big > 0) {
      const int lds_big = 2 * (256 * 128 + 256 * 128);
      ConvParams pb = p;
      pb.M = mt_big * 256;                                // rows [0, mt_big*256)
      const double frac = (double)pb.M / (double)p.M;
      {
        ProfScope ps(c.prof, c.s, (tg + (g_conv_variant == 4 ? "/conv_big_kernel" : "/conv_big_pp_kernel")).c_str(), flops * frac, bytes * frac);
        if (g_conv_variant == 4) hipLaunchKernelGGL(conv_big_kernel, dim3(mt_big * nt2), dim3(512), lds_big, c.s, pb);
        else if (g_conv_ablate == 1) hipLaunchKernelGGL(conv_big_pp_kernel<1>, dim3(mt_big * nt2), dim3(512), lds_big, c.s, pb);
        else if (g_conv_ablate == 2) hipLaunchKernelGGL(conv_big_pp_kernel<2>, dim3(mt_big * nt2), dim3(512), lds_big, c.s, pb);
        else if (g_conv_ablate == 3) hipLaunchKernelGGL(conv_big_pp_kernel<3>, dim3(mt_big * nt2), dim3(512), lds_big, c.s, pb);
        else if (g_conv_ablate == 4) hipLaunchKernelGGL(conv_big_pp_kernel<4>, dim3(mt_big * nt2), dim3(512), lds_big, c.s, pb);
        else if (g_conv_ablate == 8) hipLaunchKernelGGL(conv_big_pp_kernel<8>, dim3(mt_big * nt2), dim3(512), lds_big, c.s, pb);
        else if (g_conv_ablate == 16) hipLaunchKernelGGL(conv_big_pp_kernel<16>, dim3(mt_big * nt2), dim3(512), lds_big, c.s, pb);
        else hipLaunchKernelGGL(conv_big_pp_kernel<0>, dim3(mt_big * nt2), dim3(512), lds_big, c.s, pb);
      }
      flops *= (1.0 - frac); bytes *= (1.0 - frac);
      p.m_begin = mt_big * 256;
      if (p.m_begin >= p.M) return 0;
      mtiles = (p.M - p.m_begin + 127) / 128;
      if (g_rem_splitk && plan_splitk(p.M - p.m_begin, 384)) return 1;
    }
  }
  if (!grp && (p.m_begin > 0 || (g_rem_small && p.M >= 8192)) && g_rem_kernel && KT >= 16 && p.ksplit == 1 && L.Cout % 128 == 0) {
    // left-over rows on an otherwise idle chip: a lone workgroup per CU walks all K-steps, so per-step latency is what
    // counts: conv_512 left-overs 61 us per launch on the 2-stage 128x128 tile, 55 us on the 256x128 ping-pong, 39 us on
    // conv_deep_kernel<64> (neutral-to-slower for the 8-step Linear layers, hence KT >= 16)
    const int n128 = L.Cout / 128;
    if (g_rem_kernel == 4) {
      const dim3 grid(((p.M - p.m_begin + 127) / 128) * n128);
      ProfScope ps(c.prof, c.s, (tg + "/conv_deep_kernel").c_str(), flops, bytes);
      hipLaunchKernelGGL(conv_deep_kernel<128>, grid, dim3(256), LDS_DEEP128, c.s, p);
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
          hipLaunchKernelGGL(conv_pp_kernel<128>, dim3(((pb.M - pb.m_begin + 255) / 256) * n128), dim3(512), LDS3_128, c.s, pb);
        }
        if (!cascade || rest == 0) return 0;
        flops *= (1.0 - frac); bytes *= (1.0 - frac);
        p.m_begin += rows_pp;
        rows = rest;
      }
      ProfScope ps(c.prof, c.s, (tg + "/conv_deep_kernel").c_str(), flops, bytes);
      hipLaunchKernelGGL(conv_deep_kernel<64>, dim3(((rows + 63) / 64) * n128), dim3(256), LDS_DEEP64, c.s, p);
      return 0;
    }
    const int mt2 = (p.M - p.m_begin + 255) / 256;
    ProfScope ps(c.prof, c.s, (tg + (g_rem_kernel == 1 ? "/conv_pp_kernel(rem)" : "/conv_igemm3_kernel(rem)")).c_str(), flops, bytes);
    if (g_rem_kernel == 1) hipLaunchKernelGGL(conv_pp_kernel<128>, dim3(mt2 * (L.Cout / 128)), dim3(512), LDS3_128, c.s, p);
    else hipLaunchKernelGGL(conv_igemm3_kernel<128>, dim3(mt2 * (L.Cout / 128)), dim3(512), LDS3_128, c.s, p);
    return 0;
  }
  if (!grp && g_conv_variant == 3 && KT >= 3 && p.ksplit == 1) {
    ProfScope ps(c.prof, c.s, (tg + "/conv_pp_kernel").c_str(), flops, bytes);
    int mt2 = (p.M + 255) / 256;
    if (L.Cout % 128 == 0)
      hipLaunchKernelGGL(conv_pp_kernel<128>, dim3(mt2 * (L.Cout / 128)), dim3(512), LDS3_128, c.s, p);
    else
      hipLaunchKernelGGL(conv_pp_kernel<64>, dim3(mt2 * (L.Cout / 64)), dim3(512), LDS3_64, c.s, p);
    return 0;
  }
  if (!grp && g_conv_variant == 2 && KT >= 3) {
    ProfScope ps(c.prof, c.s, (tg + "/conv_igemm3_kernel").c_str(), flops, bytes);
    int mt2 = (p.M + 255) / 256;
    if (L.Cout % 128 == 0)
      hipLaunchKernelGGL(conv_igemm3_kernel<128>, dim3(mt2 * (L.Cout / 128)), dim3(512), LDS3_128, c.s, p);
    else
      hipLaunchKernelGGL(conv_igemm3_kernel<64>, dim3(mt2 * (L.Cout / 64)), dim3(512), LDS3_64, c.s, p);
    return 0;
  }
  if (L.Cout % 128 == 0) {
    ProfScope ps(c.prof, c.s, (tg + "/conv_igemm_kernel<128>").c_str(), flops, bytes);
    const dim3 grid(mtiles * (L.Cout / 128) * p.ksplit);
    const int lds = 2 * (128 * 128 + 128 * 128);
    switch (g_conv_variant) {
      case 11: hipLaunchKernelGGL((conv_igemm_kernel<128, 1>), grid, dim3(256), lds, c.s, p); break;
      case 12: hipLaunchKernelGGL((conv_igemm_kernel<128, 2>), grid, dim3(256), lds, c.s, p); break;
      case 13: hipLaunchKernelGGL((conv_igemm_kernel<128, 3>), grid, dim3(256), lds, c.s, p); break;
      case 10: hipLaunchKernelGGL((conv_igemm_kernel<128, 0>), grid, dim3(256), lds, c.s, p); break;
      case 17: hipLaunchKernelGGL((conv_igemm_kernel<128, 7>), grid, dim3(256), lds, c.s, p); break;   // no loads
      case 21: hipLaunchKernelGGL((conv_igemm_kernel<128, 11>), grid, dim3(256), lds, c.s, p); break;  // no MFMAs
      case 25: hipLaunchKernelGGL((conv_igemm_kernel<128, 15>), grid, dim3(256), lds, c.s, p); break;  // neither
      default: hipLaunchKernelGGL((conv_igemm_kernel<128, 3>), grid, dim3(256), lds, c.s, p); break;
    }
  } else {
    ProfScope ps(c.prof, c.s, (tg + "/conv_igemm_kernel<64>").c_str(), flops, bytes);
    hipLaunchKernelGGL((conv_igemm_kernel<64, 3>), dim3(mtiles * (L.Cout / 64) * p.ksplit), dim3(256), 2 * (128 * 128 + 64 * 128), c.s, p);
  }
  if (p.ksplit > 1) {
    ProfScope ps(c.prof, c.s, (tg + "/conv_splitk_reduce_kernel").c_str(), 0, 0);
    size_t quads = (size_t)(p.M - p.m_begin) * (p.Cout / 4);
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, c.s, p);
  }
  return 0;
}

// plain GEMM rows x Cin -> rows x Cout (Linear layer) on unpadded buffers
static int run_gemm(const Ctx &c, const char *tag, const ConvLayer &L, const __half *in, int rows, __half *out, bool relu,
                    const __half *res = nullptr, const ConvGroup *grp = nullptr) {
  return run_conv(c, tag, L, in, rows, 1, 1, 0, out, 0, relu, res, 0, 0, grp);
}

static int g_att_variant = 1;  // A/B hook (tools/ab_attention.py): 1 remap + 16-B stores, 3 no XCD remap, 5 remap + 2-B stores, 7 neither
static int run_attention(const Ctx &c, const __half *qkv, __half *out, int B, int T, int tstride = 0) {
  if (tstride == 0) tstride = T;
  double flops = 4.0 * (double)B * HEADS * (double)T * T * HDIM;
  ProfScope ps(c.prof, c.s, "attention", flops, (double)B * T * (1536 + 512) * 2.0);
  const int nq = (T + 63) / 64;
  dim3 grid((unsigned)(nq * HEADS * B)), blk(256);
  switch (g_att_variant) {
    case 1: hipLaunchKernelGGL((attention_kernel<64, true>), grid, blk, 0, c.s, qkv, out, T, nq, tstride); break;
    case 3: hipLaunchKernelGGL((attention_kernel<64, false>), grid, blk, 0, c.s, qkv, out, T, nq, tstride); break;
    case 5: hipLaunchKernelGGL((attention_kernel<64, true, false>), grid, blk, 0, c.s, qkv, out, T, nq, tstride); break;
    default: hipLaunchKernelGGL((attention_kernel<64, false, false>), grid, blk, 0, c.s, qkv, out, T, nq, tstride); break;
  }
  return 0;
}

static void run_layernorm(const Ctx &c, const __half *x, const LNParams &ln, __half *y, size_t rows, const LNParams *ln1 = nullptr,
                          size_t split_row = 0) {
  ProfScope ps(c.prof, c.s, "layernorm", 0, (double)rows * EMBED * 4.0);
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, c.s, x, ln.g, ln.b, y, rows,
                     ln1 ? ln1->g : ln.g, ln1 ? ln1->b : ln.b, ln1 ? split_row : rows);
}

static void run_small_linear(const Ctx &c, const float *x, const LinearF32 &L, float *y, int B) {
  ProfScope ps(c.prof, c.s, "small_linear", 2.0 * B * L.out * L.in, 0);
  size_t waves = (size_t)B * L.out;
  hipLaunchKernelGGL(small_linear_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, c.s, x, L.w, L.b, y, B, L.out, L.in);
}

static void run_token_mean(const Ctx &c, const __half *x, float *out, int B, int T, int tstride = 0) {
  ProfScope ps(c.prof, c.s, "token_mean", 0, (double)B * T * EMBED * 2.0);
  hipLaunchKernelGGL(token_mean_kernel, dim3(B, EMBED / 64), dim3(256), 0, c.s, x, out, T, tstride ? tstride : T);
}

// arena carve (by capacity, see ensure_scratch)
struct Arena {
  __half *stem, *x128[3], *x256[3], *x512[3], *tokens, *qkv, *att, *y1, *y2;
};
static Arena carve(NNScratch *ws) {
  Arena a;
  const size_t cap = (size_t)ws->cap;
  __half *p = ws->buf;
  a.stem = p; p += cap * SZ_STEM;
  for (int i = 0; i < 3; i++) { a.x128[i] = p; p += cap * SZ_128; }
  for (int i = 0; i < 3; i++) { a.x256[i] = p; p += cap * SZ_256; }
  for (int i = 0; i < 3; i++) { a.x512[i] = p; p += cap * SZ_512; }
  a.tokens = p; p += cap * SZ_TOK;
  a.qkv = p; p += cap * SZ_QKV;
  a.att = p; p += cap * SZ_TOK;
  a.y1 = p; p += cap * SZ_TOK;
  a.y2 = p; p += cap * SZ_TOK;
  return a;
}

// shared CNN trunk: nn_in [2N,84,84,32] (s2d, border 2) -> tokens [N,400,512] + positional embedding
// n_b = number of observed-crop (B) images following the N rendered (A) images: N, or 1 when all hypotheses share it
static int run_trunk(const Ctx &c, const Arena &a, const __half *nn_in, int N, int n_b) {
  const Net *net = c.net;
  const int NB2 = N + n_b;
  if (run_conv(c, "conv_stem", net->a0, nn_in, NB2, 80, 80, 2, a.stem, 1, true)) return 1;
  if (run_conv(c, "conv_a1", net->a1, a.stem, NB2, 80, 80, 1, a.x128[0], 1, true)) return 1;
  // encodeA residual blocks @40x40x128; the last conv writes the a|b channel concat directly
  if (run_conv(c, "conv_128", net->ra[0][0], a.x128[0], NB2, 40, 40, 1, a.x128[1], 1, true)) return 1;
  if (run_conv(c, "conv_128", net->ra[0][1], a.x128[1], NB2, 40, 40, 1, a.x128[2], 1, true, a.x128[0], 1)) return 1;
  if (run_conv(c, "conv_128", net->ra[1][0], a.x128[2], NB2, 40, 40, 1, a.x128[1], 1, true)) return 1;
  if (run_conv(c, "conv_128", net->ra[1][1], a.x128[1], NB2, 40, 40, 1, a.x256[0], 1, true, a.x128[2], 1, N)) return 1;
  if (n_b == 1 && N > 1) {  // image N landed in cat[0][..,128:256]; replicate it for the other hypotheses
    ProfScope ps(c.prof, c.s, "broadcast_b", 0, (double)N * 1600 * 256);
    size_t total = (size_t)(N - 1) * 1600 * 16;
    hipLaunchKernelGGL(broadcast_b_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.s, a.x256[0], N, 42, 42, 40, 40, 1, 128);
  }
  // encodeAB
  if (run_conv(c, "conv_256", net->rb[0][0], a.x256[0], N, 40, 40, 1, a.x256[1], 1, true)) return 1;
  if (run_conv(c, "conv_256", net->rb[0][1], a.x256[1], N, 40, 40, 1, a.x256[2], 1, true, a.x256[0], 1)) return 1;
  if (run_conv(c, "conv_256", net->rb[1][0], a.x256[2], N, 40, 40, 1, a.x256[1], 1, true)) return 1;
  if (run_conv(c, "conv_256", net->rb[1][1], a.x256[1], N, 40, 40, 1, a.x256[0], 1, true, a.x256[2], 1)) return 1;
  if (run_conv(c, "conv_b2", net->b2, a.x256[0], N, 40, 40, 1, a.x512[0], 1, true)) return 1;
  if (run_conv(c, "conv_512", net->rc[0][0], a.x512[0], N, 20, 20, 1, a.x512[1], 1, true)) return 1;
  if (run_conv(c, "conv_512", net->rc[0][1], a.x512[1], N, 20, 20, 1, a.x512[2], 1, true, a.x512[0], 1)) return 1;
  if (run_conv(c, "conv_512", net->rc[1][0], a.x512[2], N, 20, 20, 1, a.x512[1], 1, true)) return 1;
  // last conv writes the un-bordered token tensor [N,400,512]
  if (run_conv(c, "conv_512", net->rc[1][1], a.x512[1], N, 20, 20, 1, a.tokens, 0, true, a.x512[2], 1)) return 1;
  {
    size_t rows = (size_t)N * 400;
    ProfScope ps(c.prof, c.s, "add_pos_embed", 0, (double)rows * EMBED * 4.0);
    size_t chunks = rows * (EMBED / 8);
    hipLaunchKernelGGL(add_pos_embed_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, c.s, a.tokens, net->pe, 400, rows);
  }
  return 0;
}

int refiner_forward(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const __half *nn_in, int N,
                    float *trans_dev, float *rot_dev, int shared_b) {
  FP_CHECK(net && !net->scorer, "refiner_forward: wrong network");
  if (ensure_scratch(ws, N, s)) return 1;
  Ctx c{s, prof, net, ws};
  const Arena a = carve(ws);
  if (run_trunk(c, a, nn_in, N, shared_b ? 1 : N)) return 1;
  const __half *x = a.tokens;
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
    if (run_attention(c, a.qkv, a.att, 2, 400, G)) return 1;
    if (run_gemm(c, "gemm_512", net->g_out_proj, a.att, 2 * G, a.y1, false, x, &g_in)) return 1;   // + residual x (shared)
    run_layernorm(c, a.y1, T0.ln1, a.y2, 2 * G, &R0.ln1, G);
    if (run_gemm(c, "gemm_512", net->g_lin1, a.y2, 2 * G, a.y1, true, nullptr, &g_own)) return 1;
    if (run_gemm(c, "gemm_512", net->g_lin2, a.y1, 2 * G, a.att, false, a.y2, &g_own)) return 1;   // + residual x1
    run_layernorm(c, a.att, T0.ln2, a.y1, 2 * G, &R0.ln2, G);
    run_token_mean(c, a.y1, ws->f32, 2, 400, G);
    run_small_linear(c, ws->f32, T0.head, trans_dev, 1);
    run_small_linear(c, ws->f32 + EMBED, R0.head, rot_dev, 1);
    FP_HIP_OK(hipGetLastError());
    return 0;
  }
  for (int i = 0; i < 2; i++) {
    const EncLayer &L = *heads[i];
    // post-norm TransformerEncoderLayer: x1 = LN1(x + SA(x)); x2 = LN2(x1 + W2 relu(W1 x1))
    if (run_gemm(c, "gemm_qkv", L.att.in_proj, x, (int)rows, a.qkv, false)) return 1;
    if (run_attention(c, a.qkv, a.att, N, 400)) return 1;
    if (run_gemm(c, "gemm_512", L.att.out_proj, a.att, (int)rows, a.y1, false, x)) return 1;  // + residual x
    run_layernorm(c, a.y1, L.ln1, a.y2, rows);                                               // x1 = y2
    if (run_gemm(c, "gemm_512", L.lin1, a.y2, (int)rows, a.y1, true)) return 1;
    if (run_gemm(c, "gemm_512", L.lin2, a.y1, (int)rows, a.att, false, a.y2)) return 1;       // + residual x1
    run_layernorm(c, a.att, L.ln2, a.y1, rows);
    run_token_mean(c, a.y1, ws->f32, N, 400);
    run_small_linear(c, ws->f32, L.head, outs[i], N);  // Linear(512,3) commutes with the token mean
  }
  FP_HIP_OK(hipGetLastError());
  return 0;
}

int scorer_features(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const __half *nn_in, int N, float *feat_dev) {
  FP_CHECK(net && net->scorer, "scorer_features: wrong network");
  if (ensure_scratch(ws, N, s)) return 1;
  Ctx c{s, prof, net, ws};
  const Arena a = carve(ws);
  if (run_trunk(c, a, nn_in, N, N)) return 1;
  const size_t rows = (size_t)N * 400;
  if (run_gemm(c, "gemm_qkv", net->att.in_proj, a.tokens, (int)rows, a.qkv, false)) return 1;
  if (run_attention(c, a.qkv, a.att, N, 400)) return 1;
  // feature = mean_t(out_proj(att)) = out_proj(mean_t(att))  (out_proj is affine) -> 512x512 GEMV per hypothesis
  run_token_mean(c, a.att, ws->f32, N, 400);
  run_small_linear(c, ws->f32, net->att.out_proj_f32, feat_dev, N);
  FP_HIP_OK(hipGetLastError());
  return 0;
}

int scorer_head(hipStream_t s, Profiler *prof, const Net *net, NNScratch *ws, const float *feats_dev, int n_total, float *scores_dev) {
  FP_CHECK(net && net->scorer, "scorer_head: wrong network");
  if (ensure_head_scratch(ws, n_total)) return 1;
  Ctx c{s, prof, net, ws};
  const int N = n_total;
  __half *p = ws->head_buf;
  __half *xf = p; p += (size_t)N * EMBED;
  __half *qkv = p; p += (size_t)N * 3 * EMBED;
  __half *att = p; p += (size_t)N * EMBED;
  float *o32 = ws->head_f32;                  // [N,512]
  {
    ProfScope ps(c.prof, c.s, "cast", 0, (double)N * EMBED * 6.0);
    size_t n = (size_t)N * EMBED;
    hipLaunchKernelGGL(cast_f32_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.s, feats_dev, xf, n);
  }
  // att_cross: sequence = the N hypotheses, batch 1
  if (run_gemm(c, "gemm_cross", net->att_cross.in_proj, xf, N, qkv, false)) return 1;
  if (run_attention(c, qkv, att, 1, N)) return 1;
  // out_proj through the same MFMA GEMM (M = N rows), then Linear(512,1) in f32
  if (run_gemm(c, "gemm_cross", net->att_cross.out_proj, att, N, xf, false)) return 1;
  {
    // Linear(512,1) on fp16 rows: widen to f32 first (tiny)
    ProfScope ps(c.prof, c.s, "score_linear", 2.0 * N * EMBED, 0);
    // token_mean with T = 1 is a plain fp16 -> f32 copy of each row
    hipLaunchKernelGGL(token_mean_kernel, dim3(N, EMBED / 64), dim3(256), 0, c.s, xf, o32, 1, 1);
  }
  run_small_linear(c, o32, net->score_lin, scores_dev, N);
  FP_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace fp

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
}  // namespace

extern "C" {

void fpt_set_conv_variant(int v) { fp::g_conv_variant = v; }
void fpt_set_conv_ablate(int v) { fp::g_conv_ablate = v; }
void fpt_set_splitk_target(int v) { fp::g_splitk_target = v; }
void fpt_set_rem_splitk(int v) { fp::g_rem_splitk = v; }
void fpt_set_grouped_heads(int v) { fp::g_grouped_heads = v; }
void fpt_set_gemm_kernel(int v) { fp::g_gemm_kernel = v; }
void fpt_set_rem_kernel(int v) { fp::g_rem_kernel = v; }
void fpt_set_rem_small(int v) { fp::g_rem_small = v; }
void fpt_set_raster_strip_rows(int r) { fp::set_raster_strip_rows(r); }

// clock probe: allocate room for `blocks` records, run convs, then read back mean shader MHz and mean main-loop cycles
int fpt_clk_probe(int blocks, double *mhz_out, double *loop_cycles_out) {
  using namespace fp;
  if (blocks > 0) {
    if (g_clk_probe) (void)hipFree(g_clk_probe);
    FP_HIP_OK(hipMalloc((void **)&g_clk_probe, (size_t)blocks * 32));
    FP_HIP_OK(hipMemset(g_clk_probe, 0, (size_t)blocks * 32));
    return 0;
  }
  FP_CHECK(g_clk_probe, "no probe");
  int n = -blocks;
  std::vector<unsigned long long> h((size_t)n * 4);
  FP_HIP_OK(hipDeviceSynchronize());
  FP_HIP_OK(hipMemcpy(h.data(), g_clk_probe, h.size() * 8, hipMemcpyDeviceToHost));
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

// x [NB,H,W,Cin] NHWC f32, w [Cout,KH,KW,Cin] f32 (already in kernel layout), bias [Cout], res (optional) [NB,OH,OW,Cout]
// -> out f32 [NB,OH,OW,Cout] (or, with split_imgs > 0, [NB-split,OH,OW,2*Cout]).  iters > 1: returns mean ms in *ms_out.
int fpt_conv(const float *x, const float *w, const float *bias, const float *res, int NB, int H, int W, int Cin, int Cout,
             int KH, int KW, int stride, int pad, int OH, int OW, int relu, int split_imgs, float *out, int iters,
             float *ms_out) {
  using namespace fp;
  const int ip = pad;  // physical input border
  const int Hp = H + 2 * ip, Wp = W + 2 * ip;
  size_t nx = (size_t)NB * Hp * Wp * Cin, nw = (size_t)Cout * KH * KW * Cin;
  size_t M = (size_t)NB * OH * OW;
  size_t nout = M * Cout;
  DevBuf<__half> dx(nx), dw(nw), dres(nout), dout(nout * 2);
  DevBuf<float> db(Cout);
  FP_CHECK(dx.p && dw.p && dres.p && dout.p && db.p, "fpt_conv: allocation failed");
  std::vector<__half> hx(nx, __float2half(0.f));
  for (int n = 0; n < NB; n++)
    for (int y = 0; y < H; y++)
      for (int xx = 0; xx < W; xx++)
        for (int c = 0; c < Cin; c++)
          hx[(((size_t)n * Hp + y + ip) * Wp + xx + ip) * Cin + c] = __float2half(x[(((size_t)n * H + y) * W + xx) * Cin + c]);
  auto hw = permute_rows(relayout_k(to_half(w, nw), Cout, KH * KW, Cin), Cout, KH * KW * Cin);
  FP_HIP_OK(hipMemcpy(dx.p, hx.data(), nx * 2, hipMemcpyHostToDevice));
  FP_HIP_OK(hipMemcpy(dw.p, hw.data(), nw * 2, hipMemcpyHostToDevice));
  FP_HIP_OK(hipMemcpy(db.p, bias, (size_t)Cout * 4, hipMemcpyHostToDevice));
  FP_HIP_OK(hipMemset(dout.p, 0, nout * 4));
  if (res) {
    auto hr = to_half(res, nout);
    FP_HIP_OK(hipMemcpy(dres.p, hr.data(), nout * 2, hipMemcpyHostToDevice));
  }
  Net net;
  ConvLayer L;
  L.w = dw.p; L.bias = db.p; L.Cin = Cin; L.Cout = Cout; L.KH = KH; L.KW = KW; L.stride = stride; L.pad = pad;
  Ctx c{nullptr, nullptr, &net};
  (void)OH; (void)OW;
  hipEvent_t e0, e1;
  FP_HIP_OK(hipEventCreate(&e0));
  FP_HIP_OK(hipEventCreate(&e1));
  if (run_conv(c, "t", L, dx.p, NB, H, W, ip, dout.p, 0, relu != 0, res ? dres.p : nullptr, 0, split_imgs)) return 1;
  FP_HIP_OK(hipDeviceSynchronize());
  if (iters > 1) {
    FP_HIP_OK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; i++)
      if (run_conv(c, "t", L, dx.p, NB, H, W, ip, dout.p, 0, relu != 0, res ? dres.p : nullptr, 0, split_imgs)) return 1;
    FP_HIP_OK(hipEventRecord(e1, nullptr));
    FP_HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    FP_HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / iters;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  std::vector<__half> ho(nout);
  FP_HIP_OK(hipMemcpy(ho.data(), dout.p, nout * 2, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < nout; i++) out[i] = __half2float(ho[i]);
  return 0;
}

// qkv f32 [B,T,1536] -> out f32 [B,T,512]
int fpt_attention(const float *qkv, int B, int T, float *out) {
  using namespace fp;
  size_t nq = (size_t)B * T * 1536, no = (size_t)B * T * 512;
  DevBuf<__half> dq(nq), dout(no);
  FP_CHECK(dq.p && dout.p, "fpt_attention: allocation failed");
  auto hq = to_half(qkv, nq);
  FP_HIP_OK(hipMemcpy(dq.p, hq.data(), nq * 2, hipMemcpyHostToDevice));
  Ctx c{nullptr, nullptr, nullptr};
  if (run_attention(c, dq.p, dout.p, B, T)) return 1;
  FP_HIP_OK(hipDeviceSynchronize());
  std::vector<__half> ho(no);
  FP_HIP_OK(hipMemcpy(ho.data(), dout.p, no * 2, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < no; i++) out[i] = __half2float(ho[i]);
  return 0;
}


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
      (void)hipMemcpy(dx.p, hx.data(), nx * 2, hipMemcpyHostToDevice);
      (void)hipMemcpy(dw.p, hw.data(), nw * 2, hipMemcpyHostToDevice);
      (void)hipMemcpy(dres.p, hr.data(), nout * 2, hipMemcpyHostToDevice);
      (void)hipMemcpy(db.p, hb.data(), (size_t)Cout * 4, hipMemcpyHostToDevice);
      (void)hipMemset(dout.p, 0, nout * 2);
      (void)hipMemset(dref.p, 0, nout * 2);
      (void)hipMemset(dcnt.p, 0, 8);
      Net net;
      ConvLayer L;
      L.w = dw.p; L.bias = db.p; L.Cin = Cin; L.Cout = Cout; L.KH = 3; L.KW = 3; L.stride = 1; L.pad = 1;
      NNScratch ws;
      Ctx c{s, nullptr, &net, &ws};
      if (run_conv(c, "t", L, dx.p, NB, H, H, 1, dref.p, 1, true, with_res ? dres.p : nullptr, 1, 0)) return;
      (void)hipStreamSynchronize(s);
      for (int i = 0; i < iters; i++) {
        if (run_conv(c, "t", L, dx.p, NB, H, H, 1, dout.p, 1, true, with_res ? dres.p : nullptr, 1, 0)) return;
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
  if (hipMalloc((void **)&dbad, 8) != hipSuccess || hipMemset(dbad, 0, 8) != hipSuccess) return -1;
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
  (void)hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
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

void fpt_set_att_variant(int v) { fp::g_att_variant = v; }

// timing hook: random QKV resident in HBM, `iters` launches, returns ms per launch (negative on failure)
float fpt_attention_bench(int B, int T, int iters, int variant) {
  using namespace fp;
  size_t nq = (size_t)B * T * 1536, no = (size_t)B * T * 512;
  DevBuf<__half> dq(nq), dout(no);
  if (!dq.p || !dout.p) return -1.f;
  std::vector<__half> hq(nq);
  uint32_t st = 12345u;
  for (size_t i = 0; i < nq; i++) { st = st * 1664525u + 1013904223u; hq[i] = __float2half(((st >> 8) & 0xffff) / 65536.0f - 0.5f); }
  if (hipMemcpy(dq.p, hq.data(), nq * 2, hipMemcpyHostToDevice) != hipSuccess) return -1.f;
  Ctx c{nullptr, nullptr, nullptr};
  int saved = g_att_variant;
  g_att_variant = variant;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
  for (int i = 0; i < 3; i++) run_attention(c, dq.p, dout.p, B, T);
  (void)hipEventRecord(e0, nullptr);
  for (int i = 0; i < iters; i++) run_attention(c, dq.p, dout.p, B, T);
  (void)hipEventRecord(e1, nullptr);
  float ms = -1.f;
  if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) ms = -(float)iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  g_att_variant = saved;
  return ms / iters;
}

// sustained dense fp16 MFMA rate in TFLOP/s (whole chip, `waves_per_simd` waves on every SIMD); negative on failure
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
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(wgs), dim3(256), 0, s, out.p, iters / 4, zero_operands, (unsigned long long *)nullptr);  // warm-up / clock ramp
  (void)hipEventRecord(e0, s);
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(wgs), dim3(256), 0, s, out.p, iters, zero_operands, clk.p);
  (void)hipEventRecord(e1, s);
  float ms = -1.f;
  const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(s);
  if (!ok || ms <= 0.f) return -1.f;
  if (mhz) {  // shader clock during the run: cycle counter against the 100 MHz wall clock
    unsigned long long h[2] = {0, 0};
    if (hipMemcpy(h, clk.p, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -1.f;
    *mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
  }
  const double flops = (double)wgs * 4.0 * (double)iters * 8.0 * 16384.0;
  return (float)(flops / (ms * 1e-3) / 1e12);
}

}  // extern "C"
