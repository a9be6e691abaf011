// Micro-repro for the packed-f32 finding of round 2 (DESIGN.md "Concurrent models"): do packed-f32 VALU instructions
// (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) return wrong values when waves of ANOTHER queue's kernel share the SIMD?
//   victim    : short kernels on stream A; every lane evaluates the same products once with packed and once with
//               scalar instructions (inline asm, so the compiler cannot change either) and counts bitwise mismatches
//               per lane.
//   aggressor : long-running kernel on stream B (mfma / valu / lds / exp / none), small enough in registers and LDS
//               that victim waves are co-resident with it.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/repro_pkf32 tools/repro_pkf32.hip ; run: /tmp/repro_pkf32 [seconds]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

// MODE 0: v_pk_mul_f32, 1: v_pk_add_f32, 2: v_pk_fma_f32
template <int MODE>
__global__ __launch_bounds__(256) void victim_kernel(unsigned long long *hist /*[64]*/, unsigned seed, int iters) {
  const int lane = threadIdx.x & 63;
  unsigned st = seed ^ (2654435761u * (unsigned)(blockIdx.x * 256 + threadIdx.x + 1));
  unsigned bad = 0;
  for (int i = 0; i < iters; i++) {
    st = st * 1664525u + 1013904223u;
    float a0 = (float)((st >> 8) & 0xffff) * (1.0f / 65536.0f) - 0.5f;
    st = st * 1664525u + 1013904223u;
    float a1 = (float)((st >> 8) & 0xffff) * (1.0f / 65536.0f) - 0.5f;
    st = st * 1664525u + 1013904223u;
    float b0 = (float)((st >> 8) & 0xffff) * (1.0f / 65536.0f) + 0.25f;
    st = st * 1664525u + 1013904223u;
    float b1 = (float)((st >> 8) & 0xffff) * (1.0f / 65536.0f) + 0.25f;
    f2 a = {a0, a1}, b = {b0, b1}, r;
    float s0, s1;
    if (MODE == 0) {
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(a0), "v"(b0));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "v"(a1), "v"(b1));
    } else if (MODE == 1) {
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(a0), "v"(b0));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(a1), "v"(b1));
    } else {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %1" : "=v"(r) : "v"(a), "v"(b));
      asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(s0) : "v"(a0), "v"(b0));
      asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(s1) : "v"(a1), "v"(b1));
    }
    bad += (__float_as_uint(r[0]) != __float_as_uint(s0)) | (__float_as_uint(r[1]) != __float_as_uint(s1));
  }
  if (bad) atomicAdd(&hist[lane], (unsigned long long)bad);
}

// compiler-generated packed math (clang SLP-vectorised): a 3x3 transform + normalisation, like the vertex stage
__global__ __launch_bounds__(256) void victim_c_kernel(const float *__restrict__ mats, const float *__restrict__ pts, int n,
                                                       f4 *__restrict__ out) {
  __shared__ float m[16];
  if (threadIdx.x < 16) m[threadIdx.x] = mats[blockIdx.y * 16 + threadIdx.x];
  __syncthreads();
  int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= n) return;
  float x = pts[v * 3], y = pts[v * 3 + 1], z = pts[v * 3 + 2];
  float ux = m[0] * x + m[4] * y + m[8] * z;
  float uy = m[1] * x + m[5] * y + m[9] * z;
  float uz = m[2] * x + m[6] * y + m[10] * z;
  float l = sqrtf(ux * ux + uy * uy + uz * uz);
  f4 o = {ux + m[12], uy + m[13], uz + m[14], l == 0 ? 0 : -uz / l};
  out[(size_t)blockIdx.y * n + v] = o;
}

__global__ __launch_bounds__(256) void diff_kernel(const unsigned *a, const unsigned *b, size_t n, unsigned long long *hist) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && a[i] != b[i]) atomicAdd(&hist[(i / 4) & 63], 1ull);  // lane of the producing thread (n % 64 == 0)
}

// ---- aggressors (one wave per SIMD per workgroup, <= 64 VGPRs, no LDS unless stated) ----
__global__ __launch_bounds__(256) void agg_mfma(float *out, int iters) {
  h8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.01f * (threadIdx.x + i)); b[i] = (_Float16)(0.02f * (threadIdx.x - i)); }
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int i = 0; i < iters; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
  float s = 0;
  for (int j = 0; j < 4; j++) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 1234.5f) out[threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void agg_valu(float *out, int iters) {
  float x = threadIdx.x * 0.001f, y = 1.0001f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) x = __builtin_fmaf(x, y, 0.5f);
  }
  if (x == 1234.5f) out[threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void agg_exp(float *out, int iters) {
  float x = threadIdx.x * 0.001f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) x = __builtin_amdgcn_exp2f(x) * 0.25f;
  }
  if (x == 1234.5f) out[threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void agg_lds(float *out, int iters) {
  __shared__ f4 buf[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = (f4){1, 2, 3, 4};
  __syncthreads();
  f4 s = {0, 0, 0, 0};
  int idx = threadIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) { s += buf[idx]; idx = (idx + 257) & 1023; }
  }
  if (s[0] == 1234.5f) out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void agg_mem(const f4 *src, float *out, size_t n, int iters) {
  f4 s = {0, 0, 0, 0};
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (int k = 0; k < iters; k++) { s += src[i]; i = (i + 1048583) % n; }
  if (s[0] == 1234.5f) out[threadIdx.x] = s[0];
}

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 1.0;
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  unsigned long long *hist;
  float *sink;
  CK(hipMalloc(&hist, 64 * 8));
  CK(hipMalloc(&sink, 4096));
  const size_t memn = (size_t)64 << 20;
  f4 *mem;
  CK(hipMalloc(&mem, memn * 16));
  CK(hipMemset(mem, 0, memn * 16));
  // compiled-victim data
  const int NP = 2560, NM = 64;
  std::vector<float> hm(NM * 16), hp(NP * 3);
  unsigned st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto &v : hm) v = rnd();
  for (auto &v : hp) v = rnd();
  float *dm, *dp;
  f4 *dref, *dout;
  CK(hipMalloc(&dm, hm.size() * 4)); CK(hipMalloc(&dp, hp.size() * 4));
  CK(hipMalloc(&dref, (size_t)NM * NP * 16)); CK(hipMalloc(&dout, (size_t)NM * NP * 16));
  CK(hipMemcpy(dm, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dp, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(victim_c_kernel, dim3(NP / 256, NM), dim3(256), 0, sa, dm, dp, NP, dref);
  CK(hipStreamSynchronize(sa));

  const char *agg_names[] = {"none", "mfma", "valu", "exp", "lds", "mem"};
  const char *vic_names[] = {"v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "compiled (SLP packed math)"};
  for (int agg = 0; agg < 6; agg++) {
    for (int vic = 0; vic < 4; vic++) {
      CK(hipMemset(hist, 0, 64 * 8));
      auto t0 = std::chrono::steady_clock::now();
      long launches = 0;
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        // keep ~2 aggressor launches queued (each a few ms), 512 workgroups = 2 per CU
        switch (agg) {
          case 1: hipLaunchKernelGGL(agg_mfma, dim3(512), dim3(256), 0, sb, sink, 40000); break;
          case 2: hipLaunchKernelGGL(agg_valu, dim3(512), dim3(256), 0, sb, sink, 40000); break;
          case 3: hipLaunchKernelGGL(agg_exp, dim3(512), dim3(256), 0, sb, sink, 10000); break;
          case 4: hipLaunchKernelGGL(agg_lds, dim3(512), dim3(256), 0, sb, sink, 20000); break;
          case 5: hipLaunchKernelGGL(agg_mem, dim3(2048), dim3(256), 0, sb, mem, sink, memn, 2000); break;
          default: break;
        }
        for (int k = 0; k < 50; k++, launches++) {
          const unsigned seed = (unsigned)launches * 7919u;
          if (vic == 0) hipLaunchKernelGGL(victim_kernel<0>, dim3(640), dim3(256), 0, sa, hist, seed, 64);
          else if (vic == 1) hipLaunchKernelGGL(victim_kernel<1>, dim3(640), dim3(256), 0, sa, hist, seed, 64);
          else if (vic == 2) hipLaunchKernelGGL(victim_kernel<2>, dim3(640), dim3(256), 0, sa, hist, seed, 64);
          else {
            hipLaunchKernelGGL(victim_c_kernel, dim3(NP / 256, NM), dim3(256), 0, sa, dm, dp, NP, dout);
            hipLaunchKernelGGL(diff_kernel, dim3((unsigned)(((size_t)NM * NP * 4 + 255) / 256)), dim3(256), 0, sa, (const unsigned *)dout,
                               (const unsigned *)dref, (size_t)NM * NP * 4, hist);
          }
        }
        CK(hipStreamSynchronize(sa));
        if (agg) CK(hipStreamSynchronize(sb));
      }
      unsigned long long h[64];
      CK(hipMemcpy(h, hist, sizeof(h), hipMemcpyDeviceToHost));
      unsigned long long tot = 0, q[4] = {0, 0, 0, 0};
      for (int l = 0; l < 64; l++) { tot += h[l]; q[l >> 4] += h[l]; }
      printf("aggressor %-5s victim %-28s launches %7ld mismatches %8llu  by lane quarter [0-15 16-31 32-47 48-63]: %llu %llu %llu %llu\n",
             agg_names[agg], vic_names[vic], launches, tot, q[0], q[1], q[2], q[3]);
      fflush(stdout);
    }
  }
  return 0;
}
