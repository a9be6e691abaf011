// Micro-benchmark behind the Track design decision: what does a dependency between two tiny kernels cost
//   (a) as consecutive launches on a stream, (b) as nodes of a hipGraph, (c) as a grid-wide barrier inside ONE persistent kernel
// -- with the data hand-over a real layer boundary needs (each workgroup writes a slice, reads another workgroup's slice of the
// previous phase).   hipcc --offload-arch=gfx950 -O3 tools/bench_launch.hip -o /tmp/bench_launch && /tmp/bench_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int SLICE = 4096;  // floats per workgroup and phase (16 KB)

__global__ __launch_bounds__(256) void phase_kernel(const float *__restrict__ in, float *__restrict__ out, int nblk) {
  const int src = (blockIdx.x * 7 + 3) % nblk;
  for (int i = threadIdx.x; i < SLICE; i += 256) out[(size_t)blockIdx.x * SLICE + i] = in[(size_t)src * SLICE + i] * 1.0001f + 1.f;
}

// grid barrier: monotonically increasing counter, agent-scope atomics; MODE 0 = release/acquire fences (L2 write-back + invalidate),
// MODE 1 = no fences, data moved with agent-coherent (sc1) accesses instead
template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void persistent_kernel(float *a, float *b, unsigned *ctr, int phases, unsigned base) {
  const int nblk = gridDim.x;
  const int src = (blockIdx.x * 7 + 3) % nblk;
  float *in = a, *out = b;
  for (int p = 0; p < phases; p++) {
    for (int i = threadIdx.x; i < SLICE; i += 256) {
      float v;
      if (MODE == 1) v = __hip_atomic_load(in + (size_t)src * SLICE + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else v = in[(size_t)src * SLICE + i];
      v = v * 1.0001f + 1.f;
      if (MODE == 1) __hip_atomic_store(out + (size_t)blockIdx.x * SLICE + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else out[(size_t)blockIdx.x * SLICE + i] = v;
    }
    grid_barrier<MODE>(ctr, base + (unsigned)(p + 1) * nblk);
    float *t = in; in = out; out = t;
  }
}

int main() {
  const int PH = 64;
  hipStream_t s;
  OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int nblk : {64, 256, 512, 1024}) {
    float *a, *b; unsigned *ctr;
    OK(hipMalloc(&a, (size_t)nblk * SLICE * 4)); OK(hipMalloc(&b, (size_t)nblk * SLICE * 4)); OK(hipMalloc(&ctr, 4));
    OK(hipMemsetAsync(a, 0, (size_t)nblk * SLICE * 4, s)); OK(hipMemsetAsync(ctr, 0, 4, s));
    auto timeit = [&](auto fn, int reps) {
      fn(); hipStreamSynchronize(s);
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; r++) fn();
      hipStreamSynchronize(s);
      return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    };
    // (a) eager launches
    double eager = timeit([&] { for (int p = 0; p < PH; p++) hipLaunchKernelGGL(phase_kernel, dim3(nblk), dim3(256), 0, s, (p & 1) ? b : a, (p & 1) ? a : b, nblk); }, 20);
    // (b) graph
    hipGraph_t g; hipGraphExec_t ge;
    OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < PH; p++) hipLaunchKernelGGL(phase_kernel, dim3(nblk), dim3(256), 0, s, (p & 1) ? b : a, (p & 1) ? a : b, nblk);
    OK(hipStreamEndCapture(s, &g)); OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double graph = timeit([&] { hipGraphLaunch(ge, s); }, 50);
    double one = timeit([&] { hipLaunchKernelGGL(phase_kernel, dim3(nblk), dim3(256), 0, s, a, b, nblk); }, 50);
    // (c) persistent (nblk must be co-resident: 256 CUs x up to 8 workgroups of 256 threads)
    unsigned base = 0; double pers[2];
    for (int mode = 0; mode < 2; mode++) {
      OK(hipMemsetAsync(ctr, 0, 4, s)); base = 0;
      pers[mode] = timeit([&] {
        if (mode == 0) hipLaunchKernelGGL(persistent_kernel<0>, dim3(nblk), dim3(256), 0, s, a, b, ctr, PH, base);
        else hipLaunchKernelGGL(persistent_kernel<1>, dim3(nblk), dim3(256), 0, s, a, b, ctr, PH, base);
        base += (unsigned)PH * nblk;
      }, 50);
    }
    printf("workgroups %4d: eager %.2f us/phase, graph %.2f us/phase, single launch+sync %.1f us, persistent fence %.2f us/phase, persistent sc1 %.2f us/phase\n",
           nblk, eager / PH, graph / PH, one, pers[0] / PH, pers[1] / PH);
    hipFree(a); hipFree(b); hipFree(ctr);
  }
  return 0;
}
