// Micro-benchmark behind conv_smallm_kernel's weight layout: what does ONE global_load_dwordx4 wave
// instruction cost in the vector L1 (TCP) as a function of its address shape?  All data is L2-resident (a few MB re-read by
// every workgroup), one workgroup of 4 waves per CU, NLOAD independent loads in flight per wave and iteration.
//   shape 0  MFMA-operand shaped, row pitch 1024 B: lane -> (row = lane & 15, 16-byte slot = lane >> 4): 16 lines, 64 B of each
//   shape 1  the same with row pitch 128 B (a [K-step][row][128 B] weight layout): 16 half lines, consecutive
//   shape 2  fragment order: lane -> lane * 16: 1 KB contiguous, 8 whole 128-byte lines
//   shape 3  two instructions per 16 rows x 128 B tile, each 8 rows x 128 B (lane -> row = lane >> 3, slot = lane & 7)
//   hipcc --offload-arch=gfx950 -O3 tools/bench_tcp.hip -o tools/_bin/bench_tcp && tools/_bin/bench_tcp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int i4 __attribute__((ext_vector_type(4)));
constexpr int NLOAD = 8;

template <int SHAPE>
__global__ __launch_bounds__(256) void tcp_kernel(const unsigned char *__restrict__ buf, size_t bytes, int iters, int *sink, unsigned long long *clk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  size_t lane_off, step;  // byte offset of the lane inside one instruction's footprint, bytes between consecutive instructions
  if (SHAPE == 0) { lane_off = (size_t)(lane & 15) * 1024 + (lane >> 4) * 16; step = 64; }   // (NLOAD = 8: one 512-byte run of 16 rows per iteration)
  else if (SHAPE == 1) { lane_off = (size_t)(lane & 15) * 128 + (lane >> 4) * 16; step = 2048; }
  else if (SHAPE == 2) { lane_off = (size_t)lane * 16; step = 1024; }
  else { lane_off = (size_t)(lane >> 3) * 128 + (lane & 7) * 16; step = 1024; }
  // the four waves of a CU walk four different 64 KB regions (no L1 sharing between them); 16 regions = 1 MB in total, so every
  // L2 holds all of it after the first touch and the measurement is L2 -> L1, not the fabric
  const size_t region = (size_t)((blockIdx.x * 4 + wave) & 15) * 65536;
  size_t pos = 0;
  i4 acc = (i4){0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    i4 v[NLOAD];
#pragma unroll
    for (int j = 0; j < NLOAD; j++) {
      size_t o;
      if (SHAPE == 0) o = pos + lane_off + (size_t)(j & 7) * 64 + (size_t)(j >> 3) * 16384;   // 8 k-steps along a 1024-byte row, then the next 16 rows
      else o = pos + lane_off + (size_t)j * step;
      v[j] = *reinterpret_cast<const i4 *>(buf + region + (o & 65535));
    }
#pragma unroll
    for (int j = 0; j < NLOAD; j++) acc ^= v[j];
    pos += (SHAPE == 0) ? (NLOAD >= 16 ? 16384 * (NLOAD / 8) : 512) : NLOAD * step;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (acc[0] == 0x12345678 && acc[1] == 0x9abcdef0) sink[0] = acc[2] + acc[3];
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
  const size_t bytes = 8u << 20;
  unsigned char *buf; int *sink; unsigned long long *clk;
  OK(hipMalloc(&buf, bytes)); OK(hipMalloc(&sink, 4)); OK(hipMalloc(&clk, 256 * 8));
  OK(hipMemset(buf, 1, bytes));
  const int iters = 400;
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  for (int shape = 0; shape < 4; shape++) {
    for (int rep = 0; rep < 3; rep++) {
      OK(hipEventRecord(e0));
      if (shape == 0) hipLaunchKernelGGL(tcp_kernel<0>, dim3(256), dim3(256), 0, 0, buf, bytes, iters, sink, clk);
      if (shape == 1) hipLaunchKernelGGL(tcp_kernel<1>, dim3(256), dim3(256), 0, 0, buf, bytes, iters, sink, clk);
      if (shape == 2) hipLaunchKernelGGL(tcp_kernel<2>, dim3(256), dim3(256), 0, 0, buf, bytes, iters, sink, clk);
      if (shape == 3) hipLaunchKernelGGL(tcp_kernel<3>, dim3(256), dim3(256), 0, 0, buf, bytes, iters, sink, clk);
      OK(hipEventRecord(e1)); OK(hipEventSynchronize(e1));
      float ms; OK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> h(256);
      OK(hipMemcpy(h.data(), clk, 256 * 8, hipMemcpyDeviceToHost));
      double avg = 0; for (auto c : h) avg += (double)c; avg /= 256;
      const double instr_per_cu = 4.0 * iters * NLOAD;   // wave instructions a CU's L1 served
      printf("shape %d: %.1f us, %.0f clk per workgroup, %.1f clk per wave-instruction per CU, %.1f B/clk/CU\n", shape, ms * 1e3, avg, avg / instr_per_cu, instr_per_cu * 1024 / avg);
    }
  }
  return 0;
}
