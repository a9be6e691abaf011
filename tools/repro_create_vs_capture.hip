// Stand-alone stress for the host crash of DESIGN.md section 9 -- NO library, NO PyTorch: only the HIP runtime this file is linked
// against (the system ROCm; `ldd` of the binary says which).  It recreates every ingredient the crash hunt named, each on its own
// host thread of one process:
//   * servers   (2 threads): capture a 24-kernel graph once, then replay it + hipMemcpyAsync the result back + synchronise, forever;
//   * creator   (1 thread) : what fp_create does -- a fresh stream, a few hundred hipMalloc, hipMemcpyAsync uploads from pageable AND
//                            pinned sources, a synchronise, hipFree of everything, stream destroyed;
//   * foreign   (1 thread) : an unrelated stream that is never idle (the role PyTorch's stream had);
//   * transient (1 thread) : spawns short-lived threads that create a stream, copy through it, destroy it and EXIT (the faulting
//                            address of the crash lay in the stack / TLS mapping of a thread that was gone).
// It runs for `seconds` (default 30) or until `creations` creator rounds (default 10000), prints the counts and exits 0; a SIGSEGV
// inside the runtime ends it the hard way, which is the reproduction.
//   hipcc --offload-arch=gfx950 -O2 -pthread tools/repro_create_vs_capture.hip -o tools/_bin/repro_create_vs_capture
//   tools/_bin/repro_create_vs_capture [seconds] [creations]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                                      \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) {                                                                        \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));          \
      exit(2);                                                                                     \
    }                                                                                              \
  } while (0)

__global__ void axpy_kernel(float *y, const float *x, float a, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i] + y[i] * 0.5f;
}
__global__ void spin_kernel(float *y, int n, int rounds) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = y[i];
  for (int r = 0; r < rounds; r++) v = v * 1.0001f + 0.25f;
  y[i] = v;
}

static std::atomic<bool> g_stop{false};
static std::atomic<long> g_replays{0}, g_creations{0}, g_foreign{0}, g_transients{0}, g_mismatch{0};

static void server(int id) {
  CK(hipSetDevice(0));
  const int n = 1 << 16;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float *x, *y, *hy;
  CK(hipMalloc(&x, n * sizeof(float)));
  CK(hipMalloc(&y, n * sizeof(float)));
  CK(hipHostMalloc(&hy, n * sizeof(float)));
  std::vector<float> hx(n, 1.0f + id);
  CK(hipMemcpyAsync(x, hx.data(), n * sizeof(float), hipMemcpyHostToDevice, s));
  CK(hipStreamSynchronize(s));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  CK(hipMemsetAsync(y, 0, n * sizeof(float), s));
  for (int k = 0; k < 24; k++) hipLaunchKernelGGL(axpy_kernel, dim3(n / 256), dim3(256), 0, s, y, x, 1.0f + k, n);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  float first = 0.f;
  bool have = false;
  while (!g_stop.load(std::memory_order_relaxed)) {
    CK(hipGraphLaunch(ge, s));
    CK(hipMemcpyAsync(hy, y, n * sizeof(float), hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    if (!have) { first = hy[n / 2]; have = true; }
    else if (hy[n / 2] != first || hy[0] != first) g_mismatch++;
    g_replays++;
  }
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  CK(hipFree(x)); CK(hipFree(y)); CK(hipHostFree(hy));
  CK(hipStreamDestroy(s));
}

static void creator(long limit) {
  CK(hipSetDevice(0));
  std::vector<char> pageable(4 << 20, 3);
  char *pinned;
  CK(hipHostMalloc(&pinned, 4 << 20));
  memset(pinned, 5, 4 << 20);
  while (!g_stop.load(std::memory_order_relaxed)) {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<void *> bufs;
    for (int i = 0; i < 300; i++) {
      size_t bytes = (size_t)(1 + (i * 37) % 97) * 4096;
      void *p;
      CK(hipMalloc(&p, bytes));
      bufs.push_back(p);
      const char *src = (i & 1) ? pinned : pageable.data();
      CK(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, s));
      if ((i & 31) == 31) CK(hipStreamSynchronize(s));
    }
    CK(hipStreamSynchronize(s));
    for (void *p : bufs) CK(hipFree(p));
    CK(hipStreamDestroy(s));
    if (++g_creations >= limit) g_stop = true;
  }
  CK(hipHostFree(pinned));
}

static void foreign() {
  CK(hipSetDevice(0));
  const int n = 1 << 20;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float *y;
  CK(hipMalloc(&y, n * sizeof(float)));
  CK(hipMemsetAsync(y, 0, n * sizeof(float), s));
  while (!g_stop.load(std::memory_order_relaxed)) {
    for (int k = 0; k < 16; k++) hipLaunchKernelGGL(spin_kernel, dim3(n / 256), dim3(256), 0, s, y, n, 200);
    CK(hipStreamSynchronize(s));
    g_foreign += 16;
  }
  CK(hipFree(y));
  CK(hipStreamDestroy(s));
}

static void transient_spawner() {
  while (!g_stop.load(std::memory_order_relaxed)) {
    std::thread t([] {
      CK(hipSetDevice(0));
      hipStream_t s;
      CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      void *p;
      CK(hipMalloc(&p, 1 << 16));
      char host[4096];   // a source on THIS thread's stack
      memset(host, 7, sizeof host);
      CK(hipMemcpyAsync(p, host, sizeof host, hipMemcpyHostToDevice, s));
      CK(hipStreamSynchronize(s));
      CK(hipFree(p));
      CK(hipStreamDestroy(s));
    });
    t.join();
    g_transients++;
  }
}

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 30.0;
  const long limit = argc > 2 ? atol(argv[2]) : 10000;
  int rt = 0;
  CK(hipRuntimeGetVersion(&rt));
  printf("HIP runtime %d, %.0f s or %ld creations\n", rt, seconds, limit);
  fflush(stdout);
  std::vector<std::thread> th;
  th.emplace_back(server, 0);
  th.emplace_back(server, 1);
  th.emplace_back(foreign);
  th.emplace_back(transient_spawner);
  th.emplace_back(creator, limit);
  const auto t0 = std::chrono::steady_clock::now();
  while (!g_stop.load()) {
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) g_stop = true;
  }
  for (auto &t : th) t.join();
  printf("clean: %ld graph replays, %ld creation rounds (300 hipMalloc + hipMemcpyAsync each), %ld foreign kernels, %ld transient threads, "
         "%ld replay mismatches\n",
         g_replays.load(), g_creations.load(), g_foreign.load(), g_transients.load(), g_mismatch.load());
  return g_mismatch.load() ? 1 : 0;
}
