// Test double for librccl (tests only; built to tests/fake_rccl/_build/librccl.so.1 by tests/test_sharded_native_gpu.py).
//
// libfoundationpose_amd.so binds RCCL at first use with dlopen / dlsym of ncclAllGather, ncclGetErrorString, ncclCommCount,
// ncclCommUserRank and (optional) ncclCommAbort (fp_api.hip: rccl_api).  This library exports those plus ncclCommInitAll /
// ncclCommDestroy for communicators whose ranks are THREADS of one process with their models on ONE device, so that the native
// sharded Register (fp_register_sharded: slice arithmetic, ragged last shard, NaN-poisoned rows of a failing rank, stream ordering
// around the collective) executes at world 2 and 8 on the single GPU a test box has.
//
// ncclAllGather(send, recv, count, f32, comm, stream), stream-ordered like the real one:
//   1. the caller records `ready` on its stream (its send rows are complete behind it) and meets the other ranks at a host barrier;
//   2. it makes its stream wait for every peer's `ready` and enqueues one device-to-device copy per peer into its own recv buffer;
//   3. it records `copied`, meets the others again, and makes its stream wait for every peer's `copied`: nobody overwrites its send
//      buffer (the next Register) while a peer's copy of it is still in flight.
// A barrier gives up after FAKE_RCCL_TIMEOUT_S seconds (default 60) or when the communicator was aborted: the call returns an error
// instead of hanging, which is also what the tests assert for a rank that never arrives.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace {
struct Slot {
  const void *send = nullptr;
  hipEvent_t ready = nullptr, copied = nullptr;
};
struct Group {
  int world = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long generation = 0;
  bool aborted = false;
  int refs = 0;
  std::vector<Slot> slots;
};
struct Comm {
  Group *g = nullptr;
  int rank = 0;
};
enum { kSuccess = 0, kUnhandledHipError = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5, kRemoteError = 6 };

// all ranks of the group meet here; false = aborted or timed out (the group is then marked aborted for everyone)
bool barrier(Group *g) {
  static const int timeout_s = [] { const char *e = std::getenv("FAKE_RCCL_TIMEOUT_S"); return e && *e ? std::atoi(e) : 60; }();
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->aborted) return false;
  const unsigned long gen = g->generation;
  if (++g->arrived == g->world) {
    g->arrived = 0;
    g->generation++;
    g->cv.notify_all();
    return true;
  }
  const bool ok = g->cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return g->generation != gen || g->aborted; });
  if (!ok || g->aborted) {
    g->aborted = true;
    g->cv.notify_all();
    return false;
  }
  return true;
}
}  // namespace

extern "C" {

int ncclCommInitAll(void **comms, int ndev, const int * /*devlist: every rank on the current device*/) {
  if (!comms || ndev < 1) return kInvalidArgument;
  Group *g = new Group();
  g->world = ndev;
  g->refs = ndev;
  g->slots.resize(ndev);
  for (int r = 0; r < ndev; r++) {
    if (hipEventCreateWithFlags(&g->slots[r].ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->slots[r].copied, hipEventDisableTiming) != hipSuccess)
      return kUnhandledHipError;
    Comm *c = new Comm();
    c->g = g;
    c->rank = r;
    comms[r] = c;
  }
  return kSuccess;
}

int ncclCommDestroy(void *comm) {
  Comm *c = (Comm *)comm;
  if (!c) return kInvalidArgument;
  Group *g = c->g;
  bool last;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    last = --g->refs == 0;
  }
  if (last) {
    for (Slot &s : g->slots) { (void)hipEventDestroy(s.ready); (void)hipEventDestroy(s.copied); }
    delete g;
  }
  delete c;
  return kSuccess;
}

int ncclCommAbort(void *comm) {
  Comm *c = (Comm *)comm;
  if (!c) return kInvalidArgument;
  std::lock_guard<std::mutex> lk(c->g->mu);
  c->g->aborted = true;
  c->g->cv.notify_all();
  return kSuccess;
}

int ncclCommCount(void *comm, int *count) {
  if (!comm || !count) return kInvalidArgument;
  *count = ((Comm *)comm)->g->world;
  return kSuccess;
}
int ncclCommUserRank(void *comm, int *rank) {
  if (!comm || !rank) return kInvalidArgument;
  *rank = ((Comm *)comm)->rank;
  return kSuccess;
}
const char *ncclGetErrorString(int rc) {
  switch (rc) {
    case kSuccess: return "no error";
    case kUnhandledHipError: return "unhandled hip error (fake rccl)";
    case kInvalidArgument: return "invalid argument (fake rccl)";
    case kRemoteError: return "remote process exited or there was a network error (fake rccl: a rank aborted the communicator or never arrived)";
    default: return "internal error (fake rccl)";
  }
}

int ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, int datatype, void *comm, hipStream_t stream) {
  Comm *c = (Comm *)comm;
  if (!c || !sendbuff || !recvbuff || datatype != 7 /* ncclFloat32 */) return kInvalidArgument;
  Group *g = c->g;
  const size_t bytes = sendcount * sizeof(float);
  Slot &mine = g->slots[c->rank];
  mine.send = sendbuff;
  if (hipEventRecord(mine.ready, stream) != hipSuccess) return kUnhandledHipError;
  if (!barrier(g)) return kRemoteError;
  for (int p = 0; p < g->world; p++) {
    if (hipStreamWaitEvent(stream, g->slots[p].ready, 0) != hipSuccess) return kUnhandledHipError;
    if (hipMemcpyAsync((char *)recvbuff + (size_t)p * bytes, g->slots[p].send, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return kUnhandledHipError;
  }
  if (hipEventRecord(mine.copied, stream) != hipSuccess) return kUnhandledHipError;
  if (!barrier(g)) return kRemoteError;
  for (int p = 0; p < g->world; p++)
    if (p != c->rank && hipStreamWaitEvent(stream, g->slots[p].copied, 0) != hipSuccess) return kUnhandledHipError;
  // third meeting: nobody re-records its events (the next collective) before every rank has enqueued its waits on them
  if (!barrier(g)) return kRemoteError;
  return kSuccess;
}

}  // extern "C"
