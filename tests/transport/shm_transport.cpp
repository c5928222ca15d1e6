// TEST INFRASTRUCTURE — a host-staged transport for the sharded hot path (include/lanczos_hip_transport.h), loaded
// through LL_COMM_PLUGIN.  It exists for ONE purpose: to run the sharded engine with several ranks on a single GPU,
// where RCCL refuses duplicate devices (the test boxes have one GPU).  Same collective semantics through a POSIX
// shared-memory segment (device -> host slot, barrier, host -> device); slow, deterministic (sums in rank order).
// It is not part of liblanczos_hip.so and production multi-GPU runs use RCCL.
//
// Every call blocks the calling host thread but synchronises ONLY the stream it was given, so work already enqueued
// on the context's other stream keeps running on the device meanwhile (the overlapped all-gather path of
// csrc/engine.cpp is exercised with real concurrency on the device side).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/lanczos_hip.h"
#include "../../include/lanczos_hip_transport.h"

namespace {

struct Seg {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  int nranks;
  int pad;
  size_t slot_bytes;
};

constexpr size_t kSlotBytes = (size_t)64 << 20;  // per rank; enough for the test problems

struct Shm {
  int rank = 0, nranks = 1;
  Seg* seg = nullptr;
  char* slots = nullptr;
  size_t map_bytes = 0;
  std::string name;
  bool failed = false;
  char* slot(int r) { return slots + (size_t)r * seg->slot_bytes; }
  bool barrier() {
    const int gen = seg->generation.load(std::memory_order_acquire);
    if (seg->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == nranks) {
      seg->arrived.store(0, std::memory_order_relaxed);
      seg->generation.fetch_add(1, std::memory_order_release);
      return true;
    }
    // bounded wait: a peer that died must not hang the others
    for (long spins = 0; seg->generation.load(std::memory_order_acquire) == gen; ++spins) {
      if (spins < 20000) continue;  // busy-wait first (collectives are latency sized), then back off
      usleep(20);
      if (spins > 1520000) {  // ~30 s
        std::fprintf(stderr, "shm transport: barrier timed out on rank %d (a peer rank is gone?)\n", rank);
        failed = true;
        return false;
      }
    }
    return true;
  }
};

#define SHM_HIP(expr)                                                                            \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      std::fprintf(stderr, "shm transport: %s failed: %s\n", #expr, hipGetErrorString(e_));      \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)

int all_gather(void* self, const void* send, void* recv, size_t bytes, void* stream) {
  Shm* c = (Shm*)self;
  hipStream_t s = (hipStream_t)stream;
  if (c->failed || bytes > c->seg->slot_bytes) return 2;
  SHM_HIP(hipMemcpyAsync(c->slot(c->rank), send, bytes, hipMemcpyDeviceToHost, s));
  SHM_HIP(hipStreamSynchronize(s));
  if (!c->barrier()) return 3;
  for (int r = 0; r < c->nranks; ++r)
    SHM_HIP(hipMemcpyAsync((char*)recv + (size_t)r * bytes, c->slot(r), bytes, hipMemcpyHostToDevice, s));
  SHM_HIP(hipStreamSynchronize(s));
  if (!c->barrier()) return 3;
  return 0;
}

int all_reduce(void* self, double* buf, size_t n, void* stream) {
  Shm* c = (Shm*)self;
  hipStream_t s = (hipStream_t)stream;
  const size_t bytes = n * sizeof(double);
  if (c->failed || bytes > c->seg->slot_bytes) return 2;
  SHM_HIP(hipMemcpyAsync(c->slot(c->rank), buf, bytes, hipMemcpyDeviceToHost, s));
  SHM_HIP(hipStreamSynchronize(s));
  if (!c->barrier()) return 3;
  std::vector<double> sum(n, 0.0);
  for (int r = 0; r < c->nranks; ++r) {  // rank order on every rank: identical bits everywhere
    const double* p = (const double*)c->slot(r);
    for (size_t i = 0; i < n; ++i) sum[i] += p[i];
  }
  SHM_HIP(hipMemcpyAsync(buf, sum.data(), bytes, hipMemcpyHostToDevice, s));
  SHM_HIP(hipStreamSynchronize(s));
  if (!c->barrier()) return 3;
  return 0;
}

int halo_exchange(void* self, const void* send_prev, void* recv_prev, int prev, const void* send_next, void* recv_next,
                  int next, size_t bytes, void* stream) {
  Shm* c = (Shm*)self;
  hipStream_t s = (hipStream_t)stream;
  if (c->failed || 2 * bytes > c->seg->slot_bytes) return 2;
  // own slot = [message for prev | message for next]
  if (prev >= 0) SHM_HIP(hipMemcpyAsync(c->slot(c->rank), send_prev, bytes, hipMemcpyDeviceToHost, s));
  if (next >= 0) SHM_HIP(hipMemcpyAsync(c->slot(c->rank) + bytes, send_next, bytes, hipMemcpyDeviceToHost, s));
  SHM_HIP(hipStreamSynchronize(s));
  if (!c->barrier()) return 3;
  if (prev >= 0) SHM_HIP(hipMemcpyAsync(recv_prev, c->slot(prev) + bytes, bytes, hipMemcpyHostToDevice, s));
  if (next >= 0) SHM_HIP(hipMemcpyAsync(recv_next, c->slot(next), bytes, hipMemcpyHostToDevice, s));
  SHM_HIP(hipStreamSynchronize(s));
  if (!c->barrier()) return 3;
  return 0;
}

void destroy(void* self) {
  Shm* c = (Shm*)self;
  if (c->seg) munmap((void*)c->seg, c->map_bytes);
  if (c->rank == 0) shm_unlink(c->name.c_str());
  delete c;
}

}  // namespace

extern "C" {

// The "id" is the name of the shared-memory segment.
int ll_transport_unique_id(void* id128) {
  std::random_device rd;
  char name[LL_UNIQUE_ID_BYTES] = {0};
  std::snprintf(name, sizeof(name), "/ll_shm_%d_%08x", (int)getpid(), (unsigned)rd());
  std::memcpy(id128, name, LL_UNIQUE_ID_BYTES);
  return 0;
}

int ll_transport_open(const void* id128, int rank, int nranks, int device, ll_transport* out) {
  if (hipSetDevice(device) != hipSuccess) return 1;
  Shm* c = new Shm;
  c->rank = rank;
  c->nranks = nranks;
  c->name.assign((const char*)id128, strnlen((const char*)id128, LL_UNIQUE_ID_BYTES - 1));
  c->map_bytes = 4096 + kSlotBytes * (size_t)nranks;
  // rank 0 owns the segment: it removes any stale one of the same name (a crashed earlier run) and creates it afresh
  // (zero-filled); the other ranks wait for it to appear with its final size
  int fd = -1;
  if (rank == 0) {
    shm_unlink(c->name.c_str());
    fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd >= 0 && ftruncate(fd, (off_t)c->map_bytes) != 0) {
      close(fd);
      fd = -1;
    }
  } else {
    for (int tries = 0; tries < 3000 && fd < 0; ++tries) {  // ~30 s
      fd = shm_open(c->name.c_str(), O_RDWR, 0600);
      if (fd >= 0) {
        off_t sz = lseek(fd, 0, SEEK_END);
        if (sz < (off_t)c->map_bytes) {
          close(fd);
          fd = -1;
        }
      }
      if (fd < 0) usleep(10000);
    }
  }
  if (fd < 0) {
    std::fprintf(stderr, "shm transport: cannot open %s\n", c->name.c_str());
    delete c;
    return 4;
  }
  void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    delete c;
    return 5;
  }
  c->seg = (Seg*)m;  // a fresh segment is zero-filled: arrived = generation = 0
  c->slots = (char*)m + 4096;
  c->seg->nranks = nranks;
  c->seg->slot_bytes = kSlotBytes;
  if (!c->barrier()) {  // rendezvous: everybody has mapped the segment once `arrived` has counted all ranks
    destroy(c);
    return 6;
  }
  out->self = c;
  out->all_gather = all_gather;
  out->all_reduce_sum_f64 = all_reduce;
  out->halo_exchange = halo_exchange;
  out->destroy = destroy;
  return 0;
}

}  // extern "C"
