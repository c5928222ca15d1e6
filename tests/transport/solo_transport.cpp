// MEASUREMENT STAND-IN, not a transport: ONE process plays rank `rank` of an n-rank job alone, so that the production
// sharded code path (row shard of the matrix, own / remote column blocks, chunked gather, n/N-sized Gram-Schmidt
// sweeps) can be TIMED at the shard shapes of BASELINE config 4 on the pool's single-GPU boxes
// (tools/shard_compute_probe.py).  The "collectives" act as if every OTHER rank held a zero shard: all_gather writes the
// caller's own piece into its slot and zeros into the other ranks' slots (the same bytes land in local memory as with
// real peers), all_reduce leaves the sums as they are (one small kernel stands for it), halo_exchange delivers zero
// planes.  What runs is therefore a consistent Lanczos process on the DIAGONAL block of the shard (symmetric; the
// one-sweep Gram-Schmidt form relies on the recurrence being real) at the full shard's cost: every stored entry is
// still multiplied, most of them by zero.  Results are NOT those of the real problem.  Loaded through
// LL_COMM_PLUGIN like tests/transport/shm_transport.cpp; the 8-byte all-gather and the sum of ones of ll_comm_init's
// self-check are answered with what the check expects.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "lanczos_hip.h"
#include "lanczos_hip_transport.h"

namespace {
struct Solo {
  int rank = 0, nranks = 1;
  bool self_check_answered = false;
};
__global__ void scale_doubles(double* p, size_t count, double f) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) p[i] *= f;
}
__global__ void replicate16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, int copies, int own) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    for (int r = 0; r < copies; ++r) dst[(size_t)r * n16 + i] = r == own ? v : uint4{0u, 0u, 0u, 0u};
  }
}
int solo_all_gather(void* self, const void* send, void* recv, size_t bytes, void* stream) {
  Solo* c = (Solo*)self;
  hipStream_t s = (hipStream_t)stream;
  if (bytes == sizeof(double)) {  // the self-check's rank tags (capi.cpp: finish_comm_setup)
    std::vector<double> tags((size_t)c->nranks);
    for (int r = 0; r < c->nranks; ++r) tags[(size_t)r] = (double)(r + 1);
    if (hipMemcpyAsync(recv, tags.data(), tags.size() * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess) return 1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : 1;
  }
  // ONE kernel that reads the caller's piece once and writes it into every rank's slot (round 2 issued one
  // hipMemcpyAsync per rank: 16 copies of ~6 us per SpMV at N = 8, i.e. ~100 us of stand-in cost serialised in front of
  // the remote-chunk launches — more than the kernels being measured)
  if ((bytes & 15) == 0 && (((uintptr_t)send | (uintptr_t)recv) & 15) == 0) {
    const size_t n16 = bytes / 16;
    const unsigned grid = (unsigned)std::min<size_t>(2048, (n16 + 255) / 256);
    hipLaunchKernelGGL(replicate16, dim3(grid ? grid : 1), dim3(256), 0, s, (const uint4*)send, (uint4*)recv, n16, c->nranks, c->rank);
    return hipGetLastError() == hipSuccess ? 0 : 1;
  }
  for (int r = 0; r < c->nranks; ++r) {
    char* dst = (char*)recv + (size_t)r * bytes;
    const hipError_t e = r == c->rank ? hipMemcpyAsync(dst, send, bytes, hipMemcpyDeviceToDevice, s) : hipMemsetAsync(dst, 0, bytes, s);
    if (e != hipSuccess) return 1;
  }
  return 0;
}
int solo_all_reduce(void* self, double* buf, size_t count, void* stream) {
  Solo* c = (Solo*)self;
  if (count == 0) return 0;
  double f = 1.0;
  if (!c->self_check_answered) {  // the first call is the self-check's sum of ones (capi.cpp: finish_comm_setup)
    c->self_check_answered = true;
    f = (double)c->nranks;
  }
  hipLaunchKernelGGL(scale_doubles, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, buf, count, f);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
int solo_halo(void*, const void* send_prev, void* recv_prev, int prev, const void* send_next, void* recv_next, int next,
              size_t bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  (void)send_prev;
  (void)send_next;
  if (prev >= 0 && hipMemsetAsync(recv_prev, 0, bytes, s) != hipSuccess) return 1;
  if (next >= 0 && hipMemsetAsync(recv_next, 0, bytes, s) != hipSuccess) return 1;
  return 0;
}
void solo_destroy(void* self) { delete (Solo*)self; }
}  // namespace

extern "C" {
int ll_transport_unique_id(void* id128) {
  std::memset(id128, 0, LL_UNIQUE_ID_BYTES);
  std::memcpy(id128, "solo", 4);
  return 0;
}
int ll_transport_open(const void*, int rank, int nranks, int device, ll_transport* out) {
  if (hipSetDevice(device) != hipSuccess) return 1;
  Solo* c = new Solo;
  c->rank = rank;
  c->nranks = nranks;
  out->self = c;
  out->all_gather = solo_all_gather;
  out->all_reduce_sum_f64 = solo_all_reduce;
  out->halo_exchange = solo_halo;
  out->destroy = solo_destroy;
  return 0;
}
}
