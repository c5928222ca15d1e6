// MEASUREMENT STAND-IN, not a transport: ONE process plays rank `rank` of an n-rank job alone, so that the production
// sharded code path (row shard of the matrix, own / remote column blocks, chunked gather, n/N-sized Gram-Schmidt
// sweeps) can be TIMED at the shard shapes of BASELINE config 4 on the pool's single-GPU boxes
// (tools/shard_compute_probe.py).  The "collectives" keep the numbers finite and statistically plausible and nothing
// else: all_gather copies the caller's own piece into every rank's slot (device-to-device), all_reduce multiplies by
// the number of ranks (as if every rank had contributed the same partial sums), halo_exchange hands the caller's own
// boundary planes back.  Results computed through it are NOT those of the real problem.  Loaded through
// LL_COMM_PLUGIN like tests/transport/shm_transport.cpp; the 8-byte all-gather of ll_comm_init's self-check is
// answered with the tags the check expects.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "lanczos_hip.h"
#include "lanczos_hip_transport.h"

namespace {
struct Solo {
  int rank = 0, nranks = 1;
};
__global__ void scale_doubles(double* p, size_t count, double f) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) p[i] *= f;
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
  for (int r = 0; r < c->nranks; ++r)
    if (hipMemcpyAsync((char*)recv + (size_t)r * bytes, send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
  return 0;
}
int solo_all_reduce(void* self, double* buf, size_t count, void* stream) {
  Solo* c = (Solo*)self;
  if (count == 0) return 0;
  hipLaunchKernelGGL(scale_doubles, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, buf, count, (double)c->nranks);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
int solo_halo(void*, const void* send_prev, void* recv_prev, int prev, const void* send_next, void* recv_next, int next,
              size_t bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (prev >= 0 && hipMemcpyAsync(recv_prev, send_next, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
  if (next >= 0 && hipMemcpyAsync(recv_next, send_prev, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
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
