// How fast can the chip gather 8-byte elements?  150 M random int32 indices stream in (coalesced, nt), every lane
// gathers the indexed doubles from a table of T bytes and sums them.  T sweeps from L1-sized to beyond the Infinity
// Cache: the rate per memory level bounds every gather-based SpMV.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                     \
  do {                                                            \
    hipError_t e = (x);                                           \
    if (e != hipSuccess) {                                        \
      std::printf("%s: %s\n", #x, hipGetErrorString(e));          \
      return 1;                                                   \
    }                                                             \
  } while (0)

template <int U>
__global__ __launch_bounds__(256) void gather_sum(const int4* __restrict__ idx, size_t n4, const double* __restrict__ x,
                                                  double* out) {
  double acc = 0.0;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
    int4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      typedef int v4i __attribute__((ext_vector_type(4)));
      const v4i t = i + (size_t)u * 256 < n4 ? __builtin_nontemporal_load(reinterpret_cast<const v4i*>(idx + i + (size_t)u * 256))
                                              : v4i{0, 0, 0, 0};
      q[u] = int4{t.x, t.y, t.z, t.w};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += (x[q[u].x] + x[q[u].y]) + (x[q[u].z] + x[q[u].w]);
  }
  if (acc == 1.2345e-300) out[0] = acc;
}

int main() {
  const size_t n = (size_t)150 << 20;  // gathers per launch
  std::vector<int32_t> h(n);
  int32_t* d_idx;
  double *d_x, *d_out;
  const size_t max_elems = (size_t)512 << 17;  // 512 MiB table
  CK(hipMalloc(&d_idx, n * 4));
  CK(hipMalloc(&d_x, max_elems * 8));
  CK(hipMalloc(&d_out, 8));
  CK(hipMemset(d_x, 0, max_elems * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (double mb : {0.015625, 0.5, 1.0, 2.0, 3.0, 4.0, 8.0, 32.0, 80.0, 200.0, 512.0}) {
    const uint64_t elems = (uint64_t)(mb * 131072.0);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; ++i) {
      s ^= s << 13;
      s ^= s >> 7;
      s ^= s << 17;
      h[i] = (int32_t)((s >> 11) % elems);
    }
    CK(hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice));
    for (int grid : {1024, 2048, 4096}) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(gather_sum<2>, dim3(grid), dim3(256), 0, 0, (const int4*)d_idx, n / 4, d_x, d_out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = ms < best ? ms : best;
      }
      std::printf("table %8.3f MiB grid %4d: %.3f ms  %.0f G gathers/s\n", mb, grid, best, n / best / 1e6);
    }
  }
  return 0;
}
