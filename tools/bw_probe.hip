// HBM streaming ceilings on the box at hand: read-only (sum), copy and read+write-in-place (scale) kernels with 16-byte
// accesses per lane, swept over grid size and loads in flight per lane.  Context for DESIGN.md's "fraction of the
// measured ceiling" figures.   hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o /tmp/bw_probe && /tmp/bw_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e));                    \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

template <int U> __global__ __launch_bounds__(256) void read_sum(const double2* __restrict__ a, size_t n2, double* out) {
  double acc = 0.0;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n2; i += stride) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = i + (size_t)u * 256 < n2 ? a[i + (size_t)u * 256] : double2{0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
  }
  if (acc == 1.2345e-300) out[0] = acc;  // keep the loads alive
}
template <int U> __global__ __launch_bounds__(256) void copy_k(const double2* __restrict__ a, double2* __restrict__ b, size_t n2) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n2; i += stride) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = i + (size_t)u * 256 < n2 ? a[i + (size_t)u * 256] : double2{0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + (size_t)u * 256 < n2) b[i + (size_t)u * 256] = v[u];
  }
}

int main() {
  const size_t bytes = (size_t)4 << 30;  // 4 GiB per array
  const size_t n2 = bytes / sizeof(double2);
  double2 *a, *b;
  double* out;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 0, bytes));
  CK(hipMemset(b, 0, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time = [&](auto launch) {
    launch();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
  };
  for (int grid : {512, 1024, 2048, 4096, 8192, 16384}) {
    float r1 = time([&] { hipLaunchKernelGGL(read_sum<1>, dim3(grid), dim3(256), 0, 0, a, n2, out); });
    float r4 = time([&] { hipLaunchKernelGGL(read_sum<4>, dim3(grid), dim3(256), 0, 0, a, n2, out); });
    float r8 = time([&] { hipLaunchKernelGGL(read_sum<8>, dim3(grid), dim3(256), 0, 0, a, n2, out); });
    float c4 = time([&] { hipLaunchKernelGGL(copy_k<4>, dim3(grid), dim3(256), 0, 0, a, b, n2); });
    std::printf("grid %5d: read U1 %.0f GB/s, U4 %.0f, U8 %.0f | copy U4 %.0f GB/s (read+write)\n", grid, bytes / r1 / 1e6,
                bytes / r4 / 1e6, bytes / r8 / 1e6, 2.0 * bytes / c4 / 1e6);
  }
  return 0;
}
