// Why does one trip (16 basis strips, 16 B per lane each, vectors 80 KB apart) of the small-vector Gram-Schmidt kernels
// cost ~2 us per wave at n = 1e4?  One wave per workgroup reads `k` strips of 1 KB (stride = one vector) in trips of 16
// loads per lane; prints us per trip for: data last written by ANOTHER kernel (the Lanczos situation: every kernel starts
// with nothing of it in its XCD's L2), the same kernel re-reading (L2-warm), grids of 20 / 79 / 256 / 1024 workgroups,
// 8 or 16 or 32 loads in flight, and 256-lane workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/small_strip_probe.hip -o tools/_build/small_strip_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));     \
      return 1;                                                      \
    }                                                                \
  } while (0)

template <int JB>
__global__ void read_strips(const double* __restrict__ basis, long ld, int k, int passes, double* __restrict__ out) {
  const long i0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;  // 16 B per lane
  double acc0 = 0, acc1 = 0;
  for (int p = 0; p < passes; ++p)
    for (int j = 0; j + JB <= k; j += JB) {
      double2 u[JB];
#pragma unroll
      for (int b = 0; b < JB; ++b) u[b] = *reinterpret_cast<const double2*>(basis + (long)(j + b) * ld + i0);
#pragma unroll
      for (int b = 0; b < JB; ++b) { acc0 += u[b].x; acc1 += u[b].y; }
    }
  if (acc0 + acc1 == 1.2345e300) out[0] = acc0;
}
__global__ void touch(double* basis, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) basis[i] = 1.0 + 1e-9 * (double)(i & 1023);
}

int main() {
  const int k = 96;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double* out;
  CK(hipMalloc(&out, 8));
  for (long n : {10240L, 102400L}) {
    const long ld = n;
    double* basis;
    CK(hipMalloc(&basis, (size_t)k * ld * 8));
    std::printf("== n = %ld doubles per vector, k = %d vectors (%.1f MB)\n", n, k, k * ld * 8 / 1e6);
    auto run = [&](const char* name, int threads, int jb, int passes, bool rewrite) -> int {
      const int grid = (int)(n / 2 / threads);
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        if (rewrite) hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, 0, basis, (long)k * ld);  // another kernel writes the basis
        CK(hipEventRecord(e0));
        if (jb == 8) hipLaunchKernelGGL(read_strips<8>, dim3(grid), dim3(threads), 0, 0, basis, ld, k, passes, out);
        else if (jb == 16) hipLaunchKernelGGL(read_strips<16>, dim3(grid), dim3(threads), 0, 0, basis, ld, k, passes, out);
        else hipLaunchKernelGGL(read_strips<32>, dim3(grid), dim3(threads), 0, 0, basis, ld, k, passes, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      const double trips = (double)passes * (k / jb);
      std::printf("  %-64s grid %5d: kernel %7.2f us, %5.2f us per trip, %6.1f GB/s\n", name, grid, best * 1e3, best * 1e3 / trips,
                  (double)passes * k * n * 8 / (best * 1e-3) / 1e9);
      return 0;
    };
    if (run("1 wave/WG, 16 in flight, 1 pass, basis rewritten before", 64, 16, 1, true)) return 1;
    if (run("1 wave/WG, 16 in flight, 1 pass, same data again (no rewrite)", 64, 16, 1, false)) return 1;
    if (run("1 wave/WG, 16 in flight, 8 passes in one kernel (L2-warm after the 1st)", 64, 16, 8, true)) return 1;
    if (run("1 wave/WG, 8 in flight, 1 pass, rewritten", 64, 8, 1, true)) return 1;
    if (run("1 wave/WG, 32 in flight, 1 pass, rewritten", 64, 32, 1, true)) return 1;
    if (run("4 waves/WG, 16 in flight, 1 pass, rewritten", 256, 16, 1, true)) return 1;
    if (run("4 waves/WG, 32 in flight, 1 pass, rewritten", 256, 32, 1, true)) return 1;
    CK(hipFree(basis));
  }
  // kernel launch floor with events
  float ms;
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(read_strips<16>, dim3(1), dim3(64), 0, 0, out, 0, 0, 0, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::printf("empty kernel between two events: %.2f us\n", ms * 1e3);
  return 0;
}
