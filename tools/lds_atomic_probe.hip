// Rate of ds_add_f64 (what phase 2 of the propagation-blocked SpMV issues once per matrix entry) by address pattern,
// 1024-lane workgroups, one per CU, a 13 312-double (104 KiB) slice like the production kernel.  Prints adds per
// clock per CU (at the measured kernel time and a nominal 2.4 GHz) for: random rows (the SpMV's pattern), rows that
// are conflict-free within a wave (lane l of a wave -> bank pair l mod 32), consecutive rows, and for comparison the
// same with plain read-modify-write (not atomic: rate only) and with f32 atomics.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_probe.hip -o /tmp/lap && /tmp/lap
#include <hip/hip_runtime.h>

#include <cstdio>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e));                 \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

constexpr int kThreads = 1024, kRows = 13312, kIters = 2048;

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// MODE 0: ds_add_f64 ; 1: plain RMW (racy) ; 2: ds_add_f32 ; PATTERN 0 random, 1 bank-conflict-free per wave, 2 consecutive
template <int MODE, int PATTERN, int WAVES_ACTIVE>
__global__ __launch_bounds__(kThreads) void lds_adds(double* out) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < kRows; i += kThreads) lds[i] = 0.0;
  __syncthreads();
  if (wave < WAVES_ACTIVE) {
    unsigned h = hash32(tid * 2654435761u + blockIdx.x);
    for (int it = 0; it < kIters; ++it) {
      int row;
      if (PATTERN == 0) {
        h = hash32(h + it);
        row = h % kRows;
      } else if (PATTERN == 1) {
        h = hash32(h + it);
        row = ((h % (kRows / 64)) * 64) + ((lane + it) & 63);  // distinct rows mod 64 within the wave
      } else {
        row = (tid + it * 7) % kRows;
      }
      if (MODE == 0) unsafeAtomicAdd(&lds[row], 1.0);
      else if (MODE == 1) lds[row] += 1.0;
      else unsafeAtomicAdd(reinterpret_cast<float*>(lds) + row, 1.0f);
    }
  }
  __syncthreads();
  if (tid == 0 && lds[0] == -1.0) out[0] = lds[1];
}

template <int MODE, int PATTERN, int WA> float run(double* out, hipEvent_t e0, hipEvent_t e1, int grid) {
  const int lds_bytes = kRows * 8;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_adds<MODE, PATTERN, WA>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL((lds_adds<MODE, PATTERN, WA>), dim3(grid), dim3(kThreads), lds_bytes, 0, out);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((lds_adds<MODE, PATTERN, WA>), dim3(grid), dim3(kThreads), lds_bytes, 0, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  double* out;
  CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = 256;
  auto report = [&](const char* name, float ms, int waves) {
    const double adds_per_cu = (double)waves * 64 * kIters;
    std::printf("%-58s %8.3f ms  %6.2f adds/clk/CU  (chip: %.0f G adds/s)\n", name, ms, adds_per_cu / (ms * 1e-3 * 2.4e9),
                adds_per_cu * grid / (ms * 1e-3) / 1e9);
  };
  report("ds_add_f64, random rows, 16 waves", run<0, 0, 16>(out, e0, e1, grid), 16);
  report("ds_add_f64, random rows, 4 waves", run<0, 0, 4>(out, e0, e1, grid), 4);
  report("ds_add_f64, random rows, 1 wave", run<0, 0, 1>(out, e0, e1, grid), 1);
  report("ds_add_f64, conflict-free within a wave, 16 waves", run<0, 1, 16>(out, e0, e1, grid), 16);
  report("ds_add_f64, conflict-free within a wave, 1 wave", run<0, 1, 1>(out, e0, e1, grid), 1);
  report("ds_add_f64, consecutive rows, 16 waves", run<0, 2, 16>(out, e0, e1, grid), 16);
  report("plain f64 read-modify-write (racy), random rows, 16 waves", run<1, 0, 16>(out, e0, e1, grid), 16);
  report("plain f64 read-modify-write (racy), conflict-free, 16 waves", run<1, 1, 16>(out, e0, e1, grid), 16);
  report("ds_add_f32, random rows, 16 waves", run<2, 0, 16>(out, e0, e1, grid), 16);
  report("ds_add_f32, conflict-free within a wave, 16 waves", run<2, 1, 16>(out, e0, e1, grid), 16);
  std::printf("# production phase 2 needs 1.55e8 adds per SpMV = 6.05e5 per CU; at R adds/clk/CU that is 6.05e5 / R / 2.4e9 s\n");
  return 0;
}
