// Can the product buffer of the propagation-blocked SpMV live in the Infinity Cache?  (bytes per nonzero.)
// The PB SpMV writes every product once (phase 1) and reads it once (phase 2): 16 of its 28 B/nnz.  If the matrix is cut into
// TILES whose products fit the 256 MiB cache together with what streams by in between, and the same product buffer is
// reused by every tile, those 16 B need never reach HBM.  This probe replays that traffic pattern with plain streaming
// kernels: per tile of m entries
//   A: read 8 B (value) + 2 B (column) per entry from the matrix stream (advancing), write 8 B per entry to P (reused)
//   B: read 8 B per entry from P + 2 B (row) per entry from the stream, reduce
// for a fixed total of 1.5e8 entries, tiles of 1.5e8 / K entries, K = 1 (today's two phases) ... 32, launch gaps included.
//   hipcc --offload-arch=gfx950 -O3 tools/ic_tile_probe.hip -o tools/_build/ic_tile_probe && tools/_build/ic_tile_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e));                    \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

constexpr int kThreads = 1024;
constexpr int U = 3;  // 16-byte value loads in flight per lane and trip

// quads: 4 entries = 32 B of values (two 16-B loads), 8 B of columns
__global__ __launch_bounds__(kThreads) void phase_a(const double2* __restrict__ val, const uint2* __restrict__ col,
                                                    double2* __restrict__ P, size_t q0, size_t nq) {
  const size_t stride = (size_t)gridDim.x * kThreads;
  for (size_t i0 = (size_t)blockIdx.x * kThreads + threadIdx.x; i0 < nq; i0 += stride * U) {
    double2 v[U][2];
    uint2 c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      const size_t ic = i < nq ? i : nq - 1;
      v[u][0] = val[2 * (q0 + ic)];
      v[u][1] = val[2 * (q0 + ic) + 1];
      c[u] = col[q0 + ic];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      if (i < nq) {
        const double s = (double)(c[u].x & 1u) + (double)(c[u].y & 1u);
        P[2 * i] = double2{v[u][0].x * s, v[u][0].y * s};
        P[2 * i + 1] = double2{v[u][1].x * s, v[u][1].y * s};
      }
    }
  }
}
__global__ __launch_bounds__(kThreads) void phase_b(const double2* __restrict__ P, const uint2* __restrict__ row, size_t q0,
                                                    size_t nq, double* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * kThreads;
  double acc = 0.0;
  for (size_t i0 = (size_t)blockIdx.x * kThreads + threadIdx.x; i0 < nq; i0 += stride * U) {
    double2 v[U][2];
    uint2 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      const size_t ic = i < nq ? i : nq - 1;
      v[u][0] = P[2 * ic];
      v[u][1] = P[2 * ic + 1];
      r[u] = row[q0 + ic];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      if (i < nq) acc += (v[u][0].x + v[u][0].y + v[u][1].x + v[u][1].y) * (double)(r[u].x & 1u);
    }
  }
  if (acc == 1.2345e-300) out[0] = acc;
}

int main(int argc, char** argv) {
  const size_t nnz = argc > 1 ? (size_t)std::atoll(argv[1]) : (size_t)150000000;
  const size_t nq = nnz / 4;
  double2 *val, *P;
  uint2 *col, *row;
  double* out;
  CK(hipMalloc(&val, nq * 32));
  CK(hipMalloc(&col, nq * 8));
  CK(hipMalloc(&row, nq * 8));
  CK(hipMalloc(&P, nq * 32));
  CK(hipMalloc(&out, 8));
  CK(hipMemset(val, 0, nq * 32));
  CK(hipMemset(col, 1, nq * 8));
  CK(hipMemset(row, 1, nq * 8));
  CK(hipMemset(P, 0, nq * 32));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = 256;
  for (int K : {1, 2, 4, 6, 8, 12, 16, 24, 32, 48}) {
    const size_t per = (nq + K - 1) / K;
    auto sweep = [&](bool reuse) {
      for (int t = 0; t < K; ++t) {
        const size_t q0 = (size_t)t * per;
        const size_t m = q0 + per <= nq ? per : nq - q0;
        double2* Pt = reuse ? P : P + 2 * q0;
        hipLaunchKernelGGL(phase_a, dim3(grid), dim3(kThreads), 0, 0, val, col, Pt, q0, m);
        hipLaunchKernelGGL(phase_b, dim3(grid), dim3(kThreads), 0, 0, Pt, row, q0, m, out);
      }
    };
    for (int reuse = 0; reuse < 2; ++reuse) {
      sweep(reuse);
      sweep(reuse);
      CK(hipEventRecord(e0));
      for (int r = 0; r < 5; ++r) sweep(reuse);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 5;
      std::printf("K = %2d tiles of %6.1f M entries (P tile %6.1f MB) %s: %.3f ms per sweep = %.2f TB/s of 28 B/nnz, %.2f TB/s of 12 B/nnz\n",
                  K, per * 4 / 1e6, per * 32 / 1e6, reuse ? "P reused  " : "P advances", ms, 28.0 * nnz / ms / 1e9,
                  12.0 * nnz / ms / 1e9);
    }
  }
  return 0;
}
