// Kernel-quality probe for the CSR-stream SpMV (csrc/kernels.hip: spmv_stream) on matrices whose gathers are LOCAL
// (banded random, 5-point Laplacian) next to the uniformly random pattern of config 3: how much of the gap to the HBM
// roofline is the gather and how much is the kernel's own latency chain (tile_rows -> row offsets -> columns -> x ->
// LDS -> barrier -> row offsets again -> fold)?
//   V0  : the production design: 1024-entry tiles, 256 lanes, one entry per lane per trip, row offsets re-read in the fold
//   V1<TILE,NT>: TILE entries per tile, all of a lane's loads issued before the first use (TILE/NT columns, then the
//         gathers), next tile's metadata prefetched, the tile's row offsets staged in LDS during the load phase
//   ELL : no row structure at all (K entries per row, lane-per-row would be wrong for CSR): the gather + stream bound
//   hipcc --offload-arch=gfx950 -O3 tools/csr_stream_probe.hip -o tools/_build/csr_stream_probe && tools/_build/csr_stream_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                            \
  do {                                                                   \
    hipError_t e = (x);                                                  \
    if (e != hipSuccess) {                                               \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e));          \
      return 1;                                                          \
    }                                                                    \
  } while (0)

__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// pattern 0: uniform random columns; 1: columns within +-band of the row; 2: 5-point Laplacian on a side x side grid
__global__ void fill(int64_t n, int K, int pattern, int64_t band, int64_t side, int32_t* rp, int32_t* ci, double* va, double* x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  if (i == n) { rp[n] = (int32_t)(n * K); return; }
  rp[i] = (int32_t)(i * K);
  x[i] = 2.0 * ((splitmix64(i + 77) >> 11) * 0x1.0p-53) - 1.0;
  for (int j = 0; j < K; ++j) {
    const uint64_t h = splitmix64(64 * (uint64_t)i + j);
    int64_t c;
    if (pattern == 0) c = (int64_t)(h % (uint64_t)n);
    else if (pattern == 1) { c = i + (int64_t)(h % (uint64_t)(2 * band + 1)) - band; c = c < 0 ? 0 : (c >= n ? n - 1 : c); }
    else { const int64_t off[5] = {-side, -1, 0, 1, side}; c = i + off[j]; c = c < 0 ? 0 : (c >= n ? n - 1 : c); }
    ci[i * K + j] = (int32_t)c;
    va[i * K + j] = 2.0 * ((splitmix64(h) >> 11) * 0x1.0p-53) - 1.0;
  }
}

constexpr int kXcds = 8;
struct TileWalk {
  int first, step, end;
  __device__ TileWalk(int ntiles) {
    const int xcd = blockIdx.x % kXcds, local = blockIdx.x / kXcds, nlocal = gridDim.x / kXcds;
    const int per = (ntiles + kXcds - 1) / kXcds;
    first = xcd * per + local;
    step = nlocal;
    end = min(ntiles, (xcd + 1) * per);
  }
};

// ---- V0: production design
__global__ __launch_bounds__(256) void v0(int ntiles, const int32_t* __restrict__ tile_rows, const int32_t* __restrict__ rp,
                                          const int32_t* __restrict__ ci, const double* __restrict__ va,
                                          const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double prod[1024];
  const int tid = threadIdx.x;
  for (TileWalk tw(ntiles); tw.first < tw.end; tw.first += tw.step) {
    const int t = tw.first;
    const int r0 = tile_rows[t], r1 = tile_rows[t + 1];
    const int p0 = rp[r0], p1 = rp[r1];
    const int nr = r1 - r0, len = p1 - p0;
    __syncthreads();
#pragma unroll 4
    for (int i = tid; i < len; i += 256) prod[i] = va[p0 + i] * x[ci[p0 + i]];
    __syncthreads();
    int lanes = 1;
    while (lanes < 64 && nr * (lanes << 1) <= 256) lanes <<= 1;
    const int g = tid / lanes, l = tid - g * lanes;
    double acc = 0;
    if (g < nr) {
      const int a = rp[r0 + g] - p0, b = rp[r0 + g + 1] - p0;
      for (int i = a + l; i < b; i += lanes) acc += prod[i];
    }
    for (int d = lanes >> 1; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
    if (g < nr && l == 0) y[r0 + g] = acc;
  }
}

// ---- V1: deep tiles
template <int TILE, int NT, int MAXR>
__global__ __launch_bounds__(NT) void v1(int ntiles, const int32_t* __restrict__ tile_rows, const int32_t* __restrict__ rp,
                                         const int32_t* __restrict__ ci, const double* __restrict__ va,
                                         const double* __restrict__ x, double* __restrict__ y) {
  constexpr int EPT = TILE / NT;
  __shared__ double prod[TILE];
  __shared__ int rowp[MAXR + 1];
  const int tid = threadIdx.x;
  TileWalk tw(ntiles);
  int t = tw.first;
  if (t >= tw.end) return;
  int r0 = tile_rows[t], r1 = tile_rows[t + 1];
  int p0 = rp[r0], p1 = rp[r1];
  for (;;) {
    const int nr = r1 - r0, len = p1 - p0;
    int c[EPT];
    double v[EPT], xv[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int idx = tid + i * NT;
      c[i] = idx < len ? ci[p0 + idx] : 0;
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int idx = tid + i * NT;
      v[i] = idx < len ? va[p0 + idx] : 0.0;
    }
    // the tile's row offsets -> LDS (read by the fold), next tile's boundaries -> registers
    const int tn = t + tw.step;
    int nr0 = 0, nr1 = 0;
    if (tn < tw.end) { nr0 = tile_rows[tn]; nr1 = tile_rows[tn + 1]; }
    __syncthreads();  // previous fold done with prod/rowp
    for (int r = tid; r <= nr; r += NT) rowp[r] = rp[r0 + r] - p0;
#pragma unroll
    for (int i = 0; i < EPT; ++i) xv[i] = x[c[i]];
    int np0 = 0, np1 = 0;
    if (tn < tw.end) { np0 = rp[nr0]; np1 = rp[nr1]; }
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int idx = tid + i * NT;
      if (idx < len) prod[idx] = v[i] * xv[i];
    }
    __syncthreads();
    int lanes = 1;
    while (lanes < 64 && nr * (lanes << 1) <= NT) lanes <<= 1;
    const int groups = NT / lanes, l = tid & (lanes - 1);
    for (int g = tid / lanes; g < nr; g += groups) {  // uniform trip count within a lane group
      const int a = rowp[g], b = rowp[g + 1];
      double acc = 0;
      for (int i = a + l; i < b; i += lanes) acc += prod[i];
      for (int d = lanes >> 1; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
      if (l == 0) y[r0 + g] = acc;
    }
    if (tn >= tw.end) break;
    t = tn; r0 = nr0; r1 = nr1; p0 = np0; p1 = np1;
  }
}

// ---- ELL-like bound: K entries per row known, lane per entry, products summed per row through LDS-free shuffles is not
// possible in general; here each lane streams its entries and the result is a plain per-entry product sum into y by row
// blocks of 256/K... only the memory behaviour matters: stream val+col, gather x, write n doubles.
__global__ __launch_bounds__(256) void ell_bound(int64_t nnz, const int32_t* __restrict__ ci, const double* __restrict__ va,
                                                 const double* __restrict__ x, double* __restrict__ y, int K) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 8;
  for (int64_t base = (int64_t)blockIdx.x * 256 * 8; base < nnz; base += stride) {
    int c[8]; double v[8]; double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int64_t p = base + threadIdx.x + i * 256; c[i] = p < nnz ? ci[p] : 0; v[i] = p < nnz ? va[p] : 0; }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] * x[c[i]];
    if (s == 1.2345e300) y[0] = s;
    if (threadIdx.x < 256 * 8 / K / 8 * 8 && base / K + threadIdx.x < nnz / K) y[base / K + threadIdx.x] = s;  // ~n doubles written in total
  }
}

static void build_tiles(const std::vector<int32_t>& rp, int64_t n, int tile, int maxr, std::vector<int32_t>& tiles) {
  tiles.assign(1, 0);
  int64_t r = 0;
  while (r < n) {
    int64_t r1 = r;
    while (r1 < n && (r1 - r) < maxr && rp[r1 + 1] - rp[r] <= tile) ++r1;
    if (r1 == r) r1 = r + 1;
    tiles.push_back((int32_t)r1);
    r = r1;
  }
}

int main() {
  struct Case { const char* name; int64_t n; int K; int pattern; int64_t band, side; };
  const Case cases[] = {{"banded random, n=1e7, 15/row, +-65536", 10000000, 15, 1, 65536, 0},
                        {"uniform random, n=1e7, 15/row", 10000000, 15, 0, 0, 0},
                        {"5-point Laplacian 6000x6000 (n=3.6e7)", 36000000, 5, 2, 0, 6000},
                        {"banded random, n=1e7, 15/row, +-2048", 10000000, 15, 1, 2048, 0}};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (const Case& cs : cases) {
    const int64_t n = cs.n, nnz = n * cs.K;
    int32_t *rp, *ci, *tr;
    double *va, *x, *y, *yref;
    CK(hipMalloc(&rp, (n + 1) * 4));
    CK(hipMalloc(&ci, nnz * 4));
    CK(hipMalloc(&va, nnz * 8));
    CK(hipMalloc(&x, n * 8));
    CK(hipMalloc(&y, n * 8));
    CK(hipMalloc(&yref, n * 8));
    hipLaunchKernelGGL(fill, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, 0, n, cs.K, cs.pattern, cs.band, cs.side, rp, ci, va, x);
    CK(hipDeviceSynchronize());
    std::vector<int32_t> hrp((size_t)n + 1);
    CK(hipMemcpy(hrp.data(), rp, (n + 1) * 4, hipMemcpyDeviceToHost));
    const double bytes = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    std::printf("== %s: algorithmic bytes %.3f GB\n", cs.name, bytes / 1e9);
    auto run = [&](const char* name, int tile, int maxr, auto launch, bool is_ref) -> int {
      std::vector<int32_t> tiles;
      build_tiles(hrp, n, tile, maxr, tiles);
      const int ntiles = (int)tiles.size() - 1;
      CK(hipMalloc(&tr, tiles.size() * 4));
      CK(hipMemcpy(tr, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemset(y, 0, n * 8));
      launch(ntiles);
      CK(hipEventRecord(e0));
      for (int r = 0; r < 10; ++r) launch(ntiles);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 10;
      if (is_ref) CK(hipMemcpy(yref, y, n * 8, hipMemcpyDeviceToDevice));
      // max |y - yref| on a sample
      std::vector<double> a(4096), b(4096);
      CK(hipMemcpy(a.data(), y + n / 3, 4096 * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), yref + n / 3, 4096 * 8, hipMemcpyDeviceToHost));
      double d = 0;
      for (int i = 0; i < 4096; ++i) d = std::max(d, std::abs(a[i] - b[i]));
      std::printf("  %-34s %7.3f ms  %6.0f GB/s  frac %.3f  (tiles %d, maxdiff %.1e)\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000, ntiles, d);
      CK(hipFree(tr));
      return 0;
    };
#define V0RUN() run("V0 tile 1024 x 256 lanes", 1024, 256, [&](int nt) { \
      int g = nt < 2048 ? (nt + 7) / 8 * 8 : 2048; hipLaunchKernelGGL(v0, dim3(g), dim3(256), 0, 0, nt, tr, rp, ci, va, x, y); }, true)
#define V1RUN(TILE, NT, MAXR, WGPCU) run("V1 tile " #TILE " x " #NT " lanes, " #WGPCU " wg/CU", TILE, MAXR, [&](int nt) { \
      int cap = 256 * WGPCU; int g = nt < cap ? (nt + 7) / 8 * 8 : cap; \
      hipLaunchKernelGGL((v1<TILE, NT, MAXR>), dim3(g), dim3(NT), 0, 0, nt, tr, rp, ci, va, x, y); }, false)
    if (V0RUN()) return 1;
    if (V1RUN(1024, 256, 1024, 8)) return 1;
    if (V1RUN(2048, 256, 2048, 8)) return 1;
    if (V1RUN(4096, 256, 2048, 4)) return 1;
    if (V1RUN(4096, 512, 2048, 4)) return 1;
    if (V1RUN(8192, 512, 4096, 2)) return 1;
    if (V1RUN(8192, 1024, 4096, 2)) return 1;
    {
      hipLaunchKernelGGL(ell_bound, dim3(2048), dim3(256), 0, 0, nnz, ci, va, x, y, cs.K);
      CK(hipEventRecord(e0));
      for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(ell_bound, dim3(2048), dim3(256), 0, 0, nnz, ci, va, x, y, cs.K);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 10;
      std::printf("  %-34s %7.3f ms  %6.0f GB/s  frac %.3f\n", "stream + gather only (no rows)", ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000);
    }
    CK(hipFree(rp)); CK(hipFree(ci)); CK(hipFree(va)); CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(yref));
  }
  return 0;
}
