// Does it matter WHERE the concurrently running workgroups of a streaming kernel read?  The propagation-blocked SpMV
// (csrc/spmv_pb.hip) gives every workgroup its own contiguous stream (one row / column block = 1.6 - 2 MB), so the 256
// workgroups that run at a time read 256 separate places of a 1.2 GB buffer; the Gram-Schmidt kernels deal 16 KB strips
// out round-robin, so the chip sweeps a vector front to back — and a contiguous-share variant of those ran 8 % slower.
// This probe reads the same bytes with 1024-lane workgroups (one per CU, like phase 2) in both layouts:
//   contiguous : workgroup b reads [b * chunk, (b + 1) * chunk) sequentially, granule by granule
//   interleaved: granule t of workgroup b sits at ((t * G + b % G) ...): the G workgroups of a group read adjacent
//                granules at the same time (chip-wide sequential sweep)
// with one stream (8 B per entry) or two (8 B + 2 B per entry, like P + local rows), 2 or 3 granules in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_layout_probe.hip -o /tmp/slp && /tmp/slp
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

constexpr int kThreads = 1024;
constexpr int kGranuleQuads = 1024;  // one quad (4 entries = 32 B of values + 8 B of indices) per lane per granule

// nblocks logical blocks of `granules` granules each; G = interleave group size (1 = contiguous layout)
template <int D, bool TWO>
__global__ __launch_bounds__(kThreads) void stream_read(const uint4* __restrict__ val, const uint2* __restrict__ idx,
                                                        int granules, int G, double* out) {
  extern __shared__ double lds[];  // sized to force one workgroup per CU
  const int b = blockIdx.x, tid = threadIdx.x;
  const int grp = b / G, mem = b % G;
  // physical quad index of granule t: contiguous (G == 1): (b * granules + t) * 1024; interleaved: group base +
  // (t * G + mem) * 1024
  const long long gbase = (long long)grp * G * granules;
  auto quad_of = [&](int t) { return (gbase + (long long)t * G + mem) * kGranuleQuads + tid; };
  uint4 v[D][2];
  uint2 ix[D];
#pragma unroll
  for (int d = 0; d < D - 1; ++d)
    if (d < granules) {
      const long long q = quad_of(d);
      v[d][0] = val[2 * q];
      v[d][1] = val[2 * q + 1];
      if (TWO) ix[d] = idx[q];
    }
  unsigned acc = 0;
  for (int t = 0; t < granules; ++t) {
    if (t + D - 1 < granules) {
      const long long q = quad_of(t + D - 1);
      v[D - 1][0] = val[2 * q];
      v[D - 1][1] = val[2 * q + 1];
      if (TWO) ix[D - 1] = idx[q];
    }
    acc += v[0][0].x ^ v[0][0].w ^ v[0][1].y ^ v[0][1].z;
    if (TWO) acc += ix[0].x + ix[0].y;
#pragma unroll
    for (int d = 0; d < D - 1; ++d) {
      v[d][0] = v[d + 1][0];
      v[d][1] = v[d + 1][1];
      if (TWO) ix[d] = ix[d + 1];
    }
  }
  if (acc == 0x12345678u) out[0] = lds[tid];  // keep the loads alive
}

int main() {
  const int nblocks = 768, granules = 49;  // 768 blocks x 49 granules x 4096 entries = 1.54e8 entries (config 3's P)
  const size_t quads = (size_t)nblocks * granules * kGranuleQuads;
  uint4* val;
  uint2* idx;
  double* out;
  CK(hipMalloc(&val, quads * 32));
  CK(hipMalloc(&idx, quads * 8));
  CK(hipMalloc(&out, 8));
  CK(hipMemset(val, 1, quads * 32));
  CK(hipMemset(idx, 1, quads * 8));
  const int lds_bytes = 100 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_read<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_read<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_read<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_read<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time = [&](auto launch) {
    launch();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
  };
  for (int G : {1, 8, 32, 256}) {
    const float a2 = time([&] { hipLaunchKernelGGL((stream_read<2, false>), dim3(nblocks), dim3(kThreads), lds_bytes, 0, val, idx, granules, G, out); });
    const float a3 = time([&] { hipLaunchKernelGGL((stream_read<3, false>), dim3(nblocks), dim3(kThreads), lds_bytes, 0, val, idx, granules, G, out); });
    const float b2 = time([&] { hipLaunchKernelGGL((stream_read<2, true>), dim3(nblocks), dim3(kThreads), lds_bytes, 0, val, idx, granules, G, out); });
    const float b3 = time([&] { hipLaunchKernelGGL((stream_read<3, true>), dim3(nblocks), dim3(kThreads), lds_bytes, 0, val, idx, granules, G, out); });
    std::printf("interleave group %3d (%s): one stream  D2 %.3f ms %.0f GB/s | D3 %.3f ms %.0f GB/s ; two streams D2 %.3f ms %.0f GB/s | D3 %.3f ms %.0f GB/s\n",
                G, G == 1 ? "contiguous per workgroup" : "granules interleaved", a2, quads * 32 / a2 / 1e6, a3, quads * 32 / a3 / 1e6, b2,
                quads * 40 / b2 / 1e6, b3, quads * 40 / b3 / 1e6);
  }
  return 0;
}
