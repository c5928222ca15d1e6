// EXPERIMENT (never selected automatically; LL_SPMV_KERNEL=l2g): the L2-blocked gather formulation of the CSR SpMV
// that DESIGN.md section 3.1 weighs against propagation blocking.  Kept in the library so that its timing and
// counters can be reproduced (profiles/r02_spmv_alternatives.*) instead of being quoted as prose.
//
// Idea: the gather probe (profiles/r01_gather_probe.txt) gives 217 G gathers/s from an L2-resident table against 57 G/s
// from an 80 MB one.  So x is cut into column SLICES of 2^18 columns (2 MiB of fp64: half of an XCD's L2) and the
// matrix is stored per (row block, slice) tile: value + one 32-bit word (14-bit local row | 18-bit column in the slice),
// 12 bytes per nonzero like CSR.  A persistent grid of one workgroup per CU keeps the y slice of its row block in LDS
// and walks the slices in order; a soft per-XCD barrier keeps the 32 workgroups of an XCD within one slice of each
// other so that the slice they gather from is the one their L2 holds.  No product buffer: HBM traffic is the CSR
// minimum (2.0 GB for BASELINE config 3) plus the x re-reads that L2 misses cause.
//
// Measured (see profiles/): slower than propagation blocking on MI355X — the gathers leave L2 more often than the
// probe's idealised loop (the matrix stream of 1.8 GB passes through the same L2), and the 8-byte gathers are bound by
// the L1 -> L2 request rate, not by bytes.  Single GPU only.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "dev_helpers.hpp"
#include <atomic>
#include <functional>
#include <thread>

#include "ll_internal.hpp"

namespace ll {

constexpr int kL2gThreads = 1024;
constexpr int kL2gRowBits = 14;  // local row < 16384

// streamed-once matrix data: non-temporal loads for the scalar types (the builtin takes no aggregates)
__device__ __forceinline__ double nt_load(const double* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float nt_load(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ uint32_t nt_load(const uint32_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ zc nt_load(const zc* p) {
  const double* d = reinterpret_cast<const double*>(p);
  return zc{__builtin_nontemporal_load(d), __builtin_nontemporal_load(d + 1)};
}
__device__ __forceinline__ cf nt_load(const cf* p) {
  const float* d = reinterpret_cast<const float*>(p);
  return cf{__builtin_nontemporal_load(d), __builtin_nontemporal_load(d + 1)};
}

template <typename T>
__global__ __launch_bounds__(kL2gThreads) void l2g_kernel(int nrb, int nsl, int rb_rows, long long n_local,
                                                          int slice_log2, const int64_t* __restrict__ tptr,  // [nrb][nsl+1]
                                                          const T* __restrict__ val, const uint32_t* __restrict__ idx,
                                                          const T* __restrict__ x, const T* __restrict__ xl,
                                                          T* __restrict__ y, double offset,
                                                          double* __restrict__ dot_partials,
                                                          unsigned* __restrict__ sync /* [rounds][nsl][8] */) {
  constexpr int R = scalar_traits<T>::reals;
  extern __shared__ double lds[];  // [rb_rows * R]
  __shared__ double red[kL2gThreads / 64];
  const int tid = threadIdx.x;
  const int xcd = blockIdx.x % kXcds;
  const unsigned colmask = (1u << slice_log2) - 1u;
  int round = 0;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x, ++round) {
    const int in_round = min((int)gridDim.x, nrb - round * (int)gridDim.x);
    const unsigned peers = (unsigned)((in_round - xcd + kXcds - 1) / kXcds);  // workgroups of my XCD in this round
    const long long row0 = (long long)rb * rb_rows;
    const int rows = (int)min((long long)rb_rows, n_local - row0);
    __syncthreads();
    for (int i = tid; i < rb_rows * R; i += kL2gThreads) lds[i] = 0.0;
    __syncthreads();
    const int64_t* tp = tptr + (size_t)rb * (nsl + 1);
    for (int sl = 0; sl < nsl; ++sl) {
      if (sl >= 2 && tid == 0) {
        // soft barrier: do not run more than one slice ahead of the slowest workgroup of this XCD (bounded wait: the
        // barrier only buys L2 locality, correctness never depends on it)
        const unsigned* c = sync + ((size_t)round * nsl + (sl - 2)) * kXcds + xcd;
        for (int spin = 0; spin < 20000 && __atomic_load_n(c, __ATOMIC_RELAXED) < peers; ++spin) __builtin_amdgcn_s_sleep(8);
      }
      __syncthreads();
      const T* xs = x + ((long long)sl << slice_log2);
      const long long p0 = tp[sl], p1 = tp[sl + 1];
#pragma unroll 4
      for (long long p = p0 + tid; p < p1; p += kL2gThreads) {
        const T v = nt_load(val + p);
        const uint32_t ix = nt_load(idx + p);
        const T pr = mul(v, xs[ix & colmask]);
        const int rl = (int)(ix >> (32 - kL2gRowBits));
        if constexpr (scalar_traits<T>::is_complex) {
          unsafeAtomicAdd(&lds[2 * rl], (double)pr.re);
          unsafeAtomicAdd(&lds[2 * rl + 1], (double)pr.im);
        } else {
          unsafeAtomicAdd(&lds[rl], (double)pr);
        }
      }
      __syncthreads();
      if (tid == 0) atomicAdd(sync + ((size_t)round * nsl + sl) * kXcds + xcd, 1u);
    }
    __syncthreads();
    double dot_acc = 0.0;
    for (int i = tid; i < rows; i += kL2gThreads) {
      const T xi = xl[row0 + i];
      acc_t<T> acc;
      if constexpr (scalar_traits<T>::is_complex) acc = zc{lds[2 * i], lds[2 * i + 1]};
      else acc = lds[i];
      const T yi = add(narrow<T>(acc), rmul(offset, xi));
      y[row0 + i] = yi;
      dot_acc += re_cmul(xi, yi);
    }
    if (dot_partials) {
      const double v = wave_sum(dot_acc);
      if ((tid & 63) == 0) red[tid >> 6] = v;
      __syncthreads();
      if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < kL2gThreads / 64; ++w) t += red[w];
        dot_partials[rb] = t;
      }
    }
  }
}

template <typename T>
int launch_spmv_l2g(const ll_operator& op, const T* x, const T* x_local, T* y, double offset, double* dot_partials,
                    hipStream_t s) {
  static bool attr_done = false;
  const int cap = 160 * 1024 - 2048;
  if (!attr_done) {
    LL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&l2g_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    attr_done = true;
  }
  const int grid = std::min(op.l2_nrb, kCUs);
  const int rounds = (op.l2_nrb + grid - 1) / grid;
  LL_HIP(hipMemsetAsync(op.d_l2_sync, 0, (size_t)rounds * op.l2_nsl * kXcds * sizeof(unsigned), s));
  const size_t lds_bytes = (size_t)op.l2_rb_rows * sizeof(acc_t<T>);
  hipLaunchKernelGGL((l2g_kernel<T>), dim3(grid), dim3(kL2gThreads), lds_bytes, s, op.l2_nrb, op.l2_nsl, op.l2_rb_rows,
                     (long long)op.n_local, op.l2_slice_log2, op.d_l2_ptr, (const T*)op.d_l2_val, op.d_l2_idx, x, x_local, y,
                     offset, dot_partials, op.d_l2_sync);
  LL_HIP(hipGetLastError());
  return op.l2_nrb;
}

// Host-side bucket pass (an experiment: the production image of spmv_pb.hip is built on the device).
template <typename T> bool l2g_build_host(ll_operator* op, const int64_t* rp, const int32_t* ci, const T* va) {
  ll_context* ctx = op->ctx;
  if (ctx->comm != nullptr || op->nnz <= 0) return false;
  const int64_t nr = op->n_local, nc = op->n;
  int slice_log2 = 18;
  if (const char* e = std::getenv("LL_L2G_SLICE_LOG2")) slice_log2 = std::max(8, std::min(18, std::atoi(e)));
  const int64_t row_max = std::min<int64_t>((int64_t)1 << kL2gRowBits, (104 * 1024) / (int64_t)sizeof(acc_t<T>));
  int64_t m = std::max<int64_t>(1, (nr + 256 * row_max - 1) / (256 * row_max));
  int64_t rb_rows = std::min<int64_t>(row_max, std::max<int64_t>(16, (nr + 256 * m - 1) / (256 * m)));
  if (const char* e = std::getenv("LL_PB_BLOCK")) rb_rows = std::min<int64_t>(row_max, std::max(4, std::atoi(e)));
  const int64_t nrb = std::max<int64_t>(1, (nr + rb_rows - 1) / rb_rows);
  const int64_t nsl = (nc + ((int64_t)1 << slice_log2) - 1) >> slice_log2;
  if (nrb * (nsl + 1) > (int64_t)64 << 20) return false;
  std::vector<int64_t> tptr((size_t)nrb * (nsl + 1), 0);
  // host threads that end with the loop (no OpenMP runtime in the product library: its idle workers spin for 200 ms
  // after a region and disturb launch-bound runs that follow, tools/stall_probe.py)
  auto for_row_blocks = [](int64_t count, const std::function<void(int64_t)>& body) {
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(std::thread::hardware_concurrency(), 32), count / 4));
    std::atomic<int64_t> next{0};
    auto work = [&] {
      for (int64_t r0; (r0 = next.fetch_add(4)) < count;)
        for (int64_t r = r0; r < std::min(count, r0 + 4); ++r) body(r);
    };
    std::vector<std::thread> helpers;
    for (int t = 1; t < nt; ++t) helpers.emplace_back(work);
    work();
    for (std::thread& t : helpers) t.join();
  };
  for_row_blocks(nrb, [&](int64_t r) {
    const int64_t i0 = r * rb_rows, i1 = std::min(nr, i0 + rb_rows);
    int64_t* t = &tptr[(size_t)r * (nsl + 1)];
    for (int64_t p = rp[i0]; p < rp[i1]; ++p) ++t[(ci[p] >> slice_log2) + 1];
  });
  {  // global prefix: tile (r, sl) starts after all earlier tiles
    int64_t run = 0;
    for (int64_t r = 0; r < nrb; ++r) {
      int64_t* t = &tptr[(size_t)r * (nsl + 1)];
      int64_t acc = run;
      for (int64_t sl = 0; sl < nsl; ++sl) {
        const int64_t c = t[sl + 1];
        t[sl] = acc;
        acc += c;
      }
      t[nsl] = acc;
      run = acc;
    }
  }
  const size_t nnz = (size_t)op->nnz;
  std::vector<T> val(nnz);
  std::vector<uint32_t> idx(nnz);
  for_row_blocks(nrb, [&](int64_t r) {
    const int64_t i0 = r * rb_rows, i1 = std::min(nr, i0 + rb_rows);
    const int64_t* t = &tptr[(size_t)r * (nsl + 1)];
    std::vector<int64_t> fill(t, t + nsl);
    for (int64_t i = i0; i < i1; ++i)
      for (int64_t p = rp[i]; p < rp[i + 1]; ++p) {
        const int64_t sl = ci[p] >> slice_log2;
        const int64_t q = fill[(size_t)sl]++;
        val[(size_t)q] = va[p];
        idx[(size_t)q] = ((uint32_t)(i - i0) << (32 - kL2gRowBits)) | ((uint32_t)ci[p] & (((uint32_t)1 << slice_log2) - 1u));
      }
  });
  op->l2_nrb = (int)nrb;
  op->l2_nsl = (int)nsl;
  op->l2_rb_rows = (int)rb_rows;
  op->l2_slice_log2 = slice_log2;
  ctx->dev_malloc(&op->d_l2_val, nnz * sizeof(T), "l2g values");
  ctx->dev_malloc((void**)&op->d_l2_idx, nnz * sizeof(uint32_t), "l2g indices");
  ctx->dev_malloc((void**)&op->d_l2_ptr, tptr.size() * sizeof(int64_t), "l2g tile offsets");
  const int grid = std::min((int)nrb, kCUs);
  const int rounds = ((int)nrb + grid - 1) / grid;
  ctx->dev_malloc((void**)&op->d_l2_sync, (size_t)rounds * nsl * kXcds * sizeof(unsigned), "l2g sync counters");
  LL_HIP(hipMemcpy(op->d_l2_val, val.data(), nnz * sizeof(T), hipMemcpyHostToDevice));
  LL_HIP(hipMemcpy(op->d_l2_idx, idx.data(), nnz * sizeof(uint32_t), hipMemcpyHostToDevice));
  LL_HIP(hipMemcpy(op->d_l2_ptr, tptr.data(), tptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  return true;
}

#define LL_INST_L2G(T)                                                                                          \
  template int launch_spmv_l2g<T>(const ll_operator&, const T*, const T*, T*, double, double*, hipStream_t);     \
  template bool l2g_build_host<T>(ll_operator*, const int64_t*, const int32_t*, const T*);
LL_INST_L2G(double) LL_INST_L2G(zc) LL_INST_L2G(float) LL_INST_L2G(cf)

}  // namespace ll
