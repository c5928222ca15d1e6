// Device-side scalar helpers shared by the HIP translation units (kernels.hip, spmv_pb.hip).
#pragma once

#include "ll_internal.hpp"

namespace ll {

// ---------------------------------------------------------------- scalar helpers
// Storage types T: double, zc (complex double), float, cf (complex float).  Products of two stored values stay in T;
// everything that is SUMMED over many elements (dot products, row sums, norms) is carried in acc_t<T> = double / zc.
__device__ __forceinline__ double zero_of(double*) { return 0.0; }
__device__ __forceinline__ zc zero_of(zc*) { return zc{0.0, 0.0}; }
__device__ __forceinline__ float zero_of(float*) { return 0.0f; }
__device__ __forceinline__ cf zero_of(cf*) { return cf{0.0f, 0.0f}; }
template <typename T> __device__ __forceinline__ T zero() { return zero_of((T*)nullptr); }

__device__ __forceinline__ double to_acc(double a) { return a; }
__device__ __forceinline__ double to_acc(float a) { return (double)a; }
__device__ __forceinline__ zc to_acc(zc a) { return a; }
__device__ __forceinline__ zc to_acc(cf a) { return zc{(double)a.re, (double)a.im}; }
__device__ __forceinline__ void from_acc(double a, double* o) { *o = a; }
__device__ __forceinline__ void from_acc(double a, float* o) { *o = (float)a; }
__device__ __forceinline__ void from_acc(zc a, zc* o) { *o = a; }
__device__ __forceinline__ void from_acc(zc a, cf* o) { *o = cf{(float)a.re, (float)a.im}; }
template <typename T> __device__ __forceinline__ T narrow(acc_t<T> a) {
  T o;
  from_acc(a, &o);
  return o;
}

__device__ __forceinline__ double mul(double a, double b) { return a * b; }
__device__ __forceinline__ float mul(float a, float b) { return a * b; }
__device__ __forceinline__ zc mul(zc a, zc b) { return zc{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cf mul(cf a, cf b) { return cf{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ double add(double a, double b) { return a + b; }
__device__ __forceinline__ float add(float a, float b) { return a + b; }
__device__ __forceinline__ zc add(zc a, zc b) { return zc{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf add(cf a, cf b) { return cf{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ double sub(double a, double b) { return a - b; }
__device__ __forceinline__ float sub(float a, float b) { return a - b; }
__device__ __forceinline__ zc sub(zc a, zc b) { return zc{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cf sub(cf a, cf b) { return cf{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ double rmul(double r, double a) { return r * a; }
__device__ __forceinline__ float rmul(double r, float a) { return (float)r * a; }
__device__ __forceinline__ zc rmul(double r, zc a) { return zc{r * a.re, r * a.im}; }
__device__ __forceinline__ cf rmul(double r, cf a) { return cf{(float)r * a.re, (float)a.im * (float)r}; }
// acc += a*b
__device__ __forceinline__ void fma_acc(double& acc, double a, double b) { acc = fma(a, b, acc); }
__device__ __forceinline__ void fma_acc(float& acc, float a, float b) { acc = fmaf(a, b, acc); }
__device__ __forceinline__ void fma_acc(double& acc, float a, float b) { acc = fma((double)a, (double)b, acc); }
__device__ __forceinline__ void fma_acc(zc& acc, zc a, zc b) {
  acc.re = fma(a.re, b.re, fma(-a.im, b.im, acc.re));
  acc.im = fma(a.re, b.im, fma(a.im, b.re, acc.im));
}
__device__ __forceinline__ void fma_acc(cf& acc, cf a, cf b) {
  acc.re = fmaf(a.re, b.re, fmaf(-a.im, b.im, acc.re));
  acc.im = fmaf(a.re, b.im, fmaf(a.im, b.re, acc.im));
}
__device__ __forceinline__ void fma_acc(zc& acc, cf a, cf b) { fma_acc(acc, to_acc(a), to_acc(b)); }
// acc += conj(a)*b   (inner product is conjugate-linear in its first argument, LA:41,49)
__device__ __forceinline__ void cfma_acc(double& acc, double a, double b) { acc = fma(a, b, acc); }
__device__ __forceinline__ void cfma_acc(double& acc, float a, float b) { acc = fma((double)a, (double)b, acc); }
__device__ __forceinline__ void cfma_acc(zc& acc, zc a, zc b) {
  acc.re = fma(a.re, b.re, fma(a.im, b.im, acc.re));
  acc.im = fma(a.re, b.im, fma(-a.im, b.re, acc.im));
}
__device__ __forceinline__ void cfma_acc(zc& acc, cf a, cf b) { cfma_acc(acc, to_acc(a), to_acc(b)); }
// w -= h*u  (h in the accumulator type, w and u stored values)
__device__ __forceinline__ void fnma_acc(double& w, double h, double u) { w = fma(-h, u, w); }
__device__ __forceinline__ void fnma_acc(float& w, double h, float u) { w = (float)fma(-h, (double)u, (double)w); }
__device__ __forceinline__ void fnma_acc(zc& w, zc h, zc u) {
  w.re = fma(-h.re, u.re, fma(h.im, u.im, w.re));
  w.im = fma(-h.re, u.im, fma(-h.im, u.re, w.im));
}
__device__ __forceinline__ void fnma_acc(cf& w, zc h, cf u) {
  zc t = to_acc(w);
  fnma_acc(t, h, to_acc(u));
  w = cf{(float)t.re, (float)t.im};
}
// |re| + |im| (>= the modulus): the magnitude the fixed-point scales of the PB SpMV are built from
__device__ __forceinline__ double abs1(double a) { return fabs(a); }
__device__ __forceinline__ double abs1(float a) { return fabs((double)a); }
__device__ __forceinline__ double abs1(zc a) { return fabs(a.re) + fabs(a.im); }
__device__ __forceinline__ double abs1(cf a) { return fabs((double)a.re) + fabs((double)a.im); }
__device__ __forceinline__ double abs2(double a) { return a * a; }
__device__ __forceinline__ double abs2(float a) { return (double)a * (double)a; }
__device__ __forceinline__ double abs2(zc a) { return fma(a.re, a.re, a.im * a.im); }
__device__ __forceinline__ double abs2(cf a) { return fma((double)a.re, (double)a.re, (double)a.im * (double)a.im); }
// Re(conj(a)*b)
__device__ __forceinline__ double re_cmul(double a, double b) { return a * b; }
__device__ __forceinline__ double re_cmul(float a, float b) { return (double)a * (double)b; }
__device__ __forceinline__ double re_cmul(zc a, zc b) { return fma(a.re, b.re, a.im * b.im); }
__device__ __forceinline__ double re_cmul(cf a, cf b) { return fma((double)a.re, (double)b.re, (double)a.im * (double)b.im); }

__device__ __forceinline__ double shfl_down_d(double v, int delta) { return __shfl_down(v, delta, 64); }

// Sum over the 64 lanes of a wavefront; result valid in lane 0.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += shfl_down_d(v, d);
  return v;
}
__device__ __forceinline__ zc wave_sum(zc v) { return zc{wave_sum(v.re), wave_sum(v.im)}; }

// Wave sums of N accumulators at once (N a power of two <= 64), transposing as it reduces: at every step a lane hands
// half of the accumulators it still holds to its partner and keeps the partner's share of the other half, so N sums
// cost N - 1 + log2(64 / N) shuffle-adds instead of 6 N.  On return a[0] of lane L is the full sum of accumulator
// L / (64 / N) (all 64 / N lanes of that group hold it).  Fixed order: bit-reproducible.
template <int N> __device__ __forceinline__ void wave_sum_transposed(double (&a)[N], int lane) {
  static_assert(N >= 1 && N <= 64 && (N & (N - 1)) == 0, "N must be a power of two");
  int m = 32;
#pragma unroll
  for (int half = N / 2; half >= 1; half >>= 1, m >>= 1) {
    const bool upper = (lane & m) != 0;
#pragma unroll
    for (int i = 0; i < half; ++i) {
      const double keep = upper ? a[i + half] : a[i];
      const double send = upper ? a[i] : a[i + half];
      a[i] = keep + __shfl_xor(send, m, 64);
    }
  }
#pragma unroll
  for (; m >= 1; m >>= 1) a[0] += __shfl_xor(a[0], m, 64);
}

// Sum over the workgroup (kBlock = 4 waves); result valid in thread 0. `scratch` holds >= 4 doubles.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) v = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
  return v;
}

// DGKS selection (see NormRefs in ll_internal.hpp).
__device__ __forceinline__ bool second_pass_due(const NormRefs& r) { return r.force2 || *r.c1 < r.thr * *r.c0; }
__device__ __forceinline__ double final_norm2(const NormRefs& r) { return second_pass_due(r) ? *r.c2 : *r.c1; }

// XCD-aware persistent tile walk: workgroups with equal (blockIdx % 8) share an XCD (and its L2), so each such
// class walks one contiguous eighth of the tile range; neighbouring tiles (which gather neighbouring parts of x
// for banded / stencil matrices) then hit the same L2 instead of being fetched by all eight.
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

}  // namespace ll
