// Rounding a double to the nearest 64-bit integer (ties to even) without the f64 -> i64 conversion sequence.
//
// The fixed-point SpMV kernels (spmv_pb.hip: pb_phase2_fixed, tl_spmv_kernel) turn every product into an integer of up to 63
// bits.  gfx950 has no f64 -> i64 instruction: `(long long)rint(v)` compiles to v_rndne, v_ldexp, v_floor, v_fma, v_cvt_i32,
// v_cvt_u32 — five of them quarter-rate.  The same integer falls out of four full-rate additions:
//     t1 = v + 1.5 * 2^84     the sum is rounded to the ulp of [2^84, 2^85) = 2^32: the low mantissa word of t1 holds
//                             round(v / 2^32) in two's complement (|v| < 2^63)
//     hi = t1 - 1.5 * 2^84    exact: v rounded to a multiple of 2^32
//     lo = v - hi             exact (Sterbenz), |lo| <= 2^31
//     t0 = lo + 1.5 * 2^52    ulp 1: the mantissa of t0 holds 2^51 + rint(lo), ties to even
//     result = (low word of t1) << 32  +  (bits(t0) - bits(1.5 * 2^52))          (mod 2^64; exact because |result| < 2^63)
// hi is an even integer, so hi + rint(lo) = rint(hi + lo) = rint(v) including the tie rule: the SAME integer as
// `(long long)rint(v)` for every |v| < 9.0e18 (tests/cpp/fixed_round_test.cpp compares the two on the host, where the four
// additions are the same IEEE operations; the library is built with -ffp-contract=off and without fast-math, so the compiler
// keeps them).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define LL_FIXED_HD __host__ __device__ __forceinline__
#else
#define LL_FIXED_HD inline
#endif

namespace ll {
LL_FIXED_HD long long fixed_round(double v) {  // |v| < 9.0e18 (the callers test that first)
  const double m1 = 0x1.8p84, m0 = 0x1.8p52;
  const double t1 = v + m1;
  const double hi = t1 - m1;
  const double lo = v - hi;
  const double t0 = lo + m0;
  uint64_t b1, b0;
  std::memcpy(&b1, &t1, sizeof(b1));
  std::memcpy(&b0, &t0, sizeof(b0));
  return (long long)((b1 << 32) + (b0 - 0x4338000000000000ull));
}
}  // namespace ll
