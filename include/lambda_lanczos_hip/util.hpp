// Host-side helpers of the reference's `lambda_lanczos::util` namespace (util/common.hpp:80-221,
// util/linear_algebra.hpp:29-163 of mrcdr/lambda-lanczos), so that user sources and tests written against those names
// compile through include/compat unchanged.  These are small std::vector helpers a caller uses AROUND the engines
// (preparing an input, checking an output): they run on the host, like the reference's.  The Krylov loop does not use
// them — its sweeps are the HIP kernels behind include/lanczos_hip.h.
//
// Same signatures, same conventions: inner_prod conjugates its FIRST argument and folds left to right
// (linear_algebra.hpp:29-51, pinned by the reference's <(3,1+3i),(3,2+4i)> = 23 - 2i, test/lambda_lanczos_test.cpp:47-59),
// m_norm is sum |Re| + |Im| (:122-125, pinned by m_norm = 6, test :93-100), sgn(0) = +1 (common.hpp:194-201).
#ifndef LAMBDA_LANCZOS_HIP_UTIL_HPP_
#define LAMBDA_LANCZOS_HIP_UTIL_HPP_

#include <algorithm>
#include <cassert>
#include <cmath>
#include <complex>
#include <cstddef>
#include <functional>
#include <limits>
#include <numeric>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

namespace lambda_lanczos_hip {
namespace util {

// real_t<T>: T for real types, R for std::complex<R> (util/common.hpp:80-102)
template <typename T> struct realTypeMap { typedef T type; };
template <typename T> struct realTypeMap<std::complex<T>> { typedef T type; };
template <typename T> using real_t = typename realTypeMap<T>::type;

// typed_conj: identity for real T, std::conj for complex T (util/common.hpp:112-134)
template <typename T> struct TypedConjugate {
  static T invoke(const T& v) { return v; }
};
template <typename R> struct TypedConjugate<std::complex<R>> {
  static std::complex<R> invoke(const std::complex<R>& v) { return std::complex<R>(v.real(), -v.imag()); }
};
template <typename T> inline T typed_conj(const T& v) { return TypedConjugate<T>::invoke(v); }

// <v1|v2> = sum_i conj(v1_i) v2_i, sequential fold from element 0 (util/linear_algebra.hpp:29-51)
template <typename T> inline T inner_prod(const std::vector<T>& v1, const std::vector<T>& v2) {
  assert(v1.size() == v2.size());
  T acc = T();
  const std::size_t n = v1.size();
  for (std::size_t i = 0; i < n; ++i) acc = acc + typed_conj(v1[i]) * v2[i];
  return acc;
}

// Euclidean norm sqrt(Re<v|v>), unscaled (util/linear_algebra.hpp:56-60)
template <typename T> inline real_t<T> norm(const std::vector<T>& vec) { return std::sqrt(std::real(inner_prod(vec, vec))); }

// vec *= a, element by element (util/linear_algebra.hpp:65-72)
template <typename T1, typename T2> inline void scalar_mul(T1 a, std::vector<T2>& vec) {
  for (T2& e : vec) e *= a;
}

// vec *= T(1) / norm(vec) (util/linear_algebra.hpp:77-80: the reciprocal is formed first)
template <typename T> inline void normalize(std::vector<T>& vec) { scalar_mul(T(1) / norm(vec), vec); }

// sum_i |Re v_i| + |Im v_i| — the BLAS _ASUM convention, not sum |v_i| (util/linear_algebra.hpp:82-125)
template <typename T> struct ManhattanNorm {
  static T invoke(const std::vector<T>& vec) {
    T acc = T();
    for (const T& e : vec) acc = acc + std::abs(e);
    return acc;
  }
};
template <typename R> struct ManhattanNorm<std::complex<R>> {
  static R invoke(const std::vector<std::complex<R>>& vec) {
    R acc = R();
    for (const std::complex<R>& e : vec) acc = acc + std::abs(e.real()) + std::abs(e.imag());
    return acc;
  }
};
template <typename T> inline real_t<T> m_norm(const std::vector<T>& vec) { return ManhattanNorm<T>::invoke(vec); }

// Modified Gram-Schmidt of uorth against the orthonormal vectors [first, last), in iteration order
// (util/linear_algebra.hpp:132-144)
template <typename ForwardIterator, typename T>
inline void schmidt_orth(std::vector<T>& uorth, ForwardIterator first, ForwardIterator last) {
  for (ForwardIterator it = first; it != last; ++it) {
    const auto& u = *it;
    const T h = inner_prod(u, uorth);
    const std::size_t n = uorth.size();
    for (std::size_t i = 0; i < n; ++i) uorth[i] -= h * u[i];
  }
}

// a := n x n identity (util/linear_algebra.hpp:149-163)
template <typename T> void initAsIdentity(std::vector<std::vector<T>>& a, std::size_t n) {
  a.assign(n, std::vector<T>(n, T()));
  for (std::size_t i = 0; i < n; ++i) a[i][i] = T(1.0);
}

// Sort eigenvalues by `predicate` (default ascending) and, when asked, move the eigenvectors along
// (util/common.hpp:141-174; the vectors change their memory location, as there)
template <typename T>
inline void sort_eigenpairs(std::vector<real_t<T>>& eigenvalues, std::vector<std::vector<T>>& eigenvectors,
                            bool sort_eigenvector,
                            const std::function<bool(real_t<T>, real_t<T>)> predicate = std::less<real_t<T>>()) {
  const std::size_t m = eigenvalues.size();
  std::vector<std::size_t> order(m);
  std::iota(order.begin(), order.end(), std::size_t(0));
  // std::sort on (value, position) pairs in the reference; ties are not ordered by either (no stability promised)
  std::sort(order.begin(), order.end(),
            [&](std::size_t a, std::size_t b) { return predicate(eigenvalues[a], eigenvalues[b]); });
  std::vector<real_t<T>> values(m);
  for (std::size_t i = 0; i < m; ++i) values[i] = eigenvalues[order[i]];
  eigenvalues.swap(values);
  if (sort_eigenvector) {
    std::vector<std::vector<T>> vectors;
    vectors.reserve(m);
    for (std::size_t i = 0; i < m; ++i) vectors.push_back(std::move(eigenvectors[order[i]]));
    eigenvectors.swap(vectors);
  }
}

// Significant decimal digits of T and 10^-digits (util/common.hpp:180-189)
template <typename T> inline constexpr int sig_decimal_digit() {
  return (int)(std::numeric_limits<T>::digits * std::log10(std::numeric_limits<T>::radix));
}
template <typename T> inline constexpr T minimum_effective_decimal() { return std::pow(10, -sig_decimal_digit<T>()); }

// +1 for val >= 0 (zero included), -1 otherwise (util/common.hpp:194-201)
template <typename T> T sgn(T val) { return val >= 0 ? (T)1 : (T)(-1); }

// Elements joined by `delimiter`, nothing after the last one (util/common.hpp:206-221; like the reference, only the LAST
// CHARACTER of a trailing delimiter is removed)
template <typename T> std::string vectorToString(const std::vector<T>& vec, std::string delimiter = " ") {
  std::stringstream ss;
  for (const T& e : vec) ss << e << delimiter;
  std::string s = ss.str();
  if (!s.empty()) s.pop_back();
  return s;
}

}  // namespace util
}  // namespace lambda_lanczos_hip

#endif  // LAMBDA_LANCZOS_HIP_UTIL_HPP_
