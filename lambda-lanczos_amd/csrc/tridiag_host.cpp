// Host-side dense math of the Lanczos loop: eigenvalues / eigenvectors of the k x k
// Lanczos matrix T_k.  Stays on the CPU by design: it is O(k^2) scalar work on k-sized arrays, overlapped with
// the device's next iteration by the driver.
//
// The arithmetic follows the reference's implicit-shift QR formulation step for step (TRI:151-166 Givens with
// its two special cases, TRI:181-236 sweep, TRI:252-276 deflation test, TRI:290-343 driver with the 50*nsub
// stagnation guard, CM:141-174 ascending index sort) because the engine's stop decisions (LL:290-309, EX:154)
// are taken on these numbers: same operations in the same order => same decisions.  The data layout is ours:
// flat arrays, one contiguous row-major Q, no per-call vector-of-vectors.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <numeric>
#include <vector>

#include "ll_internal.hpp"

namespace ll {

namespace {

inline double sign_pos0(double v) { return v >= 0.0 ? 1.0 : -1.0; }  // CM:194-201

// Rotate rows r and r+1 of the row-major m x m matrix q (TRI:223-231).
inline void rotate_rows(double* __restrict__ a, double* __restrict__ b, int64_t m, double c, double s) {
  for (int64_t j = 0; j < m; ++j) {
    const double v0 = a[j], v1 = b[j];
    a[j] = c * v0 + s * v1;
    b[j] = -s * v0 + c * v1;
  }
}

// One Wilkinson-shift sweep on the unreduced block [lo, hi] (inclusive), TRI:196-235.
void sweep(double* al, double* be, double* q, int64_t m, int64_t lo, int64_t hi) {
  if (hi == lo) return;
  const double d = (al[hi - 1] - al[hi]) / (2 * be[hi - 1]);
  const double mu = al[hi] - be[hi - 1] / (d + sign_pos0(d) * std::sqrt(d * d + 1.0));
  double x = al[lo] - mu;
  double s = 1.0, c = 1.0, p = 0.0;
  for (int64_t i = lo; i < hi; ++i) {
    const double z = s * be[i];
    const double bprev = c * be[i];
    // one square root serves the rotation (TRI:151-166) and the new coupling (TRI:207): same expression, same bits
    const double h = std::sqrt(x * x + z * z);
    if (z == 0.0) { c = 1.0; s = 0.0; }
    else if (x == 0.0) { c = 0.0; s = 1.0; }
    else { c = x / h; s = z / h; }
    if (i > lo) be[i - 1] = h;
    const double u = (al[i + 1] - al[i] + p) * s + 2.0 * c * bprev;
    al[i] = al[i] - p + s * u;
    p = s * u;
    x = c * u - bprev;
    if (q) rotate_rows(q + i * m, q + (i + 1) * m, m, c, s);
  }
  al[hi] = al[hi] - p;
  be[hi - 1] = x;
}

}  // namespace

int64_t tridiag_qr(int64_t m, const double* alpha, const double* beta, double* ev, double* q) {
  if (m <= 0) return 0;
  std::vector<double> al(alpha, alpha + m), be((size_t)m, 0.0);
  for (int64_t i = 0; i + 1 < m; ++i) be[i] = beta[i];
  std::vector<double> qw;
  double* Q = nullptr;
  if (q) {
    qw.assign((size_t)m * m, 0.0);
    for (int64_t i = 0; i < m; ++i) qw[(size_t)i * m + i] = 1.0;
    Q = qw.data();
  }
  const double half_eps = std::numeric_limits<double>::epsilon() * 0.5;
  const double tiny = std::numeric_limits<double>::min();
  int64_t unconverged = 0, hi_prev = m - 1, stall = 1;
  // The reference re-tests EVERY coupling before every sweep (TRI:257-266).  A sweep only changes al[lo..hi] and
  // be[lo..hi-1], and be[lo-1] = be[hi] = 0 already, so after the first full scan only the block of the previous sweep
  // can change its verdict: scanning that block alone zeroes exactly the same couplings at the same times.
  int64_t scan_lo = 0, scan_hi = m - 1;
  for (;;) {
    for (int64_t i = scan_lo; i < scan_hi; ++i)
      if (std::abs(be[i]) < std::sqrt(std::abs(al[i]) * std::abs(al[i + 1])) * half_eps + tiny) be[i] = 0.0;
    // ... and locate the trailing unreduced block (TRI:268-275)
    int64_t hi = hi_prev;
    while (hi > 0 && be[hi - 1] == 0.0) --hi;
    int64_t lo = hi;
    while (lo > 0 && be[lo - 1] != 0.0) --lo;
    if (hi == 0) break;
    sweep(al.data(), be.data(), Q, m, lo, hi);
    scan_lo = lo;
    scan_hi = hi;
    const int64_t nsub = hi - lo + 1;
    if (hi == hi_prev) {
      if (stall > nsub * 50) {  // forced deflation (TRI:315-331); callers ignore the count (LL:44,268, EX:126)
        hi_prev = lo;
        ++unconverged;
        stall = 1;
      } else {
        ++stall;
      }
    } else {
      stall = 1;
      hi_prev = hi;
    }
  }
  // ascending index sort (CM:141-174: std::sort on (value, index) pairs comparing values only)
  std::vector<std::pair<double, size_t>> order;
  order.reserve((size_t)m);
  for (int64_t i = 0; i < m; ++i) order.emplace_back(al[i], (size_t)i);
  std::sort(order.begin(), order.end(),
            [](const std::pair<double, size_t>& a, const std::pair<double, size_t>& b) { return a.first < b.first; });
  for (int64_t i = 0; i < m; ++i) ev[i] = order[i].first;
  if (q)
    for (int64_t i = 0; i < m; ++i) std::copy(Q + order[i].second * m, Q + (order[i].second + 1) * m, q + i * m);
  return unconverged;
}

// Sturm-sequence bisection for the k-th smallest eigenvalue (TRI:22-88).  Used by LL_TRIDIAG_BISECT / _AUTO for
// the per-iteration stop test only: O(m) per probe instead of O(m^2) for a full QR.
double tridiag_bisect(int64_t m, const double* al, const double* be, int64_t k) {
  double r = 0.0;  // Gerschgorin-style bound sum|alpha| + 2 sum|beta| (TRI:52-58)
  for (int64_t i = 0; i < m; ++i) r += std::abs(al[i]);
  double rb = 0.0;
  for (int64_t i = 0; i + 1 < m; ++i) rb += std::abs(be[i]);
  r += 2 * rb;
  auto count_below = [&](double c) {
    double qi = al[0] - c;
    int64_t cnt = qi < 0 ? 1 : 0;
    for (int64_t i = 1; i < m; ++i) {
      qi = al[i] - c - be[i - 1] * be[i - 1] / qi;
      if (qi < 0) ++cnt;
      if (qi == 0) qi = std::numeric_limits<double>::epsilon();
    }
    return cnt;
  };
  double lo = -r, up = r, mid, pmid = std::numeric_limits<double>::max();
  while (up - lo > std::min(std::abs(lo), std::abs(up)) * std::numeric_limits<double>::epsilon()) {
    mid = (lo + up) * 0.5;
    if (count_below(mid) >= k + 1) up = mid; else lo = mid;
    if (mid == pmid) break;
    pmid = mid;
  }
  return lo;
}


// The same bisection for SEVERAL roots at once (nk <= 8): every root runs exactly the midpoint sequence of
// tridiag_bisect (same bracket [-r, r], same termination), so the results are bit-identical to nk separate calls —
// but the Sturm recurrences of the roots are evaluated in ONE pass over (alpha, beta), as independent dependency
// chains next to each other.  The recurrence is bound by the latency of its division; interleaving the chains lets
// the divider pipeline (and the vector unit) work on all roots at once: ~5x faster than root after root.
void tridiag_bisect_multi(int64_t m, const double* al, const double* be, int nk, const int64_t* ks, double* out) {
  constexpr int W = 8;
  double r = 0.0;
  for (int64_t i = 0; i < m; ++i) r += std::abs(al[i]);
  double rb = 0.0;
  for (int64_t i = 0; i + 1 < m; ++i) rb += std::abs(be[i]);
  r += 2 * rb;
  std::vector<double> b2((size_t)std::max<int64_t>(m, 1));
  for (int64_t i = 1; i < m; ++i) b2[(size_t)i] = be[i - 1] * be[i - 1];
  const double eps = std::numeric_limits<double>::epsilon();
  for (int base = 0; base < nk; base += W) {
    const int w = std::min(W, nk - base);
    double lo[W], up[W], mid[W], pmid[W], q[W];
    int64_t cnt[W];
    bool live[W];
    for (int j = 0; j < W; ++j) {
      lo[j] = -r;
      up[j] = r;
      pmid[j] = std::numeric_limits<double>::max();
      mid[j] = 0.0;
      live[j] = j < w;
    }
    for (;;) {
      bool any = false;
      for (int j = 0; j < w; ++j) {
        if (live[j] && !(up[j] - lo[j] > std::min(std::abs(lo[j]), std::abs(up[j])) * eps)) live[j] = false;
        if (live[j]) {
          mid[j] = (lo[j] + up[j]) * 0.5;
          any = true;
        }
      }
      if (!any) break;
      // Sturm counts below mid[j] for all lanes (finished lanes ride along on their last midpoint)
      for (int j = 0; j < W; ++j) {
        q[j] = al[0] - mid[j];
        cnt[j] = q[j] < 0 ? 1 : 0;
      }
      for (int64_t i = 1; i < m; ++i) {
        const double a = al[i], b = b2[(size_t)i];
#if defined(__clang__)
#pragma clang loop vectorize(enable) interleave(enable)
#elif defined(__GNUC__)
#pragma GCC ivdep
#endif
        for (int j = 0; j < W; ++j) {
          double t = a - mid[j] - b / q[j];
          cnt[j] += t < 0 ? 1 : 0;
          q[j] = t == 0 ? eps : t;
        }
      }
      for (int j = 0; j < w; ++j) {
        if (!live[j]) continue;
        if (cnt[j] >= ks[base + j] + 1) up[j] = mid[j]; else lo[j] = mid[j];
        if (mid[j] == pmid[j]) live[j] = false;
        pmid[j] = mid[j];
      }
    }
    for (int j = 0; j < w; ++j) out[base + j] = lo[j];
  }
}

// Eigenvectors of T(alpha, beta) for a FEW known eigenvalues by inverse iteration (LAPACK dstein-style): Gaussian
// elimination with partial pivoting on T - lambda I, three solves from a constant start vector, then modified
// Gram-Schmidt among vectors of (nearly) coincident eigenvalues.  O(m) per vector instead of the O(m^3) of the
// rotation-accumulating QR (TRI:223-231) — used by LL_TRIDIAG_AUTO for the final Ritz step when m is large, where
// only nroot <= 5 of the m eigenvectors are needed (LL:51-57).  out: nw rows of m entries, unit norm.
void tridiag_inverse_iteration(int64_t m, const double* al, const double* be, int64_t nw, const double* lambdas,
                               double* out) {
  double tnorm = 0.0;
  for (int64_t i = 0; i < m; ++i)
    tnorm = std::max(tnorm, std::abs(al[i]) + (i > 0 ? std::abs(be[i - 1]) : 0.0) + (i + 1 < m ? std::abs(be[i]) : 0.0));
  if (tnorm == 0.0) tnorm = 1.0;
  const double eps = std::numeric_limits<double>::epsilon();
  std::vector<double> d((size_t)m), du((size_t)m), du2((size_t)m), dl((size_t)m), x((size_t)m);
  std::vector<char> piv((size_t)m);
  for (int64_t w = 0; w < nw; ++w) {
    // tiny perturbation keeps the factorisation away from exact singularity and separates equal eigenvalues
    const double lam = lambdas[w] + (double)(w + 1) * 2.0 * eps * tnorm;
    for (int64_t i = 0; i < m; ++i) {
      d[i] = al[i] - lam;
      du[i] = i + 1 < m ? be[i] : 0.0;
      dl[i] = i + 1 < m ? be[i] : 0.0;
      du2[i] = 0.0;
    }
    // LU with partial pivoting (rows i, i+1)
    for (int64_t i = 0; i + 1 < m; ++i) {
      if (std::abs(d[i]) >= std::abs(dl[i])) {
        piv[i] = 0;
        if (d[i] == 0.0) d[i] = eps * tnorm;
        const double f = dl[i] / d[i];
        dl[i] = f;
        d[i + 1] -= f * du[i];
      } else {
        piv[i] = 1;
        const double f = d[i] / dl[i];
        d[i] = dl[i];
        dl[i] = f;
        const double t = d[i + 1];
        d[i + 1] = du[i] - f * t;
        du2[i] = i + 2 < m ? du[i + 1] : 0.0;
        du[i] = t;
        if (i + 2 < m) du[i + 1] = -f * du[i + 1];
      }
    }
    if (d[m - 1] == 0.0) d[m - 1] = eps * tnorm;
    // deterministic pseudo-random start, different for every wanted vector (a repeated eigenvalue needs start
    // vectors with independent components in its eigenspace)
    uint64_t st = 0x9E3779B97F4A7C15ull * (uint64_t)(w + 1);
    for (int64_t i = 0; i < m; ++i) {
      st = st * 6364136223846793005ull + 1442695040888963407ull;
      x[i] = ((double)(st >> 11) * (1.0 / 9007199254740992.0)) - 0.5;
    }
    for (int it = 0; it < 3; ++it) {
      // forward: apply L^-1 with the recorded row exchanges
      for (int64_t i = 0; i + 1 < m; ++i) {
        if (piv[i]) std::swap(x[i], x[i + 1]);
        x[i + 1] -= dl[i] * x[i];
      }
      // backward: U has three diagonals (d, du, du2)
      x[m - 1] /= d[m - 1];
      if (m > 1) x[m - 2] = (x[m - 2] - du[m - 2] * x[m - 1]) / d[m - 2];
      for (int64_t i = m - 3; i >= 0; --i) x[i] = (x[i] - du[i] * x[i + 1] - du2[i] * x[i + 2]) / d[i];
      // orthogonalise against previously computed vectors of close eigenvalues, then normalise
      for (int64_t v = 0; v < w; ++v)
        if (std::abs(lambdas[v] - lambdas[w]) <= 1e-3 * tnorm) {
          const double* o = out + v * m;
          double h = 0.0;
          for (int64_t i = 0; i < m; ++i) h += o[i] * x[i];
          for (int64_t i = 0; i < m; ++i) x[i] -= h * o[i];
        }
      double nn = 0.0;
      for (int64_t i = 0; i < m; ++i) nn += x[i] * x[i];
      nn = 1.0 / std::sqrt(nn);
      for (int64_t i = 0; i < m; ++i) x[i] *= nn;
    }
    std::copy(x.begin(), x.end(), out + w * m);
  }
}

}  // namespace ll
