// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the Krylov hot path of mrcdr/lambda-lanczos, written from the
// behaviour described in SURVEY.md (sections 3, 8 and Appendix A).  It is the *checker*
// for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may load the library built from this file.  The product (lambda-lanczos_amd/) never
// links, imports or calls it.
//
// Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement against
//   * the known answers of the reference's own tests (T1 = test/lambda_lanczos_test.cpp,
//     T2 = test/exponentiator_test.cpp), and
//   * fixtures under tests/golden/ that were captured from the real reference headers
//     compiled in place (oracle/ref_shim.cpp -> oracle/_ref/libref.so, see oracle/Makefile),
//   * and, whenever oracle/_ref/libref.so is present, directly against the real reference
//     on seeded inputs (alpha/beta traces, iteration counts, eigenpairs, exp(aA)v).
//
// Reference citations use the tags of SURVEY.md:
//   LL  = include/lambda_lanczos/lambda_lanczos.hpp
//   EX  = include/lambda_lanczos/exponentiator.hpp
//   LA  = include/lambda_lanczos/util/linear_algebra.hpp
//   CM  = include/lambda_lanczos/util/common.hpp
//   TRI = include/lambda_lanczos/lambda_lanczos_tridiagonal_impl.hpp
//   EPM = include/lambda_lanczos/eigenpair_manager.hpp
//
// The operator ("mv_mul", LL:120-126) is a CSR matrix here: the reference ships no sparse
// format, the CSR row loop below is what SURVEY.md section 8(d) names as the CPU baseline.
// Everything is single threaded and uses one heap vector per Lanczos vector, like the
// reference (LL:221, EX:90).

#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <random>
#include <vector>

#include <omp.h>

namespace oracle {

// Thread count of the "all cores" courtesy variant (bench.py cpu_baseline_all_cores).  1 = the reference's own
// behaviour: single thread, strict left-to-right sums (the stdpar path of the reference is dead code).
// With t > 1 every n-sized loop below is split over t threads; dot products then sum per-thread chunks in thread
// order (deterministic for a given t, rounding differs from t = 1).
static int g_threads = 1;
#define ORACLE_PAR _Pragma("omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)")

template <typename T> struct real_of { typedef T type; };                       // CM:80-102
template <typename R> struct real_of<std::complex<R>> { typedef R type; };
template <typename T> using real_t = typename real_of<T>::type;

inline double cj(double v) { return v; }                                         // CM:112-134
inline float cj(float v) { return v; }
template <typename R> inline std::complex<R> cj(const std::complex<R>& v) { return std::conj(v); }
inline double re(double v) { return v; }
inline float re(float v) { return v; }
template <typename R> inline R re(const std::complex<R>& v) { return v.real(); }

// ---------------------------------------------------------------- BLAS-1 (LA:29-163)

// <a,b> = sum conj(a_i) b_i, strict left fold (LA:29-51; conjugate-linear in arg 1, T1:47-59).
template <typename T> T inner_prod(const std::vector<T>& a, const std::vector<T>& b) {
  if (g_threads <= 1) {
    T acc = T();
    for (size_t i = 0; i < a.size(); ++i) acc = acc + cj(a[i]) * b[i];
    return acc;
  }
  std::vector<T> part((size_t)g_threads, T());
#pragma omp parallel num_threads(g_threads)
  {
    const size_t t = (size_t)omp_get_thread_num(), nt = (size_t)omp_get_num_threads(), n = a.size();
    const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
    T acc = T();
    for (size_t i = lo; i < hi; ++i) acc = acc + cj(a[i]) * b[i];
    part[t] = acc;
  }
  T acc = T();
  for (auto& p : part) acc = acc + p;
  return acc;
}
// sqrt(Re<v,v>), unscaled (LA:56-60).
template <typename T> real_t<T> norm2(const std::vector<T>& v) { return std::sqrt(re(inner_prod(v, v))); }
// v *= a (LA:65-72).
template <typename S, typename T> void scalar_mul(S a, std::vector<T>& v) {
  const int64_t n = (int64_t)v.size();
  ORACLE_PAR
  for (int64_t i = 0; i < n; ++i) v[(size_t)i] *= a;
}
// v *= T(1)/norm(v) (LA:77-80).
template <typename T> void normalize(std::vector<T>& v) { scalar_mul(T(1) / norm2(v), v); }
// sum |Re| + |Im| (LA:82-125).
inline double m_norm(const std::vector<double>& v) {
  double acc = 0;
  for (double e : v) acc = acc + std::abs(e);
  return acc;
}
inline double m_norm(const std::vector<std::complex<double>>& v) {
  double acc = 0;
  for (auto& e : v) acc = acc + std::abs(e.real()) + std::abs(e.imag());
  return acc;
}
// Modified Gram-Schmidt of w against an ordered set of orthonormal vectors (LA:132-144):
// for every basis vector: h = <u,w>, w -= h u.
template <typename T, typename It> void schmidt_orth(std::vector<T>& w, It first, It last) {
  for (It it = first; it != last; ++it) {
    const std::vector<T>& u = *it;
    T h = inner_prod(u, w);
    const int64_t n = (int64_t)w.size();
    ORACLE_PAR
    for (int64_t i = 0; i < n; ++i) w[(size_t)i] -= h * u[(size_t)i];
  }
}

// ---------------------------------------------------------------- tridiagonal QR (TRI:151-361)

template <typename R> inline R sgn(R v) { return v >= 0 ? R(1) : R(-1); }       // CM:194-201, sgn(0)=+1

// Givens pair eliminating z against x (TRI:151-166).
template <typename R> inline void givens(R x, R z, R& c, R& s) {
  if (z == 0) { c = 1; s = 0; return; }
  if (x == 0) { c = 0; s = 1; return; }
  R h = std::sqrt(x * x + z * z);
  c = x / h;
  s = z / h;
}

// One implicit Wilkinson-shift sweep over the block [off, off+ns) (TRI:181-236, Appendix A).
template <typename R>
void qr_sweep(std::vector<R>& al, std::vector<R>& be, std::vector<std::vector<R>>& q, size_t off, size_t ns,
              bool with_vectors) {
  if (ns == 1) return;
  const size_t e = off + ns - 1;
  R d = (al[e - 1] - al[e]) / (2 * be[e - 1]);
  R mu = al[e] - be[e - 1] / (d + sgn(d) * std::sqrt(d * d + R(1)));
  R x = al[off] - mu;
  R s = 1, c = 1, p = 0;
  for (size_t k = 0; k + 1 < ns; ++k) {
    R z = s * be[off + k];
    R bp = c * be[off + k];
    givens(x, z, c, s);
    if (k > 0) be[off + k - 1] = std::sqrt(x * x + z * z);
    R u = (al[off + k + 1] - al[off + k] + p) * s + R(2) * c * bp;
    al[off + k] = al[off + k] - p + s * u;
    p = s * u;
    x = c * u - bp;
    if (with_vectors) {  // rows k,k+1 of q rotated over all columns (TRI:223-231)
      std::vector<R>& r0 = q[off + k];
      std::vector<R>& r1 = q[off + k + 1];
      for (size_t j = 0; j < al.size(); ++j) {
        R v0 = r0[j], v1 = r1[j];
        r0[j] = c * v0 + s * v1;
        r1[j] = -s * v0 + c * v1;
      }
    }
  }
  al[e] = al[e] - p;
  be[e - 1] = x;
}

// Zero negligible couplings and return the trailing unreduced block (TRI:252-276).
template <typename R> void find_block(const std::vector<R>& al, std::vector<R>& be, size_t& first, size_t& last) {
  const R eps = std::numeric_limits<R>::epsilon() * R(0.5);
  const R tiny = std::numeric_limits<R>::min();
  const size_t n = al.size();
  for (size_t i = 0; i + 1 < n; ++i)
    if (std::abs(be[i]) < std::sqrt(std::abs(al[i]) * std::abs(al[i + 1])) * eps + tiny) be[i] = 0;
  while (last > 0 && be[last - 1] == 0) --last;
  first = last;
  while (first > 0 && be[first - 1] != 0) --first;
}

// All eigenpairs of T(alpha, beta); q[j][:] is eigenvector j; ascending order (TRI:290-343, CM:141-174).
// beta may be longer than n-1 (the Lanczos loop passes n entries, LL:262,268).
template <typename R>
size_t tridiag_eig(const std::vector<R>& alpha, const std::vector<R>& beta, std::vector<R>& ev,
                   std::vector<std::vector<R>>& q, bool with_vectors) {
  const size_t n = alpha.size();
  std::vector<R> al = alpha, be = beta;
  if (be.size() < n) be.resize(n, R(0));
  if (with_vectors) {  // LA:149-163
    q.assign(n, std::vector<R>(n, R(0)));
    for (size_t i = 0; i < n; ++i) q[i][i] = 1;
  }
  size_t unconverged = 0, last_prev = n - 1, loops = 1;
  while (true) {
    size_t last = last_prev, first;
    find_block(al, be, first, last);
    const size_t ns = last - first + 1;
    if (last == 0) break;
    qr_sweep(al, be, q, first, ns, with_vectors);
    if (last == last_prev) {
      if (loops > ns * 50) { last_prev = first; ++unconverged; loops = 1; }      // TRI:315-331
      else ++loops;
    } else { loops = 1; last_prev = last; }
  }
  // index sort ascending (CM:141-174)
  std::vector<std::pair<R, size_t>> order;
  for (size_t i = 0; i < n; ++i) order.emplace_back(al[i], i);
  std::sort(order.begin(), order.end(),
            [](const std::pair<R, size_t>& a, const std::pair<R, size_t>& b) { return a.first < b.first; });
  ev.resize(n);
  for (size_t i = 0; i < n; ++i) ev[i] = order[i].first;
  if (with_vectors) {
    std::vector<std::vector<R>> qs;
    for (size_t i = 0; i < n; ++i) qs.push_back(std::move(q[order[i].second]));
    q = std::move(qs);
  }
  return unconverged;
}

// Sturm count / bisection family (TRI:22-88) — unused by the engines, kept for the AUTO stop test of the
// product path (SURVEY section 7 "hard parts") so that it can be checked against the QR values.
template <typename R> size_t sturm_count(R c, const std::vector<R>& al, const std::vector<R>& be) {
  R q = al[0] - c;
  size_t cnt = q < 0 ? 1 : 0;
  for (size_t i = 1; i < al.size(); ++i) {
    q = al[i] - c - be[i - 1] * be[i - 1] / q;
    if (q < 0) ++cnt;
    if (q == 0) q = std::numeric_limits<R>::epsilon();
  }
  return cnt;
}
template <typename R> R mth_eigenvalue(const std::vector<R>& al, const std::vector<R>& be, size_t m) {
  std::vector<R> b(be.begin(), be.begin() + (al.size() - 1));
  R r = m_norm(al) + 2 * m_norm(b);  // TRI:52-58
  R lo = -r, up = r, mid, pmid = std::numeric_limits<R>::max();
  while (up - lo > std::min(std::abs(lo), std::abs(up)) * std::numeric_limits<R>::epsilon()) {
    mid = (lo + up) * R(0.5);
    if (sturm_count(mid, al, be) >= m + 1) up = mid; else lo = mid;
    if (mid == pmid) break;
    pmid = mid;
  }
  return lo;
}

// ---------------------------------------------------------------- operator: CSR row loop

template <typename T> struct Csr {
  int64_t n;
  const int64_t* rp;
  const int32_t* ci;
  const T* va;
  // out += A in  (the mv_mul contract LL:120-126: out is pre-zeroed by the engine)
  void apply(const std::vector<T>& in, std::vector<T>& out) const {
    ORACLE_PAR
    for (int64_t i = 0; i < n; ++i) {
      T acc = T();
      for (int64_t p = rp[i]; p < rp[i + 1]; ++p) acc += va[p] * in[ci[p]];
      out[i] += acc;
    }
  }
};

struct Params {  // mirrors the public fields LL:126-181 / EX:41-71
  int64_t matrix_size;
  int64_t max_iteration;
  double eps;
  int32_t find_maximum;
  int32_t full_orthogonalize;  // Exponentiator only (EX:63)
  int64_t num_eigs;
  double eigenvalue_offset;
  int64_t num_eigs_per_iteration;
};

struct Trace {  // optional instrumentation (not part of the reference API)
  double* alpha;   // capacity >= max_iteration
  double* beta;
  int64_t* len;    // number of alpha/beta entries written (last pass)
  double* t_mv;    // seconds inside mv_mul
  double* t_total; // seconds inside the run
};

static inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------- Ritz vectors (LL:33-62)

template <typename T>
std::vector<std::vector<T>> ritz_vectors(const std::vector<real_t<T>>& alpha, const std::vector<real_t<T>>& beta,
                                         const std::vector<std::vector<T>>& u, bool find_max, size_t nev) {
  typedef real_t<T> R;
  const size_t m = alpha.size(), n = u[0].size();
  std::vector<R> tev;
  std::vector<std::vector<R>> tq;
  tridiag_eig(alpha, beta, tev, tq, true);
  std::vector<std::vector<T>> x(nev, std::vector<T>(n));
  for (size_t r = 0; r < nev; ++r) {
    const size_t it = find_max ? m - r - 1 : r;
    for (size_t k = m; k-- > 0;) {                                 // k = m-1 .. 0 (LL:53)
      ORACLE_PAR
      for (int64_t i = 0; i < (int64_t)n; ++i) x[r][(size_t)i] += tq[it][k] * u[k][(size_t)i];
    }
    normalize(x[r]);
  }
  return x;
}

// ---------------------------------------------------------------- one Lanczos pass (LL:216-322)

template <typename T, typename LockedIt>
size_t lanczos_pass(const Csr<T>& A, const Params& P, const T* init, size_t nroot, LockedIt lfirst, LockedIt llast,
                    std::vector<real_t<T>>& out_ev, std::vector<std::vector<T>>& out_x, Trace* tr) {
  typedef real_t<T> R;
  const size_t n = (size_t)P.matrix_size;
  std::vector<std::vector<T>> u;
  std::vector<R> alpha, beta;
  u.emplace_back(init, init + n);                                  // init_vector hook result (LL:231-232)
  schmidt_orth(u[0], lfirst, llast);                               // LL:233
  normalize(u[0]);                                                 // LL:234
  std::vector<R> evs, pevs;
  size_t itern = (size_t)P.max_iteration;
  for (size_t k = 1; k <= (size_t)P.max_iteration; ++k) {
    std::vector<T> au(n, T());                                     // P0 LL:242
    double t0 = tr ? now_s() : 0;
    A.apply(u[k - 1], au);                                         // P1 LL:243
    if (tr && tr->t_mv) *tr->t_mv += now_s() - t0;
    ORACLE_PAR
    for (int64_t i = 0; i < (int64_t)n; ++i) au[(size_t)i] += u[k - 1][(size_t)i] * (R)P.eigenvalue_offset;   // P2 LL:244-246
    alpha.push_back(re(inner_prod(u[k - 1], au)));                 // P3 LL:248
    u.push_back(std::move(au));
    ORACLE_PAR
    for (int64_t ii = 0; ii < (int64_t)n; ++ii) {                   // P4 LL:251-257
      const size_t i = (size_t)ii;
      if (k == 1) u[k][i] = u[k][i] - alpha[k - 1] * u[k - 1][i];
      else u[k][i] = u[k][i] - beta[k - 2] * u[k - 2][i] - alpha[k - 1] * u[k - 1][i];
    }
    schmidt_orth(u[k], lfirst, llast);                             // P5 LL:259
    schmidt_orth(u[k], u.begin(), u.end() - 1);                    // P6 LL:260
    beta.push_back(norm2(u[k]));                                   // P7 LL:262
    const size_t ncalc = std::min(nroot, alpha.size());
    evs.clear();
    std::vector<R> all;
    std::vector<std::vector<R>> dummy;
    tridiag_eig(alpha, beta, all, dummy, false);                   // H1 LL:267-268
    for (size_t i = 0; i < ncalc; ++i) evs.push_back(P.find_maximum ? all[all.size() - i - 1] : all[i]);  // H2
    if (beta.back() < std::numeric_limits<R>::epsilon() * R(1e1)) { itern = k; break; }   // H3 LL:279-283
    normalize(u[k]);                                               // P8 LL:285
    bool stop = true;                                              // H4 LL:290-309
    if (pevs.size() != evs.size()) stop = false;
    else
      for (size_t r = 0; r < nroot; ++r)
        if (std::abs(evs[r] - pevs[r]) >= std::min(std::abs(evs[r]), std::abs(pevs[r])) * (R)P.eps) { stop = false; break; }
    if (stop) { itern = k; break; }
    pevs = evs;
  }
  out_ev = evs;
  beta.back() = 0;                                                 // LL:314
  if (tr && tr->alpha) {
    for (size_t i = 0; i < alpha.size(); ++i) { tr->alpha[i] = (double)alpha[i]; tr->beta[i] = (double)beta[i]; }
    *tr->len = (int64_t)alpha.size();
  }
  out_x = ritz_vectors<T>(alpha, beta, u, P.find_maximum != 0, out_ev.size());   // LL:316
  for (auto& e : out_ev) e -= (R)P.eigenvalue_offset;               // LL:317-319
  return itern;
}

// ---------------------------------------------------------------- restart loop (LL:330-366, EPM:21-80)

typedef void (*init_fn)(void* vec, int64_t n, void* user);

template <typename T>
int64_t lanczos_run(const Csr<T>& A, const Params& P, init_fn init, void* user, real_t<T>* eigvals, T* eigvecs,
                    int64_t* iter_counts, int64_t* n_pass, Trace* tr) {
  typedef real_t<T> R;
  const size_t n = (size_t)P.matrix_size;
  std::function<bool(R, R)> cmp;
  if (P.find_maximum) cmp = std::greater<R>(); else cmp = std::less<R>();
  std::multimap<R, std::vector<T>, std::function<bool(R, R)>> kept(cmp);   // EPM:32-46
  int64_t passes = 0;
  double t0 = now_s();
  while (true) {
    const size_t nroot = std::min((size_t)P.num_eigs_per_iteration, n - kept.size());   // LL:338
    std::vector<T> start(n);
    init(start.data(), (int64_t)n, user);
    std::vector<const std::vector<T>*> lockp;                      // iteration order = comparator order (CM:58-74)
    for (auto& kv : kept) lockp.push_back(&kv.second);
    struct It {                                                    // forward iterator over locked vectors
      typename std::vector<const std::vector<T>*>::const_iterator p;
      const std::vector<T>& operator*() const { return **p; }
      It& operator++() { ++p; return *this; }
      bool operator!=(const It& o) const { return p != o.p; }
    };
    std::vector<R> ev;
    std::vector<std::vector<T>> x;
    size_t it = lanczos_pass<T>(A, P, start.data(), nroot, It{lockp.cbegin()}, It{lockp.cend()}, ev, x, tr);
    iter_counts[passes++] = (int64_t)it;
    bool nothing_added = true;                                     // EPM:52-71
    for (size_t i = 0; i < ev.size(); ++i) {
      auto ins = kept.emplace(ev[i], std::move(x[i]));
      auto last = kept.end(); --last;
      if (kept.size() > (size_t)P.num_eigs) {
        if (ins != last) nothing_added = false;
        kept.erase(last);
      } else nothing_added = false;
    }
    if (nothing_added) break;                                      // LL:346-348
    if (P.num_eigs == 1) break;                                    // LL:350-353
  }
  int64_t cnt = 0;
  for (auto& kv : kept) {                                          // LL:356-365: comparator order
    eigvals[cnt] = kv.first;
    std::memcpy(eigvecs + (size_t)cnt * n, kv.second.data(), n * sizeof(T));
    ++cnt;
  }
  *n_pass = passes;
  if (tr && tr->t_total) *tr->t_total += now_s() - t0;
  return cnt;
}

// ---------------------------------------------------------------- Exponentiator::run (EX:87-173)

template <typename T>
int64_t expo_run(const Csr<T>& A, const Params& P, T a, const T* input, T* output, Trace* tr) {
  typedef real_t<T> R;
  const size_t n = (size_t)P.matrix_size;
  double t0 = now_s();
  std::vector<std::vector<T>> u;
  std::vector<R> alpha, beta;
  u.emplace_back(input, input + n);                                // EX:100
  normalize(u[0]);                                                 // EX:101
  std::vector<T> coeff_prev;
  size_t itern = (size_t)P.max_iteration;
  for (size_t k = 1; k <= (size_t)P.max_iteration; ++k) {
    u.emplace_back(n, T());                                        // EX:107
    double t1 = tr ? now_s() : 0;
    A.apply(u[k - 1], u[k]);                                       // EX:108
    if (tr && tr->t_mv) *tr->t_mv += now_s() - t1;
    alpha.push_back(re(inner_prod(u[k - 1], u[k])));               // EX:110
    for (size_t i = 0; i < n; ++i) {                               // EX:112-118
      if (k == 1) u[k][i] = u[k][i] - alpha[k - 1] * u[k - 1][i];
      else u[k][i] = u[k][i] - beta[k - 2] * u[k - 2][i] - alpha[k - 1] * u[k - 1][i];
    }
    if (P.full_orthogonalize) schmidt_orth(u[k], u.begin(), u.end() - 1);   // EX:120-122
    std::vector<R> ev;
    std::vector<std::vector<R>> p;
    tridiag_eig(alpha, beta, ev, p, true);                         // EX:124-126
    const size_t m = alpha.size();
    std::vector<T> coeff(m, T());
    for (size_t i = 0; i < m; ++i)                                 // EX:128-133
      for (size_t j = 0; j < m; ++j) coeff[i] += p[j][i] * std::exp(a * ev[j]) * p[j][0];
    beta.push_back(norm2(u[k]));                                   // EX:145
    T overlap = T();
    for (size_t i = 0; i < coeff_prev.size(); ++i) overlap += cj(coeff_prev[i]) * coeff[i];   // EX:147-150
    coeff_prev = std::move(coeff);                                 // EX:152
    if (std::abs(R(1) - std::abs(overlap)) < (R)P.eps || beta.back() < std::numeric_limits<R>::epsilon()) {   // EX:154-158
      itern = k;
      break;
    }
    normalize(u[k]);                                               // EX:160
  }
  std::vector<T> in(input, input + n);
  const T nrm = norm2(in);                                         // EX:165
  for (size_t i = 0; i < n; ++i) output[i] = T();
  for (size_t l = 0; l < coeff_prev.size(); ++l)                   // EX:166-170
    for (size_t i = 0; i < n; ++i) output[i] += nrm * coeff_prev[l] * u[l][i];
  if (tr && tr->alpha) {
    for (size_t i = 0; i < alpha.size(); ++i) { tr->alpha[i] = (double)alpha[i]; tr->beta[i] = (double)beta[i]; }
    *tr->len = (int64_t)alpha.size();
  }
  if (tr && tr->t_total) *tr->t_total += now_s() - t0;
  return (int64_t)itern;
}

// ---------------------------------------------------------------- Exponentiator::taylor_run (EX:175-210)

template <typename T>
int64_t taylor_run(const Csr<T>& A, const Params& P, T a, const T* input, T* output) {
  const size_t n = (size_t)P.matrix_size;
  if (a == T()) { std::memcpy(output, input, n * sizeof(T)); return 1; }
  std::vector<std::vector<T>> ts;
  ts.emplace_back(input, input + n);
  T factor = 1.0;
  for (size_t k = 1;; ++k) {
    factor *= a / (T)(double)k;
    ts.emplace_back(n, T());
    A.apply(ts[k - 1], ts[k]);
    if (norm2(ts[k]) * std::abs(factor) < P.eps) break;
  }
  for (size_t i = 0; i < n; ++i) output[i] = T();
  for (size_t k = ts.size(); k-- > 0;) {
    for (size_t i = 0; i < n; ++i) output[i] += ts[k][i] * factor;
    factor *= (T)(double)k / a;
  }
  return (int64_t)ts.size();
}

struct FixedInit { const void* data; size_t bytes; };
static void fixed_init(void* vec, int64_t, void* user) {
  const FixedInit* f = (const FixedInit*)user;
  std::memcpy(vec, f->data, f->bytes);
}

// LambdaLanczos<T>::run_iteration (LL:216-322) on its own: one pass, nroot pairs, caller-provided orthogonalizeTo
// (n_orth vectors, vector j at orth + j*n).  Returns the iteration count; *n_found pairs are written.
template <typename T>
int64_t run_iteration_c(const int64_t* rp, const int32_t* ci, const T* va, const Params& P, const T* init,
                               int64_t nroot, int64_t n_orth, const T* orth, double* eigvals, T* eigvecs,
                               int64_t* n_found) {
  const size_t n = (size_t)P.matrix_size;
  Csr<T> A{P.matrix_size, rp, ci, va};
  std::vector<std::vector<T>> lock;
  for (int64_t j = 0; j < n_orth; ++j) lock.emplace_back(orth + (size_t)j * n, orth + (size_t)(j + 1) * n);
  std::vector<real_t<T>> ev;
  std::vector<std::vector<T>> x;
  const size_t it = lanczos_pass<T>(A, P, init, (size_t)nroot, lock.cbegin(), lock.cend(), ev, x, nullptr);
  for (size_t i = 0; i < ev.size(); ++i) {
    eigvals[i] = (double)ev[i];
    std::memcpy(eigvecs + i * n, x[i].data(), n * sizeof(T));
  }
  *n_found = (int64_t)ev.size();
  return (int64_t)it;
}
}  // namespace oracle

// ================================================================= C ABI (ctypes)
using namespace oracle;
typedef std::complex<double> zd;

extern "C" {

// threads <= 0: all host cores.  Returns the count in effect.
int oracle_set_threads(int threads) {
  g_threads = threads > 0 ? threads : omp_get_max_threads();
  return g_threads;
}

struct oracle_params {  // layout shared with tests/oracle_lib.py
  int64_t matrix_size, max_iteration;
  double eps;
  int32_t find_maximum, full_orthogonalize;
  int64_t num_eigs;
  double eigenvalue_offset;
  int64_t num_eigs_per_iteration;
};
struct oracle_trace { double* alpha; double* beta; int64_t* len; double* t_mv; double* t_total; };

static Params cvt(const oracle_params* p) {
  Params q;
  q.matrix_size = p->matrix_size; q.max_iteration = p->max_iteration; q.eps = p->eps;
  q.find_maximum = p->find_maximum; q.full_orthogonalize = p->full_orthogonalize; q.num_eigs = p->num_eigs;
  q.eigenvalue_offset = p->eigenvalue_offset; q.num_eigs_per_iteration = p->num_eigs_per_iteration;
  return q;
}

// y = A x (y overwritten) — CSR row loop, the a1 checker.
void oracle_spmv_d(int64_t n, const int64_t* rp, const int32_t* ci, const double* va, const double* x, double* y) {
  for (int64_t i = 0; i < n; ++i) { double s = 0; for (int64_t p = rp[i]; p < rp[i + 1]; ++p) s += va[p] * x[ci[p]]; y[i] = s; }
}
void oracle_spmv_z(int64_t n, const int64_t* rp, const int32_t* ci, const zd* va, const zd* x, zd* y) {
  for (int64_t i = 0; i < n; ++i) { zd s = 0; for (int64_t p = rp[i]; p < rp[i + 1]; ++p) s += va[p] * x[ci[p]]; y[i] = s; }
}

void oracle_inner_prod_z(int64_t n, const zd* a, const zd* b, zd* out) {
  std::vector<zd> va(a, a + n), vb(b, b + n);
  *out = inner_prod(va, vb);
}
double oracle_inner_prod_d(int64_t n, const double* a, const double* b) {
  std::vector<double> va(a, a + n), vb(b, b + n);
  return inner_prod(va, vb);
}
double oracle_m_norm_z(int64_t n, const zd* a) { std::vector<zd> v(a, a + n); return m_norm(v); }
double oracle_m_norm_d(int64_t n, const double* a) { std::vector<double> v(a, a + n); return m_norm(v); }

// MGS of w (n) against nb row-major basis vectors.
void oracle_schmidt_orth_d(int64_t n, int64_t nb, const double* basis, double* w) {
  std::vector<std::vector<double>> us;
  for (int64_t j = 0; j < nb; ++j) us.emplace_back(basis + j * n, basis + (j + 1) * n);
  std::vector<double> v(w, w + n);
  schmidt_orth(v, us.begin(), us.end());
  std::copy(v.begin(), v.end(), w);
}
void oracle_schmidt_orth_z(int64_t n, int64_t nb, const zd* basis, zd* w) {
  std::vector<std::vector<zd>> us;
  for (int64_t j = 0; j < nb; ++j) us.emplace_back(basis + j * n, basis + (j + 1) * n);
  std::vector<zd> v(w, w + n);
  schmidt_orth(v, us.begin(), us.end());
  std::copy(v.begin(), v.end(), w);
}

// ev[n] ascending; q row-major n*n (q[j*n + :] = eigenvector j) when q != NULL. Returns unconverged count.
int64_t oracle_tridiag_eig(int64_t n, const double* alpha, const double* beta, int64_t nbeta, double* ev, double* q) {
  std::vector<double> al(alpha, alpha + n), be(beta, beta + nbeta), e;
  std::vector<std::vector<double>> qq;
  size_t unc = tridiag_eig(al, be, e, qq, q != nullptr);
  std::copy(e.begin(), e.end(), ev);
  if (q) for (int64_t j = 0; j < n; ++j) std::copy(qq[j].begin(), qq[j].end(), q + j * n);
  return (int64_t)unc;
}
double oracle_mth_eigenvalue(int64_t n, const double* alpha, const double* beta, int64_t m) {
  std::vector<double> al(alpha, alpha + n), be(beta, beta + n);
  return mth_eigenvalue(al, be, (size_t)m);
}

int64_t oracle_lanczos_run_d(const int64_t* rp, const int32_t* ci, const double* va, const oracle_params* p,
                             const double* init, double* eigvals, double* eigvecs, int64_t* iter_counts,
                             int64_t* n_pass, oracle_trace* tr) {
  Csr<double> A{p->matrix_size, rp, ci, va};
  FixedInit f{init, (size_t)p->matrix_size * sizeof(double)};
  return lanczos_run<double>(A, cvt(p), fixed_init, &f, eigvals, eigvecs, iter_counts, n_pass, (Trace*)tr);
}
int64_t oracle_lanczos_run_z(const int64_t* rp, const int32_t* ci, const zd* va, const oracle_params* p,
                             const zd* init, double* eigvals, zd* eigvecs, int64_t* iter_counts, int64_t* n_pass,
                             oracle_trace* tr) {
  Csr<zd> A{p->matrix_size, rp, ci, va};
  FixedInit f{init, (size_t)p->matrix_size * sizeof(zd)};
  return lanczos_run<zd>(A, cvt(p), fixed_init, &f, eigvals, eigvecs, iter_counts, n_pass, (Trace*)tr);
}
int64_t oracle_run_iteration_d(const int64_t* rp, const int32_t* ci, const double* va, const oracle_params* p,
                               const double* init, int64_t nroot, int64_t n_orth, const double* orth, double* eigvals,
                               double* eigvecs, int64_t* n_found) {
  return run_iteration_c<double>(rp, ci, va, cvt(p), init, nroot, n_orth, orth, eigvals, eigvecs, n_found);
}
int64_t oracle_run_iteration_z(const int64_t* rp, const int32_t* ci, const zd* va, const oracle_params* p,
                               const zd* init, int64_t nroot, int64_t n_orth, const zd* orth, double* eigvals,
                               zd* eigvecs, int64_t* n_found) {
  return run_iteration_c<zd>(rp, ci, va, cvt(p), init, nroot, n_orth, orth, eigvals, eigvecs, n_found);
}
int64_t oracle_expo_run_d(const int64_t* rp, const int32_t* ci, const double* va, const oracle_params* p, double a,
                          const double* input, double* output, oracle_trace* tr) {
  Csr<double> A{p->matrix_size, rp, ci, va};
  return expo_run<double>(A, cvt(p), a, input, output, (Trace*)tr);
}
int64_t oracle_expo_run_z(const int64_t* rp, const int32_t* ci, const zd* va, const oracle_params* p, double a_re,
                          double a_im, const zd* input, zd* output, oracle_trace* tr) {
  Csr<zd> A{p->matrix_size, rp, ci, va};
  return expo_run<zd>(A, cvt(p), zd(a_re, a_im), input, output, (Trace*)tr);
}
int64_t oracle_taylor_run_d(const int64_t* rp, const int32_t* ci, const double* va, const oracle_params* p, double a,
                            const double* input, double* output) {
  Csr<double> A{p->matrix_size, rp, ci, va};
  return taylor_run<double>(A, cvt(p), a, input, output);
}
int64_t oracle_taylor_run_z(const int64_t* rp, const int32_t* ci, const zd* va, const oracle_params* p, double a_re,
                            double a_im, const zd* input, zd* output) {
  Csr<zd> A{p->matrix_size, rp, ci, va};
  return taylor_run<zd>(A, cvt(p), zd(a_re, a_im), input, output);
}

}  // extern "C"
