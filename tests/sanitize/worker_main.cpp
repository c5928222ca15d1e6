// Sanitizer leg for the host step that runs on the helper thread (csrc/ritz_tracker.hpp: RitzTracker, ExpoTracker,
// StepWorker) — the only multi-threaded host code of the hot path.  Built twice by the Makefile next to it:
//   _build/worker_asan : -fsanitize=address,undefined
//   _build/worker_tsan : -fsanitize=thread          (the race detector the reference gets from its single thread)
// The driver makes (alpha, beta) sequences of Lanczos runs on known tridiagonal problems, feeds them to a threaded
// worker the way lanczos_run / expo_run do (submit while the arrays keep growing, opportunistic and fixed-lag
// consumption, early destruction with jobs outstanding) and requires the verdict sequence to equal the inline one.
// A clean run prints "worker sanitize ok".
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>

#include "ritz_tracker.hpp"

#define REQUIRE(cond)                                                                     \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      std::fprintf(stderr, "worker driver: %s failed (line %d)\n", #cond, __LINE__);      \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

namespace {
// Lanczos coefficients of diag(1..n) + a rank-one coupling, produced by the plain three-term recurrence with full
// re-orthogonalisation on the host (small n): a sequence that converges after a few dozen iterations.
void lanczos_coeffs(int n, int iters, std::vector<double>& alpha, std::vector<double>& beta) {
  std::vector<std::vector<double>> u;
  std::vector<double> v((size_t)n), w((size_t)n);
  double nrm = 0;
  for (int i = 0; i < n; ++i) v[(size_t)i] = 1.0 + 0.37 * std::sin(1.7 * i), nrm += v[(size_t)i] * v[(size_t)i];
  for (double& x : v) x /= std::sqrt(nrm);
  u.push_back(v);
  for (int k = 0; k < iters; ++k) {
    const std::vector<double>& x = u.back();
    double s = 0;
    for (int i = 0; i < n; ++i) s += x[(size_t)i];
    for (int i = 0; i < n; ++i) w[(size_t)i] = (1.0 + i * 20.0 / n) * x[(size_t)i] + 0.05 * s / n;
    double a = 0;
    for (int i = 0; i < n; ++i) a += x[(size_t)i] * w[(size_t)i];
    for (const std::vector<double>& q : u) {
      double h = 0;
      for (int i = 0; i < n; ++i) h += q[(size_t)i] * w[(size_t)i];
      for (int i = 0; i < n; ++i) w[(size_t)i] -= h * q[(size_t)i];
    }
    double b = 0;
    for (int i = 0; i < n; ++i) b += w[(size_t)i] * w[(size_t)i];
    b = std::sqrt(b);
    alpha.push_back(a);
    beta.push_back(b);
    if (b < 1e-13) break;
    for (int i = 0; i < n; ++i) w[(size_t)i] /= b;
    u.push_back(w);
  }
}

template <typename Tracker> struct Verdicts {
  std::vector<typename Tracker::Out> seq;
  int64_t stop_at = -1;
};

// lockstep_lag < 0: opportunistic policy; else fixed lag.  Mirrors the consumption loop of lanczos_run / expo_run.
template <typename Tracker>
Verdicts<Tracker> drive(const Tracker& cfg, bool threaded, int64_t lockstep_lag, const std::vector<double>& A,
                        const std::vector<double>& B, int64_t* enqueued) {
  Verdicts<Tracker> v;
  ll::StepWorker<Tracker> worker(cfg, threaded);
  std::vector<double> alpha, beta;  // grow while the worker reads its own copies
  bool stopped = false;
  auto absorb = [&](typename Tracker::Out& o) {
    v.seq.push_back(o);
    if (o.stop) v.stop_at = o.m;
    return o.stop;
  };
  const int64_t K = (int64_t)A.size();
  int64_t k = 1;
  for (; k <= K && !stopped; ++k) {
    if (k > 1) {
      alpha.push_back(A[(size_t)(k - 2)]);
      beta.push_back(B[(size_t)(k - 2)]);
      worker.submit((int64_t)alpha.size(), alpha.data(), beta.data());
      stopped = worker.consume(k - 1, lockstep_lag, threaded ? 24 : 0, absorb);
    }
  }
  *enqueued = k - 1;
  if (!stopped) {
    alpha.push_back(A[(size_t)(K - 1)]);
    beta.push_back(B[(size_t)(K - 1)]);
    worker.submit((int64_t)alpha.size(), alpha.data(), beta.data());
  }
  typename Tracker::Out r;
  while (!stopped && worker.wait_pop(r)) stopped = absorb(r);
  return v;  // ~StepWorker with jobs possibly outstanding (verdicts after the stop are abandoned)
}
}  // namespace

int main() {
  std::vector<double> A, B;
  lanczos_coeffs(400, 160, A, B);
  REQUIRE(A.size() >= 100);
  // ---- eigen-solver verdicts: QR, bisection, AUTO; 1 and 3 roots; both ends
  for (int mode : {LL_TRIDIAG_QR, LL_TRIDIAG_BISECT, LL_TRIDIAG_AUTO})
    for (int64_t nroot : {1, 3})
      for (bool fmax : {false, true}) {
        ll::RitzTracker cfg;
        cfg.nroot = nroot;
        cfg.find_maximum = fmax;
        cfg.mode = mode;
        cfg.eps = 1e-12;
        cfg.breakdown_tol = 2.2e-15;
        int64_t e0 = 0, e1 = 0, e2 = 0;
        Verdicts<ll::RitzTracker> inl = drive(cfg, false, -1, A, B, &e0);
        Verdicts<ll::RitzTracker> thr = drive(cfg, true, -1, A, B, &e1);
        Verdicts<ll::RitzTracker> fix = drive(cfg, true, 3, A, B, &e2);
        REQUIRE(inl.stop_at > 0 && inl.stop_at < (int64_t)A.size());  // the sequence converges inside the window
        REQUIRE(thr.stop_at == inl.stop_at && fix.stop_at == inl.stop_at);
        REQUIRE(thr.seq.size() == inl.seq.size() && fix.seq.size() == inl.seq.size());
        for (size_t i = 0; i < inl.seq.size(); ++i) {
          REQUIRE(thr.seq[i].m == inl.seq[i].m && thr.seq[i].evs == inl.seq[i].evs);  // bit-identical values
          REQUIRE(fix.seq[i].evs == inl.seq[i].evs && fix.seq[i].stop == inl.seq[i].stop);
        }
        REQUIRE(e2 == std::min<int64_t>((int64_t)A.size(), inl.stop_at + 3 + 1));  // fixed lag: a function of the verdicts only
      }
  // ---- exponentiator verdicts, real and complex exponent
  {
    ll::ExpoTracker<double> cfg;
    cfg.a = -0.3;
    cfg.eps = 1e-12;
    cfg.breakdown_tol = 2.2e-16;
    int64_t e0, e1, e2;
    auto inl = drive(cfg, false, -1, A, B, &e0);
    auto thr = drive(cfg, true, -1, A, B, &e1);
    auto fix = drive(cfg, true, 0, A, B, &e2);
    // real exponent: exp(a T) e_1 is not a unit vector, so the reference's overlap test (EX:154) never fires and the
    // window runs to its end — every verdict is computed and compared
    REQUIRE(inl.stop_at == -1 && thr.stop_at == -1 && fix.stop_at == -1);
    REQUIRE(inl.seq.size() == A.size() && thr.seq.size() == A.size() && fix.seq.size() == A.size());
    for (size_t i = 0; i < inl.seq.size(); ++i)
      REQUIRE(thr.seq[i].coeff == inl.seq[i].coeff && fix.seq[i].coeff == inl.seq[i].coeff);
    REQUIRE(e1 == (int64_t)A.size() && e2 == (int64_t)A.size());
  }
  {
    ll::ExpoTracker<std::complex<double>> cfg;
    cfg.a = std::complex<double>(0.0, -0.05);
    cfg.eps = 1e-12;
    cfg.breakdown_tol = 2.2e-16;
    int64_t e0, e1;
    auto inl = drive(cfg, false, -1, A, B, &e0);
    auto thr = drive(cfg, true, 2, A, B, &e1);
    REQUIRE(inl.stop_at > 0 && inl.stop_at < (int64_t)A.size() && thr.stop_at == inl.stop_at);
    REQUIRE(thr.seq.back().coeff == inl.seq.back().coeff);
    REQUIRE(e1 == inl.stop_at + 2 + 1);
    double n2 = 0;
    for (const std::complex<double>& c : inl.seq.back().coeff) n2 += std::norm(c);
    REQUIRE(std::abs(n2 - 1.0) < 1e-12);  // exp(i t T) e_1 is a unit vector
  }
  // ---- a worker that is destroyed with its queue full (run ended by an error on the enqueueing thread)
  {
    ll::RitzTracker cfg;
    cfg.mode = LL_TRIDIAG_QR;
    cfg.eps = 0.0;
    ll::StepWorker<ll::RitzTracker> worker(cfg, true);
    for (int64_t m = 1; m <= (int64_t)A.size(); ++m) worker.submit(m, A.data(), B.data());
    ll::RitzTracker::Out r;
    REQUIRE(worker.wait_pop(r) && r.m == 1);
  }
  std::printf("worker sanitize ok\n");
  return 0;
}
