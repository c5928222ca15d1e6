// The host scalar step of the eigen-solver loop (LL:264-309): Ritz values of T_m, selection of the
// nroot extremes, breakdown test, convergence test — and the helper thread that runs it OFF the enqueueing thread
// (SURVEY section 7 step 5), so that the O(m) .. O(m^2) scalar work of iteration j never delays the launches of
// iterations j+1, j+2, ...
#pragma once

#include <chrono>
#include <unistd.h>

#include <cmath>
#include <complex>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "ll_internal.hpp"
#include "trace.hpp"

namespace ll {

// One call per Lanczos iteration, in order.  Keeps the previous iteration's Ritz values (LL:266 `pevs`).
struct RitzTracker {
  // configuration
  int64_t nroot = 1;
  bool find_maximum = false;
  int mode = LL_TRIDIAG_AUTO;
  double eps = 0.0;
  double breakdown_tol = 0.0;  // LL:279: 10 * epsilon of real_t<T>

  struct Out {
    int64_t m = 0;
    bool stop = false;           // H3 (breakdown) or H4 (converged): the loop ends after m iterations
    bool evs_from_qr = true;     // the values are those of the reference's QR arithmetic (else: bisection values)
    std::vector<double> evs;     // the nroot extreme Ritz values of T_m, comparator order
    double seconds = 0.0;
  };

  std::vector<double> pevs, all;
  // LL_TRIDIAG_AUTO bookkeeping: ||T||_inf so far (noise scale of the two eigenvalue methods), and the extreme QR values
  // of the last confirmation (T_{m-1} of the next one)
  double tnorm = 0.0;
  int64_t qr_m = 0;
  std::vector<double> qr_ext;

  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

  bool converged(const std::vector<double>& cur, const std::vector<double>& before, double tol) const {
    if (before.size() != cur.size()) return false;
    for (int64_t r = 0; r < nroot; ++r)  // LL:296-303: every tracked root changed by less than eps (relative)
      if (std::abs(cur[(size_t)r] - before[(size_t)r]) >= std::min(std::abs(cur[(size_t)r]), std::abs(before[(size_t)r])) * tol)
        return false;
    return true;
  }

  // alpha[0..m), beta[0..m) (beta[m-1] = ||w_m||)
  Out step(int64_t m, const double* alpha, const double* beta) {
    TraceRange trace("ll::host_tridiag (Ritz values + stop test)");
    Out o;
    o.m = m;
    const double t0 = now();
    const int64_t ncalc = std::min<int64_t>(nroot, m);
    const bool use_qr = mode == LL_TRIDIAG_QR || (mode == LL_TRIDIAG_AUTO && m <= 64);
    if (use_qr) {
      all.resize((size_t)m);
      tridiag_qr(m, alpha, beta, all.data(), nullptr);  // H1 LL:267-268
      for (int64_t i = 0; i < ncalc; ++i) o.evs.push_back(find_maximum ? all[(size_t)(m - i - 1)] : all[(size_t)i]);  // H2
      o.evs_from_qr = true;
    } else {
      o.evs_from_qr = false;
      int64_t ks[8];
      double vals[8];
      for (int64_t i0 = 0; i0 < ncalc; i0 += 8) {  // all wanted roots in one interleaved bisection (tridiag_host.cpp)
        const int w = (int)std::min<int64_t>(8, ncalc - i0);
        for (int j = 0; j < w; ++j) ks[j] = find_maximum ? m - (i0 + j) - 1 : i0 + j;
        tridiag_bisect_multi(m, alpha, beta, w, ks, vals);
        for (int j = 0; j < w; ++j) o.evs.push_back(vals[j]);
      }
    }
    // ||T_m||_inf, updated with the new row
    {
      const double bl = m >= 2 ? std::abs(beta[m - 2]) : 0.0;
      tnorm = std::max(tnorm, std::abs(alpha[m - 1]) + bl + std::abs(beta[m - 1]));
      if (m >= 2) tnorm = std::max(tnorm, std::abs(alpha[m - 2]) + (m >= 3 ? std::abs(beta[m - 3]) : 0.0) + bl);
    }
    // H3 LL:279-283: 10 * machine epsilon of real_t<T> (float storage => the float epsilon, like the reference)
    if (beta[m - 1] < breakdown_tol) {
      o.stop = true;
      o.seconds = now() - t0;
      return o;
    }
    // H4 LL:290-309
    bool stop;
    if (use_qr) {
      stop = converged(o.evs, pevs, eps);
    } else if (mode != LL_TRIDIAG_AUTO) {
      stop = converged(o.evs, pevs, eps);  // LL_TRIDIAG_BISECT: bisection values decide alone
    } else {
      // The reference decides on the QR values of T_m and T_{m-1}.  Bisection values differ from them only by rounding
      // noise of the order eps_machine * ||T|| (both methods are backward stable; the bound below is 16 eps_machine
      // ||T||_inf, several times what either method shows in practice), so with D = |theta_m - theta_{m-1}| from
      // bisection and thr = eps * min(|theta_m|, |theta_{m-1}|):
      //   some root with D >= thr + noise   =>  the reference's test fails too: continue            (O(m) per iteration)
      //   every root with D <  thr - noise  =>  the reference's test passes too: stop, and return the QR values of T_m
      //   otherwise (a sliver of +-noise around the threshold): compute the reference's QR pair and let it decide —
      //             one QR per iteration there, T_{m-1}'s values are kept from the previous confirmation.
      // Iteration counts and returned eigenvalues therefore equal LL_TRIDIAG_QR's (checked on the reference's golden
      // traces and on runs of several hundred iterations in tests/), at O(m) instead of O(m^2) per iteration.
      const double noise = 16.0 * std::numeric_limits<double>::epsilon() * tnorm;
      bool maybe = pevs.size() == o.evs.size(), certain = maybe;
      for (int64_t r = 0; maybe && r < nroot; ++r) {
        const double thr = std::min(std::abs(o.evs[(size_t)r]), std::abs(pevs[(size_t)r])) * eps;
        const double d = std::abs(o.evs[(size_t)r] - pevs[(size_t)r]);
        if (d >= thr + noise) maybe = false;
        if (!(d < thr - noise)) certain = false;
      }
      stop = false;
      if (maybe) {
        std::vector<double> cur((size_t)m), e_now, e_before;
        tridiag_qr(m, alpha, beta, cur.data(), nullptr);
        for (int64_t i = 0; i < ncalc; ++i) e_now.push_back(find_maximum ? cur[(size_t)(m - i - 1)] : cur[(size_t)i]);
        if (certain) {
          stop = true;
        } else {
          if (qr_m == m - 1) {
            e_before = qr_ext;
          } else {
            std::vector<double> prev((size_t)(m - 1));
            tridiag_qr(m - 1, alpha, beta, prev.data(), nullptr);
            for (int64_t i = 0; i < std::min<int64_t>(nroot, m - 1); ++i)
              e_before.push_back(find_maximum ? prev[(size_t)(m - 2 - i)] : prev[(size_t)i]);
          }
          stop = converged(e_now, e_before, eps);
        }
        qr_m = m;
        qr_ext = e_now;
        if (stop) {
          o.evs = e_now;
          o.evs_from_qr = true;
        }
      }
    }
    o.stop = stop;
    if (!stop) pevs = o.evs;
    o.seconds = now() - t0;
    return o;
  }
};

// The host step of Exponentiator<T>::run (EX:124-158): eigenpairs of T_m, coeff = exp(a T_m) e_1,
// overlap with the previous iteration's coefficients, stop test.  H = double or std::complex<double>.
template <typename H> struct ExpoTracker {
  H a = H(0);
  double eps = 0.0;
  double breakdown_tol = 0.0;  // EX:154: epsilon of real_t<T>

  struct Out {
    int64_t m = 0;
    bool stop = false;
    std::vector<H> coeff;  // exp(a T_m) e_1: what the output sum uses when the loop ends at m (EX:163-170)
    double seconds = 0.0;
  };

  std::vector<H> coeff_prev, expv;
  std::vector<double> ev, p;

  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  static double conj_of(double v) { return v; }
  static std::complex<double> conj_of(std::complex<double> v) { return std::conj(v); }

  // alpha[0..m), beta[0..m): beta[m-1] = ||w_m|| (EX:145); the QR reads beta[0..m-2]
  Out step(int64_t m, const double* alpha, const double* beta) {
    TraceRange trace("ll::host_tridiag (exp(a T_k) e_1 + overlap test)");
    Out o;
    o.m = m;
    const double t0 = now();
    ev.resize((size_t)m);
    p.resize((size_t)m * m);
    tridiag_qr(m, alpha, beta, ev.data(), p.data());  // EX:124-126
    o.coeff.assign((size_t)m, H(0));
    expv.resize((size_t)m);
    for (int64_t j = 0; j < m; ++j) expv[(size_t)j] = std::exp(a * ev[(size_t)j]);  // m exponentials instead of m^2
    for (int64_t i = 0; i < m; ++i)  // EX:128-133: (exp(a T_m) e_1)_i, same product order as the reference
      for (int64_t j = 0; j < m; ++j) o.coeff[(size_t)i] += p[(size_t)j * m + i] * expv[(size_t)j] * p[(size_t)j * m];
    H overlap = H(0);
    for (size_t i = 0; i < coeff_prev.size(); ++i) overlap += conj_of(coeff_prev[i]) * o.coeff[i];  // EX:147-150
    coeff_prev = o.coeff;                                                                          // EX:152
    o.stop = std::abs(1.0 - std::abs(overlap)) < eps || beta[m - 1] < breakdown_tol;               // EX:154-158
    o.seconds = now() - t0;
    return o;
  }
};

// A single helper thread that feeds a tracker (RitzTracker / ExpoTracker) with the iterations in order.  submit()
// copies the (alpha, beta) prefixes, so the enqueueing thread may keep appending.  Results come back in order.
template <typename Tracker> class StepWorker {
 public:
  typedef typename Tracker::Out Out;
  // jitter_us (test hook, LL_TRIDIAG_TEST_JITTER_US): every verdict is held back by a pseudo-random time up to this many
  // microseconds, differently in every process, to show that ranks of a sharded run still enqueue the same iterations
  // (tests/test_gpu_multirank.py)
  StepWorker(const Tracker& cfg, bool threaded, int jitter_us = 0) : tracker_(cfg), threaded_(threaded), jitter_us_(jitter_us) {
    if (threaded_) thread_ = std::thread([this] { run(); });
  }
  ~StepWorker() {
    if (threaded_) {
      {
        std::lock_guard<std::mutex> g(mu_);
        quit_ = true;
      }
      cv_job_.notify_all();
      thread_.join();
    }
  }
  StepWorker(const StepWorker&) = delete;
  StepWorker& operator=(const StepWorker&) = delete;

  void submit(int64_t m, const double* alpha, const double* beta) {
    if (!threaded_) {
      done_.push_back(tracker_.step(m, alpha, beta));
      return;
    }
    Job j;
    j.m = m;
    j.alpha.assign(alpha, alpha + m);
    j.beta.assign(beta, beta + m);
    {
      std::lock_guard<std::mutex> g(mu_);
      jobs_.push_back(std::move(j));
      ++outstanding_;
    }
    cv_job_.notify_one();
  }
  // submitted jobs whose results have not been popped yet
  size_t outstanding() {
    if (!threaded_) return done_.size();
    std::lock_guard<std::mutex> g(mu_);
    return outstanding_;
  }
  bool try_pop(Out& out) {
    std::unique_lock<std::mutex> g(mu_, std::defer_lock);
    if (threaded_) g.lock();
    if (done_.empty()) return false;
    out = std::move(done_.front());
    done_.pop_front();
    ++popped_;
    if (threaded_) --outstanding_;
    return true;
  }
  // How the enqueueing thread consumes verdicts once iteration `collected` has been submitted; true = stop.
  //   lockstep_lag < 0 (one process): whatever has arrived, and wait only when more than max_lag are outstanding.
  //   lockstep_lag >= 0 (sharded context): exactly the verdicts up to collected - lockstep_lag, waiting for them.  Every
  //   rank holds the same (all-reduced) alpha/beta and so reaches the same verdicts, but WHEN a helper thread delivers
  //   one differs from rank to rank; a rank that saw the stop one iteration earlier than its peers would leave them
  //   alone in the next iteration's collectives.  A fixed lag makes the number of enqueued iterations a function of
  //   the verdicts only.
  template <typename Absorb> bool consume(int64_t collected, int64_t lockstep_lag, size_t max_lag, Absorb&& absorb) {
    Out r;
    bool stop = false;
    if (lockstep_lag >= 0) {
      while (!stop && popped_ < collected - lockstep_lag && wait_pop(r)) stop = absorb(r);
    } else {
      while (!stop && try_pop(r)) stop = absorb(r);
      while (!stop && outstanding() > max_lag && wait_pop(r)) stop = absorb(r);
    }
    return stop;
  }
  // blocks until the next result is there (false: nothing outstanding)
  bool wait_pop(Out& out) {
    if (!threaded_) return try_pop(out);
    std::unique_lock<std::mutex> g(mu_);
    if (outstanding_ == 0) return false;
    cv_done_.wait(g, [this] { return !done_.empty(); });
    out = std::move(done_.front());
    done_.pop_front();
    ++popped_;
    --outstanding_;
    return true;
  }

 private:
  struct Job {
    int64_t m;
    std::vector<double> alpha, beta;
  };
  void run() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_job_.wait(g, [this] { return quit_ || !jobs_.empty(); });
        if (quit_) return;  // verdicts nobody will read any more (the loop has ended) are not computed
        j = std::move(jobs_.front());
        jobs_.pop_front();
      }
      Out o = tracker_.step(j.m, j.alpha.data(), j.beta.data());
      if (jitter_us_ > 0) {
        rng_ = rng_ * 6364136223846793005ull + 1442695040888963407ull;
        std::this_thread::sleep_for(std::chrono::microseconds((long)((rng_ >> 33) % (uint64_t)jitter_us_)));
      }
      {
        std::lock_guard<std::mutex> g(mu_);
        done_.push_back(std::move(o));
      }
      cv_done_.notify_all();
    }
  }
  Tracker tracker_;
  bool threaded_;
  std::thread thread_;
  std::mutex mu_;
  std::condition_variable cv_job_, cv_done_;
  std::deque<Job> jobs_;
  std::deque<Out> done_;
  size_t outstanding_ = 0;
  int64_t popped_ = 0;
  bool quit_ = false;
  int jitter_us_ = 0;
  uint64_t rng_ = (uint64_t)::getpid() * 0x9E3779B97F4A7C15ull;
};
typedef StepWorker<RitzTracker> TridiagWorker;

}  // namespace ll
