// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// C-ABI shim around the REAL reference headers.  This file contains no reference code: it
// #includes <lambda_lanczos.hpp> / <exponentiator.hpp> from where they lie
// (-I/root/reference/include/lambda_lanczos, see oracle/Makefile) and exports the same entry
// points as oracle/lanczos_oracle.cpp under the prefix ref_.  The result goes to
// oracle/_ref/libref.so (git-ignored, not gpurun-ignored), is used to pin the restatement,
// to generate tests/golden/*.json (tests/golden/make_golden.py) and as bench.py's
// cpu_baseline ("kind": "reference").  It cannot be rebuilt on the GPU box (no /root/reference
// there); the prebuilt .so travels.
//
// alpha/beta are locals of LambdaLanczos::run_iteration (LL:222-223) and Exponentiator::run
// (EX:91-92); they are recovered with an instrumented mv_mul as SURVEY.md section 8(c)
// describes: alpha_k = Re<in_k, out_k> + offset, beta_k = Re<in_{k+1}, out_k + offset in_k>.

#include <chrono>
#include <complex>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#include <exponentiator.hpp>
#include <lambda_lanczos.hpp>

typedef std::complex<double> zd;

extern "C" {
struct oracle_params {
  int64_t matrix_size, max_iteration;
  double eps;
  int32_t find_maximum, full_orthogonalize;
  int64_t num_eigs;
  double eigenvalue_offset;
  int64_t num_eigs_per_iteration;
};
struct oracle_trace { double* alpha; double* beta; int64_t* len; double* t_mv; double* t_total; };
}

namespace {
inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline double re_(double v) { return v; }
inline double re_(zd v) { return v.real(); }
inline double cj_(double v) { return v; }
inline zd cj_(zd v) { return std::conj(v); }

template <typename T> struct Tracer {
  const int64_t* rp; const int32_t* ci; const T* va; int64_t n; double offset; oracle_trace* tr;
  std::vector<T> prev_in, prev_out;
  int64_t calls = 0;
  void operator()(const std::vector<T>& in, std::vector<T>& out) {
    double t0 = now_s();
    for (int64_t i = 0; i < n; ++i) {
      T acc = T();
      for (int64_t p = rp[i]; p < rp[i + 1]; ++p) acc += va[p] * in[ci[p]];
      out[i] += acc;
    }
    if (tr && tr->t_mv) *tr->t_mv += now_s() - t0;
    if (tr && tr->alpha) {
      if (calls > 0) {  // beta_{calls} from the previous out and this in
        T s = T();
        for (int64_t i = 0; i < n; ++i) s += cj_(in[i]) * (prev_out[i] + offset * prev_in[i]);
        tr->beta[calls - 1] = re_(s);
      }
      T a = T();
      for (int64_t i = 0; i < n; ++i) a += cj_(in[i]) * out[i];
      tr->alpha[calls] = re_(a) + offset;
      prev_in = in;
      prev_out = out;
      *tr->len = calls + 1;
    }
    ++calls;
  }
};

template <typename T>
int64_t run_eig(const int64_t* rp, const int32_t* ci, const T* va, const oracle_params* p, const T* init,
                double* eigvals, T* eigvecs, int64_t* iter_counts, int64_t* n_pass, oracle_trace* tr) {
  const int64_t n = p->matrix_size;
  Tracer<T> op{rp, ci, va, n, p->eigenvalue_offset, tr};
  double t0 = now_s();
  lambda_lanczos::LambdaLanczos<T> eng([&op](const std::vector<T>& in, std::vector<T>& out) { op(in, out); },
                                       (size_t)n, p->find_maximum != 0, (size_t)p->num_eigs);
  eng.max_iteration = (size_t)p->max_iteration;
  eng.eps = p->eps;
  eng.eigenvalue_offset = p->eigenvalue_offset;
  eng.num_eigs_per_iteration = (size_t)p->num_eigs_per_iteration;
  eng.init_vector = [init, n, &op](std::vector<T>& v) {
    std::memcpy(v.data(), init, (size_t)n * sizeof(T));
    op.calls = 0;  // a new pass: the trace keeps the last pass only
  };
  std::vector<double> ev;
  std::vector<std::vector<T>> x;
  eng.run(ev, x);
  if (tr && tr->t_total) *tr->t_total += now_s() - t0;
  for (size_t i = 0; i < ev.size(); ++i) {
    eigvals[i] = ev[i];
    std::memcpy(eigvecs + i * (size_t)n, x[i].data(), (size_t)n * sizeof(T));
  }
  const auto& ic = eng.getIterationCounts();
  for (size_t i = 0; i < ic.size(); ++i) iter_counts[i] = (int64_t)ic[i];
  *n_pass = (int64_t)ic.size();
  return (int64_t)ev.size();
}

// The reference's public run_iteration (lambda_lanczos.hpp:216-322) called directly.
template <typename T>
int64_t run_iter(const int64_t* rp, const int32_t* ci, const T* va, const oracle_params* p, const T* init, int64_t nroot,
                 int64_t n_orth, const T* orth, double* eigvals, T* eigvecs, int64_t* n_found) {
  const int64_t n = p->matrix_size;
  Tracer<T> op{rp, ci, va, n, p->eigenvalue_offset, nullptr};
  lambda_lanczos::LambdaLanczos<T> eng([&op](const std::vector<T>& in, std::vector<T>& out) { op(in, out); },
                                       (size_t)n, p->find_maximum != 0, 1);
  eng.max_iteration = (size_t)p->max_iteration;
  eng.eps = p->eps;
  eng.eigenvalue_offset = p->eigenvalue_offset;
  eng.init_vector = [init, n](std::vector<T>& v) { std::memcpy(v.data(), init, (size_t)n * sizeof(T)); };
  std::vector<std::vector<T>> lock;
  for (int64_t j = 0; j < n_orth; ++j) lock.emplace_back(orth + j * n, orth + (j + 1) * n);
  std::vector<double> ev;
  std::vector<std::vector<T>> x;
  const size_t it = eng.run_iteration(ev, x, (size_t)nroot, lock);
  for (size_t i = 0; i < ev.size(); ++i) {
    eigvals[i] = ev[i];
    std::memcpy(eigvecs + i * (size_t)n, x[i].data(), (size_t)n * sizeof(T));
  }
  *n_found = (int64_t)ev.size();
  return (int64_t)it;
}

template <typename T>
int64_t run_expo(const int64_t* rp, const int32_t* ci, const T* va, const oracle_params* p, T a, const T* input,
                 T* output, oracle_trace* tr, bool taylor) {
  const int64_t n = p->matrix_size;
  oracle_trace notrace{nullptr, nullptr, nullptr, tr ? tr->t_mv : nullptr, nullptr};
  Tracer<T> op{rp, ci, va, n, 0.0, &notrace};
  double t0 = now_s();
  lambda_lanczos::Exponentiator<T> ex([&op](const std::vector<T>& in, std::vector<T>& out) { op(in, out); },
                                      (size_t)n);
  ex.max_iteration = (size_t)p->max_iteration;
  ex.eps = p->eps;
  ex.full_orthogonalize = p->full_orthogonalize != 0;
  std::vector<T> in(input, input + n), out;  // output left unsized on purpose (T2:131)
  size_t it = taylor ? ex.taylor_run(a, in, out) : ex.run(a, in, out);
  if (tr && tr->t_total) *tr->t_total += now_s() - t0;
  std::memcpy(output, out.data(), (size_t)n * sizeof(T));
  return (int64_t)it;
}
}  // namespace

extern "C" {

int64_t ref_lanczos_run_d(const int64_t* rp, const int32_t* ci, const double* va, const oracle_params* p,
                          const double* init, double* eigvals, double* eigvecs, int64_t* iter_counts, int64_t* n_pass,
                          oracle_trace* tr) {
  return run_eig<double>(rp, ci, va, p, init, eigvals, eigvecs, iter_counts, n_pass, tr);
}
int64_t ref_lanczos_run_z(const int64_t* rp, const int32_t* ci, const zd* va, const oracle_params* p, const zd* init,
                          double* eigvals, zd* eigvecs, int64_t* iter_counts, int64_t* n_pass, oracle_trace* tr) {
  return run_eig<zd>(rp, ci, va, p, init, eigvals, eigvecs, iter_counts, n_pass, tr);
}
int64_t ref_run_iteration_d(const int64_t* rp, const int32_t* ci, const double* va, const oracle_params* p,
                            const double* init, int64_t nroot, int64_t n_orth, const double* orth, double* eigvals,
                            double* eigvecs, int64_t* n_found) {
  return run_iter<double>(rp, ci, va, p, init, nroot, n_orth, orth, eigvals, eigvecs, n_found);
}
int64_t ref_run_iteration_z(const int64_t* rp, const int32_t* ci, const zd* va, const oracle_params* p, const zd* init,
                            int64_t nroot, int64_t n_orth, const zd* orth, double* eigvals, zd* eigvecs,
                            int64_t* n_found) {
  return run_iter<zd>(rp, ci, va, p, init, nroot, n_orth, orth, eigvals, eigvecs, n_found);
}
int64_t ref_expo_run_d(const int64_t* rp, const int32_t* ci, const double* va, const oracle_params* p, double a,
                       const double* input, double* output, oracle_trace* tr) {
  return run_expo<double>(rp, ci, va, p, a, input, output, tr, false);
}
int64_t ref_expo_run_z(const int64_t* rp, const int32_t* ci, const zd* va, const oracle_params* p, double a_re,
                       double a_im, const zd* input, zd* output, oracle_trace* tr) {
  return run_expo<zd>(rp, ci, va, p, zd(a_re, a_im), input, output, tr, false);
}
int64_t ref_taylor_run_d(const int64_t* rp, const int32_t* ci, const double* va, const oracle_params* p, double a,
                         const double* input, double* output) {
  return run_expo<double>(rp, ci, va, p, a, input, output, nullptr, true);
}
int64_t ref_taylor_run_z(const int64_t* rp, const int32_t* ci, const zd* va, const oracle_params* p, double a_re,
                         double a_im, const zd* input, zd* output) {
  return run_expo<zd>(rp, ci, va, p, zd(a_re, a_im), input, output, nullptr, true);
}

void ref_inner_prod_z(int64_t n, const zd* a, const zd* b, zd* out) {
  std::vector<zd> va(a, a + n), vb(b, b + n);
  *out = lambda_lanczos::util::inner_prod(va, vb);
}
double ref_m_norm_z(int64_t n, const zd* a) {
  std::vector<zd> v(a, a + n);
  return lambda_lanczos::util::m_norm(v);
}
void ref_schmidt_orth_z(int64_t n, int64_t nb, const zd* basis, zd* w) {
  std::vector<std::vector<zd>> us;
  for (int64_t j = 0; j < nb; ++j) us.emplace_back(basis + j * n, basis + (j + 1) * n);
  std::vector<zd> v(w, w + n);
  lambda_lanczos::util::schmidt_orth(v, us.begin(), us.end());
  std::copy(v.begin(), v.end(), w);
}
int64_t ref_tridiag_eig(int64_t n, const double* alpha, const double* beta, int64_t nbeta, double* ev, double* q) {
  std::vector<double> al(alpha, alpha + n), be(beta, beta + nbeta), e;
  std::vector<std::vector<double>> qq;
  size_t unc = lambda_lanczos::tridiagonal_impl::tridiagonal_eigenpairs(al, be, e, qq, q != nullptr);
  std::copy(e.begin(), e.end(), ev);
  if (q) for (int64_t j = 0; j < n; ++j) std::copy(qq[j].begin(), qq[j].end(), q + j * n);
  return (int64_t)unc;
}
double ref_mth_eigenvalue(int64_t n, const double* alpha, const double* beta, int64_t m) {
  std::vector<double> al(alpha, alpha + n), be(beta, beta + n);
  return lambda_lanczos::tridiagonal_impl::find_mth_eigenvalue(al, be, (size_t)m);
}

// The start vector of the reference's own tests: mt19937(seed) + uniform_real_distribution(-1,1)
// (T1:25-45); complex: re then im per element.  libstdc++-specific, hence captured into fixtures.
void ref_init_mt19937_d(uint32_t seed, int64_t n, double* v) {
  std::mt19937 mt(seed);
  std::uniform_real_distribution<double> r(-1.0, 1.0);
  for (int64_t i = 0; i < n; ++i) v[i] = r(mt);
}
void ref_init_mt19937_z(uint32_t seed, int64_t n, zd* v) {
  std::mt19937 mt(seed);
  std::uniform_real_distribution<double> r(-1.0, 1.0);
  for (int64_t i = 0; i < n; ++i) { double a = r(mt); double b = r(mt); v[i] = zd(a, b); }
}

}  // extern "C"
