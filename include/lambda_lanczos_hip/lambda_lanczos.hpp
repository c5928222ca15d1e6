// lambda_lanczos::LambdaLanczos<T> on MI355X — drop-in for include/lambda_lanczos/lambda_lanczos.hpp:109-415 of
// mrcdr/lambda-lanczos: same constructor shape (mv_mul, matrix_size, find_maximum, num_eigs), same public data
// members with the same defaults, same run() overloads and getIterationCounts(), same result order.  The Krylov
// loop itself (lambda_lanczos.hpp:216-322) runs in liblanczos_hip.so with device-resident Lanczos vectors.
//
//   #include <lambda_lanczos_hip/lambda_lanczos.hpp>
//   using lambda_lanczos::LambdaLanczos;                       // alias of lambda_lanczos_hip (see bottom of file)
//   LambdaLanczos<double> engine(mv_mul, n, true, 1);          // unmodified user lambda  -> host callback path
//   LambdaLanczos<double> engine(csr_matrix, n, true, 1);      // lambda_lanczos::CsrMatrix<double> -> all on device
//   engine.run(eigenvalues, eigenvectors);
//
// Supported T: float, double, std::complex<float>, std::complex<double> (long double has no device counterpart).
#ifndef LAMBDA_LANCZOS_HIP_LAMBDA_LANCZOS_HPP_
#define LAMBDA_LANCZOS_HIP_LAMBDA_LANCZOS_HPP_

#include <limits>
#include <tuple>

#include "common.hpp"

namespace lambda_lanczos_hip {

template <typename T> class LambdaLanczos {
  static_assert(is_supported<T>::value, "LambdaLanczos<T>: T must be float, double or std::complex of those");
  template <typename n_type> using real_t = util::real_t<n_type>;

 public:
  // ---- the reference's public data members (lambda_lanczos.hpp:126-181), same names / types / defaults
  std::function<void(const std::vector<T>& in, std::vector<T>& out)> mv_mul;            // :126
  std::function<void(std::vector<T>& vec)> init_vector = VectorRandomInitializer<T>::init;  // :133 (callable default, :70-104)
  size_t matrix_size;                                                                     // :136
  size_t max_iteration;                                                                   // :138
  real_t<T> eps = std::numeric_limits<real_t<T>>::epsilon() * 1e3;                        // :150
  bool find_maximum;                                                                      // :153
  size_t num_eigs = 1;                                                                    // :156
  real_t<T> eigenvalue_offset = 0.0;                                                      // :165
  size_t num_eigs_per_iteration = 5;                                                      // :173
  size_t initial_vector_size = 200;                                                       // :181
  // ---- additions (the defaults reproduce the reference's decisions and values)
  int tridiag_mode = LL_TRIDIAG_AUTO;  // how the per-iteration Ritz values are obtained (LL_TRIDIAG_*)
  int orth_mode = LL_ORTH_CGS_DGKS;   // Gram-Schmidt variant (LL_ORTH_*)
  const T* init_vector_device = nullptr;  // non-null: start vector (local rows) already in device memory, used
                                          // instead of init_vector (copied, never modified)

  // Reference constructor (lambda_lanczos.hpp:200-208): unmodified user code, host callback operator.
  LambdaLanczos(std::function<void(const std::vector<T>&, std::vector<T>&)> mv_mul, size_t matrix_size,
                bool find_maximum, size_t num_eigs, Context ctx = Context::default_context())
      : mv_mul(mv_mul), matrix_size(matrix_size), max_iteration(matrix_size), find_maximum(find_maximum),
        num_eigs(num_eigs), ctx_(ctx) {}

  // Same shape with a device-resident operator: nothing n-sized crosses PCIe during the loop.
  LambdaLanczos(const DeviceOperator<T>& op, size_t matrix_size, bool find_maximum, size_t num_eigs)
      : matrix_size(matrix_size), max_iteration(matrix_size), find_maximum(find_maximum), num_eigs(num_eigs),
        ctx_(op.context()), csr_(new DeviceOperator<T>(op)) {}

  // run(eigenvalues, eigenvectors) (lambda_lanczos.hpp:330-366): outputs are resized by the library.
  void run(std::vector<real_t<T>>& eigenvalues, std::vector<std::vector<T>>& eigenvectors) {
    const size_t n_local = csr_ ? (size_t)csr_->local_rows() : matrix_size;
    ll_lanczos_params p = make_params(num_eigs);
    detail::InitHook<T> hook{init_vector};
    if (init_vector) {
      p.init_vector = &detail::InitHook<T>::call;
      p.init_user = &hook;
    }
    detail::HostOp<T> host{mv_mul, {}, {}};
    ll_operator* op = csr_ ? csr_->get() : detail::make_host_operator<T>(ctx_.get(), (int64_t)matrix_size, &host);
    std::vector<double> vals(num_eigs);
    std::vector<T> vecs(num_eigs * n_local);
    std::vector<int64_t> counts(4 * num_eigs + 64);
    int64_t found = 0;
    ll_run_stats st;
    const int rc = call_run(op, &p, vals.data(), vecs.data(), &found, counts.data(), (int64_t)counts.size(), &st);
    if (!csr_) ll_op_destroy(op);
    check(rc);
    eigenvalues.assign(vals.begin(), vals.begin() + found);
    eigenvectors.assign((size_t)found, std::vector<T>());
    for (int64_t i = 0; i < found; ++i)
      eigenvectors[(size_t)i].assign(vecs.begin() + (size_t)i * n_local, vecs.begin() + (size_t)(i + 1) * n_local);
    iter_counts_.assign(counts.begin(), counts.begin() + std::min<int64_t>(st.n_passes, (int64_t)counts.size()));
    last_stats_ = st;
  }

  // Addition: run() with the eigenvectors left in DEVICE memory (d_eigenvectors: capacity num_eigs * local rows,
  // eigenvector i at d_eigenvectors + i * local rows).  With init_vector_device set, nothing n-sized crosses PCIe.
  // Returns the number of pairs found.
  size_t run_device(std::vector<real_t<T>>& eigenvalues, T* d_eigenvectors) {
    if (!csr_) throw Error(LL_ERR_INVALID, "run_device needs a device operator");
    ll_lanczos_params p = make_params(num_eigs);
    detail::InitHook<T> hook{init_vector};
    if (init_vector) {
      p.init_vector = &detail::InitHook<T>::call;
      p.init_user = &hook;
    }
    std::vector<double> vals(num_eigs);
    std::vector<int64_t> counts(4 * num_eigs + 64);
    int64_t found = 0;
    ll_run_stats st;
    check(call_run(csr_->get(), &p, vals.data(), d_eigenvectors, &found, counts.data(), (int64_t)counts.size(), &st));
    eigenvalues.assign(vals.begin(), vals.begin() + found);
    iter_counts_.assign(counts.begin(), counts.begin() + std::min<int64_t>(st.n_passes, (int64_t)counts.size()));
    last_stats_ = st;
    return (size_t)found;
  }

  // run_iteration(eigvalues, eigvecs, nroot, orthogonalizeTo) (lambda_lanczos.hpp:216-322): ONE Lanczos pass that
  // tracks nroot Ritz pairs; every Lanczos vector is orthogonalised against the vectors of orthogonalizeTo (any
  // container of std::vector<T> with cbegin()/cend(), like the reference's Iterable).  Returns the iteration count.
  template <typename Iterable>
  size_t run_iteration(std::vector<real_t<T>>& eigvalues, std::vector<std::vector<T>>& eigvecs, size_t nroot,
                       Iterable orthogonalizeTo) const {
    const size_t n_local = csr_ ? (size_t)csr_->local_rows() : matrix_size;
    ll_lanczos_params p = make_params(1);
    detail::InitHook<T> hook{init_vector};
    if (init_vector) {
      p.init_vector = &detail::InitHook<T>::call;
      p.init_user = &hook;
    }
    std::vector<T> lock;
    int64_t n_orth = 0;
    for (auto it = orthogonalizeTo.cbegin(); it != orthogonalizeTo.cend(); ++it, ++n_orth) {
      const std::vector<T>& v = *it;
      if (v.size() != n_local) throw Error(LL_ERR_INVALID, "run_iteration: orthogonalizeTo vector of the wrong size");
      lock.insert(lock.end(), v.begin(), v.end());
    }
    detail::HostOp<T> host{mv_mul, {}, {}};
    ll_operator* op = csr_ ? csr_->get() : detail::make_host_operator<T>(ctx_.get(), (int64_t)matrix_size, &host);
    std::vector<double> vals(nroot);
    std::vector<T> vecs(nroot * n_local);
    int64_t found = 0, itern = 0;
    ll_run_stats st;
    const int rc = abi<T>::run_iteration(ctx_.get(), op, &p, (int64_t)nroot, n_orth, lock.empty() ? nullptr : lock.data(),
                                         vals.data(), vecs.data(), &found, &itern, &st);
    if (!csr_) ll_op_destroy(op);
    check(rc);
    eigvalues.assign(vals.begin(), vals.begin() + found);
    eigvecs.assign((size_t)found, std::vector<T>());
    for (int64_t i = 0; i < found; ++i)
      eigvecs[(size_t)i].assign(vecs.begin() + (size_t)i * n_local, vecs.begin() + (size_t)(i + 1) * n_local);
    last_stats_ = st;
    return (size_t)itern;
  }

  // C++17 multiple-value-return overload (lambda_lanczos.hpp:376-386)
  std::tuple<std::vector<real_t<T>>, std::vector<std::vector<T>>> run() {
    std::vector<real_t<T>> eigenvalues;
    std::vector<std::vector<T>> eigenvectors;
    this->run(eigenvalues, eigenvectors);
    return std::make_tuple(eigenvalues, eigenvectors);
  }

  // One eigenpair regardless of num_eigs; num_eigs is restored (lambda_lanczos.hpp:394-407)
  void run(real_t<T>& eigenvalue, std::vector<T>& eigenvector) {
    const size_t num_eigs_tmp = this->num_eigs;
    this->num_eigs = 1;
    std::vector<real_t<T>> eigenvalues(1);
    std::vector<std::vector<T>> eigenvectors(1);
    try {
      this->run(eigenvalues, eigenvectors);
    } catch (...) {
      this->num_eigs = num_eigs_tmp;
      throw;
    }
    this->num_eigs = num_eigs_tmp;
    eigenvalue = eigenvalues[0];
    eigenvector = std::move(eigenvectors[0]);
  }

  // Latest iteration counts, one entry per restart pass (lambda_lanczos.hpp:412-414)
  const std::vector<size_t>& getIterationCounts() const { return iter_counts_; }
  const ll_run_stats& getLastStats() const { return last_stats_; }

 private:
  ll_lanczos_params make_params(size_t k) const {
    ll_lanczos_params p;
    check(ll_lanczos_params_default(&p, (int64_t)matrix_size, find_maximum ? 1 : 0, (int64_t)k));
    p.max_iteration = (int64_t)max_iteration;
    p.eps = (double)eps;
    p.eigenvalue_offset = (double)eigenvalue_offset;
    p.num_eigs_per_iteration = (int64_t)num_eigs_per_iteration;
    p.initial_vector_size = (int64_t)initial_vector_size;
    p.tridiag_mode = tridiag_mode;
    p.orth_mode = orth_mode;
    p.init_vector_dev = init_vector_device;
    return p;
  }
  int call_run(ll_operator* op, const ll_lanczos_params* p, double* vals, T* vecs, int64_t* found, int64_t* counts,
               int64_t cap, ll_run_stats* st) {
    return abi<T>::run(ctx_.get(), op, p, vals, vecs, found, counts, cap, st);
  }
  Context ctx_;
  std::shared_ptr<DeviceOperator<T>> csr_;
  std::vector<size_t> iter_counts_;
  mutable ll_run_stats last_stats_{};  // run_iteration is const like the reference's (lambda_lanczos.hpp:216-220)
};

// lambda_lanczos::tridiagonal_impl::tridiagonal_eigenpairs / tridiagonal_eigenvalues
// (lambda_lanczos_tridiagonal_impl.hpp:290-361; called directly by the reference's own tests, test/lambda_lanczos_test.cpp:765,795):
// all eigenvalues (ascending) and, on request, the eigenvectors (eigenvectors[k] = k-th vector) of the symmetric tridiagonal
// matrix with diagonal alpha and sub-diagonal beta, by the library's host solver (ll_tridiag_eig: the reference's implicit-QR
// arithmetic step for step, in DOUBLE: bit-identical results for T = double; for T = float the solver still runs in double and the
// results are rounded to float at the end, so they are at least as accurate as — but need not equal — what the reference's QR in
// float arithmetic returns, and the count of forced breaks may differ); returns that count.  Host code, no device needed.
namespace tridiagonal_impl {
template <typename T>
inline size_t tridiagonal_eigenpairs(const std::vector<T>& alpha, const std::vector<T>& beta, std::vector<T>& eigenvalues,
                                     std::vector<std::vector<T>>& eigenvectors, bool compute_eigenvector = true) {
  static_assert(std::is_floating_point<T>::value, "tridiagonal_eigenpairs<T>: T must be a real floating-point type");
  const size_t m = alpha.size();
  std::vector<double> a(alpha.begin(), alpha.end()), b(m ? m : 1, 0.0), ev(m), q;
  for (size_t i = 0; i + 1 < m && i < beta.size(); ++i) b[i] = (double)beta[i];
  if (compute_eigenvector) q.resize(m * m);
  int64_t unconverged = 0;
  check(ll_tridiag_eig((int64_t)m, a.data(), b.data(), ev.data(), compute_eigenvector ? q.data() : nullptr, &unconverged));
  eigenvalues.assign(ev.begin(), ev.end());
  if (compute_eigenvector) {
    eigenvectors.assign(m, std::vector<T>(m));
    for (size_t j = 0; j < m; ++j)
      for (size_t i = 0; i < m; ++i) eigenvectors[j][i] = (T)q[j * m + i];
  }
  return (size_t)unconverged;
}
template <typename T>
inline size_t tridiagonal_eigenvalues(const std::vector<T>& alpha, const std::vector<T>& beta, std::vector<T>& eigenvalues) {
  std::vector<std::vector<T>> none;
  return tridiagonal_eigenpairs(alpha, beta, eigenvalues, none, false);
}
}  // namespace tridiagonal_impl

}  // namespace lambda_lanczos_hip

#ifndef LAMBDA_LANCZOS_HIP_NO_ALIAS
namespace lambda_lanczos = lambda_lanczos_hip;  // drop-in: existing `lambda_lanczos::LambdaLanczos<T>` code compiles unchanged
#endif

#endif  // LAMBDA_LANCZOS_HIP_LAMBDA_LANCZOS_HPP_
