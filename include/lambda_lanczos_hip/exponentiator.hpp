// lambda_lanczos::Exponentiator<T> on MI355X — drop-in for include/lambda_lanczos/exponentiator.hpp:24-211:
// output = exp(a*A) input by Krylov projection, same constructor (mv_mul, matrix_size), same public fields and
// defaults, run() / taylor_run() with the same return values.
#ifndef LAMBDA_LANCZOS_HIP_EXPONENTIATOR_HPP_
#define LAMBDA_LANCZOS_HIP_EXPONENTIATOR_HPP_

#include <limits>

#include "common.hpp"

namespace lambda_lanczos_hip {

template <typename T> class Exponentiator {
  static_assert(is_supported<T>::value, "Exponentiator<T>: T must be float, double or std::complex of those");
  template <typename n_type> using real_t = util::real_t<n_type>;

 public:
  std::function<void(const std::vector<T>& in, std::vector<T>& out)> mv_mul;   // exponentiator.hpp:41
  size_t matrix_size;                                                          // :44
  size_t max_iteration;                                                        // :46
  real_t<T> eps = std::numeric_limits<real_t<T>>::epsilon() * 1e2;             // :58
  bool full_orthogonalize = false;                                             // :63
  size_t initial_vector_size = 200;                                            // :71
  int orth_mode = LL_ORTH_CGS_DGKS;                                            // addition

  Exponentiator(std::function<void(const std::vector<T>&, std::vector<T>&)> mv_mul, size_t matrix_size,
                Context ctx = Context::default_context())
      : mv_mul(mv_mul), matrix_size(matrix_size), max_iteration(matrix_size), ctx_(ctx) {}            // :80-81
  Exponentiator(const DeviceOperator<T>& op, size_t matrix_size)
      : matrix_size(matrix_size), max_iteration(matrix_size), ctx_(op.context()), csr_(new DeviceOperator<T>(op)) {}

  // exp(a*A) input -> output (resized by the library, exponentiator_test.cpp:131); returns the iteration count (:87-173)
  size_t run(const T& a, const std::vector<T>& input, std::vector<T>& output) const { return call(a, input, output, false); }
  // plain Taylor series (:175-210); returns the number of terms
  size_t taylor_run(const T& a, const std::vector<T>& input, std::vector<T>& output) const {
    return call(a, input, output, true);
  }
  // Addition: input and output in DEVICE memory (local rows each; may be the same buffer): a time-evolution loop
  // psi <- exp(a*A) psi never crosses PCIe.  Returns the iteration count.
  size_t run_device(const T& a, const T* d_input, T* d_output) const {
    if (!csr_) throw Error(LL_ERR_INVALID, "run_device needs a device operator");
    ll_expo_params p = make_params();
    int64_t count = 0;
    check(dispatch(csr_->get(), &p, a, d_input, d_output, &count, false));
    return (size_t)count;
  }

 private:
  ll_expo_params make_params() const {
    ll_expo_params p;
    check(ll_expo_params_default(&p, (int64_t)matrix_size));
    p.max_iteration = (int64_t)max_iteration;
    p.eps = (double)eps;
    p.full_orthogonalize = full_orthogonalize ? 1 : 0;
    p.orth_mode = orth_mode;
    p.initial_vector_size = (int64_t)initial_vector_size;
    return p;
  }
  size_t call(const T& a, const std::vector<T>& input, std::vector<T>& output, bool taylor) const {
    const size_t n_local = csr_ ? (size_t)csr_->local_rows() : matrix_size;
    if (input.size() != n_local) throw Error(LL_ERR_INVALID, "input size differs from matrix_size (exponentiator.hpp:88)");
    ll_expo_params p = make_params();
    detail::HostOp<T> host{mv_mul, {}, {}};
    ll_operator* op = csr_ ? csr_->get() : detail::make_host_operator<T>(ctx_.get(), (int64_t)matrix_size, &host);
    output.assign(n_local, T());
    int64_t count = 0;
    const int rc = dispatch(op, &p, a, input.data(), output.data(), &count, taylor);
    if (!csr_) ll_op_destroy(op);
    check(rc);
    return (size_t)count;
  }
  int dispatch(ll_operator* op, const ll_expo_params* p, const double& a, const double* in, double* out, int64_t* count,
               bool taylor) const {
    return taylor ? ll_expo_taylor_run_d(ctx_.get(), op, p, a, in, out, count)
                  : ll_expo_run_d(ctx_.get(), op, p, a, in, out, count, nullptr);
  }
  int dispatch(ll_operator* op, const ll_expo_params* p, const std::complex<double>& a, const std::complex<double>* in,
               std::complex<double>* out, int64_t* count, bool taylor) const {
    return taylor ? ll_expo_taylor_run_z(ctx_.get(), op, p, a.real(), a.imag(), in, out, count)
                  : ll_expo_run_z(ctx_.get(), op, p, a.real(), a.imag(), in, out, count, nullptr);
  }
  int dispatch(ll_operator* op, const ll_expo_params* p, const float& a, const float* in, float* out, int64_t* count,
               bool taylor) const {
    return taylor ? ll_expo_taylor_run_s(ctx_.get(), op, p, (double)a, in, out, count)
                  : ll_expo_run_s(ctx_.get(), op, p, (double)a, in, out, count, nullptr);
  }
  int dispatch(ll_operator* op, const ll_expo_params* p, const std::complex<float>& a, const std::complex<float>* in,
               std::complex<float>* out, int64_t* count, bool taylor) const {
    return taylor ? ll_expo_taylor_run_c(ctx_.get(), op, p, (double)a.real(), (double)a.imag(), in, out, count)
                  : ll_expo_run_c(ctx_.get(), op, p, (double)a.real(), (double)a.imag(), in, out, count, nullptr);
  }
  Context ctx_;
  std::shared_ptr<DeviceOperator<T>> csr_;
};

}  // namespace lambda_lanczos_hip

#ifndef LAMBDA_LANCZOS_HIP_NO_ALIAS
namespace lambda_lanczos = lambda_lanczos_hip;
#endif

#endif  // LAMBDA_LANCZOS_HIP_EXPONENTIATOR_HPP_
