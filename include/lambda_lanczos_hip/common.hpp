// C++ facade over the C ABI (include/lanczos_hip.h): RAII handles, error translation and the tag dispatch from the
// template parameter T to the _d / _z entry points ("templates cannot cross a C ABI").
#ifndef LAMBDA_LANCZOS_HIP_COMMON_HPP_
#define LAMBDA_LANCZOS_HIP_COMMON_HPP_

#include <algorithm>
#include <complex>
#include <functional>
#include <cstdint>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../lanczos_hip.h"
#include "util.hpp"  // lambda_lanczos::util::{real_t, inner_prod, norm, normalize, schmidt_orth, m_norm, sort_eigenpairs, ...}

namespace lambda_lanczos_hip {

// The reference never throws (asserts only); device / RCCL failures need a channel, so the facade throws.
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error("lanczos_hip error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int status) {
  if (status != LL_OK) throw Error(status, ll_last_error());
}

template <typename T> struct is_supported : std::false_type {};
template <> struct is_supported<double> : std::true_type {};
template <> struct is_supported<std::complex<double>> : std::true_type {};
template <> struct is_supported<float> : std::true_type {};
template <> struct is_supported<std::complex<float>> : std::true_type {};

// Tag dispatch from T to the _d / _z / _s / _c entry points of the C ABI.
template <typename T> struct abi;
#define LL_FACADE_ABI(T, SFX, HOSTFN)                                                                                    \
  template <> struct abi<T> {                                                                                            \
    static int create_csr(ll_context* c, int64_t nr, int64_t nc, int64_t rb, const int64_t* rp, const int32_t* ci,       \
                          const T* va, ll_operator** o) {                                                                \
      return ll_op_create_csr_##SFX(c, nr, nc, rb, rp, ci, va, o);                                                       \
    }                                                                                                                    \
    static int create_csr_opt(ll_context* c, int64_t nr, int64_t nc, int64_t rb, const int64_t* rp, const int32_t* ci,   \
                              const T* va, const ll_csr_options* opt, ll_operator** o) {                                 \
      return ll_op_create_csr_opt_##SFX(c, nr, nc, rb, rp, ci, va, opt, o);                                              \
    }                                                                                                                    \
    static int create_dense(ll_context* c, int64_t nr, int64_t nc, int64_t rb, const T* a, ll_operator** o) {            \
      return ll_op_create_dense_##SFX(c, nr, nc, rb, a, o);                                                              \
    }                                                                                                                    \
    static int create_stencil(ll_context* c, const ll_stencil_desc* d, int64_t rb, int64_t nl, const double* onsite,     \
                              ll_operator** o) {                                                                         \
      return ll_op_create_stencil_##SFX(c, d, rb, nl, onsite, o);                                                        \
    }                                                                                                                    \
    static int create_host(ll_context* c, int64_t n, int (*fn)(const void*, void*, int64_t, void*), void* user,          \
                           ll_operator** o) {                                                                            \
      return ll_op_create_host_##SFX(c, n, reinterpret_cast<HOSTFN>(fn), user, o);                                       \
    }                                                                                                                    \
    static int run(ll_context* c, ll_operator* op, const ll_lanczos_params* p, double* vals, T* vecs, int64_t* found,    \
                   int64_t* counts, int64_t cap, ll_run_stats* st) {                                                     \
      return ll_lanczos_run_##SFX(c, op, p, vals, vecs, found, counts, cap, nullptr, nullptr, st);                       \
    }                                                                                                                    \
    static int run_iteration(ll_context* c, ll_operator* op, const ll_lanczos_params* p, int64_t nroot, int64_t n_orth,  \
                             const T* orth, double* vals, T* vecs, int64_t* found, int64_t* itern, ll_run_stats* st) {   \
      return ll_lanczos_run_iteration_##SFX(c, op, p, nroot, n_orth, orth, vals, vecs, found, itern, nullptr, nullptr,   \
                                            st);                                                                         \
    }                                                                                                                    \
  };
LL_FACADE_ABI(double, d, ll_host_mv_mul_d)
LL_FACADE_ABI(float, s, ll_host_mv_mul_s)
LL_FACADE_ABI(std::complex<double>, z, ll_host_mv_mul_z)
LL_FACADE_ABI(std::complex<float>, c, ll_host_mv_mul_z)
#undef LL_FACADE_ABI

// Device + stream + workspace.  Copyable handle (shared ownership).
class Context {
 public:
  explicit Context(int device = 0) {
    ll_context* c = nullptr;
    check(LL_ABI_CHECK());  // a binary built against an older lanczos_hip.h must not hand its structs to this library
    check(ll_ctx_create(device, &c));
    h_.reset(c, [](ll_context* p) { ll_ctx_destroy(p); });
  }
  Context(int device, void* hip_stream) {
    ll_context* c = nullptr;
    check(LL_ABI_CHECK());
    check(ll_ctx_create_on_stream(device, hip_stream, &c));
    h_.reset(c, [](ll_context* p) { ll_ctx_destroy(p); });
  }
  ll_context* get() const { return h_.get(); }
  // One process per GPU: attach an RCCL communicator (id from Context::unique_id() on rank 0, distributed by the host).
  static std::vector<char> unique_id() {
    std::vector<char> id(LL_UNIQUE_ID_BYTES);
    check(ll_comm_unique_id(id.data()));
    return id;
  }
  void init_comm(const std::vector<char>& id, int rank, int n_ranks) { check(ll_comm_init(get(), id.data(), rank, n_ranks)); }
  // Which transport answers the collectives: "rccl", "plugin:<path>", "attached" or "none" (ll_comm_transport).
  std::string transport() const {
    char buf[512];
    check(ll_comm_transport(get(), buf, sizeof(buf)));
    return std::string(buf);
  }
  // UNSTABLE: one tuning field of this context by key, on top of the environment switches (ll_ctx_set_tuning; value nullptr removes
  // the setting).  The environment carries only the user-facing switches of INTEGRATION.md section 8.
  void set_tuning(const char* key, const char* value) { check(ll_ctx_set_tuning(get(), key, value)); }
  static Context& default_context() {
    static Context ctx(0);
    return ctx;
  }

 private:
  std::shared_ptr<ll_context> h_;
};

// The device forms of the mv_mul plugin.  A DeviceOperator<T> is a shared handle to an operator resident in HBM; the
// engines accept it in place of the host std::function, and then only scalars cross PCIe per iteration.
template <typename T> class DeviceOperator {
  static_assert(is_supported<T>::value, "DeviceOperator<T>: T must be float, double or std::complex of those");

 public:
  ll_operator* get() const { return h_.get(); }
  const Context& context() const { return ctx_; }
  int64_t size() const { return n_; }
  int64_t local_rows() const { return n_local_; }
  // max_i sum_j |a_ij| over the local rows: a safe |eigenvalue_offset| (determine_eigenvalue_offset.cpp:12-29)
  double inf_norm() const {
    double v = 0;
    check(ll_op_inf_norm(get(), &v));
    return v;
  }

 protected:
  DeviceOperator(Context ctx) : ctx_(ctx) {}
  void adopt(ll_operator* op, int64_t n, int64_t n_local) {
    h_.reset(op, [](ll_operator* p) { ll_op_destroy(p); });
    n_ = n;
    n_local_ = n_local;
  }
  Context ctx_;
  std::shared_ptr<ll_operator> h_;
  int64_t n_ = 0, n_local_ = 0;
};

// CSR matrix: rows [row_begin, row_begin + n_rows) of an n_cols x n_cols symmetric/Hermitian operator (the whole
// matrix on a single GPU).
//
// Accuracy of y = A x — what stands in for the user's fp64 mv_mul loop (lambda_lanczos.hpp:120-126).  The default class
// (Accuracy::Default -> norm-wise where the propagation-blocked kernel is selected) bounds the error of every row by
// eps * sum_j |a_ij||x_j| + nnz_i 2^-60 ||A||_inf max|x|: right for Lanczos vectors of extended states and what the Krylov
// recurrence needs, but not component-wise for strongly LOCALISED vectors.  A caller who knows their vectors are localised
// passes Accuracy::Componentwise (floating-point sums in a fixed order: the accuracy of a plain fp64 row loop, 3-5 % slower
// on such matrices) — per operator, no environment variable; set_accuracy() moves an existing operator between the classes.
enum class Accuracy : int {
  Default = LL_ACCURACY_DEFAULT,
  Normwise = LL_ACCURACY_NORMWISE,
  Componentwise = LL_ACCURACY_COMPONENTWISE
};
template <typename T> class CsrMatrix : public DeviceOperator<T> {
 public:
  CsrMatrix(const std::vector<int64_t>& row_ptr, const std::vector<int32_t>& col, const std::vector<T>& val,
            Context ctx = Context::default_context(), int64_t n_cols = -1, int64_t row_begin = 0,
            Accuracy accuracy = Accuracy::Default)
      : DeviceOperator<T>(ctx) {
    const int64_t n_rows = (int64_t)row_ptr.size() - 1;
    if (n_cols < 0) n_cols = n_rows;
    ll_operator* op = nullptr;
    ll_csr_options opt;
    check(ll_csr_options_default(&opt));
    opt.accuracy = (int32_t)accuracy;
    check(abi<T>::create_csr_opt(ctx.get(), n_rows, n_cols, row_begin, row_ptr.data(), col.data(), val.data(), &opt, &op));
    this->adopt(op, n_cols, n_rows);
  }
  CsrMatrix(const std::vector<int64_t>& row_ptr, const std::vector<int32_t>& col, const std::vector<T>& val, Accuracy accuracy)
      : CsrMatrix(row_ptr, col, val, Context::default_context(), -1, 0, accuracy) {}
  void set_accuracy(Accuracy a) { check(ll_op_set_accuracy(this->get(), (int)a)); }
  Accuracy accuracy() const {
    int a = 0;
    check(ll_op_accuracy(this->get(), &a));
    return (Accuracy)a;
  }
};

// Dense row-major matrix (the operator of src/samples/sample1_simple.cpp:22-28 without the host loop).
template <typename T> class DenseMatrix : public DeviceOperator<T> {
 public:
  explicit DenseMatrix(const std::vector<std::vector<T>>& rows, Context ctx = Context::default_context(),
                       int64_t row_begin = 0)
      : DeviceOperator<T>(ctx) {
    const int64_t nr = (int64_t)rows.size(), nc = nr ? (int64_t)rows[0].size() : 0;
    std::vector<T> flat;
    flat.reserve((size_t)(nr * nc));
    for (const auto& r : rows) {
      if ((int64_t)r.size() != nc) throw Error(LL_ERR_INVALID, "DenseMatrix: ragged rows");
      flat.insert(flat.end(), r.begin(), r.end());
    }
    ll_operator* op = nullptr;
    check(abi<T>::create_dense(ctx.get(), nr, nc, row_begin, flat.data(), &op));
    this->adopt(op, nc, nr);
  }
};

// Matrix-free lattice operator (the family of src/samples/sample3_dynamic.cpp:17-22):
//   (A x)(r) = (diag + onsite[r]) x(r) + sum_d ( hop[d] x(r + e_d) + conj(hop[d]) x(r - e_d) )
// on a row-major lattice dims[0] x dims[1] x .. (last index fastest), open or periodic per dimension.
template <typename T> class LatticeOperator : public DeviceOperator<T> {
 public:
  // phase_grad (optional, complex T): ndim x ndim row-major; the bond r -> r + e_d carries
  // hop[d] * exp(i * sum_e phase_grad[d*ndim + e] * c_e(r)) — Peierls phases of a magnetic field.
  LatticeOperator(const std::vector<int64_t>& dims, double diag, const std::vector<std::complex<double>>& hop,
                  const std::vector<bool>& periodic, const std::vector<double>& onsite = {},
                  Context ctx = Context::default_context(), int64_t row_begin = 0, int64_t n_local = -1,
                  const std::vector<double>& phase_grad = {})
      : DeviceOperator<T>(ctx) {
    if (dims.empty() || dims.size() > 3 || hop.size() != dims.size() || periodic.size() != dims.size())
      throw Error(LL_ERR_INVALID, "LatticeOperator: 1 to 3 dimensions, one hop and one periodic flag per dimension");
    ll_stencil_desc d = {};
    d.ndim = (int32_t)dims.size();
    int64_t n = 1;
    for (size_t k = 0; k < dims.size(); ++k) {
      d.dims[k] = dims[k];
      d.periodic[k] = periodic[k] ? 1 : 0;
      d.hop_re[k] = hop[k].real();
      d.hop_im[k] = hop[k].imag();
      n *= dims[k];
    }
    d.diag = diag;
    if (!phase_grad.empty()) {
      if (phase_grad.size() != dims.size() * dims.size()) throw Error(LL_ERR_INVALID, "LatticeOperator: phase_grad must be ndim x ndim");
      for (size_t k = 0; k < dims.size(); ++k)
        for (size_t e = 0; e < dims.size(); ++e) d.phase_grad[k][e] = phase_grad[k * dims.size() + e];
    }
    if (n_local < 0) n_local = n;
    if (!onsite.empty() && (int64_t)onsite.size() != n_local) throw Error(LL_ERR_INVALID, "LatticeOperator: onsite size");
    ll_operator* op = nullptr;
    check(abi<T>::create_stencil(ctx.get(), &d, row_begin, n_local, onsite.empty() ? nullptr : onsite.data(), &op));
    this->adopt(op, n, n_local);
  }
};

// VectorRandomInitializer<T> (lambda_lanczos.hpp:70-104; the public init_vector member points at its init, :133): every
// element — real and imaginary part for complex T — uniform in [-1, 1] from a std::random_device-seeded mt19937.
// Callable like the reference's, so user code that invokes engine.init_vector(v) or names the class keeps working.
template <typename T> struct VectorRandomInitializer {
  static void init(std::vector<T>& v) {
    std::random_device dev;
    std::mt19937 mt(dev());
    std::uniform_real_distribution<T> rand((T)(-1.0), (T)(1.0));
    for (auto& e : v) e = rand(mt);
  }
};
template <typename R> struct VectorRandomInitializer<std::complex<R>> {
  static void init(std::vector<std::complex<R>>& v) {
    std::random_device dev;
    std::mt19937 mt(dev());
    std::uniform_real_distribution<R> rand((R)(-1.0), (R)(1.0));
    for (auto& e : v) {
      const R re = rand(mt);  // real part first, like the reference's constructor-argument order in practice
      const R im = rand(mt);
      e = std::complex<R>(re, im);
    }
  }
};

namespace detail {
// Host-callback trampoline: the reference's mv_mul signature works on std::vector, the C ABI on raw pointers.
template <typename T> struct HostOp {
  std::function<void(const std::vector<T>&, std::vector<T>&)> fn;
  std::vector<T> in, out;
  static int call(const void* in_p, void* out_p, int64_t n, void* user) {
    HostOp* self = static_cast<HostOp*>(user);
    try {
      const T* a = static_cast<const T*>(in_p);
      self->in.assign(a, a + n);
      self->out.assign((size_t)n, T());  // zero-filled on entry (lambda_lanczos.hpp:242)
      self->fn(self->in, self->out);
      std::copy(self->out.begin(), self->out.end(), static_cast<T*>(out_p));
      return 0;
    } catch (...) {
      return 1;
    }
  }
};

// The default of the public init_vector member: the reference's VectorRandomInitializer<T> under its own name (above,
// outside detail); RandomInit is the name earlier rounds of this facade used.
template <typename T> using RandomInit = VectorRandomInitializer<T>;

template <typename T> struct InitHook {
  std::function<void(std::vector<T>&)> fn;
  static void call(void* vec, int64_t n_local, int64_t /*row_begin*/, void* user) {
    InitHook* self = static_cast<InitHook*>(user);
    std::vector<T> v((size_t)n_local);
    self->fn(v);
    std::copy(v.begin(), v.end(), static_cast<T*>(vec));
  }
};

template <typename T> inline ll_operator* make_host_operator(ll_context* ctx, int64_t n, HostOp<T>* h) {
  ll_operator* op = nullptr;
  check(abi<T>::create_host(ctx, n, &HostOp<T>::call, h, &op));
  return op;
}
}  // namespace detail

}  // namespace lambda_lanczos_hip

#endif  // LAMBDA_LANCZOS_HIP_COMMON_HPP_
