// C++ facade over the C ABI (include/lanczos_hip.h): RAII handles, error translation and the tag dispatch from the
// template parameter T to the _d / _z entry points ("templates cannot cross a C ABI").
#ifndef LAMBDA_LANCZOS_HIP_COMMON_HPP_
#define LAMBDA_LANCZOS_HIP_COMMON_HPP_

#include <algorithm>
#include <complex>
#include <functional>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../lanczos_hip.h"

namespace lambda_lanczos_hip {

namespace util {
// real_t<T>: T for real types, R for std::complex<R> (reference: util/common.hpp:80-102)
template <typename T> struct realTypeMap { typedef T type; };
template <typename T> struct realTypeMap<std::complex<T>> { typedef T type; };
template <typename T> using real_t = typename realTypeMap<T>::type;
}  // namespace util

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
    static int create_host(ll_context* c, int64_t n, int (*fn)(const void*, void*, int64_t, void*), void* user,          \
                           ll_operator** o) {                                                                            \
      return ll_op_create_host_##SFX(c, n, reinterpret_cast<HOSTFN>(fn), user, o);                                       \
    }                                                                                                                    \
    static int run(ll_context* c, ll_operator* op, const ll_lanczos_params* p, double* vals, T* vecs, int64_t* found,    \
                   int64_t* counts, int64_t cap, ll_run_stats* st) {                                                     \
      return ll_lanczos_run_##SFX(c, op, p, vals, vecs, found, counts, cap, nullptr, nullptr, st);                       \
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
    check(ll_ctx_create(device, &c));
    h_.reset(c, [](ll_context* p) { ll_ctx_destroy(p); });
  }
  Context(int device, void* hip_stream) {
    ll_context* c = nullptr;
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
  static Context& default_context() {
    static Context ctx(0);
    return ctx;
  }

 private:
  std::shared_ptr<ll_context> h_;
};

// The device form of the mv_mul plugin: a CSR matrix resident in HBM (rows [row_begin, row_begin + n_rows) of an
// n_cols x n_cols symmetric/Hermitian operator; the whole matrix on a single GPU).  Accepted by the engines in place
// of the host std::function; then only scalars cross PCIe per iteration.
template <typename T> class CsrMatrix {
  static_assert(is_supported<T>::value, "CsrMatrix<T>: T must be float, double or std::complex of those");

 public:
  CsrMatrix(const std::vector<int64_t>& row_ptr, const std::vector<int32_t>& col, const std::vector<T>& val,
            Context ctx = Context::default_context(), int64_t n_cols = -1, int64_t row_begin = 0)
      : ctx_(ctx) {
    const int64_t n_rows = (int64_t)row_ptr.size() - 1;
    if (n_cols < 0) n_cols = n_rows;
    ll_operator* op = nullptr;
    check(abi<T>::create_csr(ctx_.get(), n_rows, n_cols, row_begin, row_ptr.data(), col.data(), val.data(), &op));
    h_.reset(op, [](ll_operator* p) { ll_op_destroy(p); });
    n_ = n_cols;
    n_local_ = n_rows;
  }
  ll_operator* get() const { return h_.get(); }
  const Context& context() const { return ctx_; }
  int64_t size() const { return n_; }
  int64_t local_rows() const { return n_local_; }

 private:
  Context ctx_;
  std::shared_ptr<ll_operator> h_;
  int64_t n_ = 0, n_local_ = 0;
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
