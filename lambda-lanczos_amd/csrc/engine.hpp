// Host drivers of the device-resident Lanczos loop (templated on the scalar type; instantiated for double and
// complex<double> in engine.cpp).  See DESIGN.md for the data flow.
#pragma once

#include <complex>
#include <functional>
#include <map>

#include "ll_internal.hpp"

namespace ll {

// Device scalar area (ctx->d_scal, 64 doubles):
//   [0..4)    alpha ring (slot = k % 4)
//   [8..20)   norm triples (c0,c1,c2) ring, slot s at 8 + 3*s
//   [24..27)  scratch triple for one-off orthogonalisations (start vector, primitives)
//   [32]      spare scalar (dot results)
//   [63]      constant 0
constexpr int kScalAlpha = 0, kScalNorms = 8, kScalScratch = 24, kScalSpare = 32, kScalZero = 63, kScalCount = 64;

// ---------------------------------------------------------------- chunked device slab for the Krylov basis
// The reference keeps one heap std::vector per Lanczos vector (LL:221,250; reserve(200) LL:181).  Here the basis
// lives in HBM as a few large slabs of `chunk_vecs` vectors with a common leading dimension (multiple of 256
// elements => every vector is 2 KiB aligned); slabs are appended on demand and reused across restart passes,
// nothing is allocated per iteration.
template <typename T> struct Basis {
  ll_context* ctx = nullptr;
  int64_t n_local = 0, ld = 0;
  int64_t chunk_vecs = 0;
  std::vector<T*> chunks;
  ~Basis();
  void init(ll_context* c, int64_t n_local_, int64_t ld_, int64_t chunk_vecs_);
  T* vec(int64_t k);  // pointer to vector k, growing the slab list if needed
};

// A list of (base, count) runs with a common ld, packed into kernel-argument groups.
template <typename T> struct RunList {
  std::vector<std::pair<const T*, int>> runs;
  int64_t ld = 0;
  int total() const {
    int t = 0;
    for (auto& r : runs) t += r.second;
    return t;
  }
  void add(const T* base, int64_t count) {
    if (count > 0) runs.emplace_back(base, (int)count);
  }
  void add_basis(Basis<T>& b, int64_t count) {  // vectors [0, count)
    ld = b.ld;
    for (int64_t k = 0; k < count; k += b.chunk_vecs) add(b.vec(k), std::min(b.chunk_vecs, count - k));
  }
  // launch groups of at most kMaxSegs runs and max_vecs vectors each (runs are split when needed)
  std::vector<BasisSegs<T>> groups(int max_vecs) const;
};

template <typename T> struct Engine {
  ll_context* ctx;
  ll_operator* op;  // may be null for pure BLAS-1 use
  int64_t n_local;
  static constexpr int R = scalar_traits<T>::reals;

  Engine(ll_context* c, ll_operator* o, int64_t n_local_) : ctx(c), op(o), n_local(n_local_) {}

  double* S(int i) const { return ctx->d_scal + i; }
  NormRefs plain_norm(double* c1) const { return NormRefs{S(kScalZero), c1, c1, 0}; }

  // y = A x + offset x ; Re<x,y> -> *d_alpha (device scalar, all-reduced over ranks); d_alpha nullable.
  // x_padded: x_local is readable up to the padded shard length n_shard (true for basis vectors).
  // defer (nullable): the caller's multi-dot will fold alpha itself (ThreeTerm::alpha_partials); apply then only leaves
  // the partials behind and reports them here.  Honoured on unsharded contexts only (a communicator needs the folded
  // scalar for its all-reduce): check defer->nparts > 0 afterwards.
  struct DeferredAlpha {
    const double* partials = nullptr;
    int nparts = 0;
  };
  // sc (nullable): deferred normalisation — x_local is the unnormalised w_k (ScaleIn, ll_internal.hpp); only where
  // can_defer_scale() holds.
  // xnorm2 (nullable device scalar; device operators): x_local is an unnormalised vector w with
  // ||w||^2 = *xnorm2 and the operator works with w / ||w|| (lagged Gram-Schmidt, LoopState).
  void apply(const T* x_local, T* y, double offset, double* d_alpha, bool x_padded = false, DeferredAlpha* defer = nullptr,
             const ScaleIn<T>* sc = nullptr, const double* xnorm2 = nullptr);
  // any device operator can take its input unnormalised (the PB kernels scale the x slice while they stage it, the
  // others scale the finished row sum through ScaleIn); sharded contexts gather / exchange the unnormalised shards
  bool can_scale_input() const {
    if (op == nullptr) return false;
    return op->kind == ll_operator::CSR || op->kind == ll_operator::STENCIL || op->kind == ll_operator::DENSE;
  }
  // The operator kernel can normalise its input on the fly: single GPU, and a kernel that reads x itself (CSR-stream,
  // lattice, dense).  The PB kernels keep the separate normalisation (they want max|u_k| from it), callbacks hand x to
  // user code, sharded contexts gather the normalised vector.
  bool can_defer_scale() const {
    if (ctx->comm != nullptr || op == nullptr) return false;
    if (op->kind == ll_operator::STENCIL || op->kind == ll_operator::DENSE) return true;
    return op->kind == ll_operator::CSR && op->spmv_kind == LL_SPMV_CSR_STREAM;
  }
  // Orthogonalise w against the runs with an optional fused three-term update; c = device triple for the norms.
  // Returns the NormRefs every consumer must use for ||w|| afterwards.  h_total (device, nullable): R*nb doubles.
  // first_pass_only (whole-loop drivers, LL_ORTH_CGS_DGKS): enqueue pass 1 only and return refs whose final norm is
  // c1; the driver evaluates the DGKS test on the host from the published (c0, c1) one iteration later and runs the
  // rare second pass itself (second_pass below) — no predicated no-op launches or collectives per iteration.
  // publish (nullable; single-GPU loops): where the iteration's four scalars go; when the final norm fold of this
  // call can carry them (first_pass_only, no communicator) it does and sets publish->done.
  struct Publish {
    double* host;         // pinned, device-mapped slot of 4 doubles
    const double* alpha;  // device scalar
    bool done;
    // can_defer (set by the caller): the caller scales w right after orth() and can fold + publish in that kernel
    // (launch_scale_publish); orth then launches no fold of its own and describes it here (deferred = true).
    bool can_defer = false;
    bool deferred = false;
    const double* partials = nullptr;
    int nparts = 0;
    double* c1 = nullptr;
    const double* c0 = nullptr;
    // sharded contexts with the derived norm: the caller's normalisation kernel forms ||w'||^2 = *derive_c0 - sum h_i^2
    // itself and publishes (launch_scale_derive); c0_out / c1 receive the two norms.
    bool derive = false;
    const double* derive_c0 = nullptr;
    const double* derive_h = nullptr;
    int derive_count = 0;
    double* c0_out = nullptr;
  };
  NormRefs orth(T* w, const RunList<T>& runs, int mode, const ThreeTerm<T>& tt, double* c, double* h_total,
                bool first_pass_only = false, Publish* publish = nullptr);
  // The deferred second pass on the already normalised vector u = w1/||w1||: orthogonalise against `runs` once more,
  // renormalise, and return ||u'||^2 (the factor by which beta^2 shrinks).  Synchronises the stream.
  double second_pass(T* u, const RunList<T>& runs);
  // ||v||^2 -> *d_out (device, all-reduced)
  void norm2_dev(const T* v, double* d_out);
  // <a,b> -> d_out[0..R) (device, all-reduced)
  void dot_dev(const T* a, const T* b, double* d_out);
  void all_reduce(double* d, size_t count);
  // device-time stamps around an exchange step on stream cs (no-ops unless ctx->profiling)
  void comm_timer_begin(hipStream_t cs);
  void comm_timer_end(hipStream_t cs);
  // read `count` doubles from the device scalar area (synchronises the stream)
  void fetch(const double* d, double* host, size_t count);
  // out_r = sum_k coeff[r*m+k] u_k  (coeff host, type T)
  void gemv(const RunList<T>& basis, int64_t m, int nout, const T* coeff_host, T* out, int64_t ld_out);
};

template <typename T> struct host_scalar;
template <> struct host_scalar<double> { typedef double type; };
template <> struct host_scalar<zc> { typedef std::complex<double> type; };
// float storage: the k-sized host math (tridiagonal step, exp(a T_k) e_1, coefficients) stays in double
template <> struct host_scalar<float> { typedef double type; };
template <> struct host_scalar<cf> { typedef std::complex<double> type; };

// LambdaLanczos<T>::run_iteration (LL:216-322) as a mode of lanczos_run: ONE pass with `nroot` Ritz pairs,
// Gram-Schmidt against n_orth caller-provided vectors (the reference's orthogonalizeTo; host, vector j at
// orth_host + j*n_local), every computed pair returned in comparator order without EigenPairManager filtering.
template <typename T> struct IterationSpec {
  int64_t nroot;
  int64_t n_orth;
  const T* orth_host;
};

// Bytes of one Krylov-basis slab of a default run on an operator of this shape (engine.cpp)
int64_t default_slab_bytes(int64_t n, int64_t n_local, int64_t n_shard, int elem_bytes, const Tuning& tune);

// Whole-loop drivers
template <typename T>
void lanczos_run(ll_context* ctx, ll_operator* op, const ll_lanczos_params& P, double* eigvals, T* eigvecs,
                 int64_t* n_found, int64_t* iter_counts, int64_t iter_cap, double* alpha_out, double* beta_out,
                 ll_run_stats* stats, const IterationSpec<T>* spec = nullptr);
template <typename T>
void expo_run(ll_context* ctx, ll_operator* op, const ll_expo_params& P, typename host_scalar<T>::type a,
              const T* input, T* output, int64_t* itern_out, ll_run_stats* stats);
template <typename T>
void taylor_run(ll_context* ctx, ll_operator* op, const ll_expo_params& P, typename host_scalar<T>::type a,
                const T* input, T* output, int64_t* nterms_out);

}  // namespace ll
