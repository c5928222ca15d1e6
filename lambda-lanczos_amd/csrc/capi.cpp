// extern "C" surface of liblanczos_hip.so: declared in include/lanczos_hip.h, which documents every entry point
// and cites the reference interface it replaces.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <limits>
#include <memory>

#include "engine.hpp"

namespace ll {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace ll

using namespace ll;

// ---------------------------------------------------------------- context workspace
static size_t grow(size_t have, size_t want) { return std::max(want, have + have / 2 + 64); }

void ll_context::ensure_partials(size_t doubles) {
  if (doubles <= partials_cap) return;
  if (d_partials) LL_HIP(hipFree(d_partials));
  d_partials = nullptr;
  partials_cap = grow(partials_cap, doubles);
  LL_HIP(hipMalloc((void**)&d_partials, partials_cap * sizeof(double)));
}
void ll_context::ensure_h(size_t doubles) {
  if (doubles <= h_cap) return;
  if (d_h) LL_HIP(hipFree(d_h));
  d_h = nullptr;
  h_cap = grow(h_cap, doubles);
  LL_HIP(hipMalloc((void**)&d_h, h_cap * sizeof(double)));
}
void ll_context::ensure_pinned(size_t doubles) {
  if (doubles <= pinned_cap) return;
  if (h_pinned) LL_HIP(hipHostFree(h_pinned));
  h_pinned = nullptr;
  pinned_cap = grow(pinned_cap, doubles);
  // device-mapped, coherent host memory: the publish kernel stores the per-iteration scalars straight into it
  hipError_t e = hipHostMalloc((void**)&h_pinned, pinned_cap * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    LL_HIP(hipHostMalloc((void**)&h_pinned, pinned_cap * sizeof(double), hipHostMallocDefault));
  }
}
void ll_context::ensure_coeff(size_t bytes) {
  if (bytes <= coeff_cap) return;
  if (d_coeff) LL_HIP(hipFree(d_coeff));
  d_coeff = nullptr;
  coeff_cap = grow(coeff_cap, bytes);
  LL_HIP(hipMalloc(&d_coeff, coeff_cap));
}
void ll_context::ensure_xfull(size_t bytes) {
  if (bytes <= xfull_cap) return;
  if (d_xfull) LL_HIP(hipFree(d_xfull));
  d_xfull = nullptr;
  xfull_cap = bytes;
  LL_HIP(hipMalloc(&d_xfull, xfull_cap));
}
void ll_context::ensure_halo(size_t bytes) {
  if (bytes <= halo_cap) return;
  if (d_halo) LL_HIP(hipFree(d_halo));
  d_halo = nullptr;
  halo_cap = bytes;
  LL_HIP(hipMalloc(&d_halo, halo_cap));
}
void* ll_context::ensure_stage(size_t bytes) {
  if (bytes <= stage_cap) return h_stage;
  if (h_stage) LL_HIP(hipHostFree(h_stage));
  h_stage = nullptr;
  stage_cap = grow(stage_cap, bytes);
  LL_HIP(hipHostMalloc(&h_stage, stage_cap, hipHostMallocDefault));
  return h_stage;
}
void ll_context::sync() { LL_HIP(hipStreamSynchronize(stream)); }

// ---------------------------------------------------------------- operator storage
ll_operator::~ll_operator() {
  if (ctx) (void)hipSetDevice(ctx->device);
  for (void* q : {d_row_ptr, (void*)d_tile_rows, d_dense, d_onsite, (void*)d_pb_segq, (void*)d_pb_segdest,
                  (void*)d_pb_rptr, d_pb_val, (void*)d_pb_col, (void*)d_pb_row, d_pb_prod})
    if (q) (void)hipFree(q);
  if (owns_arrays) {
    if (d_col) (void)hipFree(d_col);
    if (d_val) (void)hipFree(d_val);
  }
}

// ---------------------------------------------------------------- exception -> status
template <typename F> static int guarded(F&& f) {
  try {
    f();
    return LL_OK;
  } catch (const Failure& e) {
    return e.code;
  } catch (const std::bad_alloc&) {
    set_error("host allocation failed");
    return LL_ERR_ALLOC;
  } catch (const std::exception& e) {
    set_error(std::string("unexpected exception: ") + e.what());
    return LL_ERR_INVALID;
  }
}

static void use(ll_context* ctx) {
  LL_REQUIRE(ctx != nullptr, "null context");
  LL_HIP(hipSetDevice(ctx->device));
}

extern "C" {

const char* ll_last_error(void) { return g_last_error.c_str(); }
int ll_version(void) { return LL_VERSION_MAJOR * 1000 + LL_VERSION_MINOR; }

static int ctx_create_impl(int device, void* stream, bool own, ll_context** out) {
  return guarded([&] {
    LL_REQUIRE(out != nullptr, "null output pointer");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
      set_error(std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0") +
                "); this library has no CPU fallback");
      (void)hipGetLastError();
      throw Failure{LL_ERR_HIP};
    }
    LL_REQUIRE(device >= 0 && device < count, "device index out of range");
    LL_HIP(hipSetDevice(device));
    std::unique_ptr<ll_context> c(new ll_context);
    c->device = device;
    if (own) {
      LL_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
      c->own_stream = true;
    } else {
      c->stream = (hipStream_t)stream;
    }
    LL_HIP(hipMalloc((void**)&c->d_scal, kScalCount * sizeof(double)));
    LL_HIP(hipMemset(c->d_scal, 0, kScalCount * sizeof(double)));
    *out = c.release();
  });
}
int ll_ctx_create(int device, ll_context** out) { return ctx_create_impl(device, nullptr, true, out); }
int ll_ctx_create_on_stream(int device, void* hip_stream, ll_context** out) {
  return ctx_create_impl(device, hip_stream, false, out);
}
int ll_ctx_destroy(ll_context* ctx) {
  return guarded([&] {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    comm_destroy(ctx->comm);
    if (ctx->d_partials) (void)hipFree(ctx->d_partials);
    if (ctx->d_h) (void)hipFree(ctx->d_h);
    if (ctx->d_scal) (void)hipFree(ctx->d_scal);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->d_coeff) (void)hipFree(ctx->d_coeff);
    if (ctx->d_xfull) (void)hipFree(ctx->d_xfull);
    if (ctx->d_halo) (void)hipFree(ctx->d_halo);
    for (auto& c : ctx->slab_cache) (void)hipFree(c.first);
    if (ctx->t0) (void)hipEventDestroy(ctx->t0);
    if (ctx->t1) (void)hipEventDestroy(ctx->t1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
  });
}
int ll_ctx_stream(ll_context* ctx, void** out) {
  return guarded([&] {
    LL_REQUIRE(ctx && out, "null argument");
    *out = (void*)ctx->stream;
  });
}
int ll_ctx_synchronize(ll_context* ctx) {
  return guarded([&] {
    use(ctx);
    ctx->sync();
  });
}
int ll_ctx_release_cache(ll_context* ctx) {
  return guarded([&] {
    use(ctx);
    ctx->sync();
    for (auto& c : ctx->slab_cache) (void)hipFree(c.first);
    ctx->slab_cache.clear();
  });
}
int ll_ctx_set_profiling(ll_context* ctx, int enabled) {
  return guarded([&] {
    LL_REQUIRE(ctx != nullptr, "null context");
    ctx->profiling = enabled != 0;
  });
}

// ---------------------------------------------------------------- device timer (HIP events on the context's stream)
int ll_timer_start(ll_context* ctx) {
  return guarded([&] {
    use(ctx);
    if (!ctx->t0) {
      LL_HIP(hipEventCreate(&ctx->t0));
      LL_HIP(hipEventCreate(&ctx->t1));
    }
    LL_HIP(hipEventRecord(ctx->t0, ctx->stream));
  });
}
int ll_timer_stop(ll_context* ctx, double* ms_out) {
  return guarded([&] {
    use(ctx);
    LL_REQUIRE(ctx->t0 != nullptr && ms_out != nullptr, "timer not started");
    LL_HIP(hipEventRecord(ctx->t1, ctx->stream));
    LL_HIP(hipEventSynchronize(ctx->t1));
    float ms = 0.f;
    LL_HIP(hipEventElapsedTime(&ms, ctx->t0, ctx->t1));
    *ms_out = (double)ms;
  });
}

// ---------------------------------------------------------------- multi-GPU
int ll_comm_unique_id(void* id) {
  return guarded([&] {
    LL_REQUIRE(id != nullptr, "null id buffer");
    comm_unique_id(id);
  });
}
int ll_comm_init(ll_context* ctx, const void* id, int rank, int n_ranks) {
  return guarded([&] {
    use(ctx);
    LL_REQUIRE(id != nullptr, "null id");
    LL_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "rank out of range");
    LL_REQUIRE(ctx->comm == nullptr, "communicator already attached");
    ctx->comm = comm_create(id, rank, n_ranks, ctx->device);
    ctx->rank = rank;
    ctx->nranks = n_ranks;
  });
}
int ll_comm_rank(ll_context* ctx, int* rank, int* n_ranks) {
  return guarded([&] {
    LL_REQUIRE(ctx != nullptr, "null context");
    if (rank) *rank = ctx->rank;
    if (n_ranks) *n_ranks = ctx->nranks;
  });
}
int ll_partition(int64_t n, int n_ranks, int rank, int64_t* row_begin, int64_t* n_local) {
  return guarded([&] {
    LL_REQUIRE(n >= 0 && n_ranks >= 1 && rank >= 0 && rank < n_ranks, "bad partition request");
    const int64_t shard = (n + n_ranks - 1) / n_ranks;
    const int64_t b = std::min<int64_t>(n, shard * rank), e = std::min<int64_t>(n, shard * (rank + 1));
    if (row_begin) *row_begin = b;
    if (n_local) *n_local = e - b;
  });
}

// ---------------------------------------------------------------- memory helpers
int ll_malloc(ll_context* ctx, size_t bytes, void** out) {
  return guarded([&] {
    use(ctx);
    LL_REQUIRE(out != nullptr, "null output pointer");
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) {
      set_error(std::string("hipMalloc(") + std::to_string(bytes) + ") failed: " + hipGetErrorString(e));
      throw Failure{LL_ERR_ALLOC};
    }
  });
}
int ll_free(ll_context* ctx, void* p) {
  return guarded([&] {
    use(ctx);
    if (p) LL_HIP(hipFree(p));
  });
}
int ll_memcpy_h2d(ll_context* ctx, void* dst, const void* src, size_t bytes) {
  return guarded([&] {
    use(ctx);
    LL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    ctx->sync();
  });
}
int ll_memcpy_d2h(ll_context* ctx, void* dst, const void* src, size_t bytes) {
  return guarded([&] {
    use(ctx);
    LL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
  });
}
int ll_memset(ll_context* ctx, void* dst, int byte, size_t bytes) {
  return guarded([&] {
    use(ctx);
    LL_HIP(hipMemsetAsync(dst, byte, bytes, ctx->stream));
  });
}

// ---------------------------------------------------------------- operators
extern "C++" {
namespace {

inline double abs2_host(double v) { return v * v; }
inline double abs2_host(float v) { return (double)v * (double)v; }
inline double abs2_host(zc v) { return v.re * v.re + v.im * v.im; }
inline double abs2_host(cf v) { return (double)v.re * (double)v.re + (double)v.im * (double)v.im; }

// SpMV tiles: runs of whole rows with <= kSpmvTileNnz nonzeros and <= kBlock rows; a longer row is alone.
void build_tiles(const int64_t* rp, int64_t nrows, std::vector<int32_t>& tiles) {
  tiles.clear();
  tiles.push_back(0);
  int64_t r = 0;
  while (r < nrows) {
    int64_t r1 = r;
    while (r1 < nrows && (r1 - r) < kBlock && rp[r1 + 1] - rp[r] <= kSpmvTileNnz) ++r1;
    if (r1 == r) r1 = r + 1;
    tiles.push_back((int32_t)r1);
    r = r1;
  }
}

template <typename T>
void finish_csr(ll_operator* op, const int64_t* rp_host) {
  ll_context* ctx = op->ctx;
  const int64_t nr = op->n_local;
  std::vector<int32_t> tiles;
  build_tiles(rp_host, nr, tiles);
  op->ntiles = (int)tiles.size() - 1;
  LL_HIP(hipMalloc((void**)&op->d_tile_rows, tiles.size() * sizeof(int32_t)));
  LL_HIP(hipMemcpy(op->d_tile_rows, tiles.data(), tiles.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  // 64-bit row offsets once nnz exceeds int32 (LL_FORCE_RP64=1: exercise that kernel variant on small test matrices)
  op->rp64 = op->nnz > (int64_t)0x7fffffff || (std::getenv("LL_FORCE_RP64") && std::atoi(std::getenv("LL_FORCE_RP64")) != 0);
  if (op->rp64) {
    LL_HIP(hipMalloc(&op->d_row_ptr, (size_t)(nr + 1) * sizeof(int64_t)));
    LL_HIP(hipMemcpy(op->d_row_ptr, rp_host, (size_t)(nr + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  } else {
    std::vector<int32_t> rp32((size_t)nr + 1);
    for (int64_t i = 0; i <= nr; ++i) rp32[i] = (int32_t)rp_host[i];
    LL_HIP(hipMalloc(&op->d_row_ptr, (size_t)(nr + 1) * sizeof(int32_t)));
    LL_HIP(hipMemcpy(op->d_row_ptr, rp32.data(), (size_t)(nr + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  (void)ctx;
}

// Propagation-blocked image (kernels.hip pb_phase1/pb_phase2): the entries in column-block order (values + 16-bit
// local columns) and the matching row-block order (16-bit local rows), plus the segment tables that connect them.
// A segment = all entries of one (column block, row block) pair; inside a segment both orders agree (row-major,
// original order within a row), so destination = segment base + offset.
template <typename T>
bool build_pb(ll_operator* op, const int64_t* rp, const int32_t* ci, const T* va) {
  const int64_t nr = op->n_local, nc = op->n;
  const bool z = scalar_traits<T>::is_complex;
  // LDS budget per workgroup (160 KiB): phase 1 holds the x slice + two tables of nrb entries, phase 2 only the y slice
  (void)z;
  const int64_t col_max = std::min<int64_t>(65536, (104 * 1024) / (int64_t)sizeof(T));          // x slice <= 104 KiB
  const int64_t row_max = std::min<int64_t>(65536, (152 * 1024) / (int64_t)sizeof(acc_t<T>));   // y slice <= 152 KiB
  auto block_len = [&](int64_t len, int64_t slice_max, const char* env) {
    int64_t m = std::max<int64_t>(1, (len + 256 * slice_max - 1) / (256 * slice_max));
    int64_t b = std::max<int64_t>(16, (len + 256 * m - 1) / (256 * m));
    if (const char* e = std::getenv("LL_PB_BLOCK")) b = std::max(4, std::atoi(e));
    if (const char* e = std::getenv(env)) b = std::max(4, std::atoi(e));
    return std::min<int64_t>(b, slice_max);
  };
  const int64_t cb_cols = block_len(nc, col_max, "LL_PB_COL_BLOCK"), rb_rows = block_len(nr, row_max, "LL_PB_ROW_BLOCK");
  const int64_t ncb = (nc + cb_cols - 1) / cb_cols, nrb = std::max<int64_t>(1, (nr + rb_rows - 1) / rb_rows);
  if (ncb * nrb > (int64_t)24 << 20) return false;  // segment tables would not pay off (n beyond ~6e7): keep CSR
  // segment sizes
  std::vector<int64_t> cnt((size_t)ncb * nrb, 0);
#pragma omp parallel for schedule(dynamic, 2)
  for (int64_t r = 0; r < nrb; ++r) {
    const int64_t i0 = r * rb_rows, i1 = std::min(nr, i0 + rb_rows);
    for (int64_t p = rp[i0]; p < rp[i1]; ++p) ++cnt[(size_t)(ci[p] / cb_cols) * nrb + r];
  }
  // every segment is padded to 16 entries: kernels move quads (4 entries per lane, 16-byte accesses) and every run of
  // products written by phase 1 starts and ends on a 128-byte line (measured 3.5 % faster than quad padding)
  int64_t pad = 16;
  if (const char* e = std::getenv("LL_PB_PAD")) pad = std::max(4, std::atoi(e) / 4 * 4);
  for (auto& v : cnt) v = (v + pad - 1) / pad * pad;
  // column-block order: segments (c, r) with r fastest; row-block order: (r, c) with c fastest
  std::vector<int64_t> segq((size_t)ncb * (nrb + 1)), segdest((size_t)ncb * nrb), rptr((size_t)nrb + 1);
  {
    int64_t q = 0;
    for (int64_t c = 0; c < ncb; ++c) {
      for (int64_t r = 0; r < nrb; ++r) {
        segq[(size_t)c * (nrb + 1) + r] = q;
        q += cnt[(size_t)c * nrb + r];
      }
      segq[(size_t)c * (nrb + 1) + nrb] = q;
    }
    int64_t d = 0;
    for (int64_t r = 0; r < nrb; ++r) {
      rptr[(size_t)r] = d;
      for (int64_t c = 0; c < ncb; ++c) {
        segdest[(size_t)c * nrb + r] = d;
        d += cnt[(size_t)c * nrb + r];
      }
    }
    rptr[(size_t)nrb] = d;
  }
  const size_t nnz = (size_t)rptr[(size_t)nrb];  // padded entry count
  std::vector<T> pval(std::max<size_t>(nnz, 4));
  std::memset((void*)pval.data(), 0, pval.size() * sizeof(T));
  std::vector<uint16_t> pcol(std::max<size_t>(nnz, 4), 0), prow(std::max<size_t>(nnz, 4), 0);
#pragma omp parallel
  {
    std::vector<int64_t> fill((size_t)ncb);
#pragma omp for schedule(dynamic, 2)
    for (int64_t r = 0; r < nrb; ++r) {
      std::fill(fill.begin(), fill.end(), 0);
      const int64_t i0 = r * rb_rows, i1 = std::min(nr, i0 + rb_rows);
      for (int64_t i = i0; i < i1; ++i)
        for (int64_t p = rp[i]; p < rp[i + 1]; ++p) {
          const int64_t c = ci[p] / cb_cols;
          const int64_t off = fill[(size_t)c]++;
          const int64_t q = segq[(size_t)c * (nrb + 1) + r] + off;
          pval[(size_t)q] = va[p];
          pcol[(size_t)q] = (uint16_t)(ci[p] - c * cb_cols);
          prow[(size_t)(segdest[(size_t)c * nrb + r] + off)] = (uint16_t)(i - i0);
        }
    }
  }
  op->pb_ncb = (int)ncb;
  op->pb_nrb = (int)nrb;
  op->pb_cb_cols = (int)cb_cols;
  op->pb_rb_rows = (int)rb_rows;
  auto up = [](void** dst, const void* src, size_t bytes) {
    LL_HIP(hipMalloc(dst, std::max<size_t>(bytes, 8)));
    LL_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  };
  up((void**)&op->d_pb_segq, segq.data(), segq.size() * sizeof(int64_t));
  up((void**)&op->d_pb_segdest, segdest.data(), segdest.size() * sizeof(int64_t));
  up((void**)&op->d_pb_rptr, rptr.data(), rptr.size() * sizeof(int64_t));
  up(&op->d_pb_val, pval.data(), nnz * sizeof(T));
  up((void**)&op->d_pb_col, pcol.data(), nnz * sizeof(uint16_t));
  up((void**)&op->d_pb_row, prow.data(), nnz * sizeof(uint16_t));
  LL_HIP(hipMalloc(&op->d_pb_prod, std::max<size_t>(nnz, 4) * sizeof(T)));
  return true;
}

// Time both SpMV kernels on the device with the actual matrix and keep the faster one.
template <typename T> void autotune_spmv(ll_operator* op) {
  ll_context* ctx = op->ctx;
  hipStream_t s = ctx->stream;
  const size_t xn = (size_t)std::max<int64_t>(op->n, op->n_shard * std::max(1, ctx->nranks));
  T *x = nullptr, *y = nullptr;
  LL_HIP(hipMalloc((void**)&x, xn * sizeof(T)));
  LL_HIP(hipMalloc((void**)&y, (size_t)std::max<int64_t>(op->n_local, 1) * sizeof(T)));
  LL_HIP(hipMemsetAsync(x, 0, xn * sizeof(T), s));
  hipEvent_t e0, e1;
  LL_HIP(hipEventCreate(&e0));
  LL_HIP(hipEventCreate(&e1));
  float best = 0.f;
  int best_kind = LL_SPMV_CSR_STREAM;
  for (int kind : {LL_SPMV_CSR_STREAM, LL_SPMV_PB}) {
    float t_kind = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      LL_HIP(hipEventRecord(e0, s));
      if (kind == LL_SPMV_PB) launch_spmv_pb<T>(*op, x, x + op->row_begin, y, 0.0, nullptr, s);
      else launch_spmv<T>(*op, x, x + op->row_begin, y, 0.0, nullptr, s);
      LL_HIP(hipEventRecord(e1, s));
      LL_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      LL_HIP(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0) t_kind = std::min(t_kind, ms);
    }
    if (kind == LL_SPMV_CSR_STREAM || t_kind < best) {
      best = t_kind;
      best_kind = kind;
    }
  }
  op->spmv_kind = best_kind;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(x);
  (void)hipFree(y);
}

template <typename T>
void create_csr(ll_context* ctx, int64_t nr, int64_t nc, int64_t row_begin, const int64_t* rp, const int32_t* ci,
                const void* va, bool on_device, ll_operator** out) {
  use(ctx);
  LL_REQUIRE(out && rp && (ci || nr == 0) && (va || nr == 0), "null argument");
  LL_REQUIRE(nr >= 0 && nc >= 1 && row_begin >= 0 && row_begin + nr <= nc, "bad shape");
  LL_REQUIRE(nr < (int64_t)0x7fffffff && nc < (int64_t)0x7fffffff, "dimension exceeds int32 indices");
  std::vector<int64_t> rp_copy;
  const int64_t* rp_host = rp;
  if (on_device) {
    rp_copy.resize((size_t)nr + 1);
    LL_HIP(hipMemcpy(rp_copy.data(), rp, (size_t)(nr + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
    rp_host = rp_copy.data();
  }
  LL_REQUIRE(rp_host[0] == 0, "row_ptr must start at 0");
  for (int64_t i = 0; i < nr; ++i) LL_REQUIRE(rp_host[i + 1] >= rp_host[i], "row_ptr must be non-decreasing");
  std::unique_ptr<ll_operator> op(new ll_operator);
  op->kind = ll_operator::CSR;
  op->is_complex = scalar_traits<T>::is_complex;
  op->elem_bytes = (int)sizeof(T);
  op->ctx = ctx;
  op->n = nc;
  op->n_local = nr;
  op->row_begin = row_begin;
  op->nnz = rp_host[nr];
  if (ctx->nranks > 1) {
    op->n_shard = (nc + ctx->nranks - 1) / ctx->nranks;
    LL_REQUIRE(row_begin == std::min<int64_t>(nc, op->n_shard * ctx->rank) &&
                   nr == std::min<int64_t>(nc, op->n_shard * (ctx->rank + 1)) - row_begin,
               "sharded operators must use the ll_partition() row ranges");
  } else {
    op->n_shard = nc;
    LL_REQUIRE(row_begin == 0 && nr == nc, "a single-GPU context needs the whole matrix (row_begin 0, n_rows == n_cols)");
  }
  const size_t nnz = (size_t)op->nnz;
  if (!on_device) {  // max absolute row sum, for ll_op_inf_norm (determine_eigenvalue_offset.cpp:12-29)
    const T* v = (const T*)va;
    double mx = 0.0;
#pragma omp parallel for reduction(max : mx) schedule(static)
    for (int64_t i = 0; i < nr; ++i) {
      double rs = 0.0;
      for (int64_t p = rp_host[i]; p < rp_host[i + 1]; ++p) rs += std::sqrt(abs2_host(v[p]));
      mx = std::max(mx, rs);
    }
    op->inf_norm = mx;
  }
  if (on_device) {
    op->owns_arrays = false;
    op->d_col = const_cast<int32_t*>(ci);
    op->d_val = const_cast<void*>(va);
  } else {
    if (!on_device && nnz) {
      for (size_t p = 0; p < nnz; ++p) LL_REQUIRE(ci[p] >= 0 && ci[p] < nc, "column index out of range");
    }
    LL_HIP(hipMalloc((void**)&op->d_col, std::max<size_t>(nnz, 1) * sizeof(int32_t)));
    LL_HIP(hipMalloc(&op->d_val, std::max<size_t>(nnz, 1) * sizeof(T)));
    LL_HIP(hipMemcpy(op->d_col, ci, nnz * sizeof(int32_t), hipMemcpyHostToDevice));
    LL_HIP(hipMemcpy(op->d_val, va, nnz * sizeof(T), hipMemcpyHostToDevice));
  }
  finish_csr<T>(op.get(), rp_host);
  op->spmv_kind = LL_SPMV_CSR_STREAM;
  const char* fmt = std::getenv("LL_SPMV_KERNEL");
  const std::string want = fmt ? fmt : "auto";
  if (want != "csr" && nnz > 0) {
    // the propagation-blocked image is built on the host; arrays that are already in HBM are copied back once for it
    std::vector<int32_t> ci_copy;
    std::vector<T> va_copy;
    const int32_t* ci_host = ci;
    const T* va_host = (const T*)va;
    if (on_device) {
      ci_copy.resize(nnz);
      va_copy.resize(nnz);
      LL_HIP(hipMemcpy(ci_copy.data(), ci, nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
      LL_HIP(hipMemcpy(va_copy.data(), va, nnz * sizeof(T), hipMemcpyDeviceToHost));
      ci_host = ci_copy.data();
      va_host = va_copy.data();
      for (size_t p = 0; p < nnz; ++p) LL_REQUIRE(ci_host[p] >= 0 && ci_host[p] < nc, "column index out of range");
      double mx = 0.0;
#pragma omp parallel for reduction(max : mx) schedule(static)
      for (int64_t i = 0; i < nr; ++i) {
        double rs = 0.0;
        for (int64_t p = rp_host[i]; p < rp_host[i + 1]; ++p) rs += std::sqrt(abs2_host(va_host[p]));
        mx = std::max(mx, rs);
      }
      op->inf_norm = mx;
    }
    if (build_pb<T>(op.get(), rp_host, ci_host, va_host)) {
      if (want == "pb") op->spmv_kind = LL_SPMV_PB;
      else autotune_spmv<T>(op.get());
    }
  }
  *out = op.release();
}

// Row ranges of a sharded operator must be the ll_partition() ones (equal shard strides).
void set_partition(ll_context* ctx, ll_operator* op, int64_t n, int64_t row_begin, int64_t n_local) {
  op->n = n;
  op->n_local = n_local;
  op->row_begin = row_begin;
  if (ctx->nranks > 1) {
    op->n_shard = (n + ctx->nranks - 1) / ctx->nranks;
    LL_REQUIRE(row_begin == std::min<int64_t>(n, op->n_shard * ctx->rank) &&
                   n_local == std::min<int64_t>(n, op->n_shard * (ctx->rank + 1)) - row_begin,
               "sharded operators must use the ll_partition() row ranges");
  } else {
    op->n_shard = n;
    LL_REQUIRE(row_begin == 0 && n_local == n, "a single-GPU context needs the whole operator (row_begin 0, n_local == n)");
  }
}

template <typename T>
void create_dense(ll_context* ctx, int64_t nr, int64_t nc, int64_t row_begin, const void* a, ll_operator** out) {
  use(ctx);
  LL_REQUIRE(out && (a || nr == 0), "null argument");
  LL_REQUIRE(nr >= 0 && nc >= 1 && row_begin >= 0 && row_begin + nr <= nc, "bad shape");
  std::unique_ptr<ll_operator> op(new ll_operator);
  op->kind = ll_operator::DENSE;
  op->is_complex = scalar_traits<T>::is_complex;
  op->elem_bytes = (int)sizeof(T);
  op->ctx = ctx;
  set_partition(ctx, op.get(), nc, row_begin, nr);
  op->nnz = nr * nc;
  const T* v = (const T*)a;
  double mx = 0.0;
#pragma omp parallel for reduction(max : mx) schedule(static)
  for (int64_t i = 0; i < nr; ++i) {
    double rs = 0.0;
    for (int64_t j = 0; j < nc; ++j) rs += std::sqrt(abs2_host(v[i * nc + j]));
    mx = std::max(mx, rs);
  }
  op->inf_norm = mx;
  const size_t bytes = (size_t)nr * (size_t)nc * sizeof(T);
  LL_HIP(hipMalloc(&op->d_dense, std::max<size_t>(bytes, 16)));
  if (bytes) LL_HIP(hipMemcpy(op->d_dense, a, bytes, hipMemcpyHostToDevice));
  *out = op.release();
}

template <typename T>
void create_stencil(ll_context* ctx, const ll_stencil_desc* d, int64_t row_begin, int64_t n_local, const double* onsite,
                    ll_operator** out) {
  use(ctx);
  LL_REQUIRE(out && d, "null argument");
  LL_REQUIRE(d->ndim >= 1 && d->ndim <= 3, "ndim must be 1, 2 or 3");
  int64_t n = 1;
  for (int k = 0; k < d->ndim; ++k) {
    LL_REQUIRE(d->dims[k] >= 1, "lattice dimensions must be positive");
    LL_REQUIRE(n <= ((int64_t)1 << 40) / d->dims[k], "lattice too large");
    n *= d->dims[k];
    if (!scalar_traits<T>::is_complex) {
      LL_REQUIRE(d->hop_im[k] == 0.0, "complex hopping needs a complex storage type");
      for (int e = 0; e < 3; ++e) LL_REQUIRE(d->phase_grad[k][e] == 0.0, "Peierls phases need a complex storage type");
    }
  }
  std::unique_ptr<ll_operator> op(new ll_operator);
  op->kind = ll_operator::STENCIL;
  op->is_complex = scalar_traits<T>::is_complex;
  op->elem_bytes = (int)sizeof(T);
  op->ctx = ctx;
  set_partition(ctx, op.get(), n, row_begin, n_local);
  LL_REQUIRE(n_local < (int64_t)0x7fffffff, "shard exceeds 32-bit local indices");
  op->st = *d;
  int64_t stride = 1;
  for (int k = d->ndim - 1; k >= 0; --k) {
    op->st_stride[k] = stride;
    stride *= d->dims[k];
  }
  op->st_halo = op->st_stride[0];
  if (ctx->nranks > 1) {
    const int64_t last = n - op->n_shard * (ctx->nranks - 1);  // the shortest shard
    LL_REQUIRE(last >= op->st_halo && op->n_shard >= op->st_halo,
               "lattice operator: every shard must hold at least one hyperplane (n / dims[0] sites); use fewer ranks");
  }
  op->nnz = 0;
  double hops = 0.0;
  for (int k = 0; k < d->ndim; ++k) hops += 2.0 * std::hypot(d->hop_re[k], d->hop_im[k]);
  double diag_max = std::abs(d->diag);
  if (onsite) {
    diag_max = 0.0;
    for (int64_t i = 0; i < n_local; ++i) diag_max = std::max(diag_max, std::abs(d->diag + onsite[i]));
    typedef typename scalar_traits<T>::real R;
    std::vector<R> tmp((size_t)n_local);
    for (int64_t i = 0; i < n_local; ++i) tmp[(size_t)i] = (R)onsite[i];
    LL_HIP(hipMalloc(&op->d_onsite, std::max<size_t>((size_t)n_local * sizeof(R), 16)));
    LL_HIP(hipMemcpy(op->d_onsite, tmp.data(), (size_t)n_local * sizeof(R), hipMemcpyHostToDevice));
  }
  op->inf_norm = diag_max + hops;  // an upper bound of the max absolute row sum (equal to it for interior sites)
  *out = op.release();
}

template <typename T> void create_cb(ll_context* ctx, int64_t n, ll_operator::Kind kind, ll_operator** out) {
  use(ctx);
  LL_REQUIRE(out != nullptr && n >= 1, "bad argument");
  LL_REQUIRE(ctx->nranks == 1, "callback operators are not supported on sharded contexts");
  ll_operator* op = new ll_operator;
  op->kind = kind;
  op->is_complex = scalar_traits<T>::is_complex;
  op->elem_bytes = (int)sizeof(T);
  op->ctx = ctx;
  op->n = op->n_local = op->n_shard = n;
  *out = op;
}

}  // namespace
}  // extern "C++"

int ll_op_create_csr_d(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const int64_t* rp, const int32_t* ci,
                       const double* va, ll_operator** out) {
  return guarded([&] { create_csr<double>(ctx, nr, nc, rb, rp, ci, va, false, out); });
}
int ll_op_create_csr_z(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const int64_t* rp, const int32_t* ci,
                       const void* va, ll_operator** out) {
  return guarded([&] { create_csr<zc>(ctx, nr, nc, rb, rp, ci, va, false, out); });
}
int ll_op_create_csr_dev_d(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const int64_t* rp, const int32_t* ci,
                           const double* va, ll_operator** out) {
  return guarded([&] { create_csr<double>(ctx, nr, nc, rb, rp, ci, va, true, out); });
}
int ll_op_create_csr_dev_z(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const int64_t* rp, const int32_t* ci,
                           const void* va, ll_operator** out) {
  return guarded([&] { create_csr<zc>(ctx, nr, nc, rb, rp, ci, va, true, out); });
}
extern "C++" {
namespace {
// {row, col, value} triplets (sample2_sparse.cpp:14-47) -> CSR, stable in input order inside a row (duplicates kept).
template <typename T>
void create_coo(ll_context* ctx, int64_t n, int64_t nnz, const int32_t* rows, const int32_t* cols, const void* vals,
                ll_operator** out) {
  LL_REQUIRE(n >= 1 && nnz >= 0 && (nnz == 0 || (rows && cols && vals)), "bad argument");
  std::vector<int64_t> rp((size_t)n + 1, 0);
  for (int64_t p = 0; p < nnz; ++p) {
    LL_REQUIRE(rows[p] >= 0 && rows[p] < n, "row index out of range");
    ++rp[(size_t)rows[p] + 1];
  }
  for (int64_t i = 0; i < n; ++i) rp[(size_t)i + 1] += rp[(size_t)i];
  std::vector<int64_t> cur(rp.begin(), rp.end() - 1);
  std::vector<int32_t> ci((size_t)std::max<int64_t>(nnz, 1));
  std::vector<T> va((size_t)std::max<int64_t>(nnz, 1));
  const T* v = (const T*)vals;
  for (int64_t p = 0; p < nnz; ++p) {
    const int64_t q = cur[(size_t)rows[p]]++;
    ci[(size_t)q] = cols[p];
    va[(size_t)q] = v[p];
  }
  create_csr<T>(ctx, n, n, 0, rp.data(), ci.data(), va.data(), false, out);
}
}  // namespace
}  // extern "C++"
int ll_op_create_coo_d(ll_context* ctx, int64_t n, int64_t nnz, const int32_t* rows, const int32_t* cols,
                       const double* vals, ll_operator** out) {
  return guarded([&] { create_coo<double>(ctx, n, nnz, rows, cols, vals, out); });
}
int ll_op_create_coo_z(ll_context* ctx, int64_t n, int64_t nnz, const int32_t* rows, const int32_t* cols,
                       const void* vals, ll_operator** out) {
  return guarded([&] { create_coo<zc>(ctx, n, nnz, rows, cols, vals, out); });
}
int ll_op_inf_norm(const ll_operator* op, double* out) {
  return guarded([&] {
    LL_REQUIRE(op != nullptr && out != nullptr, "null argument");
    LL_REQUIRE(op->inf_norm >= 0.0, "the infinity norm is only known for CSR/COO/dense/lattice operators created from host data");
    *out = op->inf_norm;
  });
}
int ll_op_create_dense_d(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const double* a, ll_operator** out) {
  return guarded([&] { create_dense<double>(ctx, nr, nc, rb, a, out); });
}
int ll_op_create_stencil_d(ll_context* ctx, const ll_stencil_desc* desc, int64_t rb, int64_t nl, const double* onsite,
                           ll_operator** out) {
  return guarded([&] { create_stencil<double>(ctx, desc, rb, nl, onsite, out); });
}
int ll_op_create_dense_z(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const void* a, ll_operator** out) {
  return guarded([&] { create_dense<zc>(ctx, nr, nc, rb, a, out); });
}
int ll_op_create_stencil_z(ll_context* ctx, const ll_stencil_desc* desc, int64_t rb, int64_t nl, const double* onsite,
                           ll_operator** out) {
  return guarded([&] { create_stencil<zc>(ctx, desc, rb, nl, onsite, out); });
}
int ll_op_create_host_d(ll_context* ctx, int64_t n, ll_host_mv_mul_d fn, void* user, ll_operator** out) {
  return guarded([&] {
    LL_REQUIRE(fn != nullptr, "null callback");
    create_cb<double>(ctx, n, ll_operator::HOST_CB, out);
    (*out)->host_fn = reinterpret_cast<ll_host_mv_mul_z>(fn);  // same ABI: only the pointee types differ
    (*out)->user = user;
  });
}
int ll_op_create_host_z(ll_context* ctx, int64_t n, ll_host_mv_mul_z fn, void* user, ll_operator** out) {
  return guarded([&] {
    LL_REQUIRE(fn != nullptr, "null callback");
    create_cb<zc>(ctx, n, ll_operator::HOST_CB, out);
    (*out)->host_fn = fn;
    (*out)->user = user;
  });
}
int ll_op_create_device_d(ll_context* ctx, int64_t n, ll_dev_mv_mul fn, void* user, ll_operator** out) {
  return guarded([&] {
    LL_REQUIRE(fn != nullptr, "null callback");
    create_cb<double>(ctx, n, ll_operator::DEV_CB, out);
    (*out)->dev_fn = fn;
    (*out)->user = user;
  });
}
int ll_op_create_device_z(ll_context* ctx, int64_t n, ll_dev_mv_mul fn, void* user, ll_operator** out) {
  return guarded([&] {
    LL_REQUIRE(fn != nullptr, "null callback");
    create_cb<zc>(ctx, n, ll_operator::DEV_CB, out);
    (*out)->dev_fn = fn;
    (*out)->user = user;
  });
}
int ll_op_destroy(ll_operator* op) {
  return guarded([&] { delete op; });  // ~ll_operator releases the device arrays
}
int ll_op_select_spmv(ll_operator* op, int kind) {
  return guarded([&] {
    LL_REQUIRE(op != nullptr && op->kind == ll_operator::CSR, "not a CSR operator");
    LL_REQUIRE(kind == LL_SPMV_CSR_STREAM || kind == LL_SPMV_PB, "unknown SpMV kernel");
    LL_REQUIRE(kind != LL_SPMV_PB || op->d_pb_val != nullptr,
               "operator has no propagation-blocked image (LL_SPMV_KERNEL=csr or n too large)");
    op->spmv_kind = kind;
  });
}
int ll_op_selected_spmv(const ll_operator* op, int* kind_out) {
  return guarded([&] {
    LL_REQUIRE(op != nullptr && kind_out != nullptr, "null argument");
    *kind_out = op->spmv_kind;
  });
}
int ll_op_info(const ll_operator* op, int64_t* n, int64_t* n_local, int64_t* nnz) {
  return guarded([&] {
    LL_REQUIRE(op != nullptr, "null operator");
    if (n) *n = op->n;
    if (n_local) *n_local = op->n_local;
    if (nnz) *nnz = op->nnz;
  });
}

// ---------------------------------------------------------------- primitives
extern "C++" {
namespace {
template <typename T> void spmv_impl(ll_context* ctx, ll_operator* op, const T* x, T* y, double offset, double* dot) {
  use(ctx);
  LL_REQUIRE(op && op->ctx == ctx && x && y, "bad argument");
  LL_REQUIRE(op->is_complex == scalar_traits<T>::is_complex && op->elem_bytes == (int)sizeof(T),
             "operator scalar type mismatch");
  Engine<T> E(ctx, op, op->n_local);
  E.apply(x, y, offset, dot ? E.S(kScalSpare) : nullptr);
  if (dot) E.fetch(E.S(kScalSpare), dot, 1);
}
template <typename T> void dot_impl(ll_context* ctx, int64_t n, const T* a, const T* b, double* out) {
  use(ctx);
  LL_REQUIRE(n >= 0 && a && b && out, "bad argument");
  Engine<T> E(ctx, nullptr, n);
  E.dot_dev(a, b, E.S(kScalSpare));
  E.fetch(E.S(kScalSpare), out, scalar_traits<T>::reals);
}
template <typename T> void nrm2_impl(ll_context* ctx, int64_t n, const T* v, double* out) {
  use(ctx);
  LL_REQUIRE(n >= 0 && v && out, "bad argument");
  Engine<T> E(ctx, nullptr, n);
  E.norm2_dev(v, E.S(kScalSpare));
  double nn = 0;
  E.fetch(E.S(kScalSpare), &nn, 1);
  *out = std::sqrt(nn);
}
template <typename T> void normalize_impl(ll_context* ctx, int64_t n, T* v, double* norm_out) {
  use(ctx);
  LL_REQUIRE(n >= 0 && v, "bad argument");
  Engine<T> E(ctx, nullptr, n);
  E.norm2_dev(v, E.S(kScalSpare));
  const NormRefs nr = E.plain_norm(E.S(kScalSpare));
  launch_scale<T>(n, v, 0.0, &nr, ctx->stream);
  if (norm_out) {
    double nn = 0;
    E.fetch(E.S(kScalSpare), &nn, 1);
    *norm_out = std::sqrt(nn);
  }
}
template <typename T>
void orth_impl(ll_context* ctx, int64_t n, int64_t nb, const T* basis, int64_t ld, T* w, int mode, double* norm_out,
               double* h_out) {
  use(ctx);
  LL_REQUIRE(n >= 0 && nb >= 0 && w && (basis || nb == 0) && ld >= n, "bad argument");
  LL_REQUIRE(mode >= LL_ORTH_CGS_DGKS && mode <= LL_ORTH_MGS, "unknown orthogonalisation mode");
  constexpr int R = scalar_traits<T>::reals;
  Engine<T> E(ctx, nullptr, n);
  RunList<T> runs;
  runs.ld = ld;
  runs.add(basis, nb);
  const ThreeTerm<T> no_tt{nullptr, nullptr, nullptr, NormRefs{nullptr, nullptr, nullptr, 0}};
  double* d_htot = nullptr;
  if (h_out && nb > 0) LL_HIP(hipMalloc((void**)&d_htot, (size_t)R * nb * sizeof(double)));
  const NormRefs refs = E.orth(w, runs, mode, no_tt, E.S(kScalScratch), d_htot);
  ctx->ensure_pinned(16);
  launch_publish(ctx->h_pinned + 8, nullptr, refs, ctx->stream);
  ctx->sync();
  if (norm_out) *norm_out = std::sqrt(ctx->h_pinned[9]);
  if (d_htot) {
    LL_HIP(hipMemcpy(h_out, d_htot, (size_t)R * nb * sizeof(double), hipMemcpyDeviceToHost));
    LL_HIP(hipFree(d_htot));
  }
}
template <typename T>
void gemv_impl(ll_context* ctx, int64_t n, int64_t m, const T* basis, int64_t ld, int64_t nout, const T* coeff,
               T* out, int64_t ld_out) {
  use(ctx);
  LL_REQUIRE(n >= 0 && m >= 1 && nout >= 1 && basis && coeff && out && ld >= n && ld_out >= n, "bad argument");
  Engine<T> E(ctx, nullptr, n);
  RunList<T> runs;
  runs.ld = ld;
  runs.add(basis, m);
  E.gemv(runs, m, (int)nout, coeff, out, ld_out);
}
}  // namespace
}  // extern "C++"

int ll_spmv_d(ll_context* ctx, ll_operator* op, const double* x, double* y, double offset, double* dot) {
  return guarded([&] { spmv_impl<double>(ctx, op, x, y, offset, dot); });
}
int ll_spmv_z(ll_context* ctx, ll_operator* op, const void* x, void* y, double offset, double* dot) {
  return guarded([&] { spmv_impl<zc>(ctx, op, (const zc*)x, (zc*)y, offset, dot); });
}
int ll_dot_d(ll_context* ctx, int64_t n, const double* a, const double* b, double* out) {
  return guarded([&] { dot_impl<double>(ctx, n, a, b, out); });
}
int ll_dot_z(ll_context* ctx, int64_t n, const void* a, const void* b, double* out) {
  return guarded([&] { dot_impl<zc>(ctx, n, (const zc*)a, (const zc*)b, out); });
}
int ll_nrm2_d(ll_context* ctx, int64_t n, const double* v, double* out) {
  return guarded([&] { nrm2_impl<double>(ctx, n, v, out); });
}
int ll_nrm2_z(ll_context* ctx, int64_t n, const void* v, double* out) {
  return guarded([&] { nrm2_impl<zc>(ctx, n, (const zc*)v, out); });
}
int ll_scal_d(ll_context* ctx, int64_t n, double a, double* v) {
  return guarded([&] {
    use(ctx);
    launch_scale<double>(n, v, a, nullptr, ctx->stream);
  });
}
int ll_scal_z(ll_context* ctx, int64_t n, double a, void* v) {
  return guarded([&] {
    use(ctx);
    launch_scale<zc>(n, (zc*)v, a, nullptr, ctx->stream);
  });
}
int ll_normalize_d(ll_context* ctx, int64_t n, double* v, double* norm_out) {
  return guarded([&] { normalize_impl<double>(ctx, n, v, norm_out); });
}
int ll_normalize_z(ll_context* ctx, int64_t n, void* v, double* norm_out) {
  return guarded([&] { normalize_impl<zc>(ctx, n, (zc*)v, norm_out); });
}
int ll_three_term_d(ll_context* ctx, int64_t n, double* w, const double* up, const double* uc, double beta,
                    double alpha) {
  return guarded([&] {
    use(ctx);
    LL_REQUIRE(w && uc, "null vector");
    launch_three_term<double>(n, w, up, uc, beta, alpha, ctx->stream);
  });
}
int ll_three_term_z(ll_context* ctx, int64_t n, void* w, const void* up, const void* uc, double beta, double alpha) {
  return guarded([&] {
    use(ctx);
    LL_REQUIRE(w && uc, "null vector");
    launch_three_term<zc>(n, (zc*)w, (const zc*)up, (const zc*)uc, beta, alpha, ctx->stream);
  });
}
int ll_orth_block_d(ll_context* ctx, int64_t n, int64_t nb, const double* basis, int64_t ld, double* w, int mode,
                    double* norm_out, double* h_out) {
  return guarded([&] { orth_impl<double>(ctx, n, nb, basis, ld, w, mode, norm_out, h_out); });
}
int ll_orth_block_z(ll_context* ctx, int64_t n, int64_t nb, const void* basis, int64_t ld, void* w, int mode,
                    double* norm_out, double* h_out) {
  return guarded([&] { orth_impl<zc>(ctx, n, nb, (const zc*)basis, ld, (zc*)w, mode, norm_out, h_out); });
}
int ll_gemv_basis_d(ll_context* ctx, int64_t n, int64_t m, const double* basis, int64_t ld, int64_t nout,
                    const double* coeff, double* out, int64_t ld_out) {
  return guarded([&] { gemv_impl<double>(ctx, n, m, basis, ld, nout, coeff, out, ld_out); });
}
int ll_gemv_basis_z(ll_context* ctx, int64_t n, int64_t m, const void* basis, int64_t ld, int64_t nout,
                    const double* coeff, void* out, int64_t ld_out) {
  return guarded([&] { gemv_impl<zc>(ctx, n, m, (const zc*)basis, ld, nout, (const zc*)coeff, (zc*)out, ld_out); });
}
int ll_tridiag_eig(int64_t m, const double* alpha, const double* beta, double* ev, double* q, int64_t* unconverged) {
  return guarded([&] {
    LL_REQUIRE(m >= 1 && alpha && ev && (beta || m == 1), "bad argument");
    const int64_t u = tridiag_qr(m, alpha, beta, ev, q);
    if (unconverged) *unconverged = u;
  });
}
int ll_tridiag_eigvecs(int64_t m, const double* alpha, const double* beta, int64_t nw, const double* lambdas,
                       double* out) {
  return guarded([&] {
    LL_REQUIRE(m >= 1 && nw >= 1 && alpha && lambdas && out && (beta || m == 1), "bad argument");
    tridiag_inverse_iteration(m, alpha, beta, nw, lambdas, out);
  });
}
int ll_tridiag_bisect(int64_t m, const double* alpha, const double* beta, int64_t k, double* out) {
  return guarded([&] {
    LL_REQUIRE(m >= 1 && alpha && out && k >= 0 && k < m && (beta || m == 1), "bad argument");
    *out = tridiag_bisect(m, alpha, beta, k);
  });
}

// ---------------------------------------------------------------- whole-loop entry points
int ll_lanczos_params_default(ll_lanczos_params* p, int64_t n, int find_maximum, int64_t num_eigs) {
  return guarded([&] {
    LL_REQUIRE(p != nullptr, "null params");
    std::memset(p, 0, sizeof(*p));
    p->matrix_size = n;                                           // LL:136
    p->max_iteration = n;                                         // LL:206
    p->eps = std::numeric_limits<double>::epsilon() * 1e3;        // LL:150
    p->find_maximum = find_maximum ? 1 : 0;                       // LL:153
    p->num_eigs = num_eigs;                                       // LL:156
    p->eigenvalue_offset = 0.0;                                   // LL:165
    p->num_eigs_per_iteration = 5;                                // LL:173
    p->initial_vector_size = 200;                                 // LL:181
    p->tridiag_mode = LL_TRIDIAG_QR;
    p->orth_mode = LL_ORTH_CGS_DGKS;
  });
}
int ll_expo_params_default(ll_expo_params* p, int64_t n) {
  return guarded([&] {
    LL_REQUIRE(p != nullptr, "null params");
    std::memset(p, 0, sizeof(*p));
    p->matrix_size = n;                                           // EX:44
    p->max_iteration = n;                                         // EX:81
    p->eps = std::numeric_limits<double>::epsilon() * 1e2;        // EX:58
    p->full_orthogonalize = 0;                                    // EX:63
    p->orth_mode = LL_ORTH_CGS_DGKS;
    p->initial_vector_size = 200;                                 // EX:71
  });
}

int ll_lanczos_run_d(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, double* eigvals, double* eigvecs,
                     int64_t* n_found, int64_t* iter_counts, int64_t iter_cap, double* alpha_out, double* beta_out,
                     ll_run_stats* stats) {
  return guarded([&] {
    LL_REQUIRE(ctx && p && eigvals && n_found, "null argument");
    lanczos_run<double>(ctx, op, *p, eigvals, eigvecs, n_found, iter_counts, iter_cap, alpha_out, beta_out, stats);
  });
}
int ll_lanczos_run_z(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, double* eigvals, void* eigvecs,
                     int64_t* n_found, int64_t* iter_counts, int64_t iter_cap, double* alpha_out, double* beta_out,
                     ll_run_stats* stats) {
  return guarded([&] {
    LL_REQUIRE(ctx && p && eigvals && n_found, "null argument");
    lanczos_run<zc>(ctx, op, *p, eigvals, (zc*)eigvecs, n_found, iter_counts, iter_cap, alpha_out, beta_out, stats);
  });
}
extern "C++" {
namespace {
template <typename T>
void run_iteration_impl(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, int64_t nroot, int64_t n_orth,
                        const void* orth, double* eigvals, void* eigvecs, int64_t* n_found, int64_t* itern,
                        double* alpha_out, double* beta_out, ll_run_stats* stats) {
  LL_REQUIRE(ctx && p && eigvals && n_found, "null argument");
  ll_lanczos_params q = *p;
  q.num_eigs = 1;  // unused by the single-pass mode; keep the range check of the common driver happy
  const IterationSpec<T> spec{nroot, n_orth, (const T*)orth};
  int64_t count = 0;
  lanczos_run<T>(ctx, op, q, eigvals, (T*)eigvecs, n_found, &count, 1, alpha_out, beta_out, stats, &spec);
  if (itern) *itern = count;
}
}  // namespace
}  // extern "C++"
int ll_lanczos_run_iteration_d(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, int64_t nroot,
                               int64_t n_orth, const double* orth, double* eigvals, double* eigvecs, int64_t* n_found,
                               int64_t* itern, double* alpha_out, double* beta_out, ll_run_stats* stats) {
  return guarded([&] {
    run_iteration_impl<double>(ctx, op, p, nroot, n_orth, orth, eigvals, eigvecs, n_found, itern, alpha_out, beta_out, stats);
  });
}
int ll_lanczos_run_iteration_z(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, int64_t nroot,
                               int64_t n_orth, const void* orth, double* eigvals, void* eigvecs, int64_t* n_found,
                               int64_t* itern, double* alpha_out, double* beta_out, ll_run_stats* stats) {
  return guarded([&] {
    run_iteration_impl<zc>(ctx, op, p, nroot, n_orth, orth, eigvals, eigvecs, n_found, itern, alpha_out, beta_out, stats);
  });
}
int ll_expo_run_d(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a, const double* input,
                  double* output, int64_t* itern, ll_run_stats* stats) {
  return guarded([&] {
    LL_REQUIRE(ctx && p && input && output && itern, "null argument");
    expo_run<double>(ctx, op, *p, a, input, output, itern, stats);
  });
}
int ll_expo_run_z(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a_re, double a_im,
                  const void* input, void* output, int64_t* itern, ll_run_stats* stats) {
  return guarded([&] {
    LL_REQUIRE(ctx && p && input && output && itern, "null argument");
    expo_run<zc>(ctx, op, *p, std::complex<double>(a_re, a_im), (const zc*)input, (zc*)output, itern, stats);
  });
}
int ll_expo_taylor_run_d(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a, const double* input,
                         double* output, int64_t* nterms) {
  return guarded([&] {
    LL_REQUIRE(ctx && p && input && output && nterms, "null argument");
    taylor_run<double>(ctx, op, *p, a, input, output, nterms);
  });
}
int ll_expo_taylor_run_z(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a_re, double a_im,
                         const void* input, void* output, int64_t* nterms) {
  return guarded([&] {
    LL_REQUIRE(ctx && p && input && output && nterms, "null argument");
    taylor_run<zc>(ctx, op, *p, std::complex<double>(a_re, a_im), (const zc*)input, (zc*)output, nterms);
  });
}

// ---------------------------------------------------------------- float storage types: _s (float), _c (complex float)
// Mechanical twins of the _z entry points above (scalars stay double; data pointers are float / re,im float pairs).
int ll_op_create_csr_c(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const int64_t* rp, const int32_t* ci,
                       const void* va, ll_operator** out) {
  return guarded([&] { create_csr<cf>(ctx, nr, nc, rb, rp, ci, va, false, out); });
}
int ll_op_create_csr_s(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const int64_t* rp, const int32_t* ci,
                       const float* va, ll_operator** out) {
  return guarded([&] { create_csr<float>(ctx, nr, nc, rb, rp, ci, va, false, out); });
}
int ll_op_create_csr_dev_c(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const int64_t* rp, const int32_t* ci,
                           const void* va, ll_operator** out) {
  return guarded([&] { create_csr<cf>(ctx, nr, nc, rb, rp, ci, va, true, out); });
}
int ll_op_create_csr_dev_s(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const int64_t* rp, const int32_t* ci,
                           const float* va, ll_operator** out) {
  return guarded([&] { create_csr<float>(ctx, nr, nc, rb, rp, ci, va, true, out); });
}
int ll_op_create_coo_c(ll_context* ctx, int64_t n, int64_t nnz, const int32_t* rows, const int32_t* cols,
                       const void* vals, ll_operator** out) {
  return guarded([&] { create_coo<cf>(ctx, n, nnz, rows, cols, vals, out); });
}
int ll_op_create_coo_s(ll_context* ctx, int64_t n, int64_t nnz, const int32_t* rows, const int32_t* cols,
                       const float* vals, ll_operator** out) {
  return guarded([&] { create_coo<float>(ctx, n, nnz, rows, cols, vals, out); });
}
int ll_op_create_dense_s(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const float* a, ll_operator** out) {
  return guarded([&] { create_dense<float>(ctx, nr, nc, rb, a, out); });
}
int ll_op_create_stencil_s(ll_context* ctx, const ll_stencil_desc* desc, int64_t rb, int64_t nl, const double* onsite,
                           ll_operator** out) {
  return guarded([&] { create_stencil<float>(ctx, desc, rb, nl, onsite, out); });
}
int ll_op_create_dense_c(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const void* a, ll_operator** out) {
  return guarded([&] { create_dense<cf>(ctx, nr, nc, rb, a, out); });
}
int ll_op_create_stencil_c(ll_context* ctx, const ll_stencil_desc* desc, int64_t rb, int64_t nl, const double* onsite,
                           ll_operator** out) {
  return guarded([&] { create_stencil<cf>(ctx, desc, rb, nl, onsite, out); });
}
int ll_op_create_host_c(ll_context* ctx, int64_t n, ll_host_mv_mul_z fn, void* user, ll_operator** out) {
  return guarded([&] {
    LL_REQUIRE(fn != nullptr, "null callback");
    create_cb<cf>(ctx, n, ll_operator::HOST_CB, out);
    (*out)->host_fn = fn;
    (*out)->user = user;
  });
}
int ll_op_create_host_s(ll_context* ctx, int64_t n, ll_host_mv_mul_s fn, void* user, ll_operator** out) {
  return guarded([&] {
    LL_REQUIRE(fn != nullptr, "null callback");
    create_cb<float>(ctx, n, ll_operator::HOST_CB, out);
    (*out)->host_fn = reinterpret_cast<ll_host_mv_mul_z>(fn);
    (*out)->user = user;
  });
}
int ll_op_create_device_c(ll_context* ctx, int64_t n, ll_dev_mv_mul fn, void* user, ll_operator** out) {
  return guarded([&] {
    LL_REQUIRE(fn != nullptr, "null callback");
    create_cb<cf>(ctx, n, ll_operator::DEV_CB, out);
    (*out)->dev_fn = fn;
    (*out)->user = user;
  });
}
int ll_op_create_device_s(ll_context* ctx, int64_t n, ll_dev_mv_mul fn, void* user, ll_operator** out) {
  return guarded([&] {
    LL_REQUIRE(fn != nullptr, "null callback");
    create_cb<float>(ctx, n, ll_operator::DEV_CB, out);
    (*out)->dev_fn = fn;
    (*out)->user = user;
  });
}
int ll_spmv_c(ll_context* ctx, ll_operator* op, const void* x, void* y, double offset, double* dot) {
  return guarded([&] { spmv_impl<cf>(ctx, op, (const cf*)x, (cf*)y, offset, dot); });
}
int ll_spmv_s(ll_context* ctx, ll_operator* op, const float* x, float* y, double offset, double* dot) {
  return guarded([&] { spmv_impl<float>(ctx, op, (const float*)x, (float*)y, offset, dot); });
}
int ll_dot_c(ll_context* ctx, int64_t n, const void* a, const void* b, double* out) {
  return guarded([&] { dot_impl<cf>(ctx, n, (const cf*)a, (const cf*)b, out); });
}
int ll_dot_s(ll_context* ctx, int64_t n, const float* a, const float* b, double* out) {
  return guarded([&] { dot_impl<float>(ctx, n, (const float*)a, (const float*)b, out); });
}
int ll_nrm2_c(ll_context* ctx, int64_t n, const void* v, double* out) {
  return guarded([&] { nrm2_impl<cf>(ctx, n, (const cf*)v, out); });
}
int ll_nrm2_s(ll_context* ctx, int64_t n, const float* v, double* out) {
  return guarded([&] { nrm2_impl<float>(ctx, n, (const float*)v, out); });
}
int ll_scal_c(ll_context* ctx, int64_t n, double a, void* v) {
  return guarded([&] {
    use(ctx);
    launch_scale<cf>(n, (cf*)v, a, nullptr, ctx->stream);
  });
}
int ll_scal_s(ll_context* ctx, int64_t n, double a, float* v) {
  return guarded([&] {
    use(ctx);
    launch_scale<float>(n, (float*)v, a, nullptr, ctx->stream);
  });
}
int ll_normalize_c(ll_context* ctx, int64_t n, void* v, double* norm_out) {
  return guarded([&] { normalize_impl<cf>(ctx, n, (cf*)v, norm_out); });
}
int ll_normalize_s(ll_context* ctx, int64_t n, float* v, double* norm_out) {
  return guarded([&] { normalize_impl<float>(ctx, n, (float*)v, norm_out); });
}
int ll_three_term_c(ll_context* ctx, int64_t n, void* w, const void* up, const void* uc, double beta, double alpha) {
  return guarded([&] {
    use(ctx);
    LL_REQUIRE(w && uc, "null vector");
    launch_three_term<cf>(n, (cf*)w, (const cf*)up, (const cf*)uc, beta, alpha, ctx->stream);
  });
}
int ll_three_term_s(ll_context* ctx, int64_t n, float* w, const float* up, const float* uc, double beta, double alpha) {
  return guarded([&] {
    use(ctx);
    LL_REQUIRE(w && uc, "null vector");
    launch_three_term<float>(n, (float*)w, (const float*)up, (const float*)uc, beta, alpha, ctx->stream);
  });
}
int ll_orth_block_c(ll_context* ctx, int64_t n, int64_t nb, const void* basis, int64_t ld, void* w, int mode,
                    double* norm_out, double* h_out) {
  return guarded([&] { orth_impl<cf>(ctx, n, nb, (const cf*)basis, ld, (cf*)w, mode, norm_out, h_out); });
}
int ll_orth_block_s(ll_context* ctx, int64_t n, int64_t nb, const float* basis, int64_t ld, float* w, int mode,
                    double* norm_out, double* h_out) {
  return guarded([&] { orth_impl<float>(ctx, n, nb, (const float*)basis, ld, (float*)w, mode, norm_out, h_out); });
}
int ll_gemv_basis_c(ll_context* ctx, int64_t n, int64_t m, const void* basis, int64_t ld, int64_t nout,
                    const double* coeff, void* out, int64_t ld_out) {
  return guarded([&] {  // coefficients arrive as doubles (re,im pairs) like every scalar of the _s/_c API
    LL_REQUIRE(coeff != nullptr && m >= 1 && nout >= 1, "bad argument");
    std::vector<cf> cc((size_t)(nout * m));
    for (size_t i = 0; i < cc.size(); ++i) cc[i] = cf{(float)coeff[2 * i], (float)coeff[2 * i + 1]};
    gemv_impl<cf>(ctx, n, m, (const cf*)basis, ld, nout, cc.data(), (cf*)out, ld_out);
  });
}
int ll_gemv_basis_s(ll_context* ctx, int64_t n, int64_t m, const float* basis, int64_t ld, int64_t nout,
                    const double* coeff, float* out, int64_t ld_out) {
  return guarded([&] {
    LL_REQUIRE(coeff != nullptr && m >= 1 && nout >= 1, "bad argument");
    std::vector<float> cc((size_t)(nout * m));
    for (size_t i = 0; i < cc.size(); ++i) cc[i] = (float)coeff[i];
    gemv_impl<float>(ctx, n, m, (const float*)basis, ld, nout, cc.data(), (float*)out, ld_out);
  });
}
int ll_lanczos_run_c(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, double* eigvals, void* eigvecs,
                     int64_t* n_found, int64_t* iter_counts, int64_t iter_cap, double* alpha_out, double* beta_out,
                     ll_run_stats* stats) {
  return guarded([&] {
    LL_REQUIRE(ctx && p && eigvals && n_found, "null argument");
    lanczos_run<cf>(ctx, op, *p, eigvals, (cf*)eigvecs, n_found, iter_counts, iter_cap, alpha_out, beta_out, stats);
  });
}
int ll_lanczos_run_s(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, double* eigvals, float* eigvecs,
                     int64_t* n_found, int64_t* iter_counts, int64_t iter_cap, double* alpha_out, double* beta_out,
                     ll_run_stats* stats) {
  return guarded([&] {
    LL_REQUIRE(ctx && p && eigvals && n_found, "null argument");
    lanczos_run<float>(ctx, op, *p, eigvals, (float*)eigvecs, n_found, iter_counts, iter_cap, alpha_out, beta_out, stats);
  });
}
int ll_lanczos_run_iteration_s(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, int64_t nroot,
                               int64_t n_orth, const float* orth, double* eigvals, float* eigvecs, int64_t* n_found,
                               int64_t* itern, double* alpha_out, double* beta_out, ll_run_stats* stats) {
  return guarded([&] {
    run_iteration_impl<float>(ctx, op, p, nroot, n_orth, orth, eigvals, eigvecs, n_found, itern, alpha_out, beta_out, stats);
  });
}
int ll_lanczos_run_iteration_c(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, int64_t nroot,
                               int64_t n_orth, const void* orth, double* eigvals, void* eigvecs, int64_t* n_found,
                               int64_t* itern, double* alpha_out, double* beta_out, ll_run_stats* stats) {
  return guarded([&] {
    run_iteration_impl<cf>(ctx, op, p, nroot, n_orth, orth, eigvals, eigvecs, n_found, itern, alpha_out, beta_out, stats);
  });
}
int ll_expo_run_c(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a_re, double a_im,
                  const void* input, void* output, int64_t* itern, ll_run_stats* stats) {
  return guarded([&] {
    LL_REQUIRE(ctx && p && input && output && itern, "null argument");
    expo_run<cf>(ctx, op, *p, std::complex<double>(a_re, a_im), (const cf*)input, (cf*)output, itern, stats);
  });
}
int ll_expo_taylor_run_c(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a_re, double a_im,
                         const void* input, void* output, int64_t* nterms) {
  return guarded([&] {
    LL_REQUIRE(ctx && p && input && output && nterms, "null argument");
    taylor_run<cf>(ctx, op, *p, std::complex<double>(a_re, a_im), (const cf*)input, (cf*)output, nterms);
  });
}
int ll_expo_run_s(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a, const float* input,
                  float* output, int64_t* itern, ll_run_stats* stats) {
  return guarded([&] {
    LL_REQUIRE(ctx && p && input && output && itern, "null argument");
    expo_run<float>(ctx, op, *p, a, input, output, itern, stats);
  });
}
int ll_expo_taylor_run_s(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a, const float* input,
                         float* output, int64_t* nterms) {
  return guarded([&] {
    LL_REQUIRE(ctx && p && input && output && nterms, "null argument");
    taylor_run<float>(ctx, op, *p, a, input, output, nterms);
  });
}

}  // extern "C"
