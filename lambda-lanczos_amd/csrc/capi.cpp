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

// ---------------------------------------------------------------- tuning (ll_internal.hpp: ll::Tuning)
// ONE parser for every setting, by key.  The library reads the USER-FACING switches from the environment (kEnvSwitches:
// the list of INTEGRATION.md section 8) when a context is created; everything else — block geometries, forced code paths, the
// hooks the test suite needs — is reachable only through ll_ctx_set_tuning(ctx, key, value), an explicit call on one context
// that is documented as unstable: a stray variable in a user's environment cannot change the numerics path of a drop-in.
namespace ll {
namespace {
long long to_ll(const std::string& v) { return std::atoll(v.c_str()); }
bool to_flag(const std::string& v) { return std::atoi(v.c_str()) != 0; }
}  // namespace
bool tuning_apply(Tuning& t, const std::string& key, const std::string& v) {
  const Tuning d;  // defaults (an empty value restores the default of its key)
  const bool e = v.empty();
  // ---- user-facing (also read from the environment, kEnvSwitches below)
  if (key == "spmv_kernel") t.spmv_kernel = v == "csr" ? 1 : (v == "pb" ? 2 : (v == "tiled" ? 3 : 0));
  else if (key == "spmv_keep_both") t.keep_both = e ? d.keep_both : to_flag(v);
  else if (key == "pb_phase2") t.pb_phase2 = v == "atomic" ? LL_PB_ATOMIC : (v == "ordered" ? LL_PB_ORDERED : LL_PB_FIXED);
  else if (key == "pb_placements") t.pb_placements = e ? d.pb_placements : (int)std::max<long long>(1, std::min<long long>(16, to_ll(v)));
  else if (key == "pb_placement_trace") t.pb_placement_trace = e ? false : to_flag(v);
  else if (key == "comm_overlap") t.comm_overlap = e ? d.comm_overlap : to_flag(v);
  else if (key == "gather_chunks") t.gather_chunks = e ? 0 : (int)std::max<long long>(0, to_ll(v));
  else if (key == "csr_split") t.csr_split = e ? d.csr_split : to_flag(v);
  else if (key == "iter_trace") t.iter_trace = v;
  else if (key == "sharded_norm") t.sharded_norm_measured = v == "measured";
  else if (key == "pair_gs") t.pair_gs = e ? d.pair_gs : to_flag(v);
  else if (key == "pb_diag") t.pb_diag = e ? d.pb_diag : to_flag(v);
  else if (key == "fuse_launches") {
    const long long level = e ? 2 : to_ll(v);
    t.fuse_launches = level >= 1;
    t.lagged_gs = level >= 2;
  } else if (key == "blas_small_bytes") t.blas_small_bytes = e ? d.blas_small_bytes : to_ll(v);
  else if (key == "tridiag_thread") t.tridiag_thread = e ? d.tridiag_thread : to_flag(v);
  else if (key == "tridiag_lag") t.tridiag_lag = e ? d.tridiag_lag : (int)to_ll(v);
  else if (key == "dgks_threshold") t.dgks_threshold = e ? d.dgks_threshold : std::atof(v.c_str());
  else if (key == "slab_bytes") t.slab_bytes = e ? d.slab_bytes : std::max<long long>(1, to_ll(v));
  // ---- unstable: ll_ctx_set_tuning only (tests, tools/ probes, A/B measurements)
  else if (key == "pb_block") t.pb_block = e ? 0 : (int)std::max<long long>(0, to_ll(v));
  else if (key == "pb_row_block") t.pb_row_block = e ? 0 : (int)std::max<long long>(0, to_ll(v));
  else if (key == "pb_col_block") t.pb_col_block = e ? 0 : (int)std::max<long long>(0, to_ll(v));
  else if (key == "pb_threads1") {
    t.pb_threads1 = e ? 0 : (int)to_ll(v);
    if (t.pb_threads1 != 256 && t.pb_threads1 != 512 && t.pb_threads1 != 1024) t.pb_threads1 = 0;
  } else if (key == "pb_pad") {
    t.pb_pad = e ? 0 : (int)to_ll(v);
    if (t.pb_pad != 4 && t.pb_pad != 16) t.pb_pad = 0;
  } else if (key == "pb_xpre") t.pb_xpre = e ? d.pb_xpre : to_flag(v);
  else if (key == "pb_test_all_remote") t.pb_test_all_remote = e ? false : to_flag(v);
  else if (key == "force_rp64") t.force_rp64 = e ? false : to_flag(v);
  else if (key == "spmv_tile_balance") t.spmv_tile_balance = e ? d.spmv_tile_balance : to_flag(v);
  else if (key == "stencil_vec") t.stencil_vec = e ? d.stencil_vec : to_flag(v);
  else if (key == "tl_force") t.tl_force = e ? false : to_flag(v);
  else if (key == "tl_xcd") t.tl_xcd_order = e ? d.tl_xcd_order : to_flag(v);
  else if (key == "tl_walk") t.tl_walk_modulo = e ? d.tl_walk_modulo : to_flag(v);
  else if (key == "ritz_tail") t.ritz_tail = e ? d.ritz_tail : to_flag(v);
  else if (key == "event_in_launch") t.event_in_launch = e ? d.event_in_launch : to_flag(v);
  else if (key == "sweep_pipeline") t.sweep_pipeline = e ? d.sweep_pipeline : (int)std::max<long long>(0, std::min<long long>(2, to_ll(v)));
  else if (key == "pair_split") t.pair_split_vecs = e ? 0 : (int)std::max<long long>(0, to_ll(v));
  else if (key == "pair_max_stored") t.pair_max_stored = e ? 0 : (int)std::max<long long>(0, to_ll(v));
  else if (key == "lagged_pieces") t.lagged_pieces = e ? 0 : (int)to_ll(v);
  else if (key == "lagged_min_bytes") t.lagged_min_bytes = e ? -1 : to_ll(v);
  else if (key == "tridiag_test_jitter_us") t.tridiag_test_jitter_us = e ? 0 : (int)to_ll(v);
  else if (key == "stall_trace") t.stall_trace_ms = e ? -1.0 : std::atof(v.c_str());
  else return false;
  return true;
}
// environment variable -> key: the switches a user may set.  LL_COMM_PLUGIN and LL_ROCTX are read
// where they are used (comm.cpp, trace.hpp), once per communicator / process.
static const char* const kEnvSwitches[][2] = {
    {"LL_SPMV_KERNEL", "spmv_kernel"},       {"LL_SPMV_KEEP_BOTH", "spmv_keep_both"}, {"LL_PB_PHASE2", "pb_phase2"},
    {"LL_PB_PLACEMENTS", "pb_placements"},   {"LL_PB_PLACEMENT_TRACE", "pb_placement_trace"},
    {"LL_COMM_OVERLAP", "comm_overlap"},     {"LL_GATHER_CHUNKS", "gather_chunks"},   {"LL_CSR_SPLIT", "csr_split"},
    {"LL_ITER_TRACE", "iter_trace"},         {"LL_SHARDED_NORM", "sharded_norm"},     {"LL_PAIR_GS", "pair_gs"},
    {"LL_PB_DIAG", "pb_diag"},               {"LL_FUSE_LAUNCHES", "fuse_launches"},   {"LL_BLAS_SMALL_BYTES", "blas_small_bytes"},
    {"LL_TRIDIAG_THREAD", "tridiag_thread"}, {"LL_TRIDIAG_LAG", "tridiag_lag"},       {"LL_DGKS_THRESHOLD", "dgks_threshold"},
    {"LL_SLAB_BYTES", "slab_bytes"},
};
Tuning read_tuning(const std::map<std::string, std::string>* overrides) {
  Tuning t;
  for (auto& sw : kEnvSwitches) {
    const char* e = std::getenv(sw[0]);
    if (e && *e) (void)tuning_apply(t, sw[1], e);
  }
  if (overrides)
    for (auto& kv : *overrides) (void)tuning_apply(t, kv.first, kv.second);
  return t;
}
}  // namespace ll

// ---------------------------------------------------------------- context workspace
static size_t grow(size_t have, size_t want) { return std::max(want, have + have / 2 + 64); }

void ll_context::dev_malloc(void** out, size_t bytes, const char* what) {
  hipError_t e = hipMalloc(out, std::max<size_t>(bytes, 16));
  if (e != hipSuccess && !slab_cache.empty()) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(stream);
    for (auto& c : slab_cache) (void)hipFree(c.first);
    slab_cache.clear();
    e = hipMalloc(out, std::max<size_t>(bytes, 16));
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error(std::string("out of device memory: ") + what + " (" + std::to_string(bytes) + " bytes): " + hipGetErrorString(e));
    throw Failure{LL_ERR_ALLOC};
  }
}
void ll_context::cache_put(void* p, size_t bytes) {
  slab_cache.emplace_back(p, bytes);
  // Bounded: over the limit, a buffer of a DIFFERENT size than the one just returned goes first (oldest of those) — a Basis
  // that returns more slabs than the bound must not push out its own first slabs, which the next run of the same problem
  // would have to allocate again (hipFree synchronises the device); only when every entry has the incoming size does the
  // oldest one go.  One pass per eviction; evictions happen at the bound only, never inside a loop.
  while (slab_cache.size() > kSlabCacheMaxEntries) {
    size_t victim = 0;
    for (size_t i = 0; i + 1 < slab_cache.size(); ++i)
      if (slab_cache[i].second != bytes) {
        victim = i;
        break;
      }
    (void)hipFree(slab_cache[victim].first);
    slab_cache.erase(slab_cache.begin() + (long)victim);
  }
}
void ll_context::ensure_partials(size_t doubles) {
  if (doubles <= partials_cap) return;
  if (d_partials) LL_HIP(hipFree(d_partials));
  d_partials = nullptr;
  partials_cap = grow(partials_cap, doubles);
  dev_malloc((void**)&d_partials, partials_cap * sizeof(double), "partial sums");
}
void ll_context::ensure_alpha_partials(size_t doubles) {
  if (doubles <= alpha_partials_cap) return;
  if (d_alpha_partials) LL_HIP(hipFree(d_alpha_partials));
  d_alpha_partials = nullptr;
  alpha_partials_cap = grow(alpha_partials_cap, doubles);
  dev_malloc((void**)&d_alpha_partials, alpha_partials_cap * sizeof(double), "alpha partial sums");
}
void ll_context::ensure_h(size_t doubles) {
  if (doubles <= h_cap) return;
  if (d_h) LL_HIP(hipFree(d_h));
  d_h = nullptr;
  h_cap = grow(h_cap, doubles);
  dev_malloc((void**)&d_h, h_cap * sizeof(double), "projection coefficients");
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
  dev_malloc(&d_coeff, coeff_cap, "Ritz coefficients");
}
void ll_context::ensure_xfull(size_t bytes) {
  if (bytes <= xfull_cap) return;
  if (d_xfull) LL_HIP(hipFree(d_xfull));
  d_xfull = nullptr;
  xfull_cap = bytes;
  dev_malloc(&d_xfull, xfull_cap, "gathered vector");
}
void ll_context::ensure_halo(size_t bytes) {
  if (bytes <= halo_cap) return;
  if (d_halo) LL_HIP(hipFree(d_halo));
  d_halo = nullptr;
  halo_cap = bytes;
  dev_malloc(&d_halo, halo_cap, "halo buffer");
}
void* ll_context::ensure_stage(size_t bytes) {
  if (bytes <= stage_cap) return h_stage;
  if (h_stage) LL_HIP(hipHostFree(h_stage));
  h_stage = nullptr;
  stage_cap = grow(stage_cap, bytes);
  LL_HIP(hipHostMalloc(&h_stage, stage_cap, hipHostMallocDefault));
  return h_stage;
}
void* ll_context::ensure_cb_stage(size_t bytes) {
  if (bytes <= cb_cap) return h_cb;
  LL_HIP(hipStreamSynchronize(stream));  // an H2D copy out of the old buffer may still be in flight
  if (h_cb) LL_HIP(hipHostFree(h_cb));
  h_cb = nullptr;
  cb_cap = grow(cb_cap, bytes);
  LL_HIP(hipHostMalloc(&h_cb, cb_cap, hipHostMallocDefault));
  return h_cb;
}
void ll_context::sync() { LL_HIP(hipStreamSynchronize(stream)); }
void ll_context::drain_comm_events(double* gather_s, double* allreduce_s) {
  auto drain = [](std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, double* acc) {
    for (auto& p : v) {
      float ms = 0.f;
      if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess && acc)
        *acc += ms * 1e-3;
      (void)hipEventDestroy(p.first);
      (void)hipEventDestroy(p.second);
    }
    v.clear();
  };
  drain(ev_gather, gather_s);
  drain(ev_allreduce, allreduce_s);
  (void)hipGetLastError();
}

// ---------------------------------------------------------------- operator storage
ll_operator::~ll_operator() {
  if (ctx) (void)hipSetDevice(ctx->device);
  for (void* q : {d_row_ptr, (void*)d_tile_rows, d_dense, d_onsite, (void*)d_pb_segq, (void*)d_pb_segdest,
                  (void*)d_pb_rptr, (void*)d_pb_xoff, (void*)d_pb_ncols, d_pb_arena, (void*)d_pb_rexp, d_pb_diag,
                  (void*)d_pb_blockmax, d_rp_own, d_rp_rem, (void*)d_col_own, (void*)d_col_rem, d_val_own, d_val_rem,
                  (void*)d_tiles_own, (void*)d_tiles_rem, (void*)d_tl_first, (void*)d_tl_col, (void*)d_tl_quad, d_tl_val,
                  (void*)d_tl_idx, (void*)d_tl_rexp, (void*)d_tl_xmax, (void*)d_tl_rbmap})
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
int ll_abi_check(int caller_major, int caller_minor, size_t sizeof_run_stats, size_t sizeof_lanczos_params) {
  // minors 3 -> 4 only added entry points: a caller compiled against any of them sees the same structs
  if (caller_major == LL_VERSION_MAJOR && caller_minor >= 3 && caller_minor <= LL_VERSION_MINOR && sizeof_run_stats == sizeof(ll_run_stats) &&
      sizeof_lanczos_params == sizeof(ll_lanczos_params))
    return LL_OK;
  set_error("ABI mismatch: the caller was compiled against lanczos_hip.h " + std::to_string(caller_major) + "." +
            std::to_string(caller_minor) + " (ll_run_stats " + std::to_string(sizeof_run_stats) + " B, ll_lanczos_params " +
            std::to_string(sizeof_lanczos_params) + " B), the loaded library is " + std::to_string(LL_VERSION_MAJOR) + "." +
            std::to_string(LL_VERSION_MINOR) + " (" + std::to_string(sizeof(ll_run_stats)) + " / " +
            std::to_string(sizeof(ll_lanczos_params)) + " B): rebuild the caller");
  return LL_ERR_INVALID;
}

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
    c->tune = read_tuning(nullptr);
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
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
    ctx->drain_comm_events(nullptr, nullptr);
    comm_destroy(ctx->comm);
    if (ctx->ev_x_ready) (void)hipEventDestroy(ctx->ev_x_ready);
    for (auto e : ctx->ev_chunk)
      if (e) (void)hipEventDestroy(e);
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    if (ctx->d_partials) (void)hipFree(ctx->d_partials);
    if (ctx->d_alpha_partials) (void)hipFree(ctx->d_alpha_partials);
    if (ctx->d_h) (void)hipFree(ctx->d_h);
    if (ctx->d_scal) (void)hipFree(ctx->d_scal);
    if (ctx->d_norm_partials) (void)hipFree(ctx->d_norm_partials);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->h_cb) (void)hipHostFree(ctx->h_cb);
    if (ctx->ev_cb) (void)hipEventDestroy(ctx->ev_cb);
    if (ctx->d_coeff) (void)hipFree(ctx->d_coeff);
    if (ctx->d_xfull) (void)hipFree(ctx->d_xfull);
    if (ctx->d_halo) (void)hipFree(ctx->d_halo);
    for (auto& c : ctx->slab_cache) (void)hipFree(c.first);
    for (auto e : ctx->timer_events) (void)hipEventDestroy(e);
    if (ctx->t0) (void)hipEventDestroy(ctx->t0);
    if (ctx->t1) (void)hipEventDestroy(ctx->t1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
  });
}
int ll_ctx_reload_env(ll_context* ctx) {
  return guarded([&] {
    LL_REQUIRE(ctx != nullptr, "null context");
    ctx->tune = read_tuning(&ctx->tuning_overrides);
  });
}
int ll_ctx_set_tuning(ll_context* ctx, const char* key, const char* value) {
  return guarded([&] {
    LL_REQUIRE(ctx != nullptr && key != nullptr, "null argument");
    Tuning probe;
    LL_REQUIRE(tuning_apply(probe, key, value ? value : ""), std::string("ll_ctx_set_tuning: unknown key '") + key + "'");
    if (value) ctx->tuning_overrides[key] = value;
    else ctx->tuning_overrides.erase(key);
    ctx->tune = read_tuning(&ctx->tuning_overrides);
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
    if (!ctx->profiling) ctx->drain_comm_events(nullptr, nullptr);
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
int ll_bandwidth_probe(ll_context* ctx, size_t bytes, double* read_GBps, double* copy_GBps) {
  return guarded([&] {
    use(ctx);
    LL_REQUIRE(bytes >= ((size_t)1 << 20) && read_GBps && copy_GBps, "ll_bandwidth_probe: at least 1 MiB and two outputs");
    bytes &= ~(size_t)4095;
    struct Buf {
      void *a = nullptr, *b = nullptr;
      double* out = nullptr;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      ~Buf() {
        if (a) (void)hipFree(a);
        if (b) (void)hipFree(b);
        if (out) (void)hipFree(out);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
      }
    } w;
    hipStream_t s = ctx->stream;
    ctx->dev_malloc(&w.a, bytes, "bandwidth probe (source)");
    ctx->dev_malloc(&w.b, bytes, "bandwidth probe (destination)");
    ctx->dev_malloc((void**)&w.out, 16, "bandwidth probe (sink)");
    LL_HIP(hipMemsetAsync(w.a, 0, bytes, s));
    LL_HIP(hipMemsetAsync(w.b, 0, bytes, s));
    LL_HIP(hipEventCreate(&w.e0));
    LL_HIP(hipEventCreate(&w.e1));
    auto timed = [&](auto launch) {  // best grid of a few, three launches each behind one warm-up
      double best = 1e30;
      for (int grid : {512, 1024, 2048, 8192}) {
        launch(grid);
        LL_HIP(hipEventRecord(w.e0, s));
        for (int r = 0; r < 3; ++r) launch(grid);
        LL_HIP(hipEventRecord(w.e1, s));
        LL_HIP(hipEventSynchronize(w.e1));
        float ms = 0.f;
        LL_HIP(hipEventElapsedTime(&ms, w.e0, w.e1));
        best = std::min(best, (double)ms / 3.0);
      }
      return best;
    };
    const double ms_r = timed([&](int g) { launch_bw_read(w.a, bytes, w.out, g, s); });
    const double ms_c = timed([&](int g) { launch_bw_copy(w.a, w.b, bytes, g, s); });
    *read_GBps = (double)bytes / (ms_r * 1e-3) / 1e9;
    *copy_GBps = 2.0 * (double)bytes / (ms_c * 1e-3) / 1e9;  // bytes read + bytes written
  });
}

// ---------------------------------------------------------------- multi-GPU
int ll_comm_unique_id(void* id) {
  return guarded([&] {
    LL_REQUIRE(id != nullptr, "null id buffer");
    comm_unique_id(id);
  });
}
extern "C++" {
namespace {
// After the communicator exists: the second stream + events of the overlapped exchange, and a SELF-CHECK — every rank
// contributes (rank + 1) to an all-gather and the constant 1 to an all-reduce; a communicator that silently spans
// fewer ranks than asked for (or delivers shards in another order) fails here instead of producing a wrong spectrum.
void finish_comm_setup_impl(ll_context* ctx) {
  LL_HIP(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
  LL_HIP(hipEventCreateWithFlags(&ctx->ev_x_ready, hipEventDisableTiming));
  for (auto& e : ctx->ev_chunk) LL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const int P = ctx->nranks;
  double* d = nullptr;
  LL_HIP(hipMalloc((void**)&d, (size_t)(P + 2) * sizeof(double)));
  struct Free {
    void* p;
    ~Free() { (void)hipFree(p); }
  } guard{d};
  std::vector<double> h((size_t)P + 2, 0.0);
  h[(size_t)P] = (double)(ctx->rank + 1);  // send slot
  h[(size_t)P + 1] = 1.0;                  // all-reduce slot
  LL_HIP(hipMemcpyAsync(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  // the gather runs on the communication stream, the reduction on the compute stream: the two-stream order of the loop
  LL_HIP(hipEventRecord(ctx->ev_x_ready, ctx->stream));
  LL_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_x_ready, 0));
  comm_allgather(ctx->comm, d + P, d, sizeof(double), ctx->comm_stream);
  LL_HIP(hipEventRecord(ctx->ev_chunk[0], ctx->comm_stream));
  LL_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_chunk[0], 0));
  comm_allreduce_sum(ctx->comm, d + P + 1, 1, ctx->stream);
  LL_HIP(hipMemcpyAsync(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  LL_HIP(hipStreamSynchronize(ctx->stream));
  int seen = 0;
  for (int r = 0; r < P; ++r)
    if (h[(size_t)r] == (double)(r + 1)) ++seen;
  ctx->ranks_seen = seen;
  if (seen != P || h[(size_t)P + 1] != (double)P) {
    set_error("communicator self-check failed: all-gather delivered " + std::to_string(seen) + " of " + std::to_string(P) +
              " rank tags, all-reduce of ones gave " + std::to_string(h[(size_t)P + 1]));
    throw Failure{LL_ERR_RCCL};
  }
}
// A communicator whose set-up or self-check failed must not stay attached: the context would look sharded with a
// transport known to be broken (later operators would be created as shards, their collectives could hang, and a retry
// of ll_comm_init / ll_comm_attach would be refused).  Everything is undone and the error is passed on.
void finish_comm_setup(ll_context* ctx) {
  try {
    finish_comm_setup_impl(ctx);
  } catch (...) {
    (void)hipGetLastError();
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
    (void)hipStreamSynchronize(ctx->stream);
    comm_destroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->rank = 0;
    ctx->nranks = 1;
    ctx->ranks_seen = 0;
    if (ctx->ev_x_ready) (void)hipEventDestroy(ctx->ev_x_ready);
    ctx->ev_x_ready = nullptr;
    for (auto& e : ctx->ev_chunk) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    ctx->comm_stream = nullptr;
    (void)hipGetLastError();
    throw;
  }
}
}  // namespace
}  // extern "C++"

int ll_comm_init(ll_context* ctx, const void* id, int rank, int n_ranks) {
  return guarded([&] {
    use(ctx);
    LL_REQUIRE(id != nullptr, "null id");
    LL_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "rank out of range");
    LL_REQUIRE(ctx->comm == nullptr, "communicator already attached");
    ctx->comm = comm_create(id, rank, n_ranks, ctx->device);
    ctx->rank = rank;
    ctx->nranks = n_ranks;
    finish_comm_setup(ctx);
  });
}
int ll_comm_attach(ll_context* ctx, const ll_transport* transport, int rank, int n_ranks) {
  return guarded([&] {
    use(ctx);
    LL_REQUIRE(transport && transport->all_gather && transport->all_reduce_sum_f64 && transport->halo_exchange,
               "incomplete transport table");
    LL_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "rank out of range");
    LL_REQUIRE(ctx->comm == nullptr, "communicator already attached");
    ctx->comm = comm_attach(transport, rank, n_ranks);
    ctx->rank = rank;
    ctx->nranks = n_ranks;
    finish_comm_setup(ctx);
  });
}
int ll_comm_ranks_seen(ll_context* ctx, int* out) {
  return guarded([&] {
    LL_REQUIRE(ctx != nullptr && out != nullptr, "null argument");
    *out = ctx->comm ? ctx->ranks_seen : 1;
  });
}
int ll_comm_transport(ll_context* ctx, char* out, size_t cap) {
  return guarded([&] {
    LL_REQUIRE(ctx != nullptr && out != nullptr && cap > 0, "null argument");
    const std::string name = comm_transport_name(ctx->comm);
    std::snprintf(out, cap, "%s", name.c_str());
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

// SpMV tiles: runs of whole rows with <= cap nonzeros and <= kBlock rows; a longer row is alone.
void build_tiles_cap(const int64_t* rp, int64_t nrows, int64_t cap, std::vector<int32_t>& tiles) {
  tiles.clear();
  tiles.push_back(0);
  int64_t r = 0;
  while (r < nrows) {
    int64_t r1 = r;
    while (r1 < nrows && (r1 - r) < kBlock && rp[r1 + 1] - rp[r] <= cap) ++r1;
    if (r1 == r) r1 = r + 1;
    tiles.push_back((int32_t)r1);
    r = r1;
  }
}
// The kernel walks the tiles with a persistent grid of at most `grid_cap` workgroups, every workgroup the same number of
// tiles +-1 (TileWalk).  With only a few tiles per workgroup that +-1 is a large share of the kernel: config 2 (4 880
// tiles of 1 024 nonzeros on 2 048 workgroups) runs three rounds of which the last is 38 % full.  So when fewer than
// eight rounds are needed the tile size is lowered until the tiles fill whole rounds: every workgroup then walks exactly
// `rounds` tiles, each a little shorter: config 2's SpMV 18.35 -> 17.37 us (54.5 -> 57.5 % of the roofline).  Only from three
// rounds up: a tile costs mostly latency, so with one or two rounds (config 5: 4 883 tiles on 4 096 workgroups) a few
// workgroups walking a second full tile are cheaper than all of them walking two shorter ones (25.0 -> 30.6 us when
// balanced; gpurun A/B of round 4).  (LL_SPMV_TILE_BALANCE=0: always kSpmvTileNnz.)
void build_tiles(const int64_t* rp, int64_t nrows, std::vector<int32_t>& tiles, int grid_cap = 0, bool balance = true) {
  build_tiles_cap(rp, nrows, kSpmvTileNnz, tiles);
  const int64_t nt = (int64_t)tiles.size() - 1;
  if (!balance || grid_cap <= 0 || nt <= grid_cap / 2 || nt >= 8 * (int64_t)grid_cap) return;
  const int64_t rounds = (nt + grid_cap - 1) / grid_cap;
  if (rounds < 3) return;
  const int64_t nnz = rp[nrows];
  std::vector<int32_t> best;
  // rows do not cut evenly: shrink the cap until the tile count fits rounds x grid (a few tries)
  for (double slack : {0.995, 0.97, 0.94, 0.90}) {
    const int64_t cap = std::max<int64_t>(64, std::min<int64_t>(kSpmvTileNnz, (int64_t)((double)nnz / ((double)rounds * grid_cap * slack)) + 1));
    std::vector<int32_t> t;
    build_tiles_cap(rp, nrows, cap, t);
    if ((int64_t)t.size() - 1 <= rounds * grid_cap) {
      tiles.swap(t);
      return;
    }
  }
}

template <typename T>
void finish_csr(ll_operator* op, const int64_t* rp_host) {
  ll_context* ctx = op->ctx;
  const int64_t nr = op->n_local;
  std::vector<int32_t> tiles;
  build_tiles(rp_host, nr, tiles, sizeof(T) >= 16 ? kMaxSpmvGrid : kMaxGrid, ctx->tune.spmv_tile_balance);
  op->ntiles = (int)tiles.size() - 1;
  ctx->dev_malloc((void**)&op->d_tile_rows, tiles.size() * sizeof(int32_t), "SpMV tiles");
  LL_HIP(hipMemcpy(op->d_tile_rows, tiles.data(), tiles.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  // 64-bit row offsets once nnz exceeds int32 (LL_FORCE_RP64=1: exercise that kernel variant on small test matrices)
  op->rp64 = op->nnz > (int64_t)0x7fffffff || ctx->tune.force_rp64;
  if (op->rp64) {
    ctx->dev_malloc(&op->d_row_ptr, (size_t)(nr + 1) * sizeof(int64_t), "row offsets");
    LL_HIP(hipMemcpy(op->d_row_ptr, rp_host, (size_t)(nr + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  } else {
    std::vector<int32_t> rp32((size_t)nr + 1);
    for (int64_t i = 0; i <= nr; ++i) rp32[i] = (int32_t)rp_host[i];
    ctx->dev_malloc(&op->d_row_ptr, (size_t)(nr + 1) * sizeof(int32_t), "row offsets");
    LL_HIP(hipMemcpy(op->d_row_ptr, rp32.data(), (size_t)(nr + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  (void)ctx;
}

// Sharded contexts that keep the CSR-stream kernel: split the image by column ownership so that the own-column product
// runs under the all-gather (the PB image has its own own / remote block ranges).  Built on the device from
// the CSR arrays (they may never have been on the host); one int32 per row crosses the bus for the prefix sums.
template <typename T> void build_csr_split(ll_operator* op) {
  ll_context* ctx = op->ctx;
  hipStream_t s = ctx->stream;
  const int64_t nr = op->n_local;
  if (nr <= 0 || op->d_row_ptr == nullptr) return;
  int32_t* d_cnt = nullptr;
  ctx->dev_malloc((void**)&d_cnt, (size_t)nr * sizeof(int32_t), "own-column counts");
  struct Free1 {
    void* p;
    ~Free1() { (void)hipFree(p); }
  } free_cnt{d_cnt};
  launch_csr_count_own<T>(*op, d_cnt, s);
  std::vector<int32_t> cnt((size_t)nr);
  std::vector<int64_t> rp((size_t)nr + 1);
  LL_HIP(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)nr * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (op->rp64) {
    LL_HIP(hipMemcpyAsync(rp.data(), op->d_row_ptr, (size_t)(nr + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    LL_HIP(hipStreamSynchronize(s));
  } else {
    std::vector<int32_t> rp32((size_t)nr + 1);
    LL_HIP(hipMemcpyAsync(rp32.data(), op->d_row_ptr, (size_t)(nr + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    LL_HIP(hipStreamSynchronize(s));
    for (int64_t i = 0; i <= nr; ++i) rp[(size_t)i] = rp32[(size_t)i];
  }
  std::vector<int64_t> rp_own((size_t)nr + 1), rp_rem((size_t)nr + 1);
  rp_own[0] = rp_rem[0] = 0;
  for (int64_t i = 0; i < nr; ++i) {
    rp_own[(size_t)i + 1] = rp_own[(size_t)i] + cnt[(size_t)i];
    rp_rem[(size_t)i + 1] = rp_rem[(size_t)i] + (rp[(size_t)i + 1] - rp[(size_t)i] - cnt[(size_t)i]);
  }
  auto upload_rp = [&](const std::vector<int64_t>& v, void** dst) {
    if (op->rp64) {
      ctx->dev_malloc(dst, v.size() * sizeof(int64_t), "split row offsets");
      LL_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    } else {
      std::vector<int32_t> v32(v.size());
      for (size_t i = 0; i < v.size(); ++i) v32[i] = (int32_t)v[i];
      ctx->dev_malloc(dst, v32.size() * sizeof(int32_t), "split row offsets");
      LL_HIP(hipMemcpy(*dst, v32.data(), v32.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
  };
  auto upload_tiles = [&](const std::vector<int64_t>& v, int32_t** dst, int* ntiles) {
    std::vector<int32_t> tiles;
    build_tiles(v.data(), nr, tiles, sizeof(T) >= 16 ? kMaxSpmvGrid : kMaxGrid, ctx->tune.spmv_tile_balance);
    *ntiles = (int)tiles.size() - 1;
    ctx->dev_malloc((void**)dst, tiles.size() * sizeof(int32_t), "split SpMV tiles");
    LL_HIP(hipMemcpy(*dst, tiles.data(), tiles.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  };
  upload_rp(rp_own, &op->d_rp_own);
  upload_rp(rp_rem, &op->d_rp_rem);
  upload_tiles(rp_own, &op->d_tiles_own, &op->ntiles_own);
  upload_tiles(rp_rem, &op->d_tiles_rem, &op->ntiles_rem);
  const size_t n_own = (size_t)rp_own[(size_t)nr], n_rem = (size_t)rp_rem[(size_t)nr];
  ctx->dev_malloc((void**)&op->d_col_own, std::max<size_t>(n_own, 1) * sizeof(int32_t), "own-column indices");
  ctx->dev_malloc(&op->d_val_own, std::max<size_t>(n_own, 1) * sizeof(T), "own-column values");
  ctx->dev_malloc((void**)&op->d_col_rem, std::max<size_t>(n_rem, 1) * sizeof(int32_t), "remote-column indices");
  ctx->dev_malloc(&op->d_val_rem, std::max<size_t>(n_rem, 1) * sizeof(T), "remote-column values");
  launch_csr_split<T>(*op, s);
  LL_HIP(hipStreamSynchronize(s));
  op->csr_split = true;
}

// Undo a (possibly partial) column split: the operator runs gather-then-multiply on its unsplit image.
void release_csr_split(ll_operator* op) {
  auto drop = [](auto*& p) {
    if (p) (void)hipFree((void*)p);
    p = nullptr;
  };
  drop(op->d_rp_own);
  drop(op->d_rp_rem);
  drop(op->d_col_own);
  drop(op->d_col_rem);
  drop(op->d_val_own);
  drop(op->d_val_rem);
  drop(op->d_tiles_own);
  drop(op->d_tiles_rem);
  op->ntiles_own = op->ntiles_rem = 0;
  op->csr_split = false;
}
// The unsplit CSR arrays of an operator that runs on its column-split image (the row offsets stay: 4 bytes per row, and
// they mark the operator as one that still has a CSR-stream image).
void release_unsplit_csr(ll_operator* op) {
  auto drop = [](auto*& p) {
    if (p) (void)hipFree((void*)p);
    p = nullptr;
  };
  if (op->owns_arrays) {
    drop(op->d_col);
    drop(op->d_val);
  } else {
    op->d_col = nullptr;  // the caller's arrays: just forget them
    op->d_val = nullptr;
  }
  drop(op->d_tile_rows);
  op->ntiles = 0;
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

// Drop the device arrays of the SpMV image that is NOT selected (LL_SPMV_KEEP_BOTH=1 keeps both for A/B timing).
void release_image(ll_operator* op, int keep_kind);
void release_unselected_image(ll_operator* op) {
  if (op->ctx->tune.keep_both) return;
  release_image(op, op->spmv_kind);
}
void release_image(ll_operator* op, int keep_kind) {
  auto drop = [](auto*& p) {
    if (p) (void)hipFree((void*)p);
    p = nullptr;
  };
  if (keep_kind != LL_SPMV_CSR_STREAM) {  // CSR-stream needs row_ptr / col / val / tiles; the other kernels need none of them
    if (op->owns_arrays) {
      drop(op->d_col);
      drop(op->d_val);
    } else {
      op->d_col = nullptr;  // the caller's arrays: just forget them
      op->d_val = nullptr;
    }
    drop(op->d_row_ptr);
    drop(op->d_tile_rows);
    op->ntiles = 0;
  }
  if (keep_kind != LL_SPMV_PB) {
    drop(op->d_pb_segq);
    drop(op->d_pb_segdest);
    drop(op->d_pb_rptr);
    drop(op->d_pb_xoff);
    drop(op->d_pb_ncols);
    drop(op->d_pb_arena);
    drop(op->d_pb_rexp);
    drop(op->d_pb_blockmax);
    drop(op->d_pb_diag);
    op->d_pb_val = op->d_pb_prod = nullptr;  // interior pointers of the arena
    op->d_pb_col = op->d_pb_row = nullptr;
    op->pb_ncb = op->pb_nrb = 0;
  }
  if (keep_kind != LL_SPMV_TILED) tl_release(op);
}
// the part of the PB image that is allocated so far (a failed or refused build)
void release_pb_image(ll_operator* op) {
  auto drop = [](auto*& p) {
    if (p) (void)hipFree((void*)p);
    p = nullptr;
  };
  drop(op->d_pb_segq);
  drop(op->d_pb_segdest);
  drop(op->d_pb_rptr);
  drop(op->d_pb_xoff);
  drop(op->d_pb_ncols);
  drop(op->d_pb_arena);
  drop(op->d_pb_rexp);
  drop(op->d_pb_blockmax);
  drop(op->d_pb_diag);
  op->d_pb_val = op->d_pb_prod = nullptr;
  op->d_pb_col = op->d_pb_row = nullptr;
  op->pb_ncb = op->pb_nrb = 0;
}

// Placement of the PB image.  The same image at another address runs up to 5-8 % faster or slower (round 2: "position
// noise"; round 3, bench.py spmv.ms_by_kernel: 0.922 ms for the operator created first, 0.845 ms for one created later, same
// process, same x / y) — which HBM stacks and channels the arena's physical pages land on is the draw of the allocation,
// fixed for its life.  So the draw is repeated: the image is copied (device to device, ~1 ms per GB) into fresh
// allocations, each is timed with the real kernels, the fastest is kept and the others are freed.  Purely local: no
// collective decision depends on it.  Returns the best time (ms).
template <typename T> double tune_pb_placement(ll_operator* op) {
  ll_context* ctx = op->ctx;
  hipStream_t s = ctx->stream;
  if (op->d_pb_arena == nullptr || op->nnz < ((int64_t)1 << 22)) return -1.0;
  const size_t xn = (size_t)std::max<int64_t>(op->n, op->n_shard * std::max(1, ctx->nranks));
  struct Scratch {
    T *x = nullptr, *y = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Scratch() {
      if (x) (void)hipFree(x);
      if (y) (void)hipFree(y);
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } w;
  ctx->dev_malloc((void**)&w.x, xn * sizeof(T), "placement timing x");
  ctx->dev_malloc((void**)&w.y, (size_t)std::max<int64_t>(op->n_local, 1) * sizeof(T), "placement timing y");
  LL_HIP(hipMemsetAsync(w.x, 0, xn * sizeof(T), s));
  LL_HIP(hipEventCreate(&w.e0));
  LL_HIP(hipEventCreate(&w.e1));
  auto time_pb = [&]() {
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      LL_HIP(hipEventRecord(w.e0, s));
      launch_spmv_pb<T>(*op, w.x, w.x + op->row_begin, w.x + op->row_begin, w.y, 0.0, nullptr, s);
      LL_HIP(hipEventRecord(w.e1, s));
      LL_HIP(hipEventSynchronize(w.e1));
      float ms = 0.f;
      LL_HIP(hipEventElapsedTime(&ms, w.e0, w.e1));
      if (rep > 0) best = std::min(best, (double)ms);
    }
    return best;
  };
  auto rebase = [&](void* arena) {
    const ptrdiff_t d = (char*)arena - (char*)op->d_pb_arena;
    op->d_pb_arena = arena;
    op->d_pb_val = (char*)op->d_pb_val + d;
    op->d_pb_col = (uint16_t*)((char*)op->d_pb_col + d);
    op->d_pb_row = (uint16_t*)((char*)op->d_pb_row + d);
    op->d_pb_prod = (char*)op->d_pb_prod + d;
  };
  // Every candidate stays allocated until all have been timed (an allocation freed at once would simply be handed out
  // again for the next one); then all but the fastest are freed.
  double best = time_pb();
  if (op->ctx->tune.pb_placement_trace) std::fprintf(stderr, "[ll placement] draw 0 at %p: %.4f ms\n", op->d_pb_arena, best);
  void* best_arena = op->d_pb_arena;
  std::vector<void*> losers;
  // The candidates that lose are not returned to the device: sized like a Krylov-basis slab of a default run on this operator
  // (when that is at least the arena's size), they go into the context's slab cache and become the first basis slabs.  A process
  // that starts on a GPU another process has just left pays ~120 ms per fresh 4 GiB hipMalloc: config 3's
  // first run() to convergence needs seven slabs — the search has already paid for seven allocations.
  const size_t slab_hint = (size_t)default_slab_bytes(op->n, op->n_local, op->n_shard, op->elem_bytes, ctx->tune);
  const size_t cand_bytes = slab_hint >= op->pb_arena_bytes && slab_hint <= 2 * op->pb_arena_bytes ? slab_hint : op->pb_arena_bytes;
  void* const first_arena = op->d_pb_arena;
  bool finished = false;
  // Unwinding (a failed copy or launch, thrown through LL_HIP): the operator goes back to the best image found so far and
  // every other copy is freed exactly once — `losers` never contains the arena the operator is bound to at that point.
  struct Guard {
    std::vector<void*>& v;
    void*& best;
    decltype(rebase)& rb;
    ll_context* ctx;
    void* first;
    size_t cand_bytes, slab_hint;
    bool& finished;
    ~Guard() {
      rb(best);
      for (void* p : v) {
        if (p == best) continue;
        // (at most eight slabs of that size are kept this way: a context on which many operators are created must not pile up
        // a placement search's worth of HBM per operator)
        size_t same = 0;
        for (auto& c : ctx->slab_cache) same += c.second == cand_bytes;
        if (finished && p != first && cand_bytes == slab_hint && same < 8) ctx->cache_put(p, cand_bytes);
        else (void)hipFree(p);
      }
    }
  } guard{losers, best_arena, rebase, ctx, first_arena, cand_bytes, slab_hint, finished};
  // candidates come from the context's allocator: under memory pressure it releases the cached Krylov slabs once before
  // giving up, so a large matrix is not silently left with fewer draws
  for (int t = 1; t < ctx->tune.pb_placements; ++t) {
    void* cand = nullptr;
    try {
      ctx->dev_malloc(&cand, cand_bytes, "PB placement candidate");
    } catch (const Failure&) {  // no room for another copy: decide among what we have
      (void)hipGetLastError();
      break;
    }
    losers.push_back(cand);
    LL_HIP(hipMemcpyAsync(cand, best_arena, op->pb_arena_static_bytes, hipMemcpyDeviceToDevice, s));
    rebase(cand);
    const double ms = time_pb();
    if (op->ctx->tune.pb_placement_trace) std::fprintf(stderr, "[ll placement] draw %d at %p: %.4f ms (best so far %.4f)\n", t, cand, ms, best);
    if (ms < best) {
      best = ms;
      losers.back() = best_arena;  // the previous best becomes a loser
      best_arena = cand;
    }
    rebase(best_arena);
  }
  finished = true;
  return best;
}

// Time both SpMV kernels on the device with the actual matrix and keep the faster one.  Sharded contexts decide on
// the SUM of the per-rank times, so every rank runs the same kernel (the exchange plan depends on it).  A kernel
// whose launch fails is simply not a candidate.
template <typename T> void autotune_spmv(ll_operator* op) {
  ll_context* ctx = op->ctx;
  hipStream_t s = ctx->stream;
  const size_t xn = (size_t)std::max<int64_t>(op->n, op->n_shard * std::max(1, ctx->nranks));
  struct Scratch {
    T *x = nullptr, *y = nullptr;
    double* t = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Scratch() {
      if (x) (void)hipFree(x);
      if (y) (void)hipFree(y);
      if (t) (void)hipFree(t);
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } w;
  ctx->dev_malloc((void**)&w.x, xn * sizeof(T), "autotune x");
  ctx->dev_malloc((void**)&w.y, (size_t)std::max<int64_t>(op->n_local, 1) * sizeof(T), "autotune y");
  ctx->dev_malloc((void**)&w.t, 3 * sizeof(double), "autotune scalars");
  LL_HIP(hipMemsetAsync(w.x, 0, xn * sizeof(T), s));
  LL_HIP(hipEventCreate(&w.e0));
  LL_HIP(hipEventCreate(&w.e1));
  double t_kind[3] = {1e30, 1e30, 1e30};
  for (int kind : {LL_SPMV_CSR_STREAM, LL_SPMV_PB, LL_SPMV_TILED}) {
    if (kind == LL_SPMV_PB && op->d_pb_val == nullptr) continue;     // image not built: not a candidate
    if (kind == LL_SPMV_TILED && op->tl_nrb <= 0) continue;
    try {
      for (int rep = 0; rep < 3; ++rep) {
        LL_HIP(hipEventRecord(w.e0, s));
        if (kind == LL_SPMV_PB) launch_spmv_pb<T>(*op, w.x, w.x + op->row_begin, w.x + op->row_begin, w.y, 0.0, nullptr, s);
        else if (kind == LL_SPMV_TILED) launch_spmv_tiled<T>(*op, w.x, w.y, 0.0, nullptr, s);
        else launch_spmv<T>(*op, w.x, w.x + op->row_begin, w.y, 0.0, nullptr, s);
        LL_HIP(hipEventRecord(w.e1, s));
        LL_HIP(hipEventSynchronize(w.e1));
        float ms = 0.f;
        LL_HIP(hipEventElapsedTime(&ms, w.e0, w.e1));
        if (rep > 0) t_kind[kind] = std::min(t_kind[kind], (double)ms);
      }
    } catch (const Failure&) {  // e.g. a launch the device refuses: not a candidate, and not an error of the operator
      (void)hipGetLastError();
      t_kind[kind] = 1e30;
    }
  }
  for (int k = 0; k < 3; ++k) op->tune_ms[k] = t_kind[k] < 1e29 ? (float)t_kind[k] : -1.f;
  if (ctx->comm != nullptr) {
    LL_HIP(hipMemcpyAsync(w.t, t_kind, 3 * sizeof(double), hipMemcpyHostToDevice, s));
    comm_allreduce_sum(ctx->comm, w.t, 3, s);
    LL_HIP(hipMemcpyAsync(t_kind, w.t, 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    LL_HIP(hipStreamSynchronize(s));
  }
  op->spmv_kind = LL_SPMV_CSR_STREAM;
  if (t_kind[LL_SPMV_PB] < t_kind[op->spmv_kind]) op->spmv_kind = LL_SPMV_PB;
  if (t_kind[LL_SPMV_TILED] < t_kind[op->spmv_kind]) op->spmv_kind = LL_SPMV_TILED;
}

template <typename T>
void create_csr(ll_context* ctx, int64_t nr, int64_t nc, int64_t row_begin, const int64_t* rp, const int32_t* ci,
                const void* va, bool on_device, ll_operator** out, const ll_csr_options* opt = nullptr) {
  use(ctx);
  if (opt != nullptr) {
    LL_REQUIRE(opt->accuracy >= LL_ACCURACY_DEFAULT && opt->accuracy <= LL_ACCURACY_COMPONENTWISE, "ll_csr_options.accuracy");
    LL_REQUIRE(opt->kernel >= -1 && opt->kernel <= LL_SPMV_TILED, "ll_csr_options.kernel");
    on_device = opt->arrays_on_device != 0;
  }
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
  op->accuracy_req = opt != nullptr ? opt->accuracy : LL_ACCURACY_DEFAULT;
  set_partition(ctx, op.get(), nc, row_begin, nr);
  op->nnz = rp_host[nr];
  const size_t nnz = (size_t)op->nnz;
  if (on_device) {
    op->owns_arrays = false;
    op->d_col = const_cast<int32_t*>(ci);
    op->d_val = const_cast<void*>(va);
  } else {
    ctx->dev_malloc((void**)&op->d_col, std::max<size_t>(nnz, 1) * sizeof(int32_t), "CSR column indices");
    ctx->dev_malloc(&op->d_val, std::max<size_t>(nnz, 1) * sizeof(T), "CSR values");
    LL_HIP(hipMemcpy(op->d_col, ci, nnz * sizeof(int32_t), hipMemcpyHostToDevice));
    LL_HIP(hipMemcpy(op->d_val, va, nnz * sizeof(T), hipMemcpyHostToDevice));
  }
  finish_csr<T>(op.get(), rp_host);
  // column range check and max absolute row sum (ll_op_inf_norm; determine_eigenvalue_offset.cpp:12-29), on the device
  // for host and device inputs alike, whatever kernel gets selected
  csr_check_device<T>(op.get());
  op->spmv_kind = LL_SPMV_CSR_STREAM;
  // 0 auto, 1 csr, 2 pb (LL_SPMV_KERNEL); the caller's ll_csr_options.kernel outranks the environment
  const int want = (opt != nullptr && opt->kernel >= 0) ? opt->kernel + 1 : ctx->tune.spmv_kernel;
  // Sharded contexts take every decision below COLLECTIVELY (an empty shard, or a shard whose shape rules the image
  // out, must not leave the ranks with different kernels: the exchange plan and the collectives issued depend on it).
  auto all_ranks_agree = [&](bool mine) {
    if (ctx->comm == nullptr) return mine;
    double* d = nullptr;
    ctx->dev_malloc((void**)&d, sizeof(double), "agreement flag");
    const double v = mine ? 0.0 : 1.0;
    double sum = 0.0;
    try {
      LL_HIP(hipMemcpyAsync(d, &v, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      comm_allreduce_sum(ctx->comm, d, 1, ctx->stream);
      LL_HIP(hipMemcpyAsync(&sum, d, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      LL_HIP(hipStreamSynchronize(ctx->stream));
    } catch (...) {
      (void)hipFree(d);
      throw;
    }
    (void)hipFree(d);
    return sum == 0.0;
  };
  // 0 auto, 1 csr, 2 pb, 3 tiled.
  const bool componentwise = op->accuracy_req == LL_ACCURACY_COMPONENTWISE ||
                             (op->accuracy_req == LL_ACCURACY_DEFAULT && ctx->tune.pb_phase2 != LL_PB_FIXED);
  bool pb_ok = false, tl_ok = false;
  if (want != 1 && want != 3 && (nnz > 0 || ctx->comm != nullptr)) {
    // the propagation-blocked image is built on the device from the CSR arrays (histogram + scatter kernels)
    bool built = false;
    try {
      built = pb_build_device<T>(op.get());
    } catch (const Failure& f) {
      // The image is the matrix again plus a product buffer (peak at creation: CSR + PB + timing scratch, about 2.3 x the
      // matrix).  When it does not fit, a matrix that fits as CSR alone is still usable: keep CSR-stream (sharded
      // contexts: the peers are told below) — unless LL_SPMV_KERNEL=pb asked for this image specifically.
      const bool out_of_memory = f.code == LL_ERR_ALLOC;
      if (ctx->comm == nullptr && !(out_of_memory && want == 0)) throw;
      (void)hipGetLastError();
      built = false;
    }
    if (!built) release_pb_image(op.get());  // whatever part of the image was allocated
    pb_ok = all_ranks_agree(built);
    if (built && !pb_ok) release_pb_image(op.get());  // some rank could not build it: nobody uses it
  }
  if ((want == 0 || want == 3) && (nnz > 0 || ctx->comm != nullptr)) {
    // the 2-D tiled image: only for matrices whose row blocks touch few column tiles (tl_build_device decides).  One image serves
    // both accuracy classes: fixed-point sums (norm-wise) or the waves adding in turn in floating point (component-wise).
    op->tl_ordered = componentwise;
    bool built = false;
    try {
      built = nnz > 0 && tl_build_device<T>(op.get());
    } catch (const Failure& f) {
      if (!(f.code == LL_ERR_ALLOC && (want == 0 || ctx->comm != nullptr))) throw;
      (void)hipGetLastError();
      tl_release(op.get());
      built = false;
    }
    // (sharded contexts: the kernel choice is collective — the exchange in front of the tiled kernel carries the ranks' maxima)
    tl_ok = all_ranks_agree(built);
    if (built && !tl_ok) tl_release(op.get());
    LL_REQUIRE(!(want == 3 && !tl_ok), "this matrix is not eligible for the tiled SpMV kernel (its row blocks touch too many column tiles)");
  }
  // Asked for by name, the tiled kernel is an error where no tiled image exists (a matrix without entries) — never a silent
  // fallback, and never a kernel selected without its image (launch_spmv_tiled would write nothing).
  LL_REQUIRE(!(want == 3 && !tl_ok), "the tiled SpMV kernel was asked for by name but no tiled image was built (matrix without entries)");
  if (want == 2 && pb_ok) op->spmv_kind = LL_SPMV_PB;
  else if (want == 3 && tl_ok) op->spmv_kind = LL_SPMV_TILED;
  else if (want == 0 && (pb_ok || tl_ok)) autotune_spmv<T>(op.get());
  if (op->spmv_kind == LL_SPMV_PB && ctx->tune.pb_placements > 1) {
    const double ms = tune_pb_placement<T>(op.get());
    if (ms > 0.0 && op->tune_ms[LL_SPMV_PB] >= 0.f) op->tune_ms[LL_SPMV_PB] = (float)ms;
  }
  release_unselected_image(op.get());
  // (every rank takes this branch or none: the kernel choice above is collective, the switch comes from the environment)
  if (ctx->nranks > 1 && ctx->tune.csr_split && op->d_row_ptr != nullptr) {
    // The split image is a second copy of the matrix.  When it does not fit next to the original (a shard that already fell
    // back to CSR-stream because the PB image did not fit), the operator stays usable in the gather-then-multiply form — safe
    // per rank: split and unsplit ranks issue the same single all-gather.
    try {
      build_csr_split<T>(op.get());
    } catch (const Failure& f) {
      if (f.code != LL_ERR_ALLOC) throw;
      (void)hipGetLastError();
      release_csr_split(op.get());
    }
    // Once split, the unsplit arrays are never read again on this context: return them (steady-state footprint 1 x the matrix)
    // unless LL_SPMV_KEEP_BOTH=1 asked for every image to stay.
    if (op->csr_split && !ctx->tune.keep_both && op->spmv_kind == LL_SPMV_CSR_STREAM) release_unsplit_csr(op.get());
  }
  *out = op.release();
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
  for (int64_t i = 0; i < nr; ++i) {
    double rs = 0.0;
    for (int64_t j = 0; j < nc; ++j) rs += std::sqrt(abs2_host(v[i * nc + j]));
    mx = std::max(mx, rs);
  }
  op->inf_norm = mx;
  const size_t bytes = (size_t)nr * (size_t)nc * sizeof(T);
  ctx->dev_malloc(&op->d_dense, std::max<size_t>(bytes, 16), "dense matrix");
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
    ctx->dev_malloc(&op->d_onsite, std::max<size_t>((size_t)n_local * sizeof(R), 16), "on-site terms");
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
int ll_csr_options_default(ll_csr_options* opt) {
  return guarded([&] {
    LL_REQUIRE(opt != nullptr, "null options");
    std::memset(opt, 0, sizeof(*opt));
    opt->accuracy = LL_ACCURACY_DEFAULT;
    opt->kernel = -1;
  });
}
int ll_op_create_csr_opt_d(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const int64_t* rp, const int32_t* ci,
                           const double* va, const ll_csr_options* opt, ll_operator** out) {
  return guarded([&] { create_csr<double>(ctx, nr, nc, rb, rp, ci, va, false, out, opt); });
}
int ll_op_create_csr_opt_z(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const int64_t* rp, const int32_t* ci,
                           const void* va, const ll_csr_options* opt, ll_operator** out) {
  return guarded([&] { create_csr<zc>(ctx, nr, nc, rb, rp, ci, va, false, out, opt); });
}
int ll_op_create_csr_opt_s(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const int64_t* rp, const int32_t* ci,
                           const float* va, const ll_csr_options* opt, ll_operator** out) {
  return guarded([&] { create_csr<float>(ctx, nr, nc, rb, rp, ci, va, false, out, opt); });
}
int ll_op_create_csr_opt_c(ll_context* ctx, int64_t nr, int64_t nc, int64_t rb, const int64_t* rp, const int32_t* ci,
                           const void* va, const ll_csr_options* opt, ll_operator** out) {
  return guarded([&] { create_csr<cf>(ctx, nr, nc, rb, rp, ci, va, false, out, opt); });
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
    LL_REQUIRE(kind == LL_SPMV_CSR_STREAM || kind == LL_SPMV_PB || kind == LL_SPMV_TILED, "unknown SpMV kernel");
    LL_REQUIRE(kind != LL_SPMV_PB || op->d_pb_val != nullptr,
               "operator has no propagation-blocked image (not selected at creation; LL_SPMV_KEEP_BOTH=1 keeps every image)");
    LL_REQUIRE(kind != LL_SPMV_CSR_STREAM || (op->d_row_ptr != nullptr && (op->d_col != nullptr || op->nnz == 0 || op->csr_split)),
               "operator has released its CSR image (another kernel was selected at creation; LL_SPMV_KEEP_BOTH=1 keeps every image)");
    LL_REQUIRE(kind != LL_SPMV_TILED || op->tl_nrb > 0,
               "operator has no tiled image (matrix not eligible, or not selected at creation; LL_SPMV_KEEP_BOTH=1 keeps every image)");
    op->spmv_kind = kind;
  });
}
int ll_op_set_accuracy(ll_operator* op, int accuracy) {
  return guarded([&] {
    LL_REQUIRE(op != nullptr && op->kind == ll_operator::CSR, "not a CSR operator");
    LL_REQUIRE(accuracy == LL_ACCURACY_NORMWISE || accuracy == LL_ACCURACY_COMPONENTWISE,
               "accuracy must be LL_ACCURACY_NORMWISE or LL_ACCURACY_COMPONENTWISE");
    if (op->tl_nrb > 0) op->tl_ordered = accuracy == LL_ACCURACY_COMPONENTWISE;  // the tiled image serves both classes
    op->accuracy_req = accuracy;
    if (op->d_pb_val == nullptr) return;  // no PB image: CSR-stream is component-wise whatever is asked, the tiled kernel was set above
    if (accuracy == LL_ACCURACY_COMPONENTWISE) {
      if (op->pb_phase2 == LL_PB_FIXED) op->pb_phase2 = LL_PB_ORDERED;
    } else {
      LL_REQUIRE(op->d_pb_rexp != nullptr && op->d_pb_blockmax != nullptr,
                 "this image was built without the row exponents of the fixed-point sums (row block too large for them)");
      op->pb_phase2 = LL_PB_FIXED;
    }
  });
}
int ll_op_accuracy(const ll_operator* op, int* accuracy_out) {
  return guarded([&] {
    LL_REQUIRE(op != nullptr && accuracy_out != nullptr, "null argument");
    const bool fixed = op->kind == ll_operator::CSR && ((op->spmv_kind == LL_SPMV_PB && op->pb_phase2 == LL_PB_FIXED) ||
                                                        (op->spmv_kind == LL_SPMV_TILED && !op->tl_ordered));
    *accuracy_out = fixed ? LL_ACCURACY_NORMWISE : LL_ACCURACY_COMPONENTWISE;
  });
}
int ll_op_selected_spmv(const ll_operator* op, int* kind_out) {
  return guarded([&] {
    LL_REQUIRE(op != nullptr && kind_out != nullptr, "null argument");
    *kind_out = op->spmv_kind;
  });
}
int ll_op_autotune_ms(const ll_operator* op, double* csr_stream_ms, double* pb_ms) {
  return guarded([&] {
    LL_REQUIRE(op != nullptr, "null operator");
    if (csr_stream_ms) *csr_stream_ms = (double)op->tune_ms[LL_SPMV_CSR_STREAM];
    if (pb_ms) *pb_ms = (double)op->tune_ms[LL_SPMV_PB];
  });
}
int ll_op_autotune_ms_of(const ll_operator* op, int kind, double* ms) {
  return guarded([&] {
    LL_REQUIRE(op != nullptr && ms != nullptr && kind >= LL_SPMV_CSR_STREAM && kind <= LL_SPMV_TILED, "bad argument");
    *ms = (double)op->tune_ms[kind];
  });
}
int ll_op_tiled_layout(const ll_operator* op, int* row_blocks, int* own_column_row_blocks) {
  return guarded([&] {
    LL_REQUIRE(op != nullptr, "null operator");
    if (row_blocks) *row_blocks = op->tl_nrb;
    if (own_column_row_blocks) *own_column_row_blocks = op->ctx->nranks > 1 ? op->tl_n_interior : op->tl_nrb;
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
int ll_tridiag_bisect_multi(int64_t m, const double* alpha, const double* beta, int64_t nk, const int64_t* ks, double* out) {
  return guarded([&] {
    LL_REQUIRE(m >= 1 && alpha && out && ks && nk >= 1 && (beta || m == 1), "bad argument");
    for (int64_t j = 0; j < nk; ++j) LL_REQUIRE(ks[j] >= 0 && ks[j] < m, "root index out of range");
    tridiag_bisect_multi(m, alpha, beta, (int)nk, ks, out);
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
    p->tridiag_mode = LL_TRIDIAG_AUTO;  // decision- and value-identical to the reference's per-iteration QR, O(k) instead of O(k^2)
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
