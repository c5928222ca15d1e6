// Host drivers: the device-resident Krylov loop of LambdaLanczos<T>::run (LL:216-366) and Exponentiator<T>::run
// (EX:87-173).  Everything n-sized stays in HBM; per iteration the host receives four doubles (alpha_k, beta_k^2
// and two diagnostics) through pinned, device-mapped memory and runs the k x k tridiagonal step (a11/a12) while
// the device already executes iteration k+1 (lag-1 speculation: a speculative iteration only writes basis slots
// the results never read, so stopping one iteration "late" on the device is harmless).
#include "engine.hpp"
#include "ritz_tracker.hpp"
#include "trace.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>
#include <random>
#include <string>
#include <system_error>
#include <thread>

namespace ll {

static inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
// Environment switches used below come from ctx->tune (read once per context, ll_internal.hpp):
//   dgks_threshold  DGKS "twice is enough": a second Gram-Schmidt pass is due when the first one removed more than this
//                   fraction of ||w||^2 (LL_DGKS_THRESHOLD; a value > 1 forces the second pass in every iteration: tests).
//   tridiag_lag     sharded contexts consume the helper thread's verdicts a fixed number of iterations late
//                   (StepWorker::consume); each stop costs that many speculative iterations, a slow host step is hidden
//                   for that many (LL_TRIDIAG_LAG; negative: the single-process opportunistic policy — unsafe with more
//                   than one rank, kept to demonstrate the hang).
// Whole-loop entry points accept host OR device memory for their n-sized inputs and outputs (start vector, Ritz
// vectors, Exponentiator input/output): a device pointer keeps the vector in HBM (no PCIe crossing, no staging).
static bool is_device_ptr(const void* p) {
  if (p == nullptr) return false;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // plain (unregistered) host memory
    return false;
  }
  return a.type == hipMemoryTypeDevice;
}
// n-sized copies between two HOST buffers at the reference's std::vector boundary (pinned staging buffer -> the caller's vector):
// one thread moves 8-10 GB/s, which made this copy the longest single item of a run's epilogue (80 MB: 9 ms); four threads
// share it from 8 MiB up.
void host_copy(void* dst, const void* src, size_t bytes) {
  constexpr size_t kParallelFrom = (size_t)8 << 20;
  constexpr int kThreads = 4;
  if (bytes < kParallelFrom) {
    std::memcpy(dst, src, bytes);
    return;
  }
  const size_t piece = ((bytes / kThreads) + 4095) & ~(size_t)4095;
  std::thread th[kThreads - 1];
  int started = 0;
  for (int t = 1; t < kThreads; ++t) {
    const size_t off = std::min(bytes, (size_t)t * piece), len = std::min(bytes, (size_t)(t + 1) * piece) - off;
    try {
      th[t - 1] = std::thread([=] {
        if (len) std::memcpy((char*)dst + off, (const char*)src + off, len);
      });
      ++started;
    } catch (const std::system_error&) {
      // no thread to be had (thread limit, cgroup pids): this and the remaining ranges are copied here — never std::terminate
      // out of a joinable thread's destructor, never an error for what is only a slower copy
      const size_t rest = std::min(bytes, (size_t)t * piece);
      std::memcpy((char*)dst + rest, (const char*)src + rest, bytes - rest);
      break;
    }
  }
  std::memcpy(dst, src, std::min(bytes, piece));
  for (int t = 0; t < started; ++t) th[t].join();
}
// LL_STALL_TRACE=ms: a whole-loop call that takes longer than that prints where its time went (host timestamps at
// the phase boundaries) — for hunting one-off runtime stalls in launch-bound runs.
struct StallTrace {
  double limit_s = -1.0;
  const char* what;
  std::vector<std::pair<const char*, double>> pts;
  StallTrace(const char* w, double limit_ms) : what(w) {
    if (limit_ms >= 0) limit_s = limit_ms * 1e-3;
    if (limit_s >= 0) pts.emplace_back("start", now_s());
  }
  void at(const char* label) {
    if (limit_s >= 0) pts.emplace_back(label, now_s());
  }
  ~StallTrace() {
    if (limit_s < 0 || pts.size() < 2 || pts.back().second - pts.front().second < limit_s) return;
    std::fprintf(stderr, "[ll stall] %s took %.2f ms:", what, (pts.back().second - pts.front().second) * 1e3);
    for (size_t i = 1; i < pts.size(); ++i) std::fprintf(stderr, " %s +%.2f", pts[i].first, (pts[i].second - pts[i - 1].second) * 1e3);
    std::fprintf(stderr, "\n");
  }
};

// ================================================================= Basis / RunList
template <typename T> Basis<T>::~Basis() {
  // slabs go back to the context's cache: the next run() on this context reuses them instead of paying
  // hipMalloc/hipFree of tens of GB per call (ll_ctx_release_cache or ll_ctx_destroy frees them)
  const size_t bytes = (size_t)chunk_vecs * (size_t)ld * sizeof(T);
  for (T* p : chunks) ctx->cache_put((void*)p, bytes);
}
template <typename T> void Basis<T>::init(ll_context* c, int64_t n_local_, int64_t ld_, int64_t chunk_vecs_) {
  ctx = c;
  n_local = n_local_;
  ld = ld_;
  chunk_vecs = chunk_vecs_;
}
template <typename T> T* Basis<T>::vec(int64_t k) {
  const int64_t ci = k / chunk_vecs;
  while ((int64_t)chunks.size() <= ci) {
    T* p = nullptr;
    const size_t bytes = (size_t)chunk_vecs * (size_t)ld * sizeof(T);
    for (size_t i = 0; i < ctx->slab_cache.size(); ++i)
      if (ctx->slab_cache[i].second == bytes) {
        p = (T*)ctx->slab_cache[i].first;
        ctx->slab_cache.erase(ctx->slab_cache.begin() + (long)i);
        break;
      }
    if (p) {
      chunks.push_back(p);
      continue;
    }
    // (a 4 GiB hipMalloc was measured at 0.3-0.5 ms on every box of round 5: allocating the next slab ahead of need on a helper
    // thread changed nothing and was removed again — what does stall the loop is a hipFREE, see LoopState::begin_pass)
    hipError_t e = hipMalloc((void**)&p, bytes);
    if (e != hipSuccess && !ctx->slab_cache.empty()) {  // make room: drop cached slabs of other shapes and retry
      (void)hipGetLastError();
      for (auto& c : ctx->slab_cache) (void)hipFree(c.first);
      ctx->slab_cache.clear();
      e = hipMalloc((void**)&p, bytes);
    }
    if (e != hipSuccess) {
      set_error("out of device memory growing the Krylov basis to " + std::to_string((chunks.size() + 1) * chunk_vecs) +
                " vectors of " + std::to_string(ld * sizeof(T)) + " bytes: " + hipGetErrorString(e));
      throw Failure{LL_ERR_ALLOC};
    }
    chunks.push_back(p);
  }
  return chunks[ci] + (k % chunk_vecs) * ld;
}

template <typename T> std::vector<BasisSegs<T>> RunList<T>::groups(int max_vecs) const {
  std::vector<BasisSegs<T>> out;
  BasisSegs<T> cur;
  cur.nseg = 0;
  cur.ld = ld;
  int cur_vecs = 0;
  auto flush = [&]() {
    if (cur.nseg > 0) out.push_back(cur);
    cur.nseg = 0;
    cur_vecs = 0;
  };
  for (auto& r : runs) {
    const T* base = r.first;
    int left = r.second;
    while (left > 0) {
      if (cur.nseg == kMaxSegs || cur_vecs == max_vecs) flush();
      const int take = std::min(left, max_vecs - cur_vecs);
      cur.base[cur.nseg] = base;
      cur.count[cur.nseg] = take;
      ++cur.nseg;
      cur_vecs += take;
      base += (int64_t)take * ld;
      left -= take;
    }
  }
  flush();
  return out;
}

// ================================================================= Engine
template <typename T> void Engine<T>::all_reduce(double* d, size_t count) {
  if (ctx->comm == nullptr) return;
  if (ctx->profiling) {
    hipEvent_t a, b;
    LL_HIP(hipEventCreate(&a));
    LL_HIP(hipEventCreate(&b));
    LL_HIP(hipEventRecord(a, ctx->stream));
    comm_allreduce_sum(ctx->comm, d, count, ctx->stream);
    LL_HIP(hipEventRecord(b, ctx->stream));
    ctx->ev_allreduce.emplace_back(a, b);
  } else {
    comm_allreduce_sum(ctx->comm, d, count, ctx->stream);
  }
}
template <typename T> void Engine<T>::comm_timer_begin(hipStream_t cs) {
  if (!ctx->profiling) return;
  hipEvent_t a, b;
  LL_HIP(hipEventCreate(&a));
  LL_HIP(hipEventCreate(&b));
  ctx->ev_gather.emplace_back(a, b);
  LL_HIP(hipEventRecord(a, cs));
}
template <typename T> void Engine<T>::comm_timer_end(hipStream_t cs) {
  if (!ctx->profiling) return;
  LL_HIP(hipEventRecord(ctx->ev_gather.back().second, cs));
}

template <typename T> void Engine<T>::fetch(const double* d, double* host, size_t count) {
  LL_HIP(hipMemcpyAsync(host, d, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  LL_HIP(hipStreamSynchronize(ctx->stream));
}

template <typename T>
void Engine<T>::apply(const T* x_local, T* y, double offset, double* d_alpha, bool x_padded, DeferredAlpha* defer,
                      const ScaleIn<T>* sc, const double* xnorm2) {
  TraceRange trace("ll::apply (mv_mul + offset + alpha)");
  LL_REQUIRE(sc == nullptr || can_defer_scale(), "internal: this operator cannot normalise its input on the fly");
  LL_REQUIRE(xnorm2 == nullptr || (can_scale_input() && sc == nullptr), "internal: this operator cannot scale its input");
  ScaleIn<T> from_norm;  // the non-PB kernels take the norm as a one-element list of "partials" (nothing published)
  if (xnorm2 && !(op->kind == ll_operator::CSR && (op->spmv_kind == LL_SPMV_PB || op->spmv_kind == LL_SPMV_TILED))) {
    from_norm.partials = xnorm2;
    from_norm.nparts = 1;
    sc = &from_norm;
  }
  hipStream_t s = ctx->stream;
  ctx->ensure_alpha_partials(std::max<size_t>(kMaxSpmvGrid, (size_t)std::max(op->pb_nrb, op->tl_nrb)));
  double* const dotp = d_alpha ? ctx->d_alpha_partials : nullptr;
  int nparts = 0;
  if (op->kind == ll_operator::STENCIL) {
    // exchange step of the lattice operator: one hyperplane from each ring neighbour instead of the all-gather
    const int64_t H = op->st_halo;
    const T *lo, *hi;
    if (ctx->comm == nullptr) {  // the shard is the whole lattice: the ring neighbours are its own ends
      lo = x_local + (n_local - H);
      hi = x_local;
    } else {
      const size_t hb = (size_t)H * sizeof(T);
      ctx->ensure_halo(2 * hb);
      T* rlo = (T*)ctx->d_halo;
      T* rhi = rlo + H;
      const bool ring = op->st.periodic[0] != 0;
      const int prev = ctx->rank > 0 ? ctx->rank - 1 : (ring ? ctx->nranks - 1 : -1);
      const int next = ctx->rank + 1 < ctx->nranks ? ctx->rank + 1 : (ring ? 0 : -1);
      comm_timer_begin(s);
      comm_halo_exchange(ctx->comm, x_local, rlo, prev, x_local + (n_local - H), rhi, next, hb, s);
      comm_timer_end(s);
      lo = rlo;
      hi = rhi;
    }
    nparts = launch_stencil<T>(*op, x_local, lo, hi, y, offset, dotp, s, sc);
  } else if (op->kind == ll_operator::CSR || op->kind == ll_operator::DENSE) {
    const bool pb = op->kind == ll_operator::CSR && op->spmv_kind == LL_SPMV_PB;
    // CSR-stream on a sharded context: the image is split by column ownership (capi.cpp build_csr_split); the dense row
    // block splits by column range without a second image
    // (dense: only when the shard's column range starts and ends on 16-byte pieces of the rows — with an unaligned boundary both
    // parts would take the scalar path of dense_mv_kernel, and gather-then-multiply, which vectorises whole rows, is faster)
    constexpr int64_t V = (int64_t)(16 / sizeof(T)) > 0 ? (int64_t)(16 / sizeof(T)) : 1;
    const bool dense_split_ok = op->n % V == 0 && op->row_begin % V == 0 && (op->row_begin + op->n_local) % V == 0;
    const bool tiled = op->kind == ll_operator::CSR && op->spmv_kind == LL_SPMV_TILED;
    const bool split = ctx->comm != nullptr && ((op->kind == ll_operator::CSR && !pb && !tiled && op->csr_split) ||
                                                (op->kind == ll_operator::DENSE && ctx->tune.csr_split && dense_split_ok));
    const T* x_full = x_local;
    const T* x_own = x_local;  // what the own-column blocks of the PB kernels read
    bool remote_done = false;
    if (ctx->comm != nullptr) {
      // exchange step: every rank needs the whole x for its row block.
      // Every rank sends n_shard elements (equal strides); the last shard can be shorter.  Basis vectors are padded to
      // the stride (x_padded); a caller-provided shard of exactly n_local elements is copied into a padded send
      // buffer first (the tail of the gathered buffer beyond n is never referenced by a matrix entry).
      const int P = ctx->nranks;
      const size_t shard_bytes = (size_t)op->n_shard * sizeof(T);
      ctx->ensure_xfull(shard_bytes * (size_t)(P + 1));
      T* gathered = (T*)ctx->d_xfull;
      const T* send = x_local;
      if (!x_padded && op->n_local < op->n_shard) {
        T* pad = (T*)((char*)ctx->d_xfull + shard_bytes * (size_t)P);
        LL_HIP(hipMemsetAsync(pad, 0, shard_bytes, s));
        LL_HIP(hipMemcpyAsync(pad, x_local, (size_t)op->n_local * sizeof(T), hipMemcpyDeviceToDevice, s));
        send = pad;
      }
      x_own = send;
      x_full = gathered;
      // PB: the gather is cut into chunks (op->gather) laid out chunk-major; every other kernel needs global order
      GatherPlan plan;
      if (pb) plan = op->gather;
      else {
        plan.nchunks = 1;
        plan.start[0] = 0;
        plan.len[0] = op->n_shard;
      }
      const bool overlap = (pb || split || tiled) && ctx->tune.comm_overlap && ctx->comm_stream != nullptr;
      hipStream_t cs = overlap ? ctx->comm_stream : s;
      // tiled, fixed-point class: the grid's scale needs max |x| over the WHOLE vector before the first launch — every rank's own
      // maximum (two small kernels) travels in an 8-byte all-gather in front of the vector's
      const bool tl_max = tiled && !op->tl_ordered;
      if (tl_max) launch_tl_xmax_local<T>(*op, send, s);
      comm_timer_begin(cs);
      if (overlap) {
        LL_HIP(hipEventRecord(ctx->ev_x_ready, s));  // everything enqueued so far (x final, previous readers of the
        LL_HIP(hipStreamWaitEvent(cs, ctx->ev_x_ready, 0));  // gathered buffer done) precedes the gather
      }
      if (tl_max) {
        comm_allgather(ctx->comm, op->d_tl_xmax + tl_xmax_local_slot(), op->d_tl_xmax, sizeof(double), cs);
        if (overlap) LL_HIP(hipEventRecord(ctx->ev_chunk[1], cs));
      }
      for (int c = 0; c < plan.nchunks; ++c) {
        comm_allgather(ctx->comm, send + plan.start[c], gathered + (int64_t)P * plan.start[c],
                       (size_t)plan.len[c] * sizeof(T), cs);
        if (overlap) LL_HIP(hipEventRecord(ctx->ev_chunk[c], cs));
      }
      comm_timer_end(cs);
      if (split) {
        // the own-column product (no exchange needed) runs under the gather, the other ranks' columns are added when the
        // gathered vector has arrived; LL_COMM_OVERLAP=0 issues the same two kernels behind the gather on one stream
        if (op->kind == ll_operator::DENSE) {
          launch_dense_mv<T>(*op, x_local, x_local, y, offset, nullptr, s, sc, 1);
          if (overlap) LL_HIP(hipStreamWaitEvent(s, ctx->ev_chunk[0], 0));
          nparts = launch_dense_mv<T>(*op, gathered, x_local, y, offset, dotp, s, sc, 2);
        } else {
          launch_spmv<T>(*op, x_local, x_local, y, offset, nullptr, s, sc, 1);
          if (overlap) LL_HIP(hipStreamWaitEvent(s, ctx->ev_chunk[0], 0));
          nparts = launch_spmv<T>(*op, gathered, x_local, y, offset, dotp, s, sc, 2);
        }
        remote_done = true;
      } else if (tiled) {
        // the row blocks whose tiles are all own-column tiles run under the gather (x = the own shard), the others when the
        // vector has arrived; LL_COMM_OVERLAP=0 issues the same two launches behind the gather on one stream
        if (overlap && tl_max) LL_HIP(hipStreamWaitEvent(s, ctx->ev_chunk[1], 0));
        launch_spmv_tiled_pass<T>(*op, 0, send, op->row_begin, op->row_begin + op->n_shard, x_local, y, offset, dotp, s, xnorm2, P);
        if (overlap) LL_HIP(hipStreamWaitEvent(s, ctx->ev_chunk[0], 0));
        nparts = launch_spmv_tiled_pass<T>(*op, 1, gathered, 0, op->n, x_local, y, offset, dotp, s, xnorm2, P);
        remote_done = true;
      } else if (overlap) {
        // own-column blocks run under the gather; every chunk's remote blocks start when that chunk has arrived
        launch_pb_phase1<T>(*op, 0, op->pb_own_count, x_own, s, xnorm2);
        for (int c = 0; c < plan.nchunks; ++c) {
          LL_HIP(hipStreamWaitEvent(s, ctx->ev_chunk[c], 0));
          launch_pb_phase1<T>(*op, op->pb_chunk_first[c], op->pb_chunk_count[c], gathered, s, xnorm2);
        }
        nparts = launch_pb_phase2<T>(*op, x_local, y, offset, dotp, s, xnorm2);
        remote_done = true;
      }
    }
    if (remote_done) {
    } else if (op->kind == ll_operator::DENSE)
      nparts = launch_dense_mv<T>(*op, x_full, x_local, y, offset, dotp, s, sc);
    else if (pb)
      nparts = launch_spmv_pb<T>(*op, x_full, x_own, x_local, y, offset, dotp, s, xnorm2);
    else if (op->spmv_kind == LL_SPMV_TILED)  // single GPU (sharded contexts took the two-launch form above): x_local is the whole x
      nparts = launch_spmv_tiled<T>(*op, x_local, y, offset, dotp, s, xnorm2);
    else {
      LL_REQUIRE(op->d_col != nullptr || op->nnz == 0,
                 "this operator kept only its column-split image (created on a sharded context) and needs that communicator");
      nparts = launch_spmv<T>(*op, x_full, x_local, y, offset, dotp, s, sc);
    }
  } else {
    LL_REQUIRE(!(ctx->comm != nullptr), "callback operators are not supported on sharded contexts");
    const size_t bytes = (size_t)n_local * sizeof(T);
    if (op->kind == ll_operator::HOST_CB) {
      // unmodified user code (LL:120-126): one D2H + one H2D of an n-vector per call
      // through the context's pinned callback buffers [in | out]: full-rate DMA, no pageable bounce copies
      char* h_in = (char*)ctx->ensure_cb_stage(2 * bytes);
      char* h_out = h_in + bytes;
      if (ctx->ev_cb) LL_HIP(hipEventSynchronize(ctx->ev_cb));  // the previous call's upload out of h_out has finished
      LL_HIP(hipMemcpyAsync(h_in, x_local, bytes, hipMemcpyDeviceToHost, s));
      std::memset(h_out, 0, bytes);  // "out" is zero-filled on entry (LL:242, EX:107); overlaps the copy above
      LL_HIP(hipStreamSynchronize(s));
      int rc = op->host_fn(h_in, h_out, n_local, op->user);
      if (rc != 0) {
        set_error("mv_mul host callback returned " + std::to_string(rc));
        throw Failure{LL_ERR_CALLBACK};
      }
      if (!ctx->tune.iter_trace.empty()) {  // LL_ITER_TRACE: what the user's code saw and returned (a stale or torn buffer shows here)
        if (FILE* f = std::fopen(ctx->tune.iter_trace.c_str(), "a")) {
          double sin2 = 0.0, sout2 = 0.0, dot = 0.0;
          const typename scalar_traits<T>::real* a = (const typename scalar_traits<T>::real*)h_in;
          const typename scalar_traits<T>::real* b = (const typename scalar_traits<T>::real*)h_out;
          for (int64_t i = 0; i < n_local * R; ++i) {
            sin2 += (double)a[i] * a[i];
            sout2 += (double)b[i] * b[i];
            dot += (double)a[i] * b[i];
          }
          std::fprintf(f, "cb x=%p |in|^2=%.17g |out|^2=%.17g <in,out>=%.17g\n", (const void*)x_local, sin2, sout2, dot);
          std::fclose(f);
        }
      }
      // no second synchronisation: the next callback waits for this upload (ev_cb) before it reuses the buffer
      LL_HIP(hipMemcpyAsync(y, h_out, bytes, hipMemcpyHostToDevice, s));
      if (!ctx->ev_cb) LL_HIP(hipEventCreateWithFlags(&ctx->ev_cb, hipEventDisableTiming));
      LL_HIP(hipEventRecord(ctx->ev_cb, s));
    } else {
      LL_HIP(hipMemsetAsync(y, 0, bytes, s));
      int rc = op->dev_fn(x_local, y, n_local, (void*)s, op->user);
      if (rc != 0) {
        set_error("mv_mul device callback returned " + std::to_string(rc));
        throw Failure{LL_ERR_CALLBACK};
      }
    }
    nparts = launch_offset_dot<T>(n_local, x_local, y, offset, dotp, s);
  }
  if (d_alpha) {
    if (defer && ctx->comm == nullptr) {  // the caller's multi-dot folds them
      defer->partials = dotp;
      defer->nparts = nparts;
    } else {
      launch_reduce_cols(dotp, nparts, 1, d_alpha, nullptr, s);
      all_reduce(d_alpha, 1);
    }
  }
}

template <typename T> void Engine<T>::norm2_dev(const T* v, double* d_out) {
  ctx->ensure_partials((size_t)kMaxGrid * R);
  const int grid = launch_dot<T>(n_local, v, v, ctx->d_partials, ctx->stream);
  ctx->ensure_h(4);
  if (R == 1) {
    launch_reduce_cols(ctx->d_partials, grid, 1, d_out, nullptr, ctx->stream);
  } else {  // real part is column 0
    launch_reduce_cols(ctx->d_partials, grid, R, ctx->d_h, nullptr, ctx->stream);
    launch_copy_scalar(d_out, ctx->d_h, ctx->stream);
  }
  all_reduce(d_out, 1);
}

template <typename T> void Engine<T>::dot_dev(const T* a, const T* b, double* d_out) {
  ctx->ensure_partials((size_t)kMaxGrid * R);
  const int grid = launch_dot<T>(n_local, a, b, ctx->d_partials, ctx->stream);
  launch_reduce_cols(ctx->d_partials, grid, R, d_out, nullptr, ctx->stream);
  all_reduce(d_out, R);
}

// LDS budget of mdot / lagged_kernel: 4 waves x ncols doubles in the 160 KB of a CU (one workgroup per CU then, which is
// how the streaming kernels run on long vectors anyway)  =>  reals * nb <= 5000
template <typename T> static int max_vecs_per_launch() { return kLaggedMaxCols / scalar_traits<T>::reals; }

template <typename T>
NormRefs Engine<T>::orth(T* w, const RunList<T>& runs, int mode, const ThreeTerm<T>& tt, double* c, double* h_total,
                         bool first_pass_only, Publish* publish) {
  TraceRange trace("ll::orth (three-term + Gram-Schmidt + norm)");
  hipStream_t s = ctx->stream;
  const int nb = runs.total();
  const bool sharded = ctx->comm != nullptr;
  const ThreeTerm<T> no_tt{nullptr, nullptr, nullptr, NormRefs{nullptr, nullptr, nullptr, 0}};
  ctx->ensure_h((size_t)2 * (R * nb + 2));
  double* h1 = ctx->d_h;
  double* h2 = ctx->d_h + (R * nb + 2);

  if (nb == 0) {  // three-term update (if any) + ||w||^2 only
    BasisSegs<T> none;
    none.nseg = 0;
    none.ld = runs.ld;
    ctx->ensure_partials(kMaxGrid);
    const int grid = launch_mdot<T>(n_local, w, none, tt, nullptr, ctx->d_partials, ctx->tune.blas_small_bytes, s);
    if (publish && !sharded && publish->can_defer) {
      *publish = Publish{publish->host, publish->alpha, true, true, true, ctx->d_partials, grid, c + 1, nullptr};
    } else if (publish && !sharded) {
      launch_reduce_publish(ctx->d_partials, grid, c + 1, publish->alpha, nullptr, publish->host, s);
      publish->done = true;
    } else {
      launch_reduce_cols(ctx->d_partials, grid, 1, c + 1, nullptr, s);
      all_reduce(c + 1, 1);
    }
    return plain_norm(c + 1);
  }

  if (mode == LL_ORTH_MGS) {
    // The reference's operation order (LA:132-144): for every basis vector h = <u,w>; w -= h u, strictly sequential.
    ctx->ensure_partials((size_t)kMaxGrid * (R + 1));
    if (tt.u_cur) {
      BasisSegs<T> none;
      none.nseg = 0;
      none.ld = runs.ld;
      launch_mdot<T>(n_local, w, none, tt, nullptr, ctx->d_partials, ctx->tune.blas_small_bytes, s);
    }
    int j = 0, grid = 0;
    for (auto& r : runs.runs)
      for (int i = 0; i < r.second; ++i, ++j) {
        BasisSegs<T> one;
        one.nseg = 1;
        one.ld = runs.ld;
        one.base[0] = r.first + (int64_t)i * runs.ld;
        one.count[0] = 1;
        grid = launch_mdot<T>(n_local, w, one, no_tt, nullptr, ctx->d_partials, ctx->tune.blas_small_bytes, s);
        launch_reduce_cols(ctx->d_partials, grid, R + 1, h1 + R * j, S(kScalSpare), s);
        all_reduce(h1 + R * j, R);
        grid = launch_maxpy<T>(n_local, w, one, h1 + R * j, nullptr, ctx->d_partials, ctx->tune.blas_small_bytes, s);
      }
    launch_reduce_cols(ctx->d_partials, grid, 1, c + 1, nullptr, s);
    all_reduce(c + 1, 1);
    if (h_total) LL_HIP(hipMemcpyAsync(h_total, h1, (size_t)R * nb * sizeof(double), hipMemcpyDeviceToDevice, s));
    return plain_norm(c + 1);
  }

  const std::vector<BasisSegs<T>> groups = runs.groups(max_vecs_per_launch<T>());
  const NormRefs refs{c, c + 1, c + 2, mode == LL_ORTH_CGS2 ? 1 : 0, ctx->tune.dgks_threshold};
  const NormRefs* pred = mode == LL_ORTH_CGS2 ? nullptr : &refs;
  auto count_of = [](const BasisSegs<T>& g) {
    int t = 0;
    for (int i = 0; i < g.nseg; ++i) t += g.count[i];
    return t;
  };
  size_t max_cols = 1;
  for (auto& g : groups) max_cols = std::max(max_cols, (size_t)R * count_of(g) + 1);
  ctx->ensure_partials((size_t)kMaxGrid * max_cols);

  // ---- pass 1: h = U^H w (+ fused three-term update and ||w||^2), then w -= U h (+ fused ||w||^2)
  int off = 0;
  int grid = 0;
  double* norm_partials = ctx->d_partials;  // where the multi-axpy leaves the partial sums of ||w'||^2
  bool folded_in_maxpy = false;
  for (size_t g = 0; g < groups.size(); ++g) {
    const int nbg = count_of(groups[g]);
    const bool last = g + 1 == groups.size();
    const int mgrid = launch_mdot<T>(n_local, w, groups[g], g == 0 ? tt : no_tt, nullptr, ctx->d_partials, ctx->tune.blas_small_bytes, s);
    if (groups.size() == 1 && !sharded && ctx->tune.fuse_launches) {
      // small vectors, small grids: the multi-axpy folds the coefficients itself (one launch less per iteration)
      if (!ctx->d_norm_partials) ctx->dev_malloc((void**)&ctx->d_norm_partials, (size_t)kMaxGrid * sizeof(double), "norm partials");
      folded_in_maxpy = launch_maxpy_folding<T>(n_local, w, groups[0], ctx->d_partials, mgrid, h1, c, ctx->d_norm_partials,
                                                ctx->tune.blas_small_bytes, &grid, s);
      if (folded_in_maxpy) norm_partials = ctx->d_norm_partials;
    }
    if (!folded_in_maxpy)
      launch_reduce_cols(ctx->d_partials, mgrid, R * nbg + 1, h1 + R * off, (last && !sharded) ? c : nullptr, s);
    off += nbg;
  }
  // Sharded whole-loop passes: the norm after the pass follows from what the one all-reduce below delivers
  // (||w'||^2 = ||w||^2 - sum |h_j|^2 for an orthonormal basis) — one all-reduce per iteration less.  Its relative error
  // is eps * ||w||^2 / ||w'||^2, i.e. a few eps whenever the DGKS test (evaluated on these two numbers) does not ask
  // for a second pass anyway.  LL_SHARDED_NORM=measured restores the reduced-and-all-reduced partial norms of maxpy.
  const bool derive_norm = !ctx->tune.sharded_norm_measured;
  const bool derive = sharded && first_pass_only && mode == LL_ORTH_CGS_DGKS && derive_norm;
  if (sharded) {  // one all-reduce for all coefficients and the norm (latency-sized)
    all_reduce(h1, (size_t)R * nb + 1);
    if (!derive) launch_copy_scalar(c, h1 + R * nb, s);  // (derive: copied by the derive kernel after the update)
  }
  off = 0;
  for (size_t g = 0; g < groups.size() && !folded_in_maxpy; ++g) {
    grid = launch_maxpy<T>(n_local, w, groups[g], h1 + R * off, nullptr, ctx->d_partials, ctx->tune.blas_small_bytes, s);
    off += count_of(groups[g]);
  }
  if (derive && publish && publish->can_defer) {  // ... inside the caller's normalisation kernel (launch_scale_derive)
    publish->derive = true;
    publish->derive_c0 = h1 + R * nb;
    publish->derive_h = h1;
    publish->derive_count = R * nb;
    publish->c0_out = c;
    publish->c1 = c + 1;
  } else if (derive) {  // norm + copy of ||w||^2 + (whole-loop drivers) the publish step in one small launch
    launch_derive_norm(h1 + R * nb, h1, R * nb, c, c + 1, publish ? publish->alpha : nullptr, publish ? publish->host : nullptr, s);
    if (publish) publish->done = true;
  } else if (publish && !sharded && first_pass_only && mode == LL_ORTH_CGS_DGKS && publish->can_defer) {
    *publish = Publish{publish->host, publish->alpha, true, true, true, norm_partials, grid, c + 1, c};
  } else if (publish && !sharded && first_pass_only && mode == LL_ORTH_CGS_DGKS) {
    launch_reduce_publish(norm_partials, grid, c + 1, publish->alpha, c, publish->host, s);
    publish->done = true;
  } else {
    launch_reduce_cols(norm_partials, grid, 1, c + 1, nullptr, s);
    all_reduce(c + 1, 1);
  }
  if (first_pass_only && mode == LL_ORTH_CGS_DGKS) {
    if (h_total) LL_HIP(hipMemcpyAsync(h_total, h1, (size_t)R * nb * sizeof(double), hipMemcpyDeviceToDevice, s));
    return NormRefs{c, c + 1, c + 1, 0};  // final norm = c1; (c0, c1) go to the host through publish
  }

  // ---- pass 2: always (CGS2) or only when ||w|| dropped below ||w_before||/sqrt(2) (DGKS); decided on the device
  off = 0;
  for (size_t g = 0; g < groups.size(); ++g) {
    const int nbg = count_of(groups[g]);
    const int g2 = launch_mdot<T>(n_local, w, groups[g], no_tt, pred, ctx->d_partials, ctx->tune.blas_small_bytes, s);
    launch_reduce_cols(ctx->d_partials, g2, R * nbg + 1, h2 + R * off, S(kScalSpare), s);
    off += nbg;
  }
  if (sharded) all_reduce(h2, (size_t)R * nb);
  off = 0;
  for (size_t g = 0; g < groups.size(); ++g) {
    grid = launch_maxpy<T>(n_local, w, groups[g], h2 + R * off, pred, ctx->d_partials, ctx->tune.blas_small_bytes, s);
    off += count_of(groups[g]);
  }
  launch_reduce_cols(ctx->d_partials, grid, 1, c + 2, nullptr, s);
  all_reduce(c + 2, 1);
  if (h_total) {
    LL_HIP(hipMemcpyAsync(h_total, h1, (size_t)R * nb * sizeof(double), hipMemcpyDeviceToDevice, s));
    launch_accumulate_h(h_total, h2, R * nb, pred, s);
  }
  return refs;
}

template <typename T> double Engine<T>::second_pass(T* u, const RunList<T>& runs) {
  const ThreeTerm<T> no_tt{nullptr, nullptr, nullptr, NormRefs{nullptr, nullptr, nullptr, 0}};
  const NormRefs r = orth(u, runs, LL_ORTH_CGS_DGKS, no_tt, S(kScalScratch), nullptr, true);
  launch_scale<T>(n_local, u, 0.0, &r, ctx->stream);
  double shrink = 0.0;
  fetch(r.c1, &shrink, 1);
  return shrink;
}

template <typename T>
void Engine<T>::gemv(const RunList<T>& basis, int64_t m, int nout, const T* coeff_host, T* out, int64_t ld_out) {
  TraceRange trace("ll::gemv_basis (Ritz vectors / exp output)");
  const std::vector<BasisSegs<T>> groups = basis.groups(512);
  ctx->ensure_coeff((size_t)nout * m * sizeof(T));
  LL_HIP(hipMemcpyAsync(ctx->d_coeff, coeff_host, (size_t)nout * m * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  launch_gemv_basis<T>(n_local, m, groups.data(), (int)groups.size(), nout, (const T*)ctx->d_coeff, out, ld_out,
                       ctx->stream);
  LL_HIP(hipStreamSynchronize(ctx->stream));  // coeff_host may go away; d_coeff is reused
}

// ================================================================= helpers shared by the loops
namespace {

// A run-scoped device buffer.  Like the Krylov slabs it comes from, and goes back to, the context's slab cache: a
// hipMalloc / hipFree pair per run() costs hundreds of microseconds (hipFree synchronises the device) — most of a run on
// the small problems the reference is used for.
template <typename T> struct DevBuf {
  T* p = nullptr;
  ll_context* owner = nullptr;
  size_t bytes = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p && owner) owner->cache_put((void*)p, bytes);
    p = nullptr;
  }
  void alloc(ll_context* ctx, size_t count) {
    release();
    owner = ctx;
    bytes = std::max<size_t>(count * sizeof(T), 16);
    for (size_t i = 0; i < ctx->slab_cache.size(); ++i)
      if (ctx->slab_cache[i].second == bytes) {
        p = (T*)ctx->slab_cache[i].first;
        ctx->slab_cache.erase(ctx->slab_cache.begin() + (long)i);
        return;
      }
    ctx->dev_malloc((void**)&p, bytes, "work vectors");
  }
};

struct EventRing {
  hipEvent_t ev[4];
  EventRing() {
    for (auto& e : ev) LL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  ~EventRing() {
    for (auto& e : ev) (void)hipEventDestroy(e);
  }
};

struct PhaseTimer {  // optional per-phase device timing (HIP events on the context's stream)
  // Marks come in triples — start, after the operator, end of the iteration (or pair) — and are recorded into a RING of events
  // that the context keeps between runs: a triple is read back (its events completed long ago: the host runs a group or two ahead
  // of the device) when its slot comes round again.  (One fresh event per mark — 900 for config 3's run to convergence, 10 000 for
  // config 2's — cost the first profiled run of a process up to 0.5 s of host time in hipEventCreate on some boxes.)
  static constexpr size_t kTriples = 128;
  bool on;
  hipStream_t s;
  std::vector<hipEvent_t>& evs;
  size_t n = 0;  // marks so far
  double acc_op = 0.0, acc_rest = 0.0;
  PhaseTimer(ll_context* ctx, hipStream_t st) : on(ctx->profiling), s(st), evs(ctx->timer_events) {}
  void read(size_t first) {
    float a = 0, b = 0;
    if (hipEventSynchronize(evs[first + 2]) != hipSuccess) return;
    if (hipEventElapsedTime(&a, evs[first], evs[first + 1]) == hipSuccess) acc_op += a * 1e-3;
    if (hipEventElapsedTime(&b, evs[first + 1], evs[first + 2]) == hipSuccess) acc_rest += b * 1e-3;
  }
  void mark() {
    if (!on) return;
    const size_t slot = n % (3 * kTriples);
    if (slot % 3 == 0 && n >= 3 * kTriples) read(slot);  // the triple that used these events
    if (slot >= evs.size()) {
      hipEvent_t e;
      LL_HIP(hipEventCreate(&e));
      evs.push_back(e);
    }
    LL_HIP(hipEventRecord(evs[slot], s));
    ++n;
  }
  void collect(double& t_op, double& t_rest) {
    if (!on) return;
    // complete triples still in the ring: the last min(n / 3, kTriples) ones, minus those already read when their slot was reused
    const size_t triples = n / 3, done = n >= 3 * kTriples ? (n - 3 * kTriples) / 3 + ((n % 3) ? 1 : 0) : 0;
    for (size_t t = done; t < triples; ++t) read((t % kTriples) * 3);
    t_op += acc_op;
    t_rest += acc_rest;
    acc_op = acc_rest = 0.0;
    n = 0;
  }
};

// One Lanczos iteration as the device sees it, shared by the eigen-solver and the Exponentiator loops:
//   y = A u_{k-1} + offset u_{k-1}, alpha (a1-a3)  ->  three-term update + Gram-Schmidt against `runs` + norm (a4-a7)
//   ->  normalisation + publish of the iteration's four scalars (a8).
// The last step is DEFERRED where the operator kernel can normalise its input on the fly (Engine::can_defer_scale): the
// iteration then ends with w_k unnormalised in a work buffer and the partial sums of ||w_k||^2; the NEXT iteration's
// operator kernel folds them, works with u_k = w_k / ||w_k||, writes u_k into the basis slot and publishes — one launch
// and one read of w per iteration less (three launches become two in the Exponentiator loop, five become four in the
// eigen-solver's).  flush() does the same work with the stand-alone kernel when no next iteration follows.
template <typename T> struct LoopState {
  Engine<T>& E;
  Basis<T>& U;
  EventRing& ring;
  PhaseTimer& timer;
  int64_t nl;
  hipStream_t s;
  bool fuse_launches = true, defer = false;
  DevBuf<T> work[2];      // defer: w_k lives in work[k & 1]
  bool pending = false;   // iteration pend_k ended without its normalisation / publish
  typename Engine<T>::Publish pend{nullptr, nullptr, false};
  int pend_slot = 0;
  int64_t pend_k = 0;
  NormRefs refs_prev{nullptr, nullptr, nullptr, 0};
  double t_enqueue = 0.0;
  // Lagged block Gram-Schmidt (kernels.hip, lagged_kernel): ONE sweep over the basis per iteration.  The iteration ends
  // with the raw w_k in work[k & 1], its coefficients g_k = U^H w_k in hbuf[k & 1] and ||w_k||^2 - |g_k|^2 in *lag_c1; the
  // next iteration's operator kernel takes w_k / beta_k as its input and the next sweep writes the corrected u_k.
  bool lagged = false;
  bool lag_pending = false;
  int64_t lag_k = 0;
  const double* lag_c1 = nullptr;
  int64_t n_lagged = 0;     // iterations enqueued in the lagged form (statistics)
  bool lag_ok = false;      // this pass: every iteration so far went through enqueue_lagged (the device copy of T is complete)
  double* hbuf[2] = {nullptr, nullptr};
  double *hist_alpha = nullptr, *hist_beta = nullptr, *d_lambda = nullptr;
  static constexpr int kLaggedMaxLocked = 512;
  double lag_beta2_min = 0.0;  // passes with locked vectors: smallest beta^2 the one-sweep form accepts (begin_pass)
  size_t t_off = 0;
  int64_t n_locked = 0;       // locked eigenvectors at the front of every run list (restart passes)
  const T* locked = nullptr;
  int64_t ld = 0;
  int64_t small_bytes = 0;

  // Pair form (kernels.hip, "pair" section; tools/pair_gs_model.py): TWO iterations per sweep over the basis.  State between
  // sweeps: u_0 .. u_{pair_P-1} complete in the basis; two raw vectors pending, pr1 -> u_P and pr2 -> u_{P+1}, with their
  // measured coefficients (g1p; g2p followed by <u_P, pr2>) and the squared norms of their orthogonal parts (rho1p, rho2p).
  int64_t max_k_hint = 0;     // the loop's max_iteration (sizes the sweeps' partial sums up front, begin_pass)
  bool pair_enabled = false;
  bool pair_allowed = true;   // this pass: a coefficient above kPairGate switches the form off for the rest of the pass
  bool pair_pending = false;
  int64_t pair_P = 0;
  int64_t n_pair = 0;         // iterations enqueued in the pair form (statistics; includes speculative ones that were dropped)
  int64_t n_gate_trips = 0;   // passes that left the pair form through the coefficient gate
  DevBuf<T> pwork[2];         // with work[0..1]: the four raw vectors of a pair
  DevBuf<T> psplit;           // hand-over vector of a split sweep (more stored vectors than one launch sums columns for)
  static constexpr size_t kPresizeCols = 4096;  // columns the sweeps' partial sums are sized for at pass start
  DevBuf<double> pbuf;        // coefficient records, predictions, scalars (its own allocation: ctx->d_h may move)
  const T *pr1 = nullptr, *pr2 = nullptr;
  int pset = 0;               // which pair of buffers holds pr1 / pr2: 0 = work, 1 = pwork
  const double *g1p = nullptr, *g2p = nullptr, *rho1p = nullptr, *rho2p = nullptr;
  double *prec[4] = {nullptr, nullptr, nullptr, nullptr}, *pzero = nullptr, *pp3 = nullptr, *pp4 = nullptr, *pfold = nullptr,
         *pcols = nullptr, *pscal = nullptr;
  int prec_set = 0;           // records prec[2 * prec_set], prec[2 * prec_set + 1] hold the pending pair's coefficients
  bool slot_pair[4] = {false, false, false, false};  // the scalars of this ring slot came from a pair fold (its gate is valid)
  int ev_of_slot[4] = {0, 1, 2, 3};                   // the event that covers a ring slot's scalars (a pair's two slots share one)
  static constexpr size_t kPairRec = (size_t)kLaggedMaxCols + 32;
  static constexpr int64_t kPairSmallMinBytes = (int64_t)512 << 10;  // shortest vector of the pair form (small-vector geometry; enqueue_pair)
  // Pointer table of the software-pipelined sweep (kernels.hip, pair_sweep_pipe_kernel): entry c = stored column c of this pass —
  // the locked eigenvectors, then u_0, u_1, ... — written on the device, slab by slab (launch_fill_ptrs), when a pass starts and
  // whenever the basis has grown by a slab.
  DevBuf<const T*> vtab;
  static constexpr size_t kVtabCap = (size_t)kLaggedMaxCols + (size_t)kLaggedMaxLocked + 64;
  size_t vtab_chunks = 0;     // slabs of U whose slots are in the table
  void vtab_begin_pass() {
    if (!vtab.p) return;
    launch_fill_ptrs<T>(vtab.p, 0, (int)std::min<int64_t>(n_locked, (int64_t)kVtabCap), locked, ld, s);
    vtab_chunks = 0;
  }
  const T* const* vtab_sync() {
    if (!vtab.p) return nullptr;
    for (; vtab_chunks < U.chunks.size(); ++vtab_chunks) {
      const int64_t start = n_locked + (int64_t)vtab_chunks * U.chunk_vecs;
      const int64_t count = std::min<int64_t>(U.chunk_vecs, (int64_t)kVtabCap - start);
      launch_fill_ptrs<T>(vtab.p, (int)start, (int)count, U.chunks[vtab_chunks], ld, s);
    }
    return vtab.p;
  }

  LoopState(Engine<T>& e, Basis<T>& u, EventRing& r, PhaseTimer& t, int64_t n_local, hipStream_t st)
      : E(e), U(u), ring(r), timer(t), nl(n_local), s(st) {}
  void enable_pair() {  // after enable_lagged
    if (!lagged) return;
    pair_enabled = true;
    for (auto& w : pwork)
      if (!w.p) w.alloc(E.ctx, (size_t)ld);
    // 4 records + zero record + p3 + p4 + fold scratch + folded columns (two per stored vector: twice a record) + 64 scalars
    pbuf.alloc(E.ctx, 10 * kPairRec + 64);
    if (E.ctx->tune.sweep_pipeline > 0) vtab.alloc(E.ctx, kVtabCap);
    double* b = pbuf.p;
    for (int i = 0; i < 4; ++i) prec[i] = b + (size_t)i * kPairRec;
    pzero = b + 4 * kPairRec;
    pp3 = b + 5 * kPairRec;
    pp4 = b + 6 * kPairRec;
    pfold = b + 7 * kPairRec;
    pcols = b + 8 * kPairRec;
    pscal = b + 10 * kPairRec;
    LL_HIP(hipMemsetAsync(pzero, 0, kPairRec * sizeof(double), s));
    launch_set_scalar(pscal + 0, 1.0, s);  // pscal[0] = 1 (rho1^2 of a vector that is already complete)
    // pscal[8 + 2 i], [9 + 2 i]: rho^2 pair i (alternating); pscal[16 ..]: |r3|^2, <r1, r3>
  }
  // partial sums of the sweeps: sized at pass start for up to kPresizeCols columns (begin_pass); beyond that in powers of two —
  // every growth is a hipFree, i.e. a device synchronisation
  void want_partial_cols(size_t cols) {
    if (cols > kPresizeCols) {
      size_t p2 = kPresizeCols;
      while (p2 < cols) p2 *= 2;
      cols = p2;
    }
    E.ctx->ensure_partials((size_t)kMaxGrid * cols);
  }
  void enable_defer(int64_t ld_) {
    defer = true;
    ld = ld_;
    for (auto& w : work)
      if (!w.p) w.alloc(E.ctx, (size_t)ld_);
  }
  void enable_lagged(int64_t ld_) {
    lagged = true;
    ld = ld_;
    small_bytes = E.ctx->tune.blas_small_bytes;
    for (auto& w : work)
      if (!w.p) w.alloc(E.ctx, (size_t)ld_);
    bind_buffers();
  }
  // Everything up front: growing ctx->d_h in the middle of a pass would free the pending coefficients.  Per parity of k:
  // g (coefficients) and, t_off further, t (lagged_fold_kernel); then the device copy of alpha / beta and the locked
  // eigenvalues.  Called again at the start of every pass: a two-sweep iteration with more than ~7000 coefficient
  // columns (Engine::orth) may have grown, i.e. moved, ctx->d_h since.
  void bind_buffers() {
    constexpr size_t R = (size_t)Engine<T>::R;
    t_off = (size_t)kLaggedMaxCols + 8;
    const size_t half = 2 * t_off + 2 * R + 8;
    E.ctx->ensure_h(2 * half + 2 * t_off + (size_t)kLaggedMaxLocked);
    hbuf[0] = E.ctx->d_h;
    hbuf[1] = E.ctx->d_h + half;
    hist_alpha = E.ctx->d_h + 2 * half;
    hist_beta = hist_alpha + t_off;
    d_lambda = hist_beta + t_off;
  }
  // a new Lanczos pass: k restarts at 1.  The compensation of the lagged form needs the image under the operator of every
  // vector it orthogonalises against: the recurrence for the Lanczos vectors, lambda_i z_i for a locked EIGENvector
  // (lambda_shifted: eigenvalues of the operator the loop applies, i.e. including eigenvalue_offset).  A caller's
  // arbitrary orthogonalizeTo vectors (run_iteration) have no such relation: lambda_shifted = nullptr keeps the
  // two-sweep form for that pass.
  // What the compensation neglects for a locked column is c_z r with r = A z - lambda z and c_z ~ ||r|| / beta, i.e. the
  // SQUARE of the locked vector's residual: it is measured here (one operator application per locked vector and pass) and
  // the pass takes the one-sweep form only if every ||r_i|| <= 3e-8 max|lambda| (effect on the recurrence ~ 1e-15 max|lambda|
  // at a typical beta).  Ritz vectors of clustered or degenerate eigenvalues, or of a pass cut off by max_iteration, do not meet
  // that and keep the two-sweep form.  All numbers are all-reduced: the same decision on every rank.
  // norm_scale: a rank-independent estimate of the OPERATOR's size (the previous passes' ||T||_inf, lanczos_run: at least
  // ||A + offset||_2 restricted to the Krylov space, at most 3 x ||A + offset||_2 — so "3e-8 scale" below means at most
  // 9e-8 ||A + offset||_2 and the neglected term at most ~1e-14 ||A + offset||_2 at a typical beta): the gate is
  // relative to the OPERATOR's size, not to max|lambda + offset|, which collapses when a locked eigenvalue sits near -offset.
  void begin_pass(const T* locked_vecs, int64_t n_lock, const double* lambda_shifted = nullptr, double offset = 0.0,
                  double norm_scale = 0.0) {
    locked = locked_vecs;
    n_locked = n_lock;
    lag_pending = false;
    pair_pending = false;
    pair_allowed = true;
    for (auto& b : slot_pair) b = false;
    // (lambda_shifted == nullptr with locked vectors — a caller's orthogonalizeTo list, run_iteration LL:216-220,259: their Rayleigh
    // quotients theta_i = <z_i, (A + offset) z_i> are MEASURED below and take the eigenvalues' place; the residual gate then decides
    // whether the list consists of eigenvectors to the accuracy the one-sweep forms need)
    lag_ok = lagged && n_lock <= kLaggedMaxLocked;
    lag_beta2_min = 0.0;
    if (lag_ok) vtab_begin_pass();
    if (lagged) {
      bind_buffers();
      // The partial sums of the sweeps — one column per coefficient, kMaxGrid rows — are sized HERE for the longest basis this pass can
      // reach (max_k_hint: max_iteration; the column limits of the one-sweep forms bound it): growing them in the loop means a
      // hipFree, i.e. a device synchronisation, plus a hipMalloc a dozen times in a run's first call on a context (geometric growth
      // up to 600 columns for config 3's 301 iterations) — 0.5-0.7 s of the 1.15-1.37 s that call took on some boxes of round 5,
      // against 0.62 s for the second call.
      constexpr size_t R = (size_t)Engine<T>::R;
      const size_t reach = (size_t)std::max<int64_t>(0, std::min<int64_t>(max_k_hint, (int64_t)kLaggedMaxCols)) + (size_t)n_lock + 2;
      const size_t cols = std::min<size_t>(kPresizeCols, 2 * R * reach + 5 * R + 1);  // (longer runs: powers of two, enqueue_pair)
      E.ctx->ensure_partials((size_t)kMaxGrid * cols);
    }
    if (!lag_ok || n_lock == 0) return;
    const bool measure_theta = lambda_shifted == nullptr;
    if (!measure_theta) {
      LL_HIP(hipMemcpyAsync(d_lambda, lambda_shifted, (size_t)n_lock * sizeof(double), hipMemcpyHostToDevice, s));
      LL_HIP(hipStreamSynchronize(s));  // (pageable source: the caller's array may go away)
    }
    BasisSegs<T> none;
    none.nseg = 0;
    none.ld = ld;
    E.ctx->ensure_partials(kMaxGrid);
    double* r2_dev = hbuf[0];  // free until the first iteration of the pass: ||A z_i - lambda_i z_i||^2, i < n_lock
    for (int64_t i = 0; i < n_lock; ++i) {
      const T* z = locked + i * ld;
      T* y = work[0].p;
      E.apply(z, y, offset, measure_theta ? d_lambda + i : nullptr, true);  // (theta_i = Re <z_i, y>: the fused dot of the operator kernel)
      const ThreeTerm<T> tt{nullptr, z, d_lambda + i, NormRefs{nullptr, nullptr, nullptr, 0}};  // y <- y - lambda_i z, ||y||^2
      const int grid = launch_mdot<T>(nl, y, none, tt, nullptr, E.ctx->d_partials, small_bytes, s);
      launch_reduce_cols(E.ctx->d_partials, grid, 1, r2_dev + i, nullptr, s);
    }
    E.all_reduce(r2_dev, (size_t)n_lock);  // one collective and one fetch for all locked vectors
    std::vector<double> r2((size_t)n_lock), theta;
    E.fetch(r2_dev, r2.data(), (size_t)n_lock);
    if (measure_theta) {
      theta.resize((size_t)n_lock);
      E.fetch(d_lambda, theta.data(), (size_t)n_lock);
      lambda_shifted = theta.data();
    }
    double scale = norm_scale, worst = 0.0;
    for (int64_t i = 0; i < n_lock; ++i) {
      worst = std::max(worst, std::sqrt(std::max(r2[(size_t)i], 0.0)));
      scale = std::max(scale, std::fabs(lambda_shifted[i]));
    }
    if (!(worst <= 3e-8 * scale)) {
      lag_ok = false;
      return;
    }
    // c_z ~ ||r|| / beta: the neglected term is <= ||r||^2 / beta; below this beta^2 it would exceed 1e-13 scale and the
    // loop leaves the one-sweep form for the rest of the pass (lanczos_run, collect)
    const double bmin = worst * worst / (1e-13 * scale);
    lag_beta2_min = bmin * bmin;
  }
  // iteration j ended with a beta too small for the first-order treatment of the locked columns: complete u_j with the
  // two-sweep kernels (unless the speculative sweep already has) and continue in the two-sweep form
  void leave_lagged(int64_t j) {
    make_final(j);
    lag_pending = false;
    pair_pending = false;
    lag_ok = false;
  }
  // beta_j changed on the host (second Gram-Schmidt pass on u_{j+1})
  void set_beta(int64_t j, double value) {
    if (lag_ok) launch_set_scalar(hist_beta + j, value, s);
  }
  RunList<T> basis_runs(int64_t count) {  // locked vectors, then u_0 .. u_{count-1}
    RunList<T> runs;
    runs.ld = ld;
    runs.add(locked, n_locked);
    runs.add_basis(U, count);
    return runs;
  }
  // u_{lag_k} = (w - U g) / beta with the two-sweep kernels: the pending late update, applied now (the vector is needed
  // complete: a second Gram-Schmidt pass on it, or the loop leaves the lagged form)
  void flush_lag() {
    if (!lag_pending) return;
    T* dst = U.vec(lag_k);
    LL_HIP(hipMemcpyAsync(dst, work[lag_k & 1].p, (size_t)nl * sizeof(T), hipMemcpyDeviceToDevice, s));
    const RunList<T> runs = basis_runs(lag_k);
    int off = 0;
    for (auto& g : runs.groups(max_vecs_per_launch<T>())) {
      launch_maxpy<T>(nl, dst, g, hbuf[lag_k & 1] + Engine<T>::R * off, nullptr, E.ctx->d_partials, small_bytes, s);
      for (int i = 0; i < g.nseg; ++i) off += g.count[i];
    }
    const NormRefs nr{lag_c1, lag_c1, lag_c1, 0};
    launch_scale<T>(nl, dst, 0.0, &nr, s);
    lag_pending = false;
  }
  // u_j must be complete in its basis slot (second Gram-Schmidt pass on it)
  void make_final(int64_t j) {
    if (pair_pending && j >= pair_P) pair_flush(j + 1);
    if (lag_pending && lag_k == j) flush_lag();
  }
  // Leave the pair form: complete the two pending vectors with their measured coefficients (two-sweep kernels).  Afterwards
  // u_0 .. u_{P+1} are complete, nothing is pending, and iteration P + 2 can be enqueued from a clean state.
  // count: only the vectors u_j with j < count are needed (end of a pass: the Ritz vectors use u_0 .. u_{m-1}; a repair of u_j:
  // nothing behind u_j survives it) — a pending vector beyond that is dropped instead of completed.
  void pair_flush(int64_t count = std::numeric_limits<int64_t>::max()) {
    if (!pair_pending) return;
    constexpr int R = Engine<T>::R;
    const int64_t P = pair_P;
    const T* src[2] = {pr1, pr2};
    const double* coef[2] = {g1p, g2p};  // g2p: R * P coefficients against the basis, then <u_P, pr2>: one contiguous list
    const double* rho[2] = {rho1p, rho2p};
    for (int v = 0; v < 2 && P + v < count; ++v) {
      T* dst = U.vec(P + v);
      if (dst != src[v]) LL_HIP(hipMemcpyAsync(dst, src[v], (size_t)nl * sizeof(T), hipMemcpyDeviceToDevice, s));
      const RunList<T> runs = basis_runs(P + v);
      int off = 0;
      for (auto& g : runs.groups(max_vecs_per_launch<T>())) {
        launch_maxpy<T>(nl, dst, g, coef[v] + R * off, nullptr, E.ctx->d_partials, small_bytes, s);
        for (int i = 0; i < g.nseg; ++i) off += g.count[i];
      }
      const NormRefs nr{rho[v], rho[v], rho[v], 0};
      launch_scale<T>(nl, dst, 0.0, &nr, s);
    }
    pair_pending = false;
    lag_pending = false;
    refs_prev = NormRefs{rho2p, rho2p, rho2p, 0};  // beta^2 of the last completed vector, for the next three-term update
  }
  // End of a pass with a pair pending: the Ritz vectors need u_0 .. u_{count-1}, of which u_P (and u_{P+1}) exist only as raw
  // vectors with their measured coefficients.  Instead of completing them with a sweep of their own (pair_flush: the whole basis
  // read once per pending vector — 1.4 ms of a 131 ms step on config 3), the caller folds the late update into the COEFFICIENTS of the
  // Ritz GEMV:  u_P = (r1 - S g1) / rho1,  u_{P+1} = (r2 - S g2 - gam u_P) / rho2  =>  sum_k s_k u_k is a combination of S, r1, r2.
  struct PairTail {
    bool active = false;
    int64_t P = 0;            // Lanczos vectors complete in the basis
    int nvec = 0;             // pending vectors the result needs (1: u_P; 2: u_P and u_{P+1})
    const T* src[2] = {nullptr, nullptr};
    const double* g[2] = {nullptr, nullptr};     // reals * K coefficients each; g[1] is followed by gam (reals)
    const double* rho2[2] = {nullptr, nullptr};  // squared norms of the orthogonal parts
  };
  PairTail take_tail(int64_t count) {
    PairTail t;
    if (!pair_pending) return t;
    const int nv = (int)std::max<int64_t>(0, std::min<int64_t>(2, count - pair_P));
    pair_pending = false;
    lag_pending = false;
    if (nv == 0) return t;
    t.active = true;
    t.P = pair_P;
    t.nvec = nv;
    t.src[0] = pr1;
    t.src[1] = pr2;
    t.g[0] = g1p;
    t.g[1] = g2p;
    t.rho2[0] = rho1p;
    t.rho2[1] = rho2p;
    return t;
  }
  // Iterations k and k + 1 in the pair form.  Entered from the one-sweep state (iteration k - 1 pending with its measured
  // coefficients: u_{k-2} plays the part of an already complete first vector, g1 = 0, rho1 = 1) or continued from a pair.
  bool enqueue_pair(int64_t k, double offset) {
    constexpr int R = Engine<T>::R;
    // (restart passes: lag_ok already says that the locked vectors are eigenvectors to the one-sweep form's gate, begin_pass)
    if (!pair_enabled || !pair_allowed || !lag_ok) return false;
    // never beyond the loop's max_iteration: iteration k + 1 would be an operator application the caller did not ask for, and with
    // max_iteration == n its input is the normalised remainder of a vanishing vector; the last odd iteration runs in the one-sweep form
    if (max_k_hint > 0 && k + 1 > max_k_hint) return false;
    const int64_t Lk = n_locked;
    int64_t P;
    const T *r1, *r2;
    const double *g1, *g2, *rho1sq, *rho2sq;
    int out_set, out_rec;
    if (pair_pending) {
      if (pair_P + 2 != k) return false;
      P = pair_P;
      r1 = pr1;
      r2 = pr2;
      g1 = g1p;
      g2 = g2p;
      rho1sq = rho1p;
      rho2sq = rho2p;
      out_set = pset ^ 1;
      out_rec = prec_set ^ 1;
    } else if (lag_pending && lag_k == k - 1 && k >= 3) {
      P = k - 2;
      r1 = U.vec(k - 2);
      r2 = work[(k - 1) & 1].p;
      g1 = pzero;
      g2 = hbuf[(k - 1) & 1];   // L + k - 1 = K + 1 coefficients: against the locked vectors and u_0 .. u_{P-1}, then <u_P, r2>
      rho1sq = pscal + 0;
      rho2sq = lag_c1;
      out_set = 1;
      out_rec = 0;
    } else {
      return false;
    }
    const int64_t K = Lk + P;  // stored columns of the sweep
    const int ncols = 2 * R * (int)K + 5 * R + 1;
    const int64_t stream_bytes = std::min<int64_t>(small_bytes, (int64_t)1 << 20);
    const int64_t len = E.ctx->comm != nullptr ? E.op->n_shard : nl;  // (sharded: decided on the shard stride, the same on every rank)
    // the coefficient records hold reals * (K + 2) (+ reals) numbers, the recorded tridiagonal kLaggedMaxCols + 8 entries; the
    // sweep's 2 reals K + 5 reals + 1 columns are summed in as many launches as one workgroup's LDS asks for (pair_sweep_max_vecs)
    // below the streaming geometry, down to the one-sweep form's lower limit (320 KiB by default), the sweep runs in the small-vector
    // geometry (pair_small_kernel: four waves per 1 KiB strip split the stored vectors), one launch, as many columns as 64 KiB of LDS hold
    // (Laplacian, window 100, it/s with / without the pair form: n = 5.0e4 (401 KB) 25.8 k / 26.3 k, n = 1.0e5 (800 KB) 24.3 k / 20.9 k:
    // seven launches per pair against four per iteration, half the basis traffic — the pair form takes over from 512 KiB)
    const int64_t min_default = std::min<int64_t>(stream_bytes, kPairSmallMinBytes);
    const int64_t min_bytes = E.ctx->tune.lagged_min_bytes >= 0 ? std::min<int64_t>(E.ctx->tune.lagged_min_bytes, stream_bytes) : min_default;
    const bool small_geometry = len * (int64_t)sizeof(T) < stream_bytes;
    if ((int64_t)R * (K + 8) > kLaggedMaxCols || len * (int64_t)sizeof(T) < min_bytes) return false;
    if (small_geometry && (!pair_small_fits<T>((int)K) || K > max_vecs_per_launch<T>() || basis_runs(P).runs.size() > (size_t)kMaxSegs)) return false;
    if (E.ctx->tune.pair_max_stored > 0 && K > E.ctx->tune.pair_max_stored) return false;  // (test hook: the hand-over to the one-sweep form)
    const RunList<T> stored = basis_runs(P);
    int per_launch = pair_sweep_max_vecs<T>();
    if (E.ctx->tune.pair_split_vecs > 0) per_launch = std::min(per_launch, std::max(1, E.ctx->tune.pair_split_vecs));
    const std::vector<BasisSegs<T>> groups = stored.groups(per_launch);
    if (!small_geometry && groups.size() > 1 && !psplit.p) psplit.alloc(E.ctx, (size_t)ld);
    const double te0 = now_s();
    T* r3 = out_set ? pwork[0].p : work[0].p;
    T* r4 = out_set ? pwork[1].p : work[1].p;
    double* rec3 = prec[2 * out_rec];
    double* rec4 = prec[2 * out_rec + 1];
    double* nxt = pscal + 8 + 2 * out_rec;
    double* t3 = pscal + 16;  // |r3|^2, <r1, r3>
    const double* gam = g2 + R * K;
    const int sa = (int)(k % 4), sb = (int)((k + 1) % 4);
    double* e1 = E.S(kScalAlpha + sa);
    double* e2 = E.S(kScalAlpha + sb);
    want_partial_cols((size_t)std::max(ncols, 1 + R));
    // ---- iteration k: operator on r2 / rho2, three-term with raw vectors
    timer.mark();
    typename Engine<T>::DeferredAlpha da1, da2;
    E.apply(r2, r3, offset, e1, true, fuse_launches ? &da1 : nullptr, nullptr, rho2sq);
    timer.mark();
    // Where the second operator kernel reads x itself (CSR-stream, lattice, dense on one GPU) it folds the three-term kernel's
    // partial sums of |r3|^2 on the fly (ScaleIn, like the deferred normalisation of 3.3) and pair_predict_kernel folds <r1, r3>:
    // no fold launch in between.  The PB / tiled kernels and sharded contexts want the folded scalar (all-reduced).
    const bool fold_in_consumers = fuse_launches && E.can_defer_scale();
    int grid = launch_pair_three_term<T>(nl, r3, r2, r1, e1, da1.nparts > 0 ? da1.partials : nullptr, da1.nparts, rho2sq, rho1sq,
                                         E.ctx->d_partials, fold_in_consumers, s);
    const int tt_grid = grid;
    if (!fold_in_consumers) {
      launch_reduce_cols(E.ctx->d_partials, grid, 1 + R, t3, nullptr, s);
      if (E.ctx->comm != nullptr) E.all_reduce(t3, (size_t)(1 + R));  // |r3|^2 and <r1, r3> over the shards
    }
    timer.mark();
    // ---- iteration k + 1: operator on r3 / |r3|; its three-term update is formed inside the sweep
    timer.mark();
    if (fold_in_consumers) {
      ScaleIn<T> sc;
      sc.partials = E.ctx->d_partials;  // column 0: |r3|^2 per workgroup
      sc.nparts = tt_grid;
      sc.c1_out = t3;                   // the folded |r3|^2, for the predict / sweep / fold kernels
      E.apply(r3, r4, offset, e2, true, &da2, &sc, nullptr);
    } else {
      E.apply(r3, r4, offset, e2, true, fuse_launches ? &da2 : nullptr, nullptr, t3);
    }
    timer.mark();
    // ---- one sweep for both
    launch_pair_predict((int)P, (int)Lk, R, g1, g2, rho1sq, rho2sq, gam, t3, fold_in_consumers ? E.ctx->d_partials : nullptr, tt_grid,
                        e1, e2, da2.nparts > 0 ? da2.partials : nullptr, da2.nparts, hist_alpha, hist_beta, d_lambda, pp3, pp4, s);
    T* const uP = U.vec(P);
    T* const uQ = U.vec(P + 1);  // (may add a slab: the pointer table is brought up to date after it)
    if (small_geometry) {
      BasisSegs<T> none;
      none.nseg = 0;
      none.ld = ld;
      const std::vector<BasisSegs<T>> one = stored.groups(max_vecs_per_launch<T>());  // a single group (checked above)
      LL_REQUIRE(launch_pair_sweep_small<T>(nl, one.empty() ? none : one[0], (int)K, r1, r2, r3, r4, uP, uQ, g1, g2, gam, pp4, rho1sq, rho2sq,
                                            e2, t3, E.ctx->d_partials, &grid, s),
                 "internal: the small-geometry pair sweep refused a launch that was checked to fit");
    } else {
      grid = launch_pair_sweep<T>(nl, groups, (int)K, r1, r2, r3, r4, uP, uQ, psplit.p, g1, g2, gam, pp4, rho1sq, rho2sq, e2, t3,
                                  E.ctx->d_partials, E.ctx->tune.lagged_pieces, s, vtab_sync(), E.ctx->tune.sweep_pipeline >= 2);
    }
    launch_reduce_cols(E.ctx->d_partials, grid, ncols, pcols, nullptr, s);
    // sharded: ONE all-reduce carries both iterations' columns; every rank then folds the same numbers to the same bits
    if (E.ctx->comm != nullptr) E.all_reduce(pcols, (size_t)ncols);
    launch_pair_fold(pcols, (int)P, (int)Lk, R, d_lambda, pp4, g2, gam, rho2sq, t3, e1, e2, rec3, rec4, nxt, hist_alpha, hist_beta, pfold,
                     E.ctx->h_pinned + 4 * sa, E.ctx->h_pinned + 4 * sb, E.ctx->h_pinned + 16 + sa, E.ctx->h_pinned + 16 + sb, s,
                     E.ctx->tune.event_in_launch ? ring.ev[sb] : nullptr);
    // ONE event for both iterations of the pair (their scalars are published by the same fold kernel): every event record is a marker
    // packet between two dependent kernels of a loop that is bound by exactly those gaps at small sizes
    if (!E.ctx->tune.event_in_launch) LL_HIP(hipEventRecord(ring.ev[sb], s));
    ev_of_slot[sa] = ev_of_slot[sb] = sb;
    timer.mark();
    slot_pair[sa] = slot_pair[sb] = true;
    pair_pending = true;
    pair_P = P + 2;
    pr1 = r3;
    pr2 = r4;
    pset = out_set;
    prec_set = out_rec;
    g1p = rec3;
    g2p = rec4;
    rho1p = nxt;
    rho2p = nxt + 1;
    lag_pending = false;
    n_pair += 2;
    n_lagged += 2;  // (the pair form is a one-sweep form: ll_run_stats.lagged_iterations counts it, pair_iterations singles it out)
    t_enqueue += now_s() - te0;
    return true;
  }
  bool enqueue_lagged(int64_t k, double offset, int64_t nb_total) {
    constexpr int R = Engine<T>::R;
    if (!lag_ok) return false;
    const RunList<T> in_memory = basis_runs(lag_pending ? k - 1 : k);  // u_{k-1} is not in memory while its update is pending
    const std::vector<BasisSegs<T>> groups = in_memory.groups(max_vecs_per_launch<T>());
    // (very short vectors keep the two-sweep form of the small-vector kernels; sharded: decided on the shard stride, the same
    // on every rank).  The one sweep of the streaming geometry overtakes the two small-vector sweeps from about 1 MiB per
    // vector, well below the 4 MiB at which the streaming two-sweep kernels do (Laplacian, window 100: n = 2.0e5 14.3 ->
    // 15.4 k it/s, 3.6e5 10.9 -> 14.0 k, 5.0e5 8.5 -> 12.4 k; n = 1.0e5 would lose 5 %; profiles/r03_small_vector_kernel_gaps.txt)
    const int64_t len = E.ctx->comm != nullptr ? E.op->n_shard : nl;
    const int64_t stream_bytes = std::min<int64_t>(small_bytes, (int64_t)1 << 20);  // from here the streaming geometry
    // ... and below it, down to 320 KiB, the one-sweep kernel of the small-vector geometry (lagged_small_kernel): four
    // launches per iteration against the three of the two small-vector sweeps, but one pass over the basis: Laplacian,
    // window 100: n = 5.0e4 20.8 -> 25.8 k it/s, 1.0e5 18.3 -> 21.0 k; n = 3.0e4 26.3 -> 25.3 k and n = 1e4 26.4 -> 19.1 k
    // would lose (profiles/r03_small_vector_kernel_gaps.txt)
    const int64_t min_default = std::min<int64_t>(stream_bytes, (int64_t)320 << 10);
    const int64_t min_bytes = E.ctx->tune.lagged_min_bytes >= 0 ? std::min<int64_t>(E.ctx->tune.lagged_min_bytes, stream_bytes)
                                                                  : min_default;
    if (nb_total != k + n_locked || R * nb_total > kLaggedMaxCols || groups.size() > 1 || len * (int64_t)sizeof(T) < min_bytes) {
      lag_ok = false;  // for the rest of the pass: the two-sweep iterations do not record T on the device
      return false;
    }
    const double te0 = now_s();
    const int slot = (int)(k % 4);
    slot_pair[slot] = false;
    T* y = work[k & 1].p;
    const T* x = lag_pending ? work[(k - 1) & 1].p : U.vec(k - 1);
    timer.mark();
    typename Engine<T>::DeferredAlpha da;
    E.apply(x, y, offset, E.S(kScalAlpha + slot), true, fuse_launches ? &da : nullptr, nullptr, lag_pending ? lag_c1 : nullptr);
    timer.mark();
    ThreeTerm<T> tt{k > 1 ? U.vec(k - 2) : nullptr, U.vec(k - 1), E.S(kScalAlpha + slot), refs_prev};
    if (da.nparts > 0) {
      tt.alpha_partials = da.partials;
      tt.alpha_nparts = da.nparts;
      tt.alpha_out = E.S(kScalAlpha + slot);
    }
    const int ncols = R * (int)nb_total + 1;
    want_partial_cols((size_t)ncols);
    BasisSegs<T> none;
    none.nseg = 0;
    none.ld = ld;
    int grid;
    if (lag_pending) {
      const Lagged<T> lg{work[(k - 1) & 1].p, U.vec(k - 1), hbuf[(k - 1) & 1], hbuf[(k - 1) & 1] + t_off, lag_c1};
      grid = launch_lagged<T>(nl, y, groups.empty() ? none : groups[0], lg, tt, E.ctx->d_partials, E.ctx->tune.lagged_pieces,
                              stream_bytes, s);
      ++n_lagged;
    } else {
      grid = launch_mdot<T>(nl, y, groups.empty() ? none : groups[0], tt, nullptr, E.ctx->d_partials, small_bytes, s);
    }
    double* c = E.S(kScalNorms + 3 * slot);
    double* hb = hbuf[k & 1];
    const double* c0 = c;
    if (E.ctx->comm == nullptr) {
      launch_reduce_cols(E.ctx->d_partials, grid, ncols, hb, c, s);  // coefficients -> hb, ||w||^2 -> c[0]
    } else {  // one all-reduce for the coefficients and ||w||^2; every rank then folds the same numbers to the same bits
      launch_reduce_cols(E.ctx->d_partials, grid, ncols, hb, nullptr, s);
      E.all_reduce(hb, (size_t)ncols);
      c0 = hb + R * nb_total;
    }
    const double* pg = lag_pending ? hbuf[(k - 1) & 1] : nullptr;
    launch_lagged_fold(hb, (int)nb_total, (int)n_locked, R, hb + t_off, c0, c, c + 1, E.S(kScalAlpha + slot), pg,
                       pg ? pg + t_off : nullptr, lag_c1, hist_alpha, hist_beta, d_lambda, E.ctx->h_pinned + 4 * slot, s,
                       E.ctx->tune.event_in_launch ? ring.ev[slot] : nullptr);
    ev_of_slot[slot] = slot;
    if (!E.ctx->tune.event_in_launch) LL_HIP(hipEventRecord(ring.ev[slot], s));
    timer.mark();
    lag_pending = true;
    lag_k = k;
    lag_c1 = c + 1;
    refs_prev = NormRefs{c, c + 1, c + 1, 0};
    t_enqueue += now_s() - te0;
    return true;
  }
  void enqueue(int64_t k, double offset, const RunList<T>& runs, int mode) {
    pair_flush();  // (a pending pair is completed first: the forms below start from complete vectors)
    if (mode == LL_ORTH_CGS_DGKS && !pending && enqueue_lagged(k, offset, runs.total())) return;
    flush_lag();  // (leaving the lagged form: u_{k-1} must be complete)
    lag_ok = false;
    const double te0 = now_s();
    const int slot = (int)(k % 4);
    slot_pair[slot] = false;
    const T* x = U.vec(k - 1);
    T* y = defer ? work[k & 1].p : U.vec(k);
    ScaleIn<T> sc;
    if (pending) {  // u_{k-1} is still w_{k-1} in its work buffer: this operator kernel normalises it on the fly
      x = work[(k - 1) & 1].p;
      sc.partials = pend.partials;
      sc.nparts = pend.nparts;
      sc.c1_out = pend.c1;
      sc.alpha = pend.alpha;
      sc.c0 = pend.c0;
      sc.host = pend.host;
      sc.u_out = U.vec(k - 1);
    }
    timer.mark();
    typename Engine<T>::DeferredAlpha da;
    // the operator kernel that publishes iteration k-1's scalars completes that iteration's event itself where its launcher can
    // (LL_LAUNCH_STOP: no marker packet between it and the sweep's first kernel); otherwise the event is recorded behind it
    const bool ev_in_launch = pending && E.ctx->tune.event_in_launch;
    if (ev_in_launch) E.ctx->stop_next = ring.ev[pend_slot];
    E.apply(x, y, offset, E.S(kScalAlpha + slot), true, fuse_launches ? &da : nullptr, pending ? &sc : nullptr);  // P0-P3
    if (pending) {
      ev_of_slot[pend_slot] = pend_slot;
      if (!ev_in_launch || E.ctx->stop_next != nullptr) LL_HIP(hipEventRecord(ring.ev[pend_slot], s));  // iteration k-1's scalars are on their way to the host
      E.ctx->stop_next = nullptr;
      pending = false;
    }
    timer.mark();
    ThreeTerm<T> tt{k > 1 ? U.vec(k - 2) : nullptr, U.vec(k - 1), E.S(kScalAlpha + slot), refs_prev};  // P4
    if (da.nparts > 0) {  // the multi-dot folds alpha itself
      tt.alpha_partials = da.partials;
      tt.alpha_nparts = da.nparts;
      tt.alpha_out = E.S(kScalAlpha + slot);
    }
    typename Engine<T>::Publish pub{E.ctx->h_pinned + 4 * slot, E.S(kScalAlpha + slot), false};
    pub.can_defer = fuse_launches;
    const NormRefs refs = E.orth(y, runs, mode, tt, E.S(kScalNorms + 3 * slot), nullptr, true, &pub);  // P5-P7
    if (pub.deferred && defer) {  // P8 rides in the next operator kernel
      pending = true;
      pend = pub;
      pend_slot = slot;
      pend_k = k;
    } else if (pub.deferred) {  // norm fold + publish + normalisation in one launch (P8)
      launch_scale_publish<T>(nl, y, pub.partials, pub.nparts, pub.c1, pub.alpha, pub.c0, pub.host, s);
      ev_of_slot[slot] = slot;
      LL_HIP(hipEventRecord(ring.ev[slot], s));
    } else if (pub.derive) {  // sharded: derived norm + publish + normalisation in one launch
      launch_scale_derive<T>(nl, y, pub.derive_c0, pub.derive_h, pub.derive_count, pub.c0_out, pub.c1, pub.alpha, pub.host, s);
      ev_of_slot[slot] = slot;
      LL_HIP(hipEventRecord(ring.ev[slot], s));
    } else {
      LL_REQUIRE(!defer, "internal: deferred normalisation needs the fused norm fold");
      if (!pub.done) launch_publish(pub.host, pub.alpha, refs, s);
      ev_of_slot[slot] = slot;
      LL_HIP(hipEventRecord(ring.ev[slot], s));
      launch_scale<T>(nl, y, 0.0, &refs, s);  // P8
    }
    timer.mark();
    refs_prev = refs;
    t_enqueue += now_s() - te0;
  }
  // the pending iteration is the last one: normalise it into its basis slot and publish its scalars now
  void flush() {
    if (!pending) return;
    launch_scale_publish<T>(nl, U.vec(pend_k), pend.partials, pend.nparts, pend.c1, pend.alpha, pend.c0, pend.host, s,
                            work[pend_k & 1].p);
    ev_of_slot[pend_slot] = pend_slot;
    LL_HIP(hipEventRecord(ring.ev[pend_slot], s));
    pending = false;
  }
};

template <typename T> void default_init(T* v, int64_t n);
// LL:70-104: std::random_device-seeded mt19937, uniform [-1,1]; complex: both parts.
template <> void default_init<double>(double* v, int64_t n) {
  std::random_device dev;
  std::mt19937 mt(dev());
  std::uniform_real_distribution<double> r(-1.0, 1.0);
  for (int64_t i = 0; i < n; ++i) v[i] = r(mt);
}
template <> void default_init<float>(float* v, int64_t n) {
  std::random_device dev;
  std::mt19937 mt(dev());
  std::uniform_real_distribution<float> r(-1.0f, 1.0f);
  for (int64_t i = 0; i < n; ++i) v[i] = r(mt);
}
template <> void default_init<cf>(cf* v, int64_t n) {
  std::random_device dev;
  std::mt19937 mt(dev());
  std::uniform_real_distribution<float> r(-1.0f, 1.0f);
  for (int64_t i = 0; i < n; ++i) {
    v[i].re = r(mt);
    v[i].im = r(mt);
  }
}
template <> void default_init<zc>(zc* v, int64_t n) {
  std::random_device dev;
  std::mt19937 mt(dev());
  std::uniform_real_distribution<double> r(-1.0, 1.0);
  for (int64_t i = 0; i < n; ++i) {
    v[i].re = r(mt);
    v[i].im = r(mt);
  }
}

inline double as_real_coeff(double v, double*) { return v; }
inline zc as_real_coeff(double v, zc*) { return zc{v, 0.0}; }
inline float as_real_coeff(double v, float*) { return (float)v; }
inline cf as_real_coeff(double v, cf*) { return cf{(float)v, 0.0f}; }

// Vectors per basis slab.  The reference's initial_vector_size (LL:181, default 200) only RESERVES the outer
// std::vector; its Lanczos vectors are allocated one by one.  Here a slab is one hipMalloc, so it is capped by BYTES
// (4 GiB, LL_SLAB_BYTES overrides): a run that converges after 30 iterations of an n = 1e8 problem must not need
// 200 vectors of HBM up front.  Slabs are appended on demand and cached in the context between runs.
int64_t pick_chunk_vecs(int64_t initial_vector_size, int64_t max_iteration, int64_t vec_bytes, int64_t cap_bytes) {
  int64_t want = initial_vector_size > 0 ? initial_vector_size : 200;
  want = std::min(want, max_iteration + 2);
  want = std::min(want, cap_bytes / std::max<int64_t>(vec_bytes, 1));
  return std::max<int64_t>(want, 4);
}

}  // namespace

// Bytes of one Krylov-basis slab of a run with default parameters on this operator (initial_vector_size = 200, max_iteration = n):
// what operator creation sizes its spare placement candidates to, so that they can serve as the first basis slabs (capi.cpp).
int64_t default_slab_bytes(int64_t n, int64_t n_local, int64_t n_shard, int elem_bytes, const Tuning& tune) {
  const int64_t ld = round_up(std::max(n_local, n_shard), 256);
  const int64_t vec_bytes = ld * (int64_t)elem_bytes;
  return pick_chunk_vecs(200, std::max<int64_t>(n, 1), vec_bytes, tune.slab_bytes) * vec_bytes;
}

// ================================================================= LambdaLanczos<T>::run
template <typename T>
void lanczos_run(ll_context* ctx, ll_operator* op, const ll_lanczos_params& P_in, double* eigvals, T* eigvecs,
                 int64_t* n_found, int64_t* iter_counts, int64_t iter_cap, double* alpha_out, double* beta_out,
                 ll_run_stats* stats, const IterationSpec<T>* spec) {
  ll_lanczos_params P = P_in;
  // ll_lanczos_params_default() fills in the DOUBLE tolerance (LL:150 with real_t<T> = double); the reference scales it
  // with the epsilon of real_t<T>, so a float run that was left at that default gets 1e3 * FLT_EPSILON instead of a
  // tolerance float data can never meet (which would run to max_iteration = n).
  if (sizeof(typename scalar_traits<T>::real) == 4 && P.eps == std::numeric_limits<double>::epsilon() * 1e3)
    P.eps = (double)std::numeric_limits<float>::epsilon() * 1e3;
  LL_REQUIRE(op && op->ctx == ctx, "operator belongs to another context");
  LL_REQUIRE(op->is_complex == scalar_traits<T>::is_complex && op->elem_bytes == (int)sizeof(T),
             "operator scalar type mismatch");
  LL_REQUIRE(P.matrix_size == op->n, "matrix_size differs from the operator dimension");
  LL_REQUIRE(P.num_eigs >= 1 && P.num_eigs <= P.matrix_size, "num_eigs out of range");
  if (spec) {
    LL_REQUIRE(spec->nroot >= 1 && spec->nroot <= P.matrix_size, "nroot out of range");
    LL_REQUIRE(spec->n_orth >= 0 && (spec->n_orth == 0 || spec->orth_host != nullptr), "bad orthogonalizeTo list");
  }
  LL_REQUIRE(P.max_iteration >= 1, "max_iteration must be >= 1");
  LL_REQUIRE(P.num_eigs_per_iteration >= 1, "num_eigs_per_iteration must be >= 1");
  LL_HIP(hipSetDevice(ctx->device));
  const double t_start = now_s();
  hipStream_t s = ctx->stream;
  const int64_t n = op->n, nl = op->n_local;
  const int64_t ld = round_up(std::max(nl, op->n_shard), 256);
  const int mode = P.orth_mode;
  const double dgks_thr = ctx->tune.dgks_threshold;
  // Two launches per iteration less on single-GPU runs: alpha is folded by the multi-dot that needs it, and the fold of
  // the post-pass norm + the publish step ride in the normalisation kernel.  LL_FUSE_LAUNCHES=0: separate kernels (A/B).
  const bool fuse_launches = ctx->tune.fuse_launches;
  Engine<T> E(ctx, op, nl);
  constexpr int R = scalar_traits<T>::reals;

  Basis<T> U;
  U.init(ctx, nl, ld, pick_chunk_vecs(P.initial_vector_size, P.max_iteration, ld * (int64_t)sizeof(T), ctx->tune.slab_bytes));
  DevBuf<T> d_locked, d_ritz;
  int64_t d_ritz_cap = 0;
  if (spec) {
    if (spec->n_orth > 0) d_locked.alloc(ctx, (size_t)spec->n_orth * ld);
  } else if (P.num_eigs > 1) {
    d_locked.alloc(ctx, (size_t)P.num_eigs * ld);
  }
  const int64_t nroot_max = std::min<int64_t>(P.num_eigs_per_iteration, n);
  ctx->ensure_pinned(32);  // 4 ring slots of 4 scalars, then the 4 gate values of the pair form
  EventRing ring;
  PhaseTimer timer(ctx, s);

  // EigenPairManager (EPM:21-80): best num_eigs pairs, ordered by the comparator
  std::function<bool(double, double)> cmp;
  if (P.find_maximum) cmp = std::greater<double>(); else cmp = std::less<double>();
  std::multimap<double, std::vector<T>, std::function<bool(double, double)>> kept(cmp);

  int64_t passes = 0, total_iters = 0, second_passes = 0;
  double t_inf_prev = 0.0;  // max over the passes so far of ||T_m||_inf (same numbers on every rank; between 1 and 3 x ||A + offset||_2)
  double t_tridiag = 0.0, t_wait = 0.0, t_setup = 0.0, t_finish = 0.0;
  LoopState<T> LS(E, U, ring, timer, nl, s);
  LS.fuse_launches = fuse_launches;
  if (E.can_defer_scale() && fuse_launches && mode == LL_ORTH_CGS_DGKS) LS.enable_defer(ld);
  LS.max_k_hint = P.max_iteration;
  if (E.can_scale_input() && fuse_launches && ctx->tune.lagged_gs && mode == LL_ORTH_CGS_DGKS) LS.enable_lagged(ld);
  // two iterations per sweep (device operators, streaming vectors, no locked vectors; LoopState::enqueue_pair decides per iteration)
  {
    const int64_t stream_bytes = std::min<int64_t>(ctx->tune.blas_small_bytes, (int64_t)1 << 20);
    const int64_t pair_min = ctx->tune.lagged_min_bytes >= 0 ? std::min<int64_t>(ctx->tune.lagged_min_bytes, stream_bytes)
                                                             : std::min<int64_t>(stream_bytes, LoopState<T>::kPairSmallMinBytes);
    if (LS.lagged && ctx->tune.pair_gs && (ctx->comm != nullptr ? op->n_shard : nl) * (int64_t)sizeof(T) >= pair_min) LS.enable_pair();
  }
  std::vector<double> alpha, beta;
  // Pinned staging buffer owned by the context (reused across runs): the init_vector hook fills it directly and the
  // Ritz vectors land in it, so n-sized host<->device copies run at full PCIe rate and nothing n-sized is zero-filled
  // or page-faulted per call.
  T* stage = (T*)ctx->ensure_stage((size_t)std::max<int64_t>(nl, 1) * sizeof(T));
  const bool single_pair = P.num_eigs == 1 && !spec;  // one pass, one survivor: its vector goes stage -> caller directly
  bool result_in_stage = false, result_in_caller = false;
  const bool out_dev = is_device_ptr(eigvecs);
  auto to_caller = [&](T* dst, const T* src_host) {  // host -> the caller's buffer, wherever it lives
    if (out_dev) LL_HIP(hipMemcpy(dst, src_host, (size_t)nl * sizeof(T), hipMemcpyHostToDevice));
    else host_copy(dst, src_host, (size_t)nl * sizeof(T));
  };

  struct TraceFile {  // LL_ITER_TRACE
    FILE* f = nullptr;
    ~TraceFile() {
      if (f) std::fclose(f);
    }
  } trace_holder;
  if (!ctx->tune.iter_trace.empty()) trace_holder.f = std::fopen(ctx->tune.iter_trace.c_str(), "a");
  if (trace_holder.f) std::setvbuf(trace_holder.f, nullptr, _IOLBF, 0);  // line by line: the callback lines (Engine::apply) interleave in order
  FILE* const trace_file = trace_holder.f;

  while (true) {  // restart loop LL:334-354
    const int64_t nroot = spec ? spec->nroot : std::min<int64_t>(P.num_eigs_per_iteration, n - (int64_t)kept.size());  // LL:338
    const double t_pass0 = now_s();
    // ---- start vector (LL:231-234)
    if (P.init_vector_dev) {  // start vector already in HBM (copied: the caller's buffer is left untouched)
      LL_HIP(hipMemcpyAsync(U.vec(0), P.init_vector_dev, (size_t)nl * sizeof(T), hipMemcpyDeviceToDevice, s));
    } else {
      if (P.init_vector) P.init_vector(stage, nl, op->row_begin, P.init_user);
      else default_init<T>(stage, nl);
      LL_HIP(hipMemcpyAsync(U.vec(0), stage, (size_t)nl * sizeof(T), hipMemcpyHostToDevice, s));
    }
    const int64_t L = spec ? spec->n_orth : (int64_t)kept.size();
    if (spec) {
      for (int64_t j = 0; j < L; ++j)  // the caller's orthogonalizeTo, in the caller's order
        LL_HIP(hipMemcpyAsync(d_locked.p + j * ld, spec->orth_host + j * nl, (size_t)nl * sizeof(T), hipMemcpyDefault, s));  // host or device
    } else {
      int64_t j = 0;
      for (auto& kv : kept) {  // comparator order, like MapValueIterable (CM:58-74)
        LL_HIP(hipMemcpyAsync(d_locked.p + j * ld, kv.second.data(), (size_t)nl * sizeof(T), hipMemcpyHostToDevice, s));
        ++j;
      }
    }
    const ThreeTerm<T> no_tt{nullptr, nullptr, nullptr, NormRefs{nullptr, nullptr, nullptr, 0}};
    NormRefs refs0;
    if (L > 0) {
      RunList<T> lk;
      lk.ld = ld;
      lk.add(d_locked.p, L);
      refs0 = E.orth(U.vec(0), lk, mode, no_tt, E.S(kScalScratch), nullptr);  // LL:233
    } else {
      E.norm2_dev(U.vec(0), E.S(kScalScratch) + 1);
      refs0 = E.plain_norm(E.S(kScalScratch) + 1);
    }
    launch_scale<T>(nl, U.vec(0), 0.0, &refs0, s);  // LL:234
    t_setup += now_s() - t_pass0;

    // ---- the Lanczos loop (LL:240-310)
    alpha.clear();
    beta.clear();
    std::vector<double> evs, all;
    bool evs_from_qr = true;  // whether `evs` hold the values of the reference's QR arithmetic (else: bisection values)
    int64_t itern = P.max_iteration;
    bool stopped = false;
    LS.refs_prev = refs0;
    LS.pending = false;
    // One-sweep form against locked vectors: they must be eigenvectors (LoopState::begin_pass measures their residuals);
    // a caller's orthogonalizeTo list (run_iteration) is not, and keeps the two-sweep form.
    std::vector<double> locked_lambda;  // of the operator the loop applies (A + eigenvalue_offset)
    if (!spec)
      for (auto& kv : kept) locked_lambda.push_back(kv.first + P.eigenvalue_offset);
    LS.begin_pass(d_locked.p, L, locked_lambda.empty() ? nullptr : locked_lambda.data(), P.eigenvalue_offset, t_inf_prev);
    RunList<T> locked_runs;
    locked_runs.ld = ld;
    locked_runs.add(d_locked.p, L);  // P5
    // Enqueue the next iteration(s) from k on: two at once where the pair form applies (one sweep over the basis for both),
    // else one.  Returns how many.
    auto enqueue = [&](int64_t k) -> int64_t {
      if (mode == LL_ORTH_CGS_DGKS && !LS.pending && LS.enqueue_pair(k, P.eigenvalue_offset)) return 2;
      RunList<T> runs = locked_runs;
      runs.add_basis(U, k);  // P6
      LS.enqueue(k, P.eigenvalue_offset, runs, mode);
      return 1;
    };
    // Host half of iteration j, part 1 (this thread): wait for the four scalars, take the DGKS decision, append
    // alpha_j / beta_j and hand T_j to the Ritz tracker.  kRedone: a second Gram-Schmidt pass changed u_j, the
    // speculative iteration j+1 must be enqueued again.
    enum { kContinue = 0, kRedone = 2 };
    RitzTracker tracker_cfg;
    tracker_cfg.nroot = nroot;
    tracker_cfg.find_maximum = P.find_maximum != 0;
    tracker_cfg.mode = P.tridiag_mode;
    tracker_cfg.eps = P.eps;
    tracker_cfg.breakdown_tol = (double)std::numeric_limits<typename scalar_traits<T>::real>::epsilon() * 1e1;  // H3 LL:279
    // Callback operators run WITHOUT speculation: the user's mv_mul must be called exactly as often as the reference
    // calls it (LL:243: once per executed iteration) and never on the 1/sqrt(~0)-scaled vector that follows a
    // breakdown; a host callback synchronises the stream anyway, so there is nothing to overlap.
    const bool speculate = !(op->kind == ll_operator::HOST_CB || op->kind == ll_operator::DEV_CB);
    // Part 2 (H1-H4: Ritz values, breakdown, convergence) runs on a helper thread, in iteration order; this thread
    // keeps enqueuing and looks at the verdicts as they arrive, at most kMaxLag iterations late.  A verdict that
    // arrives late only means a few speculative iterations more on the device (they write basis slots the results
    // never read).  The lag is only ever used when the helper is slower than the device — in practice the O(m^2) QR
    // confirmations of LL_TRIDIAG_AUTO near convergence at large m (190 ms at m = 3300 against 9 ms per device
    // iteration at n = 1e6) — so the bound is generous; while the helper keeps up the verdicts are one iteration late
    // like before.  LL_TRIDIAG_THREAD=0 computes the verdicts inline (lag 1, the round-1 behaviour).
    const bool threaded = speculate && ctx->tune.tridiag_thread;
    const size_t kMaxLag = threaded ? 24 : 0;
    const int64_t lockstep_lag = threaded && ctx->comm != nullptr ? std::max(-1, ctx->tune.tridiag_lag) : -1;  // see StepWorker::consume
    TridiagWorker worker(tracker_cfg, threaded, ctx->tune.tridiag_test_jitter_us);
    RitzTracker::Out last;
    auto absorb = [&](RitzTracker::Out& r) {
      t_tridiag += r.seconds;
      last = std::move(r);
      return last.stop;
    };
    // the gate of the pair form: what is neglected is the SQUARE of a relative coefficient, which must stay below the rounding
    // of the storage type (float vectors carry coefficients of ~1e-6 by rounding alone)
    const double pair_gate = sizeof(typename scalar_traits<T>::real) == 4 ? 2e-4 : kPairGate;
    auto collect = [&](int64_t j) -> int {
      const int slot = (int)(j % 4);
      const double tw0 = now_s();
      LL_HIP(hipEventSynchronize(ring.ev[LS.ev_of_slot[slot]]));
      t_wait += now_s() - tw0;
      const volatile double* hp = ctx->h_pinned + 4 * slot;
      const double alpha_j = hp[0], c0_j = hp[2], c1_j = hp[3];
      double beta2_j = hp[1];
      int verdict = kContinue;
      if (mode == LL_ORTH_CGS_DGKS && c1_j < dgks_thr * c0_j) {
        // DGKS "twice is enough", decided here from the published norms: the first pass removed more than half of
        // ||w||^2, so Gram-Schmidt is repeated on u_j (already scaled to unit norm on the device) and beta_j shrinks
        // by the norm that survives.  Rare (near breakdown / deflation); costs one pipeline drain.
        if (c1_j > 0.0 && std::isfinite(c1_j)) {
          RunList<T> again;
          again.ld = ld;
          again.add(d_locked.p, L);
          again.add_basis(U, j);
          LS.make_final(j);
          beta2_j = c1_j * E.second_pass(U.vec(j), again);
          ++second_passes;
          double* cj = E.S(kScalNorms + 3 * slot);
          launch_set_scalar(cj + 1, beta2_j, s);  // what the next three-term update reads as beta_j^2
          LS.refs_prev = NormRefs{cj, cj + 1, cj + 1, 0};
          LS.pending = false;  // the speculative iteration j+1 was computed from the old u_j: it is enqueued again
          LS.lag_pending = false;
          LS.pair_pending = false;
          LS.set_beta(j - 1, std::sqrt(beta2_j));
          verdict = kRedone;
        } else {
          beta2_j = 0.0;  // w vanished exactly: breakdown (H3)
        }
      }
      if (verdict == kContinue && LS.lag_ok && LS.n_locked > 0 && beta2_j < LS.lag_beta2_min) {
        LS.leave_lagged(j);
        double* cj = E.S(kScalNorms + 3 * slot);
        LS.refs_prev = NormRefs{cj, cj + 1, cj + 1, 0};
        verdict = kRedone;  // the speculative iteration j+1 took the one-sweep form: it is enqueued again
      }
      if (verdict == kContinue && LS.slot_pair[slot] && !(ctx->h_pinned[16 + slot] <= pair_gate)) {
        // A coefficient of this iteration's raw vector grew beyond what the pair form tracks to first order (beta -> eps: an
        // exhausted Krylov space, breakdown).  The iteration itself stands — its coefficients were MEASURED, its alpha / beta
        // are exact — but whatever took the vector as an operator input (the second iteration of its pair, the next pair) is
        // second-order inaccurate: u_j is completed with its measured coefficients, everything after it is enqueued again, and
        // the rest of the pass runs in the one-sweep form (exact for coefficients of any size).
        LS.pair_allowed = false;
        ++LS.n_gate_trips;
        LS.make_final(j);
        LS.pending = false;
        LS.lag_pending = false;
        LS.pair_pending = false;
        double* cj = E.S(kScalNorms + 3 * slot);
        launch_set_scalar(cj + 1, beta2_j, s);  // what the next three-term update reads as beta_j^2
        LS.refs_prev = NormRefs{cj, cj + 1, cj + 1, 0};
        verdict = kRedone;
      }
      alpha.push_back(alpha_j);
      beta.push_back(std::sqrt(beta2_j));
      if (trace_file)
        std::fprintf(trace_file, "iter %lld %lld %.17g %.17g %.17g %.17g %d\n", (long long)passes, (long long)j, alpha_j, beta2_j,
                     c0_j, c1_j, verdict);
      worker.submit((int64_t)alpha.size(), alpha.data(), beta.data());
      return verdict;
    };

    RitzTracker::Out r;
    if (speculate) {
      // One group of iterations (one, or the two of a pair) is enqueued ahead of the group whose scalars are collected.
      int64_t enq = 0, col = 0;  // iterations enqueued / collected so far
      int64_t ahead_first = 1, ahead_last = 0;  // the group enqueued last, not yet collected (empty: first > last)
      while (!stopped && col < P.max_iteration) {
        const int64_t grp_first = ahead_first, grp_last = ahead_last;
        if (enq < P.max_iteration) {
          ahead_first = enq + 1;
          enq += enqueue(enq + 1);
          ahead_last = enq;
        } else {
          LS.flush();  // nothing follows: the last iteration's normalisation / publish step happens now
          ahead_first = 1;
          ahead_last = 0;
        }
        for (int64_t j = grp_first; j <= std::min(grp_last, P.max_iteration) && !stopped; ++j) {
          const bool redo = collect(j) == kRedone;
          col = j;
          if (redo) {  // u_j changed under everything enqueued after it: enqueue again from j + 1
            enq = j;
            ahead_first = 1;
            ahead_last = 0;
          }
          stopped = worker.consume(j, lockstep_lag, kMaxLag, absorb);
          if (redo) break;
        }
      }
    } else {
      for (int64_t k = 1; k <= P.max_iteration && !stopped; ++k) {
        RunList<T> runs = locked_runs;
        runs.add_basis(U, k);
        LS.enqueue(k, P.eigenvalue_offset, runs, mode);
        collect(k);  // kRedone: u_k was repaired in place, nothing ran ahead
        while (!stopped && worker.wait_pop(r)) stopped = absorb(r);
      }
    }
    while (!stopped && worker.wait_pop(r)) stopped = absorb(r);  // the first stop verdict wins; else the last iteration's values
    itern = last.m;  // == max_iteration without a stop (LL:239,312)
    // (a pending pair: the Ritz vectors below need u_0 .. u_{itern-1}; the pending ones enter the GEMV through their raw vectors)
    if (!ctx->tune.ritz_tail) LS.pair_flush(itern);  // (A/B: complete the pending vectors with a sweep of their own)
    const typename LoopState<T>::PairTail tail = LS.take_tail(itern);
    if (trace_file) {
      std::fprintf(trace_file, "stop %lld %lld %d collected %zu\n", (long long)passes, (long long)itern, (int)stopped, alpha.size());
      std::fflush(trace_file);
    }
    evs = last.evs;
    evs_from_qr = last.evs_from_qr;
    alpha.resize((size_t)itern);  // iterations the device ran ahead of the verdict are dropped
    beta.resize((size_t)itern);
    for (size_t i = 0; i < alpha.size(); ++i)  // ||T||_inf of this pass: the operator-size scale of the next pass's gate
      t_inf_prev = std::max(t_inf_prev, std::fabs(alpha[i]) + (i > 0 ? beta[i - 1] : 0.0) + (i + 1 < alpha.size() ? beta[i] : 0.0));
    LL_HIP(hipStreamSynchronize(s));

    // ---- Ritz pairs (LL:312-319, LL:33-62)
    TraceRange trace_ritz("ll::ritz (tridiagonal eigenvectors + GEMV + copy back)");
    const double t_fin0 = now_s();
    const int64_t m = (int64_t)alpha.size();  // == itern
    (void)itern;
    if (P.tridiag_mode == LL_TRIDIAG_AUTO && !evs_from_qr && m > 0) {
      // the loop ended without a convergence stop (max_iteration or breakdown) while bisection was tracking the
      // roots: return the values of the reference's QR arithmetic, like every other exit of this mode
      all.resize((size_t)m);
      const double t0 = now_s();
      tridiag_qr(m, alpha.data(), beta.data(), all.data(), nullptr);
      t_tridiag += now_s() - t0;
      for (size_t i = 0; i < evs.size(); ++i) evs[i] = P.find_maximum ? all[(size_t)m - i - 1] : all[i];
    }
    const int64_t nev = (int64_t)evs.size();
    // Eigenvectors of T_m: the reference accumulates all m of them by QR (LL:44, O(m^3)); LL_TRIDIAG_AUTO switches to
    // inverse iteration for the few wanted ones once m is large.
    const bool few_vectors = P.tridiag_mode == LL_TRIDIAG_AUTO && m > 256;
    std::vector<double> tev, tq;
    if (!few_vectors) {
      tev.resize((size_t)m);
      tq.resize((size_t)m * m);
      const double t0 = now_s();
      tridiag_qr(m, alpha.data(), beta.data(), tev.data(), tq.data());  // beta[m-1] is never read (LL:314)
      t_tridiag += now_s() - t0;
    }
    const std::vector<double> evs_raw = evs;  // Ritz values of the shifted operator, comparator order
    for (auto& e : evs) e -= P.eigenvalue_offset;  // LL:317-319
    // EigenPairManager::insertEigenpairs (EPM:52-71) decides from the VALUES alone which of the nev new pairs
    // survive; replay it on (value, index) first so that only surviving Ritz vectors are formed and copied to the
    // host (the reference forms all nroot = 5 and throws 4 away when one pair is requested, LL:338, EPM:60-64).
    std::vector<char> survives((size_t)nev, 0);
    bool nothing_added = true;
    {
      std::multimap<double, int64_t, std::function<bool(double, double)>> sim(cmp);
      for (auto& kv : kept) sim.emplace(kv.first, (int64_t)-1);
      for (int64_t i = 0; i < nev; ++i) {
        auto ins = sim.emplace(evs[i], i);
        auto last = sim.end();
        --last;
        if ((int64_t)sim.size() > P.num_eigs) {
          if (ins != last) nothing_added = false;
          sim.erase(last);
        } else {
          nothing_added = false;
        }
      }
      for (auto& kv : sim)
        if (kv.second >= 0) survives[(size_t)kv.second] = 1;
    }
    if (spec) std::fill(survives.begin(), survives.end(), (char)1);  // run_iteration returns every computed pair
    std::vector<int64_t> want;
    for (int64_t i = 0; i < nev; ++i)
      if (survives[(size_t)i]) want.push_back(i);
    const int64_t nw = (int64_t)want.size();
    std::vector<std::vector<T>> xs((size_t)nev);
    if (nw > 0) {
      std::vector<T> coeff((size_t)nw * m);
      if (few_vectors) {
        std::vector<double> lam((size_t)nw), sv((size_t)nw * m);
        for (int64_t w = 0; w < nw; ++w) lam[(size_t)w] = evs_raw[(size_t)want[w]];
        const double t0 = now_s();
        tridiag_inverse_iteration(m, alpha.data(), beta.data(), nw, lam.data(), sv.data());
        t_tridiag += now_s() - t0;
        for (size_t i = 0; i < sv.size(); ++i) coeff[i] = as_real_coeff(sv[i], (T*)nullptr);
      } else {
        for (int64_t w = 0; w < nw; ++w) {
          const int64_t it = P.find_maximum ? m - want[w] - 1 : want[w];
          for (int64_t k = 0; k < m; ++k) coeff[(size_t)w * m + k] = as_real_coeff(tq[(size_t)it * m + k], (T*)nullptr);
        }
      }
      RunList<T> basis;
      basis.ld = ld;
      int64_t mm = m;
      if (!tail.active) {
        basis.add_basis(U, m);
      } else {
        // x = sum_{k<m} s_k u_k with the pending vectors expanded (LoopState::PairTail): columns = the stored columns of the last
        // sweep (locked vectors first, then u_0 .. u_{P-1}), then r1 (and r2)
        typedef std::complex<double> Z;
        const int64_t Pt = tail.P, K = L + Pt;
        std::vector<double> g1h((size_t)R * K + 1), g2h((size_t)R * (K + 1) + 1), rho(2, 1.0);
        if (K > 0) E.fetch(tail.g[0], g1h.data(), (size_t)R * K);
        E.fetch(tail.g[1], g2h.data(), (size_t)R * (K + 1));
        E.fetch(tail.rho2[0], &rho[0], 1);
        E.fetch(tail.rho2[1], &rho[1], 1);
        const double rho1 = std::sqrt(rho[0]), rho2 = std::sqrt(rho[1]);
        auto gz = [&](const std::vector<double>& g, int64_t j) { return R == 2 ? Z(g[(size_t)2 * j], g[(size_t)2 * j + 1]) : Z(g[(size_t)j], 0.0); };
        auto to_t = [&](Z v, T* o) {
          if constexpr (scalar_traits<T>::is_complex) {
            o->re = (decltype(o->re))v.real();
            o->im = (decltype(o->im))v.imag();
          } else {
            *o = (T)v.real();
          }
        };
        auto from_t = [&](const T& v) {
          if constexpr (scalar_traits<T>::is_complex) return Z((double)v.re, (double)v.im);
          else return Z((double)v, 0.0);
        };
        mm = K + tail.nvec;
        std::vector<T> c2((size_t)nw * mm);
        std::vector<Z> cs((size_t)K);
        for (int64_t w = 0; w < nw; ++w) {
          const T* sw = coeff.data() + (size_t)w * m;
          for (int64_t j = 0; j < K; ++j) cs[(size_t)j] = j < L ? Z(0.0, 0.0) : from_t(sw[j - L]);
          Z a = from_t(sw[Pt]);                                              // coefficient of u_P
          Z b = tail.nvec == 2 ? from_t(sw[Pt + 1]) : Z(0.0, 0.0);           // ... of u_{P+1}
          Z on_r2(0.0, 0.0);
          if (tail.nvec == 2) {
            on_r2 = b / rho2;
            for (int64_t j = 0; j < K; ++j) cs[(size_t)j] -= on_r2 * gz(g2h, j);
            a -= on_r2 * gz(g2h, K);                                         // gam = <u_P, r2>
          }
          const Z on_r1 = a / rho1;
          for (int64_t j = 0; j < K; ++j) cs[(size_t)j] -= on_r1 * gz(g1h, j);
          T* out = c2.data() + (size_t)w * mm;
          for (int64_t j = 0; j < K; ++j) to_t(cs[(size_t)j], out + j);
          to_t(on_r1, out + K);
          if (tail.nvec == 2) to_t(on_r2, out + K + 1);
        }
        coeff.swap(c2);
        basis = LS.basis_runs(Pt);
        basis.add(tail.src[0], 1);
        if (tail.nvec == 2) basis.add(tail.src[1], 1);
      }
      if (!d_ritz.p || d_ritz_cap < nw) {  // only the surviving vectors are formed; sized by what a pass can return at
        d_ritz_cap = std::max<int64_t>(nw, std::min<int64_t>(nroot_max, spec ? spec->nroot : P.num_eigs));  // most, so that
        d_ritz.alloc(ctx, (size_t)d_ritz_cap * ld);  // repeated runs of one problem reuse ONE cached buffer size
      }
      E.gemv(basis, mm, (int)nw, coeff.data(), d_ritz.p, ld);
      for (int64_t w = 0; w < nw; ++w) {
        E.norm2_dev(d_ritz.p + w * ld, E.S(kScalScratch) + 1);
        const NormRefs nr = E.plain_norm(E.S(kScalScratch) + 1);
        launch_scale<T>(nl, d_ritz.p + w * ld, 0.0, &nr, s);  // LL:58
        if (single_pair && out_dev) {  // the one survivor goes straight to the caller's device buffer
          LL_HIP(hipMemcpyAsync(eigvecs, d_ritz.p + w * ld, (size_t)nl * sizeof(T), hipMemcpyDeviceToDevice, s));
          LL_HIP(hipStreamSynchronize(s));
          result_in_caller = true;
          continue;
        }
        LL_HIP(hipMemcpyAsync(stage, d_ritz.p + w * ld, (size_t)nl * sizeof(T), hipMemcpyDeviceToHost, s));
        LL_HIP(hipStreamSynchronize(s));
        if (single_pair) {
          result_in_stage = true;  // copied to the caller once, at the end
        } else {
          xs[(size_t)want[w]].assign(stage, stage + nl);
        }
      }
    }

    t_finish += now_s() - t_fin0;
    if (passes < iter_cap && iter_counts) iter_counts[passes] = m;
    ++passes;
    total_iters += m;

    if (spec) {  // LL:312-321: hand the pairs back as they are
      for (int64_t i = 0; i < nev; ++i) {
        eigvals[i] = evs[(size_t)i];
        if (eigvecs) to_caller(eigvecs + (size_t)i * nl, xs[(size_t)i].data());
      }
      *n_found = nev;
      break;
    }
    // ---- EigenPairManager::insertEigenpairs (EPM:52-71), now with the vectors of the survivors
    {
      bool check_nothing = true;
      for (int64_t i = 0; i < nev; ++i) {
        auto ins = kept.emplace(evs[i], std::move(xs[(size_t)i]));
        auto last = kept.end();
        --last;
        if ((int64_t)kept.size() > P.num_eigs) {
          if (ins != last) check_nothing = false;
          kept.erase(last);
        } else {
          check_nothing = false;
        }
      }
      (void)check_nothing;
    }
    if (nothing_added) break;    // LL:346-348
    if (P.num_eigs == 1) break;  // LL:350-353
  }

  int64_t cnt = spec ? *n_found : 0;
  for (auto kv = kept.begin(); !spec && kv != kept.end(); ++kv) {  // comparator order (LL:356-365)
    eigvals[cnt] = kv->first;
    if (eigvecs && !(single_pair && result_in_caller)) {
      const T* src = (single_pair && result_in_stage) ? stage : kv->second.data();
      to_caller(eigvecs + (size_t)cnt * nl, src);
    }
    ++cnt;
  }
  *n_found = cnt;
  if (alpha_out) std::copy(alpha.begin(), alpha.end(), alpha_out);
  if (beta_out) std::copy(beta.begin(), beta.end(), beta_out);
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->n_passes = passes;
    stats->total_iterations = total_iters;
    stats->seconds_host_tridiag = t_tridiag;
    stats->last_alpha_len = (int64_t)alpha.size();
    stats->seconds_host_enqueue = LS.t_enqueue;
    stats->seconds_host_wait = t_wait;
    stats->seconds_setup = t_setup;
    stats->seconds_finish = t_finish;
    stats->second_passes = second_passes;
    stats->lagged_iterations = LS.n_lagged;
    stats->pair_iterations = LS.n_pair;
    stats->pair_gate_trips = LS.n_gate_trips;
    timer.collect(stats->seconds_spmv, stats->seconds_orth);
    ctx->drain_comm_events(&stats->seconds_comm_gather, &stats->seconds_comm_allreduce);
    stats->seconds_total = now_s() - t_start;
  }
  ctx->drain_comm_events(nullptr, nullptr);
  (void)R;
}

template void lanczos_run<double>(ll_context*, ll_operator*, const ll_lanczos_params&, double*, double*, int64_t*,
                                  int64_t*, int64_t, double*, double*, ll_run_stats*, const IterationSpec<double>*);
template void lanczos_run<zc>(ll_context*, ll_operator*, const ll_lanczos_params&, double*, zc*, int64_t*, int64_t*,
                              int64_t, double*, double*, ll_run_stats*, const IterationSpec<zc>*);
template void lanczos_run<float>(ll_context*, ll_operator*, const ll_lanczos_params&, double*, float*, int64_t*,
                                 int64_t*, int64_t, double*, double*, ll_run_stats*, const IterationSpec<float>*);
template void lanczos_run<cf>(ll_context*, ll_operator*, const ll_lanczos_params&, double*, cf*, int64_t*, int64_t*,
                              int64_t, double*, double*, ll_run_stats*, const IterationSpec<cf>*);

// ================================================================= Exponentiator<T>::run
namespace {
inline void from_std(double v, double* o) { *o = v; }
inline void from_std(std::complex<double> v, zc* o) { o->re = v.real(); o->im = v.imag(); }
inline void from_std(double v, float* o) { *o = (float)v; }
inline void from_std(std::complex<double> v, cf* o) { o->re = (float)v.real(); o->im = (float)v.imag(); }
}  // namespace

template <typename T>
void expo_run(ll_context* ctx, ll_operator* op, const ll_expo_params& P_in, typename host_scalar<T>::type a,
              const T* input, T* output, int64_t* itern_out, ll_run_stats* stats) {
  typedef typename host_scalar<T>::type H;
  ll_expo_params P = P_in;
  if (sizeof(typename scalar_traits<T>::real) == 4 && P.eps == std::numeric_limits<double>::epsilon() * 1e2)
    P.eps = (double)std::numeric_limits<float>::epsilon() * 1e2;  // EX:58 with real_t<T> = float (see lanczos_run)
  LL_REQUIRE(op && op->ctx == ctx, "operator belongs to another context");
  LL_REQUIRE(op->is_complex == scalar_traits<T>::is_complex && op->elem_bytes == (int)sizeof(T),
             "operator scalar type mismatch");
  LL_REQUIRE(P.matrix_size == op->n, "matrix_size differs from the operator dimension");
  LL_REQUIRE(P.max_iteration >= 1, "max_iteration must be >= 1");
  LL_HIP(hipSetDevice(ctx->device));
  const double t_start = now_s();
  hipStream_t s = ctx->stream;
  StallTrace st("expo_run", ctx->tune.stall_trace_ms);
  const int64_t nl = op->n_local;
  const int64_t ld = round_up(std::max(nl, op->n_shard), 256);
  Engine<T> E(ctx, op, nl);
  st.at("engine");
  const double dgks_thr = ctx->tune.dgks_threshold;
  // Two launches per iteration less on single-GPU runs: alpha is folded by the multi-dot that needs it, and the fold of
  // the post-pass norm + the publish step ride in the normalisation kernel.  LL_FUSE_LAUNCHES=0: separate kernels (A/B).
  const bool fuse_launches = ctx->tune.fuse_launches;
  Basis<T> U;
  U.init(ctx, nl, ld, pick_chunk_vecs(P.initial_vector_size, P.max_iteration, ld * (int64_t)sizeof(T), ctx->tune.slab_bytes));
  st.at("basis");
  ctx->ensure_pinned(32);  // 4 ring slots of 4 scalars, then the 4 gate values of the pair form
  EventRing ring;
  PhaseTimer timer(ctx, s);
  double t_tridiag = 0.0;
  st.at("events");

  // u[0] = input / ||input||  (EX:100-101); ||input|| is kept for the output scaling (EX:165)
  LL_HIP(hipMemcpyAsync(U.vec(0), input, (size_t)nl * sizeof(T), hipMemcpyDefault, s));
  st.at("input-copy-enqueued");
  E.norm2_dev(U.vec(0), E.S(kScalScratch) + 1);
  double in_norm2 = 0.0;
  E.fetch(E.S(kScalScratch) + 1, &in_norm2, 1);
  st.at("input-norm-fetched");
  const double in_norm = std::sqrt(in_norm2);
  NormRefs refs_prev = E.plain_norm(E.S(kScalScratch) + 1);
  launch_scale<T>(nl, U.vec(0), 0.0, &refs_prev, s);

  std::vector<double> alpha, beta;
  std::vector<H> coeff_prev;
  int64_t second_passes = 0;
  int64_t itern = P.max_iteration;
  bool stopped = false;

  double t_wait = 0.0;
  LoopState<T> LS(E, U, ring, timer, nl, s);
  LS.fuse_launches = fuse_launches;
  LS.refs_prev = refs_prev;
  LS.max_k_hint = P.max_iteration;
  if (E.can_defer_scale() && fuse_launches && (!P.full_orthogonalize || P.orth_mode == LL_ORTH_CGS_DGKS)) LS.enable_defer(ld);
  if (E.can_scale_input() && fuse_launches && ctx->tune.lagged_gs && P.full_orthogonalize && P.orth_mode == LL_ORTH_CGS_DGKS)
    LS.enable_lagged(ld);
  // two iterations per sweep with full_orthogonalize (EX:120-122), exactly as in the eigen-solver loop (LoopState::enqueue_pair)
  {
    const int64_t stream_bytes = std::min<int64_t>(ctx->tune.blas_small_bytes, (int64_t)1 << 20);
    const int64_t pair_min = ctx->tune.lagged_min_bytes >= 0 ? std::min<int64_t>(ctx->tune.lagged_min_bytes, stream_bytes)
                                                             : std::min<int64_t>(stream_bytes, LoopState<T>::kPairSmallMinBytes);
    if (LS.lagged && ctx->tune.pair_gs && (ctx->comm != nullptr ? op->n_shard : nl) * (int64_t)sizeof(T) >= pair_min) LS.enable_pair();
  }
  LS.begin_pass(nullptr, 0);
  auto enqueue = [&](int64_t k) -> int64_t {  // EX:107-118 (+ EX:120-122 with full_orthogonalize), EX:145, EX:160
    if (P.full_orthogonalize && P.orth_mode == LL_ORTH_CGS_DGKS && !LS.pending && LS.enqueue_pair(k, 0.0)) return 2;
    RunList<T> runs;
    runs.ld = ld;
    if (P.full_orthogonalize) runs.add_basis(U, k);
    LS.enqueue(k, 0.0, runs, P.orth_mode);
    return 1;
  };
  // Host half of iteration j, part 1 (this thread): the four scalars, the DGKS decision, alpha_j / beta_j; part 2 (EX:124-158:
  // exp(a T_j) e_1 and the overlap test, O(j^3)) runs on the helper thread like the eigen-solver's Ritz step.
  enum { kContinue = 0, kRedone = 2 };
  ExpoTracker<H> tracker_cfg;
  tracker_cfg.a = a;
  tracker_cfg.eps = P.eps;
  tracker_cfg.breakdown_tol = (double)std::numeric_limits<typename scalar_traits<T>::real>::epsilon();  // EX:154
  const bool speculate = !(op->kind == ll_operator::HOST_CB || op->kind == ll_operator::DEV_CB);  // see lanczos_run
  const bool threaded = speculate && ctx->tune.tridiag_thread;
  const size_t kMaxLag = threaded ? 24 : 0;
  const int64_t lockstep_lag = threaded && ctx->comm != nullptr ? std::max(-1, ctx->tune.tridiag_lag) : -1;  // see StepWorker::consume
  StepWorker<ExpoTracker<H>> worker(tracker_cfg, threaded, ctx->tune.tridiag_test_jitter_us);
  typename ExpoTracker<H>::Out last, r;
  auto absorb = [&](typename ExpoTracker<H>::Out& o) {
    t_tridiag += o.seconds;
    last = std::move(o);
    return last.stop;
  };
  const double pair_gate = sizeof(typename scalar_traits<T>::real) == 4 ? 2e-4 : kPairGate;
  auto collect = [&](int64_t j) -> int {
    const int slot = (int)(j % 4);
    const double tw0 = now_s();
    LL_HIP(hipEventSynchronize(ring.ev[LS.ev_of_slot[slot]]));
    t_wait += now_s() - tw0;
    const volatile double* hp = ctx->h_pinned + 4 * slot;
    const double alpha_j = hp[0], c0_j = hp[2], c1_j = hp[3];
    double beta2_j = hp[1];
    int verdict = kContinue;
    if (P.full_orthogonalize && P.orth_mode == LL_ORTH_CGS_DGKS && c1_j < dgks_thr * c0_j) {  // see lanczos_run
      if (c1_j > 0.0 && std::isfinite(c1_j)) {
        RunList<T> again;
        again.ld = ld;
        again.add_basis(U, j);
        LS.make_final(j);
        beta2_j = c1_j * E.second_pass(U.vec(j), again);
        ++second_passes;
        double* cj = E.S(kScalNorms + 3 * slot);
        launch_set_scalar(cj + 1, beta2_j, s);
        LS.refs_prev = NormRefs{cj, cj + 1, cj + 1, 0};
        LS.pending = false;  // (see lanczos_run)
        LS.lag_pending = false;
        LS.pair_pending = false;
        LS.set_beta(j - 1, std::sqrt(beta2_j));
        verdict = kRedone;
      } else {
        beta2_j = 0.0;
      }
    }
    if (verdict == kContinue && LS.slot_pair[slot] && !(ctx->h_pinned[16 + slot] <= pair_gate)) {  // the pair form's gate (see lanczos_run)
      LS.pair_allowed = false;
      ++LS.n_gate_trips;
      LS.make_final(j);
      LS.pending = false;
      LS.lag_pending = false;
      LS.pair_pending = false;
      double* cj = E.S(kScalNorms + 3 * slot);
      launch_set_scalar(cj + 1, beta2_j, s);
      LS.refs_prev = NormRefs{cj, cj + 1, cj + 1, 0};
      verdict = kRedone;
    }
    alpha.push_back(alpha_j);
    beta.push_back(std::sqrt(beta2_j));  // EX:145
    worker.submit((int64_t)alpha.size(), alpha.data(), beta.data());
    return verdict;
  };
  if (speculate) {
    // One group of iterations (one, or the two of a pair) is enqueued ahead of the group whose scalars are collected (lanczos_run).
    int64_t enq = 0, col = 0;
    int64_t ahead_first = 1, ahead_last = 0;
    while (!stopped && col < P.max_iteration) {
      const int64_t grp_first = ahead_first, grp_last = ahead_last;
      if (enq < P.max_iteration) {
        ahead_first = enq + 1;
        enq += enqueue(enq + 1);
        ahead_last = enq;
      } else {
        LS.flush();
        ahead_first = 1;
        ahead_last = 0;
      }
      for (int64_t j = grp_first; j <= std::min(grp_last, P.max_iteration) && !stopped; ++j) {
        const bool redo = collect(j) == kRedone;
        col = j;
        if (redo) {
          enq = j;
          ahead_first = 1;
          ahead_last = 0;
        }
        stopped = worker.consume(j, lockstep_lag, kMaxLag, absorb);
        if (redo) break;
      }
    }
  } else {
    for (int64_t k = 1; k <= P.max_iteration && !stopped; ++k) {
      enqueue(k);
      collect(k);
      while (!stopped && worker.wait_pop(r)) stopped = absorb(r);
    }
  }
  while (!stopped && worker.wait_pop(r)) stopped = absorb(r);
  itern = last.m;
  coeff_prev = last.coeff;
  LS.pair_flush((int64_t)coeff_prev.size());  // (a pending pair: the output below needs u_0 .. u_{m-1} complete in the basis)
  alpha.resize((size_t)itern);
  beta.resize((size_t)itern);
  st.at("loop");
  LL_HIP(hipStreamSynchronize(s));
  st.at("drained");

  // output = ||input|| * sum_l coeff_prev[l] u[l]  (EX:163-170)
  const int64_t m = (int64_t)coeff_prev.size();
  std::vector<T> c((size_t)m);
  for (int64_t l = 0; l < m; ++l) from_std(H(in_norm) * coeff_prev[l], &c[l]);
  DevBuf<T> d_out;
  d_out.alloc(ctx, (size_t)ld);
  RunList<T> basis;
  basis.ld = ld;
  basis.add_basis(U, m);
  st.at("out-alloc");
  E.gemv(basis, m, 1, c.data(), d_out.p, ld);
  LL_HIP(hipMemcpyAsync(output, d_out.p, (size_t)nl * sizeof(T), hipMemcpyDefault, s));
  st.at("gemv+copy-enqueued");
  LL_HIP(hipStreamSynchronize(s));
  st.at("output-done");
  *itern_out = itern;
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->n_passes = 1;
    stats->total_iterations = itern;
    stats->seconds_host_tridiag = t_tridiag;
    stats->last_alpha_len = (int64_t)alpha.size();
    stats->second_passes = second_passes;
    stats->lagged_iterations = LS.n_lagged;
    stats->pair_iterations = LS.n_pair;
    stats->pair_gate_trips = LS.n_gate_trips;
    stats->seconds_host_enqueue = LS.t_enqueue;
    stats->seconds_host_wait = t_wait;
    timer.collect(stats->seconds_spmv, stats->seconds_orth);
    ctx->drain_comm_events(&stats->seconds_comm_gather, &stats->seconds_comm_allreduce);
    stats->seconds_total = now_s() - t_start;
  }
  ctx->drain_comm_events(nullptr, nullptr);
}
template void expo_run<double>(ll_context*, ll_operator*, const ll_expo_params&, double, const double*, double*,
                               int64_t*, ll_run_stats*);
template void expo_run<zc>(ll_context*, ll_operator*, const ll_expo_params&, std::complex<double>, const zc*, zc*,
                           int64_t*, ll_run_stats*);
template void expo_run<float>(ll_context*, ll_operator*, const ll_expo_params&, double, const float*, float*, int64_t*,
                              ll_run_stats*);
template void expo_run<cf>(ll_context*, ll_operator*, const ll_expo_params&, std::complex<double>, const cf*, cf*,
                           int64_t*, ll_run_stats*);

// ================================================================= Exponentiator<T>::taylor_run (EX:175-210)
template <typename T>
void taylor_run(ll_context* ctx, ll_operator* op, const ll_expo_params& P, typename host_scalar<T>::type a,
                const T* input, T* output, int64_t* nterms_out) {
  typedef typename host_scalar<T>::type H;
  LL_REQUIRE(op && op->ctx == ctx, "operator belongs to another context");
  LL_REQUIRE(op->is_complex == scalar_traits<T>::is_complex && op->elem_bytes == (int)sizeof(T),
             "operator scalar type mismatch");
  LL_REQUIRE(P.matrix_size == op->n, "matrix_size differs from the operator dimension");
  LL_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int64_t nl = op->n_local;
  if (a == H(0)) {  // EX:179-182; input / output may be host or device memory, and may be the same buffer
    if (output != input) {
      LL_HIP(hipMemcpyAsync(output, input, (size_t)nl * sizeof(T), hipMemcpyDefault, s));
      LL_HIP(hipStreamSynchronize(s));
    }
    *nterms_out = 1;
    return;
  }
  const int64_t ld = round_up(std::max(nl, op->n_shard), 256);
  Engine<T> E(ctx, op, nl);
  Basis<T> V;
  V.init(ctx, nl, ld, 32);
  LL_HIP(hipMemcpyAsync(V.vec(0), input, (size_t)nl * sizeof(T), hipMemcpyDefault, s));
  H factor = 1.0;
  int64_t terms = 1;
  for (int64_t k = 1;; ++k) {  // EX:187-195
    factor *= a / H((double)k);
    E.apply(V.vec(k - 1), V.vec(k), 0.0, nullptr, true);
    ++terms;
    E.norm2_dev(V.vec(k), E.S(kScalScratch) + 1);
    double nn = 0.0;
    E.fetch(E.S(kScalScratch) + 1, &nn, 1);
    if (std::sqrt(nn) * std::abs(factor) < P.eps) break;
  }
  std::vector<T> c((size_t)terms);
  for (int64_t k = terms; k-- > 0;) {  // backward sum with the reference's factor recurrence (EX:198-206)
    from_std(factor, &c[k]);
    factor *= H((double)k) / a;
  }
  DevBuf<T> d_out;
  d_out.alloc(ctx, (size_t)ld);
  RunList<T> basis;
  basis.ld = ld;
  basis.add_basis(V, terms);
  E.gemv(basis, terms, 1, c.data(), d_out.p, ld);
  LL_HIP(hipMemcpyAsync(output, d_out.p, (size_t)nl * sizeof(T), hipMemcpyDefault, s));
  LL_HIP(hipStreamSynchronize(s));
  *nterms_out = terms;
}
template void taylor_run<double>(ll_context*, ll_operator*, const ll_expo_params&, double, const double*, double*,
                                 int64_t*);
template void taylor_run<zc>(ll_context*, ll_operator*, const ll_expo_params&, std::complex<double>, const zc*, zc*,
                             int64_t*);
template void taylor_run<float>(ll_context*, ll_operator*, const ll_expo_params&, double, const float*, float*,
                                int64_t*);
template void taylor_run<cf>(ll_context*, ll_operator*, const ll_expo_params&, std::complex<double>, const cf*, cf*,
                             int64_t*);

template struct Basis<double>;
template struct Basis<zc>;
template struct Basis<float>;
template struct Basis<cf>;
template struct RunList<double>;
template struct RunList<zc>;
template struct RunList<float>;
template struct RunList<cf>;
template struct Engine<double>;
template struct Engine<zc>;
template struct Engine<float>;
template struct Engine<cf>;

}  // namespace ll
