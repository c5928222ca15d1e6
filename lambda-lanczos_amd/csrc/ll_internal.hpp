// Internal declarations shared by the HIP kernels, the host drivers and the C ABI.
// Not installed; the public surface is include/lanczos_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#if defined(__HIPCC__)
#include <hip/hip_ext.h>  // hipExtLaunchKernelGGL (device-code translation units only: the host-only sanitizer builds use g++)
#endif

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/lanczos_hip.h"
#include "../../include/lanczos_hip_transport.h"

#if defined(__HIPCC__)
// A launch whose completion IS an event: hipExtLaunchKernelGGL hangs the event on the kernel's own dispatch packet; hipEventRecord
// behind the launch is a marker packet of its own between two dependent kernels (config 5: 5.9 -> 4.5 us in front of the next kernel,
// 16.5 k -> 16.9 k it/s; n = 1e5: + 1.4 %; same-box A/B through the key event_in_launch).
#define LL_LAUNCH_STOP(stop, kernel, grid, block, lds, s, ...)                                      \
  do {                                                                                              \
    if (stop) hipExtLaunchKernelGGL(kernel, grid, block, lds, s, nullptr, stop, 0, __VA_ARGS__);    \
    else hipLaunchKernelGGL(kernel, grid, block, lds, s, __VA_ARGS__);                              \
  } while (0)
#endif

namespace ll {

// ---------------------------------------------------------------- scalar types
// Device-side complex<double>: interleaved (re, im), 16-byte aligned so that one element is one dwordx4 access.
struct alignas(16) zc {
  double re, im;
};

// Device-side complex<float>: interleaved (re, im), 8-byte aligned.
struct alignas(8) cf {
  float re, im;
};

// `reals` = doubles per REDUCED value (all reductions, coefficients and norms are carried in double / zc whatever the
// storage type: float inputs are widened at the first accumulation, which is at least the reference's accuracy);
// `acc` = that accumulator type.
template <typename T> struct scalar_traits;
template <> struct scalar_traits<double> {
  static constexpr bool is_complex = false;
  static constexpr int reals = 1;
  typedef double acc;
  typedef double real;
};
template <> struct scalar_traits<zc> {
  static constexpr bool is_complex = true;
  static constexpr int reals = 2;
  typedef zc acc;
  typedef double real;
};
template <> struct scalar_traits<float> {
  static constexpr bool is_complex = false;
  static constexpr int reals = 1;
  typedef double acc;
  typedef float real;
};
template <> struct scalar_traits<cf> {
  static constexpr bool is_complex = true;
  static constexpr int reals = 2;
  typedef zc acc;
  typedef float real;
};
template <typename T> using acc_t = typename scalar_traits<T>::acc;

// ---------------------------------------------------------------- errors
void set_error(const std::string& msg);
struct Failure {
  int code;
};

#define LL_HIP(expr)                                                                                         \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess) {                                                                                  \
      ::ll::set_error(std::string(#expr) + " failed: " + hipGetErrorString(e_) + " (" + __FILE__ + ":" +     \
                      std::to_string(__LINE__) + ")");                                                       \
      throw ::ll::Failure{LL_ERR_HIP};                                                                       \
    }                                                                                                        \
  } while (0)

#define LL_REQUIRE(cond, msg)                                                      \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      ::ll::set_error(std::string("invalid argument: ") + (msg));                  \
      throw ::ll::Failure{LL_ERR_INVALID};                                         \
    }                                                                              \
  } while (0)

// ---------------------------------------------------------------- launch geometry
constexpr int kBlock = 256;          // threads per workgroup (4 waves of 64)
constexpr int kCUs = 256;            // MI355X
constexpr int kXcds = 8;
constexpr int kMaxGrid = kCUs * 8;   // persistent grids: at most 8 workgroups per CU (multiple of 8 XCDs)
constexpr int kMaxSpmvGrid = kCUs * 16;  // CSR-stream SpMV with 16-byte values (its partial dot products: d_alpha_partials)
constexpr int kSpmvTileNnz = 1024;   // nonzeros staged through LDS per SpMV tile
constexpr int kMaxSegs = 26;         // basis segments (slabs) per multi-dot / multi-axpy launch: 5000 vectors in slabs of 200

// A run of basis vectors stored with a common leading dimension: vector j at base + j*ld.
template <typename T> struct BasisSegs {
  const T* base[kMaxSegs];
  int count[kMaxSegs];
  int nseg;
  long long ld;
};

// The three squared norms of one Gram-Schmidt call, as device scalars:
//   c0 = ||w||^2 before pass 1, c1 = after pass 1, c2 = after pass 2 (valid only if the second pass ran).
// Device-predicated form (the one-call primitive ll_orth_block_*): the second pass runs iff force2 (LL_ORTH_CGS2) or
// c1 < c0/2 (DGKS "twice is enough" test) and every consumer (scale, next three-term update, host read-back) applies
// the same selection.  The whole-loop drivers enqueue pass 1 only (c2 aliases c1, force2 = 0) and take the DGKS
// decision on the host from the published (c0, c1) one iteration later (engine.hpp, Engine::second_pass).
struct NormRefs {
  const double* c0;
  const double* c1;
  const double* c2;
  int force2;
  double thr = 0.5;  // DGKS threshold of the device-predicated form (LL_DGKS_THRESHOLD overrides it, like on the host)
};

// ---------------------------------------------------------------- communicator (RCCL, lazily loaded; or an external transport)
struct Comm;
Comm* comm_attach(const struct ::ll_transport* t, int rank, int nranks);
Comm* comm_create(const void* id128, int rank, int nranks, int device);
void comm_destroy(Comm*);
std::string comm_transport_name(const Comm*);  // "none" | "rccl" | "plugin:<path>" | "attached"
void comm_unique_id(void* id128);
void comm_allgather(Comm*, const void* send, void* recv, size_t bytes, hipStream_t s);
void comm_allreduce_sum(Comm*, double* buf, size_t n_doubles, hipStream_t s);
// Ring halo exchange: `bytes` from send_prev go to rank `prev` (they become its recv_next) and `bytes` from send_next
// go to rank `next` (its recv_prev); the matching messages arrive in recv_prev / recv_next.  prev / next = -1: no such
// neighbour (open boundary).  prev == next (two ranks on a ring) and prev == next == own rank are legal.
void comm_halo_exchange(Comm*, const void* send_prev, void* recv_prev, int prev, const void* send_next, void* recv_next,
                        int next, size_t bytes, hipStream_t s);

}  // namespace ll

// ---------------------------------------------------------------- exchange plan of a sharded vector
// The all-gather of a vector whose shards have the stride n_shard is cut into nchunks pieces: piece c = elements
// [start[c], start[c] + len[c]) of EVERY shard.  Region c of the gathered buffer starts at nranks * start[c] elements
// and holds rank s's piece at + s * len[c].  One chunk => the buffer is the vector in global order.
namespace ll {
constexpr int kMaxGatherChunks = 8;
struct GatherPlan {
  int nchunks = 1;
  int64_t start[kMaxGatherChunks] = {0};
  int64_t len[kMaxGatherChunks] = {0};
};
}  // namespace ll

// ---------------------------------------------------------------- tuning: environment switches and per-context settings
// The USER-FACING LL_* switches are read from the environment ONCE, when a context is created
// (ll_ctx_create*), into the context; operators copy what shapes their image when THEY are created.  Nothing on a launch path
// calls getenv.  ll_ctx_reload_env() reads them again.  Every other field below — geometry overrides, forced code paths, the
// hooks of the test suite — is NOT read from the environment: it is set per context through ll_ctx_set_tuning(ctx, key, value)
// (capi.cpp tuning_apply holds the one parser; the comments below name the key).
namespace ll {
struct Tuning {
  // --- operator creation
  int spmv_kernel = 0;             // LL_SPMV_KERNEL = auto (0, time both and keep the faster) | csr (1) | pb (2)
  bool keep_both = false;          // LL_SPMV_KEEP_BOTH=1: keep the image that lost the timing (ll_op_select_spmv A/B)
  int pb_phase2 = 4;               // LL_PB_PHASE2 = fixed (4, default) | ordered (1) | atomic (0); see spmv_pb.hip
  int pb_block = 0;                // LL_PB_BLOCK: rows AND columns per block (0: automatic); tests force ragged blocks
  int pb_row_block = 0;            // LL_PB_ROW_BLOCK / LL_PB_COL_BLOCK: one of the two only
  int pb_col_block = 0;
  int pair_max_stored = 0;         // key pair_max_stored = n: the pair form hands over to the one-sweep form beyond n stored vectors (test hook; by itself at 4 992 real / 2 492 complex)
  int pair_split_vecs = 0;         // key pair_split = n: at most n stored vectors per launch of the pair sweep (test hook: split sweeps on small problems)
  int pb_threads1 = 0;             // LL_PB_THREADS1 = 256 | 512 | 1024: lanes per workgroup of PB phase 1 (0: automatic — 512 for the thin column blocks of a sharded image, 1024 on one GPU); read at creation
  int pb_pad = 0;                  // LL_PB_PAD = 4 | 16: entries every segment of the PB image is padded to (0: automatic — 4 sharded, 16 on one GPU); read at creation
  int pb_placements = 8;           // LL_PB_PLACEMENTS: arena placements timed at creation (1: keep the first; LL_PB_PLACEMENT_TRACE=1 prints every draw); capi.cpp
  bool pb_xpre = true;             // LL_PB_XPRE=0: phase 2 of the PB SpMV loads x_i in its epilogue (A/B of the early request)
  bool pb_diag = true;             // LL_PB_DIAG=0: the diagonal entries travel through the PB streams like every other entry (A/B)
  int gather_chunks = 0;           // LL_GATHER_CHUNKS: pieces of the all-gather (0: 4 on two ranks, 2 on more)
  bool spmv_tile_balance = true;   // LL_SPMV_TILE_BALANCE=0: CSR-stream tiles always hold up to 1024 nonzeros (capi.cpp build_tiles)
  bool csr_split = true;           // LL_CSR_SPLIT=0: sharded CSR-stream / dense operators gather first, then multiply (round-3 form)
  bool comm_overlap = true;        // LL_COMM_OVERLAP=0: exchange and compute on one stream (serial A/B reference)
  // --- the loops
  bool tridiag_thread = true;      // LL_TRIDIAG_THREAD=0: host Ritz step inline instead of on the helper thread
  int tridiag_lag = 3;             // LL_TRIDIAG_LAG: fixed verdict lag of sharded runs (engine.cpp)
  double dgks_threshold = 0.5;     // LL_DGKS_THRESHOLD: second Gram-Schmidt pass when ||w'||^2 < thr * ||w||^2
  bool sharded_norm_measured = false;  // LL_SHARDED_NORM=measured: all-reduce the post-pass norm instead of deriving it
  int64_t slab_bytes = (int64_t)4 << 30;       // LL_SLAB_BYTES: cap of one Krylov-basis slab
  int64_t blas_small_bytes = (int64_t)4 << 20;  // LL_BLAS_SMALL_BYTES: vectors below this use the small-vector kernels
  bool fuse_launches = true;       // LL_FUSE_LAUNCHES=0: separate fold / publish kernels (A/B of the launch fusion)
  long long lagged_min_bytes = -1; // key lagged_min_bytes: shortest vector of the one-sweep form (-1 = default)
  int lagged_pieces = 0;           // key lagged_pieces: strip geometry of the one-sweep kernel (0 = by length)
  bool lagged_gs = true;           // LL_FUSE_LAUNCHES=1: fused folds but the two-sweep Gram-Schmidt form; 2 (default): one sweep
  bool event_in_launch = true;     // key event_in_launch = 0: iteration events as marker packets behind the publishing kernel (A/B)
  bool ritz_tail = true;           // key ritz_tail = 0: a pair pending at the end of a pass is completed by sweeps of its own instead of entering the Ritz GEMV through its raw vectors (A/B)
  int sweep_pipeline = 1;          // key sweep_pipeline: 1 (default) the software-pipelined pair sweep on streaming vectors (> ~9 MiB), 0 never (A/B: same bits), 2 on every length (parity tests on small cases)
  bool pair_gs = true;             // LL_PAIR_GS=0: never two iterations per sweep (the one-sweep form throughout; A/B and parity hunts)
  // --- test hooks (not for users)
  bool force_rp64 = false;         // LL_FORCE_RP64=1: 64-bit row offsets on small matrices
  bool tl_xcd_order = true;        // LL_TL_XCD=0: row blocks of the tiled kernel in launch order instead of one contiguous eighth per XCD (A/B)
  bool tl_walk_modulo = true;      // LL_TL_WALK=0: a row block's tiles in ascending column order instead of by column index modulo the longest tile list (A/B; read at creation)
  bool tl_force = false;           // LL_TL_FORCE=1: build the tiled image even for matrices that are not eligible (parity tests on small cases)
  bool pb_test_all_remote = false; // LL_PB_TEST_ALL_REMOTE=1: own columns are read from the gathered buffer too
  int tridiag_test_jitter_us = 0;  // LL_TRIDIAG_TEST_JITTER_US: random delay of every helper-thread verdict
  bool stencil_vec = true;         // LL_STENCIL_VEC=0: scalar lattice kernel on shapes the vector kernel would take
  double stall_trace_ms = -1.0;    // LL_STALL_TRACE: print where a whole-loop call longer than this spent its time
  std::string iter_trace;          // LL_ITER_TRACE=path: the eigen-solver loop appends one line per collected iteration
                                   // (pass k alpha beta^2 c0 c1 second-pass) and one per stop verdict — for parity hunts
  bool pb_placement_trace = false; // LL_PB_PLACEMENT_TRACE=1: print every placement draw of the PB image (capi.cpp)
};
// capi.cpp: defaults <- the user-facing environment switches <- the context's overrides (ll_ctx_set_tuning), in that order
Tuning read_tuning(const std::map<std::string, std::string>* overrides);
bool tuning_apply(Tuning& t, const std::string& key, const std::string& value);  // false: unknown key

}  // namespace ll

// ---------------------------------------------------------------- context
struct ll_context {
  ll::Tuning tune;
  std::map<std::string, std::string> tuning_overrides;  // ll_ctx_set_tuning: key -> value, applied on top of the environment
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  ll::Comm* comm = nullptr;
  int rank = 0, nranks = 1;
  int ranks_seen = 0;                 // result of the rank self-check at ll_comm_init (== nranks when healthy)
  // Exchange overlap: the all-gather of a sharded vector is issued on comm_stream, chunk by chunk, so that
  // SpMV work on the rank's own columns runs under it and every chunk's remote-column work starts when that chunk
  // has arrived.  LL_COMM_OVERLAP=0 issues everything on `stream` instead (serial reference path for A/B tests).
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_x_ready = nullptr;
  hipEvent_t ev_chunk[ll::kMaxGatherChunks] = {};
  bool profiling = false;
  hipEvent_t t0 = nullptr, t1 = nullptr;  // ll_timer_*

  // workspace, all sized lazily
  double* d_partials = nullptr;  // [grid][ncols] block partial sums
  double* d_alpha_partials = nullptr;  // the operator kernels' partial <x, Ax> (kept apart: the multi-dot that follows
  size_t alpha_partials_cap = 0;       // may fold them itself while it writes its own partials)
  size_t partials_cap = 0;       // doubles
  double* d_h = nullptr;         // reduced projection coefficients / small scalars
  size_t h_cap = 0;              // doubles
  double* d_scal = nullptr;      // 64 doubles of device scalars (ring slots, flags)
  double* d_norm_partials = nullptr;  // kMaxGrid norm partials of the folding multi-axpy (must not alias d_partials)
  double* h_pinned = nullptr;    // pinned host mirror for scalar read-back
  size_t pinned_cap = 0;         // doubles
  void* d_coeff = nullptr;       // coefficient upload area for gemv_basis
  size_t coeff_cap = 0;          // bytes
  std::vector<hipEvent_t> timer_events;  // PhaseTimer's ring of timing events (profiling mode), created once and kept between runs
  std::vector<std::pair<void*, size_t>> slab_cache;  // Krylov-basis slabs kept between runs (ptr, bytes), oldest first
  // Return a buffer to the cache.  The cache is bounded (kSlabCacheMaxEntries): a long-lived context that solves problems
  // of many different shapes frees its oldest cached buffers instead of accumulating them (hipFree synchronises the device;
  // it happens only when the bound is hit, never inside a loop).
  static constexpr size_t kSlabCacheMaxEntries = 64;
  void cache_put(void* p, size_t bytes);
  void* d_xfull = nullptr;       // all-gather target (sharded runs)
  size_t xfull_cap = 0;          // bytes
  void* d_halo = nullptr;        // received halos of the lattice operator: [from prev | from next]
  size_t halo_cap = 0;           // bytes

  // hipMalloc that makes room first when the device is full: the cached Krylov slabs of earlier runs are returned to
  // the device and the allocation is retried; LL_ERR_ALLOC (with the size in the message) if it still fails.
  void dev_malloc(void** out, size_t bytes, const char* what);
  void ensure_partials(size_t doubles);
  void ensure_alpha_partials(size_t doubles);
  void ensure_h(size_t doubles);
  void ensure_pinned(size_t doubles);
  void ensure_coeff(size_t bytes);
  void ensure_xfull(size_t bytes);
  void ensure_halo(size_t bytes);
  // device-time stamps of the exchange steps (only with profiling on): (start, end) pairs on the stream they ran on
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_gather, ev_allreduce;
  void drain_comm_events(double* gather_s, double* allreduce_s);  // adds the elapsed device seconds, frees the events
  void* h_stage = nullptr;       // pinned host staging buffer for n-sized transfers (start vector, Ritz vectors)
  size_t stage_cap = 0;          // bytes
  void* ensure_stage(size_t bytes);
  void* h_cb = nullptr;          // pinned [in | out] buffers of the host-callback operator
  size_t cb_cap = 0;
  hipEvent_t ev_cb = nullptr;    // recorded after the upload of a callback result
  hipEvent_t stop_next = nullptr;  // set by the loop in front of an operator application whose kernel publishes an iteration's scalars: that
                                   // launch completes the event itself (LL_LAUNCH_STOP) and clears the field; still set afterwards = not taken
  void* ensure_cb_stage(size_t bytes);
  void sync();
};

// ---------------------------------------------------------------- operator
struct ll_operator {
  enum Kind { CSR, HOST_CB, DEV_CB, DENSE, STENCIL } kind = CSR;
  bool is_complex = false;
  int elem_bytes = 8;  // sizeof(T): 4 float, 8 double / complex float, 16 complex double
  ll_context* ctx = nullptr;
  int64_t n = 0, n_local = 0, row_begin = 0, nnz = 0;
  int64_t n_shard = 0;  // padded shard length used by the all-gather (= n when not sharded)
  double inf_norm = -1.0;  // max absolute row sum of the local rows (-1: unknown)
  // CSR
  void* d_row_ptr = nullptr;  // int32 or int64
  bool rp64 = false;
  int32_t* d_col = nullptr;
  void* d_val = nullptr;
  bool owns_arrays = true;
  int32_t* d_tile_rows = nullptr;  // ntiles+1 row boundaries of the SpMV tiles
  int ntiles = 0;
  // Sharded contexts, CSR-stream selected: the same rows split by column ownership (capi.cpp build_csr_split) — the
  // entries over the rank's OWN columns (indices rebased to the local shard; their product needs no exchange and runs
  // under the all-gather) and the entries over the other ranks' columns (global indices into the gathered vector).
  bool csr_split = false;
  void *d_rp_own = nullptr, *d_rp_rem = nullptr;  // row offsets, int32 or int64 like d_row_ptr
  int32_t *d_col_own = nullptr, *d_col_rem = nullptr;
  void *d_val_own = nullptr, *d_val_rem = nullptr;
  int32_t *d_tiles_own = nullptr, *d_tiles_rem = nullptr;
  int ntiles_own = 0, ntiles_rem = 0;
  // propagation-blocked image of the same matrix (spmv_pb.hip pb_phase1 / pb_phase2)
  int spmv_kind = 0;                 // LL_SPMV_*
  float tune_ms[3] = {-1.f, -1.f, -1.f};  // what the creation-time autotune measured per LL_SPMV_* kernel (-1: not timed)
  int pb_ncb = 0, pb_nrb = 0, pb_cb_cols = 0, pb_rb_rows = 0;   // cb_cols = longest column block (LDS sizing)
  int64_t* d_pb_segq = nullptr;      // [ncb][nrb+1] entry offsets of the segments in column-block order
  int64_t* d_pb_segdest = nullptr;   // [ncb][nrb]   position of each segment in row-block order
  int64_t* d_pb_rptr = nullptr;      // [nrb+1]      entry offsets of the row blocks in row-block order
  int64_t* d_pb_xoff = nullptr;      // [ncb]        element offset of the block's x slice in ITS source buffer
  int32_t* d_pb_ncols = nullptr;     // [ncb]        columns of the block
  void* d_pb_arena = nullptr;        // the one allocation that holds the four big streams below (interior pointers)
  size_t pb_arena_bytes = 0, pb_arena_static_bytes = 0;  // its size; the part in front of the product buffer
  void* d_pb_val = nullptr;          // values, column-block order
  uint16_t* d_pb_col = nullptr;      // local column, column-block order
  uint16_t* d_pb_row = nullptr;      // local row, row-block order
  void* d_pb_prod = nullptr;         // product buffer P (nnz elements of T), row-block order
  int16_t* d_pb_rexp = nullptr;      // LL_PB_PHASE2=fixed: exponent of every local row's absolute sum
  double* d_pb_blockmax = nullptr;   // LL_PB_PHASE2=fixed: max |x| per column block, left by phase 1
  void* d_pb_diag = nullptr;         // a_ii of every local row (T): the diagonal is kept outside the streams (spmv_pb.hip pb_diag_kernel)
  int64_t pb_entries = 0;            // padded entry count of the image
  int pb_phase2 = 4;                 // form of phase 2 (ll::LL_PB_FIXED / _ORDERED / _ATOMIC): set when the image is built,
                                     // changed by ll_op_set_accuracy (both forms read the same image)
  int accuracy_req = 0;              // LL_ACCURACY_* asked for at creation (ll_csr_options.accuracy); 0 = the environment decides
  // 2-D tiled image (spmv_pb.hip, tl_*; LL_SPMV_TILED): row blocks with the y slice in LDS, each walking its non-empty
  // column tiles of 16 KiB of x; entries = value (pre-scaled by the row's exponent) + packed 16-bit local column / row
  int tl_nrb = 0, tl_rb_rows = 0, tl_ncb = 0;
  int tl_n_interior = 0;             // sharded: row blocks whose tiles are all own-column tiles (first in d_tl_rbmap; they run under the all-gather)
  int32_t* d_tl_rbmap = nullptr;     // sharded: [nrb] row blocks in launch order (interior first); nullptr on one GPU
  int64_t tl_entries = 0, tl_tiles = 0;
  int32_t* d_tl_first = nullptr;     // [nrb + 1]     first tile of each row block in the tile list
  int32_t* d_tl_col = nullptr;       // [ntiles]      column tile index
  int64_t* d_tl_quad = nullptr;      // [ntiles + 1]  first quad (4 entries) of each tile in the entry stream
  void* d_tl_val = nullptr;          // values in tile order
  uint32_t* d_tl_idx = nullptr;      // local column | local row << 16
  int16_t* d_tl_rexp = nullptr;      // exponent of every row's absolute sum (the scale the values were divided by)
  double* d_tl_xmax = nullptr;       // maxima of |x| the kernel folds (per workgroup of the pre-pass; sharded: one per rank), then scratch of the own-shard pre-pass
  bool tl_ordered = false;           // the tiled kernel sums in floating point, the waves in turn (component-wise class) instead of in fixed point
  // Column-block table order: the blocks over the rank's OWN columns first (their x slice is the local shard, no
  // exchange needed), then, gather chunk by gather chunk, the blocks over the other ranks' columns (x slice in the
  // gathered buffer).  One phase-1 launch per range, so own-column work runs under the all-gather.
  int pb_threads1 = 1024;                      // lanes per workgroup of phase 1 (1024; 512 for the thin column blocks of a sharded image)
  bool pb_xpre = true;                         // fixed-point phase 2 requests the epilogue's x_i before its stream (LL_PB_XPRE)
  int pb_own_count = 0;                        // table range [0, own_count)
  int pb_chunk_first[ll::kMaxGatherChunks] = {0};  // remote blocks of gather chunk c: [first, first + count)
  int pb_chunk_count[ll::kMaxGatherChunks] = {0};
  ll::GatherPlan gather;             // how a sharded vector is all-gathered when the PB kernels are selected
  // dense row-major block (kind DENSE): n_local x n values of T
  void* d_dense = nullptr;
  // lattice operator (kind STENCIL)
  ll_stencil_desc st = {};
  int64_t st_stride[3] = {0, 0, 0};  // flattened-index stride of each dimension (last index fastest)
  int64_t st_halo = 0;               // sites of one hyperplane = reach of the operator in the flattened index
  void* d_onsite = nullptr;          // n_local on-site terms in the real type of T (nullable)
  // callbacks
  ll_host_mv_mul_z host_fn = nullptr;  // every host callback is stored under the void* signature
  ll_dev_mv_mul dev_fn = nullptr;
  void* user = nullptr;
  ll_operator() = default;
  ll_operator(const ll_operator&) = delete;
  ll_operator& operator=(const ll_operator&) = delete;
  ~ll_operator();  // frees every device array the operator owns (capi.cpp)
};

namespace ll {

// ---------------------------------------------------------------- kernel launchers (kernels.hip)
// All launchers enqueue on `s` and return immediately.

// Deferred normalisation (a8 folded into the next a1; single-GPU whole-loop drivers, operators that gather x themselves):
// the operator kernel is handed the UNNORMALISED vector w_k together with the `nparts` partial sums of ||w_k||^2.  Every
// workgroup folds them in the same fixed order, works with u_k = w_k / ||w_k|| (the operator is linear: the factor is
// applied to the row sums and to x_i), writes u_k to u_out (the basis slot: every later reader wants it normalised) and
// workgroup 0 stores ||w_k||^2 to *c1_out and iteration k's four scalars to the pinned host slot — the work of
// scale_publish_kernel without its launch and without its read of w.
template <typename T> struct ScaleIn {
  const double* partials = nullptr;  // null: x is already normalised (everything below is ignored)
  int nparts = 0;
  double* c1_out = nullptr;
  const double* alpha = nullptr;
  const double* c0 = nullptr;
  double* host = nullptr;
  T* u_out = nullptr;
};

// y = A x_full(cols) + offset * x_local ; dot_partials (nullable): one double per workgroup, Re<x_local, y>.
// Returns the number of partials written.
// part: 0 = the whole image; 1 / 2 = the own-column / other-columns half of a column-split image (ll_operator::csr_split).
template <typename T>
int launch_spmv(const ll_operator& op, const T* x_full, const T* x_local, T* y, double offset, double* dot_partials,
                hipStream_t s, const ScaleIn<T>* sc = nullptr, int part = 0);
// build helpers of the column split (kernels.hip): own-column entries per row; scatter into the two halves
template <typename T> void launch_csr_count_own(const ll_operator& op, int32_t* own_cnt, hipStream_t s);
template <typename T> void launch_csr_split(const ll_operator& op, hipStream_t s);
// Same contract, propagation-blocked kernels (op.spmv_kind == LL_SPMV_PB; spmv_pb.hip): phase 1 over the own-column
// blocks (x slices from x_own: the local shard readable up to the shard stride), then over every gather chunk's remote blocks (x slices from x_gathered, laid out
// per op.gather), then phase 2.  The pieces are exposed so that the sharded driver can run the own-column part under
// the all-gather and each chunk's part as soon as that chunk has arrived.
// xnorm2 (nullable device scalar): x is an UNNORMALISED vector w with ||w||^2 = *xnorm2; the kernels work with w / ||w||
// (lagged Gram-Schmidt, kernels.hip).
template <typename T>
int launch_spmv_pb(const ll_operator& op, const T* x_gathered, const T* x_own, const T* x_local, T* y, double offset,
                   double* dot_partials, hipStream_t s, const double* xnorm2 = nullptr);
template <typename T>
void launch_pb_phase1(const ll_operator& op, int blk_first, int blk_count, const T* xsrc, hipStream_t s,
                      const double* xnorm2 = nullptr);
template <typename T>
int launch_pb_phase2(const ll_operator& op, const T* x_local, T* y, double offset, double* dot_partials, hipStream_t s,
                     const double* xnorm2 = nullptr);
// forms of PB phase 2 (Tuning::pb_phase2, ll_operator::pb_phase2)
constexpr int LL_PB_ATOMIC = 0, LL_PB_ORDERED = 1, LL_PB_FIXED = 4;
// Build the propagation-blocked image on the device from the operator's CSR arrays (false: shape not supported).
template <typename T> bool pb_build_device(ll_operator* op);
// The 2-D tiled kernel for matrices with column locality (spmv_pb.hip): same contract as launch_spmv on a single GPU
// (x = the whole vector); build returns false when the matrix is not eligible (too many column tiles per row block).
template <typename T> bool tl_build_device(ll_operator* op);
void tl_release(ll_operator* op);
template <typename T>
int launch_spmv_tiled(const ll_operator& op, const T* x, T* y, double offset, double* dot_partials, hipStream_t s,
                      const double* xnorm2 = nullptr);
// Sharded form (engine.cpp): own-shard maximum, then the two launches (own-column row blocks under the all-gather, the rest behind it).
template <typename T> void launch_tl_xmax_local(const ll_operator& op, const T* x_own, hipStream_t s);
int tl_xmax_local_slot();
template <typename T>
int launch_spmv_tiled_pass(const ll_operator& op, int pass, const T* x, int64_t col0, int64_t col_end, const T* x_local, T* y,
                           double offset, double* dot_partials, hipStream_t s, const double* xnorm2, int n_xmax);
// Column range check + max absolute row sum of the local rows (sets op->inf_norm), on the device.
template <typename T> void csr_check_device(ll_operator* op);
// Same contract for the dense row block (op.kind == DENSE).
template <typename T>
int launch_dense_mv(const ll_operator& op, const T* x_full, const T* x_local, T* y, double offset, double* dot_partials,
                    hipStream_t s, const ScaleIn<T>* sc = nullptr, int part = 0);
// Lattice operator (op.kind == STENCIL): site li of the shard reads x at li + off, |off| <= op.st_halo, from
// halo_lo[st_halo + j] for j < 0, x_local[j] for 0 <= j < n_local and halo_hi[j - n_local] beyond.
template <typename T>
int launch_stencil(const ll_operator& op, const T* x_local, const T* halo_lo, const T* halo_hi, T* y, double offset,
                   double* dot_partials, hipStream_t s, const ScaleIn<T>* sc = nullptr);
// y += offset * x ; partials of Re<x,y> (post-pass for callback operators).
template <typename T>
int launch_offset_dot(int64_t n, const T* x, T* y, double offset, double* dot_partials, hipStream_t s);

// out[j] = sum_b partials[b*ncols + j], j < ncols (deterministic tree, fixed order).
// last_out (nullable) redirects the last column.
void launch_reduce_cols(const double* partials, int nparts, int ncols, double* out, double* last_out, hipStream_t s);
void launch_copy_scalar(double* dst, const double* src, hipStream_t s);
// reduce_one + publish in one launch: out[0] = sum(partials); host[0..4) = {*alpha, sum, *c0, sum} (alpha, c0 nullable)
void launch_reduce_publish(const double* partials, int nparts, double* out, const double* alpha, const double* c0,
                           double* host_mapped, hipStream_t s);
void launch_set_scalar(double* dst, double value, hipStream_t s);
// *c1 = max(0, *c0 - sum_i h[i]^2)  (one workgroup, fixed order)
// c0 = *c0_src (||w||^2 before the pass, from the all-reduced buffer), c1 = max(c0 - sum_i h_i^2, 0); host_mapped (nullable):
// the iteration's four scalars (alpha, c1, c0, c1) go to the pinned host slot in the same launch.
void launch_derive_norm(const double* c0_src, const double* h, int count, double* c0, double* c1, const double* alpha,
                        double* host_mapped, hipStream_t s);

// Multi-dot with optional fused three-term update.
//   if (three_term) w = w - beta*u_prev - alpha*u_cur   (u_prev nullable; alpha = *alpha_dev; beta = beta_from(norms_prev))
//   partial columns: for every basis vector j (segments in order): <u_j, w> (1 or 2 doubles), then ||w||^2.
// pred (nullable): the launch is a no-op unless the second pass is due according to *pred.
// Returns grid size (= number of partial rows); ncols = reals*nb + 1.
template <typename T> struct ThreeTerm {
  const T* u_prev;      // nullable (k == 1)
  const T* u_cur;       // nullable => no three-term update
  const double* alpha;  // device scalar
  NormRefs prev;        // norms of the previous iteration: beta = sqrt(final norm^2)
  // Deferred alpha (single-GPU whole-loop drivers): alpha has NOT been folded yet; every workgroup of the multi-dot
  // sums the operator kernel's `alpha_nparts` partials itself (same fixed order everywhere) and workgroup 0 stores the
  // result to alpha_out for the consumers that follow (publish).  One launch per iteration less.
  const double* alpha_partials = nullptr;
  int alpha_nparts = 0;
  double* alpha_out = nullptr;
};
// small_bytes: vectors shorter than this many bytes take the small-vector geometry (Tuning::blas_small_bytes).
template <typename T>
int launch_mdot(int64_t n, T* w, const BasisSegs<T>& segs, const ThreeTerm<T>& tt, const NormRefs* pred,
                double* partials, int64_t small_bytes, hipStream_t s);
// Lagged block Gram-Schmidt (kernels.hip, lagged_kernel): one sweep that applies the previous iteration's update to
// r -> u_out, forms w = w - alpha r/beta - beta u_prev minus the compensation of the perturbed operator input, and all
// coefficients <u_j, .> (segments, then u_out) + ||w||^2.  Partial columns: reals * (nb + 1) + 1 per workgroup.
// tt.u_cur is ignored.  g: reals * nb coefficients of r, t: reals * (nb + 1) + 1 values, both from launch_lagged_fold.
// LDS: 4 ncols doubles (the waves' partial columns; g and t are read through the scalar cache), 160 KB per workgroup:
// reals * nb <= kLaggedMaxCols.  Streaming geometry only.
constexpr int kLaggedMaxCols = 5000;
template <typename T> struct Lagged {
  const T* r;
  T* u_out;
  const double* g;
  const double* t;
  const double* beta2;
};
// pieces: 0 = pick the strip geometry from the vector length (16-byte pieces per lane: 4 from kLaggedFullStrips 16 KiB
// strips up, else 2); 2 / 4 force it (key lagged_pieces).
constexpr int kLaggedFullStrips = 200;
template <typename T>
int launch_lagged(int64_t n, T* w, const BasisSegs<T>& segs, const Lagged<T>& lg, const ThreeTerm<T>& tt, double* partials,
                  int pieces, int64_t small_limit, hipStream_t s);  // vectors below small_limit bytes: lagged_small_kernel
// The pair form (two iterations per sweep; kernels.hip, "pair" section; tools/pair_gs_model.py is the executable specification).
// Streaming geometry only; 2 * reals * K + 5 * reals + 1 <= kLaggedMaxCols columns per workgroup (K stored columns).
constexpr double kPairGate = 1e-8;  // largest relative coefficient the pair form accepts in double precision (second-order terms
                                    // stay below 1e-16; float storage: 2e-4, engine.cpp)
template <typename T>
int launch_pair_three_term(int64_t n, T* y, const T* x, const T* p, double* e, const double* e_partials, int e_nparts,
                           const double* cx2, const double* cp2, double* partials, bool colmajor,
                           hipStream_t s);  // y <- y - (e / sqrt(cx2)) x - sqrt(cx2 / cp2) p; partials [grid][1 + reals] (colmajor:
                                            // [1 + reals][grid]): |y|^2, <p, y>; e_partials (nullable): the operator kernel's
                                            // partial sums of e, folded here into *e
// P Lanczos vectors behind L locked eigenvectors (eigenvalues lambda[0..L) of the operator the loop applies): K = L + P columns
void launch_pair_predict(int P, int L, int reals, const double* g1, const double* g2, const double* rho1sq, const double* rho2sq,
                         const double* gam, double* n3sq, const double* d13_partials, int d13_nparts, const double* e1, double* e2,
                         const double* e2_partials, int e2_nparts, const double* hist_alpha, const double* hist_beta,
                         const double* lambda, double* p3, double* p4, hipStream_t s);
// r4 holds y2 = A (r3 / |r3|) on entry; the sweep forms r4 = y2 - (e2 / |r3|) r3 - (|r3| / rho2) r2 on the fly
// One workgroup keeps 4 x (2 reals Pl + 5 reals + 1) columns in LDS: a sweep over more stored vectors than that is split into
// launches over consecutive groups of them (same results bit for bit, kernels.hip); part4: scratch n-vector for the hand-over
// (touched only when there is more than one group).
constexpr int kPairFirst = 1, kPairLast = 2;
template <typename T> constexpr int pair_sweep_max_vecs() {
  return (kLaggedMaxCols - 5 * scalar_traits<T>::reals - 1) / (2 * scalar_traits<T>::reals);
}
template <typename T>
int launch_pair_sweep(int64_t n, const std::vector<BasisSegs<T>>& groups, int P, const T* r1, const T* r2, const T* r3, T* r4,
                      T* uP_out, T* uQ_out, T* part4, const double* g1, const double* g2, const double* gam, const double* p4,
                      const double* rho1sq, const double* rho2sq, const double* e2, const double* n3sq, double* partials, int pieces,
                      hipStream_t s, const T* const* vtab = nullptr,  // vtab (device; column c -> pointer of stored vector c): the software-
                      bool force_pipeline = false);                   // pipelined kernel on streaming vectors (force: on any); null: the reference kernel
// The same sweep in the small-vector geometry (pair_small_kernel: four waves per 1 KiB strip split the stored vectors; vectors of
// 320 KiB .. 1 MiB).  One launch over ONE group of segments; false when the columns do not fit one workgroup's LDS (nothing launched).
template <typename T>
bool launch_pair_sweep_small(int64_t n, const BasisSegs<T>& segs, int P, const T* r1, const T* r2, const T* r3, T* r4, T* uP_out,
                             T* uQ_out, const double* g1, const double* g2, const double* gam, const double* p4, const double* rho1sq,
                             const double* rho2sq, const double* e2, const double* n3sq, double* partials, int* grid_out,
                             hipStream_t s);
template <typename T> bool pair_small_fits(int P);  // P stored columns fit the small-geometry sweep's LDS
// tab[start + i] = base + i * ld, i < count (the pointer table of the pipelined sweeps; one launch per slab)
template <typename T> void launch_fill_ptrs(const T** tab, int start, int count, const T* base, int64_t ld, hipStream_t s);
void launch_pair_fold(const double* m, int P, int L, int reals, const double* lambda, const double* p4, const double* g2, const double* gam,
                      const double* rho2sq, const double* n3sq, const double* e1, const double* e2, double* rec3, double* rec4,
                      double* nxt, double* hist_alpha, double* hist_beta, double* scratch, double* host_a, double* host_b,
                      double* gate_a, double* gate_b, hipStream_t s, hipEvent_t stop = nullptr);
// Fold of a lagged iteration (K = L + k columns: L locked eigenvectors with eigenvalues lambda[0..L), then k Lanczos
// vectors; m: reals * K folded columns, *c0 = ||w||^2, copied to *c0_out): compensated coefficients in place,
// *c1 = *c0 - |g|^2, t_out (reals * (K + 1) + 1) for the next sweep, alpha / beta appended to hist_*[k - 1], *alpha
// replaced by its corrected value, the iteration's four scalars published.  prev_* = nullptr after a clean iteration.
void launch_lagged_fold(double* m, int K, int L, int reals, double* t_out, const double* c0, double* c0_out, double* c1,
                        double* alpha, const double* prev_g, const double* prev_t, const double* prev_c1,
                        double* hist_alpha, double* hist_beta, const double* lambda, double* host_mapped, hipStream_t s, hipEvent_t stop = nullptr);
// w -= sum_j h_j u_j over the segments; partial ||w||^2 per workgroup. h: reals*nb doubles on the device.
template <typename T>
int launch_maxpy(int64_t n, T* w, const BasisSegs<T>& segs, const double* h, const NormRefs* pred, double* partials,
                 int64_t small_bytes, hipStream_t s);
// The same update with the fold of the multi-dot's partials ([mparts][reals*nb + 1]) done inside the kernel (small-vector
// geometry, small grids): h_out receives the coefficients, *c0_out (nullable) the ||w||^2 column; `partials` (the norm
// partials of the result) must not alias mdot_partials.  false: not applicable, nothing was launched.
template <typename T>
bool launch_maxpy_folding(int64_t n, T* w, const BasisSegs<T>& segs, const double* mdot_partials, int mparts, double* h_out,
                          double* c0_out, double* partials, int64_t small_bytes, int* grid_out, hipStream_t s);
// v *= factor, factor = a (host value) when norms == nullptr, else 1/sqrt(final norm^2).
template <typename T> void launch_scale(int64_t n, T* v, double a, const NormRefs* norms, hipStream_t s);
// scale fused with the fold of the post-pass norm and the publish step (single-GPU whole-loop drivers): every workgroup
// folds the `nparts` norm partials in the same fixed order, v *= 1/sqrt(sum); workgroup 0 stores the sum to *out and the
// iteration's four scalars (alpha, sum, c0, sum) to the pinned host slot.
// a8 fused with launch_derive_norm (sharded whole-loop drivers): v *= 1 / sqrt(max(*c0_src - sum_i h_i^2, 0)).
template <typename T>
void launch_scale_derive(int64_t n, T* v, const double* c0_src, const double* h, int count, double* c0, double* c1,
                         const double* alpha, double* host_mapped, hipStream_t s);
// src (nullable): read the unnormalised vector from there instead of from v (out of place).
template <typename T>
int launch_scale_publish(int64_t n, T* v, const double* partials, int nparts, double* out, const double* alpha,
                         const double* c0, double* host_mapped, hipStream_t s, const T* src = nullptr);
// Plain three-term update with host scalars (primitive API).
template <typename T>
void launch_three_term(int64_t n, T* w, const T* u_prev, const T* u_cur, double beta, double alpha, hipStream_t s);
// partials of <a,b> (reals per workgroup). Returns grid.
template <typename T> int launch_dot(int64_t n, const T* a, const T* b, double* partials, hipStream_t s);
// out_r = sum_{k=m-1..0} coeff[r*m+k] * u_k, r < nout (<= 8); coeff on device, type T. scale (nullable device
// scalar) multiplies every output.
template <typename T>
void launch_gemv_basis(int64_t n, int64_t m, const BasisSegs<T>* segs, int nlaunch, int nout, const T* coeff, T* out,
                       int64_t ld_out, hipStream_t s);
// small helpers
// h_acc += h_add when the second pass ran
void launch_accumulate_h(double* h_acc, const double* h_add, int count, const NormRefs* pred, hipStream_t s);
// out_host_visible[0] = *alpha (0 if null), [1] = final norm^2, [2] = c0, [3] = c1  (pinned, device-mapped memory)
void launch_publish(double* out_mapped, const double* alpha, const NormRefs& norms, hipStream_t s);

// streaming kernels of ll_bandwidth_probe (kernels.hip): a read-only sum and a copy, 16-byte accesses
void launch_bw_read(const void* a, size_t bytes, double* out, int grid, hipStream_t s);
void launch_bw_copy(const void* a, void* b, size_t bytes, int grid, hipStream_t s);

// ---------------------------------------------------------------- host tridiagonal solver (tridiag_host.cpp)
// Flat-array implicit-shift QR; same arithmetic as the reference's (TRI:151-343) so that
// convergence decisions coincide.  q (nullable) row-major m x m, row j = eigenvector j.
int64_t tridiag_qr(int64_t m, const double* alpha, const double* beta, double* ev, double* q);
// Unit eigenvectors for nw given eigenvalues by inverse iteration (O(m) each); out = nw rows of m entries.
void tridiag_inverse_iteration(int64_t m, const double* alpha, const double* beta, int64_t nw, const double* lambdas,
                               double* out);
// k-th smallest eigenvalue by Sturm bisection (TRI:22-88).
double tridiag_bisect(int64_t m, const double* alpha, const double* beta, int64_t k);
// The same for nk roots at once (bit-identical to nk calls of tridiag_bisect; interleaved recurrences, ~5x faster).
void tridiag_bisect_multi(int64_t m, const double* alpha, const double* beta, int nk, const int64_t* ks, double* out);

}  // namespace ll
