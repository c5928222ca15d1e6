/*
 * lanczos_hip.h — C ABI of the MI355X-native Lanczos hot path (liblanczos_hip.so).
 *
 * This is the drop-in boundary.  The reference (mrcdr/lambda-lanczos) is a
 * header-only C++ template library, so its "FFI" is the template API itself; templates cannot cross a
 * C ABI, therefore every entry point exists per scalar type with the BLAS-style suffixes
 *     _d  = double                  (reference T = double)
 *     _z  = double complex          (reference T = std::complex<double>, interleaved re,im)
 *     _s  = float                   (reference T = float)
 *     _c  = float complex           (reference T = std::complex<float>, interleaved re,im)
 *   (the _s/_c declarations are at the end of this file: same arguments as _d/_z with float data pointers; every
 *   scalar — offsets, eps, alpha, eigenvalues, norms — stays double, and all reductions are accumulated in double,
 *   which is at least the accuracy of the reference's float arithmetic.  long double has no device counterpart.)
 * and the C++ facade include/lambda_lanczos_hip/{lambda_lanczos,exponentiator}.hpp re-creates
 * lambda_lanczos::LambdaLanczos<T> / Exponentiator<T> on top by tag dispatch.
 *
 * Reference citations (paths relative to the reference repository root):
 *   LL  = include/lambda_lanczos/lambda_lanczos.hpp
 *   EX  = include/lambda_lanczos/exponentiator.hpp
 *   LA  = include/lambda_lanczos/util/linear_algebra.hpp
 *   TRI = include/lambda_lanczos/lambda_lanczos_tridiagonal_impl.hpp
 *
 * Conventions
 *   - every function returns an int status: 0 = LL_OK, otherwise an LL_ERR_* code; ll_last_error()
 *     returns a thread-local human readable message for the last failure.  (The reference has no error
 *     channel at all: asserts only, LA:31, EX:88 — this is additive.)
 *   - pointers named *_dev are device (HBM) pointers owned by the caller (ll_malloc or any HIP
 *     allocation on the context's device); pointers named *_host are host memory.
 *   - all device work is enqueued on the context's HIP stream; primitive calls are asynchronous unless
 *     they return a host scalar, whole-loop calls (ll_lanczos_run_*, ll_expo_run_*) are synchronous.
 *   - a context is not re-entrant; one context per host thread (the reference is single threaded and has
 *     no globals, SURVEY 8b "Threading").
 *   - no CPU fallback exists: without a HIP device every call that touches the device fails with
 *     LL_ERR_HIP.
 */
#ifndef LANCZOS_HIP_H_
#define LANCZOS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LL_VERSION_MAJOR 0
#define LL_VERSION_MINOR 5  /* 2: ll_lanczos_params.init_vector_dev, device pointers accepted for n-sized buffers
                             * 3: ll_run_stats.lagged_iterations + reserved tail (the struct grew: code compiled against a
                             *    minor-2 header must be rebuilt — ll_abi_check refuses it), ll_ctx_reload_env, ll_abi_check
                             * 4: additive — ll_csr_options / ll_op_create_csr_opt_*, ll_op_set_accuracy, ll_op_accuracy, the tiled
                             *    SpMV kernel id; no struct changed, callers built against minor 3 keep working
                             * 5: additive — ll_ctx_set_tuning (the test hooks and geometry overrides left the environment),
                             *    ll_comm_transport, ll_bandwidth_probe, ll_op_tiled_layout; no struct changed */

enum {
  LL_OK = 0,
  LL_ERR_INVALID = 1, /* bad argument */
  LL_ERR_HIP = 2,     /* a HIP runtime call failed (or no device) */
  LL_ERR_RCCL = 3,    /* RCCL missing or a collective failed */
  LL_ERR_ALLOC = 4,   /* out of device or host memory */
  LL_ERR_CALLBACK = 5 /* a user callback reported failure */
};

typedef struct ll_context ll_context; /* device + stream + workspace (+ RCCL communicator) */
typedef struct ll_operator ll_operator; /* the mv_mul plugin: device CSR, host callback or device callback */

/* ------------------------------------------------------------------ library / context */

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* ll_last_error(void);
/* LL_VERSION_MAJOR * 1000 + LL_VERSION_MINOR */
int ll_version(void);
/* ABI handshake: pass the LL_VERSION_* the CALLER was compiled with and sizeof(ll_run_stats) / sizeof(ll_lanczos_params) as
 * the caller sees them.  LL_OK when the loaded library lays the structs out the same way; LL_ERR_INVALID (with a message
 * naming both versions) otherwise — the library fills ll_run_stats and reads the parameter structs by ITS layout, so a
 * stale binary would be overrun.  Minors that only ADD entry points accept older callers (any minor from 3 up to the library's
 * own, same struct sizes).  The handshake exists since minor 3: a binary built against an older header never calls it and is
 * not protected — it must be rebuilt.  The C++ facade (both Context constructors) and the Python binding call it once per
 * process. */
int ll_abi_check(int caller_major, int caller_minor, size_t sizeof_run_stats, size_t sizeof_lanczos_params);
#define LL_ABI_CHECK() ll_abi_check(LL_VERSION_MAJOR, LL_VERSION_MINOR, sizeof(ll_run_stats), sizeof(ll_lanczos_params))

/* Create a context on HIP device `device` with its own non-blocking stream. */
int ll_ctx_create(int device, ll_context** out);
/* Same, but enqueue on an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = the
 * null stream.  The stream stays owned by the caller. */
int ll_ctx_create_on_stream(int device, void* hip_stream, ll_context** out);
int ll_ctx_destroy(ll_context* ctx);
/* The library's LL_* environment switches are read ONCE, when a context is created, and
 * copied into the operators created on it — never on a launch path.  This reads them again into an existing context
 * (test suites and tuning scripts that flip a switch inside one process); operators that already exist keep theirs. */
int ll_ctx_reload_env(ll_context* ctx);
/* UNSTABLE (tests, probes and A/B measurements; keys may change between minors): set one tuning field of THIS context by key,
 * on top of what the environment said.  The environment carries only the user-facing switches of INTEGRATION.md section 8; block
 * geometries, forced code paths and the hooks of the test suite ("pb_block", "pair_split", "lagged_min_bytes", "tl_force",
 * "force_rp64", ... — the list is INTEGRATION.md section 8, second table) exist only here, so that a stray variable in a user's
 * environment can never change the numerics path.  Every user-facing switch is also a key (its name in lower case without the
 * LL_ prefix: "spmv_kernel", "pair_gs", ...).  value NULL removes the setting again; an unknown key is LL_ERR_INVALID.  Like the
 * environment switches it takes effect for operators / runs created afterwards. */
int ll_ctx_set_tuning(ll_context* ctx, const char* key, const char* value);
/* hipStream_t of the context (for callers that enqueue their own work in between). */
int ll_ctx_stream(ll_context* ctx, void** hip_stream_out);
/* Block until the context's stream is idle. */
int ll_ctx_synchronize(ll_context* ctx);
/* The Krylov-basis slabs of a finished run stay cached in the context so that the next run does not pay
 * hipMalloc/hipFree of tens of GB; this returns them to the device (ll_ctx_destroy does it too). */
int ll_ctx_release_cache(ll_context* ctx);

/* Device-side stopwatch: HIP events recorded on the context's stream (what bench.py uses to time a kernel on the
 * stream it is launched on).  ll_timer_stop waits for the stream and returns the milliseconds since ll_timer_start. */
int ll_timer_start(ll_context* ctx);
int ll_timer_stop(ll_context* ctx, double* ms_out);
/* Streaming ceilings of the device at hand, measured now on the context's stream with two plain kernels over `bytes` (>= 1 MiB;
 * two scratch buffers of that size are allocated and freed): a read-only stream (GB/s of bytes read) and a copy (GB/s of bytes
 * read + written).  bench.py reports its roofline fractions against these next to the 8 TB/s spec peak (SURVEY 8d "Bound"). */
int ll_bandwidth_probe(ll_context* ctx, size_t bytes, double* read_GBps, double* copy_GBps);

/* ------------------------------------------------------------------ multi-GPU
 * One process per GPU.  Rows of A and every n-vector are partitioned 1-D and contiguously:
 * rank r owns rows [row_begin, row_begin + n_local).  The library talks RCCL directly on the context's
 * stream: one all-gather of the current Lanczos vector per SpMV and small all-reduces for the dot products.
 * librccl is loaded lazily (dlopen), single-GPU use needs no RCCL. */

#define LL_UNIQUE_ID_BYTES 128
/* Rank 0 creates the id and distributes the 128 bytes by any host mechanism (MPI, torch.distributed, file). */
int ll_comm_unique_id(void* id_out_128);
/* Collective over all ranks. After it, operators created on this context are row shards. */
int ll_comm_init(ll_context* ctx, const void* id_128, int rank, int n_ranks);
int ll_comm_rank(ll_context* ctx, int* rank, int* n_ranks);
/* ll_comm_init ends with a self-check (every rank's tag through an all-gather on the communication stream, a sum of
 * ones through an all-reduce on the compute stream) and fails with LL_ERR_RCCL when the communicator does not span
 * n_ranks ranks in rank order.  This returns how many rank tags arrived (== n_ranks after a successful init; 1 without
 * a communicator) so that a launcher can print it next to its results. */
int ll_comm_ranks_seen(ll_context* ctx, int* out);
/* Which transport answers this context's collectives, as text: "rccl" (the communicator ll_comm_init created with librccl),
 * "plugin:<path>" (the shared object LL_COMM_PLUGIN named), "attached" (ll_comm_attach) or "none" (no communicator).  A launcher
 * prints it next to ll_comm_ranks_seen: a rank count alone does not say that RCCL ran. */
int ll_comm_transport(ll_context* ctx, char* out, size_t cap);
/* The contiguous row range of `rank`: shards of ceil(n / n_ranks) rows (the last ones may be shorter or empty).
 * Sharded operators and vectors must use exactly these ranges (the all-gather relies on equal shard strides). */
int ll_partition(int64_t n, int n_ranks, int rank, int64_t* row_begin, int64_t* n_local);

/* ------------------------------------------------------------------ device memory helpers */

int ll_malloc(ll_context* ctx, size_t bytes, void** dev_out);
int ll_free(ll_context* ctx, void* dev);
int ll_memcpy_h2d(ll_context* ctx, void* dst_dev, const void* src_host, size_t bytes); /* synchronous */
int ll_memcpy_d2h(ll_context* ctx, void* dst_host, const void* src_dev, size_t bytes); /* synchronous */
int ll_memset(ll_context* ctx, void* dst_dev, int byte, size_t bytes);

/* ------------------------------------------------------------------ the operator plugin: mv_mul (LL:120-126, EX:35-41)
 *
 * Reference contract: std::function<void(const vector<T>& in, vector<T>& out)>, `out` zero-filled on entry,
 * accumulate or overwrite both legal, called once per Lanczos iteration with the unit-norm u[k-1]
 * (LL:242-243, EX:107-108).  Five realisations: */

/* (1) device-resident CSR (the reference ships no sparse format; this is the new operator of SURVEY 8a-a1).
 *     n_rows_local rows [row_begin, row_begin+n_rows_local) of a global n_cols x n_cols symmetric/Hermitian
 *     matrix; column indices are GLOBAL.  Single GPU: row_begin = 0, n_rows_local = n_cols.
 *     row_ptr_host has n_rows_local+1 entries starting at 0.  Host arrays stay owned by the caller and may be
 *     freed after the call; the library keeps device copies until ll_op_destroy. */
int ll_op_create_csr_d(ll_context* ctx, int64_t n_rows_local, int64_t n_cols, int64_t row_begin,
                       const int64_t* row_ptr_host, const int32_t* col_host, const double* val_host,
                       ll_operator** out);
int ll_op_create_csr_z(ll_context* ctx, int64_t n_rows_local, int64_t n_cols, int64_t row_begin,
                       const int64_t* row_ptr_host, const int32_t* col_host, const void* val_host /* re,im pairs */,
                       ll_operator** out);
/*     Same with arrays that already live on the device (e.g. built by a GPU generator). */
int ll_op_create_csr_dev_d(ll_context* ctx, int64_t n_rows_local, int64_t n_cols, int64_t row_begin,
                           const int64_t* row_ptr_dev, const int32_t* col_dev, const double* val_dev,
                           ll_operator** out);
int ll_op_create_csr_dev_z(ll_context* ctx, int64_t n_rows_local, int64_t n_cols, int64_t row_begin,
                           const int64_t* row_ptr_dev, const int32_t* col_dev, const void* val_dev,
                           ll_operator** out);

/*     {row, col, value} triplets, the format of the reference's sparse sample (src/samples/sample2_sparse.cpp:14-47):
 *     converted to CSR on the host (stable inside a row, duplicates kept as separate entries); single GPU only. */
int ll_op_create_coo_d(ll_context* ctx, int64_t n, int64_t nnz, const int32_t* rows_host, const int32_t* cols_host,
                       const double* vals_host, ll_operator** out);
int ll_op_create_coo_z(ll_context* ctx, int64_t n, int64_t nnz, const int32_t* rows_host, const int32_t* cols_host,
                       const void* vals_host, ll_operator** out);
/*     max_i sum_j |a_ij| over the LOCAL rows of a CSR/COO operator created from
 *     host arrays: a safe eigenvalue_offset bound, the idea of src/determine_eigenvalue_offset/
 *     determine_eigenvalue_offset.cpp:12-29 (which does not build upstream).  For "smallest" problems use
 *     eigenvalue_offset = -inf_norm so that the wanted Ritz value is large in magnitude. */
int ll_op_inf_norm(const ll_operator* op, double* out);

/* (2) unmodified user code: a host callback with exactly the reference semantics; costs one D2H + one H2D
 *     of an n-vector per iteration (SURVEY 8b "Operator contract").  Return non-zero from fn to abort. */
typedef int (*ll_host_mv_mul_d)(const double* in, double* out_zeroed, int64_t n, void* user);
typedef int (*ll_host_mv_mul_z)(const void* in, void* out_zeroed, int64_t n, void* user); /* also used by _c */
typedef int (*ll_host_mv_mul_s)(const float* in, float* out_zeroed, int64_t n, void* user);
int ll_op_create_host_d(ll_context* ctx, int64_t n, ll_host_mv_mul_d fn, void* user, ll_operator** out);
int ll_op_create_host_z(ll_context* ctx, int64_t n, ll_host_mv_mul_z fn, void* user, ll_operator** out);

/* (3) a device callback: fn enqueues out += A*in on `hip_stream` for device pointers (out zero-filled). */
typedef int (*ll_dev_mv_mul)(const void* in_dev, void* out_dev_zeroed, int64_t n_local, void* hip_stream,
                             void* user);
int ll_op_create_device_d(ll_context* ctx, int64_t n, ll_dev_mv_mul fn, void* user, ll_operator** out);
int ll_op_create_device_z(ll_context* ctx, int64_t n, ll_dev_mv_mul fn, void* user, ll_operator** out);

/* (4) dense row-major matrix, the operator of the reference's first sample and of its small known-answer tests
 *     (src/samples/sample1_simple.cpp:22-28; T1:130, T1:446-453): a_host holds n_rows_local x n_cols values of T,
 *     rows [row_begin, row_begin + n_rows_local) of the global matrix.  Sharded exactly like CSR (row block +
 *     all-gather of x).  One apply streams the matrix once: sizeof(T) * n_rows_local * n_cols bytes. */
int ll_op_create_dense_d(ll_context* ctx, int64_t n_rows_local, int64_t n_cols, int64_t row_begin,
                         const double* a_host, ll_operator** out);
int ll_op_create_dense_z(ll_context* ctx, int64_t n_rows_local, int64_t n_cols, int64_t row_begin,
                         const void* a_host, ll_operator** out);

/* (5) matrix-free lattice operator: the family of the reference's "dynamic matrix" sample and tests
 *     (src/samples/sample3_dynamic.cpp:17-22 and T1:265-273: open chain; T2:113-121: periodic ring; BASELINE
 *     config 2: 5-point Laplacian).  Sites of a row-major lattice dims[0] x .. x dims[ndim-1] (LAST index fastest,
 *     n = product of dims);
 *         (A x)(r) = (diag + onsite[r]) x(r) + sum_d ( t_d(r) x(r + e_d) + conj(t_d(r - e_d)) x(r - e_d) ),  t_d = hop[d]
 *     (times a position-dependent Peierls phase when phase_grad is set, see the struct)
 *     with open (periodic[d] = 0: the missing neighbour contributes nothing) or periodic boundaries per dimension.
 *     No matrix is stored: one apply moves 2 * sizeof(T) * n bytes (+ 8n for onsite) instead of the CSR image.
 *     Sharded contexts: flattened sites [row_begin, row_begin + n_local) per ll_partition; the exchange step is a
 *     HALO exchange of one lattice hyperplane (n / dims[0] sites) with the two neighbouring ranks instead of the
 *     all-gather of x; every shard must hold at least one hyperplane.
 *     hop_im must be 0 for the real types.  onsite_host_local: n_local real values, NULL = none. */
typedef struct ll_stencil_desc {
  int32_t ndim;        /* 1..3 */
  int32_t periodic[3];
  int64_t dims[3];
  double diag;
  double hop_re[3];
  double hop_im[3];
  /* Peierls phases (complex types only; all 0 = none): the bond r -> r + e_d carries
   *   hop[d] * exp(i * sum_e phase_grad[d][e] * c_e(r)),   c(r) = lattice coordinates of the bond's lower site r,
   * and conj of that in the other direction.  Example (BASELINE config 5, Landau gauge on an N x N torus with dims =
   * {y, x}): hop = {-1, -1}, phase_grad[1][0] = 2 pi 3 / N. */
  double phase_grad[3][3];
} ll_stencil_desc;
int ll_op_create_stencil_d(ll_context* ctx, const ll_stencil_desc* desc, int64_t row_begin, int64_t n_local,
                           const double* onsite_host_local, ll_operator** out);
int ll_op_create_stencil_z(ll_context* ctx, const ll_stencil_desc* desc, int64_t row_begin, int64_t n_local,
                           const double* onsite_host_local, ll_operator** out);

/* Which SpMV kernel a CSR operator uses (both are bit-reproducible run to run; their results agree to rounding IN THE
 * NORM-WISE SENSE stated below):
 *   LL_SPMV_CSR_STREAM  plain CSR, products staged in LDS; best when the x gathers hit L1/L2 (stencils, narrow bands).
 *   LL_SPMV_PB          propagation blocking: the same matrix re-ordered once at upload (on the device) so that one
 *                       SpMV is two fully coalesced streaming sweeps with x and y slices in LDS and no global gather;
 *                       best for matrices without column locality (BASELINE config 3).  On sharded contexts its
 *                       own-column part runs under the all-gather.
 *   LL_SPMV_TILED       2-D tiling for matrices WITH column locality (bands, stencils, lattices — the operators the reference
 *                       itself ships, sample3_dynamic.cpp:17-22): one workgroup per row block keeps the y slice AND, tile by
 *                       tile, the x slice in LDS and streams 12 B per nonzero (fp64) — no global gather, no product buffer.
 *                       Built only when the row blocks touch few column tiles (a random matrix is not eligible).  Sharded
 *                       contexts launch it twice per product: the row blocks whose tiles all lie inside the rank's own columns
 *                       run under the all-gather, the others when the vector has arrived (ll_op_tiled_layout reports the
 *                       split); the ranks' maxima of |x| travel in an 8-byte all-gather in front of the vector's, so the
 *                       shards stitch to the bits of the single-GPU product.  By default it sums in fixed point like LL_SPMV_PB's default form (same integers: the result does
 *                       not depend on the tiling), i.e. the NORM-wise accuracy class below; with LL_ACCURACY_COMPONENTWISE the same
 *                       image is summed in floating point, the waves of a workgroup adding in turn (a fixed order).
 * ll_op_create_csr_{d,z} and _csr_dev_ build the images, time them on the device with the actual matrix (sharded
 * contexts: summed over the ranks, so every rank takes the same decision), keep the fastest and RELEASE the others
 * (for BASELINE config 3 that returns 1.8 GB of CSR arrays).  Environment: LL_SPMV_KERNEL=csr|pb|tiled skips the timing,
 * LL_SPMV_KEEP_BOTH=1 keeps every image so that ll_op_select_spmv can switch later (A/B timing, tests).
 *
 * ACCURACY of y = A x (this is what replaces the user's fp64 mv_mul, LL:243 / EX:108):
 *   LL_SPMV_CSR_STREAM, and LL_SPMV_PB / LL_SPMV_TILED with floating-point sums (LL_ACCURACY_COMPONENTWISE; LL_PB_PHASE2=ordered|atomic):
 *     |y_i - (A x)_i| <= ~nnz_i * eps * sum_j |a_ij| |x_j|             (COMPONENT-wise, like a plain fp64 row loop).
 *   LL_SPMV_PB and LL_SPMV_TILED in their default form (LL_PB_PHASE2=fixed) round every product to a per-row fixed-point grid and add
 *   64-bit integers (order-independent: same bits for every launch, block geometry and partition of the matrix):
 *     |y_i - (A x)_i| <= eps * sum_j |a_ij| |x_j|  +  nnz_i * 2^-60 * (sum_j |a_ij|) * max_k |x_k|   (NORM-wise)
 *   where max_k runs over the WHOLE input vector.  For vectors whose entries are of comparable size (Lanczos vectors of
 *   extended states, random vectors) the second term is 2^7 or more times below the first.  For a vector with a huge dynamic
 *   range (x = e_0, a strongly localised state, an entry of 1e20 next to O(1) entries) rows whose terms are all far
 *   below (sum_j |a_ij|) max|x| lose RELATIVE accuracy: their absolute error stays below nnz_i 2^-60 ||A||_inf ||x||_inf, which
 *   is what the Lanczos recurrence and the Exponentiator need (tests/test_gpu_round3.py asserts both bounds and whole
 *   runs from x = e_0 against the real reference), but it is not the component-wise accuracy of an fp64 row loop.
 *   Callers who need that on such vectors ask for LL_ACCURACY_COMPONENTWISE (ll_csr_options.accuracy / ll_op_set_accuracy below;
 *   3-5 % slower); the environment's LL_PB_PHASE2=ordered or LL_SPMV_KERNEL=csr does the same for every operator of a context.
 *   Rows that meet an Inf / NaN are reported as NaN.  float / complex float storage: the product a_ij x_j is rounded to
 *   the storage type once (exactly what a float multiply gives) before it is summed in fixed point / double. */
enum { LL_SPMV_CSR_STREAM = 0, LL_SPMV_PB = 1, LL_SPMV_TILED = 2 };
/* The ACCURACY CLASS above as a per-operator choice of the caller (who knows whether the vectors are localised), not of
 * the environment:
 *   LL_ACCURACY_DEFAULT        what the context's environment says (LL_PB_PHASE2; norm-wise when unset)
 *   LL_ACCURACY_NORMWISE       the order-independent fixed-point sums where the PB kernel is selected (fastest, bit-identical
 *                              for every partition of the matrix); CSR-stream operators are component-wise anyway
 *   LL_ACCURACY_COMPONENTWISE  floating-point sums in a fixed order in every kernel: the accuracy of the user's own fp64
 *                              row loop (LL:120-126), still bit-reproducible run to run, 3-5 % slower on PB operators
 * ll_csr_options carries it (and the kernel choice, and where the arrays live) into ll_op_create_csr_opt_*; an existing
 * operator is moved between the classes by ll_op_set_accuracy (same image, another phase-2 kernel: no rebuild);
 * ll_op_accuracy reports the class of the kernel that is selected now (never LL_ACCURACY_DEFAULT). */
enum { LL_ACCURACY_DEFAULT = 0, LL_ACCURACY_NORMWISE = 1, LL_ACCURACY_COMPONENTWISE = 2 };
typedef struct ll_csr_options {
  int32_t accuracy;          /* LL_ACCURACY_* */
  int32_t kernel;            /* -1: the environment decides (timing of the candidates unless LL_SPMV_KERNEL); LL_SPMV_CSR_STREAM / LL_SPMV_PB /
                              * LL_SPMV_TILED (an error when the matrix is not eligible or has no entries: never a silent fallback) */
  int32_t arrays_on_device;  /* 0: row_ptr / col / val are host arrays (copied); 1: device arrays as in ll_op_create_csr_dev_* */
  int32_t reserved[5];       /* zero */
} ll_csr_options;
int ll_csr_options_default(ll_csr_options* opt); /* {LL_ACCURACY_DEFAULT, -1, 0, zeros}: then identical to ll_op_create_csr_* */
int ll_op_create_csr_opt_d(ll_context* ctx, int64_t n_rows, int64_t n_cols, int64_t row_begin, const int64_t* row_ptr,
                           const int32_t* col, const double* val, const ll_csr_options* opt, ll_operator** out);
int ll_op_create_csr_opt_z(ll_context* ctx, int64_t n_rows, int64_t n_cols, int64_t row_begin, const int64_t* row_ptr,
                           const int32_t* col, const void* val, const ll_csr_options* opt, ll_operator** out);
int ll_op_create_csr_opt_s(ll_context* ctx, int64_t n_rows, int64_t n_cols, int64_t row_begin, const int64_t* row_ptr,
                           const int32_t* col, const float* val, const ll_csr_options* opt, ll_operator** out);
int ll_op_create_csr_opt_c(ll_context* ctx, int64_t n_rows, int64_t n_cols, int64_t row_begin, const int64_t* row_ptr,
                           const int32_t* col, const void* val, const ll_csr_options* opt, ll_operator** out);
int ll_op_set_accuracy(ll_operator* op, int accuracy);        /* LL_ACCURACY_NORMWISE | LL_ACCURACY_COMPONENTWISE */
int ll_op_accuracy(const ll_operator* op, int* accuracy_out);
int ll_op_select_spmv(ll_operator* op, int kind);
int ll_op_selected_spmv(const ll_operator* op, int* kind_out);
/* Milliseconds the creation-time timing measured per kernel on this rank (-1: that kernel was not timed). */
int ll_op_autotune_ms(const ll_operator* op, double* csr_stream_ms, double* pb_ms);
int ll_op_autotune_ms_of(const ll_operator* op, int kind /* LL_SPMV_* */, double* ms);
/* The tiled image's row blocks (0: no tiled image) and how many of them need no column of another rank (all of them on one GPU). */
int ll_op_tiled_layout(const ll_operator* op, int* row_blocks, int* own_column_row_blocks);
int ll_op_destroy(ll_operator* op);
/* Global dimension n, local rows, nnz held locally (0 for callbacks). */
int ll_op_info(const ll_operator* op, int64_t* n, int64_t* n_local, int64_t* nnz_local);

/* ------------------------------------------------------------------ hot-path primitives
 * Exposed so that every kernel can be parity-checked on its own.  Vectors are LOCAL shards (n_local elements);
 * scalar results are already all-reduced over ranks when a communicator is attached. */

/* a1+a2+a3: y = A x + offset*x ; if dot_host != NULL also returns Re<x, y> (LL:243-248, EX:108-110).
 * x_dev is the LOCAL shard (the library all-gathers it when sharded). */
int ll_spmv_d(ll_context* ctx, ll_operator* op, const double* x_dev, double* y_dev, double offset,
              double* dot_host);
int ll_spmv_z(ll_context* ctx, ll_operator* op, const void* x_dev, void* y_dev, double offset, double* dot_host);

/* a3: <a,b> = sum conj(a_i) b_i (LA:29-51; conjugate-linear in the FIRST argument).  out_host: 1 double (_d) or
 * re,im (_z). */
int ll_dot_d(ll_context* ctx, int64_t n_local, const double* a_dev, const double* b_dev, double* out_host);
int ll_dot_z(ll_context* ctx, int64_t n_local, const void* a_dev, const void* b_dev, double* out_host_reim);
/* a7: sqrt(Re<v,v>), unscaled (LA:56-60). */
int ll_nrm2_d(ll_context* ctx, int64_t n_local, const double* v_dev, double* out_host);
int ll_nrm2_z(ll_context* ctx, int64_t n_local, const void* v_dev, double* out_host);
/* a8: v *= a (LA:65-72) with a real factor; normalize = nrm2 + scal(1/nrm2) (LA:77-80). */
int ll_scal_d(ll_context* ctx, int64_t n_local, double a, double* v_dev);
int ll_scal_z(ll_context* ctx, int64_t n_local, double a, void* v_dev);
int ll_normalize_d(ll_context* ctx, int64_t n_local, double* v_dev, double* norm_host /* nullable */);
int ll_normalize_z(ll_context* ctx, int64_t n_local, void* v_dev, double* norm_host /* nullable */);
/* a4: w = w - beta*u_prev - alpha*u_cur (LL:251-257, EX:112-118); u_prev_dev may be NULL (k == 1). */
int ll_three_term_d(ll_context* ctx, int64_t n_local, double* w_dev, const double* u_prev_dev,
                    const double* u_cur_dev, double beta, double alpha);
int ll_three_term_z(ll_context* ctx, int64_t n_local, void* w_dev, const void* u_prev_dev, const void* u_cur_dev,
                    double beta, double alpha);
/* a5+a6+a7: orthogonalise w against nb orthonormal vectors stored row-major with leading dimension ld
 * (vector j at basis_dev + j*ld elements) and return ||w|| afterwards (LA:132-144 at LL:259-262, EX:121,145).
 * mode: LL_ORTH_CGS_DGKS (default, block classical Gram-Schmidt + a second pass when the norm drops below
 * 1/sqrt(2)), LL_ORTH_CGS2 (always two passes), LL_ORTH_MGS (sequential dot->axpy per vector: the reference's
 * operation order, 2*nb launches). h_host (nullable) receives the nb projection coefficients summed over passes
 * (re,im pairs for _z). */
enum { LL_ORTH_CGS_DGKS = 0, LL_ORTH_CGS2 = 1, LL_ORTH_MGS = 2 };
int ll_orth_block_d(ll_context* ctx, int64_t n_local, int64_t nb, const double* basis_dev, int64_t ld,
                    double* w_dev, int mode, double* norm_host, double* h_host);
int ll_orth_block_z(ll_context* ctx, int64_t n_local, int64_t nb, const void* basis_dev, int64_t ld, void* w_dev,
                    int mode, double* norm_host, double* h_host);
/* a9+a10: out_r = sum_{k=m-1..0} coeff[r*m + k] * basis_k for r < nout in ONE pass over the basis
 * (LL:51-57: Ritz vectors, real coefficients; EX:166-170: exp(aA)v, coefficients of type T).
 * coeff_host: nout*m values of type T (re,im pairs for _z).  out_dev: nout vectors, leading dimension ld_out. */
int ll_gemv_basis_d(ll_context* ctx, int64_t n_local, int64_t m, const double* basis_dev, int64_t ld, int64_t nout,
                    const double* coeff_host, double* out_dev, int64_t ld_out);
int ll_gemv_basis_z(ll_context* ctx, int64_t n_local, int64_t m, const void* basis_dev, int64_t ld, int64_t nout,
                    const double* coeff_host_reim, void* out_dev, int64_t ld_out);

/* a11 (host, TRI:290-361): all eigenvalues (ascending) and optionally eigenvectors (q_host row-major m*m,
 * row j = eigenvector j, nullable) of the symmetric tridiagonal T(alpha[0..m), beta[0..m-1)).
 * Returns the reference's "unconverged" count through unconverged_out (nullable). */
int ll_tridiag_eig(int64_t m, const double* alpha_host, const double* beta_host, double* ev_host, double* q_host,
                   int64_t* unconverged_out);
/* k-th smallest eigenvalue (k = 0 .. m-1) by Sturm bisection (TRI:22-88, find_mth_eigenvalue); what
 * LL_TRIDIAG_BISECT / LL_TRIDIAG_AUTO use for the per-iteration stop test. */
int ll_tridiag_bisect(int64_t m, const double* alpha_host, const double* beta_host, int64_t k, double* out);
/* The same for nk roots ks[0..nk) in one call: bit-identical to nk calls of ll_tridiag_bisect, but the Sturm recurrences
 * of the roots run interleaved (the recurrence is bound by the latency of its division), ~5x faster.  This is what the
 * per-iteration stop test of LL_TRIDIAG_AUTO uses for the nroot tracked Ritz values. */
int ll_tridiag_bisect_multi(int64_t m, const double* alpha_host, const double* beta_host, int64_t nk,
                            const int64_t* ks, double* out_nk);
/* Unit eigenvectors for nw given eigenvalues by inverse iteration, O(m) each (out_host: nw rows of m entries); what
 * LL_TRIDIAG_AUTO uses for the final Ritz step when m > 256 instead of accumulating all m vectors (LL:44, O(m^3)). */
int ll_tridiag_eigvecs(int64_t m, const double* alpha_host, const double* beta_host, int64_t nw,
                       const double* lambdas_host, double* out_host);

/* ------------------------------------------------------------------ whole-loop entry points */

/* Start-vector hook (LL:133 init_vector): fill the LOCAL shard vec_host[0..n_local) holding global rows
 * [row_begin, row_begin+n_local) (re,im pairs for _z).  Called once per restart pass (LL:231-232).
 * NULL = the reference's default: nondeterministic uniform [-1,1] (LL:70-104). */
typedef void (*ll_init_vector_fn)(void* vec_host, int64_t n_local, int64_t row_begin, void* user);

/* LL_TRIDIAG_QR: the reference's implicit-shift QR of all k Ritz values every iteration (O(k^2), TRI:290-361).
 * LL_TRIDIAG_BISECT: Sturm bisection of the nroot wanted values only (O(k), TRI:22-88).
 * LL_TRIDIAG_AUTO: QR up to k = 64, bisection beyond, and the reference's QR arithmetic decides whenever a root's
 *   change comes within 4*eps of the stop threshold: same iteration counts and eigenvalues as LL_TRIDIAG_QR; few Ritz
 *   vectors by inverse iteration once k > 256.  Recommended for runs of more than a few hundred iterations. */
enum { LL_TRIDIAG_QR = 0, LL_TRIDIAG_BISECT = 1, LL_TRIDIAG_AUTO = 2 };

typedef struct ll_lanczos_params {
  /* --- the reference's public fields, same meaning and defaults (LL:126-181) --- */
  int64_t matrix_size;            /* LL:136  global n */
  int64_t max_iteration;          /* LL:138,206  default n */
  double eps;                     /* LL:150  default 1e3 * DBL_EPSILON */
  int32_t find_maximum;           /* LL:153 */
  int32_t reserved0;
  int64_t num_eigs;               /* LL:156  default 1 */
  double eigenvalue_offset;       /* LL:165  default 0 */
  int64_t num_eigs_per_iteration; /* LL:173  default 5 */
  int64_t initial_vector_size;    /* LL:181  default 200 — initial capacity (vectors) of the device basis slab */
  /* --- additions (0 = reference-faithful behaviour) --- */
  int32_t tridiag_mode;           /* LL_TRIDIAG_*: how the per-iteration Ritz values (LL:267-268) are obtained */
  int32_t orth_mode;              /* LL_ORTH_* */
  ll_init_vector_fn init_vector;  /* LL:133 */
  void* init_user;
  const void* init_vector_dev;    /* non-NULL: the start vector (n_local elements) is already in DEVICE memory; used
                                   * instead of the hook in every pass, copied, never modified */
} ll_lanczos_params;

/* Fill *p with the reference defaults for an n x n problem (LL:200-208); tridiag_mode = LL_TRIDIAG_AUTO (same stop
 * decisions and values as the reference's per-iteration QR).  eps is the DOUBLE default 1e3 * DBL_EPSILON (LL:150 with
 * real_t<T> = double); the reference scales it with the epsilon of real_t<T>, so the _s / _c entry points replace
 * exactly that value by 1e3 * FLT_EPSILON (likewise ll_expo_params_default's 1e2 * DBL_EPSILON, EX:58) — a float run
 * left at the defaults converges like the reference's float instantiation instead of iterating to max_iteration. */
int ll_lanczos_params_default(ll_lanczos_params* p, int64_t n, int find_maximum, int64_t num_eigs);

typedef struct ll_run_stats {
  int64_t n_passes;         /* restart passes = getIterationCounts().size() (LL:412-414) */
  int64_t total_iterations; /* sum of the iteration counts */
  double seconds_total;     /* wall time of the call */
  double seconds_host_tridiag; /* host time spent in the tridiagonal eigen-solver (a11/a12) */
  double seconds_spmv;      /* device time inside the operator (HIP events; 0 unless ll_ctx_set_profiling) */
  double seconds_orth;      /* device time in three-term + orthogonalisation + norm + scale */
  int64_t last_alpha_len;   /* entries written to alpha_out/beta_out (last pass) */
  double seconds_host_enqueue; /* host time spent enqueuing device work inside the loop */
  double seconds_host_wait;    /* host time spent waiting for the per-iteration scalars */
  double seconds_setup;        /* start vector, locked vectors (per pass) */
  double seconds_finish;       /* Ritz step: tridiagonal eigenvectors, GEMV over the basis, copy back */
  int64_t second_passes;       /* iterations whose Gram-Schmidt was repeated (DGKS test decided on the host) */
  double seconds_comm_gather;    /* device time of the all-gathers / halo exchanges (their own stream; 0 unless profiling) */
  double seconds_comm_allreduce; /* device time of the all-reduces (0 unless profiling) */
  int64_t lagged_iterations;     /* iterations that ran in the one-sweep (lagged) Gram-Schmidt form */
  int64_t pair_iterations;       /* ... of which: iterations that shared ONE sweep over the basis with their neighbour (two
                                  * iterations per sweep; taken from the reserved tail: same struct size) */
  int64_t pair_gate_trips;       /* times a pass left the two-iterations-per-sweep form because a coefficient of a raw vector
                                  * exceeded its gate (exhausted Krylov space / breakdown): 0 in ordinary runs */
  int64_t reserved[6];           /* zero; later statistics are taken from here, so the struct size stays what it is */
} ll_run_stats;
int ll_ctx_set_profiling(ll_context* ctx, int enabled);

/* LambdaLanczos<T>::run(eigenvalues, eigenvectors) (LL:330-366): restart loop + EigenPairManager semantics.
 *   eigvals_host   : num_eigs doubles, comparator order (descending for find_maximum, else ascending; LL:362-365)
 *   eigvecs_host   : num_eigs * n_local values of T, row-major, LOCAL shard of each eigenvector (nullable).
 *                    Despite the name the buffer may live in HOST or in DEVICE memory (ll_malloc / hipMalloc); a
 *                    device buffer receives the Ritz vectors without crossing PCIe (with p->init_vector_dev the whole
 *                    call then moves nothing n-sized over the bus).  The same holds for ll_lanczos_run_iteration_*.
 *   n_found        : number of pairs returned (<= num_eigs)
 *   iter_counts    : capacity iter_cap entries (getIterationCounts, LL:412); nullable
 *   alpha_out/beta_out : optional traces of the LAST pass (capacity max_iteration each); nullable
 */
int ll_lanczos_run_d(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, double* eigvals_host,
                     double* eigvecs_host, int64_t* n_found, int64_t* iter_counts, int64_t iter_cap,
                     double* alpha_out, double* beta_out, ll_run_stats* stats);
int ll_lanczos_run_z(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, double* eigvals_host,
                     void* eigvecs_host, int64_t* n_found, int64_t* iter_counts, int64_t iter_cap, double* alpha_out,
                     double* beta_out, ll_run_stats* stats);

/* LambdaLanczos<T>::run_iteration(eigvalues, eigvecs, nroot, orthogonalizeTo) (LL:216-322): ONE Lanczos pass that
 * tracks `nroot` Ritz pairs, with every Lanczos vector orthogonalised against the caller's n_orth vectors first
 * (LL:233,259).  No restart loop, no EigenPairManager: every computed pair comes back, in comparator order.
 *   orth_host      : n_orth * n_local values of T (vector j at orth_host + j*n_local; LOCAL shards); NULL if n_orth = 0
 *   eigvals_host   : capacity nroot doubles;  eigvecs_host: capacity nroot * n_local values of T (nullable)
 *   n_found        : pairs returned = min(nroot, iterations done) (LL:264,312)
 *   itern_out      : the method's return value, the Lanczos-iteration count
 * p->num_eigs and p->num_eigs_per_iteration are ignored; every other field acts as in ll_lanczos_run_*. */
int ll_lanczos_run_iteration_d(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, int64_t nroot,
                               int64_t n_orth, const double* orth_host, double* eigvals_host, double* eigvecs_host,
                               int64_t* n_found, int64_t* itern_out, double* alpha_out, double* beta_out,
                               ll_run_stats* stats);
int ll_lanczos_run_iteration_z(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, int64_t nroot,
                               int64_t n_orth, const void* orth_host, double* eigvals_host, void* eigvecs_host,
                               int64_t* n_found, int64_t* itern_out, double* alpha_out, double* beta_out,
                               ll_run_stats* stats);

typedef struct ll_expo_params {
  /* the reference's public fields (EX:41-71) */
  int64_t matrix_size;         /* EX:44 */
  int64_t max_iteration;       /* EX:46,81  default n */
  double eps;                  /* EX:58  default 1e2 * DBL_EPSILON */
  int32_t full_orthogonalize;  /* EX:63  default false */
  int32_t orth_mode;           /* LL_ORTH_* used when full_orthogonalize */
  int64_t initial_vector_size; /* EX:71  default 200 */
} ll_expo_params;
int ll_expo_params_default(ll_expo_params* p, int64_t n);

/* Exponentiator<T>::run(a, input, output) (EX:87-173): output = exp(a*A) input; returns the iteration count
 * through itern_out.  input/output are LOCAL shards (n_local values of T) in HOST or in DEVICE memory (decided per
 * pointer): a time-evolution loop psi <- exp(a*A) psi that keeps psi in device buffers never crosses PCIe.  input and
 * output may be the same buffer.  The same holds for ll_expo_taylor_run_*. */
int ll_expo_run_d(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a, const double* input_host,
                  double* output_host, int64_t* itern_out, ll_run_stats* stats);
int ll_expo_run_z(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a_re, double a_im,
                  const void* input_host, void* output_host, int64_t* itern_out, ll_run_stats* stats);
/* Exponentiator<T>::taylor_run (EX:175-210). */
int ll_expo_taylor_run_d(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a,
                         const double* input_host, double* output_host, int64_t* nterms_out);
int ll_expo_taylor_run_z(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a_re, double a_im,
                         const void* input_host, void* output_host, int64_t* nterms_out);

/* ------------------------------------------------------------------ float storage types (_s float, _c complex float)
 * Same entry points as above; see the comments on the _d / _z versions. */
int ll_op_create_csr_c(ll_context* ctx, int64_t n_rows_local, int64_t n_cols, int64_t row_begin,
                       const int64_t* row_ptr_host, const int32_t* col_host, const void* val_host ,
                       ll_operator** out);
int ll_op_create_csr_s(ll_context* ctx, int64_t n_rows_local, int64_t n_cols, int64_t row_begin,
                       const int64_t* row_ptr_host, const int32_t* col_host, const float* val_host ,
                       ll_operator** out);
int ll_op_create_csr_dev_c(ll_context* ctx, int64_t n_rows_local, int64_t n_cols, int64_t row_begin,
                           const int64_t* row_ptr_dev, const int32_t* col_dev, const void* val_dev,
                           ll_operator** out);
int ll_op_create_csr_dev_s(ll_context* ctx, int64_t n_rows_local, int64_t n_cols, int64_t row_begin,
                           const int64_t* row_ptr_dev, const int32_t* col_dev, const float* val_dev,
                           ll_operator** out);
int ll_op_create_coo_c(ll_context* ctx, int64_t n, int64_t nnz, const int32_t* rows_host, const int32_t* cols_host,
                       const void* vals_host, ll_operator** out);
int ll_op_create_coo_s(ll_context* ctx, int64_t n, int64_t nnz, const int32_t* rows_host, const int32_t* cols_host,
                       const float* vals_host, ll_operator** out);
int ll_op_create_host_c(ll_context* ctx, int64_t n, ll_host_mv_mul_z fn, void* user, ll_operator** out);
int ll_op_create_host_s(ll_context* ctx, int64_t n, ll_host_mv_mul_s fn, void* user, ll_operator** out);
int ll_op_create_device_c(ll_context* ctx, int64_t n, ll_dev_mv_mul fn, void* user, ll_operator** out);
int ll_op_create_device_s(ll_context* ctx, int64_t n, ll_dev_mv_mul fn, void* user, ll_operator** out);
int ll_op_create_dense_c(ll_context* ctx, int64_t n_rows_local, int64_t n_cols, int64_t row_begin,
                         const void* a_host, ll_operator** out);
int ll_op_create_dense_s(ll_context* ctx, int64_t n_rows_local, int64_t n_cols, int64_t row_begin,
                         const float* a_host, ll_operator** out);
int ll_op_create_stencil_c(ll_context* ctx, const ll_stencil_desc* desc, int64_t row_begin, int64_t n_local,
                           const double* onsite_host_local, ll_operator** out);
int ll_op_create_stencil_s(ll_context* ctx, const ll_stencil_desc* desc, int64_t row_begin, int64_t n_local,
                           const double* onsite_host_local, ll_operator** out);
int ll_spmv_c(ll_context* ctx, ll_operator* op, const void* x_dev, void* y_dev, double offset, double* dot_host);
int ll_spmv_s(ll_context* ctx, ll_operator* op, const float* x_dev, float* y_dev, double offset, double* dot_host);
int ll_dot_c(ll_context* ctx, int64_t n_local, const void* a_dev, const void* b_dev, double* out_host_reim);
int ll_dot_s(ll_context* ctx, int64_t n_local, const float* a_dev, const float* b_dev, double* out_host_reim);
int ll_nrm2_c(ll_context* ctx, int64_t n_local, const void* v_dev, double* out_host);
int ll_nrm2_s(ll_context* ctx, int64_t n_local, const float* v_dev, double* out_host);
int ll_scal_c(ll_context* ctx, int64_t n_local, double a, void* v_dev);
int ll_scal_s(ll_context* ctx, int64_t n_local, double a, float* v_dev);
int ll_normalize_c(ll_context* ctx, int64_t n_local, void* v_dev, double* norm_host );
int ll_normalize_s(ll_context* ctx, int64_t n_local, float* v_dev, double* norm_host );
int ll_three_term_c(ll_context* ctx, int64_t n_local, void* w_dev, const void* u_prev_dev, const void* u_cur_dev,
                    double beta, double alpha);
int ll_three_term_s(ll_context* ctx, int64_t n_local, float* w_dev, const float* u_prev_dev, const float* u_cur_dev,
                    double beta, double alpha);
int ll_orth_block_c(ll_context* ctx, int64_t n_local, int64_t nb, const void* basis_dev, int64_t ld, void* w_dev,
                    int mode, double* norm_host, double* h_host);
int ll_orth_block_s(ll_context* ctx, int64_t n_local, int64_t nb, const float* basis_dev, int64_t ld, float* w_dev,
                    int mode, double* norm_host, double* h_host);
int ll_gemv_basis_c(ll_context* ctx, int64_t n_local, int64_t m, const void* basis_dev, int64_t ld, int64_t nout,
                    const double* coeff_host_reim, void* out_dev, int64_t ld_out);
int ll_gemv_basis_s(ll_context* ctx, int64_t n_local, int64_t m, const float* basis_dev, int64_t ld, int64_t nout,
                    const double* coeff_host_reim, float* out_dev, int64_t ld_out);
int ll_lanczos_run_c(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, double* eigvals_host,
                     void* eigvecs_host, int64_t* n_found, int64_t* iter_counts, int64_t iter_cap, double* alpha_out,
                     double* beta_out, ll_run_stats* stats);
int ll_lanczos_run_s(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, double* eigvals_host,
                     float* eigvecs_host, int64_t* n_found, int64_t* iter_counts, int64_t iter_cap, double* alpha_out,
                     double* beta_out, ll_run_stats* stats);
int ll_lanczos_run_iteration_c(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, int64_t nroot,
                               int64_t n_orth, const void* orth_host, double* eigvals_host, void* eigvecs_host,
                               int64_t* n_found, int64_t* itern_out, double* alpha_out, double* beta_out,
                               ll_run_stats* stats);
int ll_lanczos_run_iteration_s(ll_context* ctx, ll_operator* op, const ll_lanczos_params* p, int64_t nroot,
                               int64_t n_orth, const float* orth_host, double* eigvals_host, float* eigvecs_host,
                               int64_t* n_found, int64_t* itern_out, double* alpha_out, double* beta_out,
                               ll_run_stats* stats);
int ll_expo_run_c(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a_re, double a_im,
                  const void* input_host, void* output_host, int64_t* itern_out, ll_run_stats* stats);
int ll_expo_run_s(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a, const float* input_host,
                  float* output_host, int64_t* itern_out, ll_run_stats* stats);
int ll_expo_taylor_run_c(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a_re, double a_im,
                         const void* input_host, void* output_host, int64_t* nterms_out);
int ll_expo_taylor_run_s(ll_context* ctx, ll_operator* op, const ll_expo_params* p, double a,
                         const float* input_host, float* output_host, int64_t* nterms_out);

#ifdef __cplusplus
}
#endif
#endif /* LANCZOS_HIP_H_ */
