/*
 * lanczos_hip_transport.h — the exchange-step plug-in of the sharded hot path.
 *
 * The reference (mrcdr/lambda-lanczos) is single-process: its only "exchange" is the mv_mul call site
 * include/lambda_lanczos/lambda_lanczos.hpp:243 reading the whole vector.  With the rows partitioned over several
 * GPUs the library needs three collectives per iteration (lanczos_hip.h, "multi-GPU"): an all-gather of the current
 * Lanczos vector, all-reduces of a few doubles, and — for the matrix-free lattice operator — a halo exchange with
 * the two ring neighbours.  By default they go to RCCL over xGMI (ll_comm_init).  An application that already owns a
 * transport (MPI with GPU-aware buffers, its own RCCL communicator, a test harness) can attach it instead:
 *
 *     ll_transport t = { self, my_all_gather, my_all_reduce, my_halo, my_destroy };
 *     ll_comm_attach(ctx, &t, rank, n_ranks);
 *
 * or name a shared object in LL_COMM_PLUGIN that exports
 *
 *     int ll_transport_unique_id(void* id_out_128);
 *     int ll_transport_open(const void* id_128, int rank, int n_ranks, int device, ll_transport* out);
 *
 * in which case ll_comm_unique_id / ll_comm_init are served by it (this is how the repository's multi-rank tests
 * run several ranks on ONE GPU, where RCCL refuses duplicate devices: tests/transport/shm_transport.cpp).
 *
 * Semantics every implementation must provide:
 *   - all pointers are DEVICE pointers on the context's device; `hip_stream` is a hipStream_t;
 *   - every call is STREAM-ORDERED: it may return before the data has moved, but the exchange must observe all work
 *     enqueued on `hip_stream` before the call, and work enqueued on `hip_stream` after the call must observe the
 *     result (a blocking implementation trivially qualifies);
 *   - the library calls the collectives in the same order on every rank, possibly on two different streams of the
 *     same context (the all-gather runs on a communication stream so that own-column work overlaps it);
 *   - all_reduce_sum_f64 must produce bit-identical results on all ranks (the replicated host decisions rely on it);
 *   - return 0 on success, anything else aborts the calling entry point with LL_ERR_RCCL.
 */
#ifndef LANCZOS_HIP_TRANSPORT_H_
#define LANCZOS_HIP_TRANSPORT_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

struct ll_context;

typedef struct ll_transport {
  void* self;
  /* every rank contributes bytes_per_rank bytes from send_dev; recv_dev receives n_ranks * bytes_per_rank bytes,
   * rank r's contribution at offset r * bytes_per_rank (ncclAllGather semantics). */
  int (*all_gather)(void* self, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* hip_stream);
  /* in-place sum over ranks of `count` doubles. */
  int (*all_reduce_sum_f64)(void* self, double* buf_dev, size_t count, void* hip_stream);
  /* ring halo exchange: `bytes` from send_prev_dev go to rank `prev` (they arrive in ITS recv_next_dev) and `bytes`
   * from send_next_dev go to rank `next` (its recv_prev_dev); prev / next = -1: no such neighbour.  prev == next and
   * prev == next == own rank are legal. */
  int (*halo_exchange)(void* self, const void* send_prev_dev, void* recv_prev_dev, int prev, const void* send_next_dev,
                       void* recv_next_dev, int next, size_t bytes, void* hip_stream);
  /* called once from ll_ctx_destroy (nullable). */
  void (*destroy)(void* self);
} ll_transport;

/* Attach an application-provided transport instead of RCCL.  Collective over all ranks (it runs the same rank
 * self-check as ll_comm_init).  The table is copied; `self` stays owned by the transport (released by destroy). */
int ll_comm_attach(struct ll_context* ctx, const ll_transport* transport, int rank, int n_ranks);

#ifdef __cplusplus
}
#endif

#endif /* LANCZOS_HIP_TRANSPORT_H_ */
