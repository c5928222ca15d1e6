// Propagation-blocked SpMV for gfx950 (a1/a2/a3: mv_mul LL:243 / EX:108, offset update LL:244-246, alpha dot LL:248)
// and the device-side construction of its matrix image.
//
// Measured on MI355X (profiles/r01_gather_probe.txt): a gather that misses the CU's 32 KiB L1 moves a whole 128-byte
// line for 8 useful bytes and the chip sustains only 57 G such gathers/s from an 80 MB x (217 G/s from an L2-resident
// table).  For a matrix without column locality (BASELINE config 3: 1.5e8 gathers) that caps ANY gather-based CSR
// kernel at >= 0.69 ms per SpMV (measured: 2.7 ms CSR-stream) while the matrix itself streams in 0.35 ms.
//
// These kernels therefore never gather from global memory.  The same matrix is stored in two sweeps' order (built
// once at upload, on the device) and one SpMV is two fully coalesced streaming kernels with LDS-resident slices:
//   phase 1 (one workgroup per COLUMN block): the x slice of the block is loaded into LDS; the block's entries
//            (value, 16-bit local column) stream in, ordered by destination row block; product = value * x_lds[col]
//            is written to the product buffer P at its position in row-block order (contiguous runs of one segment
//            = one (column block, row block) pair);
//   phase 2 (one workgroup per ROW block): the y slice lives in LDS; the row block's range of P and the 16-bit
//            local row indices stream in (perfectly sequential) and are added into the slice with ds_add_f64; the
//            epilogue adds offset*x_i (a2), writes y once and accumulates Re(conj(x_i) y_i) (a3).
// HBM traffic is 2*sizeof(T) + sizeof(T) + 4 bytes per nonzero (28 B for fp64 against 12 B for CSR) but every byte
// is streamed at full line efficiency and every x/y element is touched in LDS.
//
// Determinism: the LDS adds of phase 2 are issued WAVE BY WAVE in a fixed order (the 16 waves of the workgroup take
// turns inside every trip, a barrier between turns), so every y_i is summed in the same order on every launch: the
// kernel is bit-reproducible like every other reduction of the library (LL_PB_PHASE2=atomic restores the arrival-order
// adds for A/B timing).  The next trip's loads are in flight while the turns run, so the turns cost no bandwidth.
//
// Column blocks carry their own x-slice offset, so one image serves the single-GPU case (x in place), the sharded
// case (own columns from the local shard, the other ranks' columns from the gathered buffer, launched separately so
// that own-column work overlaps the all-gather) and a gather that arrives in several chunks.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <string>

#include "dev_helpers.hpp"
#include "fixed_round.hpp"
#include "ll_internal.hpp"

namespace ll {

constexpr int kPbThreads = 1024;
constexpr int kPbWaves = kPbThreads / 64;

__device__ __forceinline__ void lds_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// Entries are handled in QUADS: every segment is padded to a multiple of 16 entries (zero value, local index 0), so
// a lane always moves four consecutive entries with 16-byte accesses (2 x dwordx4 of values / products, one dwordx2
// of four 16-bit indices), all four share one segment, and every run of products starts and ends on a 128-byte line.
template <typename T> struct quad {
  T e[4];
};
template <typename T> __device__ __forceinline__ quad<T> load_quad(const T* __restrict__ p) {
  constexpr int NCH = (int)(4 * sizeof(T) / 16);  // 16-byte pieces of four entries (float: 1, double / cf: 2, zc: 4)
  const uint4* src = reinterpret_cast<const uint4*>(p);
  uint4 c[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) c[i] = src[i];
  quad<T> q;
  __builtin_memcpy(&q, c, sizeof(q));
  return q;
}
template <typename T> __device__ __forceinline__ void store_quad(T* __restrict__ p, const quad<T>& q) {
  constexpr int NCH = (int)(4 * sizeof(T) / 16);
  uint4 c[NCH];
  __builtin_memcpy(c, &q, sizeof(q));
  uint4* dst = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < NCH; ++i) dst[i] = c[i];
}

// ================================================================= software pipelines of the two phases
// One workgroup owns a CU (its LDS slice is > 100 KB), so nothing but the workgroup's own 16 waves hides memory latency:
// every lane keeps D trips of loads in flight in a STATIC ring of register slots — the trip loop is unrolled D times and
// every slot index is a compile-time constant, so a trip is consumed while the D - 1 requested after it are still in
// flight and the compiler can wait for exactly the oldest one (s_waitcnt vmcnt(N)).  (Rounds 1-2 rotated the slots by
// register copies at the end of every trip; a copy of slot d+1 needs ITS data, so every trip ended with a full drain of
// the memory pipeline and the stream and the LDS work ran one after the other.)
// Every vector-memory instruction of the trip loops is UNCONDITIONAL and sits in uniform straight-line code: a lane
// beyond the end of its range re-reads the range's last quad (a cache hit) and stores to a dump quad behind the image,
// only the LDS work is predicated.  With loads under divergent branches the number of outstanding loads is unknown at
// compile time and every wait degenerates to vmcnt(0) again.
// The quads of a workgroup's range are dealt out in WAVE CHUNKS (64 lanes x U quads) from a counter in LDS instead of
// by a fixed stride: the hardware favours a workgroup's oldest waves, so with a fixed partition wave 0 is done long
// before wave 15 (measured: 94 us against 160 us in phase 1) and the tail of every workgroup runs with a fraction of its
// loads in flight.  The first D - 1 chunks of every wave are fixed (they are requested before the counter is live).
template <int U, int W = kPbWaves> struct WaveChunks {  // W: waves of the workgroup
  static constexpr int kQuads = 64 * U;  // quads per chunk
  unsigned* ctr;                         // LDS: chunks handed out so far (starts at (D - 1) * W)
  long long g0;
  __device__ __forceinline__ long long fixed(int d) const { return g0 + (long long)((threadIdx.x >> 6) + d * W) * kQuads; }
  __device__ __forceinline__ long long grab() const {
    unsigned c = 0;
    if ((threadIdx.x & 63) == 0) c = atomicAdd(ctr, 1u);
    c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
    return g0 + (long long)c * kQuads;
  }
};
template <typename T, int D, int TH = kPbThreads> struct ColStream {  // phase 1: (4 values, 4 local columns) per lane per trip; TH lanes
  quad<T> v[D];
  ushort4 c[D];
  long long base[D];  // first quad of the chunk held in each slot
  const T* val;
  const ushort4* col;
  long long g1, glast;  // end of the range; last valid quad (>= its first quad; the image is padded behind its end)
  WaveChunks<1, TH / 64> chunks;
  __device__ __forceinline__ void issue(int slot, long long cbase) {
    base[slot] = cbase;
    const long long g = cbase + (threadIdx.x & 63);
    const long long gc = g < glast ? g : glast;
    v[slot] = load_quad<T>(val + 4 * gc);
    c[slot] = col[gc];
  }
  __device__ __forceinline__ void prologue() {
#pragma unroll
    for (int d = 0; d < D - 1; ++d) issue(d, chunks.fixed(d));
  }
  // consume(values, columns, quad index of this lane) for every chunk this wave gets; the counter must be live.
  // (The D phases are spelled out — slot indices must be compile-time constants, and a `return` inside a `#pragma
  // unroll` loop keeps the compiler from unrolling it.)
  template <int PH, typename F> __device__ __forceinline__ bool step(F&& consume) {
    const long long cur = base[PH];
    if (cur >= g1) return false;  // wave-uniform; chunks are handed out in ascending order
    issue((PH + D - 1) % D, chunks.grab());
    consume(v[PH], c[PH], cur + (threadIdx.x & 63));
    return true;
  }
  // (Measured and rejected in round 5: TWO rounds of the ring per loop iteration.  The waitcnt pass merges the pending loads of
  // the loop's two entries conservatively at its header, which costs a full drain in the first trip behind it — once per D
  // trips; with two rounds per iteration it is once per 2 D, yet config 3's SpMV came out 1 % slower on the same box,
  // 0.911 against 0.899 ms, profiles/r05_pb_ring_two_rounds_ab.txt: the other 15 waves of the workgroup cover the drain.)
  template <typename F> __device__ __forceinline__ void run(F&& consume) {
    static_assert(D >= 2 && D <= 4, "pipeline depth");
    for (;;) {
      if (!step<0>(consume)) return;
      if (!step<1>(consume)) return;
      if constexpr (D > 2) {
        if (!step<2>(consume)) return;
      }
      if constexpr (D > 3) {
        if (!step<3>(consume)) return;
      }
    }
  }
};
// phase 2: U x (4 products, 4 local rows) per lane per trip.  DYN: wave chunks as above; otherwise the fixed stride of
// the whole workgroup (the wave-ordered form needs every wave in every trip).
template <typename T, int U, int D, bool DYN> struct RowStream {
  quad<T> p[D][U];
  ushort4 r[D][U];
  long long base[D];
  const T* P;
  const ushort4* row;
  long long g1, glast;  // end of the range; last valid quad (>= the first quad: the image is padded behind its end)
  WaveChunks<U> chunks;
  static constexpr long long kTrip = (long long)U * kPbThreads;
  __device__ __forceinline__ long long quad_of(long long cbase, int u) const {
    return DYN ? cbase + (threadIdx.x & 63) + (long long)u * 64 : cbase + threadIdx.x + (long long)u * kPbThreads;
  }
  __device__ __forceinline__ void issue(int slot, long long cbase) {
    base[slot] = cbase;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long g = quad_of(cbase, u);
      const long long gc = g < glast ? g : glast;
      p[slot][u] = load_quad<T>(P + 4 * gc);
      r[slot][u] = row[gc];
    }
  }
  __device__ __forceinline__ void prologue() {
#pragma unroll
    for (int d = 0; d < D - 1; ++d) issue(d, DYN ? chunks.fixed(d) : chunks.g0 + d * kTrip);
  }
  // consume(products, rows, quad indices of this lane) for every trip; DYN: the counter must be live
  long long next_static;
  template <int PH, typename F> __device__ __forceinline__ bool step(F&& consume) {
    const long long cur = base[PH];
    if (cur >= g1) return false;  // uniform over the wave (DYN) / the workgroup
    long long nxt;
    if constexpr (DYN) nxt = chunks.grab();
    else {
      nxt = next_static;
      next_static += kTrip;
    }
    issue((PH + D - 1) % D, nxt);
    long long g[U];
#pragma unroll
    for (int u = 0; u < U; ++u) g[u] = quad_of(cur, u);
    consume(p[PH], r[PH], g);
    return true;
  }
  template <typename F> __device__ __forceinline__ void run(F&& consume) {
    static_assert(D >= 2 && D <= 4, "pipeline depth");
    next_static = chunks.g0 + (D - 1) * kTrip;
    for (;;) {
      if (!step<0>(consume)) return;
      if (!step<1>(consume)) return;
      if constexpr (D > 2) {
        if (!step<2>(consume)) return;
      }
      if constexpr (D > 3) {
        if (!step<3>(consume)) return;
      }
    }
  }
};

__device__ __forceinline__ double pow2(int k) {  // 2^k for |k| <= 1022
  return __longlong_as_double((long long)(1023 + k) << 52);
}

// ---- fixed-point sums (LL_PB_PHASE2=fixed, the default)
// Integer addition is associative, so if every product is first rounded to a fixed-point grid the LDS adds may arrive in
// ANY order — all waves add concurrently like in the arrival-order form — and the result is still the same bits on every
// launch, for every kernel geometry and every partition of the matrix.
//   grid of row i:  q_i = 2^(E_i - 62),  E_i = er_i + e_x + 1,  sum_j |a_ij| < 2^er_i (computed when the image is built),
//   max_k |x_k| < 2^e_x, so that the exact sum of the row, scaled by 1/q_i, fits a 64-bit integer with room to spare.
// Error per row: each product is rounded to q_i once (<= q_i / 2), the sum itself is exact:
//   |y_i - exact| <= nnz_i * 2^-63 * 2^E_i  <=  nnz_i * 2^-60 * (sum_j |a_ij|) max|x|      (lanczos_hip.h states this bound)
// — finer than the unit roundoff of a double-precision sum of terms of that size, and far below eps * ||A|| ||x||, the
// scale that matters to the Lanczos recurrence; NOT component-wise accurate for rows whose terms are all tiny against
// (sum_j |a_ij|) max|x| (LL_PB_PHASE2=ordered is).
// Phase 1 writes fl(a_ij x_j) and the maximum of |x| over its slice; phase 2 looks up the row's exponent per entry, scales,
// rounds and adds.  Rows that meet an Inf / NaN are reported as NaN.
// (Round 3 measured a PRE-SCALED form — image values a_ij 2^-er_i, max|x| known before phase 1, 64-bit integers in P,
// phase 2 a pure stream with one integer atomic per entry: the LDS work of phase 2 turned out to be free (removing the adds
// altogether changes nothing, profiles/r03_spmv_variants.txt runs C and G) while the float -> int64 conversion makes
// phase 1 borderline ALU-bound; no gain, removed again.)
constexpr long long kPbBadProduct = (long long)0x8000000000000000ull;  // "not a finite number below 2^63"
constexpr int kPbXPre = 16;      // rows per lane whose x_i the fixed-point phase 2 holds in registers (row blocks of <= 16 384 rows)
constexpr int kPbXInf = 20000;   // e_x when max|x| is not finite: every row is reported as NaN

// (fixed_round.hpp: rint() to a 64-bit integer in four full-rate additions instead of the six-instruction, mostly quarter-rate
// f64 -> i64 conversion sequence — the same integers, checked value by value on the host in tests/cpp/fixed_round_test.cpp)
__device__ __forceinline__ long long pb_to_fixed(double sc) {
  if (!(fabs(sc) < 9.0e18)) return kPbBadProduct;
  return fixed_round(sc);
}
__device__ __forceinline__ long long pb_to_fixed(double p, int k) {
  return pb_to_fixed(ldexp(p, k));  // v_ldexp_f64: one instruction, no range restrictions
}
// maximum over the workgroup of a per-lane value (result in every lane); scratch: kPbWaves doubles + 1
template <int W = kPbWaves> __device__ __forceinline__ double pb_block_max(double m, double* scratch) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = fmax(m, __shfl_down(m, d, 64));
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = scratch[0];
    for (int w = 1; w < W; ++w) t = fmax(t, scratch[w]);
    scratch[W] = t;
  }
  __syncthreads();
  return scratch[W];
}

// ================================================================= phase 1
// Workgroup b handles column block blk_first + b: the block's entries (4 values + 4 local columns per lane and trip)
// stream in, ordered by destination row block; product = value * x_lds[col] goes to the product buffer P at its position
// in row-block order.  The first D - 1 trips are requested before the x slice is staged.
// blockmax (nullable; fixed-point phase 2) receives the maximum of |x| over the slice.
template <typename T, int D, int TH>
__global__ __launch_bounds__(TH) void pb_phase1(int nrb, int blk_first, const int64_t* __restrict__ xoff,
                                                        const int32_t* __restrict__ ncols_tab,
                                                        const int64_t* __restrict__ seg_q,     // [ncb][nrb+1]
                                                        const int64_t* __restrict__ seg_dest,  // [ncb][nrb]
                                                        const T* __restrict__ val, const ushort4* __restrict__ col,
                                                        const T* __restrict__ xsrc, T* __restrict__ P, int cb_cols,
                                                        double* __restrict__ blockmax, long long p_dump,
                                                        const double* __restrict__ xnorm2) {
  extern __shared__ double lds[];
  // xnorm2 (nullable): the input is an UNNORMALISED vector w with ||w||^2 = *xnorm2 (lagged Gram-Schmidt, kernels.hip): the
  // slice is scaled by 1 / ||w|| while it is staged
  const double xs_fac = xnorm2 ? 1.0 / sqrt(*xnorm2) : 1.0;
  __shared__ double bm_red[TH / 64 + 1];
  __shared__ unsigned chunk_ctr;
  T* xs = reinterpret_cast<T*>(lds);                                               // [cb_cols]
  long long* qs = reinterpret_cast<long long*>(reinterpret_cast<char*>(lds) +
                                              (((size_t)cb_cols * sizeof(T) + 15) & ~(size_t)15));  // [nrb + 1]
  long long* db = qs + (nrb + 1);                                                  // [nrb]
  const int tid = threadIdx.x;
  const int c = blk_first + blockIdx.x;
  const int64_t* sq = seg_q + (size_t)c * (nrb + 1);
  const long long g0 = sq[0] >> 2, g1 = sq[nrb] >> 2;

  ColStream<T, D, TH> st;
  st.val = val;
  st.col = col;
  st.g1 = g1;
  st.glast = g1 > g0 ? g1 - 1 : g0;
  st.chunks.ctr = &chunk_ctr;
  st.chunks.g0 = g0;
  st.prologue();
  if (tid == 0) chunk_ctr = (D - 1) * (TH / 64);  // (published by the barriers below)
  {  // stage the x slice (16-byte loads when the slice is 16-byte aligned) and the segment tables.  All loads of a batch are
    // requested before the first LDS store (clamped addresses, no load under a divergent branch): a load -> store loop
    // per piece is a chain of memory latencies — 9-17 us per workgroup with nothing else running on the CU.
    const int ncols = ncols_tab[c];
    const T* src = xsrc + xoff[c];
    constexpr int V = (int)(16 / sizeof(T));
    if (V >= 1 && ncols >= V && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      const int nv = ncols / V;
      const uint4* s4 = reinterpret_cast<const uint4*>(src);
      uint4* d4 = reinterpret_cast<uint4*>(xs);
      constexpr int SB = 7;  // 7 x 1024 x 16 B = 112 KiB: every slice in one batch (256 lanes: 28 KiB, more than their slices)
      for (int i0 = 0; i0 < nv; i0 += SB * TH) {
        uint4 piece[SB];
#pragma unroll
        for (int b = 0; b < SB; ++b) {
          const int i = i0 + b * TH + tid;
          piece[b] = s4[i < nv ? i : nv - 1];
        }
#pragma unroll
        for (int b = 0; b < SB; ++b) {
          const int i = i0 + b * TH + tid;
          if (xnorm2) {
            T el[V];
            __builtin_memcpy(el, &piece[b], sizeof(uint4));
#pragma unroll
            for (int q = 0; q < V; ++q) el[q] = rmul(xs_fac, el[q]);
            __builtin_memcpy(&piece[b], el, sizeof(uint4));
          }
          if (i < nv) d4[i] = piece[b];
        }
      }
      for (int i = nv * V + tid; i < ncols; i += TH) xs[i] = rmul(xs_fac, src[i]);
    } else {
      for (int i = tid; i < ncols; i += TH) xs[i] = rmul(xs_fac, src[i]);
    }
    const int64_t* sd = seg_dest + (size_t)c * nrb;
    for (int i0 = 0; i0 <= nrb; i0 += 2 * TH) {
      long long tq[2], td[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int i = i0 + b * TH + tid;
        tq[b] = sq[i <= nrb ? i : nrb];
        td[b] = sd[i < nrb ? i : (nrb > 0 ? nrb - 1 : 0)];
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int i = i0 + b * TH + tid;
        if (i <= nrb) qs[i] = tq[b];
        if (i < nrb) db[i] = td[b];
      }
    }
  }
  __syncthreads();
  if (blockmax != nullptr) {  // the largest |x| of the slice (phase 2 needs max |x| over all columns)
    double m = 0.0;
    const int ncols = ncols_tab[c];
    for (int i = tid; i < ncols; i += TH) m = fmax(m, abs1(xs[i]));
    m = pb_block_max<TH / 64>(m, bm_red);
    if (tid == 0) blockmax[c] = m;  // NaN in the slice: fmax drops it; the products carry it into P and phase 2 reports it
  }
  int r = 0;
  T* const dump = P + p_dump;  // where lanes beyond the block's last quad store (a quad behind the image)
  auto consume = [&](const quad<T>& v, const ushort4& cl, long long g) {
    const long long qq = 4 * g;
    while (r + 1 < nrb && qq >= qs[r + 1]) ++r;
    const unsigned short cc[4] = {cl.x, cl.y, cl.z, cl.w};
    quad<T> pr;
#pragma unroll
    for (int e = 0; e < 4; ++e) pr.e[e] = mul(v.e[e], xs[cc[e]]);
    store_quad<T>(g < g1 ? P + db[r] + (qq - qs[r]) : dump, pr);
  };
  st.run(consume);
}

// ================================================================= phase 2
// One workgroup per ROW block (y slice in LDS): the block's range of P and of the 16-bit local rows streams in and is
// added into the slice; the epilogue adds offset * x_i (a2), writes y once and accumulates Re(conj(x_i) y_i) (a3).
template <typename T> __device__ __forceinline__ void lds_add_elem(double* lds, int rl, const T& v) {
  if constexpr (scalar_traits<T>::is_complex) {
    lds_add(&lds[2 * rl], (double)v.re);
    lds_add(&lds[2 * rl + 1], (double)v.im);
  } else {
    lds_add(&lds[rl], (double)v);
  }
}
// the shared epilogue: value(i, x_i) gives row i's sum (x_i: the row's own input element, for the diagonal term that the
// PB image keeps outside its streams); y = value + offset x, partial Re<x, y> per workgroup
template <typename T, typename F, typename G>
__device__ __forceinline__ void pb_phase2_epilogue(int rb, int64_t row0, int rows, const T* __restrict__ xl, T* __restrict__ y,
                                                   double offset, double* __restrict__ dot_partials, double* red,
                                                   const double* __restrict__ xnorm2, F&& value, G&& pre) {
  // value(i, x_i, pre(i)): pre(i) is what row i's value needs from GLOBAL memory besides x_i (its diagonal entry, its exponent).
  // All loads of a round — x_i and pre(i) of EU rows per lane, clamped addresses, no load under a divergent branch — are requested
  // before the first is used: requested inside the per-row `if (i < rows)` they form a chain of one memory latency per row, which
  // made this epilogue 12-17 us per workgroup with nothing else running on the CU.
  const double xs_fac = xnorm2 ? 1.0 / sqrt(*xnorm2) : 1.0;  // unnormalised input (see pb_phase1)
  const int tid = threadIdx.x;
  constexpr int EU = 8;  // rows per lane per round
  double dot_acc = 0.0;
  for (int i0 = tid; i0 < rows; i0 += EU * kPbThreads) {
    T xi[EU];
    decltype(pre(0)) pl[EU];
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int i = i0 + u * kPbThreads;
      const int ic = i < rows ? i : rows - 1;
      xi[u] = rmul(xs_fac, xl[row0 + ic]);
      pl[u] = pre(ic);
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int i = i0 + u * kPbThreads;
      if (i < rows) {
        const T yi = add(narrow<T>(value(i, xi[u], pl[u])), rmul(offset, xi[u]));
        y[row0 + i] = yi;
        dot_acc += re_cmul(xi[u], yi);
      }
    }
  }
  if (dot_partials) {
    const double v = wave_sum(dot_acc);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < kPbWaves; ++w) t += red[w];
      dot_partials[rb] = t;
    }
  }
}

// Floating-point sums.  ORDERED: the 16 waves add in turn (a barrier between turns) — a fixed order, so every y_i is the
// same bits on every launch, and component-wise accurate like the reference's fp64 mv_mul; otherwise arrival order
// (LL_PB_PHASE2=atomic: not reproducible, A/B timing reference).
template <typename T, bool ORDERED, int D>
__global__ __launch_bounds__(kPbThreads) void pb_phase2(int rb_rows, int64_t n_local,
                                                        const int64_t* __restrict__ rptr,  // [nrb + 1]
                                                        const ushort4* __restrict__ row, const T* __restrict__ P,
                                                        const T* __restrict__ xl, T* __restrict__ y, double offset,
                                                        double* __restrict__ dot_partials, const double* __restrict__ xnorm2,
                                                        const T* __restrict__ diag) {
  constexpr int R = scalar_traits<T>::reals;
  constexpr int U = 2;
  extern __shared__ double lds[];  // [rb_rows * R]
  __shared__ double red[kPbWaves];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int rb = blockIdx.x;
  const int64_t row0 = (int64_t)rb * rb_rows;
  const int rows = (int)min((int64_t)rb_rows, n_local - row0);
  const long long g0 = rptr[rb] >> 2, g1 = rptr[rb + 1] >> 2;
  __shared__ unsigned chunk_ctr;
  RowStream<T, U, D, !ORDERED> st;
  st.P = P;
  st.row = row;
  st.g1 = g1;
  st.glast = g1 > g0 ? g1 - 1 : g0;
  st.chunks.ctr = &chunk_ctr;
  st.chunks.g0 = g0;
  st.prologue();
  if (tid == 0) chunk_ctr = (D - 1) * kPbWaves;
  for (int i = tid; i < rb_rows * R; i += kPbThreads) lds[i] = 0.0;
  __syncthreads();
  st.run([&](const quad<T>(&p)[U], const ushort4(&r)[U], const long long(&g)[U]) {
    auto add_mine = [&]() {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (g[u] < g1) {
          lds_add_elem<T>(lds, r[u].x, p[u].e[0]);
          lds_add_elem<T>(lds, r[u].y, p[u].e[1]);
          lds_add_elem<T>(lds, r[u].z, p[u].e[2]);
          lds_add_elem<T>(lds, r[u].w, p[u].e[3]);
        }
      }
    };
    if constexpr (ORDERED) {
      for (int w = 0; w < kPbWaves; ++w) {
        if (wave == w) add_mine();
        __syncthreads();
      }
    } else {
      add_mine();
    }
  });
  __syncthreads();
  pb_phase2_epilogue<T>(
      rb, row0, rows, xl, y, offset, dot_partials, red, xnorm2,
      [&](int i, const T& xi, const T& dg) {
        acc_t<T> a;
        if constexpr (scalar_traits<T>::is_complex) a = zc{lds[2 * i], lds[2 * i + 1]};
        else a = lds[i];
        if (diag) a = add(a, to_acc(mul(dg, xi)));  // the diagonal entry, kept outside the streams (pb_diag_kernel)
        return a;
      },
      [&](int i) { return diag ? diag[row0 + i] : zero<T>(); });
}

// Fixed-point sums (see above): P holds fl(a_ij x_j) in the storage type.
template <typename T, int D>
__global__ __launch_bounds__(kPbThreads) void pb_phase2_fixed(int rb_rows, int64_t n_local, int ncb,
                                                              const int64_t* __restrict__ rptr,
                                                              const ushort4* __restrict__ row, const T* __restrict__ P,
                                                              const int16_t* __restrict__ rexp,
                                                              const double* __restrict__ blockmax,
                                                              const T* __restrict__ xl, T* __restrict__ y, double offset,
                                                              double* __restrict__ dot_partials,
                                                              const double* __restrict__ xnorm2, int xprefetch,
                                                              const T* __restrict__ diag) {
  constexpr int R = scalar_traits<T>::reals;
  constexpr int U = 2;
  extern __shared__ double lds_raw[];
  long long* acc = reinterpret_cast<long long*>(lds_raw);                   // [rb_rows * R]
  int16_t* ex = reinterpret_cast<int16_t*>(acc + (size_t)rb_rows * R);      // [rb_rows]: 62 - E_i, or kBadRow
  __shared__ double red[kPbWaves + 1];
  constexpr int kBadRow = 32767;
  const int tid = threadIdx.x;
  const int rb = blockIdx.x;
  const int64_t row0 = (int64_t)rb * rb_rows;
  const int rows = (int)min((int64_t)rb_rows, n_local - row0);
  const long long g0 = rptr[rb] >> 2, g1 = rptr[rb + 1] >> 2;
  __shared__ unsigned chunk_ctr;
  // The epilogue needs x_i of the block's rows (offset term, alpha).  Requested HERE, before the stream's first trips and
  // held in registers: the loads are older than every load of the stream, so they have arrived long before the epilogue,
  // which otherwise spends 3-4 dependent rounds of global loads per workgroup with nothing else running on the CU
  // (8-byte types with at most kPbXPre rows per lane; LL_PB_XPRE=0 in the launcher: load them in the epilogue, round-3 form).
  T xpre[kPbXPre];
  if constexpr (sizeof(T) <= 8) {
    if (xprefetch) {
#pragma unroll
      for (int u = 0; u < kPbXPre; ++u) {
        const int i = tid + u * kPbThreads;
        xpre[u] = xl[row0 + (i < rows ? i : rows - 1)];
      }
    }
  }
  RowStream<T, U, D, true> st;
  st.P = P;
  st.row = row;
  st.g1 = g1;
  st.glast = g1 > g0 ? g1 - 1 : g0;
  st.chunks.ctr = &chunk_ctr;
  st.chunks.g0 = g0;
  st.prologue();
  if (tid == 0) chunk_ctr = (D - 1) * kPbWaves;
  {  // exponent of max |x| over ALL columns (every column block left its slice maximum)
    double m = 0.0;
    for (int i = tid; i < ncb; i += kPbThreads) m = fmax(m, blockmax[i]);
    for (int i = tid; i < rb_rows * R; i += kPbThreads) acc[i] = 0;
    const double t = pb_block_max(m, red);
    int e_x = -2000;  // x == 0: any scale does
    if (t > 0.0 && isfinite(t)) (void)frexp(t, &e_x);  // t < 2^e
    else if (!(t == 0.0)) e_x = kPbXInf;               // Inf: every row of the block is reported as NaN
    for (int i = tid; i < rb_rows; i += kPbThreads) {
      int k = kBadRow;
      if (i < rows) {
        const int er = rexp[row0 + i];
        if (er != 32767 && e_x != kPbXInf) k = max(-1000, min(1000, 62 - (er + e_x + 1)));
      } else {
        k = 0;
      }
      ex[i] = (int16_t)k;
    }
    __syncthreads();
  }
  st.run([&](const quad<T>(&p)[U], const ushort4(&r)[U], const long long(&g)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (g[u] < g1) {
        const unsigned short rr[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = ex[rr[e]];
          const double sc = pow2(k == kBadRow ? 0 : k);
          long long w[R];
          if constexpr (scalar_traits<T>::is_complex) {
            w[0] = pb_to_fixed((double)p[u].e[e].re * sc);
            w[1] = pb_to_fixed((double)p[u].e[e].im * sc);
          } else {
            w[0] = pb_to_fixed((double)p[u].e[e] * sc);
          }
          bool ok = true;
#pragma unroll
          for (int q = 0; q < R; ++q) {
            if (w[q] == kPbBadProduct) ok = false;
            else atomicAdd(reinterpret_cast<unsigned long long*>(&acc[R * rr[e] + q]), (unsigned long long)w[q]);
          }
          if (!ok) ex[rr[e]] = (int16_t)kBadRow;  // the row's result is NaN (same value from every writer)
        }
      }
    }
  });
  __syncthreads();
  // (the row's diagonal entry is kept outside the streams, pb_diag_kernel: its product joins the row's integers here —
  // rounded to the same grid, added to the same sum: the same bits as if it had travelled through the product buffer)
  auto value = [&](int i, const T& xi, const T& dg) {  // dg: the row's diagonal entry (requested ahead, see the epilogue)
    int k = ex[i];
    long long s[R];
#pragma unroll
    for (int q = 0; q < R; ++q) s[q] = acc[R * i + q];
    if (diag != nullptr && k != kBadRow) {
      const T p = mul(dg, xi);
      const double sc = pow2(k);
      long long w[R];
      if constexpr (scalar_traits<T>::is_complex) {
        w[0] = pb_to_fixed((double)p.re * sc);
        w[1] = pb_to_fixed((double)p.im * sc);
      } else {
        w[0] = pb_to_fixed((double)p * sc);
      }
#pragma unroll
      for (int q = 0; q < R; ++q) {
        if (w[q] == kPbBadProduct) k = kBadRow;
        else s[q] = (long long)((unsigned long long)s[q] + (unsigned long long)w[q]);
      }
    }
    const double back = k == kBadRow ? __longlong_as_double(0x7ff8000000000000ll) : pow2(-k);  // NaN for unusable rows
    acc_t<T> a;
    if constexpr (scalar_traits<T>::is_complex) a = zc{(double)s[0] * back, (double)s[1] * back};
    else a = (double)s[0] * back;
    return a;
  };
  if constexpr (sizeof(T) <= 8) {
    if (xprefetch) {  // same arithmetic, same order of the per-lane dot sum (rows tid, tid + 1024, ...) as the loading form
      const double xs_fac = xnorm2 ? 1.0 / sqrt(*xnorm2) : 1.0;
      double dot_acc = 0.0;
      // the diagonal entries of the lane's rows in one batch (clamped addresses, no load under the per-row branch): the ring's
      // registers are free here; requested row by row they were a chain of up to 13 memory latencies — the 17 us of this epilogue
      T dgs[kPbXPre];
#pragma unroll
      for (int u = 0; u < kPbXPre; ++u) {
        const int i = tid + u * kPbThreads;
        dgs[u] = diag ? diag[row0 + (i < rows ? i : rows - 1)] : zero<T>();
      }
#pragma unroll
      for (int u = 0; u < kPbXPre; ++u) {
        const int i = tid + u * kPbThreads;
        if (i < rows) {
          const T xi = rmul(xs_fac, xpre[u]);
          const T yi = add(narrow<T>(value(i, xi, dgs[u])), rmul(offset, xi));
          y[row0 + i] = yi;
          dot_acc += re_cmul(xi, yi);
        }
      }
      if (dot_partials) {
        const double v = wave_sum(dot_acc);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
          double t = 0.0;
          for (int w = 0; w < kPbWaves; ++w) t += red[w];
          dot_partials[rb] = t;
        }
      }
      return;
    }
  }
  pb_phase2_epilogue<T>(rb, row0, rows, xl, y, offset, dot_partials, red, xnorm2, value,
                        [&](int i) { return diag ? diag[row0 + i] : zero<T>(); });
}

// exponent of every local row's absolute sum: sum_j |a_ij| < 2^rexp[i]  (32767: the row holds Inf / NaN)
template <typename T, typename RP>
__global__ __launch_bounds__(256) void pb_rowexp_kernel(long long n_local, const RP* __restrict__ rp,
                                                        const T* __restrict__ va, int16_t* __restrict__ rexp) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_local; i += (long long)gridDim.x * 256) {
    double s = 0.0;
    for (long long p = (long long)rp[i]; p < (long long)rp[i + 1]; ++p) s += abs1(va[p]);
    int e = -1100;
    if (s > 0.0 && isfinite(s)) (void)frexp(s, &e);
    else if (!(s == 0.0)) e = 32767;
    rexp[i] = (int16_t)e;
  }
}

// ================================================================= launchers
namespace {
constexpr int kPbLdsCap = 160 * 1024 - 2048;

// the opt-in to > 64 KiB of dynamic LDS is per device and per kernel symbol
template <typename K> void pb_opt_in(K kernel) {
  LL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kPbLdsCap));
}
// trips of loads in flight per lane: three in both phases (two for the 16-byte types in phase 2, where a third trip spills
// registers).  Measured on config 3 (profiles/r03_spmv_variants.txt): 3/3 0.898 ms, 2/2 0.937, 4/3 0.892, 3/2 0.890, 2/3
// 0.889 on one box — beyond two trips the differences are inside the 2-4 % position noise of one process.
template <typename T> constexpr int pb_depth2() { return sizeof(T) < 16 ? 3 : 2; }
constexpr int kPbDepth1 = 3;

template <typename T> void pb_opt_in_lds() {
  static std::atomic<unsigned long long> mask{0};
  int dev = 0;
  LL_HIP(hipGetDevice(&dev));
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask.load(std::memory_order_acquire) & bit) return;
  constexpr int D2 = pb_depth2<T>();
  pb_opt_in(&pb_phase1<T, kPbDepth1, kPbThreads>);
  pb_opt_in(&pb_phase1<T, kPbDepth1, 256>);
  pb_opt_in(&pb_phase1<T, kPbDepth1, 512>);
  pb_opt_in(&pb_phase2<T, false, D2>);
  pb_opt_in(&pb_phase2<T, true, D2>);
  pb_opt_in(&pb_phase2_fixed<T, D2>);
  mask.fetch_or(bit, std::memory_order_release);
}

template <typename T>
void phase1_launch(const ll_operator& op, int blk_first, int blk_count, const T* xsrc, const double* xnorm2, hipStream_t s) {
  const size_t lds1 = (((size_t)op.pb_cb_cols * sizeof(T) + 15) & ~(size_t)15) + (size_t)(2 * op.pb_nrb + 1) * sizeof(long long);
  double* bm = op.pb_phase2 == LL_PB_FIXED ? op.d_pb_blockmax : nullptr;
  // Fewer lanes per workgroup for the thin column blocks of a sharded image: at N = 8 a block holds 16 K entries — four trips of
  // 1024 lanes, two of them requested by the prologue: the ring never reaches its steady state — but 8 trips of 512 lanes
  // (measured on config 4's shards through the stand-in transport, profiles/r05_shard_phase1_lanes.txt: 60.8 -> 41.9 us per
  // launch at N = 8 with the same 68 KB slices; smaller slices with more workgroups per CU lose to their segment count)
  if (op.pb_threads1 == 256)
    hipLaunchKernelGGL((pb_phase1<T, kPbDepth1, 256>), dim3(blk_count), dim3(256), lds1, s, op.pb_nrb, blk_first, op.d_pb_xoff,
                       op.d_pb_ncols, op.d_pb_segq, op.d_pb_segdest, (const T*)op.d_pb_val, (const ushort4*)op.d_pb_col, xsrc,
                       (T*)op.d_pb_prod, op.pb_cb_cols, bm, (long long)op.pb_entries, xnorm2);
  else if (op.pb_threads1 == 512)
    hipLaunchKernelGGL((pb_phase1<T, kPbDepth1, 512>), dim3(blk_count), dim3(512), lds1, s, op.pb_nrb, blk_first, op.d_pb_xoff,
                       op.d_pb_ncols, op.d_pb_segq, op.d_pb_segdest, (const T*)op.d_pb_val, (const ushort4*)op.d_pb_col, xsrc,
                       (T*)op.d_pb_prod, op.pb_cb_cols, bm, (long long)op.pb_entries, xnorm2);
  else
    hipLaunchKernelGGL((pb_phase1<T, kPbDepth1, kPbThreads>), dim3(blk_count), dim3(kPbThreads), lds1, s, op.pb_nrb, blk_first,
                       op.d_pb_xoff, op.d_pb_ncols, op.d_pb_segq, op.d_pb_segdest, (const T*)op.d_pb_val,
                       (const ushort4*)op.d_pb_col, xsrc, (T*)op.d_pb_prod, op.pb_cb_cols, bm, (long long)op.pb_entries, xnorm2);
  LL_HIP(hipGetLastError());
}
}  // namespace

template <typename T>
void launch_pb_phase1(const ll_operator& op, int blk_first, int blk_count, const T* xsrc, hipStream_t s, const double* xnorm2) {
  if (blk_count <= 0 || op.pb_nrb <= 0) return;
  pb_opt_in_lds<T>();
  phase1_launch<T>(op, blk_first, blk_count, xsrc, xnorm2, s);
}

template <typename T>
int launch_pb_phase2(const ll_operator& op, const T* x_local, T* y, double offset, double* dot_partials,
                     hipStream_t s, const double* xnorm2) {
  if (op.pb_nrb <= 0) return 0;
  pb_opt_in_lds<T>();
  const dim3 grid(op.pb_nrb), block(kPbThreads);
  constexpr int D2 = pb_depth2<T>();
  if (op.pb_phase2 == LL_PB_FIXED) {  // order-independent fixed-point sums, LATE form
    const size_t ldsf = (size_t)op.pb_rb_rows * (sizeof(acc_t<T>) + sizeof(int16_t)) + 16;
    hipLaunchKernelGGL((pb_phase2_fixed<T, D2>), grid, block, ldsf, s, op.pb_rb_rows, op.n_local, op.pb_ncb, op.d_pb_rptr,
                       (const ushort4*)op.d_pb_row, (const T*)op.d_pb_prod, op.d_pb_rexp, op.d_pb_blockmax, x_local, y,
                       offset, dot_partials, xnorm2, (op.pb_xpre && op.pb_rb_rows <= kPbXPre * kPbThreads) ? 1 : 0,
                       (const T*)op.d_pb_diag);
  } else {
    const size_t lds2 = (size_t)op.pb_rb_rows * sizeof(acc_t<T>);
    if (op.pb_phase2 == LL_PB_ORDERED)
      hipLaunchKernelGGL((pb_phase2<T, true, D2>), grid, block, lds2, s, op.pb_rb_rows, op.n_local, op.d_pb_rptr,
                         (const ushort4*)op.d_pb_row, (const T*)op.d_pb_prod, x_local, y, offset, dot_partials, xnorm2,
                         (const T*)op.d_pb_diag);
    else
      hipLaunchKernelGGL((pb_phase2<T, false, D2>), grid, block, lds2, s, op.pb_rb_rows, op.n_local, op.d_pb_rptr,
                         (const ushort4*)op.d_pb_row, (const T*)op.d_pb_prod, x_local, y, offset, dot_partials, xnorm2,
                         (const T*)op.d_pb_diag);
  }
  LL_HIP(hipGetLastError());
  return op.pb_nrb;
}

template <typename T>
int launch_spmv_pb(const ll_operator& op, const T* x_gathered, const T* x_own, const T* x_local, T* y, double offset,
                   double* dot_partials, hipStream_t s, const double* xnorm2) {
  launch_pb_phase1<T>(op, 0, op.pb_own_count, x_own, s, xnorm2);
  for (int c = 0; c < op.gather.nchunks; ++c)
    launch_pb_phase1<T>(op, op.pb_chunk_first[c], op.pb_chunk_count[c], x_gathered, s, xnorm2);
  return launch_pb_phase2<T>(op, x_local, y, offset, dot_partials, s, xnorm2);
}

// ================================================================= device-side construction of the image
// Column -> (table index of its column block, local column).  Blocks never straddle a (rank, gather chunk) region,
// so the x slice of a block is contiguous both in the local shard and in the gathered buffer.
struct PbColMap {
  long long shard;  // shard stride of the vector partition (n when not sharded)
  int nranks, rank, nchunks;
  long long cstart[kMaxGatherChunks];  // chunk start within a shard
  int clen[kMaxGatherChunks];          // chunk length
  int bl[kMaxGatherChunks];            // block length inside the chunk
  int nb[kMaxGatherChunks];            // blocks per (rank, chunk)
  int own_base[kMaxGatherChunks];      // table index of the first own block of chunk c
  int rem_base[kMaxGatherChunks];      // table index of the first remote block of chunk c
};
__host__ __device__ inline int pb_block_of(const PbColMap& m, long long colg, int* local) {
  const long long s = colg / m.shard;
  const long long l = colg - s * m.shard;
  int c = 0;
  while (c + 1 < m.nchunks && l >= m.cstart[c + 1]) ++c;
  const int within = (int)(l - m.cstart[c]);
  const int b = within / m.bl[c];
  *local = within - b * m.bl[c];
  if ((int)s == m.rank) return m.own_base[c] + b;
  const int so = (m.rank >= 0 && (int)s > m.rank) ? (int)s - 1 : (int)s;  // rank < 0: no own blocks at all
  return m.rem_base[c] + so * m.nb[c] + b;
}

// Pass 1: entries per (column block, row block).  One workgroup per row block, LDS histogram over the column blocks.
template <typename RP>
__global__ __launch_bounds__(256) void pb_count_kernel(PbColMap m, int ncb, int nrb, int rb_rows, long long n_local,
                                                       const RP* __restrict__ rp, const int32_t* __restrict__ ci,
                                                       int32_t* __restrict__ cnt /* [ncb][nrb] */,
                                                       const uint32_t* __restrict__ skip = nullptr) {
  extern __shared__ int hist[];
  const int r = blockIdx.x;
  for (int i = threadIdx.x; i < ncb; i += 256) hist[i] = 0;
  __syncthreads();
  const long long i0 = (long long)r * rb_rows, i1 = min(n_local, i0 + rb_rows);
  const long long p0 = (long long)rp[i0], p1 = (long long)rp[i1];
  for (long long p = p0 + threadIdx.x; p < p1; p += 256) {
    if (skip != nullptr && ((skip[p >> 5] >> (p & 31)) & 1u)) continue;  // the row's diagonal entry: kept outside the image
    int local;
    atomicAdd(&hist[pb_block_of(m, ci[p], &local)], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ncb; i += 256) cnt[(size_t)i * nrb + r] = hist[i];
}

// Pass 2: scatter.  ONE wavefront per row block walks the block's entries in CSR order, 64 at a time; entries of the
// chunk that fall into the same segment get consecutive slots in lane order (ballot ranking), so the position of
// every entry is a pure function of the matrix: the image — hence the summation order of phase 2 — is identical
// from build to build.  Inside a segment both orders agree (row-major, original order within a row).
// TILED (the 2-D tiled image, tl_* below): ONE order — the segments of a row block are consecutive (segq is laid out row
// block by row block, segdest is not used) — the two 16-bit local indices of an entry are packed into one 32-bit word
// (pcol is that array, prow is not used) and the value is stored PRE-SCALED by 2^-er_i (rexp: exponent of the row's
// absolute sum), an exact operation that takes the per-row exponent out of the kernel's inner loop.
__device__ __forceinline__ double scale_pow2(double v, int k) { return ldexp(v, k); }
__device__ __forceinline__ float scale_pow2(float v, int k) { return ldexpf(v, k); }
__device__ __forceinline__ zc scale_pow2(zc v, int k) { return zc{ldexp(v.re, k), ldexp(v.im, k)}; }
__device__ __forceinline__ cf scale_pow2(cf v, int k) { return cf{ldexpf(v.re, k), ldexpf(v.im, k)}; }
template <typename T, typename RP, bool TILED = false>
__global__ __launch_bounds__(64) void pb_scatter_kernel(PbColMap m, int ncb, int nrb, int rb_rows, long long n_local,
                                                        const RP* __restrict__ rp, const int32_t* __restrict__ ci,
                                                        const T* __restrict__ va, const int64_t* __restrict__ segq,
                                                        const int64_t* __restrict__ segdest, T* __restrict__ pval,
                                                        uint16_t* __restrict__ pcol, uint16_t* __restrict__ prow,
                                                        const int16_t* __restrict__ rexp = nullptr,
                                                        const uint32_t* __restrict__ skip = nullptr) {
  extern __shared__ int fill[];  // [ncb]
  const int r = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < ncb; i += 64) fill[i] = 0;
  __syncthreads();
  const long long i0 = (long long)r * rb_rows, i1 = min(n_local, i0 + rb_rows);
  const long long p0 = (long long)rp[i0], p1 = (long long)rp[i1];
  long long row_hint = i0;  // row of the chunk's first entry (monotone over the chunks)
  for (long long base = p0; base < p1; base += 64) {
    const long long p = base + lane;
    const bool in_range = p < p1;
    const bool valid = in_range && !(skip != nullptr && ((skip[p >> 5] >> (p & 31)) & 1u));  // (the diagonal entry stays outside)
    // row of entry p: the last row i with rp[i] <= p (binary search from the hint; rows of a chunk are few)
    long long lo = row_hint, hi = i1 - 1;
    if (in_range) {
      while (lo < hi) {
        const long long mid = (lo + hi + 1) >> 1;
        if ((long long)rp[mid] <= p) lo = mid; else hi = mid - 1;
      }
    }
    const long long rowi = lo;
    row_hint = __shfl(rowi, 0, 64);
    int local = 0;
    const int key = valid ? pb_block_of(m, ci[p], &local) : -1;
    int off = 0;
    unsigned long long todo = __ballot(valid);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int k = __shfl(key, leader, 64);
      const unsigned long long same = __ballot(valid && key == k);
      if (valid && key == k) off = fill[k] + __popcll(same & ((1ull << lane) - 1ull));
      __syncthreads();  // single wave: orders the LDS read above before the update below
      if (lane == leader) fill[k] += __popcll(same);
      __syncthreads();
      todo &= ~same;
    }
    if (valid) {
      const long long q = segq[(size_t)key * (nrb + 1) + r] + off;
      if constexpr (TILED) {
        const int er = rexp[rowi];
        pval[q] = (er == 32767 || er <= -1100) ? va[p] : scale_pow2(va[p], -er);  // (rows with Inf / NaN, empty rows: as is)
        reinterpret_cast<uint32_t*>(pcol)[q] = (uint32_t)local | ((uint32_t)(rowi - i0) << 16);
      } else {
        const long long qd = segdest[(size_t)key * nrb + r] + off;
        pval[q] = va[p];
        pcol[q] = (uint16_t)local;
        prow[qd] = (uint16_t)(rowi - i0);
      }
    }
  }
}

// The first diagonal entry of every local row leaves the image: it is the one entry whose x element the row's own epilogue
// holds anyway (offset term, alpha), so its product needs no trip through the product buffer — 28 bytes of traffic per row
// saved for 8 (the diag array), 1/15 of config 3's entries.  diag[i] = a_ii (0 when the row stores none), skip: one bit
// per CSR entry, set for the entries taken out (read by the count and scatter kernels).  Further diagonal duplicates of a
// row stay in the streams.
template <typename T, typename RP>
__global__ __launch_bounds__(256) void pb_diag_kernel(long long n_local, long long row_begin, const RP* __restrict__ rp,
                                                      const int32_t* __restrict__ ci, const T* __restrict__ va,
                                                      T* __restrict__ diag, uint32_t* __restrict__ skip) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_local; i += (long long)gridDim.x * 256) {
    T d = zero<T>();
    const long long g = row_begin + i;
    for (long long p = (long long)rp[i]; p < (long long)rp[i + 1]; ++p)
      if ((long long)ci[p] == g) {
        d = va[p];
        atomicOr(&skip[p >> 5], 1u << (p & 31));
        break;
      }
    diag[i] = d;
  }
}

// max_i sum_j |a_ij| over the local rows and the column range check, on the device (inputs may never have been on
// the host).  out[0] = max row sum, out[1] = number of out-of-range columns.
template <typename T, typename RP>
__global__ __launch_bounds__(256) void csr_check_kernel(long long n_local, long long n_cols, const RP* __restrict__ rp,
                                                        const int32_t* __restrict__ ci, const T* __restrict__ va,
                                                        double* __restrict__ part_max, unsigned long long* __restrict__ bad) {
  __shared__ double red[4];
  double mx = 0.0;
  unsigned long long nbad = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_local; i += (long long)gridDim.x * 256) {
    double rs = 0.0;
    for (long long p = (long long)rp[i]; p < (long long)rp[i + 1]; ++p) {
      rs += sqrt(abs2(va[p]));
      const int c = ci[p];
      if (c < 0 || c >= n_cols) ++nbad;
    }
    mx = fmax(mx, rs);
  }
  // fold: max over the workgroup (wave shuffles + LDS), count via one atomic per lane that saw a bad column
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mx = fmax(mx, __shfl_down(mx, d, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) part_max[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  if (nbad) atomicAdd(bad, nbad);
}

template <typename T> void csr_check_device(ll_operator* op) {
  const long long nr = op->n_local;
  const int grid = (int)std::max<long long>(1, std::min<long long>(kMaxGrid, (nr + 255) / 256));
  double* d_part = nullptr;
  unsigned long long* d_bad = nullptr;
  op->ctx->dev_malloc((void**)&d_part, (size_t)grid * sizeof(double), "row-sum partials");
  op->ctx->dev_malloc((void**)&d_bad, sizeof(unsigned long long), "column check");
  struct Guard {
    void *a, *b;
    ~Guard() {
      (void)hipFree(a);
      (void)hipFree(b);
    }
  } guard{d_part, d_bad};
  hipStream_t s = op->ctx->stream;
  LL_HIP(hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), s));
  if (op->rp64)
    hipLaunchKernelGGL((csr_check_kernel<T, int64_t>), dim3(grid), dim3(256), 0, s, nr, (long long)op->n,
                       (const int64_t*)op->d_row_ptr, op->d_col, (const T*)op->d_val, d_part, d_bad);
  else
    hipLaunchKernelGGL((csr_check_kernel<T, int32_t>), dim3(grid), dim3(256), 0, s, nr, (long long)op->n,
                       (const int32_t*)op->d_row_ptr, op->d_col, (const T*)op->d_val, d_part, d_bad);
  LL_HIP(hipGetLastError());
  std::vector<double> part((size_t)grid);
  unsigned long long bad = 0;
  LL_HIP(hipMemcpyAsync(part.data(), d_part, (size_t)grid * sizeof(double), hipMemcpyDeviceToHost, s));
  LL_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, s));
  LL_HIP(hipStreamSynchronize(s));
  LL_REQUIRE(bad == 0, "column index out of range");
  double mx = 0.0;
  for (double v : part) mx = std::max(mx, v);
  op->inf_norm = mx;
}

// Block geometry + tables on the host (O(ncb * nrb)), entry placement on the device.  Returns false when the image
// cannot be built for this shape (segment tables too large, or the LDS budget cannot hold a slice and its tables):
// the operator then keeps CSR-stream.
template <typename T> bool pb_build_device(ll_operator* op) {
  ll_context* ctx = op->ctx;
  hipStream_t s = ctx->stream;
  const int64_t nr = op->n_local;
  const int P = std::max(1, ctx->nranks);
  const int64_t S = P > 1 ? op->n_shard : op->n;

  // ---- row blocks (y slice in LDS, 152 KiB at most)
  // (the fixed-point form of phase 2 keeps a 16-bit exponent per row next to the accumulator)
  // (the fixed-point form of phase 2 keeps a 16-bit exponent per row next to the accumulator)
  const int64_t row_max = std::min<int64_t>(65536, (152 * 1024) / (int64_t)(sizeof(acc_t<T>) + sizeof(int16_t)));
  const Tuning& tune = ctx->tune;
  auto block_len = [&](int64_t len, int64_t slice_max, int forced) {
    int64_t m = std::max<int64_t>(1, (len + 256 * slice_max - 1) / (256 * slice_max));
    int64_t b = std::max<int64_t>(16, (len + 256 * m - 1) / (256 * m));
    if (tune.pb_block > 0) b = std::max(4, tune.pb_block);
    if (forced > 0) b = std::max(4, forced);
    return std::min<int64_t>(b, slice_max);
  };
  const int64_t rb_rows = block_len(nr, row_max, tune.pb_row_block);
  const int64_t nrb = std::max<int64_t>(1, (nr + rb_rows - 1) / rb_rows);
  // ---- LDS budget of phase 1: x slice + two tables of nrb entries (the advisor's round-1 finding: check it here)
  const int64_t table_bytes = (2 * nrb + 1) * (int64_t)sizeof(long long) + 16;
  if (table_bytes + 1024 > kPbLdsCap) return false;
  const int64_t col_max = std::min<int64_t>(65536, std::min<int64_t>(104 * 1024, kPbLdsCap - table_bytes) / (int64_t)sizeof(T));
  if (col_max < 16) return false;

  // ---- gather chunks and column blocks
  PbColMap m;
  std::memset(&m, 0, sizeof(m));
  m.shard = S;
  m.nranks = P;
  m.rank = ctx->comm != nullptr ? ctx->rank : 0;
  // test hook: treat the rank's own columns like everybody else's (every x slice is read from the gathered buffer), so
  // that a 1-rank RCCL communicator exercises the gather -> remote-block dependency of the overlapped path
  if (ctx->comm != nullptr && tune.pb_test_all_remote) m.rank = -1;
  const int nrem = m.rank >= 0 ? P - 1 : P;  // ranks whose columns are remote
  GatherPlan gp;
  {
    // Chunks pipeline the remote-column part of phase 1 behind the transfer, but every extra collective costs launch
    // latency and small messages use xGMI less efficiently: 4 pieces on 2 GPUs (half of the columns are own, long
    // transfer over one link pair), 2 pieces on more (the own part is 1/P, the transfer uses all links at once).
    int want = P > 1 ? (tune.gather_chunks > 0 ? tune.gather_chunks : (P == 2 ? 4 : 2)) : 1;
    want = std::max(1, std::min(want, kMaxGatherChunks));
    int64_t lc = (S + want - 1) / want;
    lc = std::max<int64_t>(256, (lc + 255) / 256 * 256);  // chunk starts stay 2 KiB aligned
    gp.nchunks = (int)std::max<int64_t>(1, (S + lc - 1) / lc);
    for (int c = 0; c < gp.nchunks; ++c) {
      gp.start[c] = c * lc;
      gp.len[c] = std::min<int64_t>(lc, S - c * lc);
    }
  }
  m.nchunks = gp.nchunks;
  int own_total = 0;
  for (int c = 0; c < gp.nchunks; ++c) {
    m.cstart[c] = gp.start[c];
    m.clen[c] = (int)gp.len[c];
    int64_t bl;
    if (P == 1) bl = block_len(gp.len[c], col_max, tune.pb_col_block);
    else {
      // Sharded: one launch of phase 1 covers the nrem other ranks' blocks of this chunk, one workgroup per block and
      // per CU at a time.  Its duration is rounds x (fixed + per-column work) with rounds = ceil(blocks / CUs), so the
      // block count per (rank, chunk) is chosen to fill whole rounds instead of the fewest blocks that fit the LDS:
      // N = 8, two chunks: 48 blocks of 13 021 columns per rank give launches of 336 blocks = 1.31 -> 2 rounds; 73
      // blocks of 8 562 columns give 511 blocks = 2 full rounds of 2/3 the length (measured on the shard shapes with
      // tools/shard_compute_probe.py).  kFixedCols: the per-workgroup fixed cost in units of one column's work.
      const int64_t nb_min = (gp.len[c] + col_max - 1) / col_max;
      const double kFixedCols = 1500.0;
      int64_t best_nb = nb_min;
      double best_cost = 1e300;
      for (int64_t nb = nb_min; nb <= 4 * nb_min + 8; ++nb) {
        const int64_t len_b = (gp.len[c] + nb - 1) / nb;
        const int64_t launch = (int64_t)std::max(1, nrem) * nb;
        const double cost = (double)((launch + kCUs - 1) / kCUs) * (kFixedCols + (double)len_b);
        if (cost < best_cost * 0.999) {
          best_cost = cost;
          best_nb = nb;
        }
      }
      bl = (gp.len[c] + best_nb - 1) / best_nb;
      bl = std::min<int64_t>(col_max, (bl + 3) / 4 * 4);
      if (tune.pb_block > 0) bl = std::min<int64_t>(col_max, std::max(4, tune.pb_block));
      if (tune.pb_col_block > 0) bl = std::min<int64_t>(col_max, std::max(4, tune.pb_col_block));
    }
    m.bl[c] = (int)bl;
    m.nb[c] = (int)((gp.len[c] + bl - 1) / bl);
    m.own_base[c] = own_total;
    if (m.rank >= 0) own_total += m.nb[c];
  }
  int64_t ncb = own_total;
  for (int c = 0; c < gp.nchunks; ++c) {
    m.rem_base[c] = (int)ncb;
    op->pb_chunk_first[c] = (int)ncb;
    op->pb_chunk_count[c] = nrem * m.nb[c];
    ncb += (int64_t)nrem * m.nb[c];
  }
  if (ncb * nrb > (int64_t)24 << 20) return false;  // segment tables would not pay off (n beyond ~6e7): keep CSR
  if (ncb * (int64_t)sizeof(int) > 60 * 1024) return false;  // LDS histogram of the build kernels
  int cb_cols = 0;
  std::vector<int64_t> xoff((size_t)ncb);
  std::vector<int32_t> ncols((size_t)ncb);
  for (int c = 0; c < gp.nchunks; ++c) {
    cb_cols = std::max(cb_cols, m.bl[c]);
    for (int sr = 0; sr < P; ++sr)
      for (int b = 0; b < m.nb[c]; ++b) {
        const bool own = sr == m.rank;
        const int idx = own ? m.own_base[c] + b
                            : m.rem_base[c] + ((m.rank >= 0 && sr > m.rank) ? sr - 1 : sr) * m.nb[c] + b;
        ncols[(size_t)idx] = (int32_t)std::min<int64_t>(m.bl[c], gp.len[c] - (int64_t)b * m.bl[c]);
        xoff[(size_t)idx] = own ? gp.start[c] + (int64_t)b * m.bl[c]
                                : (int64_t)P * gp.start[c] + (int64_t)sr * gp.len[c] + (int64_t)b * m.bl[c];
      }
  }

  // ---- the diagonal leaves the image (pb_diag_kernel): one bit per CSR entry marks what the passes below skip
  uint32_t* d_skip = nullptr;
  struct Free0 {
    uint32_t*& p;
    ~Free0() {
      if (p) (void)hipFree(p);
    }
  } free_skip{d_skip};
  if (tune.pb_diag && nr > 0 && op->nnz > 0) {
    const size_t words = ((size_t)op->nnz + 31) / 32 + 1;
    ctx->dev_malloc((void**)&d_skip, words * sizeof(uint32_t), "diagonal marks");
    ctx->dev_malloc(&op->d_pb_diag, std::max<size_t>((size_t)nr, 2) * sizeof(T), "diagonal of the PB image");
    LL_HIP(hipMemsetAsync(d_skip, 0, words * sizeof(uint32_t), s));
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>(kMaxGrid, (nr + 255) / 256));
    if (op->rp64)
      hipLaunchKernelGGL((pb_diag_kernel<T, int64_t>), dim3(g), dim3(256), 0, s, (long long)nr, (long long)op->row_begin,
                         (const int64_t*)op->d_row_ptr, op->d_col, (const T*)op->d_val, (T*)op->d_pb_diag, d_skip);
    else
      hipLaunchKernelGGL((pb_diag_kernel<T, int32_t>), dim3(g), dim3(256), 0, s, (long long)nr, (long long)op->row_begin,
                         (const int32_t*)op->d_row_ptr, op->d_col, (const T*)op->d_val, (T*)op->d_pb_diag, d_skip);
    LL_HIP(hipGetLastError());
  }
  // ---- pass 1 on the device: segment sizes
  int32_t* d_cnt = nullptr;
  ctx->dev_malloc((void**)&d_cnt, (size_t)ncb * nrb * sizeof(int32_t), "segment counts");
  struct Free1 {
    void* p;
    ~Free1() { (void)hipFree(p); }
  } free_cnt{d_cnt};
  const size_t hist_bytes = (size_t)ncb * sizeof(int);
  if (op->rp64)
    hipLaunchKernelGGL((pb_count_kernel<int64_t>), dim3((int)nrb), dim3(256), hist_bytes, s, m, (int)ncb, (int)nrb,
                       (int)rb_rows, (long long)nr, (const int64_t*)op->d_row_ptr, op->d_col, d_cnt, d_skip);
  else
    hipLaunchKernelGGL((pb_count_kernel<int32_t>), dim3((int)nrb), dim3(256), hist_bytes, s, m, (int)ncb, (int)nrb,
                       (int)rb_rows, (long long)nr, (const int32_t*)op->d_row_ptr, op->d_col, d_cnt, d_skip);
  LL_HIP(hipGetLastError());
  std::vector<int32_t> cnt32((size_t)ncb * nrb);
  LL_HIP(hipMemcpyAsync(cnt32.data(), d_cnt, cnt32.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  LL_HIP(hipStreamSynchronize(s));
  // every segment is padded to 16 entries: kernels move quads (4 entries per lane, 16-byte accesses) and every run of
  // products written by phase 1 starts and ends on a 128-byte line (measured 3.5 % faster than quad padding)
  // (sharded images: quad padding.  Their segments hold 105 entries at N = 8 instead of 840, the 16-entry padding costs 8 % of both
  // streams there, and phase 2 measured 54.9 -> 47.8 us without it)
  const int64_t pad = tune.pb_pad > 0 ? tune.pb_pad : (P > 1 ? 4 : 16);
  // column-block order: segments (c, r) with r fastest; row-block order: (r, c) with c fastest
  std::vector<int64_t> segq((size_t)ncb * (nrb + 1)), segdest((size_t)ncb * nrb), rptr((size_t)nrb + 1);
  {
    int64_t q = 0;
    for (int64_t c = 0; c < ncb; ++c) {
      for (int64_t r = 0; r < nrb; ++r) {
        segq[(size_t)c * (nrb + 1) + r] = q;
        q += ((int64_t)cnt32[(size_t)c * nrb + r] + pad - 1) / pad * pad;
      }
      segq[(size_t)c * (nrb + 1) + nrb] = q;
    }
    int64_t d = 0;
    for (int64_t r = 0; r < nrb; ++r) {
      rptr[(size_t)r] = d;
      for (int64_t c = 0; c < ncb; ++c) {
        segdest[(size_t)c * nrb + r] = d;
        d += ((int64_t)cnt32[(size_t)c * nrb + r] + pad - 1) / pad * pad;
      }
    }
    rptr[(size_t)nrb] = d;
  }
  const size_t entries = (size_t)rptr[(size_t)nrb];
  op->pb_ncb = (int)ncb;
  op->pb_nrb = (int)nrb;
  op->pb_cb_cols = cb_cols;
  op->pb_rb_rows = (int)rb_rows;
  op->pb_own_count = own_total;
  op->pb_entries = (int64_t)entries;
  op->pb_xpre = tune.pb_xpre;
  op->pb_threads1 = tune.pb_threads1 > 0 ? tune.pb_threads1 : (P > 1 ? 512 : kPbThreads);
  op->gather = gp;
  // Phase 2 form, fixed per operator at creation (LL_PB_PHASE2).  Default "fixed": order-independent fixed-point sums
  // (integer LDS adds, all waves at once) — bit-reproducible for every launch, kernel geometry and partition, and 2-5 %
  // faster than the wave-ordered form (profiles/r02_spmv_variants_run10_fixed_default.jsonl).  "ordered": floating-
  // point adds, the 16 waves in turn (barriers) — reproducible and component-wise accurate; "atomic": floating-point
  // adds in arrival order (not reproducible; A/B timing reference).
  // The caller's choice through ll_csr_options.accuracy / ll_op_set_accuracy (include/lanczos_hip.h) outranks the environment.
  op->pb_phase2 = op->accuracy_req == LL_ACCURACY_COMPONENTWISE
                      ? (tune.pb_phase2 == LL_PB_ATOMIC ? LL_PB_ATOMIC : LL_PB_ORDERED)
                      : (op->accuracy_req == LL_ACCURACY_NORMWISE ? LL_PB_FIXED : tune.pb_phase2);
  auto up = [&](void** dst, const void* src, size_t bytes) {
    ctx->dev_malloc(dst, bytes, "propagation-blocking tables");
    LL_HIP(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, s));
  };
  up((void**)&op->d_pb_segq, segq.data(), segq.size() * sizeof(int64_t));
  up((void**)&op->d_pb_segdest, segdest.data(), segdest.size() * sizeof(int64_t));
  up((void**)&op->d_pb_rptr, rptr.data(), rptr.size() * sizeof(int64_t));
  up((void**)&op->d_pb_xoff, xoff.data(), xoff.size() * sizeof(int64_t));
  up((void**)&op->d_pb_ncols, ncols.data(), ncols.size() * sizeof(int32_t));
  // 16 entries behind the image: the dump quad of phase 1 and the clamped reads of an empty last block
  const size_t cap = entries + 16;
  // ONE allocation for the four big streams (values, local columns, local rows, product buffer), starts 2 MiB aligned
  // (staggering the starts against the HBM channel interleave was measured in round 2: no effect).
  {
    const size_t stagger = 0;
    auto up2m = [](size_t v) { return (v + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1); };
    const size_t o_val = 0;
    const size_t o_col = up2m(o_val + cap * sizeof(T)) + 1 * stagger;
    const size_t o_row = up2m(o_col + cap * sizeof(uint16_t)) + 2 * stagger;
    const size_t o_prod = up2m(o_row + cap * sizeof(uint16_t)) + 3 * stagger;
    const size_t total = o_prod + cap * sizeof(T);
    ctx->dev_malloc(&op->d_pb_arena, total, "propagation-blocked image (values, indices, product buffer)");
    op->pb_arena_bytes = total;
    op->pb_arena_static_bytes = o_prod;  // everything in front of the product buffer is the matrix (copied on re-placement)
    char* base = (char*)op->d_pb_arena;
    op->d_pb_val = base + o_val;
    op->d_pb_col = (uint16_t*)(base + o_col);
    op->d_pb_row = (uint16_t*)(base + o_row);
    op->d_pb_prod = base + o_prod;
  }
  LL_HIP(hipMemsetAsync(op->d_pb_val, 0, cap * sizeof(T), s));  // padding entries: value 0, local indices 0
  LL_HIP(hipMemsetAsync(op->d_pb_col, 0, cap * sizeof(uint16_t), s));
  LL_HIP(hipMemsetAsync(op->d_pb_row, 0, cap * sizeof(uint16_t), s));
  // ---- pass 2 on the device: place the entries
  if (op->rp64)
    hipLaunchKernelGGL((pb_scatter_kernel<T, int64_t>), dim3((int)nrb), dim3(64), hist_bytes, s, m, (int)ncb, (int)nrb,
                       (int)rb_rows, (long long)nr, (const int64_t*)op->d_row_ptr, op->d_col, (const T*)op->d_val,
                       op->d_pb_segq, op->d_pb_segdest, (T*)op->d_pb_val, op->d_pb_col, op->d_pb_row, nullptr, d_skip);
  else
    hipLaunchKernelGGL((pb_scatter_kernel<T, int32_t>), dim3((int)nrb), dim3(64), hist_bytes, s, m, (int)ncb, (int)nrb,
                       (int)rb_rows, (long long)nr, (const int32_t*)op->d_row_ptr, op->d_col, (const T*)op->d_val,
                       op->d_pb_segq, op->d_pb_segdest, (T*)op->d_pb_val, op->d_pb_col, op->d_pb_row, nullptr, d_skip);
  LL_HIP(hipGetLastError());
  // Fixed-point sums need per-row exponents of the absolute row sums and per-block maxima of |x|.  They are built for EVERY
  // image whose row block leaves room for them (2 bytes per row; the CSR arrays they come from may be released after creation),
  // so that ll_op_set_accuracy can move the operator between the two accuracy classes later.
  // the y slice holds 64-bit integers + one 16-bit exponent per row: it must still fit the LDS
  const bool fixed_fits = (size_t)rb_rows * (sizeof(acc_t<T>) + sizeof(int16_t)) + 16 <= (size_t)kPbLdsCap;
  if (op->pb_phase2 == LL_PB_FIXED)
    LL_REQUIRE(fixed_fits, "LL_PB_PHASE2=fixed: row block too large for the LDS (lower LL_PB_ROW_BLOCK)");
  if (fixed_fits) {
    ctx->dev_malloc((void**)&op->d_pb_rexp, std::max<size_t>((size_t)nr, 8) * sizeof(int16_t), "row exponents");
    ctx->dev_malloc((void**)&op->d_pb_blockmax, (size_t)ncb * sizeof(double), "x slice maxima");
    LL_HIP(hipMemsetAsync(op->d_pb_blockmax, 0, (size_t)ncb * sizeof(double), s));
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>(kMaxGrid, (nr + 255) / 256));
    if (op->rp64)
      hipLaunchKernelGGL((pb_rowexp_kernel<T, int64_t>), dim3(g), dim3(256), 0, s, (long long)nr, (const int64_t*)op->d_row_ptr,
                         (const T*)op->d_val, op->d_pb_rexp);
    else
      hipLaunchKernelGGL((pb_rowexp_kernel<T, int32_t>), dim3(g), dim3(256), 0, s, (long long)nr, (const int32_t*)op->d_row_ptr,
                         (const T*)op->d_val, op->d_pb_rexp);
    LL_HIP(hipGetLastError());
  }
  LL_HIP(hipStreamSynchronize(s));  // the host tables above go out of scope
  return true;
}


// ================================================================= 2-D tiled SpMV for matrices with column locality
// (a1/a2/a3 like the kernels above; third candidate of the creation-time timing, LL_SPMV_TILED)
// Propagation blocking pays a 16 B/nnz round trip of the products through memory because, for a matrix WITHOUT column
// locality, the x values a row block needs are spread over the whole vector.  When the columns of a row block fall into
// few column tiles (bands, stencils, lattices — every operator the reference itself ships: sample3_dynamic.cpp:17-22,
// exponentiator_test.cpp:113-121) both slices fit the LDS at once and nothing has to leave the CU:
//   one workgroup per ROW block (y slice = 64-bit fixed-point accumulators in LDS, as in pb_phase2_fixed);
//   it walks the block's NON-EMPTY column tiles (16 KiB of x each, double-buffered in LDS, re-staged from L2 / Infinity
//   Cache: neighbouring row blocks share their tiles) and streams the tile's entries — value + 16-bit local column +
//   16-bit local row = 12 B/nnz for fp64, the algorithmic bytes of CSR — multiplying out of one LDS slice and adding
//   into the other.  No global gather, no product buffer: HBM sees the matrix once, x a few times, y once.
// Pipeline: every lane keeps D trips of entry loads in flight in a static ring (as above); the NEXT tile's x piece
// (16 bytes per lane) is requested one tile ahead and parked in registers; one workgroup barrier per tile.
// Sums: the same order-independent fixed-point scheme as pb_phase2_fixed, with the per-row exponent folded into the
// stored values when the image is built (a power of two: exact) — so the integers that are added are the ones the PB
// kernel adds, the result does not depend on the tile geometry, and the accuracy class is the NORM-wise one stated in
// lanczos_hip.h.  max|x| over the whole vector comes from a small pre-pass (tl_xmax_kernel: one read of x).
constexpr int kTlTileBytes = 16 * 1024;  // one x tile: one 16-byte piece per lane of the workgroup
constexpr int kTlXmaxParts = 512;
constexpr int kTlDepth = 3;
constexpr int kTlSlots = 4;       // x tiles resident in LDS (a ring; power of two)
constexpr int kTlNewPerTrip = 2;  // x tiles a trip may bring in

template <typename T>
__global__ __launch_bounds__(256) void tl_xmax_kernel(long long n, const T* __restrict__ x, double* __restrict__ parts, int aligned) {
  __shared__ double red[4];
  constexpr int V = 16 / (int)sizeof(T);  // elements per 16-byte piece
  constexpr int U = 4;                     // pieces per lane and trip, all requested before the first is used
  double m = 0.0;
  const long long nv = aligned ? n / V : 0;  // whole pieces (the vector is 16-byte aligned)
  const uint4* x4 = reinterpret_cast<const uint4*>(x);
  const long long stride = (long long)gridDim.x * 256;
  for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < nv; i0 += U * stride) {
    uint4 piece[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = i0 + u * stride;
      piece[u] = x4[i < nv ? i : nv - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      T el[V];
      __builtin_memcpy(el, &piece[u], sizeof(uint4));
#pragma unroll
      for (int q = 0; q < V; ++q) m = fmax(m, abs1(el[q]));
    }
  }
  for (long long i = nv * V + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = fmax(m, abs1(x[i]));
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = fmax(m, __shfl_down(m, d, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  // (an Inf in x survives fmax; a NaN is dropped here and reaches the rows it touches through its products)
  if (threadIdx.x == 0) parts[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// ORDERED: component-wise accurate floating-point sums (ll_csr_options.accuracy = LL_ACCURACY_COMPONENTWISE, LL_PB_PHASE2=ordered)
// instead of the fixed-point ones — the y slice holds doubles, every product fl(a_ij 2^-e_i x_j) is added with ds_add_f64, the 16 waves
// of the workgroup in turn (a barrier between turns, as in pb_phase2<ORDERED>): a fixed order, so the result is the same bits on every
// launch, and its error is the reference's own fp64 row loop's, gamma_n sum_j |a_ij x_j| (the row's power-of-two scale is exact).  No
// max|x| pre-pass, no grid scale on the staged x.
template <typename T, int D, bool ALIGNED, bool ORDERED>
__global__ __launch_bounds__(kPbThreads) void tl_spmv_kernel(int rb_rows, int64_t n_local, int64_t n_cols,
                                                             const int32_t* __restrict__ tfirst,   // [nrb + 1]
                                                             const int32_t* __restrict__ tcol,     // [ntiles]
                                                             const int64_t* __restrict__ tquad,    // [ntiles + 1]
                                                             const T* __restrict__ val, const uint4* __restrict__ idx,
                                                             const int16_t* __restrict__ rexp,
                                                             const double* __restrict__ xmax_parts, int n_xmax,
                                                             const T* __restrict__ x, const T* __restrict__ xl,
                                                             T* __restrict__ y, double offset,
                                                             double* __restrict__ dot_partials,
                                                             const double* __restrict__ xnorm2, int xcd_order,
                                                             const int32_t* __restrict__ rb_map, int rb_first, int64_t col0) {
  // Sharded contexts launch the kernel twice: x holds the global columns [col0, n_cols) — the rank's own shard
  // for the row blocks whose tiles are all own-column tiles (they run under the all-gather), the gathered vector (col0 = 0) for
  // the others — and rb_map[rb_first + i] is the i-th row block of the launch.  Single GPU: rb_map = nullptr, col0 = 0.
  constexpr int R = scalar_traits<T>::reals;
  constexpr int C = kTlTileBytes / (int)sizeof(T);  // columns per tile
  constexpr int V = 16 / (int)sizeof(T);            // elements per 16-byte piece
  extern __shared__ double lds_raw[];
  long long* acc = reinterpret_cast<long long*>(lds_raw);                        // [rb_rows * R]
  T* xs = reinterpret_cast<T*>(acc + (size_t)rb_rows * R);                       // [kTlSlots][C]
  unsigned* bad = reinterpret_cast<unsigned*>(xs + kTlSlots * C);                       // [(rb_rows + 31) / 32] rows that met Inf / NaN
  __shared__ double red[kPbWaves + 1];
  const int tid = threadIdx.x;
  // XCD-aware row-block order: workgroups with equal blockIdx % 8 share an XCD and its L2, and neighbouring row blocks share
  // most of their x tiles (94 % for the banded config 3) — so every XCD takes a CONTIGUOUS eighth of the row blocks: the
  // 32 blocks it runs at a time then stage their tiles out of its own L2 instead of each fetching them through the fabric.
  int rb = blockIdx.x;
  if (xcd_order) {
    const int nb = gridDim.x, q = nb / kXcds, r = nb % kXcds;
    const int x = blockIdx.x % kXcds, l = blockIdx.x / kXcds;
    rb = x * q + min(x, r) + l;  // XCD x owns q + (x < r) consecutive blocks; a bijection on [0, nb)
  }
  if (rb_map) rb = rb_map[rb_first + rb];
  const int64_t row0 = (int64_t)rb * rb_rows;
  const int rows = (int)min((int64_t)rb_rows, n_local - row0);
  const int t0 = tfirst[rb], t1 = tfirst[rb + 1];
  const double xs_fac = xnorm2 ? 1.0 / sqrt(*xnorm2) : 1.0;  // unnormalised input (see pb_phase1)
  const long long q_begin = tquad[t0];
  // (the planner works on 32-bit quad numbers RELATIVE to the row block's first quad: its comparisons are scalar instructions;
  // 64-bit compares go through vector-register temporaries)
  const int q_end = (int)(tquad[t1] - q_begin);
  const int q_last = q_end > 0 ? q_end - 1 : 0;  // (the image is padded by one quad behind its end)

  // ---- trips.  A trip is kPbThreads consecutive quads of the row block's stream, WHEREVER the tile boundaries fall: a lane's
  // quad belongs to one tile (tiles are whole quads), the lanes of a trip to up to three consecutive tiles of the block's list.
  // (Round 5's first form ended every trip at the end of its tile: the tiles of the banded config 3 hold 0.75 trips, a quarter of
  // the lane slots of every trip carried no entry.)  Tile number tau of the list lives in slot (tau - t0) % kTlSlots of the x ring
  // in LDS.  A trip brings in at most kTlNewPerTrip new tiles — one 16-byte piece per lane and new tile, requested with the trip's
  // entries and parked in registers until the trip is consumed — and never more than the ring has room for next to the tiles of the
  // trip before it (which slower waves may still be reading): span(k) + new(k + 1) <= kTlSlots consecutive tiles, hence distinct
  // slots; a trip that would need more ends early, at a tile boundary.  ONE barrier per trip (new tiles visible; every wave has left
  // the trip before the previous one).  The planner below is uniform over the workgroup and runs at REQUEST time, D - 1 trips ahead.
  struct Plan {
    int q0, q1;  // quads [q0, q1) of the stream (relative to q_begin)
    int e1, e2;  // ends of tiles a and a + 1: a lane's tile is a + (g >= e1) + (g >= e2)
    int a;       // tile of the trip's first quad
    int first_new, n_new;
  };
  int pq = 0;       // next quad to plan
  int pcur = t0;    // tile that holds pq
  int pb = t0 - 1;  // last tile some earlier trip brought in
  int pspan = 0;    // tiles of the trip before
  auto tq = [&](int t) { return (int)(tquad[t] - q_begin); };
  auto plan = [&]() {
    Plan p;
    p.q0 = pq;
    p.a = pcur;
    p.first_new = pb + 1;
    if (pq >= q_end) {  // beyond the row block: nothing valid (q0 >= q_end ends the loop when it is consumed)
      p.q1 = pq;
      p.e1 = p.e2 = pq;
      p.n_new = 0;
      return p;
    }
    const int cap = min(kTlNewPerTrip, kTlSlots - pspan);  // >= 1: a trip spans at most three tiles
    const int bmax = min(t1 - 1, pb + cap);
    const int end = min(pq + kPbThreads, q_end);
    const int e1 = tq(min(pcur + 1, t1)), e2 = tq(min(pcur + 2, t1)), e3 = tq(min(pcur + 3, t1));
    int b = pcur;
    int eb = e1;
    if (b < bmax && eb < end) {
      ++b;
      eb = e2;
      if (b < bmax && eb < end) {
        ++b;
        eb = e3;
      }
    }
    p.e1 = e1;
    p.e2 = e2;
    p.q1 = min(end, eb);
    p.n_new = max(0, b - pb);
    pspan = b - pcur + 1;
    pb = max(pb, b);
    pq = p.q1;
    pcur = p.q1 == eb ? b + 1 : b;
    return p;
  };
  quad<T> v[D];
  uint4 ix[D], xp[D][kTlNewPerTrip];
  int gq[D];
  Plan pl[D];
  int ctn[D][kTlNewPerTrip];  // column tiles of the (up to) two new tiles of the trip
  // Every trip requests the SAME loads, unconditionally and in straight-line code — the lane's quad of values (2 x 16 B), its packed
  // indices (16 B) and its 16-byte piece of each of the trip's new x tiles (a trip with fewer new tiles requests its last one again:
  // a cache hit) — so the compiler can wait for exactly the oldest trip (s_waitcnt vmcnt(N), see above).
  auto issue = [&](int slot) {
    const Plan p = plan();
    pl[slot] = p;
    const int g = p.q0 + tid;
    const long long gc = q_begin + (g < q_last ? g : q_last);
    gq[slot] = g;
    v[slot] = load_quad<T>(val + 4 * gc);
    ix[slot] = idx[gc];
#pragma unroll
    for (int j = 0; j < kTlNewPerTrip; ++j) {
      // (a row block without tiles reads the table's padding entry; elements beyond the vector's end are never referenced)
      const int tj = max(t0, min(p.first_new + min(j, max(p.n_new - 1, 0)), t1 - 1));
      const int ct = tcol[tj];
      ctn[slot][j] = ct;
      const long long off = (long long)ct * C + (long long)tid * V;
      if constexpr (ALIGNED) {
        // a piece that would reach beyond the end is read V-aligned from the last whole piece position that is still inside
        // (n_cols >= V is guaranteed by the launcher) and shifted into place below (store_piece)
        // (... and never from in front of the buffer: only a re-requested or padding tile can lie there, and it is never stored)
        const long long lim = n_cols - V;
        const long long o = off <= lim ? off : lim;
        xp[slot][j] = *reinterpret_cast<const uint4*>(x + ((o >= col0 ? o : col0) - col0));
      } else {
        T el[V];
#pragma unroll
        for (int q = 0; q < V; ++q) {
          const long long o = off + q < n_cols ? off + q : n_cols - 1;
          el[q] = x[(o >= col0 ? o : col0) - col0];
        }
        __builtin_memcpy(&xp[slot][j], el, sizeof(uint4));
      }
    }
  };
  // (double / complex<double>: the scale 2^kx of the fixed-point grid rides on the staged x elements — one exact scaling per
  // element and tile instead of one v_ldexp_f64 per entry; |kx| <= 1000 and every product below 2^-1022 rounds to the integer 0
  // either way, so the integers are the same.  The float types keep the scaling behind their float product, whose underflow
  // threshold is within reach of the scale.)
  constexpr bool kFoldScale = sizeof(typename scalar_traits<T>::real) == 8 && !ORDERED;
  int kx = 0;  // set below, before the first tile is staged
  auto store_piece = [&](int tile, int ct, uint4 piece) {
    const int buf = (tile - t0) & (kTlSlots - 1);
    T el[V];
    __builtin_memcpy(el, &piece, sizeof(uint4));
    if constexpr (ALIGNED) {
      const long long off = (long long)ct * C + (long long)tid * V;
      const int d = (int)max(0ll, off - (n_cols - V));  // the piece was read d elements further left (only at the vector's end)
      if (d > 0) {
        T sh[V];
#pragma unroll
        for (int q = 0; q < V; ++q) sh[q] = zero<T>();
#pragma unroll
        for (int q = 0; q < V; ++q)
          if (q + d < V) sh[q] = el[q + d];
#pragma unroll
        for (int q = 0; q < V; ++q) el[q] = sh[q];
      }
    }
    if (xnorm2) {
#pragma unroll
      for (int q = 0; q < V; ++q) el[q] = rmul(xs_fac, el[q]);
    }
    if constexpr (kFoldScale) {
#pragma unroll
      for (int q = 0; q < V; ++q) el[q] = scale_pow2(el[q], kx);
    }
    uint4 out;
    __builtin_memcpy(&out, el, sizeof(uint4));
    reinterpret_cast<uint4*>(xs + (size_t)buf * C)[tid] = out;
  };
  // ---- accumulators, scale of the fixed-point grid
  for (int i = tid; i < rb_rows * R; i += kPbThreads) acc[i] = 0;
  for (int i = tid; i < (rb_rows + 31) / 32; i += kPbThreads) bad[i] = 0u;
  int e_x = -2000;  // x == 0: any scale does
  if constexpr (!ORDERED) {
    double m = 0.0;
    for (int i = tid; i < n_xmax; i += kPbThreads) m = fmax(m, xmax_parts[i]);
    const double t = pb_block_max(m, red) * xs_fac;  // (ends with a barrier: the LDS stores above are visible)
    if (t > 0.0 && isfinite(t)) (void)frexp(t, &e_x);
    else if (!(t == 0.0)) e_x = kPbXInf;
    // integer = (PRE-SCALED value * x) * 2^kx; row i's sum is acc_i * 2^(er_i - kx)   (pb_phase2_fixed: k = 62 - (er + e_x + 1))
    kx = e_x == kPbXInf ? 0 : max(-1000, min(1000, 61 - e_x));
  } else {
    __syncthreads();  // the accumulators are zero before the first add
  }
  double* const accd = reinterpret_cast<double*>(acc);  // ORDERED: the same slice as doubles
  const int wave = tid >> 6;

  // (No branch on `valid`: a lane beyond the end of its trip holds a re-read of a valid quad and adds ZERO to that quad's
  // rows — every trip then waits for and uses its loads on every path, which keeps the wait counts of the ring exact.)
  auto consume = [&](const quad<T>& vv, const uint4& ii, bool valid, int buf) {
    const T* xb = xs + (size_t)buf * C;
    const unsigned w4[4] = {ii.x, ii.y, ii.z, ii.w};
    if constexpr (ORDERED) {
      T pe[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) pe[e] = mul(vv.e[e], xb[w4[e] & 0xffffu]);
      for (int w = 0; w < kPbWaves; ++w) {  // the waves add in turn: a fixed order (Inf / NaN travel through the sums by themselves)
        if (wave == w && valid) {
#pragma unroll
          for (int e = 0; e < 4; ++e) lds_add_elem<T>(accd, (int)(w4[e] >> 16), pe[e]);
        }
        __syncthreads();
      }
      return;
    }
    // the four x elements first, in one batch of LDS reads: behind the first atomic the compiler may not move a read of the
    // same LDS array forward (it cannot prove that xs and acc do not overlap), and entry by entry every read's latency is exposed
    T xe[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) xe[e] = xb[w4[e] & 0xffffu];
    unsigned badmask = 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned lr = w4[e] >> 16;
      const T p = mul(vv.e[e], xe[e]);
      double pr[R];
      if constexpr (scalar_traits<T>::is_complex) {
        pr[0] = (double)p.re;
        pr[1] = (double)p.im;
      } else {
        pr[0] = (double)p;
      }
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const double sc = kFoldScale ? pr[q] : ldexp(pr[q], kx);
        const bool good = fabs(sc) < 9.0e18;  // false for Inf / NaN too; fixed_round's value is not used then
        if (!good) badmask |= 1u << e;
        atomicAdd(reinterpret_cast<unsigned long long*>(&acc[R * lr + q]), (valid && good) ? (unsigned long long)fixed_round(sc) : 0ull);
      }
    }
    // rows that met an Inf / NaN (rare: one wave-uniform branch per trip instead of a masked LDS OR per entry)
    if (__builtin_expect(__any(valid && badmask != 0u), 0)) {
      if (valid) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if ((badmask >> e) & 1u) {
            const unsigned lr = w4[e] >> 16;
            atomicOr(&bad[lr >> 5], 1u << (lr & 31));
          }
      }
    }
  };
  // (The first D - 1 trips are requested HERE, right in front of the loop and after everything else of the prologue has
  // drained: the loop is then entered with exactly the pending loads its back edge carries, and the compiler's wait counts
  // stay exact in every phase; requested earlier, the merge of the two paths costs a full drain in the loop's first phase.)
#pragma unroll
  for (int d = 0; d < D - 1; ++d) {
    __builtin_amdgcn_sched_barrier(0);  // keep the trips' loads in trip order (the wait counts of the loop assume it)
    issue(d);
  }
  __builtin_amdgcn_sched_barrier(0);
  auto step = [&](auto ph) -> bool {
    constexpr int PH = decltype(ph)::value;
    const Plan p = pl[PH];
    if (p.q0 >= q_end) return false;  // uniform over the workgroup
    if (p.n_new > 0) store_piece(p.first_new, ctn[PH][0], xp[PH][0]);
    if (p.n_new > 1) store_piece(p.first_new + 1, ctn[PH][1], xp[PH][1]);
    __syncthreads();
    const int g = gq[PH];
    const bool valid = g < p.q1;
    const int buf = (p.a - t0 + (g >= p.e1 ? 1 : 0) + (g >= p.e2 ? 1 : 0)) & (kTlSlots - 1);
    const quad<T> vv = v[PH];
    const uint4 ii = ix[PH];
    issue((PH + D - 1) % D);
    consume(vv, ii, valid, buf);
    return true;
  };
  // (The waitcnt pass merges the pending loads of the prologue and of the back edge conservatively at the loop header: the FIRST
  // trip behind it drains the memory pipeline, one trip in D.  Two rounds of the ring per loop iteration halve that and change
  // nothing measurable — 0.548 against 0.550 ms on the banded config 3, profiles/r05_tl_ring_two_rounds_ab.txt: the other waves
  // of the workgroup cover the drain — so the loop keeps one round.)
  for (;;) {
    if (!step(std::integral_constant<int, 0>{})) break;
    if (!step(std::integral_constant<int, 1>{})) break;
    if constexpr (D > 2) {
      if (!step(std::integral_constant<int, 2>{})) break;
    }
    if constexpr (D > 3) {
      if (!step(std::integral_constant<int, 3>{})) break;
    }
  }
  __syncthreads();
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  auto value = [&](int i, const T&, int er) {  // er: the row's exponent (requested ahead, see the epilogue)
    if constexpr (ORDERED) {  // 2^er restores the row's scale (exact); a row whose absolute sum is not finite has no usable image
      acc_t<T> a;
      if constexpr (scalar_traits<T>::is_complex)
        a = er == 32767 ? zc{nan, nan} : zc{ldexp(accd[2 * i], er), ldexp(accd[2 * i + 1], er)};
      else
        a = er == 32767 ? nan : ldexp(accd[i], er);
      return a;
    }
    const bool unusable = er == 32767 || e_x == kPbXInf || ((bad[i >> 5] >> (i & 31)) & 1u);
    const int back = er - kx;  // 2^back restores the row's scale (empty rows: er = -1100, acc = 0)
    acc_t<T> a;
    if constexpr (scalar_traits<T>::is_complex)
      a = unusable ? zc{nan, nan} : zc{ldexp((double)acc[2 * i], back), ldexp((double)acc[2 * i + 1], back)};
    else
      a = unusable ? nan : ldexp((double)acc[i], back);
    return a;
  };
  pb_phase2_epilogue<T>(rb, row0, rows, xl, y, offset, dot_partials, red, xnorm2, value, [&](int i) { return (int)rexp[row0 + i]; });
}

namespace {
template <typename T> constexpr int tl_cols() { return kTlTileBytes / (int)sizeof(T); }
template <typename T> size_t tl_lds_bytes(int rb_rows) {
  return (size_t)rb_rows * sizeof(acc_t<T>) + (size_t)kTlSlots * kTlTileBytes + (size_t)((rb_rows + 31) / 32) * sizeof(unsigned) + 16;
}
template <typename T> void tl_opt_in_lds() {
  static std::atomic<unsigned long long> mask{0};
  int dev = 0;
  LL_HIP(hipGetDevice(&dev));
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask.load(std::memory_order_acquire) & bit) return;
  pb_opt_in(&tl_spmv_kernel<T, kTlDepth, true, false>);
  pb_opt_in(&tl_spmv_kernel<T, kTlDepth, false, false>);
  pb_opt_in(&tl_spmv_kernel<T, kTlDepth, true, true>);
  pb_opt_in(&tl_spmv_kernel<T, kTlDepth, false, true>);
  mask.fetch_or(bit, std::memory_order_release);
}
}  // namespace

namespace {
// One launch over the row blocks [rb_first, rb_first + rb_count) of the image's order (op.d_tl_rbmap; identity without a map); x holds the
// global columns [col0, col_end), x_local the rank's own rows; the fixed-point class reads n_xmax maxima of |x| from op.d_tl_xmax.
template <typename T>
void tl_launch_rows(const ll_operator& op, int rb_first, int rb_count, const T* x, int64_t col0, int64_t col_end, const T* x_local, T* y,
                    double offset, double* dot_partials, hipStream_t s, const double* xnorm2, int n_xmax) {
  if (rb_count <= 0) return;
  constexpr int64_t V = (int64_t)(16 / sizeof(T));
  // 16-byte pieces of x: the fast form needs an aligned window of at least one piece that starts on a piece boundary
  const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && col0 % V == 0 && col_end - col0 >= V;
#define LL_TL_LAUNCH(AL, ORD)                                                                                                         \
  hipLaunchKernelGGL((tl_spmv_kernel<T, kTlDepth, AL, ORD>), dim3(rb_count), dim3(kPbThreads), tl_lds_bytes<T>(op.tl_rb_rows), s,      \
                     op.tl_rb_rows, op.n_local, col_end, op.d_tl_first, op.d_tl_col, op.d_tl_quad, (const T*)op.d_tl_val,              \
                     (const uint4*)op.d_tl_idx, op.d_tl_rexp, op.d_tl_xmax, n_xmax, x, x_local, y, offset, dot_partials, xnorm2,       \
                     op.ctx->tune.tl_xcd_order ? 1 : 0, (const int32_t*)op.d_tl_rbmap, rb_first, col0)
  if (aligned && op.tl_ordered) LL_TL_LAUNCH(true, true);
  else if (aligned) LL_TL_LAUNCH(true, false);
  else if (op.tl_ordered) LL_TL_LAUNCH(false, true);
  else LL_TL_LAUNCH(false, false);
#undef LL_TL_LAUNCH
  LL_HIP(hipGetLastError());
}
}  // namespace

// x = the WHOLE vector (single GPU; sharded: the gathered vector, x + row_begin the rank's own rows): max |x|, then every row block.
template <typename T>
int launch_spmv_tiled(const ll_operator& op, const T* x, T* y, double offset, double* dot_partials, hipStream_t s,
                      const double* xnorm2) {
  if (op.tl_nrb <= 0) return 0;
  tl_opt_in_lds<T>();
  const int xgrid = (int)std::max<int64_t>(1, std::min<int64_t>(kTlXmaxParts, (op.n + 255) / 256));
  if (!op.tl_ordered)  // (the component-wise form has no fixed-point grid to scale: no max|x| pre-pass)
    hipLaunchKernelGGL((tl_xmax_kernel<T>), dim3(xgrid), dim3(256), 0, s, (long long)op.n, x, op.d_tl_xmax,
                       (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? 1 : 0);
  tl_launch_rows<T>(op, 0, op.tl_nrb, x, 0, op.n, x + op.row_begin, y, offset, dot_partials, s, xnorm2, xgrid);
  return op.tl_nrb;
}

// Sharded contexts (engine.cpp apply_operator): the maximum of |x| over the rank's own shard, one double at
// op.d_tl_xmax[tl_xmax_local_slot()] — the ranks' maxima are then all-gathered into op.d_tl_xmax[0, nranks), the table the kernel
// folds (a maximum does not depend on how the vector is cut, so every partition scales its fixed-point grid exactly like one GPU).
template <typename T> void launch_tl_xmax_local(const ll_operator& op, const T* x_own, hipStream_t s) {
  if (op.tl_nrb <= 0 || op.tl_ordered) return;
  double* parts = op.d_tl_xmax + kTlXmaxParts;
  const int g = (int)std::max<int64_t>(1, std::min<int64_t>(kTlXmaxParts, (op.n_local + 255) / 256));
  hipLaunchKernelGGL((tl_xmax_kernel<T>), dim3(g), dim3(256), 0, s, (long long)op.n_local, x_own, parts,
                     (reinterpret_cast<uintptr_t>(x_own) & 15) == 0 ? 1 : 0);
  hipLaunchKernelGGL((tl_xmax_kernel<double>), dim3(1), dim3(256), 0, s, (long long)g, (const double*)parts, parts + kTlXmaxParts, 1);
  LL_HIP(hipGetLastError());
}
int tl_xmax_local_slot() { return 2 * kTlXmaxParts; }

// pass 0: the row blocks whose tiles are all own-column tiles (x = the rank's shard, global columns [col0, col_end));
// pass 1: the others (x = the gathered vector, col0 = 0).  n_xmax maxima (one per rank) wait in op.d_tl_xmax.
template <typename T>
int launch_spmv_tiled_pass(const ll_operator& op, int pass, const T* x, int64_t col0, int64_t col_end, const T* x_local, T* y,
                           double offset, double* dot_partials, hipStream_t s, const double* xnorm2, int n_xmax) {
  if (op.tl_nrb <= 0) return 0;
  tl_opt_in_lds<T>();
  const int first = pass == 0 ? 0 : op.tl_n_interior;
  const int count = pass == 0 ? op.tl_n_interior : op.tl_nrb - op.tl_n_interior;
  tl_launch_rows<T>(op, first, count, x, col0, col_end, x_local, y, offset, dot_partials, s, xnorm2, n_xmax);
  return op.tl_nrb;
}

// Build the tiled image on the device from the operator's CSR arrays.  false: the matrix is not eligible — its row
// blocks touch too many column tiles (re-staging x would cost more than the matrix stream itself: matrices without
// column locality, which keep PB) or the shape does not fit the tables — and nothing stays allocated.
template <typename T> bool tl_build_device(ll_operator* op) {
  ll_context* ctx = op->ctx;
  hipStream_t s = ctx->stream;
  const int64_t nr = op->n_local;
  if (nr <= 0 || op->nnz <= 0 || ctx->nranks > kTlXmaxParts) return false;
  const Tuning& tune = ctx->tune;
  constexpr int C = tl_cols<T>();
  // ---- row blocks: as long as the LDS allows (the longer the block, the smaller the share of re-staged x per entry),
  //      in whole rounds of 256 workgroups
  const int64_t row_max = std::min<int64_t>(65536, ((int64_t)kPbLdsCap - kTlSlots * kTlTileBytes - 2048 - 64) / (int64_t)sizeof(acc_t<T>) / 256 * 256);
  int64_t rounds = std::max<int64_t>(1, (nr + 256 * row_max - 1) / (256 * row_max));
  int64_t rb_rows = std::min<int64_t>(row_max, std::max<int64_t>(16, (nr + 256 * rounds - 1) / (256 * rounds)));
  if (tune.pb_block > 0) rb_rows = std::min<int64_t>(row_max, std::max(4, tune.pb_block));
  if (tune.pb_row_block > 0) rb_rows = std::min<int64_t>(row_max, std::max(4, tune.pb_row_block));
  rb_rows = (rb_rows + 1) & ~(int64_t)1;  // even: the x buffers behind the accumulators stay 16-byte aligned
  const int64_t nrb = (nr + rb_rows - 1) / rb_rows;
  const int64_t ncb = (op->n + C - 1) / C;
  if (ncb * (int64_t)sizeof(int) > 60 * 1024) return false;   // LDS histogram of the build kernels
  if (ncb * nrb > (int64_t)24 << 20) return false;            // count table
  PbColMap m;
  std::memset(&m, 0, sizeof(m));
  m.shard = op->n;
  m.nranks = 1;
  m.rank = 0;
  m.nchunks = 1;
  m.cstart[0] = 0;
  m.clen[0] = (int)std::min<int64_t>(op->n, 0x7fffffff);
  m.bl[0] = C;
  m.nb[0] = (int)ncb;
  m.own_base[0] = 0;
  m.rem_base[0] = (int)ncb;
  // ---- entries per (column tile, row block)
  int32_t* d_cnt = nullptr;
  ctx->dev_malloc((void**)&d_cnt, (size_t)ncb * nrb * sizeof(int32_t), "tile counts");
  struct Free1 {
    void* p;
    ~Free1() {
      if (p) (void)hipFree(p);
    }
  } free_cnt{d_cnt};
  const size_t hist_bytes = (size_t)ncb * sizeof(int);
  if (op->rp64)
    hipLaunchKernelGGL((pb_count_kernel<int64_t>), dim3((int)nrb), dim3(256), hist_bytes, s, m, (int)ncb, (int)nrb,
                       (int)rb_rows, (long long)nr, (const int64_t*)op->d_row_ptr, op->d_col, d_cnt);
  else
    hipLaunchKernelGGL((pb_count_kernel<int32_t>), dim3((int)nrb), dim3(256), hist_bytes, s, m, (int)ncb, (int)nrb,
                       (int)rb_rows, (long long)nr, (const int32_t*)op->d_row_ptr, op->d_col, d_cnt);
  LL_HIP(hipGetLastError());
  std::vector<int32_t> cnt32((size_t)ncb * nrb);
  LL_HIP(hipMemcpyAsync(cnt32.data(), d_cnt, cnt32.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  LL_HIP(hipStreamSynchronize(s));
  // ---- tile list (non-empty tiles only), row block by row block; every tile padded to 16 entries (whole quads, the
  //      stream of a tile starts on a 128-byte line of both arrays)
  const int64_t pad = 16;
  std::vector<int32_t> tfirst((size_t)nrb + 1), tcol;
  std::vector<int64_t> tquad, segq((size_t)ncb * (nrb + 1), 0);
  // Order of a row block's tiles (the sums are integers: any order gives the same bits).  Neighbouring row blocks of a banded
  // matrix share most of their column tiles, and with one contiguous eighth of the row blocks per XCD (tl_xcd_order) they
  // run at the same time behind the same L2: walking the tiles by column index MODULO the longest tile list makes every
  // row block reach the tile of residue t at about step t, so a staged x slice is fetched from the fabric once per group of
  // row blocks that share it rather than once per row block (LL_TL_WALK=0: ascending column index; A/B in DESIGN.md §3.1).
  int64_t period = 1;
  for (int64_t r = 0; r < nrb; ++r) {
    int64_t k = 0;
    for (int64_t c = 0; c < ncb; ++c) k += cnt32[(size_t)c * nrb + r] != 0;
    period = std::max(period, k);
  }
  if (!tune.tl_walk_modulo) period = ncb;
  std::vector<int32_t> order;
  order.reserve((size_t)ncb);
  for (int64_t c0 = 0; c0 < std::min(period, ncb); ++c0)
    for (int64_t c = c0; c < ncb; c += period) order.push_back((int32_t)c);
  int64_t q = 0;
  for (int64_t r = 0; r < nrb; ++r) {
    tfirst[(size_t)r] = (int32_t)tcol.size();
    for (const int32_t c : order) {
      const int64_t k = cnt32[(size_t)c * nrb + r];
      segq[(size_t)c * (nrb + 1) + r] = q;
      if (k == 0) continue;
      tcol.push_back(c);
      tquad.push_back(q >> 2);
      q += (k + pad - 1) / pad * pad;
    }
  }
  tfirst[(size_t)nrb] = (int32_t)tcol.size();
  tquad.push_back(q >> 2);
  // Sharded contexts: the row blocks whose tiles all lie inside the rank's OWN columns come first — their launch needs nothing from
  // the other ranks and runs under the all-gather; the rest follows behind it (banded matrices: the blocks next to the shard's ends).
  std::vector<int32_t> rbmap;
  int n_interior = 0;
  if (ctx->nranks > 1) {
    std::vector<int32_t> rest;
    for (int64_t r = 0; r < nrb; ++r) {
      bool own = true;
      for (int32_t t = tfirst[(size_t)r]; t < tfirst[(size_t)r + 1] && own; ++t) {
        const int64_t c0 = (int64_t)tcol[(size_t)t] * C, c1 = std::min<int64_t>(c0 + C, op->n);
        own = c0 >= op->row_begin && c1 <= op->row_begin + nr;
      }
      (own ? rbmap : rest).push_back((int32_t)r);
    }
    n_interior = (int)rbmap.size();
    rbmap.insert(rbmap.end(), rest.begin(), rest.end());
  }
  const int64_t entries = q;
  const int64_t ntiles = (int64_t)tcol.size();
  tcol.push_back(tcol.empty() ? 0 : tcol.back());  // padding entry: what a row block without tiles reads (and ignores)
  // ---- eligibility: the re-staged x slices must cost less than the matrix stream, and the padding must stay small
  const double staged = (double)ntiles * kTlTileBytes, stream = (double)entries * (sizeof(T) + 4);
  // (measured at twice the stream — an eighth of the banded config 3, row blocks of 4 883 rows: 0.109 ms against CSR-stream's 0.099 ms)
  const bool eligible = staged <= stream && (double)entries <= 1.25 * (double)op->nnz + 16.0 * (double)nrb;
  if (ntiles == 0 || !(eligible || tune.tl_force)) return false;
  if (ntiles > 0x7ffffff0) return false;
  op->tl_nrb = (int)nrb;
  op->tl_rb_rows = (int)rb_rows;
  op->tl_ncb = (int)ncb;
  op->tl_entries = entries;
  op->tl_tiles = ntiles;
  op->tl_n_interior = n_interior;
  auto up = [&](void** dst, const void* src, size_t bytes) {
    ctx->dev_malloc(dst, std::max<size_t>(bytes, 16), "tiled-image tables");
    LL_HIP(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, s));
  };
  int64_t* d_segq = nullptr;
  struct Free2 {
    int64_t*& p;
    ~Free2() {
      if (p) (void)hipFree(p);
    }
  } free_segq{d_segq};
  try {
    up((void**)&op->d_tl_first, tfirst.data(), tfirst.size() * sizeof(int32_t));
    up((void**)&op->d_tl_col, tcol.data(), tcol.size() * sizeof(int32_t));
    up((void**)&op->d_tl_quad, tquad.data(), tquad.size() * sizeof(int64_t));
    up((void**)&d_segq, segq.data(), segq.size() * sizeof(int64_t));
    if (!rbmap.empty()) up((void**)&op->d_tl_rbmap, rbmap.data(), rbmap.size() * sizeof(int32_t));
    const size_t cap = (size_t)entries + 16;  // one quad behind the image: clamped reads of lanes beyond the end
    ctx->dev_malloc(&op->d_tl_val, cap * sizeof(T), "tiled image (values)");
    ctx->dev_malloc((void**)&op->d_tl_idx, cap * sizeof(uint32_t), "tiled image (local indices)");
    ctx->dev_malloc((void**)&op->d_tl_rexp, std::max<size_t>((size_t)nr, 8) * sizeof(int16_t), "row exponents");
    // [0, parts): what the kernel folds; [parts, 2 parts): the own shard's partial maxima; [2 parts]: the own shard's maximum
    ctx->dev_malloc((void**)&op->d_tl_xmax, (size_t)(2 * kTlXmaxParts + 8) * sizeof(double), "x maxima");
    LL_HIP(hipMemsetAsync(op->d_tl_val, 0, cap * sizeof(T), s));  // padding entries: value 0, local indices 0
    LL_HIP(hipMemsetAsync(op->d_tl_idx, 0, cap * sizeof(uint32_t), s));
    LL_HIP(hipMemsetAsync(op->d_tl_xmax, 0, (size_t)(2 * kTlXmaxParts + 8) * sizeof(double), s));
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>(kMaxGrid, (nr + 255) / 256));
    if (op->rp64)
      hipLaunchKernelGGL((pb_rowexp_kernel<T, int64_t>), dim3(g), dim3(256), 0, s, (long long)nr, (const int64_t*)op->d_row_ptr,
                         (const T*)op->d_val, op->d_tl_rexp);
    else
      hipLaunchKernelGGL((pb_rowexp_kernel<T, int32_t>), dim3(g), dim3(256), 0, s, (long long)nr, (const int32_t*)op->d_row_ptr,
                         (const T*)op->d_val, op->d_tl_rexp);
    LL_HIP(hipGetLastError());
    if (op->rp64)
      hipLaunchKernelGGL((pb_scatter_kernel<T, int64_t, true>), dim3((int)nrb), dim3(64), hist_bytes, s, m, (int)ncb, (int)nrb,
                         (int)rb_rows, (long long)nr, (const int64_t*)op->d_row_ptr, op->d_col, (const T*)op->d_val, d_segq,
                         nullptr, (T*)op->d_tl_val, reinterpret_cast<uint16_t*>(op->d_tl_idx), nullptr, op->d_tl_rexp);
    else
      hipLaunchKernelGGL((pb_scatter_kernel<T, int32_t, true>), dim3((int)nrb), dim3(64), hist_bytes, s, m, (int)ncb, (int)nrb,
                         (int)rb_rows, (long long)nr, (const int32_t*)op->d_row_ptr, op->d_col, (const T*)op->d_val, d_segq,
                         nullptr, (T*)op->d_tl_val, reinterpret_cast<uint16_t*>(op->d_tl_idx), nullptr, op->d_tl_rexp);
    LL_HIP(hipGetLastError());
    LL_HIP(hipStreamSynchronize(s));  // the host tables above go out of scope
  } catch (...) {
    tl_release(op);
    throw;
  }
  return true;
}

void tl_release(ll_operator* op) {
  auto drop = [](auto*& p) {
    if (p) (void)hipFree((void*)p);
    p = nullptr;
  };
  drop(op->d_tl_first);
  drop(op->d_tl_col);
  drop(op->d_tl_quad);
  drop(op->d_tl_val);
  drop(op->d_tl_idx);
  drop(op->d_tl_rexp);
  drop(op->d_tl_xmax);
  drop(op->d_tl_rbmap);
  op->tl_nrb = 0;
  op->tl_n_interior = 0;
}

#define LL_INST_PB(T)                                                                                              \
  template void launch_pb_phase1<T>(const ll_operator&, int, int, const T*, hipStream_t, const double*);            \
  template int launch_pb_phase2<T>(const ll_operator&, const T*, T*, double, double*, hipStream_t, const double*);  \
  template int launch_spmv_pb<T>(const ll_operator&, const T*, const T*, const T*, T*, double, double*, hipStream_t, \
                                 const double*);                                                                    \
  template bool pb_build_device<T>(ll_operator*);                                                                   \
  template bool tl_build_device<T>(ll_operator*);                                                                   \
  template int launch_spmv_tiled<T>(const ll_operator&, const T*, T*, double, double*, hipStream_t, const double*);  \
  template void launch_tl_xmax_local<T>(const ll_operator&, const T*, hipStream_t);                                 \
  template int launch_spmv_tiled_pass<T>(const ll_operator&, int, const T*, int64_t, int64_t, const T*, T*, double, \
                                         double*, hipStream_t, const double*, int);                                 \
  template void csr_check_device<T>(ll_operator*);
LL_INST_PB(double) LL_INST_PB(zc) LL_INST_PB(float) LL_INST_PB(cf)

}  // namespace ll
