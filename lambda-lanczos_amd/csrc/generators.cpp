// Host-only synthetic inputs for bench.py / tests (SURVEY.md 8d "Concrete synthetic inputs").  Not part of the hot
// path and not an oracle: this is workload synthesis (the reference ships no matrices beyond its inline test
// lambdas).  Everything is a pure function of (index, seed) through splitmix64 so that the CPU baseline, the GPU
// path and every rank of a sharded run see bit-identical inputs without exchanging data.
//
// The numpy twin lives in lambda-lanczos_amd/generators.py (used for small sizes and to cross-check this file).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

inline uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline double u01(uint64_t x) { return (double)(splitmix64(x) >> 11) * (1.0 / 9007199254740992.0); }

constexpr int kOut = 7;  // out-entries per row of B

// column of entry j of row i of B
inline int64_t b_col(int64_t n, int64_t i, int j, int64_t band) {
  const uint64_t h = splitmix64((uint64_t)(64 * i + j));
  if (band <= 0) {  // uniform over the other n-1 columns
    int64_t c = (int64_t)(h % (uint64_t)(n - 1));
    if (c >= i) ++c;
    return c;
  }
  // banded: within +-band of i (never i), wrapped into [0,n)
  const int64_t off = (int64_t)(h % (uint64_t)(2 * band));  // 0 .. 2*band-1
  int64_t d = off - band;                                      // -band .. band-1
  if (d >= 0) ++d;                                             // skip 0 -> -band..-1, 1..band
  int64_t c = (i + d) % n;
  if (c < 0) c += n;
  return c;
}
inline double b_val(int64_t i, int j) { return 2.0 * u01((uint64_t)(64 * i + j + 32)) - 1.0; }

struct Ent {
  int32_t c;
  double v;
};

}  // namespace

extern "C" {

uint64_t llgen_splitmix64(uint64_t x) { return splitmix64(x); }

// Start vector (all configs, through the init_vector hook): v_i = 2*u01(seed*2^40 + i) - 1 for global index i.
void llgen_start_vector_d(uint64_t seed, int64_t row_begin, int64_t n_local, double* v) {
  const uint64_t base = seed << 40;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_local; ++i) v[i] = 2.0 * u01(base + (uint64_t)(row_begin + i)) - 1.0;
}
// complex: re from 2i, im from 2i+1
void llgen_start_vector_z(uint64_t seed, int64_t row_begin, int64_t n_local, double* v_reim) {
  const uint64_t base = seed << 40;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_local; ++i) {
    const uint64_t g = (uint64_t)(row_begin + i);
    v_reim[2 * i] = 2.0 * u01(base + 2 * g) - 1.0;
    v_reim[2 * i + 1] = 2.0 * u01(base + 2 * g + 1) - 1.0;
  }
}

// ---- C2: 5-point Dirichlet Laplacian on an N x N grid, row-major i = y*N + x, diagonal 4, neighbours -1,
// columns sorted.  Rows [row_begin, row_begin + n_local).  nnz(total) = 5 N^2 - 4 N.
int64_t llgen_laplace2d_count(int64_t N, int64_t row_begin, int64_t n_local) {
  int64_t nnz = 0;
  for (int64_t r = row_begin; r < row_begin + n_local; ++r) {
    const int64_t y = r / N, x = r % N;
    nnz += 1 + (y > 0) + (x > 0) + (x + 1 < N) + (y + 1 < N);
  }
  return nnz;
}
void llgen_laplace2d_fill(int64_t N, int64_t row_begin, int64_t n_local, int64_t* rp, int32_t* ci, double* va) {
  int64_t p = 0;
  rp[0] = 0;
  for (int64_t k = 0; k < n_local; ++k) {
    const int64_t r = row_begin + k, y = r / N, x = r % N;
    if (y > 0) { ci[p] = (int32_t)(r - N); va[p++] = -1.0; }
    if (x > 0) { ci[p] = (int32_t)(r - 1); va[p++] = -1.0; }
    ci[p] = (int32_t)r; va[p++] = 4.0;
    if (x + 1 < N) { ci[p] = (int32_t)(r + 1); va[p++] = -1.0; }
    if (y + 1 < N) { ci[p] = (int32_t)(r + N); va[p++] = -1.0; }
    rp[k + 1] = p;
  }
}

// ---- C3/C4: A = B + B^T + 7 I with 7 random out-entries per row of B (duplicates kept as separate entries),
// exactly 15 n stored entries.  band <= 0: uniformly random columns; band > 0: columns within +-band of the row.
// Two-call protocol: count -> caller allocates -> fill.  Rows [row_begin, row_begin+n_local); entries of a row are
// sorted by column (ties: B entries by j, then B^T entries by source row, then j).
// The transpose part needs every row's out-entries, so both calls scan all n rows (O(n) memory for counts).
static void incoming_counts(int64_t n, int64_t band, int64_t row_begin, int64_t n_local, std::vector<int32_t>& cnt) {
  cnt.assign((size_t)n_local, 0);
  for (int64_t r = 0; r < n; ++r)
    for (int j = 0; j < kOut; ++j) {
      const int64_t c = b_col(n, r, j, band);
      if (c >= row_begin && c < row_begin + n_local) ++cnt[(size_t)(c - row_begin)];
    }
}
int64_t llgen_randsym_count(int64_t n, int64_t band, int64_t row_begin, int64_t n_local) {
  std::vector<int32_t> cnt;
  incoming_counts(n, band, row_begin, n_local, cnt);
  int64_t nnz = 0;
  for (int64_t i = 0; i < n_local; ++i) nnz += kOut + 1 + cnt[(size_t)i];
  return nnz;
}
void llgen_randsym_fill(int64_t n, int64_t band, int64_t row_begin, int64_t n_local, int64_t* rp, int32_t* ci,
                        double* va) {
  std::vector<int32_t> cnt;
  incoming_counts(n, band, row_begin, n_local, cnt);
  rp[0] = 0;
  for (int64_t i = 0; i < n_local; ++i) rp[i + 1] = rp[i] + kOut + 1 + cnt[(size_t)i];
  std::vector<int64_t> cur((size_t)n_local);
  // own entries + diagonal
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < n_local; ++k) {
    const int64_t i = row_begin + k;
    int64_t p = rp[k];
    for (int j = 0; j < kOut; ++j) {
      ci[p] = (int32_t)b_col(n, i, j, band);
      va[p++] = b_val(i, j);
    }
    ci[p] = (int32_t)i;
    va[p++] = 7.0;
    cur[(size_t)k] = p;
  }
  // transpose entries, in (source row, j) order
  for (int64_t r = 0; r < n; ++r)
    for (int j = 0; j < kOut; ++j) {
      const int64_t c = b_col(n, r, j, band);
      if (c >= row_begin && c < row_begin + n_local) {
        const int64_t p = cur[(size_t)(c - row_begin)]++;
        ci[p] = (int32_t)r;
        va[p] = b_val(r, j);
      }
    }
  // stable sort of every row by column
#pragma omp parallel
  {
    std::vector<Ent> tmp;
#pragma omp for schedule(static)
    for (int64_t k = 0; k < n_local; ++k) {
      const int64_t a = rp[k], b = rp[k + 1];
      tmp.resize((size_t)(b - a));
      for (int64_t p = a; p < b; ++p) tmp[(size_t)(p - a)] = Ent{ci[p], va[p]};
      std::stable_sort(tmp.begin(), tmp.end(), [](const Ent& x, const Ent& y) { return x.c < y.c; });
      for (int64_t p = a; p < b; ++p) { ci[p] = tmp[(size_t)(p - a)].c; va[p] = tmp[(size_t)(p - a)].v; }
    }
  }
}

// ---- C5: complex Hermitian tight-binding torus N x N, i = y*N + x: on-site real u01(i) - 0.5, hops to x+1 / x-1
// carry -exp(+i phi y) / -exp(-i phi y) with phi = 2 pi 3 / N, hops to y+-1 are -1.  5 entries per row, sorted.
void llgen_torus_fill(int64_t N, int64_t row_begin, int64_t n_local, int64_t* rp, int32_t* ci, double* va_reim) {
  const double phi = 2.0 * M_PI * 3.0 / (double)N;
  rp[0] = 0;
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < n_local; ++k) {
    const int64_t r = row_begin + k, y = r / N, x = r % N;
    struct E { int64_t c; double re, im; } e[5];
    e[0] = {y * N + (x + 1) % N, -std::cos(phi * (double)y), -std::sin(phi * (double)y)};
    e[1] = {y * N + (x + N - 1) % N, -std::cos(phi * (double)y), std::sin(phi * (double)y)};
    e[2] = {((y + 1) % N) * N + x, -1.0, 0.0};
    e[3] = {((y + N - 1) % N) * N + x, -1.0, 0.0};
    e[4] = {r, u01((uint64_t)r) - 0.5, 0.0};
    std::sort(e, e + 5, [](const E& a, const E& b) { return a.c < b.c; });
    for (int t = 0; t < 5; ++t) {
      const int64_t p = 5 * k + t;
      ci[p] = (int32_t)e[t].c;
      va_reim[2 * p] = e[t].re;
      va_reim[2 * p + 1] = e[t].im;
    }
  }
  for (int64_t k = 0; k < n_local; ++k) rp[k + 1] = 5 * (k + 1);
}

}  // extern "C"
