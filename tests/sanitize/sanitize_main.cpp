// AddressSanitizer + UndefinedBehaviorSanitizer leg for the HOST code of the hot path (the
// reference's tests build with -fsanitize=address, /root/reference/test/CMakeLists.txt:3).  CPU build only — GPU
// sanitizers are not available on the pool.  One executable links
//   * csrc/tridiag_host.cpp   (the product's tridiagonal solver, bisection, inverse iteration),
//   * csrc/generators.cpp     (the synthetic-matrix generators bench.py and the tests use),
//   * oracle/lanczos_oracle.cpp (the CPU checker: whole Lanczos / Exponentiator / Taylor runs),
// all compiled with -fsanitize=address,undefined -fno-sanitize-recover=all, and drives them on small problems,
// including the degenerate shapes the reference tests (1x1, near-singular off-diagonals, empty shards).
// Any sanitizer report aborts; a clean run prints "sanitize ok".
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace ll {
int64_t tridiag_qr(int64_t m, const double* alpha, const double* beta, double* ev, double* q);
void tridiag_inverse_iteration(int64_t m, const double* alpha, const double* beta, int64_t nw, const double* lambdas,
                               double* out);
double tridiag_bisect(int64_t m, const double* alpha, const double* beta, int64_t k);
}  // namespace ll

extern "C" {
void llgen_start_vector_d(uint64_t seed, int64_t row_begin, int64_t n_local, double* v);
void llgen_start_vector_z(uint64_t seed, int64_t row_begin, int64_t n_local, double* v_reim);
int64_t llgen_laplace2d_count(int64_t N, int64_t row_begin, int64_t n_local);
void llgen_laplace2d_fill(int64_t N, int64_t row_begin, int64_t n_local, int64_t* rp, int32_t* ci, double* va);
int64_t llgen_randsym_count(int64_t n, int64_t band, int64_t row_begin, int64_t n_local);
void llgen_randsym_fill(int64_t n, int64_t band, int64_t row_begin, int64_t n_local, int64_t* rp, int32_t* ci, double* va);
void llgen_torus_fill(int64_t N, int64_t row_begin, int64_t n_local, int64_t* rp, int32_t* ci, double* va_reim);

struct oracle_params {
  int64_t matrix_size, max_iteration;
  double eps;
  int32_t find_maximum, full_orthogonalize;
  int64_t num_eigs;
  double eigenvalue_offset;
  int64_t num_eigs_per_iteration;
};
struct oracle_trace { double* alpha; double* beta; int64_t* len; double* t_mv; double* t_total; };
int64_t oracle_lanczos_run_d(const int64_t*, const int32_t*, const double*, const oracle_params*, const double*, double*,
                             double*, int64_t*, int64_t*, oracle_trace*);
int64_t oracle_expo_run_z(const int64_t*, const int32_t*, const std::complex<double>*, const oracle_params*, double,
                          double, const std::complex<double>*, std::complex<double>*, oracle_trace*);
int64_t oracle_taylor_run_z(const int64_t*, const int32_t*, const std::complex<double>*, const oracle_params*, double,
                            double, const std::complex<double>*, std::complex<double>*);
}

#define REQUIRE(cond)                                                          \
  do {                                                                         \
    if (!(cond)) {                                                             \
      std::fprintf(stderr, "sanitize driver: %s failed (line %d)\n", #cond, __LINE__); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

int main() {
  // ---- tridiagonal solver: sizes 1 .. 200, the reference's known answer (T1:758-768) and a near-singular case
  {
    const double a3[3] = {1, 2, 3}, b3[2] = {2, 2};
    double ev[3], q[9];
    REQUIRE(ll::tridiag_qr(3, a3, b3, ev, q) == 0);
    REQUIRE(std::fabs(ev[0] + 1) < 1e-12 && std::fabs(ev[1] - 2) < 1e-12 && std::fabs(ev[2] - 5) < 1e-12);
    for (int k = 0; k < 3; ++k) REQUIRE(std::fabs(ll::tridiag_bisect(3, a3, b3, k) - ev[k]) < 1e-10);
    double one_a[1] = {7.5}, one_ev[1], one_q[1];
    REQUIRE(ll::tridiag_qr(1, one_a, nullptr, one_ev, one_q) == 0 && one_ev[0] == 7.5);
    for (int m : {2, 5, 17, 64, 65, 200}) {
      std::vector<double> al(m), be(m), e(m), qq((size_t)m * m), out((size_t)3 * m);
      for (int i = 0; i < m; ++i) {
        al[i] = std::sin(0.37 * i) * 3.0;
        be[i] = (i % 7 == 3) ? 1e-300 : 0.5 + std::cos(0.11 * i);  // some vanishing couplings (breakdown shapes)
      }
      ll::tridiag_qr(m, al.data(), be.data(), e.data(), qq.data());
      for (int i = 1; i < m; ++i) REQUIRE(e[i] >= e[i - 1]);
      ll::tridiag_qr(m, al.data(), be.data(), e.data(), nullptr);
      const int nw = m < 3 ? m : 3;
      ll::tridiag_inverse_iteration(m, al.data(), be.data(), nw, e.data(), out.data());
      REQUIRE(std::fabs(ll::tridiag_bisect(m, al.data(), be.data(), m - 1) - e[m - 1]) < 1e-9 * (1 + std::fabs(e[m - 1])));
    }
  }
  // ---- generators: whole matrices and shards (including an empty shard), structural checks
  std::vector<int64_t> rp;
  std::vector<int32_t> ci;
  std::vector<double> va;
  const int64_t n = 777;
  {
    const int64_t nnz = llgen_randsym_count(n, 0, 0, n);
    rp.resize(n + 1), ci.resize(nnz), va.resize(nnz);
    llgen_randsym_fill(n, 0, 0, n, rp.data(), ci.data(), va.data());
    REQUIRE(rp[n] == nnz && nnz == 15 * n);
    for (int64_t p = 0; p < nnz; ++p) REQUIRE(ci[p] >= 0 && ci[p] < n);
    // a shard in the middle and an empty one at the end
    for (auto sh : {std::pair<int64_t, int64_t>{300, 211}, {777, 0}}) {
      const int64_t c = llgen_randsym_count(n, 40, sh.first, sh.second);
      std::vector<int64_t> r2(sh.second + 1);
      std::vector<int32_t> c2(c > 0 ? c : 1);
      std::vector<double> v2(c > 0 ? c : 1);
      llgen_randsym_fill(n, 40, sh.first, sh.second, r2.data(), c2.data(), v2.data());
      REQUIRE(r2[sh.second] == c);
    }
    const int64_t N = 13, ln = llgen_laplace2d_count(N, 0, N * N);
    std::vector<int64_t> r3(N * N + 1);
    std::vector<int32_t> c3(ln);
    std::vector<double> v3(ln);
    llgen_laplace2d_fill(N, 0, N * N, r3.data(), c3.data(), v3.data());
    REQUIRE(r3[N * N] == ln && ln == 5 * N * N - 4 * N);
  }
  // ---- oracle: a Lanczos run to convergence with a restart pass (two roots) on the random matrix
  {
    std::vector<double> init(n), vals(2), vecs(2 * n), al(n), be(n);
    llgen_start_vector_d(1, 0, n, init.data());
    oracle_params p{n, n, 2.220446049250313e-13, 1, 0, 2, 0.0, 5};
    std::vector<int64_t> counts(64);
    int64_t npass = 0, len = 0;
    double t_mv = 0, t_tot = 0;
    oracle_trace tr{al.data(), be.data(), &len, &t_mv, &t_tot};
    const int64_t found = oracle_lanczos_run_d(rp.data(), ci.data(), va.data(), &p, init.data(), vals.data(), vecs.data(),
                                               counts.data(), &npass, &tr);
    REQUIRE(found == 2 && vals[0] >= vals[1] && npass >= 1 && len >= 1);
  }
  // ---- oracle: Exponentiator + Taylor on the complex torus
  {
    const int64_t N = 9, nt = N * N, tnnz = 5 * nt;  // 5 entries per row
    std::vector<int64_t> rt(nt + 1);
    std::vector<int32_t> ct(tnnz);
    std::vector<std::complex<double>> vt(tnnz), in(nt), out(nt), out2(nt);
    llgen_torus_fill(N, 0, nt, rt.data(), ct.data(), reinterpret_cast<double*>(vt.data()));
    llgen_start_vector_z(1, 0, nt, reinterpret_cast<double*>(in.data()));
    oracle_params p{nt, nt, 2.220446049250313e-14, 0, 1, 1, 0.0, 5};
    const int64_t it = oracle_expo_run_z(rt.data(), ct.data(), vt.data(), &p, 0.0, -0.5, in.data(), out.data(), nullptr);
    const int64_t terms = oracle_taylor_run_z(rt.data(), ct.data(), vt.data(), &p, 0.0, -0.5, in.data(), out2.data());
    REQUIRE(it >= 2 && terms >= 2);
    double diff = 0, nrm = 0;
    for (int64_t i = 0; i < nt; ++i) diff += std::norm(out[i] - out2[i]), nrm += std::norm(out[i]);
    // the Lanczos form stops on the overlap of successive approximations (EX:154), the Taylor form on the size of the
    // last term (EX:187-195): they agree to ~1e-8 here; this is a memory/UB check, parity lives in the other tests
    REQUIRE(diff <= 1e-12 * nrm);
  }
  std::printf("sanitize ok\n");
  return 0;
}
