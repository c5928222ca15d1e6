// Literal drop-in check: this file is written against the REFERENCE's include paths and API only
// (README.md:20-21 of mrcdr/lambda-lanczos: <lambda_lanczos/lambda_lanczos.hpp>, <lambda_lanczos/exponentiator.hpp>,
// lambda_lanczos::LambdaLanczos / Exponentiator, std::function mv_mul, run overloads).  It is compiled with
//   -I include/compat
// in place of the reference's include directory; nothing here names this repository's headers or types.
#include <lambda_lanczos/exponentiator.hpp>
#include <lambda_lanczos/lambda_lanczos.hpp>

#include <cmath>
#include <complex>
#include <cstdio>
#include <exception>
#include <random>
#include <vector>

using lambda_lanczos::Exponentiator;
using lambda_lanczos::LambdaLanczos;

static int failures = 0;
static void check(bool ok, const char* what) {
  if (!ok) {
    ++failures;
    std::printf("FAILED: %s\n", what);
  }
}

// a seeded start vector through the reference's public init_vector hook (lambda_lanczos.hpp:133), like the reference's
// own tests do: the checks below do not depend on what std::random_device returns on the test machine
static void seeded_start(std::vector<double>& v) {
  std::mt19937 mt(1);
  std::uniform_real_distribution<double> rand(-1.0, 1.0);
  for (double& e : v) e = rand(mt);
}

int main() {
  try {
    // ---- 1. the 3 x 3 matrix of the reference's first known-answer test (eigenvalues 4, 1, 1), a user lambda as mv_mul
    std::printf("[case] dense 3x3 through a std::function mv_mul, largest eigenpair\n");
    const std::vector<std::vector<double>> a = {{2, 1, 1}, {1, 2, 1}, {1, 1, 2}};
    auto dense = [&a](const std::vector<double>& in, std::vector<double>& out) {
      for (size_t r = 0; r < a.size(); ++r)
        for (size_t c = 0; c < a.size(); ++c) out[r] += a[r][c] * in[c];
    };
    LambdaLanczos<double> engine(dense, 3, true, 1);
    engine.init_vector = seeded_start;
    std::vector<double> values;
    std::vector<std::vector<double>> vectors;
    engine.run(values, vectors);
    check(values.size() == 1 && std::fabs(values[0] - 4.0) < 1e-12, "largest eigenvalue is 4");
    check(vectors.size() == 1 && vectors[0].size() == 3, "one eigenvector of length 3");
    const double s = vectors[0][0] < 0 ? -1.0 : 1.0;
    for (double x : vectors[0]) check(std::fabs(s * x - 1.0 / std::sqrt(3.0)) < 1e-10, "eigenvector (1,1,1)/sqrt3");
    check(engine.getIterationCounts().size() == 1, "one pass recorded");

    // ---- 2. the tuple-returning overload and the public tuning fields, smallest pair of an open chain
    std::printf("[case] run() returning a tuple, find_maximum = false, eigenvalue_offset\n");
    const int n = 10;
    auto chain = [n](const std::vector<double>& in, std::vector<double>& out) {
      for (int i = 0; i < n; ++i) {
        if (i > 0) out[i] -= in[i - 1];
        if (i + 1 < n) out[i] -= in[i + 1];
      }
    };
    LambdaLanczos<double> low(chain, n, false, 1);
    low.eigenvalue_offset = -4.0;
    low.eps = 1e-14;
    low.init_vector = seeded_start;
    auto result = low.run();
    const double pi = std::acos(-1.0);
    check(std::fabs(std::get<0>(result)[0] + 2.0 * std::cos(pi / (n + 1))) < 1e-12, "lowest level of the open chain");

    // ---- 3. the single-pair overload
    std::printf("[case] run(eigenvalue, eigenvector)\n");
    double top = 0.0;
    std::vector<double> top_vec;
    LambdaLanczos<double> single(dense, 3, true, 1);
    single.init_vector = seeded_start;
    single.run(top, top_vec);
    check(std::fabs(top - 4.0) < 1e-12 && top_vec.size() == 3, "single-pair overload");

    // ---- 4. Exponentiator: exp(a A) v on a periodic chain with a = i, norm conservation and a = 0 identity
    std::printf("[case] Exponentiator<std::complex<double>>::run on a ring, anti-Hermitian exponent\n");
    typedef std::complex<double> cplx;
    const int m = 16;
    auto ring = [m](const std::vector<cplx>& in, std::vector<cplx>& out) {
      for (int i = 0; i < m; ++i) out[i] += -in[(i + 1) % m] - in[(i + m - 1) % m];
    };
    Exponentiator<cplx> evolve(ring, m);
    std::vector<cplx> psi(m, cplx(0.0, 0.0)), next;
    psi[0] = cplx(1.0, 0.0);
    const size_t steps = evolve.run(cplx(0.0, 1.0), psi, next);
    double norm2 = 0.0;
    for (const cplx& z : next) norm2 += std::norm(z);
    check(steps >= 2 && next.size() == (size_t)m && std::fabs(norm2 - 1.0) < 1e-12, "unitary step keeps the norm");
    std::vector<cplx> same;
    evolve.run(cplx(0.0, 0.0), psi, same);
    check(same.size() == (size_t)m && std::abs(same[0] - psi[0]) < 1e-14 && std::abs(same[5]) < 1e-14, "a = 0 is the identity");
  } catch (const std::exception& e) {
    std::printf("device error: %s\n", e.what());
    return 2;
  }
  std::printf(failures ? "FAILED (%d)\n" : "PASSED\n", failures);
  return failures ? 1 : 0;
}
