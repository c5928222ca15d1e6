// The `lambda_lanczos::util::` / `tridiagonal_impl::` / `VectorRandomInitializer` seam: user sources and the reference's own
// tests call these helpers next to the engines (test/lambda_lanczos_test.cpp:54,75-87,99,107,122,302,765 and
// test/exponentiator_test.cpp:21,66,130 of mrcdr/lambda-lanczos).  This file knows the REFERENCE's include lines and names only;
// it is compiled once with -I include/compat (this repository's facade) and once with the real reference's include directory,
// and must pass both ways.  Host code only: it needs no device.
#include <lambda_lanczos/exponentiator.hpp>
#include <lambda_lanczos/lambda_lanczos.hpp>

#include <cmath>
#include <complex>
#include <cstdio>
#include <functional>
#include <random>
#include <string>
#include <vector>

namespace util = lambda_lanczos::util;
using std::complex;
using std::vector;

static int failures = 0;
static void check(bool ok, const char* what) {
  if (!ok) {
    ++failures;
    std::printf("FAILED: %s\n", what);
  }
}

int main() {
  // ---- inner_prod conjugates its first argument: <(3, 1+3i), (3, 2+4i)> = 23 - 2i (the reference's pin)
  std::printf("[case] inner_prod / typed_conj convention\n");
  {
    const vector<complex<double>> v1{3.0, complex<double>(1.0, 3.0)}, v2{3.0, complex<double>(2.0, 4.0)};
    const complex<double> r = util::inner_prod(v1, v2);
    check(r.real() == 23.0 && r.imag() == -2.0, "<v1|v2> = 23 - 2i");
    check(util::typed_conj(complex<double>(1.0, 3.0)) == complex<double>(1.0, -3.0), "typed_conj(complex)");
    check(util::typed_conj(2.5) == 2.5, "typed_conj(real) is the identity");
    const vector<double> a{1.0, 2.0, 3.0}, b{4.0, -5.0, 6.0};
    check(util::inner_prod(a, b) == 12.0, "real inner product");
    util::real_t<complex<float>> one = 1.0f;  // real_t maps complex<R> to R
    check(sizeof(one) == sizeof(float), "real_t<complex<float>> is float");
  }

  // ---- m_norm: sum |Re| + |Im| (= 6 for the reference's pin), norm, scalar_mul, normalize
  std::printf("[case] m_norm / norm / scalar_mul / normalize\n");
  {
    const vector<complex<double>> v{complex<double>(1.0, 3.0), complex<double>(-1.0, -1.0)};
    check(util::m_norm(v) == 6.0, "m_norm = 6");
    const vector<double> r{-1.5, 2.0, -0.5};
    check(util::m_norm(r) == 4.0, "real m_norm");
    vector<double> w{3.0, 4.0};
    check(util::norm(w) == 5.0, "norm (3,4) = 5");
    util::scalar_mul(2.0, w);
    check(w[0] == 6.0 && w[1] == 8.0, "scalar_mul");
    util::normalize(w);
    check(std::fabs(w[0] - 0.6) < 1e-15 && std::fabs(w[1] - 0.8) < 1e-15 && std::fabs(util::norm(w) - 1.0) < 1e-15, "normalize");
    vector<complex<double>> z{complex<double>(0.0, 2.0), complex<double>(0.0, 0.0)};
    util::normalize(z);
    check(std::abs(z[0] - complex<double>(0.0, 1.0)) < 1e-15, "normalize keeps the phase");
  }

  // ---- schmidt_orth: the reference's own experiment (n = 10, five complex vectors), <v|u> vanishes afterwards
  std::printf("[case] schmidt_orth against an orthonormal set\n");
  {
    const size_t n = 10;
    std::mt19937 eng(1);
    std::uniform_real_distribution<double> dist(-10.0, 10.0);
    vector<vector<complex<double>>> us;
    for (size_t k = 0; k < n / 2; ++k) {
      vector<complex<double>> u(n);
      for (auto& e : u) e = complex<double>(dist(eng), dist(eng));
      util::schmidt_orth(u, us.begin(), us.end());
      util::normalize(u);
      us.push_back(u);
    }
    vector<complex<double>> v(n);
    for (auto& e : v) e = complex<double>(dist(eng), dist(eng));
    util::schmidt_orth(v, us.begin(), us.end());
    for (const auto& u : us) {
      const complex<double> ip = util::inner_prod(v, u);
      check(std::fabs(ip.real()) <= 1e-15 * n * 20 && std::fabs(ip.imag()) <= 1e-15 * n * 20, "<v|u_k> = 0 after schmidt_orth");
    }
    check(util::norm(v) > 1.0, "something is left of v");
  }

  // ---- sort_eigenpairs: ascending by default, a predicate for descending, vectors follow on request
  std::printf("[case] sort_eigenpairs\n");
  {
    vector<double> vals{2, -1, 0};
    vector<vector<complex<double>>> vecs{{2, 2, 2}, {0, 0, 0}, {1, 1, 1}};
    util::sort_eigenpairs(vals, vecs, true);
    check(vals[0] == -1 && vals[1] == 0 && vals[2] == 2, "values ascending");
    check(vecs[0][0].real() == 0 && vecs[1][0].real() == 1 && vecs[2][0].real() == 2, "vectors moved along");
    util::sort_eigenpairs<complex<double>>(vals, vecs, false, std::greater<double>());
    check(vals[0] == 2 && vals[2] == -1, "descending with a predicate");
    check(vecs[0][0].real() == 0, "vectors untouched when not requested");
  }

  // ---- vectorToString, sgn, initAsIdentity
  std::printf("[case] vectorToString / sgn / initAsIdentity\n");
  {
    check(util::vectorToString(vector<double>{1, 2, 3}) == std::string("1 2 3"), "vectorToString default delimiter");
    check(util::vectorToString(vector<int>{4, 5}, ",") == std::string("4,5"), "vectorToString with a delimiter");
    check(util::vectorToString(vector<double>{}) == std::string(""), "vectorToString of an empty vector");
    check(util::sgn(0.0) == 1.0 && util::sgn(-2.0) == -1.0 && util::sgn(3.0) == 1.0, "sgn(0) = +1");
    vector<vector<double>> id;
    util::initAsIdentity(id, 3);
    check(id.size() == 3 && id[1].size() == 3 && id[1][1] == 1.0 && id[1][2] == 0.0, "initAsIdentity");
  }

  // ---- tridiagonal_eigenpairs: alpha = {1,2,3}, beta = {2,2} -> {-1, 2, 5} (the reference's known answer)
  std::printf("[case] tridiagonal_impl::tridiagonal_eigenpairs / tridiagonal_eigenvalues\n");
  {
    const vector<double> alpha{1, 2, 3}, beta{2, 2};
    vector<double> ev;
    vector<vector<double>> q;
    const size_t unconverged = lambda_lanczos::tridiagonal_impl::tridiagonal_eigenpairs(alpha, beta, ev, q);
    check(unconverged == 0 && ev.size() == 3 && q.size() == 3, "three pairs, no forced break");
    const double want[3] = {-1.0, 2.0, 5.0};
    const double wantv[3][3] = {{2, -2, 1}, {2, 1, -2}, {1, 2, 2}};  // unnormalised eigenvectors
    for (int j = 0; j < 3 && ev.size() == 3 && q.size() == 3; ++j) {
      check(std::fabs(ev[j] - want[j]) < 1e-13, "eigenvalue of the 3 x 3 tridiagonal");
      vector<double> v(wantv[j], wantv[j] + 3);
      util::normalize(v);
      const double s = util::inner_prod(v, q[j]) < 0 ? -1.0 : 1.0;
      for (int i = 0; i < 3; ++i) check(std::fabs(s * q[j][i] - v[i]) < 1e-13, "eigenvector of the 3 x 3 tridiagonal");
    }
    vector<double> only;
    check(lambda_lanczos::tridiagonal_impl::tridiagonal_eigenvalues(alpha, beta, only) == 0 && only.size() == 3 &&
              std::fabs(only[0] + 1.0) < 1e-13 && std::fabs(only[2] - 5.0) < 1e-13,
          "tridiagonal_eigenvalues");
  }

  // ---- VectorRandomInitializer: the default of init_vector, under its public name
  std::printf("[case] VectorRandomInitializer\n");
  {
    vector<double> v(64, 9.0);
    lambda_lanczos::VectorRandomInitializer<double>::init(v);
    bool in_range = true, moved = false;
    for (double e : v) {
      in_range = in_range && e >= -1.0 && e <= 1.0;
      moved = moved || e != 9.0;
    }
    check(in_range && moved, "real elements uniform in [-1, 1]");
    vector<complex<double>> z(64);
    lambda_lanczos::VectorRandomInitializer<complex<double>>::init(z);
    bool ok = true, has_imag = false;
    for (const auto& e : z) {
      ok = ok && std::fabs(e.real()) <= 1.0 && std::fabs(e.imag()) <= 1.0;
      has_imag = has_imag || e.imag() != 0.0;
    }
    check(ok && has_imag, "complex elements: both parts in [-1, 1]");
  }

  if (failures == 0) std::printf("PASSED\n");
  return failures == 0 ? 0 : 1;
}
