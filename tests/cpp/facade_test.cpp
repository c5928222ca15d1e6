// The reference's own tests, re-typed against the drop-in facade (include/lambda_lanczos_hip/): they read like
// test/lambda_lanczos_test.cpp / test/exponentiator_test.cpp of mrcdr/lambda-lanczos (same engines, same public-field
// idiom, same expectations and tolerances) and run on the GPU through liblanczos_hip.so.
// Built and executed by tests/test_gpu_cpp_facade.py; exits non-zero on the first failed expectation.
#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include <lambda_lanczos_hip/exponentiator.hpp>
#include <lambda_lanczos_hip/lambda_lanczos.hpp>

using lambda_lanczos::LambdaLanczos;
template <typename T> using vector = std::vector<T>;
template <typename T> using complex = std::complex<T>;

static int g_failures = 0;
#define EXPECT_NEAR(expected, actual, tol)                                                                \
  do {                                                                                                    \
    const double e_ = (expected), a_ = (actual), t_ = (tol);                                             \
    if (!(std::abs(e_ - a_) <= t_)) {                                                                     \
      std::printf("  FAIL %s:%d  expected %.17g got %.17g (tol %.3g)\n", __FILE__, __LINE__, e_, a_, t_); \
      ++g_failures;                                                                                       \
    }                                                                                                     \
  } while (0)
#define EXPECT_EQ(expected, actual)                                                                \
  do {                                                                                             \
    if (!((expected) == (actual))) {                                                               \
      std::printf("  FAIL %s:%d  %s != %s\n", __FILE__, __LINE__, #expected, #actual);             \
      ++g_failures;                                                                                \
    }                                                                                              \
  } while (0)
#define TEST(suite, name) static void suite##_##name()
#define RUN(suite, name)                    \
  do {                                      \
    std::printf("[ RUN ] %s.%s\n", #suite, #name); \
    suite##_##name();                       \
  } while (0)

// T1:25-45 — fixed-seed start vector
template <typename T> void vector_initializer(vector<T>& v);
template <> void vector_initializer(vector<double>& v) {
  std::mt19937 mt(1);
  std::uniform_real_distribution<double> rand(-1.0, 1.0);
  for (auto& e : v) e = rand(mt);
}
template <> void vector_initializer(vector<complex<double>>& v) {
  std::mt19937 mt(1);
  std::uniform_real_distribution<double> rand(-1.0, 1.0);
  for (auto& e : v) {
    double a = rand(mt), b = rand(mt);
    e = complex<double>(a, b);
  }
}

TEST(DIAGONALIZE_TEST, SIMPLE_MATRIX) {  // T1:128-161
  const size_t n = 3;
  double matrix[n][n] = {{2.0, 1.0, 1.0}, {1.0, 2.0, 1.0}, {1.0, 1.0, 2.0}};
  auto matmul = [&](const vector<double>& in, vector<double>& out) {
    for (size_t i = 0; i < n; ++i)
      for (size_t j = 0; j < n; ++j) out[i] += matrix[i][j] * in[j];
  };
  LambdaLanczos<double> engine(matmul, n, true, 1);
  engine.init_vector = vector_initializer<double>;
  engine.eigenvalue_offset = 6.0;
  vector<double> eigvalues;
  vector<vector<double>> eigvecs;
  engine.run(eigvalues, eigvecs);
  double eigvalue = eigvalues[0];
  auto& eigvec = eigvecs[0];
  auto sign = eigvec[0] / std::abs(eigvec[0]);
  vector<double> correct_eigvec{sign / std::sqrt(3.0), sign / std::sqrt(3.0), sign / std::sqrt(3.0)};
  double correct_eigvalue = 4.0;
  EXPECT_NEAR(correct_eigvalue, eigvalue, std::abs(correct_eigvalue * engine.eps));
  for (size_t i = 0; i < n; ++i) EXPECT_NEAR(correct_eigvec[i], eigvec[i], std::abs(correct_eigvalue * engine.eps * 10));
  EXPECT_EQ(size_t(1), engine.getIterationCounts().size());
}

TEST(DIAGONALIZE_TEST, SIMPLE_MATRIX_MULTIPLE_VALUE_RETURN_FEATURE) {  // T1:231-260 (C++17 structured binding)
  const size_t n = 3;
  double matrix[n][n] = {{2.0, 1.0, 1.0}, {1.0, 2.0, 1.0}, {1.0, 1.0, 2.0}};
  auto matmul = [&](const vector<double>& in, vector<double>& out) {
    for (size_t i = 0; i < n; ++i)
      for (size_t j = 0; j < n; ++j) out[i] += matrix[i][j] * in[j];
  };
  LambdaLanczos<double> engine(matmul, n, true, 1);
  engine.eigenvalue_offset = 6.0;
  auto [eigvalues, eigvecs] = engine.run();
  EXPECT_NEAR(4.0, eigvalues[0], 4.0 * engine.eps);
  EXPECT_NEAR(1.0 / std::sqrt(3.0), std::abs(eigvecs[0][1]), 4.0 * engine.eps * 10);
}

TEST(DIAGONALIZE_TEST, HERMITIAN_MATRIX) {  // T1:375-409
  const size_t n = 3;
  const auto I_ = complex<double>(0.0, 1.0);
  complex<double> matrix[n][n] = {{0.0, I_, 1.0}, {-I_, 0.0, I_}, {1.0, -I_, 0.0}};
  auto matmul = [&](const vector<complex<double>>& in, vector<complex<double>>& out) {
    for (size_t i = 0; i < n; ++i)
      for (size_t j = 0; j < n; ++j) out[i] += matrix[i][j] * in[j];
  };
  LambdaLanczos<complex<double>> engine(matmul, n, false, 1);
  engine.init_vector = vector_initializer<complex<double>>;
  double eigvalue;
  vector<complex<double>> eigvec(n);
  engine.run(eigvalue, eigvec);
  vector<complex<double>> correct_eigvec{1.0, I_, -1.0};
  auto phase_factor = std::polar(1.0, std::arg(eigvec[0]));
  for (auto& c : correct_eigvec) c *= phase_factor / std::sqrt(3.0);
  double correct_eigvalue = -2.0;
  EXPECT_NEAR(correct_eigvalue, eigvalue, std::abs(correct_eigvalue * engine.eps));
  for (size_t i = 0; i < n; ++i) {
    EXPECT_NEAR(correct_eigvec[i].real(), eigvec[i].real(), std::abs(correct_eigvalue * engine.eps * 10));
    EXPECT_NEAR(correct_eigvec[i].imag(), eigvec[i].imag(), std::abs(correct_eigvalue * engine.eps * 10));
  }
}

TEST(DIAGONALIZE_TEST, MULTIPLE_EIGENPAIRS) {  // T1:442-488
  const int n = 8;
  const size_t nroot = 3;
  double matrix[n][n] = {{6, -3, -3, 0, -1, 1, -1, 1},  {-3, -4, 2, 2, -1, -5, 0, -4}, {-3, 2, 2, -3, 0, 0, -1, -1},
                         {0, 2, -3, 0, -3, 3, 2, 2},    {-1, -1, 0, -3, -2, 0, -5, -4}, {1, -5, 0, 3, 0, -4, 5, 0},
                         {-1, 0, -1, 2, -5, 5, -4, 4},  {1, -4, -1, 2, -4, 0, 4, 2}};
  auto mv_mul = [&](const vector<double>& in, vector<double>& out) {
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) out[i] += matrix[i][j] * in[j];
  };
  LambdaLanczos<double> engine(mv_mul, n, false, 1);
  engine.num_eigs = nroot;
  engine.eps = 1e-7;
  vector<double> eigenvalues;
  vector<vector<double>> eigenvectors;
  engine.run(eigenvalues, eigenvectors);
  const double correct_eigvals[3] = {-13.21508597, -8.50033154, -4.26674892};
  double correct_eigvecs[3][n] = {
      {0.02081752, -0.49222707, 0.13202088, 0.24048092, 0.15089223, -0.60850056, 0.48079787, -0.24043829},
      {0.16645991, 0.51818471, -0.00646562, -0.09493495, 0.60595718, 0.02042567, 0.52346924, 0.23043415},
      {0.03381669, -0.07999997, 0.32090331, 0.61650970, 0.41812886, -0.01782613, -0.45571810, 0.35575946}};
  EXPECT_EQ(nroot, eigenvalues.size());
  for (size_t iroot = 0; iroot < nroot; ++iroot) {
    EXPECT_NEAR(correct_eigvals[iroot], eigenvalues[iroot], std::abs(correct_eigvals[iroot] * engine.eps));
    auto sign = eigenvectors[iroot][0] / std::abs(eigenvectors[iroot][0]);
    for (int i = 0; i < n; ++i)
      EXPECT_NEAR(correct_eigvecs[iroot][i] * sign, eigenvectors[iroot][i], std::abs(correct_eigvals[iroot] * engine.eps * 10));
  }
}

TEST(DIAGONALIZE_TEST, DEVICE_CSR_OPERATOR) {  // the device-resident operator form (SURVEY 8b "Operator contract")
  const int64_t N = 40, n = N * N;  // 5-point Laplacian, analytic spectrum
  vector<int64_t> rp{0};
  vector<int32_t> ci;
  vector<double> va;
  for (int64_t r = 0; r < n; ++r) {
    const int64_t y = r / N, x = r % N;
    if (y > 0) { ci.push_back((int32_t)(r - N)); va.push_back(-1.0); }
    if (x > 0) { ci.push_back((int32_t)(r - 1)); va.push_back(-1.0); }
    ci.push_back((int32_t)r); va.push_back(4.0);
    if (x + 1 < N) { ci.push_back((int32_t)(r + 1)); va.push_back(-1.0); }
    if (y + 1 < N) { ci.push_back((int32_t)(r + N)); va.push_back(-1.0); }
    rp.push_back((int64_t)ci.size());
  }
  lambda_lanczos::CsrMatrix<double> A(rp, ci, va);
  LambdaLanczos<double> engine(A, (size_t)n, false, 1);
  engine.init_vector = vector_initializer<double>;
  engine.eigenvalue_offset = -8.0;
  double eigvalue;
  vector<double> eigvec;
  engine.run(eigvalue, eigvec);
  const double correct = 4.0 - 4.0 * std::cos(M_PI / (N + 1));
  EXPECT_NEAR(correct, eigvalue, 8.0 * engine.eps * 10);
  EXPECT_EQ((size_t)n, eigvec.size());
}

TEST(EXPONENTIATOR_TEST, EXPONENTIATE_LARGE_MATRIX) {  // T2:106-162
  const size_t n = 100;
  const double t = -1.0;
  auto mv_mul = [&](const vector<complex<double>>& in, vector<complex<double>>& out) {
    for (size_t i = 0; i < n - 1; ++i) {
      out[i] += t * in[i + 1];
      out[i + 1] += t * in[i];
    }
    out[0] += t * in[n - 1];
    out[n - 1] += t * in[0];
  };
  complex<double> a(0.0, 3.0);
  lambda_lanczos::Exponentiator<complex<double>> exponentiator(mv_mul, n);
  vector<complex<double>> input(n);
  input[0] = complex<double>(1, 2);
  input[n - 1] = complex<double>(1, 2);
  input[n / 2] = complex<double>(8, 2);
  double nrm = 0;
  for (auto& c : input) nrm += std::norm(c);
  for (auto& c : input) c /= std::sqrt(nrm);
  vector<complex<double>> output;  // left unsized on purpose (T2:131)
  size_t itern = exponentiator.run(a, input, output);
  // analytic plane waves (T2:83-104)
  vector<complex<double>> exact(n);
  const complex<double> I_(0.0, 1.0);
  for (size_t j = 0; j < n; ++j) {
    const double k = 2 * M_PI / n * j, ev = 2 * t * std::cos(k);
    complex<double> proj = 0;
    for (size_t i = 0; i < n; ++i) proj += std::conj(std::exp(I_ * k * (double)i) / std::sqrt((double)n)) * input[i];
    for (size_t i = 0; i < n; ++i) exact[i] += std::exp(I_ * k * (double)i) / std::sqrt((double)n) * std::exp(a * ev) * proj;
  }
  complex<double> ov = 0;
  double ne = 0, no = 0;
  for (size_t i = 0; i < n; ++i) { ov += std::conj(exact[i]) * output[i]; ne += std::norm(exact[i]); no += std::norm(output[i]); }
  EXPECT_NEAR(1.0, std::abs(ov) / std::sqrt(ne * no), exponentiator.eps * 10);
  EXPECT_EQ(size_t(19), itern);  // what the reference reports for this input (tests/golden/exponentiator.json)
  vector<complex<double>> tout;
  size_t terms = exponentiator.taylor_run(a, input, tout);
  EXPECT_EQ(size_t(37), terms);
}

int main() {
  try {
    RUN(DIAGONALIZE_TEST, SIMPLE_MATRIX);
    RUN(DIAGONALIZE_TEST, SIMPLE_MATRIX_MULTIPLE_VALUE_RETURN_FEATURE);
    RUN(DIAGONALIZE_TEST, HERMITIAN_MATRIX);
    RUN(DIAGONALIZE_TEST, MULTIPLE_EIGENPAIRS);
    RUN(DIAGONALIZE_TEST, DEVICE_CSR_OPERATOR);
    RUN(EXPONENTIATOR_TEST, EXPONENTIATE_LARGE_MATRIX);
  } catch (const std::exception& e) {
    std::printf("EXCEPTION: %s\n", e.what());
    return 2;
  }
  std::printf("%s (%d failed expectations)\n", g_failures ? "FAILED" : "PASSED", g_failures);
  return g_failures ? 1 : 0;
}
