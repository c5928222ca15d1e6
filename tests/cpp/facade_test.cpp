// Drop-in check of the C++ facade (include/lambda_lanczos_hip/): user code written against the reference's API —
// constructor (mv_mul, n, find_maximum, num_eigs), public data members, run() overloads, getIterationCounts(),
// Exponentiator(mv_mul, n).run()/taylor_run() — compiles unchanged against the facade and produces the reference's
// known answers on the GPU.  The PROBLEMS (matrices, expected eigenpairs) are the known-answer data of the reference's
// test suite (cited per case); the harness is ours: one table of cases, one generic checker.
// Built and executed by tests/test_cpp_facade.py; exit code 0 = all expectations met.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>

#include <lambda_lanczos_hip/exponentiator.hpp>
#include <lambda_lanczos_hip/lambda_lanczos.hpp>

namespace ll = lambda_lanczos;  // the alias the facade installs: existing `lambda_lanczos::` code keeps compiling
typedef std::complex<double> cplx;

static int failures = 0;
static void expect(bool ok, const std::string& what) {
  if (!ok) {
    std::printf("  FAIL: %s\n", what.c_str());
    ++failures;
  }
}
static void expect_close(double want, double got, double tol, const std::string& what) {
  if (!(std::abs(want - got) <= tol)) {
    std::printf("  FAIL: %s: want %.16g got %.16g (tol %.3g)\n", what.c_str(), want, got, tol);
    ++failures;
  }
}

// A dense operator in the reference's callback form: out += M in, `out` arrives zero-filled.
template <typename T> std::function<void(const std::vector<T>&, std::vector<T>&)> dense(const std::vector<std::vector<T>>& m) {
  return [m](const std::vector<T>& in, std::vector<T>& out) {
    for (size_t i = 0; i < m.size(); ++i)
      for (size_t j = 0; j < m.size(); ++j) out[i] += m[i][j] * in[j];
  };
}

// phase-insensitive distance between unit vectors
template <typename T> double misfit(const std::vector<T>& a, const std::vector<T>& b) {
  std::complex<double> ov = 0;
  for (size_t i = 0; i < a.size(); ++i) ov += std::conj(std::complex<double>(a[i])) * std::complex<double>(b[i]);
  return std::abs(1.0 - std::abs(ov));
}

// Deterministic start vectors, like the reference's own tests (lambda_lanczos_test.cpp:25-45 seed mt19937 with 1): the
// known-answer cases below must not depend on what std::random_device returns on the test box.
template <typename T> struct SeededInit {
  static void fill(std::vector<T>& v, unsigned seed) {
    std::mt19937 mt(seed);
    std::uniform_real_distribution<T> rand((T)(-1.0), (T)(1.0));
    for (auto& e : v) e = rand(mt);
  }
};
template <typename R> struct SeededInit<std::complex<R>> {
  static void fill(std::vector<std::complex<R>>& v, unsigned seed) {
    std::mt19937 mt(seed);
    std::uniform_real_distribution<R> rand((R)(-1.0), (R)(1.0));
    for (auto& e : v) {
      const R re = rand(mt), im = rand(mt);
      e = std::complex<R>(re, im);
    }
  }
};
template <typename T> std::function<void(std::vector<T>&)> seeded_init(unsigned seed = 1) {
  return [seed](std::vector<T>& v) { SeededInit<T>::fill(v, seed); };
}

template <typename T> struct EigCase {
  const char* name;                       // reference test this data comes from
  std::vector<std::vector<T>> matrix;
  bool find_maximum;
  size_t num_eigs;
  double offset, eps;                     // eps <= 0: keep the default
  std::vector<double> values;
  std::vector<std::vector<T>> vectors;    // unit vectors, any phase
};

template <typename T> void run_case(const EigCase<T>& c) {
  std::printf("[case] %s\n", c.name);
  const size_t n = c.matrix.size();
  ll::LambdaLanczos<T> engine(dense<T>(c.matrix), n, c.find_maximum, 1);
  engine.num_eigs = c.num_eigs;           // public data members, as in the reference's tests
  engine.eigenvalue_offset = c.offset;
  if (c.eps > 0) engine.eps = c.eps;
  engine.init_vector = seeded_init<T>(1);
  std::vector<ll::util::real_t<T>> values;   // real_t<T>, as in the reference (lambda_lanczos.hpp:330)
  std::vector<std::vector<T>> vectors;
  engine.run(values, vectors);            // outputs are sized by the library
  expect(values.size() == c.num_eigs && vectors.size() == c.num_eigs, "result count");
  for (size_t r = 0; r < c.values.size() && r < values.size(); ++r) {
    expect_close(c.values[r], values[r], std::max(std::abs(c.values[r]) * (double)engine.eps, 1e-8 * (engine.eps > 1e-8)), "eigenvalue");
    expect(vectors[r].size() == n, "eigenvector length");
    expect(misfit(c.vectors[r], vectors[r]) <= std::max(100.0 * (double)engine.eps, 1e-12), "eigenvector direction");
  }
  expect(!engine.getIterationCounts().empty(), "iteration counts recorded");
  if (c.num_eigs == 1) expect(engine.getIterationCounts().size() == 1, "one pass for one eigenpair");
}

static void eigen_cases() {
  const double s3 = 1.0 / std::sqrt(3.0);
  // 3x3 all-ones + identity: eigenvalues {4,1,1} (lambda_lanczos_test.cpp:128-161, README sample)
  run_case<double>({"T1:128 SIMPLE_MATRIX", {{2, 1, 1}, {1, 2, 1}, {1, 1, 2}}, true, 1, 6.0, -1, {4.0}, {{s3, s3, s3}}});
  run_case<cplx>({"T1:310 SIMPLE_MATRIX_USE_COMPLEX_TYPE", {{2, 1, 1}, {1, 2, 1}, {1, 1, 2}}, true, 1, 0.0, -1, {4.0}, {{s3, s3, s3}}});
  // the same matrix in single precision: default eps = 1e3 * FLT_EPSILON (lambda_lanczos_test.cpp:163-193)
  const float f3 = 1.0f / std::sqrt(3.0f);
  run_case<float>({"T1:163 SIMPLE_MATRIX_FLOAT", {{2, 1, 1}, {1, 2, 1}, {1, 1, 2}}, true, 1, 0.0, -1, {4.0}, {{f3, f3, f3}}});
  run_case<std::complex<float>>({"complex<float> Hermitian 3x3 (data of T1:375)",
                                 {{0, {0, 1}, 1}, {{0, -1}, 0, {0, 1}}, {1, {0, -1}, 0}}, false, 1, 0.0, -1, {-2.0},
                                 {{f3, {0, f3}, -f3}}});
  const cplx I(0, 1);
  run_case<cplx>({"T1:375 HERMITIAN_MATRIX", {{0, I, 1}, {-I, 0, I}, {1, -I, 0}}, false, 1, 0.0, -1, {-2.0},
                  {{s3, I * s3, -s3}}});
  run_case<double>({"T1:411 SINGLE_ELEMENT_MATRIX", {{2}}, true, 1, 0.0, -1, {2.0}, {{1.0}}});
  run_case<double>({"T1:442 MULTIPLE_EIGENPAIRS",
                    {{6, -3, -3, 0, -1, 1, -1, 1}, {-3, -4, 2, 2, -1, -5, 0, -4}, {-3, 2, 2, -3, 0, 0, -1, -1},
                     {0, 2, -3, 0, -3, 3, 2, 2}, {-1, -1, 0, -3, -2, 0, -5, -4}, {1, -5, 0, 3, 0, -4, 5, 0},
                     {-1, 0, -1, 2, -5, 5, -4, 4}, {1, -4, -1, 2, -4, 0, 4, 2}},
                    false, 3, 0.0, 1e-7, {-13.21508597, -8.50033154, -4.26674892},
                    {{0.02081752, -0.49222707, 0.13202088, 0.24048092, 0.15089223, -0.60850056, 0.48079787, -0.24043829},
                     {0.16645991, 0.51818471, -0.00646562, -0.09493495, 0.60595718, 0.02042567, 0.52346924, 0.23043415},
                     {0.03381669, -0.07999997, 0.32090331, 0.61650970, 0.41812886, -0.01782613, -0.45571810, 0.35575946}}});
}

static void api_shapes() {
  std::printf("[case] run() overloads, init_vector hook, num_eigs restored (lambda_lanczos.hpp:376-407)\n");
  std::vector<std::vector<double>> m = {{2, 1, 1}, {1, 2, 1}, {1, 1, 2}};
  ll::LambdaLanczos<double> engine(dense<double>(m), 3, true, 2);
  int hook_calls = 0;
  engine.init_vector = [&hook_calls](std::vector<double>& v) {  // reference hook signature (lambda_lanczos.hpp:133)
    ++hook_calls;
    for (size_t i = 0; i < v.size(); ++i) v[i] = 1.0 + 0.25 * (double)i;
  };
  double value = 0;
  std::vector<double> vec;
  engine.run(value, vec);                                       // single pair regardless of num_eigs
  expect_close(4.0, value, 4.0 * engine.eps, "single-pair overload");
  expect(engine.num_eigs == 2, "num_eigs restored");
  expect(hook_calls >= 1, "init_vector hook used");
  auto both = engine.run();                                     // tuple overload
  expect(std::get<0>(both).size() == 2 && std::get<1>(both).size() == 2, "tuple overload sizes");
  expect_close(4.0, std::get<0>(both)[0], 4.0 * engine.eps, "largest first for find_maximum");
  expect_close(1.0, std::get<0>(both)[1], 1e-10, "second eigenvalue");
}

static void run_iteration_direct() {
  std::printf("[case] run_iteration called directly with an orthogonalizeTo list (lambda_lanczos.hpp:216-322)\n");
  // chain matrix tridiag(1, 2, 1): eigenvalues 2 + sqrt2, 2, 2 - sqrt2; the top eigenvector (1, sqrt2, 1)/2 is locked,
  // so one pass for the two largest remaining pairs must return 2 and 2 - sqrt2
  std::vector<std::vector<double>> m = {{2, 1, 0}, {1, 2, 1}, {0, 1, 2}};
  ll::LambdaLanczos<double> engine(dense<double>(m), 3, true, 1);
  engine.init_vector = [](std::vector<double>& v) { v = {0.3, -0.8, 0.5}; };
  std::vector<std::vector<double>> locked = {{0.5, std::sqrt(0.5), 0.5}};
  std::vector<double> values;
  std::vector<std::vector<double>> vectors;
  const size_t itern = engine.run_iteration(values, vectors, 2, locked);
  expect(itern == 2, "two iterations span the deflated space");
  expect(values.size() == 2 && vectors.size() == 2, "two pairs returned");
  const double want[2] = {2.0, 2.0 - std::sqrt(2.0)};
  for (size_t r = 0; r < values.size() && r < 2; ++r) {
    expect_close(want[r], values[r], 1e-12, "deflated eigenvalue");
    double dot = 0;
    for (size_t i = 0; i < 3; ++i) dot += vectors[r][i] * locked[0][i];
    expect(std::abs(dot) <= 1e-12, "orthogonal to the locked vector");
  }
}

static void const_and_default_hook() {
  std::printf("[case] const engine: run_iteration is const and init_vector defaults to a callable (lambda_lanczos.hpp:133,216-220)\n");
  std::vector<std::vector<double>> m = {{2, 1, 0}, {1, 2, 1}, {0, 1, 2}};
  const ll::LambdaLanczos<double> engine(dense<double>(m), 3, true, 1);
  // user code that invokes the public hook itself, like code written against the reference may do
  std::vector<double> v(1000, 7.0);
  engine.init_vector(v);
  double lo = 1e9, hi = -1e9, sum = 0;
  for (double e : v) lo = std::min(lo, e), hi = std::max(hi, e), sum += e;
  expect(lo >= -1.0 && hi <= 1.0 && lo < -0.5 && hi > 0.5 && std::abs(sum) < 200.0, "default init_vector fills uniform [-1, 1]");
  std::vector<std::complex<double>> vz(500);
  ll::LambdaLanczos<cplx> ez(dense<cplx>({{cplx(1, 0)}}), 1, true, 1);
  ez.init_vector(vz);
  bool imag_used = false;
  for (auto& e : vz) imag_used = imag_used || std::abs(e.imag()) > 0.1;
  expect(imag_used, "complex default initialiser fills both parts");
  // run_iteration on a CONST engine with the (random) default start vector
  std::vector<double> values;
  std::vector<std::vector<double>> vectors;
  std::vector<std::vector<double>> none;
  const size_t itern = engine.run_iteration(values, vectors, 1, none);
  expect(itern >= 2 && itern <= 3, "run_iteration on a const engine");
  expect(!values.empty(), "one pair returned");
  if (!values.empty()) expect_close(2.0 + std::sqrt(2.0), values[0], 1e-10, "largest eigenvalue of tridiag(1,2,1)");
}

static void device_operator() {
  std::printf("[case] device-resident CsrMatrix operator, 5-point Laplacian 40x40 (analytic spectrum)\n");
  const int64_t N = 40, n = N * N;
  std::vector<int64_t> rp{0};
  std::vector<int32_t> ci;
  std::vector<double> va;
  for (int64_t r = 0; r < n; ++r) {
    const int64_t y = r / N, x = r % N;
    const int64_t nb[5] = {y > 0 ? r - N : -1, x > 0 ? r - 1 : -1, r, x + 1 < N ? r + 1 : -1, y + 1 < N ? r + N : -1};
    for (int t = 0; t < 5; ++t)
      if (nb[t] >= 0) {
        ci.push_back((int32_t)nb[t]);
        va.push_back(t == 2 ? 4.0 : -1.0);
      }
    rp.push_back((int64_t)ci.size());
  }
  ll::CsrMatrix<double> A(rp, ci, va);
  ll::LambdaLanczos<double> engine(A, (size_t)n, false, 1);
  engine.eigenvalue_offset = -8.0;
  engine.init_vector = seeded_init<double>(1);
  double value;
  std::vector<double> vec;
  engine.run(value, vec);
  expect_close(4.0 - 4.0 * std::cos(M_PI / (N + 1)), value, 8.0 * engine.eps * 10, "smallest Laplacian eigenvalue");
  expect(vec.size() == (size_t)n, "eigenvector length");
}

static void operator_zoo() {
  std::printf("[case] matrix-free LatticeOperator: T1:262 DYNAMIC_MATRIX (open chain n=10), lambda_min = -2cos(pi/11)\n");
  {
    const size_t n = 10;
    ll::LatticeOperator<double> chain({(int64_t)n}, 0.0, {-1.0}, {false});
    ll::LambdaLanczos<double> engine(chain, n, false, 1);
    engine.eps = 1e-14;
    engine.eigenvalue_offset = -chain.inf_norm() * 5;  // the reference's test uses -10
    engine.init_vector = seeded_init<double>(1);
    double value;
    std::vector<double> vec;
    engine.run(value, vec);
    const double want = -2.0 * std::cos(M_PI / (double)(n + 1));
    expect_close(want, value, std::abs(want) * engine.eps * 10, "open-chain ground state energy");
    std::vector<double> sine(n);
    for (size_t i = 0; i < n; ++i) sine[i] = std::sin((double)(i + 1) * M_PI / (double)(n + 1));
    double nn = 0;
    for (double v : sine) nn += v * v;
    for (double& v : sine) v /= std::sqrt(nn);
    expect(misfit(sine, vec) <= 1e-13, "open-chain ground state vector");
  }
  std::printf("[case] DenseMatrix: T1:442 MULTIPLE_EIGENPAIRS matrix resident on the device\n");
  {
    std::vector<std::vector<double>> m = {{6, -3, -3, 0, -1, 1, -1, 1}, {-3, -4, 2, 2, -1, -5, 0, -4},
                                          {-3, 2, 2, -3, 0, 0, -1, -1}, {0, 2, -3, 0, -3, 3, 2, 2},
                                          {-1, -1, 0, -3, -2, 0, -5, -4}, {1, -5, 0, 3, 0, -4, 5, 0},
                                          {-1, 0, -1, 2, -5, 5, -4, 4}, {1, -4, -1, 2, -4, 0, 4, 2}};
    ll::DenseMatrix<double> A(m);
    ll::LambdaLanczos<double> engine(A, 8, false, 3);
    engine.eps = 1e-7;
    engine.init_vector = seeded_init<double>(1);
    std::vector<double> values;
    std::vector<std::vector<double>> vectors;
    engine.run(values, vectors);
    const double want[3] = {-13.21508597, -8.50033154, -4.26674892};
    expect(values.size() == 3, "three eigenvalues");
    for (size_t r = 0; r < values.size() && r < 3; ++r) expect_close(want[r], values[r], 1e-7, "dense eigenvalue");
  }
  std::printf("[case] LatticeOperator<complex>: T2:106 ring n=100 as a matrix-free operator, a = 3i\n");
  {
    const size_t n = 100;
    ll::LatticeOperator<cplx> ring({(int64_t)n}, 0.0, {-1.0}, {true});
    ll::Exponentiator<cplx> ex(ring, n);
    std::vector<cplx> input(n, 0.0), output;
    input[0] = input[n - 1] = cplx(1, 2);
    input[n / 2] = cplx(8, 2);
    double nn = 0;
    for (auto& c : input) nn += std::norm(c);
    for (auto& c : input) c /= std::sqrt(nn);
    expect(ex.run(cplx(0, 3), input, output) == 19, "iteration count of the reference (tests/golden/exponentiator.json)");
    double on = 0;
    for (auto& c : output) on += std::norm(c);
    expect(std::abs(std::sqrt(on) - 1.0) <= 1e-12, "unitary evolution keeps the norm");
  }
}

static void exponentiator() {
  std::printf("[case] T2:106 EXPONENTIATE_LARGE_MATRIX — periodic chain n=100, a = 3i, analytic plane waves\n");
  const size_t n = 100;
  const double t = -1.0;
  auto hop = [n, t](const std::vector<cplx>& in, std::vector<cplx>& out) {
    for (size_t i = 0; i < n; ++i) out[i] += t * (in[(i + 1) % n] + in[(i + n - 1) % n]);
  };
  ll::Exponentiator<cplx> ex(hop, n);
  std::vector<cplx> input(n, 0.0), output;  // output deliberately unsized (exponentiator_test.cpp:131)
  input[0] = input[n - 1] = cplx(1, 2);
  input[n / 2] = cplx(8, 2);
  double nn = 0;
  for (auto& c : input) nn += std::norm(c);
  for (auto& c : input) c /= std::sqrt(nn);
  const cplx a(0.0, 3.0);
  const size_t itern = ex.run(a, input, output);
  std::vector<cplx> exact(n, 0.0);
  for (size_t j = 0; j < n; ++j) {  // exp(a H) through the plane-wave eigenbasis, eigenvalue 2 t cos k
    const double k = 2 * M_PI * (double)j / (double)n;
    cplx proj = 0;
    for (size_t i = 0; i < n; ++i) proj += std::conj(std::polar(1.0, k * (double)i)) * input[i];
    const cplx w = std::exp(a * (2 * t * std::cos(k))) * proj / (double)n;
    for (size_t i = 0; i < n; ++i) exact[i] += std::polar(1.0, k * (double)i) * w;
  }
  expect(misfit(exact, output) <= ex.eps * 10, "exp(aA)v against the analytic result (exponentiator_test.cpp:147-153)");
  expect(itern == 19, "iteration count of the reference for this input (tests/golden/exponentiator.json)");
  std::vector<cplx> tout;
  expect(ex.taylor_run(a, input, tout) == 37, "taylor_run term count of the reference");
  expect(misfit(exact, tout) <= ex.eps * 10, "taylor_run result");
}

static void device_resident_io() {
  std::printf("[case] device-resident I/O: init_vector_device + run_device, Exponentiator::run_device in place\n");
  {
    const size_t n = 100;
    ll::LatticeOperator<cplx> ring({(int64_t)n}, 0.0, {-1.0}, {true});
    ll::Exponentiator<cplx> ex(ring, n);
    std::vector<cplx> psi(n, 0.0), host_out, dev_out(n);
    psi[0] = psi[n - 1] = cplx(1, 2);
    psi[n / 2] = cplx(8, 2);
    const size_t it_host = ex.run(cplx(0, 3), psi, host_out);
    ll_context* c = ring.context().get();
    void* d = nullptr;
    expect(ll_malloc(c, n * sizeof(cplx), &d) == LL_OK, "ll_malloc");
    ll_memcpy_h2d(c, d, psi.data(), n * sizeof(cplx));
    const size_t it_dev = ex.run_device(cplx(0, 3), (const cplx*)d, (cplx*)d);  // psi <- exp(aA) psi in place
    ll_memcpy_d2h(c, dev_out.data(), d, n * sizeof(cplx));
    ll_free(c, d);
    expect(it_dev == it_host, "same iteration count as the host-buffer call");
    expect(std::memcmp(dev_out.data(), host_out.data(), n * sizeof(cplx)) == 0, "same bits as the host-buffer call");
  }
  {
    const size_t n = 64;
    ll::LatticeOperator<double> chain({(int64_t)n}, 0.0, {-1.0}, {false});
    std::vector<double> start(n);
    for (size_t i = 0; i < n; ++i) start[i] = std::cos(0.37 * (double)i) + 0.1;
    ll::LambdaLanczos<double> host_engine(chain, n, false, 2), dev_engine(chain, n, false, 2);
    host_engine.init_vector = [&start](std::vector<double>& v) { v = start; };
    std::vector<double> hv, dv;
    std::vector<std::vector<double>> hx;
    host_engine.run(hv, hx);
    ll_context* c = chain.context().get();
    void *d_start = nullptr, *d_vecs = nullptr;
    ll_malloc(c, n * sizeof(double), &d_start);
    ll_malloc(c, 2 * n * sizeof(double), &d_vecs);
    ll_memcpy_h2d(c, d_start, start.data(), n * sizeof(double));
    dev_engine.init_vector_device = (const double*)d_start;
    const size_t found = dev_engine.run_device(dv, (double*)d_vecs);
    std::vector<double> dx(2 * n);
    ll_memcpy_d2h(c, dx.data(), d_vecs, 2 * n * sizeof(double));
    ll_free(c, d_start);
    ll_free(c, d_vecs);
    expect(found == hv.size() && dv == hv, "same eigenvalues as the host-buffer call");
    expect(dev_engine.getIterationCounts() == host_engine.getIterationCounts(), "same iteration counts");
    bool same = found == 2;
    for (size_t r = 0; same && r < 2; ++r) same = std::memcmp(&dx[r * n], hx[r].data(), n * sizeof(double)) == 0;
    expect(same, "same eigenvector bits as the host-buffer call");
  }
}

// The context's round-6 additions: which transport answers the collectives, and the per-context tuning keys (unknown keys throw).
static void context_queries() {
  std::printf("[case] Context::transport / set_tuning\n");
  ll::Context ctx(0);
  expect(ctx.transport() == "none", "no communicator: transport() == \"none\"");
  ctx.set_tuning("pair_gs", "0");
  ctx.set_tuning("pair_gs", nullptr);
  bool threw = false;
  try {
    ctx.set_tuning("no_such_key", "1");
  } catch (const std::exception& e) {
    threw = std::string(e.what()).find("unknown key") != std::string::npos;
  }
  expect(threw, "an unknown tuning key is an error");
}

int main() {
  try {
    context_queries();
    eigen_cases();
    api_shapes();
    run_iteration_direct();
    const_and_default_hook();
    device_operator();
    operator_zoo();
    exponentiator();
    device_resident_io();
  } catch (const std::exception& e) {
    std::printf("EXCEPTION: %s\n", e.what());
    return 2;
  }
  std::printf("%s (%d failed expectations)\n", failures ? "FAILED" : "PASSED", failures);
  return failures ? 1 : 0;
}
