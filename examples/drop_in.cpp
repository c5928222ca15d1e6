// Switching user code from lambda-lanczos to the MI355X implementation, in two steps.
//   step 1: change the include — the user's own mv_mul lambda keeps working (host callback, one vector round trip per
//           iteration), same constructor, same public fields, same run() overloads;
//   step 2: hand the matrix over instead of the lambda — the whole Krylov loop stays in device memory.
// The matrix is a small spin-chain-like Hamiltonian given as (row, column, value) triplets, the format of the
// reference's sparse sample.
//   g++ -std=c++17 -Iinclude examples/drop_in.cpp -o drop_in -Llambda-lanczos_amd/lib -llanczos_hip -Wl,-rpath,$PWD/lambda-lanczos_amd/lib
#include <lambda_lanczos_hip/lambda_lanczos.hpp>  // was: <lambda_lanczos/lambda_lanczos.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <tuple>
#include <vector>

using lambda_lanczos::LambdaLanczos;  // the namespace alias keeps existing code compiling

int main(int argc, char** argv) {
  // A fixed start vector makes the run reproducible (the reference's default draws from std::random_device,
  // lambda_lanczos.hpp:70-104); any seed works, the optional argument picks another one.
  const unsigned seed = argc > 1 ? (unsigned)std::strtoul(argv[1], nullptr, 10) : 1u;
  auto seeded_start = [seed](std::vector<double>& v) {
    std::mt19937 mt(seed);
    std::uniform_real_distribution<double> rand(-1.0, 1.0);
    for (auto& e : v) e = rand(mt);
  };
  // a ring of n sites with alternating on-site energies and nearest-neighbour hopping, as triplets
  const int n = 2000;
  std::vector<std::tuple<int, int, double>> triplets;
  for (int i = 0; i < n; ++i) {
    triplets.emplace_back(i, i, (i % 2 ? 0.3 : -0.3));
    triplets.emplace_back(i, (i + 1) % n, -1.0);
    triplets.emplace_back(i, (i + n - 1) % n, -1.0);
  }

  // ---- step 1: unmodified user code — the operator is the user's lambda
  auto mv_mul = [&](const std::vector<double>& in, std::vector<double>& out) {
    for (const auto& t : triplets) out[std::get<0>(t)] += std::get<2>(t) * in[std::get<1>(t)];
  };
  LambdaLanczos<double> engine(mv_mul, n, false, 2);  // two lowest eigenpairs
  engine.eigenvalue_offset = -3.0;                     // so that the lowest eigenvalues have the largest magnitude
  engine.init_vector = seeded_start;
  std::vector<double> values;
  std::vector<std::vector<double>> vectors;
  engine.run(values, vectors);
  std::printf("host lambda : E0 = %.12f  E1 = %.12f  (%zu + %zu iterations)\n", values[0], values[1],
              engine.getIterationCounts()[0], engine.getIterationCounts().back());

  // ---- step 2: the same matrix resident on the device (CSR built from the triplets)
  std::sort(triplets.begin(), triplets.end());
  std::vector<int64_t> row_ptr(n + 1, 0);
  std::vector<int32_t> col;
  std::vector<double> val;
  for (const auto& t : triplets) {
    ++row_ptr[std::get<0>(t) + 1];
    col.push_back(std::get<1>(t));
    val.push_back(std::get<2>(t));
  }
  for (int i = 0; i < n; ++i) row_ptr[i + 1] += row_ptr[i];
  lambda_lanczos::CsrMatrix<double> A(row_ptr, col, val);
  LambdaLanczos<double> device_engine(A, n, false, 2);
  device_engine.eigenvalue_offset = -A.inf_norm();     // a safe offset from the matrix itself
  device_engine.init_vector = seeded_start;
  std::vector<double> dvalues;
  std::vector<std::vector<double>> dvectors;
  device_engine.run(dvalues, dvectors);
  std::printf("device CSR  : E0 = %.12f  E1 = %.12f\n", dvalues[0], dvalues[1]);

  // residual of the first pair, computed with the user's own lambda
  std::vector<double> r(n, 0.0);
  mv_mul(dvectors[0], r);
  double res = 0;
  for (int i = 0; i < n; ++i) res += (r[i] - dvalues[0] * dvectors[0][i]) * (r[i] - dvalues[0] * dvectors[0][i]);
  std::printf("residual |A v - E0 v| = %.2e\n", std::sqrt(res));
  const bool ok = std::abs(values[0] - dvalues[0]) < 1e-9 && std::abs(values[1] - dvalues[1]) < 1e-9 && std::sqrt(res) < 1e-6;
  std::printf("%s\n", ok ? "OK" : "MISMATCH");
  return ok ? 0 : 1;
}
