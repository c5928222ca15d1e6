// Stress harness for the host-callback path on an exhausted Krylov space: the ring of
// examples/drop_in.cpp, many mt19937 start vectors in ONE process; prints every run whose counts / values deviate.
//   g++ -std=c++17 -O2 -Iinclude tools/dropin_stress.cpp -o tools/_build/dropin_stress -Llambda-lanczos_amd/lib -llanczos_hip
#include <lambda_lanczos_hip/lambda_lanczos.hpp>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <tuple>
#include <vector>

int main(int argc, char** argv) {
  const unsigned first = argc > 1 ? (unsigned)std::strtoul(argv[1], nullptr, 10) : 1u;
  const unsigned count = argc > 2 ? (unsigned)std::strtoul(argv[2], nullptr, 10) : 100u;
  const size_t num_eigs = argc > 3 ? (size_t)std::strtoul(argv[3], nullptr, 10) : 2u;  // 1: the first pass only
  const int n = 2000;
  std::vector<std::tuple<int, int, double>> triplets;
  for (int i = 0; i < n; ++i) {
    triplets.emplace_back(i, i, (i % 2 ? 0.3 : -0.3));
    triplets.emplace_back(i, (i + 1) % n, -1.0);
    triplets.emplace_back(i, (i + n - 1) % n, -1.0);
  }
  size_t calls = 0;
  auto mv_mul = [&](const std::vector<double>& in, std::vector<double>& out) {
    ++calls;
    for (const auto& t : triplets) out[std::get<0>(t)] += std::get<2>(t) * in[std::get<1>(t)];
  };
  int bad = 0;
  const bool from_device = first == 0;  // first = 0: every init_vector call draws its seed from std::random_device (and logs it)
  std::vector<unsigned> drawn;
  for (unsigned seed = first; seed < first + count; ++seed) {
    drawn.clear();
    lambda_lanczos::LambdaLanczos<double> engine(mv_mul, n, false, num_eigs);
    engine.eigenvalue_offset = -3.0;
    engine.init_vector = [seed, from_device, &drawn](std::vector<double>& v) {
      unsigned sd = seed;
      if (from_device) {
        std::random_device dev;
        sd = dev();
        drawn.push_back(sd);
      }
      std::mt19937 mt(sd);
      std::uniform_real_distribution<double> rand(-1.0, 1.0);
      for (auto& e : v) e = rand(mt);
    };
    std::vector<double> values;
    std::vector<std::vector<double>> vectors;
    calls = 0;
    engine.run(values, vectors);
    const auto& c = engine.getIterationCounts();
    // pass 1 exhausts the Krylov space at 1002 and stops at 1003 for every start vector seen so far; a LATER pass that starts
    // from the same vector converges near 985, one that starts from a fresh random vector ends at 1001, 1002 or ~1850
    const bool ok = !c.empty() && c[0] == 1003 && std::abs(values[0] + 2.022374841616) < 2e-11 &&
                    (values.size() < 2 || std::abs(values[1] + 2.022365081214) < 2e-11);
    if (!ok) {
      ++bad;
      std::printf("seed %u: E0 = %.12f E1 = %.12f counts", seed, values[0], values.size() > 1 ? values[1] : 0.0);
      for (auto x : c) std::printf(" %zu", x);
      std::printf(" calls %zu seeds", calls);
      for (auto d : drawn) std::printf(" %u", d);
      std::printf("\n");
      std::fflush(stdout);
    }
  }
  std::printf("bad = %d of %u\n", bad, count);
  return bad ? 1 : 0;
}
