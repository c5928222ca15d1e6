// Real-time evolution psi(t + dt) = exp(-i H dt) psi(t) with the wave function kept in DEVICE memory between the steps
// (Exponentiator<T>::run_device): nothing n-sized crosses PCIe inside the loop.  H is a tight-binding ring handed over as
// a matrix-free lattice operator; a Gaussian wave packet with momentum k0 moves at the group velocity 2 sin(k0).
//   g++ -std=c++17 -Iinclude examples/time_evolution.cpp -o time_evolution -Llambda-lanczos_amd/lib -llanczos_hip -Wl,-rpath,$PWD/lambda-lanczos_amd/lib
#include <lambda_lanczos_hip/exponentiator.hpp>  // was: <lambda_lanczos/exponentiator.hpp>

#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>

using cplx = std::complex<double>;

int main() {
  const size_t n = 1 << 16;
  const double k0 = 1.0, dt = 0.5;
  const int steps = 200;
  lambda_lanczos::LatticeOperator<cplx> H({(int64_t)n}, 0.0, {-1.0}, {true});  // ring, hopping -1
  lambda_lanczos::Exponentiator<cplx> evolve(H, n);

  std::vector<cplx> psi(n);
  double norm = 0;
  for (size_t i = 0; i < n; ++i) {
    const double x = (double)i - (double)n / 4;
    psi[i] = std::exp(-x * x / (2.0 * 400.0 * 400.0)) * std::polar(1.0, k0 * (double)i);
    norm += std::norm(psi[i]);
  }
  for (auto& c : psi) c /= std::sqrt(norm);

  ll_context* ctx = H.context().get();
  void* d_psi = nullptr;
  if (ll_malloc(ctx, n * sizeof(cplx), &d_psi) != LL_OK) return 2;
  ll_memcpy_h2d(ctx, d_psi, psi.data(), n * sizeof(cplx));
  size_t iterations = 0;
  for (int s = 0; s < steps; ++s)  // psi <- exp(-i H dt) psi, in place, on the device
    iterations += evolve.run_device(cplx(0.0, -dt), (const cplx*)d_psi, (cplx*)d_psi);
  ll_memcpy_d2h(ctx, psi.data(), d_psi, n * sizeof(cplx));
  ll_free(ctx, d_psi);

  double n2 = 0, mean = 0;
  for (size_t i = 0; i < n; ++i) {
    n2 += std::norm(psi[i]);
    mean += (double)i * std::norm(psi[i]);
  }
  mean /= n2;
  const double expect = (double)n / 4 + 2.0 * std::sin(k0) * dt * steps;  // group velocity of E(k) = -2 cos k
  std::printf("%d steps, %zu Krylov iterations in total, norm %.12f, <x> = %.2f (expected %.2f)\n", steps, iterations,
              std::sqrt(n2), mean, expect);
  const bool ok = std::abs(std::sqrt(n2) - 1.0) < 1e-9 && std::abs(mean - expect) < 2.0;
  std::printf("%s\n", ok ? "OK" : "MISMATCH");
  return ok ? 0 : 1;
}
