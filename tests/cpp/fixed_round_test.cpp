// fixed_round (lambda-lanczos_amd/csrc/fixed_round.hpp) against (long long)rint(v) on the host: the four additions are IEEE
// operations on either side, so what holds here holds in the kernels.  Built and run by tests/test_fixed_round.py.
#include <cfenv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "fixed_round.hpp"

static long long bad = 0, seen = 0;
static void check(double v) {
  if (!(std::fabs(v) < 9.0e18)) return;
  ++seen;
  const long long want = (long long)std::rint(v), got = ll::fixed_round(v);
  if (want != got && bad++ < 10) std::printf("MISMATCH v=%a want=%lld got=%lld\n", v, want, got);
}

int main() {
  std::fesetround(FE_TONEAREST);
  std::mt19937_64 rng(12345);
  // every binade from 2^-1074 to 2^63, random mantissas, both signs
  for (int e = -1074; e <= 63; ++e)
    for (int i = 0; i < 4000; ++i) {
      const double m = 1.0 + (double)(rng() >> 11) * 0x1p-53;
      const double v = std::ldexp(m, e);
      check(v);
      check(-v);
    }
  // ties and neighbours of ties: k + 0.5 and the doubles next to it, small and large k; multiples of 2^31 and 2^32 +- 0.5
  for (int i = 0; i < 2000000; ++i) {
    const int sh = (int)(rng() % 52);
    const long long k = (long long)(rng() >> (12 + sh));
    const double v = (double)k + 0.5;
    for (double w : {v, std::nextafter(v, 1e300), std::nextafter(v, -1e300), -v})
      check(w);
    const double u = std::ldexp((double)(long long)(rng() % 4000000000ull), 31) + 0.5 * (double)((int)(rng() % 5) - 2);
    check(u);
    check(-u);
    check(u + std::ldexp(1.0, 31));
  }
  // integers at the top of the range (the products of a row at the bound of its grid) and around the word boundary
  for (int i = 0; i < 2000000; ++i) {
    const double v = (double)(long long)(rng() >> 1) * ((rng() & 1) ? 1.0 : -1.0);
    check(v);
    check(std::ldexp(1.0, 32) * (double)(int)(rng() % 100000) + (double)(int)(rng() % 7) - 3.0);
  }
  for (double v : {0.0, -0.0, 0.5, -0.5, 1.5, -1.5, 2.5, 0x1p31, -0x1p31, 0x1p31 + 0.5, 0x1p32 - 0.5, 0x1p62, -0x1p62, 8.99e18, -8.99e18,
                   0x1p52 + 1.0, 0x1p53, 0x1p51 + 0.5, 4.9e-324, -4.9e-324})
    check(v);
  std::printf("%lld values, %lld mismatches\n", seen, bad);
  return bad == 0 ? 0 : 1;
}
