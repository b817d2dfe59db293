// Host-side check of rpt_amd/csrc/math_converged.h against include/rpt_math.h: bit-for-bit on every range boundary
// (+- a few ulps), the special values, and N random arguments per function.  Prints "ok <cases>" or the first mismatch.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../rpt_amd/csrc/math_converged.h"

static uint64_t bits(double v) { uint64_t b; std::memcpy(&b, &v, 8); return b; }
static double from_bits(uint64_t b) { double v; std::memcpy(&v, &b, 8); return v; }
static bool same(double a, double b) { return bits(a) == bits(b) || (a != a && b != b); }
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next() { // splitmix64
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char** argv) {
  const long n = argc > 1 ? std::atol(argv[1]) : 10000000;
  std::vector<double> xs;
  // the high words the range tests compare against, and their neighbours, with both signs
  const uint32_t hi[] = {0x44100000, 0x3fdc0000, 0x3e200000, 0x3fe60000, 0x3ff30000, 0x40038000, 0x3ff00000, 0x3fe00000,
                         0x3c600000, 0x7ff00000, 0x00000000, 0x00100000, 0x3ff80000, 0x400921fb,
                         0x40862E42, 0x3fd62e42, 0x3FF0A2B2, 0x3e300000, 0x4002d97c, 0x3fe921fb, 0x3ff921fb, 0x3e400000, 0x3FD33333,
                         0x3fe90000, 0x40874910, 0x4086232b};
  for (uint32_t h : hi)
    for (int d = -3; d <= 3; d++)
      for (uint32_t lo : {0u, 1u, 0xffffffffu, 0x80000000u})
        for (int sgn = 0; sgn < 2; sgn++) {
          uint64_t b = ((uint64_t)h << 32 | lo) + (uint64_t)(int64_t)d;
          xs.push_back(from_bits(b | ((uint64_t)sgn << 63)));
        }
  for (double v : {0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1e-310, -1e-310, 1e300, -1e300, 0.4375, 0.6875, 1.1875, 2.4375})
    xs.push_back(v);
  long cases = 0;
  for (double x : xs) {
    if (!same(rpt_atan(x), rptc_atan(x))) { std::printf("atan %a: %a != %a\n", x, rpt_atan(x), rptc_atan(x)); return 1; }
    if (!same(rpt_acos(x), rptc_acos(x))) { std::printf("acos %a: %a != %a\n", x, rpt_acos(x), rptc_acos(x)); return 1; }
    if (!same(rpt_exp(x), rptc_exp(x))) { std::printf("exp %a: %a != %a\n", x, rpt_exp(x), rptc_exp(x)); return 1; }
    {
      double s0, c0, s1, c1;
      rpt_sincos_pio2(x, &s0, &c0);
      rptc_sincos_pio2(x, &s1, &c1);
      if (!same(s0, s1) || !same(c0, c1)) { std::printf("sincos %a: %a %a != %a %a\n", x, s0, c0, s1, c1); return 1; }
    }
    for (double y : xs) {
      if (!same(rpt_atan2(y, x), rptc_atan2(y, x))) { std::printf("atan2 %a %a: %a != %a\n", y, x, rpt_atan2(y, x), rptc_atan2(y, x)); return 1; }
      cases++;
    }
    cases += 2;
  }
  for (long i = 0; i < n; i++) {
    // atan: log-uniform magnitudes over 2^-40 .. 2^70 and uniform around the reductions' ranges
    const double m = (i & 1) ? std::ldexp(1.0 + unit(), (int)(next() % 111) - 40) : 4.0 * unit();
    const double x = (next() & 1) ? m : -m;
    if (!same(rpt_atan(x), rptc_atan(x))) { std::printf("atan %a: %a != %a\n", x, rpt_atan(x), rptc_atan(x)); return 1; }
    const double c = 2.0 * unit() - 1.0;
    const double c2 = (i & 3) == 0 ? std::copysign(1.0 - std::ldexp(unit(), -(int)(next() % 50)), c) : c; // towards +-1
    if (!same(rpt_acos(c2), rptc_acos(c2))) { std::printf("acos %a: %a != %a\n", c2, rpt_acos(c2), rptc_acos(c2)); return 1; }
    // atan2: unit-vector components (what Hdri::get_color passes) and wild magnitude ratios
    double y = 2.0 * unit() - 1.0, xx = 2.0 * unit() - 1.0;
    if ((i & 7) == 0) { y = std::ldexp(y, (int)(next() % 200) - 100); xx = std::ldexp(xx, (int)(next() % 200) - 100); }
    if ((i & 1023) == 0) xx = 1.0;
    if (!same(rpt_atan2(y, xx), rptc_atan2(y, xx))) { std::printf("atan2 %a %a: %a != %a\n", y, xx, rpt_atan2(y, xx), rptc_atan2(y, xx)); return 1; }
    // exp: the Beckmann exponent's range, the reduction's boundaries, the whole finite range
    const double e = (i & 3) == 0 ? 1500.0 * unit() - 750.0 : ((i & 3) == 1 ? 3.0 * unit() - 1.5 : -std::ldexp(1.0 + unit(), (int)(next() % 40) - 30));
    if (!same(rpt_exp(e), rptc_exp(e))) { std::printf("exp %a: %a != %a\n", e, rpt_exp(e), rptc_exp(e)); return 1; }
    // sincos: [-3pi/4, 3pi/4], denser towards 0 and around pi/4, pi/2
    double th = (i & 1) ? 2.3561944901923448 * (2.0 * unit() - 1.0) : std::ldexp(1.0 + unit(), -(int)(next() % 32));
    if ((i & 15) == 0) th = 1.5707963267948966 + std::ldexp(2.0 * unit() - 1.0, -(int)(next() % 40));
    if ((i & 15) == 1) th = 0.7853981633974483 + std::ldexp(2.0 * unit() - 1.0, -(int)(next() % 40));
    double s0, c0, s1, c1;
    rpt_sincos_pio2(th, &s0, &c0);
    rptc_sincos_pio2(th, &s1, &c1);
    if (!same(s0, s1) || !same(c0, c1)) { std::printf("sincos %a: %a %a != %a %a\n", th, s0, c0, s1, c1); return 1; }
    cases += 5;
  }
  std::printf("ok %ld\n", cases);
  return 0;
}
