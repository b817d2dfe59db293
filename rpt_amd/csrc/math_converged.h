// math_converged.h — atan, acos, atan2, exp and sincos of include/rpt_math.h (the fdlibm restatements the CPU checker
// evaluates too) with their range cases CONVERGED: what fdlibm writes as an if-chain of argument ranges, each with its own
// division (and, in acos, its own pair of polynomials and square root; in sincos its own pair of kernels), is here ONE
// division, ONE polynomial pair, ONE square root behind per-lane selected operands.  On a 64-lane wave whose lanes fall
// into different ranges the if-chain runs every branch in turn — Hdri::get_color's atan2 + acos (environment.rs:25-52)
// four divisions for the atan and three rounds of polynomials + division for the acos, Beckmann sampling and the
// Beckmann distribution (material.rs:221-254) two divisions per exp and two kernel pairs per sincos — the converged
// form runs one.  (-DRPT_MATH_PLAIN: the rptc_ names are rpt_math.h's functions, for A/B builds.)
//
// Every lane evaluates, operation for operation, what its own fdlibm branch evaluates: the selected operands only
// name that branch's constants, and the rewrites used to share an expression are exact in IEEE arithmetic
// (1.0 * x == x; 0.0 * x - 1.0 == -1.0 and 0.0 + x == x for finite x > 0; a + b == b + a; 1.0 - x == 1.0 + (-x)).
// Hence the same bits as rpt_math.h for every input, NaNs and signed zeros included: tests/test_math_converged.py
// compares them on the host over every range boundary and 10^7 random arguments, tests/test_gpu_parity.py on the device.
#pragma once
#include "../../include/rpt_math.h"

#ifdef RPT_MATH_PLAIN
#define rptc_atan rpt_atan
#define rptc_acos rpt_acos
#define rptc_atan2 rpt_atan2
#define rptc_exp rpt_exp
#define rptc_sincos_pio2 rpt_sincos_pio2
#else

/* fdlibm s_atan.c, as rpt_math.h rpt_atan */
RPT_MATH_FN double rptc_atan(double x) {
  const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
               aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
               aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
               aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
               aT10 = 1.62858201153657823623e-02;
  const int32_t hx = rptm_hi(x), ix = hx & 0x7fffffff;
  if (ix >= 0x44100000) { /* |x| >= 2^66, inf, NaN */
    if (ix > 0x7ff00000 || (ix == 0x7ff00000 && rptm_lo(x) != 0)) return x + x;
    if (hx > 0) return 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
    return -1.57079632679489655800e+00 - 6.12323399573676603587e-17;
  }
  if (ix < 0x3e200000) return x; /* |x| < 2^-29 */
  const bool small = ix < 0x3fdc0000; /* |x| < 0.4375: no reduction */
  /* the four reductions (2x-1)/(2+x), (x-1)/(x+1), (x-1.5)/(1+1.5x), -1/x are (a x - b) / (a + b x) */
  const bool r0 = ix < 0x3fe60000, r1 = ix < 0x3ff30000, r2 = ix < 0x40038000;
  const double a = r0 ? 2.0 : (r2 ? 1.0 : 0.0);
  const double b = (r2 && !r1) ? 1.5 : 1.0;
  const double ax = rptm_fabs(x);
  const double num = a * ax - b, den = a + b * ax;
  const double xr = small ? x : num / den;
  const double z = xr * xr, w = z * z;
  const double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  const double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (small) return xr - xr * (s1 + s2);
  const double hi_ = r0 ? 4.63647609000806093515e-01 : r1 ? 7.85398163397448278999e-01
                   : r2 ? 9.82793723247329054082e-01 : 1.57079632679489655800e+00;
  const double lo_ = r0 ? 2.26987774529616870924e-17 : r1 ? 3.06161699786838301793e-17
                   : r2 ? 1.39033110312309984516e-17 : 6.12323399573676603587e-17;
  const double zz = hi_ - ((xr * (s1 + s2) - lo_) - xr);
  return (hx < 0) ? -zz : zz;
}

/* fdlibm e_acos.c, as rpt_math.h rpt_acos */
RPT_MATH_FN double rptc_acos(double x) {
  const double one = 1.0, pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00,
               pio2_lo = 6.12323399573676603587e-17, pS0 = 1.66666666666666657415e-01,
               pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
               pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04,
               pS5 = 3.47933107596021167570e-05, qS1 = -2.40339491173441421878e+00,
               qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
               qS4 = 7.70381505559019352791e-02;
  const int32_t hx = rptm_hi(x), ix = hx & 0x7fffffff;
  if (ix >= 0x3ff00000) { /* |x| >= 1 */
    if (((uint32_t)(ix - 0x3ff00000) | rptm_lo(x)) == 0) {
      if (hx > 0) return 0.0;
      return pi + 2.0 * pio2_lo;
    }
    return (x - x) / (x - x);
  }
  if (ix <= 0x3c600000) return pio2_hi + pio2_lo; /* |x| <= 2^-57 */
  const bool small = ix < 0x3fe00000, neg = hx < 0; /* |x| < 0.5; else x < -0.5 or x > 0.5 */
  const double z = small ? x * x : (one + (neg ? x : -x)) * 0.5;
  const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  const double r = p / q;
  if (small) return pio2_hi - (x - (pio2_lo - r * x));
  const double s = rptm_sqrt(z);
  if (neg) {
    const double w = r * s - pio2_lo;
    return pi - 2.0 * (s + w);
  }
  const double df = rptm_words(rptm_hi(s), 0);
  const double c = (z - df * df) / (s + df);
  const double w = r * s + c;
  return 2.0 * (df + w);
}

/* fdlibm e_atan2.c, as rpt_math.h rpt_atan2: the special cases as they are there (no lane of a renderer takes them),
 * ONE atan behind them — of y when x = 1, else of |y / x| */
RPT_MATH_FN double rptc_atan2(double y, double x) {
  const double tiny = 1.0e-300, pi_o_4 = 7.8539816339744827900E-01, pi_o_2 = 1.5707963267948965580E+00,
               pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
  const int32_t hx = rptm_hi(x), ix = hx & 0x7fffffff, hy = rptm_hi(y), iy = hy & 0x7fffffff;
  const uint32_t lx = rptm_lo(x), ly = rptm_lo(y);
  if (((uint32_t)ix | ((lx | (0u - lx)) >> 31)) > 0x7ff00000u || ((uint32_t)iy | ((ly | (0u - ly)) >> 31)) > 0x7ff00000u)
    return x + y; /* NaN */
  const bool x_is_one = ((uint32_t)(hx - 0x3ff00000) | lx) == 0;
  const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2); /* 2*sign(x) + sign(y) */
  if (!x_is_one) {
    if (((uint32_t)iy | ly) == 0) { /* y = 0 */
      switch (m) {
        case 0: case 1: return y;
        case 2: return pi + tiny;
        default: return -pi - tiny;
      }
    }
    if (((uint32_t)ix | lx) == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny; /* x = 0 */
    if (ix == 0x7ff00000) { /* x = inf */
      if (iy == 0x7ff00000) {
        switch (m) {
          case 0: return pi_o_4 + tiny;
          case 1: return -pi_o_4 - tiny;
          case 2: return 3.0 * pi_o_4 + tiny;
          default: return -3.0 * pi_o_4 - tiny;
        }
      } else {
        switch (m) {
          case 0: return 0.0;
          case 1: return -0.0;
          case 2: return pi + tiny;
          default: return -pi - tiny;
        }
      }
    }
    if (iy == 0x7ff00000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny; /* y = inf */
  }
  const int32_t k = (iy - ix) >> 20;
  const bool huge = !x_is_one && k > 60;               /* |y/x| > 2^60 */
  const bool none = !x_is_one && !huge && hx < 0 && k < -60; /* |y|/x < -2^60 */
  const double t = rptc_atan(x_is_one ? y : rptm_fabs(y / x));
  if (x_is_one) return t;
  const double z = huge ? pi_o_2 + 0.5 * pi_lo : (none ? 0.0 : t);
  switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

/* fdlibm e_exp.c, as rpt_math.h rpt_exp: the reduction by k ln2 for |x| < 1.5 ln2 (k = +-1) is the general one with
 * t = +-1.0 (t * ln2HI and t * ln2LO are exact), for k = 0 it is the identity (x - 0.0 * ln2HI - 0.0 * ln2LO == x);
 * the two quotients (x c) / (c - 2) and (x c) / (2 - c) are one division behind a selected denominator */
RPT_MATH_FN double rptc_exp(double x) {
  const double one = 1.0, huge = 1.0e+300, twom1000 = 9.33263618503218878990e-302,
               o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02,
               ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
               P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  uint32_t hx = (uint32_t)rptm_hi(x);
  const int32_t xsb = (int32_t)((hx >> 31) & 1u);
  hx &= 0x7fffffffu;
  if (hx >= 0x40862E42u) { /* |x| >= 709.78 */
    if (hx >= 0x7ff00000u) {
      if (((hx & 0xfffffu) | rptm_lo(x)) != 0) return x + x; /* NaN */
      return (xsb == 0) ? x : 0.0;                            /* exp(+-inf) = {inf, 0} */
    }
    if (x > o_threshold) return huge * huge;
    if (x < u_threshold) return twom1000 * twom1000;
  }
  if (hx < 0x3e300000u) return one + x; /* |x| < 2^-28 */
  const bool reduce = hx > 0x3fd62e42u; /* |x| > 0.5 ln2 */
  const int32_t k_far = (int32_t)(invln2 * x + (xsb ? -0.5 : 0.5));
  const int32_t k = reduce ? (hx < 0x3FF0A2B2u ? 1 - xsb - xsb : k_far) : 0;
  const double t = (double)k;
  const double hi = x - t * ln2HI, lo = t * ln2LO;
  const double xr = hi - lo;
  const double tt = xr * xr;
  const double c = xr - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
  const double q = (xr * c) / (k == 0 ? c - 2.0 : 2.0 - c);
  if (k == 0) return one - (q - xr);
  double y = one - ((lo - q) - hi);
  if (k >= -1021) return rptm_words(rptm_hi(y) + (int32_t)((uint32_t)k << 20), rptm_lo(y));
  y = rptm_words(rptm_hi(y) + (int32_t)((uint32_t)(k + 1000) << 20), rptm_lo(y));
  return y * twom1000;
}

/* fdlibm k_sin.c / k_cos.c / the first two cases of e_rem_pio2.c, as rpt_math.h rpt_sincos_pio2: ONE sine kernel and
 * ONE cosine kernel on (x, 0) for |x| <= pi/4, on the reduced (y0, y1) otherwise, their results swapped and signed
 * per lane.  k_cos's three forms are one: with qx = 0 for |x| < 0.3, a = 1 - qx and hz = 0.5 z - qx are 1 and 0.5 z. */
RPT_MATH_FN void rptc_sincos_pio2(double x, double* s, double* c) {
  const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11,
               pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const int32_t hx = rptm_hi(x), ix = hx & 0x7fffffff;
  if (ix >= 0x4002d97c) { *s = *c = (x - x) / (x - x); return; } /* out of contract (or NaN/inf) */
  const bool n0 = ix <= 0x3fe921fb; /* |x| <= pi/4: no reduction */
  double a = x, b = 0.0;
  if (!n0) { /* n = +-1: x -+ pi/2 in two pieces (x + p == x - (-p)) */
    const double sg = hx > 0 ? 1.0 : -1.0;
    double z = x - sg * pio2_1;
    if (ix != 0x3ff921fb) { a = z - sg * pio2_1t; b = (z - a) - sg * pio2_1t; }
    else { z -= sg * pio2_2; a = z - sg * pio2_2t; b = (z - a) - sg * pio2_2t; }
  }
  const int32_t ia = rptm_hi(a) & 0x7fffffff;
  const double z = a * a;
  /* sine kernel: rptm_kernel_sin(a, b, iy = !n0) */
  double ks;
  {
    const double v = z * a;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double plain = a + v * (S1 + z * r), tail = a - ((z * (0.5 * b - v * r) - b) - v * S1);
    ks = ia < 0x3e400000 ? a : (n0 ? plain : tail); /* |a| < 2^-27 */
  }
  /* cosine kernel: rptm_kernel_cos(a, b) */
  double kc;
  {
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double qx = ia < 0x3FD33333 ? 0.0 : (ia > 0x3fe90000 ? 0.28125 : rptm_words(ia - 0x00200000, 0)); /* 0, x/4 */
    const double hz = 0.5 * z - qx, aa = 1.0 - qx;
    kc = ia < 0x3e400000 ? 1.0 : aa - (hz - (z * r - a * b));
  }
  if (n0) { *s = ks; *c = kc; }
  else if (hx > 0) { *s = kc; *c = -ks; }
  else { *s = -kc; *c = ks; }
}
#endif // RPT_MATH_PLAIN
