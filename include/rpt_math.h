/*
 * rpt_math.h — the transcendental functions of the parity arithmetic contract.
 *
 * The reference calls f64::exp / ln / atan / sin_cos / acos / atan2 (material.rs:143,247-248,
 * 260-261; environment.rs:27-28), i.e. whatever libm the platform links — results differ by an
 * ulp between glibc, musl and a GPU's device library, and in a path tracer one differing ulp
 * flips a rejection-sampling or self-intersection decision a few bounces later.  To make
 * "same seed -> same image" a bit-exact statement between the CPU oracle and the gfx950
 * kernels, both evaluate THESE functions: the classic fdlibm algorithms (Sun Microsystems,
 * "Developed at SunSoft ... Permission to use, copy, modify, and distribute this software is
 * freely granted, provided that this notice is preserved") restated as header-only inline code
 * that uses only IEEE +,-,*,/,sqrt and integer bit operations — no FMA when compiled with
 * -ffp-contract=off, hence identical bits on x86-64 and gfx950.
 * Accuracy is < 1 ulp, the same class as glibc's; tests/test_rpt_math.py checks every function
 * against the host libm over millions of arguments.
 *
 * Usable from C++ host code and from HIP device code (RPT_MATH_FN supplies the qualifiers).
 */
#ifndef RPT_MATH_H
#define RPT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RPT_MATH_FN __host__ __device__ static inline __attribute__((always_inline))
#else
#define RPT_MATH_FN static inline
#endif

RPT_MATH_FN uint64_t rptm_bits(double x) {
  uint64_t u;
  __builtin_memcpy(&u, &x, 8);
  return u;
}
RPT_MATH_FN double rptm_from_bits(uint64_t u) {
  double x;
  __builtin_memcpy(&x, &u, 8);
  return x;
}
RPT_MATH_FN int32_t rptm_hi(double x) { return (int32_t)(rptm_bits(x) >> 32); }
RPT_MATH_FN uint32_t rptm_lo(double x) { return (uint32_t)rptm_bits(x); }
RPT_MATH_FN double rptm_words(int32_t hi, uint32_t lo) {
  return rptm_from_bits(((uint64_t)(uint32_t)hi << 32) | lo);
}
RPT_MATH_FN double rptm_fabs(double x) { return rptm_from_bits(rptm_bits(x) & 0x7FFFFFFFFFFFFFFFull); }

/* ---------------------------------------------------------------- exp (fdlibm e_exp.c) */
RPT_MATH_FN double rpt_exp(double x) {
  const double one = 1.0, huge = 1.0e+300, twom1000 = 9.33263618503218878990e-302,
               o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02,
               ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
               P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  double y, hi = 0.0, lo = 0.0, c, t;
  int32_t k = 0, xsb;
  uint32_t hx = (uint32_t)rptm_hi(x);
  xsb = (int32_t)((hx >> 31) & 1u);
  hx &= 0x7fffffffu;
  if (hx >= 0x40862E42u) { /* |x| >= 709.78 */
    if (hx >= 0x7ff00000u) {
      if (((hx & 0xfffffu) | rptm_lo(x)) != 0) return x + x; /* NaN */
      return (xsb == 0) ? x : 0.0;                            /* exp(+-inf) = {inf, 0} */
    }
    if (x > o_threshold) return huge * huge;
    if (x < u_threshold) return twom1000 * twom1000;
  }
  if (hx > 0x3fd62e42u) {   /* |x| > 0.5 ln2 */
    if (hx < 0x3FF0A2B2u) { /* and |x| < 1.5 ln2 */
      hi = x - (xsb ? -ln2HI : ln2HI);
      lo = xsb ? -ln2LO : ln2LO;
      k = 1 - xsb - xsb;
    } else {
      k = (int32_t)(invln2 * x + (xsb ? -0.5 : 0.5));
      t = (double)k;
      hi = x - t * ln2HI;
      lo = t * ln2LO;
    }
    x = hi - lo;
  } else if (hx < 0x3e300000u) { /* |x| < 2^-28 */
    return one + x;
  } else {
    k = 0;
  }
  t = x * x;
  c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return one - ((x * c) / (c - 2.0) - x);
  y = one - ((lo - (x * c) / (2.0 - c)) - hi);
  if (k >= -1021) {
    return rptm_words(rptm_hi(y) + (int32_t)((uint32_t)k << 20), rptm_lo(y));
  }
  y = rptm_words(rptm_hi(y) + (int32_t)((uint32_t)(k + 1000) << 20), rptm_lo(y));
  return y * twom1000;
}

/* ---------------------------------------------------------------- log (fdlibm e_log.c) */
RPT_MATH_FN double rpt_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               two54 = 1.80143985094819840000e+16, Lg1 = 6.666666666666735130e-01,
               Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01,
               Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
  double hfsq, f, s, z, R, w, t1, t2, dk;
  int32_t k = 0, hx = rptm_hi(x), i, j;
  uint32_t lx = rptm_lo(x);
  if (hx < 0x00100000) { /* x < 2^-1022 */
    if ((((uint32_t)hx & 0x7fffffffu) | lx) == 0) return -two54 / 0.0; /* log(+-0) = -inf */
    if (hx < 0) return (x - x) / 0.0;                                   /* log(-#) = NaN */
    k -= 54;
    x *= two54;
    hx = rptm_hi(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  i = (hx + 0x95f64) & 0x100000;
  x = rptm_words(hx | (i ^ 0x3ff00000), rptm_lo(x)); /* normalise x or x/2 */
  k += (i >> 20);
  f = x - 1.0;
  if ((0x000fffff & (2 + hx)) < 3) { /* |f| < 2^-20 */
    if (f == 0.0) {
      if (k == 0) return 0.0;
      dk = (double)k;
      return dk * ln2_hi + dk * ln2_lo;
    }
    R = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0) return f - R;
    dk = (double)k;
    return dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  s = f / (2.0 + f);
  dk = (double)k;
  z = s * s;
  i = hx - 0x6147a;
  w = z * z;
  j = 0x6b851 - hx;
  t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  R = t2 + t1;
  if (i > 0) {
    hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  if (k == 0) return f - s * (f - R);
  return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* ---------------------------------------------------------------- atan (fdlibm s_atan.c) */
RPT_MATH_FN double rpt_atan(double x) {
  const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
               aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
               aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
               aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
               aT10 = 1.62858201153657823623e-02;
  double w, s1, s2, z, hi_, lo_;
  int32_t hx = rptm_hi(x), ix = hx & 0x7fffffff, id;
  if (ix >= 0x44100000) { /* |x| >= 2^66 */
    if (ix > 0x7ff00000 || (ix == 0x7ff00000 && rptm_lo(x) != 0)) return x + x; /* NaN */
    if (hx > 0) return 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
    return -1.57079632679489655800e+00 - 6.12323399573676603587e-17;
  }
  if (ix < 0x3fdc0000) {            /* |x| < 0.4375 */
    if (ix < 0x3e200000) return x;  /* |x| < 2^-29 */
    id = -1;
  } else {
    x = rptm_fabs(x);
    if (ix < 0x3ff30000) {   /* |x| < 1.1875 */
      if (ix < 0x3fe60000) { /* 7/16 <= |x| < 11/16 */
        id = 0;
        x = (2.0 * x - 1.0) / (2.0 + x);
      } else { /* 11/16 <= |x| < 19/16 */
        id = 1;
        x = (x - 1.0) / (x + 1.0);
      }
    } else {
      if (ix < 0x40038000) { /* |x| < 2.4375 */
        id = 2;
        x = (x - 1.5) / (1.0 + 1.5 * x);
      } else { /* 2.4375 <= |x| < 2^66 */
        id = 3;
        x = -1.0 / x;
      }
    }
  }
  z = x * x;
  w = z * z;
  s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) return x - x * (s1 + s2);
  if (id == 0) { hi_ = 4.63647609000806093515e-01; lo_ = 2.26987774529616870924e-17; }
  else if (id == 1) { hi_ = 7.85398163397448278999e-01; lo_ = 3.06161699786838301793e-17; }
  else if (id == 2) { hi_ = 9.82793723247329054082e-01; lo_ = 1.39033110312309984516e-17; }
  else { hi_ = 1.57079632679489655800e+00; lo_ = 6.12323399573676603587e-17; }
  z = hi_ - ((x * (s1 + s2) - lo_) - x);
  return (hx < 0) ? -z : z;
}

/* ------------------------------------------------ sin / cos kernels (fdlibm k_sin.c, k_cos.c) */
RPT_MATH_FN double rptm_kernel_sin(double x, double y, int iy) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z, r, v;
  int32_t ix = rptm_hi(x) & 0x7fffffff;
  if (ix < 0x3e400000) return x; /* |x| < 2^-27 */
  z = x * x;
  v = z * x;
  r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

RPT_MATH_FN double rptm_kernel_cos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double a, hz, z, r, qx;
  int32_t ix = rptm_hi(x) & 0x7fffffff;
  if (ix < 0x3e400000) return 1.0; /* |x| < 2^-27 */
  z = x * x;
  r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  if (ix < 0x3FD33333) return 1.0 - (0.5 * z - (z * r - x * y)); /* |x| < 0.3 */
  if (ix > 0x3fe90000) qx = 0.28125; /* x > 0.78125 */
  else qx = rptm_words(ix - 0x00200000, 0); /* x/4 */
  hz = 0.5 * z - qx;
  a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}

/* sin and cos of x for |x| < 3*pi/4 (fdlibm e_rem_pio2.c, first two cases; the only caller is
 * Beckmann sampling with theta = atan(.) in [0, pi/2], material.rs:247-248).  Outside that
 * range the result is NaN: the contract is explicit, not silently wrong. */
RPT_MATH_FN void rpt_sincos_pio2(double x, double* s, double* c) {
  const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11,
               pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
  int32_t hx = rptm_hi(x), ix = hx & 0x7fffffff;
  if (ix <= 0x3fe921fb) { /* |x| <= pi/4 */
    *s = rptm_kernel_sin(x, 0.0, 0);
    *c = rptm_kernel_cos(x, 0.0);
    return;
  }
  if (ix < 0x4002d97c) { /* |x| < 3pi/4: n = +-1 */
    double z, y0, y1;
    if (hx > 0) {
      z = x - pio2_1;
      if (ix != 0x3ff921fb) { y0 = z - pio2_1t; y1 = (z - y0) - pio2_1t; }
      else { z -= pio2_2; y0 = z - pio2_2t; y1 = (z - y0) - pio2_2t; }
      *s = rptm_kernel_cos(y0, y1);          /* n = 1 */
      *c = -rptm_kernel_sin(y0, y1, 1);
    } else {
      z = x + pio2_1;
      if (ix != 0x3ff921fb) { y0 = z + pio2_1t; y1 = (z - y0) + pio2_1t; }
      else { z += pio2_2; y0 = z + pio2_2t; y1 = (z - y0) + pio2_2t; }
      *s = -rptm_kernel_cos(y0, y1);         /* n = -1 */
      *c = rptm_kernel_sin(y0, y1, 1);
    }
    return;
  }
  *s = *c = (x - x) / (x - x); /* out of contract (or NaN/inf) */
}

/* ---------------------------------------------------------------- acos (fdlibm e_acos.c) */
RPT_MATH_FN double rptm_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_sqrt(x); /* correctly rounded on gfx950, as on the host */
#else
  return __builtin_sqrt(x);
#endif
}

RPT_MATH_FN double rpt_acos(double x) {
  const double one = 1.0, pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00,
               pio2_lo = 6.12323399573676603587e-17, pS0 = 1.66666666666666657415e-01,
               pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
               pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04,
               pS5 = 3.47933107596021167570e-05, qS1 = -2.40339491173441421878e+00,
               qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
               qS4 = 7.70381505559019352791e-02;
  double z, p, q, r, w, s, c, df;
  int32_t hx = rptm_hi(x), ix = hx & 0x7fffffff;
  if (ix >= 0x3ff00000) { /* |x| >= 1 */
    if (((uint32_t)(ix - 0x3ff00000) | rptm_lo(x)) == 0) {
      if (hx > 0) return 0.0;
      return pi + 2.0 * pio2_lo;
    }
    return (x - x) / (x - x);
  }
  if (ix < 0x3fe00000) { /* |x| < 0.5 */
    if (ix <= 0x3c600000) return pio2_hi + pio2_lo;
    z = x * x;
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    r = p / q;
    return pio2_hi - (x - (pio2_lo - r * x));
  } else if (hx < 0) { /* x < -0.5 */
    z = (one + x) * 0.5;
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    s = rptm_sqrt(z);
    r = p / q;
    w = r * s - pio2_lo;
    return pi - 2.0 * (s + w);
  } else { /* x > 0.5 */
    z = (one - x) * 0.5;
    s = rptm_sqrt(z);
    df = rptm_words(rptm_hi(s), 0);
    c = (z - df * df) / (s + df);
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    r = p / q;
    w = r * s + c;
    return 2.0 * (df + w);
  }
}

/* ---------------------------------------------------------------- atan2 (fdlibm e_atan2.c) */
RPT_MATH_FN double rpt_atan2(double y, double x) {
  const double tiny = 1.0e-300, pi_o_4 = 7.8539816339744827900E-01, pi_o_2 = 1.5707963267948965580E+00,
               pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
  double z;
  int32_t k, m, hx = rptm_hi(x), ix = hx & 0x7fffffff, hy = rptm_hi(y), iy = hy & 0x7fffffff;
  uint32_t lx = rptm_lo(x), ly = rptm_lo(y);
  if (((uint32_t)ix | ((lx | (0u - lx)) >> 31)) > 0x7ff00000u || ((uint32_t)iy | ((ly | (0u - ly)) >> 31)) > 0x7ff00000u)
    return x + y; /* NaN */
  if (((uint32_t)(hx - 0x3ff00000) | lx) == 0) return rpt_atan(y); /* x = 1 */
  m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                          /* 2*sign(x) + sign(y) */
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
  k = (iy - ix) >> 20;
  if (k > 60) z = pi_o_2 + 0.5 * pi_lo;    /* |y/x| > 2^60 */
  else if (hx < 0 && k < -60) z = 0.0;     /* |y|/x < -2^60 */
  else z = rpt_atan(rptm_fabs(y / x));
  switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

#endif /* RPT_MATH_H */
