// fp64 sin / cos / exp for IMPORTED element kernels (csrc/mpcx_ufcx.cpp puts this text in front of the user's C source and
// maps the libm names onto it: FFCx-generated tabulate_tensor functions call sin / cos / exp at every quadrature point, and
// the device libm spends 189 instructions on a sin, 62 on an exp -- general Payne-Hanek reduction, denormal and special-case
// paths inlined into every call).  Valid on the FULL double range: arguments outside the fast range, infinities and NaNs
// take the libm function (a rarely taken branch).  Accuracy (tests/test_fastmath_host.py compiles this header with g++
// and compares with long double libm on dense and random arguments):
//   mpcx_fast_sin / mpcx_fast_cos   |x| <= 2^19 pi: <= 2.5 ulp (Cody-Waite reduction to [-pi/2, pi/2] with a three-part
//                                   pi, odd near-minimax polynomial of degree 17); beyond: libm
//   mpcx_fast_exp                   -708 <= x <= 709: <= 1.5 ulp (x = n ln2 + r, |r| <= ln2 / 2, near-minimax degree 11, ldexp);
//                                   beyond (underflow into the denormals, overflow), NaN: libm.  (A 64-entry 2^(j/64) table
//                                   with a degree-5 polynomial saves 7 of the 21 instructions but its per-lane table load
//                                   stalls the in-order wave: config 2's imported right-hand side 3.45 -> 3.80 ms; not kept)
// No includes, builtins only: the same text compiles under hipRTC (device) and g++ (host test).
#pragma once
// the libm functions the slow paths fall back to (hipRTC: the device library's entry points)
#ifndef MPCX_FM_LIBM_SIN
#define MPCX_FM_LIBM_SIN(x) __builtin_sin(x)
#define MPCX_FM_LIBM_COS(x) __builtin_cos(x)
#define MPCX_FM_LIBM_EXP(x) __builtin_exp(x)
#endif
#ifndef MPCX_UFCX_MATH_FN
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC_RTC__)
#define MPCX_UFCX_MATH_FN static __device__ __host__ __attribute__((always_inline)) inline
#else
#define MPCX_UFCX_MATH_FN static inline
#endif
#endif

// Polynomial / reduction constants.  On the device they live in one __constant__ table and reach the fma's as scalar
// (SGPR) operands: an fp64 literal cannot be an inline operand, so every Horner step on literals costs a register copy
// next to its fma (measured on the benchmark's right-hand side: 150 instead of ~110 instructions per quadrature point).
#define MPCX_FM_TABLE_INIT                                                                                             \
  {                                                                                                                    \
    /* 0..7: sin(r) = r + r^3 P(r^2) on |r| <= pi/2, near-minimax (Chebyshev fit of (sin r - r) / r^3 in r^2, degree 7: */  \
    /* 1.3e-18 absolute), highest power first */                                                                       \
    2.7314446665270123e-15, -7.643970288741763e-13, 1.605897731221174e-10, -2.505210761699229e-08,                     \
        2.7557319219163205e-06, -0.00019841269841254974, 0.008333333333333316, -0.16666666666666666,                  \
        /* 8..10: pi in three parts; 11: 1/pi; 12: 2^19 pi */                                                          \
        2.0 * 1.57079632673412561417e+00, 2.0 * 6.07710050630396597660e-11, 2.0 * 2.02226624879595063154e-21,          \
        3.18309886183790671538e-01, 1647099.0, /* 13: 1/ln2; 14, 15: ln2 in two parts */                               \
        1.44269504088896338700e+00, 6.93147180369123816490e-01, 1.90821492927058770002e-10,                            \
        /* 16..25: e^r = 1 + r + r^2 Q(r) on |r| <= ln2 / 2, near-minimax (Chebyshev fit of (e^r - 1 - r) / r^2, degree 9: */ \
        /* 1.3e-17), highest power first */                                                                            \
        0x1.af389f20208c6p-26, 0x1.28917ccaf39d3p-22, 0x1.71de0db2d4e97p-19, 0x1.a019b91463588p-16,                    \
        0x1.a01a01a7c2f89p-13, 0x1.6c16c17889fd3p-10, 0x1.11111111109b5p-7, 0x1.5555555553d68p-5,                      \
        0x1.5555555555556p-3, 0x1.0000000000001p-1                                                                     \
  }
#if defined(MPCX_FM_DEVICE_TABLE)
__constant__ double mpcx_fm_tab[26] = MPCX_FM_TABLE_INIT;
#else
static const double mpcx_fm_tab[26] = MPCX_FM_TABLE_INIT;
#endif
#define MPCX_FM_K(i) mpcx_fm_tab[i]

MPCX_UFCX_MATH_FN double mpcx_fm_flip(double v, int odd)
{
  unsigned long long u;
  __builtin_memcpy(&u, &v, 8);
  u ^= (unsigned long long)((unsigned)odd & 1u) << 63;
  __builtin_memcpy(&v, &u, 8);
  return v;
}

// sin(r) for |r| <= pi/2 (+ a few ulp): r + r^3 P(r^2), eight coefficients
MPCX_UFCX_MATH_FN double mpcx_fm_sin_poly(double r)
{
  const double r2 = r * r;
  double p = MPCX_FM_K(0);
  p = __builtin_fma(p, r2, MPCX_FM_K(1));
  p = __builtin_fma(p, r2, MPCX_FM_K(2));
  p = __builtin_fma(p, r2, MPCX_FM_K(3));
  p = __builtin_fma(p, r2, MPCX_FM_K(4));
  p = __builtin_fma(p, r2, MPCX_FM_K(5));
  p = __builtin_fma(p, r2, MPCX_FM_K(6));
  p = __builtin_fma(p, r2, MPCX_FM_K(7));
  return __builtin_fma(r * r2, p, r);
}

// x - m * pi, pi in three parts of 33 + 33 + 53 bits (twice fdlibm's pio2_1, pio2_2, pio2_3): m * part is exact for |2 m| < 2^20
MPCX_UFCX_MATH_FN double mpcx_fm_reduce(double x, double m)
{
  double r = __builtin_fma(-m, MPCX_FM_K(8), x);
  r = __builtin_fma(-m, MPCX_FM_K(9), r);
  r = __builtin_fma(-m, MPCX_FM_K(10), r);
  return r;
}

MPCX_UFCX_MATH_FN double mpcx_fast_sin(double x)
{
  if (!(__builtin_fabs(x) <= MPCX_FM_K(12))) // beyond 2^19 pi, inf, NaN
    return MPCX_FM_LIBM_SIN(x);
  const double n = __builtin_rint(x * MPCX_FM_K(11)); // x / pi
  const double r = mpcx_fm_reduce(x, n);
  return mpcx_fm_flip(mpcx_fm_sin_poly(r), (int)n);
}

MPCX_UFCX_MATH_FN double mpcx_fast_cos(double x)
{
  if (!(__builtin_fabs(x) <= MPCX_FM_K(12)))
    return MPCX_FM_LIBM_COS(x);
  // x = (n + 1/2) pi + r:  cos(x) = (-1)^(n+1) sin(r)
  const double n = __builtin_rint(__builtin_fma(x, MPCX_FM_K(11), -0.5));
  const double r = mpcx_fm_reduce(x, n + 0.5);
  return mpcx_fm_flip(mpcx_fm_sin_poly(r), (int)n + 1);
}

MPCX_UFCX_MATH_FN double mpcx_fast_exp(double x)
{
  if (!(x >= -708.0 && x <= 709.0))
    return MPCX_FM_LIBM_EXP(x);
  const double n = __builtin_rint(x * MPCX_FM_K(13));
  double r = __builtin_fma(-n, MPCX_FM_K(14), x); // ln2 high part (33 bits): exact
  r = __builtin_fma(-n, MPCX_FM_K(15), r);
  double p = MPCX_FM_K(16);
  p = __builtin_fma(p, r, MPCX_FM_K(17));
  p = __builtin_fma(p, r, MPCX_FM_K(18));
  p = __builtin_fma(p, r, MPCX_FM_K(19));
  p = __builtin_fma(p, r, MPCX_FM_K(20));
  p = __builtin_fma(p, r, MPCX_FM_K(21));
  p = __builtin_fma(p, r, MPCX_FM_K(22));
  p = __builtin_fma(p, r, MPCX_FM_K(23));
  p = __builtin_fma(p, r, MPCX_FM_K(24));
  p = __builtin_fma(p, r, MPCX_FM_K(25));
  // e^r = 1 + r + r^2 Q(r)
  const double v = __builtin_fma(r * r, p, r) + 1.0;
  return __builtin_ldexp(v, (int)n);
}
