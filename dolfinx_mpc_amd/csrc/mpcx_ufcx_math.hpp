// fp64 sin / cos / exp for IMPORTED element kernels (csrc/mpcx_ufcx.cpp puts this text in front of the user's C source and
// maps the libm names onto it: FFCx-generated tabulate_tensor functions call sin / cos / exp at every quadrature point, and
// the device libm spends 189 instructions on a sin, 62 on an exp -- general Payne-Hanek reduction, denormal and special-case
// paths inlined into every call).  Valid on the FULL double range: arguments outside the fast range, infinities and NaNs
// take the libm function (a rarely taken branch).  Accuracy (tests/test_fastmath_host.py compiles this header with g++
// and compares with long double libm on dense and random arguments):
//   mpcx_fast_sin / mpcx_fast_cos   |x| <= 2^19 pi: <= 2 ulp (Cody-Waite reduction to [-pi/2, pi/2] with a three-part
//                                   pi/2, odd Taylor polynomial of degree 21); beyond: libm
//   mpcx_fast_exp                   -708 <= x <= 709: <= 2 ulp (x = n ln2 + r, |r| <= ln2 / 2, Taylor degree 13, ldexp);
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
    /* 0..9: sin Taylor, -1/21! .. 1/3! (alternating) */                                                               \
    -1.9572941063391263e-20, 8.2206352466243295e-18, -2.8114572543455206e-15, 7.6471637318198164e-13,                  \
        -1.6059043836821613e-10, 2.5052108385441720e-08, -2.7557319223985893e-06, 1.9841269841269841e-04,              \
        -8.3333333333333332e-03, 1.6666666666666666e-01, /* 10..12: pi in three parts; 13: 1/pi; 14: 2^19 pi */           \
        2.0 * 1.57079632673412561417e+00, 2.0 * 6.07710050630396597660e-11, 2.0 * 2.02226624879595063154e-21,           \
        3.18309886183790671538e-01, 1647099.0, /* 15: 1/ln2; 16, 17: ln2 in two parts; 18..29: exp Taylor 1/13! .. 1/2! */ \
        1.44269504088896338700e+00, 6.93147180369123816490e-01, 1.90821492927058770002e-10, 1.6059043836821613e-10,     \
        2.0876756987868099e-09, 2.5052108385441720e-08, 2.7557319223985888e-07, 2.7557319223985893e-06,                 \
        2.4801587301587302e-05, 1.9841269841269841e-04, 1.3888888888888889e-03, 8.3333333333333332e-03,                 \
        4.1666666666666664e-02, 1.6666666666666666e-01, 0.5                                                             \
  }
#if defined(MPCX_FM_DEVICE_TABLE)
__constant__ double mpcx_fm_tab[30] = MPCX_FM_TABLE_INIT;
#else
static const double mpcx_fm_tab[30] = MPCX_FM_TABLE_INIT;
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

// sin(r) for |r| <= pi/2 (+ a few ulp): r - r^3 p(r^2), Taylor to degree 21 (x^23/23! < 2e-19 at pi/2)
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
  p = __builtin_fma(p, r2, MPCX_FM_K(8));
  p = __builtin_fma(p, r2, MPCX_FM_K(9));
  return __builtin_fma(-(r * r2), p, r);
}

// x - m * pi, pi in three parts of 33 + 33 + 53 bits (twice fdlibm's pio2_1, pio2_2, pio2_3): m * part is exact for |2 m| < 2^20
MPCX_UFCX_MATH_FN double mpcx_fm_reduce(double x, double m)
{
  double r = __builtin_fma(-m, MPCX_FM_K(10), x);
  r = __builtin_fma(-m, MPCX_FM_K(11), r);
  r = __builtin_fma(-m, MPCX_FM_K(12), r);
  return r;
}

MPCX_UFCX_MATH_FN double mpcx_fast_sin(double x)
{
  if (!(__builtin_fabs(x) <= MPCX_FM_K(14))) // beyond 2^19 pi, inf, NaN
    return MPCX_FM_LIBM_SIN(x);
  const double n = __builtin_rint(x * MPCX_FM_K(13)); // x / pi
  const double r = mpcx_fm_reduce(x, n);
  return mpcx_fm_flip(mpcx_fm_sin_poly(r), (int)n);
}

MPCX_UFCX_MATH_FN double mpcx_fast_cos(double x)
{
  if (!(__builtin_fabs(x) <= MPCX_FM_K(14)))
    return MPCX_FM_LIBM_COS(x);
  // x = (n + 1/2) pi + r:  cos(x) = (-1)^(n+1) sin(r)
  const double n = __builtin_rint(__builtin_fma(x, MPCX_FM_K(13), -0.5));
  const double r = mpcx_fm_reduce(x, n + 0.5);
  return mpcx_fm_flip(mpcx_fm_sin_poly(r), (int)n + 1);
}

MPCX_UFCX_MATH_FN double mpcx_fast_exp(double x)
{
  if (!(x >= -708.0 && x <= 709.0))
    return MPCX_FM_LIBM_EXP(x);
  const double n = __builtin_rint(x * MPCX_FM_K(15));
  double r = __builtin_fma(-n, MPCX_FM_K(16), x); // ln2 high part (33 bits): exact
  r = __builtin_fma(-n, MPCX_FM_K(17), r);
  double p = MPCX_FM_K(18);
  p = __builtin_fma(p, r, MPCX_FM_K(19));
  p = __builtin_fma(p, r, MPCX_FM_K(20));
  p = __builtin_fma(p, r, MPCX_FM_K(21));
  p = __builtin_fma(p, r, MPCX_FM_K(22));
  p = __builtin_fma(p, r, MPCX_FM_K(23));
  p = __builtin_fma(p, r, MPCX_FM_K(24));
  p = __builtin_fma(p, r, MPCX_FM_K(25));
  p = __builtin_fma(p, r, MPCX_FM_K(26));
  p = __builtin_fma(p, r, MPCX_FM_K(27));
  p = __builtin_fma(p, r, MPCX_FM_K(28));
  p = __builtin_fma(p, r, MPCX_FM_K(29));
  // e^r = 1 + r + r^2 p
  const double v = __builtin_fma(r * r, p, r) + 1.0;
  return __builtin_ldexp(v, (int)n);
}
