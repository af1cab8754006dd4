// fp64 sin(pi t) and exp(y) for the analytic right-hand sides: exact range
// reduction + short Taylor polynomials (< 2 ulp on the reduced ranges; accuracy
// is checked on the host by tests/test_fastmath_host.py, which compiles this
// header with g++).  The library versions spend most of their instructions on
// the general range reduction; the right-hand side of the periodic benchmark
// (python/benchmarks/bench_periodic.py:85-89) evaluates 14 sin + 14 exp per
// cell, which makes the vector kernel VALU-bound otherwise.
#pragma once
#include <cmath>

#ifndef MPCX_HD
#if defined(__HIPCC__)
#define MPCX_HD __host__ __device__
#else
#define MPCX_HD
#endif
#endif

namespace mpcx
{

// sin(pi * t), |t| < 2^30
MPCX_HD inline double fast_sinpi(double t)
{
  const double n = std::rint(2.0 * t);
  const double r = std::fma(-0.5, n, t); // exact, |r| <= 1/4
  const int q = static_cast<int>(n);
  // x = pi * r in two pieces
  const double PI_HI = 3.141592653589793116, PI_LO = 1.2246467991473532e-16;
  const double x = r * PI_HI;
  const double xlo = std::fma(r, PI_HI, -x) + r * PI_LO;
  const double x2 = x * x;
  // sin(x)/x and cos(x), |x| <= pi/4
  double s = -1.0 / 1307674368000.0;
  s = std::fma(s, x2, 1.0 / 6227020800.0);
  s = std::fma(s, x2, -1.0 / 39916800.0);
  s = std::fma(s, x2, 1.0 / 362880.0);
  s = std::fma(s, x2, -1.0 / 5040.0);
  s = std::fma(s, x2, 1.0 / 120.0);
  s = std::fma(s, x2, -1.0 / 6.0);
  s = std::fma(s * x2, x, x); // x + x^3 * (...)
  double c = 1.0 / 20922789888000.0;
  c = std::fma(c, x2, -1.0 / 87178291200.0);
  c = std::fma(c, x2, 1.0 / 479001600.0);
  c = std::fma(c, x2, -1.0 / 3628800.0);
  c = std::fma(c, x2, 1.0 / 40320.0);
  c = std::fma(c, x2, -1.0 / 720.0);
  c = std::fma(c, x2, 1.0 / 24.0);
  c = std::fma(c, x2, -0.5);
  c = std::fma(c, x2, 1.0);
  // first-order correction for the low part of x
  const double sv = std::fma(xlo, c, s);
  const double cv = std::fma(-xlo, s, c);
  const double v = (q & 1) ? cv : sv;
  return (q & 2) ? -v : v;
}

// exp(y); underflows to 0 / overflows to inf through ldexp
MPCX_HD inline double fast_exp(double y)
{
  const double L2E = 1.44269504088896338700e+00;
  const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  const bool under = y < -745.2; // below the smallest denormal
  y = under ? -745.2 : (y > 710.0 ? 710.0 : y);
  const double n = std::rint(y * L2E);
  double r = std::fma(-n, LN2_HI, y);
  r = std::fma(-n, LN2_LO, r); // |r| <= ln2/2
  double p = 1.0 / 6227020800.0; // 1/13!
  p = std::fma(p, r, 1.0 / 479001600.0);
  p = std::fma(p, r, 1.0 / 39916800.0);
  p = std::fma(p, r, 1.0 / 3628800.0);
  p = std::fma(p, r, 1.0 / 362880.0);
  p = std::fma(p, r, 1.0 / 40320.0);
  p = std::fma(p, r, 1.0 / 5040.0);
  p = std::fma(p, r, 1.0 / 720.0);
  p = std::fma(p, r, 1.0 / 120.0);
  p = std::fma(p, r, 1.0 / 24.0);
  p = std::fma(p, r, 1.0 / 6.0);
  p = std::fma(p, r, 0.5);
  p = std::fma(p, r, 1.0);
  p = std::fma(p, r, 1.0);
  return under ? 0.0 : std::ldexp(p, static_cast<int>(n));
}

} // namespace mpcx
