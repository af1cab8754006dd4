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

// sin(pi * t), |t| < 2^30.  r = t - rint(t) in [-1/2, 1/2] exactly, one odd
// polynomial in r (Taylor coefficients (-1)^k pi^(2k+1)/(2k+1)!, truncation
// < 2e-18), leading term split as r*PI_HI + r*PI_LO: ~17 fp64 instructions
// against ~32 for a quarter-range sin/cos pair.
MPCX_HD inline double fast_sinpi(double t)
{
  const double n = std::rint(t);
  const double r = t - n; // exact
  const double r2 = r * r;
  double p = 0x1.2877020d52cf0p-31;       // pi^21/21!
  p = std::fma(p, r2, -0x1.8a404211f9547p-26);
  p = std::fma(p, r2, 0x1.aaec32af93359p-21);
  p = std::fma(p, r2, -0x1.6fadb9f155744p-16);
  p = std::fma(p, r2, 0x1.e8f434d018d63p-12);
  p = std::fma(p, r2, -0x1.e3074fde8871fp-8);
  p = std::fma(p, r2, 0x1.50783487ee782p-4);
  p = std::fma(p, r2, -0x1.32d2cce62bd86p-1);
  p = std::fma(p, r2, 0x1.466bc6775aae2p+1);
  p = std::fma(p, r2, -0x1.4abbce625be53p+2); // -pi^3/6
  const double PI_HI = 0x1.921fb54442d18p+1, PI_LO = 1.2246467991473532e-16;
  const double tl = std::fma(r2, p, PI_LO);
  const double v = std::fma(r, PI_HI, r * tl);
  return (static_cast<long long>(n) & 1) ? -v : v;
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
