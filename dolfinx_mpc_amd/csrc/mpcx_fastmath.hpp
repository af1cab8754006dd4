// fp64 sin(pi t) and exp(y) for the analytic right-hand sides: exact range
// reduction + short Taylor polynomials (< 2 ulp on the reduced ranges; accuracy
// is checked on the host by tests/test_fastmath_host.py, which compiles this
// header with g++).  The library versions spend most of their instructions on
// the general range reduction; the right-hand side of the periodic benchmark
// (python/benchmarks/bench_periodic.py:85-89) evaluates 14 sin + 14 exp per
// cell, which makes the vector kernel VALU-bound otherwise.
#pragma once
#include <cmath>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
#ifndef MPCX_HD
#if defined(__HIPCC__)
#define MPCX_HD __host__ __device__
#else
#define MPCX_HD
#endif
#endif

namespace mpcx
{

// bit helpers (__builtin_memcpy compiles to register moves on host and device)
MPCX_HD inline double flip_sign(double v, unsigned sign_in_bit31)
{
  unsigned long long u;
  __builtin_memcpy(&u, &v, 8);
  u ^= static_cast<unsigned long long>(sign_in_bit31 & 0x80000000u) << 32;
  __builtin_memcpy(&v, &u, 8);
  return v;
}
// v * 2^m for normal v and a normal result (|m| small enough): add m to the exponent field
MPCX_HD inline double scale_pow2(double v, int m)
{
  long long u;
  __builtin_memcpy(&u, &v, 8);
  u += static_cast<long long>(m) << 52;
  __builtin_memcpy(&v, &u, 8);
  return v;
}

// rint(t) and its low 32 bits as an integer, |t| < 2^31, by the add-the-magic-number trick: t + 1.5 * 2^52 rounds t to the
// nearest integer (ties to even, the default rounding mode) into the low mantissa bits, two full-rate fp64 adds instead of
// v_rndne_f64 + v_cvt_i32_f64 (round 5: 1.4 % fewer VALU instructions in the cluster vector kernel of config 2, 2.78 against
// 2.79-2.82 ms -- the conversions were not the slow instructions the pipe-bound kernel was suspected of)
MPCX_HD inline double rint_magic(double t, int& k)
{
  const double MAGIC = 0x1.8p52;
  const double tn = t + MAGIC;
  unsigned long long u;
  __builtin_memcpy(&u, &tn, 8);
  k = static_cast<int>(static_cast<unsigned>(u));
  return tn - MAGIC;
}

// sin(pi * t), |t| < 2^30.  r = t - rint(t) in [-1/2, 1/2] exactly, one odd
// polynomial in r (Taylor coefficients (-1)^k pi^(2k+1)/(2k+1)!, truncation
// < 2e-18), leading term split as r*PI_HI + r*PI_LO: ~17 fp64 instructions
// against ~32 for a quarter-range sin/cos pair.
MPCX_HD inline double fast_sinpi(double t)
{
  int k;
  const double n = rint_magic(t, k);
  const double r = t - n; // exact
  const double r2 = r * r;
  // near-minimax fit of (sin(pi r) / r - pi) / r^2 in r^2 on |r| <= 1/2, degree 7 (absolute error 1.3e-18 in sin(pi r)): two
  // Horner steps less than the Taylor polynomial of degree 21
  double p = 0x1.9ec5cd6e85639p-21;
  p = std::fma(p, r2, -0x1.6f866b6ea7cc5p-16);
  p = std::fma(p, r2, 0x1.e8f3b0121051dp-12);
  p = std::fma(p, r2, -0x1.e3074ee5f48c3p-8);
  p = std::fma(p, r2, 0x1.50783486f190ap-4);
  p = std::fma(p, r2, -0x1.32d2cce62adb9p-1);
  p = std::fma(p, r2, 0x1.466bc6775aad6p+1);
  p = std::fma(p, r2, -0x1.4abbce625be53p+2); // ~ -pi^3/6
  const double PI_HI = 0x1.921fb54442d18p+1, PI_LO = 1.2246467991473532e-16;
  const double tl = std::fma(r2, p, PI_LO);
  const double v = std::fma(r, PI_HI, r * tl);
  // odd n: flip the sign bit (integer xor on the high word instead of a compare/select pair)
  return flip_sign(v, static_cast<unsigned>(k) << 31);
}

// 2^(j/64), j = 0..63, correctly rounded
static constexpr double EXP2_64[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};

#if defined(__HIPCC__)
// On the GPU the table is read from LDS (a global-memory gather per evaluation exposes a
// cache round trip per quadrature point: measured 10% slower than no table at all).  Every
// kernel that evaluates fast_exp calls fastmath_init_lds() first (all threads, ends in a barrier).
__device__ inline double* exp2_table_lds()
{
  __shared__ double t[64];
  return t;
}
__device__ inline void fastmath_init_lds()
{
  if (threadIdx.x < 64)
    exp2_table_lds()[threadIdx.x] = EXP2_64[threadIdx.x];
  __syncthreads();
}
#endif

// exp(y) = 2^m * 2^(j/64) * e^r with y = (64 m + j) ln2/64 + r, |r| <= ln2/128: table value T
// times a degree-5 Taylor polynomial (truncation r^6/720 < 4e-17), assembled as T + T*q with
// q = e^r - 1 small, so the rounding of q is damped by |r| -- ~1 ulp, 9 fp64 instructions
// against 17 for the degree-13 polynomial on |r| <= ln2/2.  Underflows to 0 and
// overflows to inf through ldexp.
MPCX_HD inline double fast_exp(double y)
{
  const double INV = 0x1.71547652b82fep+6;                                     // 64/ln2
  const double L_HI = 0x1.62e42fee00000p-7, L_LO = 0x1.a39ef35793c76p-39;     // ln2/64, HI has 32 bits
  const bool under = y < -745.2; // below the smallest denormal
  y = under ? -745.2 : (y > 710.0 ? 710.0 : y);
  const double n = std::rint(y * INV);
  double r = std::fma(-n, L_HI, y); // exact
  r = std::fma(-n, L_LO, r);
  const int k = static_cast<int>(n);
#if defined(__HIP_DEVICE_COMPILE__)
  const double T = exp2_table_lds()[k & 63];
#else
  const double T = EXP2_64[k & 63];
#endif
  double p = 1.0 / 120.0;
  p = std::fma(p, r, 1.0 / 24.0);
  p = std::fma(p, r, 1.0 / 6.0);
  p = std::fma(p, r, 0.5);
  p = std::fma(p, r, 1.0);
  const double v = std::fma(T, p * r, T);
  return under ? 0.0 : std::ldexp(v, k >> 6);
}

// exp(y) for y <= 0 (Gaussians): one clamp, and 2^m goes into the exponent field of the table
// value with an integer add.  Results below 2^-1021 (y < -707.7) are flushed to 0 instead of
// going through the denormals; otherwise identical to fast_exp.
MPCX_HD inline double fast_exp_nonpos(double y)
{
  const double INV = 0x1.71547652b82fep+6;
  const double L_HI = 0x1.62e42fee00000p-7, L_LO = 0x1.a39ef35793c76p-39;
  y = std::fmax(y, -708.0); // one v_max_f64; exp(-708) = 3.3e-308, still normal after the scaling below
  int k;
  const double n = rint_magic(y * INV, k);
  double r = std::fma(-n, L_HI, y);
  r = std::fma(-n, L_LO, r);
#if defined(__HIP_DEVICE_COMPILE__)
  const double T = scale_pow2(exp2_table_lds()[k & 63], k >> 6);
#else
  const double T = scale_pow2(EXP2_64[k & 63], k >> 6);
#endif
  double p = 1.0 / 120.0;
  p = std::fma(p, r, 1.0 / 24.0);
  p = std::fma(p, r, 1.0 / 6.0);
  p = std::fma(p, r, 0.5);
  p = std::fma(p, r, 1.0);
  return std::fma(T, p * r, T);
}

#if defined(__HIPCC__)
// The same two functions with their coefficients read from __constant__ memory into scalar registers ONCE
// per kernel call instead of being re-materialised as literals in front of every fma (an fp64 literal costs
// two s_mov + a v_mov per use: measured 17 of ~75 VALU instructions per quadrature point of the benchmark's
// right-hand side).  v_fma_f64 takes one SGPR pair as an operand, so Horner steps become single instructions.
struct FmConsts
{
  double s[8];         // sinpi: near-minimax coefficients (fast_sinpi), highest first
  double pi_hi, pi_lo;
  double inv, l_hi, l_lo; // exp: 64/ln2, ln2/64 split
  double e[5];         // exp: 1/120, 1/24, 1/6, 1/2, 1
};
__constant__ FmConsts g_fm_consts = {
    {0x1.9ec5cd6e85639p-21, -0x1.6f866b6ea7cc5p-16, 0x1.e8f3b0121051dp-12, -0x1.e3074ee5f48c3p-8, 0x1.50783486f190ap-4,
     -0x1.32d2cce62adb9p-1, 0x1.466bc6775aad6p+1, -0x1.4abbce625be53p+2},
    0x1.921fb54442d18p+1, 1.2246467991473532e-16,
    0x1.71547652b82fep+6, 0x1.62e42fee00000p-7, 0x1.a39ef35793c76p-39,
    {1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0}};

__device__ inline double fast_sinpi_k(double t, const FmConsts& K)
{
  int k;
  const double n = rint_magic(t, k);
  const double r = t - n;
  const double r2 = r * r;
  double p = K.s[0];
#pragma unroll
  for (int i = 1; i < 8; ++i)
    p = fma(p, r2, K.s[i]);
  const double tl = fma(r2, p, K.pi_lo);
  const double v = fma(r, K.pi_hi, r * tl);
  return flip_sign(v, static_cast<unsigned>(k) << 31);
}

__device__ inline double fast_exp_nonpos_k(double y, const FmConsts& K)
{
  y = fmax(y, -708.0);
  int k;
  const double n = rint_magic(y * K.inv, k);
  double r = fma(-n, K.l_hi, y);
  r = fma(-n, K.l_lo, r);
  const double T = scale_pow2(exp2_table_lds()[k & 63], k >> 6);
  double p = K.e[0];
#pragma unroll
  for (int i = 1; i < 5; ++i)
    p = fma(p, r, K.e[i]);
  return fma(T, p * r, T);
}
#endif

} // namespace mpcx
