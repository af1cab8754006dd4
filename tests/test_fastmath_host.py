"""Accuracy of the device math helpers (csrc/mpcx_fastmath.hpp), checked on the
host: the header is plain C++ (MPCX_HD expands to nothing under g++), so the
same code is compiled with g++ and compared with libm.  Bars: <= 2 ulp for
sin(pi t) relative to max(|value|, tiny) on the benchmark's argument range, and
< 2 ulp for exp on [-200, 5]."""

import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include "mpcx_fastmath.hpp"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv)
{
  int which = atoi(argv[1]);
  double lo = atof(argv[2]), hi = atof(argv[3]);
  int n = atoi(argv[4]);
  for (int i = 0; i < n; ++i)
  {
    double t = lo + (hi - lo) * ((i + 0.37) / n);
    double v = which == 0 ? mpcx::fast_sinpi(t) : (which == 1 ? mpcx::fast_exp(t) : mpcx::fast_exp_nonpos(t));
    printf("%.17g %.17g\n", t, v);
  }
  return 0;
}
"""


def _run(tmp_path, which, lo, hi, n):
    exe = os.path.join(str(tmp_path), "fm")
    if not os.path.exists(exe):
        src = os.path.join(str(tmp_path), "fm.cpp")
        open(src, "w").write(SRC)
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "dolfinx_mpc_amd", "csrc"),
                        src, "-o", exe], check=True)
    out = subprocess.run([exe, str(which), repr(lo), repr(hi), str(n)], check=True, capture_output=True, text=True).stdout
    a = np.array(out.split(), dtype=np.float64).reshape(-1, 2)
    return a[:, 0], a[:, 1]


def test_fast_sinpi(tmp_path):
    t, v = _run(tmp_path, 0, -3.0, 9.0, 200001)
    import mpmath

    mpmath.mp.dps = 40
    ref = np.array([float(mpmath.sinpi(mpmath.mpf(x))) for x in t[::97]])
    err = np.abs(v[::97] - ref)
    ulp = np.spacing(np.maximum(np.abs(ref), 1e-3))
    assert (err / ulp).max() <= 2.0, (err / ulp).max()
    # against libm over the whole sample (libm itself is ~1 ulp on sin(pi*t) through the rounded argument)
    assert np.abs(v - np.sin(np.pi * t)).max() < 4e-15
    # exact zeros / ones at half-integers
    t2, v2 = _run(tmp_path, 0, 0.0, 8.0, 16)
    # sample points are t = 8*(i+0.37)/16, not special; check a few special values through python
    for x, want in ((0.0, 0.0), (0.5, 1.0), (1.0, 0.0), (1.5, -1.0), (2.0, 0.0), (-0.5, -1.0)):
        tt, vv = _run(tmp_path, 0, x - 0.37 * 1e-300, x - 0.37 * 1e-300 + 1e-300, 1)
        assert abs(vv[0] - want) < 1e-15


def test_fast_exp(tmp_path):
    t, v = _run(tmp_path, 1, -200.0, 5.0, 200001)
    import mpmath

    mpmath.mp.dps = 40
    ref = np.array([float(mpmath.exp(mpmath.mpf(x))) for x in t[::97]])
    rel = np.abs(v[::97] - ref) / ref
    assert rel.max() < 2.0 * np.finfo(np.float64).eps, rel.max()
    assert (np.abs(v - np.exp(t)) / np.exp(t)).max() < 4.5e-16
    # deep underflow goes to zero, no NaN
    t3, v3 = _run(tmp_path, 1, -2000.0, -800.0, 50)
    assert np.all(v3 == 0.0)


def test_fast_exp_nonpos(tmp_path):
    """the Gaussian variant (y <= 0, exponent spliced into the table value): same accuracy down to
    the flush threshold, exact 0-free tail, 1 at 0"""
    t, v = _run(tmp_path, 2, -700.0, 0.0, 200001)
    import mpmath

    mpmath.mp.dps = 40
    ref = np.array([float(mpmath.exp(mpmath.mpf(x))) for x in t[::97]])
    rel = np.abs(v[::97] - ref) / ref
    assert rel.max() < 2.0 * np.finfo(np.float64).eps, rel.max()
    assert (np.abs(v - np.exp(t)) / np.exp(t)).max() < 4.5e-16
    t2, v2 = _run(tmp_path, 2, -5000.0, -709.0, 40)
    assert np.all((v2 >= 0.0) & (v2 < 1e-307))
    t3, v3 = _run(tmp_path, 2, -1e-300, 0.0, 3)
    assert np.all(v3 == 1.0)


UFCX_SRC = r"""
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "mpcx_ufcx_math.hpp"
static double ulps(double got, long double ref)
{
  if (ref == 0)
    return got == 0 ? 0 : 1e9;
  int e;
  std::frexp((double)fabsl(ref), &e);
  return (double)(fabsl((long double)got - ref) / std::ldexp(1.0L, e - 53));
}
int main()
{
  double ms = 0, mc = 0, me = 0;
  srand(1);
  for (int i = 0; i < 2000000; ++i)
  {
    const double t = rand() / (double)RAND_MAX;
    const double x = (i % 4 == 0) ? (t - 0.5) * 20 : (i % 4 == 1) ? (t - 0.5) * 2e3 : (i % 4 == 2) ? (t - 0.5) * 3.2e6 : (t - 0.5) * 1e-3;
    ms = fmax(ms, ulps(mpcx_fast_sin(x), sinl((long double)x)));
    mc = fmax(mc, ulps(mpcx_fast_cos(x), cosl((long double)x)));
    const double y = (i % 2) ? (t - 0.5) * 1400 : (t - 0.5) * 20;
    me = fmax(me, ulps(mpcx_fast_exp(y), expl((long double)y)));
  }
  for (int k = 1; k < 200000; k += 7) // next to the zeros of sin and cos
    for (int d = -2; d <= 2; ++d)
    {
      double x = k * 3.14159265358979323846;
      for (int s = 0; s < (d < 0 ? -d : d); ++s)
        x = std::nextafter(x, d > 0 ? 1e300 : -1e300);
      ms = fmax(ms, ulps(mpcx_fast_sin(x), sinl((long double)x)));
      const double xc = x + 1.5707963267948966;
      mc = fmax(mc, ulps(mpcx_fast_cos(xc), cosl((long double)xc)));
    }
  printf("%.4f %.4f %.4f\n", ms, mc, me);
  // outside the fast range: libm's answers, bit for bit
  const double big[] = {1e7, -3.3e9, 1e300, 2e6};
  int same = 1;
  for (double x : big)
    same = same && mpcx_fast_sin(x) == std::sin(x) && mpcx_fast_cos(x) == std::cos(x);
  same = same && mpcx_fast_exp(-720.0) == std::exp(-720.0) && mpcx_fast_exp(709.5) == std::exp(709.5) && std::isinf(mpcx_fast_exp(800.0))
         && mpcx_fast_exp(-800.0) == 0.0 && std::isnan(mpcx_fast_sin(NAN)) && std::isnan(mpcx_fast_exp(NAN)) && std::isnan(mpcx_fast_cos(INFINITY));
  printf("%d\n", same);
  printf("%.17g %.17g %.17g\n", mpcx_fast_sin(0.0), mpcx_fast_cos(0.0), mpcx_fast_exp(0.0));
  return 0;
}
"""


def test_imported_kernel_math_full_range(tmp_path):
    """csrc/mpcx_ufcx_math.hpp -- the sin / cos / exp that imported (FFCx-shaped) kernels get instead of the device libm:
    <= 2.5 ulp for sin / cos on |x| <= 2^19 pi including the neighbourhoods of their zeros, <= 1.5 ulp for exp on
    [-700, 700]; outside the fast ranges, and for NaN / inf, the libm result itself"""
    src = os.path.join(str(tmp_path), "um.cpp")
    exe = os.path.join(str(tmp_path), "um")
    open(src, "w").write(UFCX_SRC)
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "dolfinx_mpc_amd", "csrc"), src, "-o", exe],
                   check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split("\n")
    ms, mc, me = (float(v) for v in out[0].split())
    assert ms <= 2.5 and mc <= 2.5 and me <= 1.5, (ms, mc, me)
    assert out[1].strip() == "1"
    s0, c0, e0 = (float(v) for v in out[2].split())
    assert s0 == 0.0 and c0 == 1.0 and e0 == 1.0
