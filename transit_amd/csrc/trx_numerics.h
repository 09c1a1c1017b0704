// trx_numerics.h -- small numerical building blocks shared by the host side
// (g++) and the HIP kernels (hipcc).  Each routine reproduces the arithmetic of
// the reference routine it names, operation for operation, so that discrete
// decisions (nearest index, bracketing) and quadrature weights agree with it.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TRX_HD __host__ __device__ __forceinline__
#else
#define TRX_HD inline
#endif

namespace trx {

// physical constants, values of transit/include/constants_tr.h:35-63 (cgs)
constexpr double kPi      = 3.141592653589793;
constexpr double kAmu     = 1.66053886e-24;
constexpr double kEc      = 4.8032068e-10;
constexpr double kLs      = 2.99792458e10;
constexpr double kMe      = 9.1093897e-28;
constexpr double kKb      = 1.380658e-16;
constexpr double kH       = 6.6260755e-27;
constexpr double kAmagat  = 2.68678e19;
constexpr double kSigCte  = kPi*kEc*kEc/kLs/kLs/kMe/kAmu;
constexpr double kExpCte  = kH*kLs/kKb;
constexpr double kSqrtLn2 = 0.83255461115769775635;
constexpr double kDeg     = kPi/180.0;
constexpr double kNavo    = 6.02214076e23;
constexpr double kE0H2    = 4.911e-23;
constexpr double kMicron  = 1e-4;
constexpr double kSunRadius = 6.957e10;
constexpr double kTliWfct = 1e-4;   // readlineinfo.c:6
constexpr double kTliEfct = 1.0;    // readlineinfo.c:7

// pu/src/iomisc.c:1088-1108 (binsearchapprox): nearest element by the
// reference's bisection; lo/hi are inclusive search bounds as passed there.
TRX_HD int nearest_index(const double *a, double v, int lo, int hi)
{
  while (hi - lo > 1) {
    const int mid = (hi + lo) / 2;
    if (a[mid] > v) hi = mid; else lo = mid;
  }
  if (hi - lo == 1)
    return (fabs(a[hi] - v) < fabs(a[lo] - v)) ? hi : lo;
  return lo;
}

// pu/src/numerical.c:16-45 (binsearchie)
TRX_HD int bracket_ie(const double *a, long i, long f, double v)
{
  if (a[i] > v) return -1;
  if (a[f] < v) return -2;
  if (a[f] == v) return -5;
  if (i == f && a[i] != v) return -3;
  while (f - i > 1) {
    const long m = (f + i) >> 1;
    if (a[m] > v) f = m; else i = m;
  }
  return (int)i;
}

// Parabola through three nodes taken as equispaced (only the first spacing is used), in the
// coordinates t = x/step: the restatement of pu/src/numerical.c:182-195 (interp_parab) with
// scalar arguments.  The operations and their order are the reference's on purpose -- the
// result's rounding noise (tests/tolerances.py) is part of what parity is measured against.
//   bend = second difference of the values; the three polynomial coefficients follow.
TRX_HD double parab3(double node0, double node1, double v0, double v1, double v2, double at)
{
  const double step = node1 - node0;
  const double t0   = node0 / step;
  const double bend = v0 + v2 - 2*v1;
  const double quad = bend / (2.0 * step * step);
  const double lin  = (v2 - v1 - (t0 + 1.5) * bend) / step;
  const double cst  = v0 + t0 * (v2 - 4*v1 + 3*v0 + t0 * bend) / 2.0;
  return at * at * quad + at * lin + cst;
}

// The same with the node-only terms handed in (they are the same for every wavenumber of a layer:
// a kernel computes them once per layer, off its per-ray dependency chain).  step = node1 - node0,
// t0 = node0 / step, twice_sq = 2.0 * step * step: the very expressions of parab3, so the same bits.
TRX_HD double parab3_nodes(double step, double t0, double twice_sq, double v0, double v1, double v2, double at)
{
  const double bend = v0 + v2 - 2*v1;
  const double quad = bend / twice_sq;
  const double lin  = (v2 - v1 - (t0 + 1.5) * bend) / step;
  const double cst  = v0 + t0 * (v2 - 4*v1 + 3*v0 + t0 * bend) / 2.0;
  return at * at * quad + at * lin + cst;
}

// x / d, correctly rounded, for a divisor whose reciprocal rd = 1.0 / d (itself a correctly
// rounded division) is known beforehand: two residual corrections of x * rd, five dependent
// multiply-adds instead of the ~13 instructions (one of them the slow reciprocal) of a division.
// After the first correction the quotient is within one unit of the last place; a residual
// x - q d taken exactly (fused) and applied with the correctly rounded reciprocal then lands on
// the rounded quotient itself (Markstein, IBM J. Res. Dev. 34, 1990, theorem 8.5; the exceptional
// divisors of that theorem are those whose reciprocal is not correctly rounded, excluded here).
// Holds while no intermediate leaves the normal range: |x| between ~1e-290 and ~1e290 or zero
// (an optical depth or extinction outside of it has no physical meaning; it comes out within a
// few units of the last denormal place instead).  tests/test_numerics.py checks 10^7 operands
// against the division, with the divisors the kernels use.
TRX_HD double quotient_rn(double x, double d, double rd)
{
  const double q0 = x * rd;
  const double q1 = __builtin_fma(__builtin_fma(-q0, d, x), rd, q0);
  return __builtin_fma(__builtin_fma(-q1, d, x), rd, q1);
}

// parab3_nodes with the two divisors' reciprocals handed in as well (r_step = 1.0 / step,
// r_twice_sq = 1.0 / twice_sq): the same bits again, without a division on the chain.
TRX_HD double parab3_recip(double step, double r_step, double t0, double twice_sq, double r_twice_sq,
                           double v0, double v1, double v2, double at)
{
  const double bend = v0 + v2 - 2*v1;
  const double quad = quotient_rn(bend, twice_sq, r_twice_sq);
  const double lin  = quotient_rn(v2 - v1 - (t0 + 1.5) * bend, step, r_step);
  const double cst  = v0 + t0 * (v2 - 4*v1 + 3*v0 + t0 * bend) / 2.0;
  return at * at * quad + at * lin + cst;
}

// parab3_recip with the two node-only factors of its polynomial handed in too (t0p = t0 + 1.5,
// at_sq = at * at: the very sub-expressions of parab3, so the same bits) -- a chain of parabolas,
// each fed by the one before, issues what is left and nothing else.
TRX_HD double parab3_chain(double step, double r_step, double t0, double t0p, double twice_sq, double r_twice_sq,
                           double v0, double v1, double v2, double at, double at_sq)
{
  const double bend = v0 + v2 - 2*v1;
  const double quad = quotient_rn(bend, twice_sq, r_twice_sq);
  const double lin  = quotient_rn(v2 - v1 - t0p * bend, step, r_step);
  const double cst  = v0 + t0 * (v2 - 4*v1 + 3*v0 + t0 * bend) / 2.0;
  return at_sq * quad + at * lin + cst;
}

// pu/src/spline.c:12-48 + 186-206 (tri / spline_init).  Scratch arrays u, v of
// length n are supplied by the caller (no allocation: usable in a kernel).
// Strided access so that a kernel can keep column-major scratch.
TRX_HD void spline_second_derivs(double *z, const double *x, const double *y, long n,
                                 double *u, double *v, long zs = 1, long xs = 1, long ys = 1,
                                 long us = 1)
{
  // The recurrences carry u[i-1], v[i-1], z[i+1] and the previous interval in
  // registers: same operations in the same order as the reference, without a
  // store->load round trip per step (the arrays may alias as far as the compiler knows).
  double up = 0, vp = 0;
  if (n > 2) {
    const double h0 = x[1*xs] - x[0], h1 = x[2*xs] - x[1*xs];
    const double b0 = (y[1*ys] - y[0]) / h0, b1 = (y[2*ys] - y[1*ys]) / h1;
    up = 2 * (h1 + h0);
    vp = 6 * (b1 - b0);
    u[1*us] = up;
    v[1*us] = vp;
  }
  if (n > 3) {
    double xi = x[2*xs], yi = y[2*ys];
    double him = xi - x[1*xs], bim = (yi - y[1*ys]) / him;
    for (long i = 2; i < n - 1; i++) {
      const double xn = x[(i+1)*xs], yn = y[(i+1)*ys];
      const double hi = xn - xi, bi = (yn - yi) / hi;
      const double un = 2*(hi + him) - him*him/up;
      const double vn = 6*(bi - bim) - vp*him/up;
      u[i*us] = un;
      v[i*us] = vn;
      up = un; vp = vn; him = hi; bim = bi; xi = xn; yi = yn;
    }
  }
  z[0] = 0; z[(n-1)*zs] = 0;
  if (n > 2) {
    double zn = 0, xn = x[(n-1)*xs];
    for (long i = n - 2; i > 0; i--) {
      const double xi = x[i*xs];
      const double hi = xn - xi;
      const double zi = (v[i*us] - hi*zn) / u[i*us];
      z[i*zs] = zi;
      zn = zi; xn = xi;
    }
  }
}

// bracketing rule shared by splinterp_pt and spline3 (spline.c:76-80, 151-155)
TRX_HD int spline_interval(const double *x, long n, double xo, long xs = 1)
{
  // nearest_index on a strided array
  int lo = 0, hi = (int)n - 1;
  while (hi - lo > 1) {
    const int mid = (hi + lo) / 2;
    if (x[mid*xs] > xo) hi = mid; else lo = mid;
  }
  int k = lo;
  if (hi - lo == 1) k = (fabs(x[hi*xs] - xo) < fabs(x[lo*xs] - xo)) ? hi : lo;
  if (k == n - 1 || xo < x[k*xs]) k--;
  return k;
}

// pu/src/spline.c:131-183 (splinterp_pt): Horner form
// (the evaluation in interval k on its own: a caller that evaluates many splines over the same
// abscissa at the same point locates the interval once)
TRX_HD double spline_eval_at(int k, const double *z, const double *x, const double *y, double xo,
                             long zs = 1, long xs = 1, long ys = 1);
TRX_HD double spline_eval_pt(const double *z, long n, const double *x, const double *y, double xo,
                             long zs = 1, long xs = 1, long ys = 1)
{
  return spline_eval_at(spline_interval(x, n, xo, xs), z, x, y, xo, zs, xs, ys);
}
TRX_HD double spline_eval_at(int k, const double *z, const double *x, const double *y, double xo,
                             long zs, long xs, long ys)
{
  const double xl = x[k*xs], xh = x[(k+1)*xs], yl = y[k*ys], yh = y[(k+1)*ys];
  const double h = xh - xl, dy = yh - yl;
  if (xl == xo) return yl;
  if (h > 0) {
    const double dx = xo - xl;
    const double zk = z[k*zs], zk1 = z[(k+1)*zs];
    const double a = (zk1 - zk) / (6*h);
    const double b = 0.5 * zk;
    const double c = dy/h - h/6 * (zk1 + 2*zk);
    return yl + dx*(c + dx*(b + dx*a));
  }
  return 0;
}

// pu/src/spline.c:55-91 (spline3): power form used by splinterp()
TRX_HD double spline_eval_pow(const double *z, long n, const double *x, const double *y, double xo)
{
  const int k = spline_interval(x, n, xo);
  const double h = x[k+1] - x[k];
  const double B = (y[k+1] - y[k]) / h - h/6 * (z[k+1] + 2 * z[k]);
  const double d = xo - x[k];
  // (at a node -- a retrieval loop resamples its atmosphere onto its own radii -- the two powers are
  // +0 without asking the library: pow(+0, 2) and pow(+0, 3) are exactly that)
  const double d2 = d == 0.0 ? 0.0 : pow(d, 2), d3 = d == 0.0 ? 0.0 : pow(d, 3);
  return y[k] + d * B + d2 * 0.5*z[k] + d3 * (z[k+1] - z[k]) / (6*h);
}

}  // namespace trx
