// Test program for trx::quotient_rn (transit_amd/csrc/trx_numerics.h): the residual-corrected
// product must be the division's own bits.  Built and run by tests/test_numerics.py.
#include "trx_numerics.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline double mant() { uint64_t m = (rnd() >> 12) | 0x3ff0000000000000ull; double d; memcpy(&d, &m, 8); return d; }
int main(int argc, char **argv)
{
  long n = argc > 1 ? atol(argv[1]) : 10000000, bad = 0, specials = 0;
  for (long i = 0; i < n; i++) {
    double d;
    switch (i % 4) {
      case 0: d = 6.0; break;
      case 1: d = ldexp(mant(), 10 + (int)(rnd() % 30)); break;            // layer spacing, cm
      case 2: { double st = ldexp(mant(), 10 + (int)(rnd() % 30)); d = 2.0 * st * st; break; }
      default: { uint64_t m = 0x3ff0000000000000ull | (0xfffffffffffffull - (rnd() % 64)); memcpy(&d, &m, 8); d = ldexp(d, (int)(rnd() % 60)); }  // mantissa near all ones
    }
    const double rd = 1.0 / d;
    double x = ldexp(mant(), -200 + (int)(rnd() % 260));
    if (rnd() & 1) x = -x;
    if (i % 16 == 5) { x = d * mant(); }                                    // near exact quotients
    if (i % 16 == 9) { double q = mant(); x = q * d; x = nextafter(x, (rnd() & 1) ? 1e300 : -1e300); specials++; }
    if (i % 1024 == 7) x = 0.0;
    const double a = trx::quotient_rn(x, d, rd), b = x / d;
    if (memcmp(&a, &b, 8) != 0 && !(a == 0 && b == 0)) { if (bad < 5) printf("x %a d %a: %a vs %a\n", x, d, a, b); bad++; }
  }
  printf("%ld operands, %ld differ\n", n, bad);
  // the three forms of the bottom-point parabola must be the same bits: parab3 (the reference's
  // operations, pu/src/numerical.c:182-195), parab3_recip (its two divisions as residual-corrected
  // products) and parab3_chain (the node-only factors handed in as well: the chain of k_ray_tail and
  // k_optical_depth_vertical)
  long pbad = 0, pn = n / 20;
  for (long i = 0; i < pn; i++) {
    const double r0 = ldexp(mant(), 29 + (int)(rnd() % 5)), step = ldexp(mant(), 18 + (int)(rnd() % 8));      // radii ~1e9..1e10 cm, layers ~3..500 km
    const double r1 = r0 + step;
    const double v0 = ldexp(mant(), -40 + (int)(rnd() % 50)), v1 = v0 * (0.5 + mant() / 2), v2 = v1 * (0.25 + mant() / 2);
    const double st = r1 - r0, t0 = r0 / st, tw = 2.0 * st * st;
    const double a = trx::parab3(r0, r1, v0, v1, v2, r0);
    const double b = trx::parab3_recip(st, 1.0 / st, t0, tw, 1.0 / tw, v0, v1, v2, r0);
    const double c = trx::parab3_chain(st, 1.0 / st, t0, t0 + 1.5, tw, 1.0 / tw, v0, v1, v2, r0, r0 * r0);
    if (memcmp(&a, &b, 8) != 0 || memcmp(&a, &c, 8) != 0) { if (pbad < 5) printf("parabola: %a %a %a\n", a, b, c); pbad++; }
  }
  printf("%ld parabolas, %ld differ\n", pn, pbad);
  return bad != 0 || pbad != 0;
}
