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
  return bad != 0;
}
