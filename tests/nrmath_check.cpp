// host-side accuracy check of mapdn_amd/csrc/nrmath.hpp (compiled by tests/test_nrmath.py with g++ -mfma): one line per function,
// reference = long-double libm.
#include <cstdio>
#include <cmath>
#include <cstdint>
#include "nrmath.hpp"

static uint64_t sm = 88172645463325252ull;
static double u() { sm ^= sm << 13; sm ^= sm >> 7; sm ^= sm << 17; return (double)(sm >> 11) / 9007199254740992.0; }
static double relerr(double got, long double want) { return (double)(fabsl((long double)got - want) / fabsl(want)); }

int main() {
  double e_small = 0, e_mid = 0, e_mid_far = 0, e_bowl = 0, e_id = 0;
  for (int i = 0; i < 2000000; ++i) {
    double s, c;
    const double x = (2 * u() - 1) * 0.5;                               // the Newton steps of the polynomial path
    mapdn::sincos_small(x, &s, &c);
    if (x != 0.0) e_small = fmax(e_small, relerr(s, sinl((long double)x)));
    e_small = fmax(e_small, relerr(c, cosl((long double)x)));
    double s2, c2;
    mapdn::sincos_mid(x, &s2, &c2);                                     // on |x| <= 0.5 the large-step form IS the polynomial (k = 0)
    if (s2 != s || c2 != c) e_id = 1.0;
    const double y = (2 * u() - 1) * 1e5;                               // large steps: absolute error (sin / cos pass through zero)
    mapdn::sincos_mid(y, &s, &c);
    e_mid = fmax(e_mid, fmax(fabs((double)(s - sinl((long double)y))), fabs((double)(c - cosl((long double)y)))));
    const double z = (2 * u() - 1) * 1e8;
    mapdn::sincos_mid(z, &s, &c);
    e_mid_far = fmax(e_mid_far, fmax(fabs((double)(s - sinl((long double)z))), fabs((double)(c - cosl((long double)z)))));
    const double v = 0.95 + 0.1 * u();                                  // the band of the bowl barrier
    const long double ref = -0.01L * (1.0L / sqrtl(2.0L * 3.14159265358979323846264338327950288L * 0.1L * 0.1L)) * expl(-0.5L * ((long double)v - 1.0L) * ((long double)v - 1.0L) / (0.1L * 0.1L)) + 0.04L;
    e_bowl = fmax(e_bowl, fabs((double)(mapdn::bowl_inside(v) - ref)));
  }
  printf("sincos_small_rel %.3e\nsincos_mid_abs_1e5 %.3e\nsincos_mid_abs_1e8 %.3e\nmid_equals_small_inside %d\nbowl_abs %.3e\n", e_small, e_mid, e_mid_far, e_id == 0.0, e_bowl);
  return 0;
}
