// nrmath.hpp — the small elementary functions of the solver kernels, compilable for the HOST as well: tests/test_nrmath.py builds
// this header with g++ and measures each function against long-double libm on its argument range (the device code and the check compile
// the same source).
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define NRMATH_FN __device__ __forceinline__
#else
#define NRMATH_FN static inline
#endif

namespace mapdn {

// cos/sin of a Newton angle step |x| <= 0.5: Taylor to x^16 / x^17 (remainder < 2e-23)
NRMATH_FN void sincos_small(double x, double* s, double* c) {
  const double z = x * x;
  double ps = 1.0 / 355687428096000.0;                         // 1/17!
  ps = fma(ps, z, -1.0 / 1307674368000.0);                     // -1/15!
  ps = fma(ps, z, 1.0 / 6227020800.0);                         // 1/13!
  ps = fma(ps, z, -1.0 / 39916800.0);                          // -1/11!
  ps = fma(ps, z, 1.0 / 362880.0);                             // 1/9!
  ps = fma(ps, z, -1.0 / 5040.0);                              // -1/7!
  ps = fma(ps, z, 1.0 / 120.0);                                // 1/5!
  ps = fma(ps, z, -1.0 / 6.0);                                 // -1/3!
  *s = fma(ps * z, x, x);
  double pc = 1.0 / 20922789888000.0;                          // 1/16!
  pc = fma(pc, z, -1.0 / 87178291200.0);                       // -1/14!
  pc = fma(pc, z, 1.0 / 479001600.0);                          // 1/12!
  pc = fma(pc, z, -1.0 / 3628800.0);                           // -1/10!
  pc = fma(pc, z, 1.0 / 40320.0);                              // 1/8!
  pc = fma(pc, z, -1.0 / 720.0);                               // -1/6!
  pc = fma(pc, z, 1.0 / 24.0);                                 // 1/4!
  pc = fma(pc, z, -0.5);                                       // -1/2!
  *c = fma(pc, z, 1.0);
}

// cos/sin of a LARGE Newton angle step (|x| > 0.5: a diverging or badly conditioned iterate; rare).  libm's sincos is ~200
// instructions with its Payne-Hanek branch, and it was inlined into every (unrolled / peeled) backward row of the layouts that
// update inside the rows — the rare path made the hot loops ~40 KB larger than the instruction cache likes (case141 x 8192:
// solver launch 111 -> 100 us without it).  This form is ~35 instructions: Cody-Waite reduction by pi/2 in three FMAs (fdlibm's
// pio2_1 / pio2_2 / pio2_3: exact products for |x| up to ~1e5 rad), the same Taylor pair on [-pi/4, pi/4] (remainders 8e-20 /
// 2e-18), quadrant fix-up.  Within 1-2 ulp there; beyond ~1e5 rad it loses accuracy gracefully (finite values) — such an iterate is
// diverging and ends in the non-convergence branch whatever its digits.  Every k_nr_tree geometry uses THIS function for large steps, so
// the tree solver's results stay bit-identical across launch geometries (k_nr_sparse / k_nr_dense call libm's sincos: they agree with
// the tree solver to the last ulp or two, not to the bit — the tests compare them with the oracle, at 1e-9).
NRMATH_FN void sincos_mid(double x, double* s, double* c) {
  const double k = fmin(fmax(rint(x * 6.36619772367581382433e-01), -2.0e9), 2.0e9);   // 2 / pi; clamped: the (int) conversion below must not
                                                                                      // depend on how a device saturates (|x| > 3e9 rad: garbage in, finite out)
  double r = fma(-k, 1.57079632673412561417e+00, x);
  r = fma(-k, 6.07710050630396597660e-11, r);
  r = fma(-k, 2.02226624871116645580e-21, r);
  double sr, cr;
  sincos_small(r, &sr, &cr);
  const int n = (int)k;
  const double ss = (n & 1) ? cr : sr, cc = (n & 1) ? sr : cr;
  *s = (n & 2) ? -ss : ss;
  *c = ((n + 1) & 2) ? -cc : cc;
}

// the bowl barrier (voltage_barrier/bowl.py:6-12) INSIDE its band |v - 1| <= 0.05: -0.01 N(v; 1, 0.1) + 0.04.  The exponent
// x = -(v - 1)^2 / (2 scale^2) lies in [-0.125, 0]: exp as its Taylor polynomial to x^11 (remainder < 3e-20; libm's exp with its range
// reduction and the division by scale^2 were a third of the epilogue's bus loop, which runs at one wave per SIMD), the division as a
// multiplication by the rounded 1 / (2 scale^2) — within 2 ulp of the reference's value
NRMATH_FN double bowl_inside(double v) {
  const double x = (v - 1.0) * (v - 1.0) * -50.0;
  double ex = 1.0 / 39916800.0;
  ex = fma(ex, x, 1.0 / 3628800.0); ex = fma(ex, x, 1.0 / 362880.0); ex = fma(ex, x, 1.0 / 40320.0); ex = fma(ex, x, 1.0 / 5040.0);
  ex = fma(ex, x, 1.0 / 720.0); ex = fma(ex, x, 1.0 / 120.0); ex = fma(ex, x, 1.0 / 24.0); ex = fma(ex, x, 1.0 / 6.0);
  ex = fma(ex, x, 0.5); ex = fma(ex, x, 1.0); ex = fma(ex, x, 1.0);
  return fma(-0.01 * 3.9894228040143270, ex, 0.04);       // 1 / sqrt(2 pi scale^2) = 3.98942280401432...
}

}  // namespace mapdn
