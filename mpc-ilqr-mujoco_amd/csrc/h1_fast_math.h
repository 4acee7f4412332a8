// sin and cos of a double in ~35 VALU instructions (the ocml sincos is ~4x that: the dynamics step evaluates
// 57 of them per knot, half of its instruction count).  Cody-Waite reduction by pi/2 with FMA (exact for |x| < 1e5,
// the regime of joint angles and half rotation increments -- branch-free on purpose: no large-argument path), fdlibm's
// __kernel_sin / __kernel_cos minimax polynomials on [-pi/4, pi/4]: error <= 1-2 ulp.
#pragma once
#include <hip/hip_runtime.h>

#ifndef DEVFN
#define DEVFN __device__ __forceinline__
#endif

namespace h1f {

DEVFN void sincos_fast(double x, double* sn, double* cs) {
  const double k = rint(x * 6.36619772367581382433e-01);    // 2 / pi
  double r = fma(-k, 1.57079632673412561417e+00, x);        // pio2_1 (33 bits): exact product
  r = fma(-k, 6.07710050650619224932e-11, r);               // pio2_1t
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                      -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double s = fma(z * r, ps, r);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                      2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int n = (int)k & 3;
  const double s1 = (n & 1) ? c : s, c1 = (n & 1) ? s : c;
  *sn = (n & 2) ? -s1 : s1;
  *cs = ((n + 1) & 2) ? -c1 : c1;
}

}  // namespace h1f
