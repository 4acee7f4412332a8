// host shim so the device headers compile with g++ for sanitizer runs (CPU build only)
#pragma once
#include <cmath>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __noinline__
#define __constant__
#define __shared__ static
#define __launch_bounds__(...)
inline void sincos(double a, double* s, double* c) { *s = std::sin(a); *c = std::cos(a); }
using std::sqrt; using std::fma; using std::fabs; using std::fmax; using std::fmin;
using std::rint;
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
inline void __syncthreads() {}
