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
// lane-exchange intrinsics: single-lane host runs never take the split-lane paths; these only have to compile
#include <cstring>
inline int __double2loint(double d) { long long b; std::memcpy(&b, &d, 8); return (int)(b & 0xffffffff); }
inline int __double2hiint(double d) { long long b; std::memcpy(&b, &d, 8); return (int)(b >> 32); }
inline double __hiloint2double(int hi, int lo) { long long b = ((long long)hi << 32) | (unsigned)lo; double d; std::memcpy(&d, &b, 8); return d; }
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rmask, bmask, bc) (src)
