// fp64 VALU issue rate of one wave per SIMD vs several: independent v_fma_f64 chains, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -o fp64_issue fp64_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(64) k_fma(double* out, int iters, double a, double b) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
    x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
    x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main() {
  double* d; CK(hipMalloc(&d, 64 * 16384 * sizeof(double)));
  const int iters = 200000;
  for (int waves_per_simd = 1; waves_per_simd <= 4; ++waves_per_simd) {
    const int grid = 1024 * waves_per_simd;   // 256 CUs x 4 SIMDs
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_fma, dim3(grid), dim3(64), 0, 0, d, 1000, 0.999, 0.001);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_fma, dim3(grid), dim3(64), 0, 0, d, iters, 0.999, 0.001);
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double fl = 2.0 * 8.0 * iters * 64.0 * grid;
    printf("waves/SIMD %d: %.3f ms, %.2f TFLOP/s fp64, %.2f cycles per wave-FMA at 2.4 GHz\n", waves_per_simd, ms, fl / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / (8.0 * iters) / waves_per_simd);
  }
  return 0;
}
