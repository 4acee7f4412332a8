// Issue rate of v_mfma_f64_4x4x4_4b_f64 on gfx950, one wave per SIMD, 1 / 2 / 4 / 8 independent accumulators (the line search's feedback
// product, dyn_split_kernels.hip): is it a quarter of the 16 x 16 x 4 form's 64 cycles, as its quarter of the flops suggests?
//   hipcc --offload-arch=gfx950 -O3 -o mfma444_issue tools/probes/mfma_f64_4x4x4_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void __launch_bounds__(64) k(double* out, int iters, double a, double b) {
  double c[8];
  for (int i = 0; i < 8; ++i) c[i] = 0.0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) c[r % CH] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[r % CH], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0.0;
  for (int i = 0; i < 8; ++i) s += c[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[64 * 4096] = (double)(t1 - t0);
}
template <int CH> void run(double* d, const char* name) {
  const int iters = 20000, grid = 1024;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<CH>), dim3(grid), dim3(64), 0, 0, d, 100, 1.0, 0.5);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<CH>), dim3(grid), dim3(64), 0, 0, d, iters, 1.0, 0.5);
  hipEventRecord(b); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  double ticks; hipMemcpy(&ticks, d + 64 * 4096, 8, hipMemcpyDeviceToHost);
  const double n = 8.0 * iters;
  printf("%-12s %7.3f ms  %6.1f ns / MFMA and wave  %6.1f clock64 ticks  (%.2f TFLOP/s)\n", name, ms, ms * 1e6 / n, ticks / n, 512.0 * n * grid / (ms * 1e-3) / 1e12);
}
int main() {
  double* d; hipMalloc(&d, (64 * 4096 + 8) * sizeof(double));
  run<1>(d, "1 chain"); run<2>(d, "2 chains"); run<4>(d, "4 chains"); run<8>(d, "8 chains");
  return 0;
}
