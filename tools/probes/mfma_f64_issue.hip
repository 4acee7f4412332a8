// fp64 MFMA issue on gfx950, one wave per SIMD: cycles per v_mfma_f64_16x16x4_f64 with 1 / 2 / 4 / 8 independent accumulator chains, and
// with independent VALU work (v_fma_f64) or LDS reads interleaved -- what overlaps with the matrix instruction and what does not.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o mfma_f64_issue mfma_f64_issue.hip
// (without -amdgpu-mfma-vgpr-form the compiler parks the accumulators of the multi-chain loops in the accumulator file and copies all of
// them in and out every trip: 16 v_accvgpr moves per MFMA in the 8-chain loop, which is then what the loop measures)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int CH, int VALU, int LDSR>
__global__ void __launch_bounds__(64) k(double* out, int iters, double a, double b) {
  __shared__ double lds[64 * 8];
  lds[threadIdx.x] = a; lds[threadIdx.x + 64] = b;
  __syncthreads();
  v4d c[8];
  for (int i = 0; i < 8; ++i) c[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  double acc = 0.0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      c[r % CH] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[r % CH], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < VALU; ++v) x[(r + v) & 7] = __builtin_fma(x[(r + v) & 7], a, b);
#pragma unroll
      for (int v = 0; v < LDSR; ++v) acc += lds[(threadIdx.x + 64 * ((r + v) & 7)) & 511];
    }
  }
  const long long t1 = clock64();
  double s = acc;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3] + x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s + (double)(t1 - t0) * 1e-300;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[64 * 4096] = (double)(t1 - t0);
}
template <int CH, int VALU, int LDSR>
int run(double* d, const char* name, int grid = 1024) {
  const int iters = 20000;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<CH, VALU, LDSR>), dim3(grid), dim3(64), 0, 0, d, 100, 1.0, 0.5);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<CH, VALU, LDSR>), dim3(grid), dim3(64), 0, 0, d, iters, 1.0, 0.5);
  CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double ticks; CK(hipMemcpy(&ticks, d + 64 * 4096, 8, hipMemcpyDeviceToHost));
  const double n = 8.0 * iters;
  printf("%-44s %7.3f ms  %6.1f ns / MFMA and wave  %6.1f clock64 ticks  (%.2f TFLOP/s)\n", name, ms, ms * 1e6 / n, ticks / n, 2048.0 * n * grid / (ms * 1e-3) / 1e12);
  return 0;
}
int main() {
  double* d; CK(hipMalloc(&d, (64 * 4096 + 8) * sizeof(double)));
  if (run<1, 0, 0>(d, "1 chain")) return 1;
  if (run<2, 0, 0>(d, "2 chains")) return 1;
  if (run<4, 0, 0>(d, "4 chains")) return 1;
  if (run<8, 0, 0>(d, "8 chains")) return 1;
  if (run<4, 1, 0>(d, "4 chains + 1 v_fma_f64 per MFMA")) return 1;
  if (run<4, 4, 0>(d, "4 chains + 4 v_fma_f64 per MFMA")) return 1;
  if (run<4, 8, 0>(d, "4 chains + 8 v_fma_f64 per MFMA")) return 1;
  if (run<1, 4, 0>(d, "1 chain  + 4 v_fma_f64 per MFMA")) return 1;
  if (run<4, 0, 2>(d, "4 chains + 2 ds_read_b64 per MFMA")) return 1;
  if (run<1, 0, 0>(d, "1 chain, 2 waves per SIMD", 2048)) return 1;
  if (run<4, 0, 0>(d, "4 chains, 2 waves per SIMD", 2048)) return 1;
  if (run<1, 0, 0>(d, "1 chain, 4 waves per SIMD", 4096)) return 1;
  if (run<4, 4, 0>(d, "4 chains + 4 v_fma_f64, 2 waves per SIMD", 2048)) return 1;
  return 0;
}
