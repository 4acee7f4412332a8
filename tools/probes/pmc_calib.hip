// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access widths the solver's kernels use
// (MI355X_MICROARCH.md, "HBM": widths other than 16 B/lane are uncalibrated).  Every kernel moves a known
// byte count; run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes).
//   hipcc --offload-arch=gfx950 -O3 -o pmc_calib pmc_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// 8 B per lane, coalesced read + write: n doubles in, n doubles out
__global__ void calib_copy8(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// 16 B per lane, coalesced read + write (the guide's calibrated case)
__global__ void calib_copy16(const double2* __restrict__ src, double2* __restrict__ dst, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// read only (8 B/lane), one tiny write per block
__global__ void calib_read8(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += src[i];
  if (s == 12345.678) dst[blockIdx.x] = s;
}
// the Riccati gain store: a 19 x 51 row-major block per (workgroup, knot), written as 16-double row segments
// (lane = 16*lk + lr -> row 16I + lk + 4r, column 16w + lr), 25 knots per workgroup
__global__ void __launch_bounds__(256) calib_gain_store(double* __restrict__ K, int knots) {
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  for (int t = knots - 1; t >= 0; --t) {
    double* Kg = K + ((size_t)blockIdx.x * knots + t) * 19 * 51;
    const int col = 16 * w + lr;
    for (int I = 0; I < 2; ++I)
      for (int r = 0; r < 4; ++r) {
        const int a = 16 * I + lk + 4 * r;
        if (a < 19 && col < 51) Kg[a * 51 + col] = (double)(a + col);
      }
    __syncthreads();
  }
}

int main() {
  const size_t n = (size_t)1 << 27;   // 1 GiB of doubles
  double *a, *b;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
  CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(calib_copy8, dim3(4096), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(calib_copy16, dim3(4096), dim3(256), 0, 0, (const double2*)a, (double2*)b, n / 2);
    hipLaunchKernelGGL(calib_read8, dim3(4096), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(calib_gain_store, dim3(4096), dim3(256), 0, 0, b, 25);
    CK(hipDeviceSynchronize());
  }
  printf("bytes: copy8 r/w %zu, copy16 r/w %zu, read8 r %zu, gain_store w %zu\n", n * 8, n * 8, n * 8, (size_t)4096 * 25 * 19 * 51 * 8);
  return 0;
}
