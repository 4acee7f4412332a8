// Probe: does a kernel with a large per-lane private (scratch) footprint run on this box?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int N>
__global__ void __launch_bounds__(64) probe(double* out, int n, int stride) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  double a[N];
  for (int i = 0; i < N; ++i) a[i] = (double)(i + gid);
  double s = 0.0;
  int j = gid % N;
  for (int i = 0; i < n; ++i) { s += a[j]; j = (j * stride + 1) % N; a[j] += s * 1e-9; }
  out[gid] = s;
}
template <int N> int run(int blocks) {
  double* d; hipMalloc(&d, (size_t)blocks * 64 * sizeof(double));
  hipLaunchKernelGGL(probe<N>, dim3(blocks), dim3(64), 0, 0, d, 1000, 7);
  hipError_t e = hipDeviceSynchronize();
  double h = 0; hipMemcpy(&h, d, sizeof(double), hipMemcpyDeviceToHost);
  printf("N=%d (%d B/lane) blocks=%d -> %s out0=%g\n", N, N * 8, blocks, hipGetErrorString(e), h);
  hipFree(d);
  return e != hipSuccess;
}
int main(int argc, char** argv) {
  int which = atoi(argv[1]), blocks = atoi(argv[2]);
  if (which == 0) return run<768>(blocks);
  if (which == 1) return run<1280>(blocks);
  if (which == 2) return run<1600>(blocks);
  if (which == 3) return run<2048>(blocks);
  return run<4096>(blocks);
}
