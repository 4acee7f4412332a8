// Probe: does the SIZE of a kernel's private segment cost when the segment is never touched?  One wave per SIMD (launch bounds ask for
// the whole register file), a fixed dependent fp64 chain; a private array of N doubles is only reached through a runtime flag that is
// never set.  Prints the kernel time for several N at 1024 and 2048 waves.   hipcc --offload-arch=gfx950 -O3 scratch_size_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int N>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) probe(double* out, int iters, int touch) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  double a[N > 0 ? N : 1];
  double s = 1.0 + 1e-9 * gid;
  if (touch) { for (int i = 0; i < N; ++i) a[i] = s * i; int j = gid % (N > 0 ? N : 1); for (int i = 0; i < iters; ++i) { s += a[j]; j = (j * 7 + 1) % (N > 0 ? N : 1); a[j] = s; } }
  for (int i = 0; i < iters; ++i) s = __builtin_fma(s, 1.0000001, 1e-12);
  out[gid] = s;
}
template <int N> void run(int blocks, int iters) {
  double* d; hipMalloc(&d, (size_t)blocks * 64 * sizeof(double));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<N>, dim3(blocks), dim3(64), 0, 0, d, iters, 0);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<N>, dim3(blocks), dim3(64), 0, 0, d, iters, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("private segment %5d B/lane  waves %5d  -> %.3f ms\n", N * 8, blocks, best);
  hipFree(d);
}
int main() {
  const int iters = 200000;
  for (int blocks : {1024, 2048}) {
    run<0>(blocks, iters); run<16>(blocks, iters); run<36>(blocks, iters); run<64>(blocks, iters); run<100>(blocks, iters);
    run<137>(blocks, iters); run<160>(blocks, iters); run<224>(blocks, iters); run<300>(blocks, iters); run<440>(blocks, iters);
  }
  return 0;
}
