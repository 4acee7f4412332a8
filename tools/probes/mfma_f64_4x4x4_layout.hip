// Lane layout of v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 products per instruction), found by brute force: one wave,
// A = 1 + lane, B = 101 + 7 lane, D compared on the host against every assignment (i, k) / (k, n) / (i, n) <- lane & 15.
//   hipcc --offload-arch=gfx950 tools/probes/mfma_f64_4x4x4_layout.hip -o /tmp/mfma444 && /tmp/mfma444
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out) {
  const int l = threadIdx.x;
  const double a = 1.0 + l, b = 101.0 + 7.0 * l;
  double d = 0.0;
  d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, 0, 0, 0);
  out[l] = d;
}
int main() {
  double* d; hipMalloc(&d, 64 * sizeof(double));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  auto A = [](int l) { return 1.0 + l; };
  auto B = [](int l) { return 101.0 + 7.0 * l; };
  // hypotheses: for operand X, lane j = l & 15 of block l >> 4 holds element (p, q) with p = j % 4, q = j / 4 (form 0) or p = j / 4, q = j % 4 (form 1)
  for (int fa = 0; fa < 2; ++fa) for (int fb = 0; fb < 2; ++fb) for (int fd = 0; fd < 2; ++fd) {
    int ok = 1;
    for (int l = 0; l < 64 && ok; ++l) {
      const int blk = l >> 4, j = l & 15;
      const int i = fd ? j / 4 : j % 4, n = fd ? j % 4 : j / 4;
      double s = 0.0;
      for (int kk = 0; kk < 4; ++kk) {
        const int ja = fa ? (i * 4 + kk) : (kk * 4 + i);      // lane of A[i][kk]
        const int jb = fb ? (kk * 4 + n) : (n * 4 + kk);      // lane of B[kk][n]
        s += A(16 * blk + ja) * B(16 * blk + jb);
      }
      if (s != h[l]) ok = 0;
    }
    if (ok) printf("layout: A[i][k] at lane 16 b + %s, B[k][n] at lane 16 b + %s, D[i][n] at lane 16 b + %s\n", fa ? "4 i + k" : "i + 4 k", fb ? "4 k + n" : "k + 4 n", fd ? "4 i + n" : "i + 4 n");
  }
  for (int l = 0; l < 64; ++l) printf("%.0f%c", h[l], (l & 15) == 15 ? '\n' : ' ');
  return 0;
}
