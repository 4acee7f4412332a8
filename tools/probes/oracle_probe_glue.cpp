// TEST INFRASTRUCTURE: exposes the oracle's AD linearisation of one knot to the CPU probes.
#include "../../oracle/h1_costs.hpp"
#include <vector>
void oracle_linearize(const double* x, const double* u, double h, const double* g, double* A, double* B) {
  using namespace orc;
  typedef D1<H1_NX + H1_NU> T;
  DynParams P; P.h = h; P.g[0] = g[0]; P.g[1] = g[1]; P.g[2] = g[2];
  std::vector<T> xs(H1_NX), us(H1_NU), xn(H1_NX);
  for (int i = 0; i < H1_NX; ++i) xs[i] = T::var(x[i], i);
  for (int i = 0; i < H1_NU; ++i) us[i] = T::var(u[i], H1_NX + i);
  h1_step<T>(xs.data(), us.data(), P, xn.data());
  for (int i = 0; i < H1_NX; ++i) { for (int j = 0; j < H1_NX; ++j) A[i * H1_NX + j] = xn[i].g[j]; for (int j = 0; j < H1_NU; ++j) B[i * H1_NU + j] = xn[i].g[H1_NX + j]; }
}
