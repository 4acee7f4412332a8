// CPU check (sanitizers on) of the register-resident ABA (h1_aba_reg.h) against the templated scalar ABA.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "../../mpc-ilqr-mujoco_amd/csrc/h1_dynamics_dev.h"
#include "../../mpc-ilqr-mujoco_amd/csrc/h1_aba_reg.h"
int main() {
  srand(5);
  double worst = 0;
  for (int trial = 0; trial < 6; ++trial) {
    double x[51] = {0}, u[19], xa[51], xb[51];
    x[2] = 1.0432; double q[4] = {1, 0.3 * trial * (rand() / (double)RAND_MAX - 0.5), 0.2 * trial, -0.1 * trial};
    double n = std::sqrt(q[0]*q[0]+q[1]*q[1]+q[2]*q[2]+q[3]*q[3]); for (int k = 0; k < 4; ++k) x[3+k] = q[k] / n * (trial == 5 ? 1.2 : 1.0);
    for (int i = 7; i < 51; ++i) x[i] = (trial == 0 && i >= 26) ? 0.0 : 1.5 * (rand() / (double)RAND_MAX - 0.5);
    for (int i = 0; i < 19; ++i) u[i] = 60.0 * (rand() / (double)RAND_MAX - 0.5);
    h1::DynParams P{0.02, {0.1 * trial, 0, -9.81 + trial}};
    h1::step<double>(x, u, P, xa);
    double lds[h1r::LDS_SLOTS];
    h1r::LaneLds L{lds, 1, 0};
    h1r::step(x, u, P.h, P.g, L, xb);
    double e = 0; for (int i = 0; i < 51; ++i) e = std::fmax(e, std::fabs(xa[i] - xb[i]));
    double ca[3], cb[3]; h1::com_mj(x, ca); h1r::com_mj(x, cb);
    double ec = 0; for (int k = 0; k < 3; ++k) ec = std::fmax(ec, std::fabs(ca[k] - cb[k]));
    printf("trial %d: max|step diff| = %.3e  max|com diff| = %.3e\n", trial, e, ec);
    worst = std::fmax(worst, std::fmax(e, ec));
  }
  return worst < 1e-11 ? 0 : 1;
}
