// CPU check of the analytic linearisation device code (h1_linearize_dev.h) against the oracle's AD Jacobians.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../../mpc-ilqr-mujoco_amd/csrc/h1_linearize_dev.h"
#include "../../oracle/ilqr_oracle_for_probe.h"
using namespace h1;

static void qacc_for(const double* x, const double* u, const DynParams& P, const double* tau_base, const double* dtau, double* qacc, KnotDump* dump) {
  const double h = P.h;
  double qn = std::sqrt(x[3]*x[3]+x[4]*x[4]+x[5]*x[5]+x[6]*x[6]);
  double qh[4] = {x[3]/qn, x[4]/qn, x[5]/qn, x[6]/qn};
  double tau[H1_NU];
  for (int i = 0; i < H1_NU; ++i) { double ui = u[i]; if (ui < H1_CTRLRANGE[i][0]) ui = H1_CTRLRANGE[i][0]; if (ui > H1_CTRLRANGE[i][1]) ui = H1_CTRLRANGE[i][1]; tau[i] = ui - H1_DAMPING * x[H1_NQ+6+i] + (dtau ? dtau[i] : 0.0); }
  if (dump) forward_dynamics<double, true>(qh, x+7, x+H1_NQ, tau, H1_ARMATURE + h*H1_DAMPING, P.g, qacc, tau_base, dump);
  else forward_dynamics<double, false>(qh, x+7, x+H1_NQ, tau, H1_ARMATURE + h*H1_DAMPING, P.g, qacc, tau_base, nullptr);
}

int main() {
  DynParams P{0.02, {0, 0, -1.0}};
  srand(3);
  double worst = 0;
  for (int trial = 0; trial < 4; ++trial) {
    double x[H1_NX] = {0}, u[H1_NU];
    x[2] = 1.0432; double q[4] = {1, 0.2*(rand()/(double)RAND_MAX-0.5)*trial, 0.3*(rand()/(double)RAND_MAX-0.5)*trial, 0.1*trial};
    double n = std::sqrt(q[0]*q[0]+q[1]*q[1]+q[2]*q[2]+q[3]*q[3]); for (int k=0;k<4;++k) x[3+k] = q[k]/n;
    for (int i = 7; i < H1_NX; ++i) x[i] = (trial==0 && i>=26) ? 0.0 : 0.8 * (rand() / (double)RAND_MAX - 0.5);
    for (int i = 0; i < H1_NU; ++i) u[i] = 20.0 * (rand() / (double)RAND_MAX - 0.5);
    if (trial == 2) u[4] = 100.0;  // clamped actuator
    static LinShared L;
    L.h = P.h;
    for (int i = 0; i < H1_NX; ++i) L.x[i] = x[i];
    for (int i = 0; i < H1_NU; ++i) L.u_[i] = u[i];
    double y0[H1_NV];
    static KnotDump KD;
    qacc_for(x, u, P, nullptr, nullptr, y0, &KD);
    // what lin_load_dump + lin_accumulate_forces leave in LDS on the device
    for (int k = 0; k < 9; ++k) L.D.R0[k] = KD.R0[k];
    for (int k = 0; k < 3; ++k) L.D.aL[k] = KD.aL[k];
    for (int k = 0; k < H1_NV; ++k) L.D.qacc[k] = KD.qacc[k];
    for (int k = 0; k < 36; ++k) L.u.m.IA0inv[k] = KD.IA0inv[k];
    for (int i = 0; i < H1_NB; ++i) {
      for (int k = 0; k < 9; ++k) L.D.Rj[i][k] = KD.Rj[i][k];
      for (int k = 0; k < 6; ++k) { L.D.v[i][k] = KD.v[i][k]; L.D.F[i][k] = KD.F[i][k]; L.u.m.U[i][k] = KD.U[i][k]; }
      L.u.m.Dinv[i] = KD.Dinv[i];
      inertia_mul(i, KD.v[i], L.Iv[i]);
      if (i > 0) xf_motion(KD.Rj[i], H1_POS[i], KD.a[H1_PARENT[i]], L.xa[i]);
    }
    static double Mref[H1_NV][H1_NV];
    for (int c = 0; c < H1_NV; ++c) {
      double tb[6] = {0,0,0,0,0,0}, dt[H1_NU] = {0}, y[H1_NV];
      if (c < 6) tb[c] = 1.0; else dt[c-6] = 1.0;
      qacc_for(x, u, P, tb, dt, y, nullptr);
      for (int r = 0; r < H1_NV; ++r) Mref[r][c] = y[r] - y0[r];
    }
    for (int lane = 0; lane < 64; ++lane) lin_minv_lane(L, lane);
    { double em = 0; for (int r = 0; r < H1_NV; ++r) for (int c = 0; c < H1_NV; ++c) em = std::fmax(em, std::fabs(Mref[r][c] - L.Minv[MINV_IDX(r, c)])); printf("  Minv sweep vs unit-force differences: %.3e\n", em); }
    lin_prologue(L);
    for (int lane = 0; lane < 64; ++lane) lin_tangent_zero(L, lane);
    for (int lane = 0; lane < 64; ++lane) lin_tangent_chains(L, lane);
    for (int lane = 0; lane < 64; ++lane) lin_tangent_pelvis(L, lane);
    for (int lane = 0; lane < 64; ++lane) lin_apply_minv_lane(L, lane);
    std::vector<double> A(51*51), B(51*19), Ao(51*51), Bo(51*19);
    for (int k = 0; k < 51; ++k) lin_column(L, 0, k, [&](int r, double v) { A[r*51+k] = v; });
    for (int k = 0; k < 19; ++k) lin_column(L, 1, k, [&](int r, double v) { B[r*19+k] = v; });
    oracle_linearize(x, u, P.h, P.g, Ao.data(), Bo.data());
    double ea = 0, eb = 0; int wa = 0;
    for (int i = 0; i < 51*51; ++i) { double d = std::fabs(A[i]-Ao[i]); if (d > ea) { ea = d; wa = i; } }
    for (int i = 0; i < 51*19; ++i) eb = std::fmax(eb, std::fabs(B[i]-Bo[i]));
    printf("trial %d: max|A-Ao| = %.3e at (%d,%d) [%g vs %g]  max|B-Bo| = %.3e\n", trial, ea, wa/51, wa%51, A[wa], Ao[wa], eb);
    worst = std::fmax(worst, std::fmax(ea, eb));
  }
  return worst < 1e-9 ? 0 : 1;
}
