// Analytic Jacobians of the STANCE-CONSTRAINED step (contact row f4): A_t = df/dx, B_t = df/du of
//   Mhat qacc + bias(q, v) - J^T lambda = tau,      J qacc + gamma(q, v) + v_f / h + soft lambda = 0
// (h1_dynamics_dev.h forward_dynamics_stance; reference plant: RobotUtils::rolloutOneStep with MuJoCo's floor contacts,
// robot_utils.cpp:106-117, which the reference differentiates by forward differences, robot_utils.cpp:120-160).
// Differentiating the two equations with qacc, lambda as the dependent variables (active set fixed):
//   Mhat dqacc - J^T dlambda = -dT,   J dqacc + soft dlambda = -dR
//   dT = tangent of the inverse dynamics WITH the contact wrench as an external force fixed in link coordinates
//        (the tangent-RNEA sweeps of h1_linearize_dev.h, with -lambda added to the foot bodies' forces),
//   dR = tangent of (true foot acceleration + v_f / h) with qacc fixed (foot body's dv, da of the same sweeps, minus the
//        tangent of the rotated gravity offset),
//   => dlambda = (C + soft)^-1 (G^T dT - dR),  dqacc = -Mhat^-1 dT + G dlambda,   G = Mhat^-1 J^T,  C = J G.
// G and C come from twelve unit-wrench responses of the articulated-body recursion (lanes 25..36 beside the 25 Minv lanes);
// controls: dqacc/du_i = (Minv[:, 6+i] - G (C + soft)^-1 G[6+i, :]^T) free_u.
// The primal dump is the FREE solve (k_lin_primal_r); lambda, the constrained qacc and body accelerations are rebuilt here.
#pragma once
#include "h1_linearize_dev.h"

namespace h1 {

struct LinContact {
  double a[H1_NB][6];        // body accelerations (gravity-offset form): free solve, then constrained
  double G[H1_NV][12];       // Mhat^-1 J^T, columns = wrench components (left foot 0..5, right foot 6..11)
  double C[12][12];          // J Mhat^-1 J^T
  double F[12][12];          // Cholesky factor of the masked C + soft I: lower, reciprocal pivots on the diagonal
  double b[12], lam[12];
  double dq[H1_NV];          // G lambda
  double da[H1_NB][6];       // body acceleration increments
  double offb[11][3];        // gravity offset R_i^T R0^T (-g) at the leg bodies 1..10 ([0]: pelvis)
  double zl[2][3];           // world up axis in the feet's link coordinates
  double dR[2][6][19];       // tangent of (a_f,true + v_f / h) of foot g per chain-group slot
  double W[12][LIN_LD];      // dlambda per direction
  double WU[12][20];         // dlambda per control column
  int act[2];                // active stance feet after the unilateral check
};
// Sliding feet (contact modes 3 / 4, h1_aba_split.h stance_correct): what the multiplier tangents need beside LinContact.
// Two more arrays live in LinContact storage that is dead by then: dZ[2][3][16] (tangent of the up axis in the feet's link coordinates
// per chain-group slot) over Cc.da, and rho = (C lambda - b) on the translation rows of a sliding foot over Cc.b; in mode 4 the factor of
// the STICKING system keeps the strict upper triangle of Cc.F (transposed) and its reciprocal pivots are Fsd.
struct LinSlide {
  double zlb[10][3];         // world up axis in the coordinates of the leg bodies 1..10
  double n[2];               // normal multiplier of a sliding foot
  double t[2][3];            // mode 4: unit direction of the sticking solution's tangential force
  double fs[2][3];           // mode 4: force part of the sticking solution
  double fn[2], nt[2];       // mode 4: its normal component, the norm of its tangential part
  double Z[2][12];           // mode 4: K^-1 c_f (Woodbury)
  double Mi[4];              // mode 4: (I + U^T Z)^-1, row-major 2 x 2
  double Fsd[12];            // mode 4: reciprocal pivots of the sticking system's factor
  double ls[12];             // mode 4: the sticking solution (first pass of the tangent sweeps, see k_lin_tangent2c)
  int sl[2];                 // foot slides
};
DEVFN double (*slide_dZ(LinContact& Cc))[3][16] { return reinterpret_cast<double (*)[3][16]>(&Cc.da[0][0]); }
DEVFN const double (*slide_dZ(const LinContact& Cc))[3][16] { return reinterpret_cast<const double (*)[3][16]>(&Cc.da[0][0]); }
static_assert(sizeof(double) * 2 * 3 * 16 <= sizeof(double) * H1_NB * 6, "dZ must fit in LinContact::da");

// columns of Minv (lanes 0..24, as lin_minv_lane) and of G = Mhat^-1 J^T (lanes 25..36: unit wrench component c on foot g);
// the feet's accelerations of the wrench columns are the rows of C
template <int FIRST, int LEN> struct MinvChainOutC {
  template <int K> static DEVFN void step(LinShared& L, LinContact& Cc, const MinvPath& P, const double* ap, int lane) {
    constexpr int I = FIRST + K, ax = h1c::C_AXIS[I], dep = h1c::C_DEPTH[I];
    const double r[3] = {h1c::C_POS[I][0], h1c::C_POS[I][1], h1c::C_POS[I][2]};
    double a[6]; xf_motion(L.D.Rj[I], r, ap, a);
    double s = (P.body[dep] == I) ? P.du[dep] : 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) s -= L.u.m.U[I][q] * a[q];
    const double qdd = s * L.u.m.Dinv[I];
    a[ax] += qdd;
    if (lane < H1_NV) { if (5 + I >= lane) L.Minv[MINV_IDX(5 + I, lane)] = qdd; }
    else {
      Cc.G[5 + I][lane - H1_NV] = qdd;
      if constexpr (I == 5 || I == 10) {
#pragma unroll
        for (int q = 0; q < 6; ++q) Cc.C[(I == 5 ? 0 : 6) + q][lane - H1_NV] = a[q];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (K + 1 < LEN) step<K + 1>(L, Cc, P, a, lane);
  }
};
DEVFN void lin_minv_lane_c(LinShared& L, LinContact& Cc, int lane) {
  if (lane >= H1_NV + 12) return;
  const int c = lane;
  const bool wrench = c >= H1_NV;
  const int wc = c - H1_NV;                 // wrench column: foot wc / 6, component wc % 6
  double p0[6] = {0, 0, 0, 0, 0, 0};
  MinvPath P;
#pragma unroll
  for (int d = 0; d < 6; ++d) { P.body[d] = -1; P.du[d] = 0.0; }
  {
    int i = wrench ? (wc < 6 ? 5 : 10) : ((c >= 6) ? c - 5 : 0);        // first body on the path (0: none)
    double acc[6] = {0, 0, 0, 0, 0, 0};
    if (wrench) {
      const int k = wc % 6;
#pragma unroll
      for (int q = 0; q < 6; ++q) acc[q] = (q == k) ? -1.0 : 0.0;      // bias-force increment of a unit wrench on the foot
    }
    bool first = !wrench;
#pragma unroll
    for (int d = 5; d >= 1; --d) {
      if (i > 0 && H1_DEPTH[i] == d) {
        const int ax = H1_AXIS[i];
        const double du = (first ? 1.0 : 0.0) - (ax == 0 ? acc[0] : (ax == 1 ? acc[1] : acc[2]));
        first = false;
        P.body[d] = i; P.du[d] = du;
        const double s = du * L.u.m.Dinv[i];
        double pa[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) { pa[r] = acc[r] + L.u.m.U[i][r] * s; acc[r] = 0.0; }
        xf_force_acc(L.D.Rj[i], H1_POS[i], pa, acc);
        i = H1_PARENT[i];
      }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) p0[r] = acc[r];
  }
  double rhs[6] = {-p0[0], -p0[1], -p0[2], -p0[3], -p0[4], -p0[5]};
  if (!wrench) {
    if (c < 3) { rhs[3] += L.D.R0[3 * c]; rhs[4] += L.D.R0[3 * c + 1]; rhs[5] += L.D.R0[3 * c + 2]; }
    else if (c < 6) rhs[c - 3] += 1.0;
  }
  double a0[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) { double s = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s += L.u.m.IA0inv[6 * r + k] * rhs[k];
    a0[r] = s; }
  double lw[3]; mv3(L.D.R0, a0 + 3, lw);
  if (!wrench) {
#pragma unroll
    for (int r = 0; r < 3; ++r) if (r >= lane) L.Minv[MINV_IDX(r, lane)] = lw[r];
#pragma unroll
    for (int r = 0; r < 3; ++r) if (3 + r >= lane) L.Minv[MINV_IDX(3 + r, lane)] = a0[r];
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r) { Cc.G[r][wc] = lw[r]; Cc.G[3 + r][wc] = a0[r]; }
  }
  double a11[6];
  {
    constexpr int I = 11, ax = h1c::C_AXIS[11];
    const double r[3] = {h1c::C_POS[I][0], h1c::C_POS[I][1], h1c::C_POS[I][2]};
    xf_motion(L.D.Rj[I], r, a0, a11);
    double s = (P.body[1] == I) ? P.du[1] : 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) s -= L.u.m.U[I][q] * a11[q];
    const double qdd = s * L.u.m.Dinv[I];
    a11[ax] += qdd;
    if (!wrench) { if (5 + I >= lane) L.Minv[MINV_IDX(5 + I, lane)] = qdd; }
    else Cc.G[5 + I][wc] = qdd;
  }
  MinvChainOutC<12, 4>::step<0>(L, Cc, P, a11, lane);
  MinvChainOutC<16, 4>::step<0>(L, Cc, P, a11, lane);
  MinvChainOutC<1, 5>::step<0>(L, Cc, P, a0, lane);
  MinvChainOutC<6, 5>::step<0>(L, Cc, P, a0, lane);
}

// load by 128 threads: as lin_load_dump2, but the body accelerations are kept (they are corrected for the constraint before
// X a_parent and the body forces are formed)
// (every load requested before the first use, selection flags included, as in lin_load_dump2; false = rollout not selected)
DEVFN bool lin_load_dump2c(LinShared& L, LinContact& Cc, const double* g, int tid, const double* xg, const double* ug, const int* flag, const int* flag2) {
  LinDump& D = L.D;
  const int wv = tid >> 6, lane = tid & 63;
  const int f1 = flag ? *flag : 1, f2 = flag2 ? *flag2 : 1;
  const double r0 = g[LinDumpG_R0 + (tid < 9 ? tid : 0)];
  const double al = g[LinDumpG_aL + (tid < 3 ? tid : 0)];
  const double qa = g[LinDumpG_qacc + (tid < H1_NV ? tid : 0)];
  const int ev = tid < H1_NB * 6 ? tid : 0;
  const double vv = g[ldg_v_lin(ev)], uu = g[ldg_U_lin(ev)], aa = g[ldg_a_lin(ev)];
  const double di = g[ldg_Dinv(tid < H1_NB ? tid : 0)];
  const double ia = g[LinDumpG_IA0inv + (tid < 36 ? tid : 0)];
  const int i0 = (lane >= 1 && lane < H1_NB) ? lane : 1, i1 = lane < H1_NB ? lane : 0;
  const double s = g[ldg_s(i0)], c = g[ldg_c(i0)];
  double v6[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) v6[k] = g[ldg_v(i1, k)];
  const double xu = (tid < 64) ? xg[tid < H1_NX ? tid : 0] : ug[(tid - 64) < H1_NU ? tid - 64 : 0];
  if (!(f1 && f2)) return false;
  if (tid < H1_NX) L.x[tid] = xu;
  if (tid >= 64 && tid < 64 + H1_NU) L.u_[tid - 64] = xu;
  if (tid < 9) D.R0[tid] = r0;
  if (tid < 3) D.aL[tid] = al;
  if (tid < H1_NV) D.qacc[tid] = qa;
  if (tid < H1_NB * 6) { (&D.v[0][0])[tid] = vv; (&L.u.m.U[0][0])[tid] = uu; (&Cc.a[0][0])[tid] = aa; }
  if (tid < H1_NB) L.u.m.Dinv[tid] = di;
  if (tid < 36) L.u.m.IA0inv[tid] = ia;
  if (wv == 0 && lane >= 1 && lane < H1_NB) {
    const int i = lane, a = H1_AXIS[i], b = (a + 1) % 3, d = (a + 2) % 3;
    for (int r = 0; r < 3; ++r) {
      const double fa = H1_RFIX[i][r][a], fb = H1_RFIX[i][r][b], fd = H1_RFIX[i][r][d];
      D.Rj[i][3 * r + a] = fa; D.Rj[i][3 * r + b] = fb * c + fd * s; D.Rj[i][3 * r + d] = fd * c - fb * s;
    }
  }
  if (wv == 1 && lane < H1_NB) {
    const int i = lane;
    double Iv[6];
    inertia_mul(i, v6, Iv);
    for (int k = 0; k < 6; ++k) L.Iv[i][k] = Iv[k];
  }
  return true;
}

// one lane per foot: gravity offset and world up axis rotated down the leg, right-hand side of the constraint
DEVFN void lin_contact_rhs(LinShared& L, LinContact& Cc, const double* grav, int g, LinSlide* Zs = nullptr) {
  const LinDump& D = L.D;
  const double mg[3] = {-grav[0], -grav[1], -grav[2]};
  double off[3]; mtv3(D.R0, mg, off);
  double zl[3] = {D.R0[6], D.R0[7], D.R0[8]};
  if (g == 0) { Cc.offb[0][0] = off[0]; Cc.offb[0][1] = off[1]; Cc.offb[0][2] = off[2]; }
  const int first = g == 0 ? 1 : 6;
  for (int i = first; i < first + 5; ++i) {
    double o2[3], z2[3]; mtv3(D.Rj[i], off, o2); mtv3(D.Rj[i], zl, z2);
    for (int k = 0; k < 3; ++k) { off[k] = o2[k]; zl[k] = z2[k]; Cc.offb[i][k] = o2[k]; }
    if (Zs) for (int k = 0; k < 3; ++k) Zs->zlb[i - 1][k] = z2[k];
  }
  const int fbody = first + 4;
  for (int k = 0; k < 3; ++k) {
    Cc.zl[g][k] = zl[k];
    Cc.b[6 * g + k] = -D.v[fbody][k] / L.h - Cc.a[fbody][k];
    Cc.b[6 * g + 3 + k] = -D.v[fbody][3 + k] / L.h - (Cc.a[fbody][3 + k] - off[k]);
  }
}
// x <- (C + soft)^-1 x with the factor left by lin_contact_solve_w (wave-uniform LDS reads; inactive rows are identity)
DEVFN void lin_contact_backsolve(const LinContact& Cc, double* x) {
  // (a scheduling fence per row: left alone the compiler hoists all 78 factor entries into registers first and spills)
#pragma unroll
  for (int i = 0; i < 12; ++i) { double t = x[i];
#pragma unroll
    for (int k = 0; k < 12; ++k) if (k < i) t -= Cc.F[i][k] * x[k];
    x[i] = t * Cc.F[i][i];
    __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
  for (int i = 11; i >= 0; --i) { double t = x[i];
#pragma unroll
    for (int k = 0; k < 12; ++k) if (k > i) t -= Cc.F[k][i] * x[k];
    x[i] = t * Cc.F[i][i];
    __builtin_amdgcn_sched_barrier(0); }
}
// wave 0: masked Cholesky of C + soft I with lane i holding ROW i in twelve registers (right-looking, fully unrolled): the
// pivot and the column below it travel by v_readlane, no LDS round trip and no synchronisation inside the factorisation.
// By symmetry lane i also ends up with column i of L (its registers k > i), so the back substitution needs broadcasts of
// the solution only.  (A lane-per-row version through LDS took 30 k cycles, a one-lane unrolled one 49 k with 200 spills.)
// Leaves the factor in Cc.F (lower, reciprocal pivots on the diagonal), the multipliers in Cc.lam, the active set in Cc.act.
DEVFN double bcast_lane(double x, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), src), hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}
DEVFN void lin_contact_solve_w(LinContact& Cc, const int* stance, double soft, int mode, int lane) {
  const int i = lane < 12 ? lane : 11;
  int actL = stance[0] == 1, actR = stance[1] == 1;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const bool ai = (i < 6) ? actL : actR;
    double F[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const bool aj = (j < 6) ? actL : actR;
      const double c = Cc.C[i > j ? i : j][i > j ? j : i];          // lower triangle, mirrored
      F[j] = (ai && aj) ? c + (i == j ? soft : 0.0) : (i == j ? 1.0 : 0.0);
    }
    double rinv = 1.0;                                               // 1 / L_ii of this lane's row
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const double d = bcast_lane(F[j], j);                          // pivot: element (j, j), held by lane j
      // 1 / sqrt(d): hardware estimate (v_rsq_f64, ~2^-26) + two Newton steps (the library's sqrt followed by a division is
      // ~70 instructions of a dependent chain, at every one of the twelve pivots)
      double ri = __builtin_amdgcn_rsq(d);
      ri = ri * __builtin_fma(-0.5 * d * ri, ri, 1.5);
      ri = ri * __builtin_fma(-0.5 * d * ri, ri, 1.5);
      if (i == j) rinv = ri;
      const double lij = F[j] * ri;                                  // lane i > j: L_ij; lane j itself: sqrt(d)
#pragma unroll
      for (int k = 0; k < 12; ++k) if (k > j) {
        const double lkj = bcast_lane(lij, k);                       // L_kj, held by lane k
        if (i > j) F[k] -= lij * lkj;
      }
      if (i > j) F[j] = lij;
      else if (i == j) {                                             // row j is final: its registers k > j become column j of L
#pragma unroll
        for (int k = 0; k < 12; ++k) if (k > j) F[k] *= ri;
      }
    }
    // forward substitution L y = b, then L^T x = y; x_k broadcast from lane k
    double y = ai ? Cc.b[i] : 0.0;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const double yk = bcast_lane(y * rinv, k);                     // y_k final once all k' < k have been subtracted
      if (i == k) y = yk;
      else if (i > k) y -= F[k] * yk;
    }
    double x = y;
#pragma unroll
    for (int k = 11; k >= 0; --k) {
      const double xk = bcast_lane(x * rinv, k);
      if (i == k) x = xk;
      else if (i < k) x -= F[k] * xk;                                // F[k], k > i: L_ki
    }
    int again = 0;
    if (mode == 2 && pass == 0) {
      const double lx3 = bcast_lane(x, 3), lx4 = bcast_lane(x, 4), lx5 = bcast_lane(x, 5), rx3 = bcast_lane(x, 9), rx4 = bcast_lane(x, 10), rx5 = bcast_lane(x, 11);
      const double fzL = Cc.zl[0][0] * lx3 + Cc.zl[0][1] * lx4 + Cc.zl[0][2] * lx5, fzR = Cc.zl[1][0] * rx3 + Cc.zl[1][1] * rx4 + Cc.zl[1][2] * rx5;
      if (actL && fzL < 0.0) { actL = 0; again = 1; }
      if (actR && fzR < 0.0) { actR = 0; again = 1; }
    }
    if (!again) {
      if (lane < 12) {
        Cc.lam[i] = x;
#pragma unroll
        for (int j = 0; j < 12; ++j) if (j < i) Cc.F[i][j] = F[j];
        Cc.F[i][i] = rinv;
      }
      break;
    }
  }
  if (lane == 0) { Cc.act[0] = actL; Cc.act[1] = actR; }
}
// ---- contact modes 3 / 4: the same solve with the Coulomb limit (h1_aba_split.h stance_correct, oracle forward_dynamics_mj_stance) -------
// lane i = row i of a 12 x 12 SPD matrix in twelve registers -> its Cholesky factor (the factorisation of lin_contact_solve_w as a
// routine): on exit F[j < i] = L_ij, F[k > i] = L_ki (column i of L), rinv = 1 / L_ii
DEVFN void chol12_rows(double (&F)[12], double& rinv, int i) {
  rinv = 1.0;
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const double d = bcast_lane(F[j], j);
    double ri = __builtin_amdgcn_rsq(d);
    ri = ri * __builtin_fma(-0.5 * d * ri, ri, 1.5);
    ri = ri * __builtin_fma(-0.5 * d * ri, ri, 1.5);
    if (i == j) rinv = ri;
    const double lij = F[j] * ri;
#pragma unroll
    for (int k = 0; k < 12; ++k) if (k > j) {
      const double lkj = bcast_lane(lij, k);
      if (i > j) F[k] -= lij * lkj;
    }
    if (i > j) F[j] = lij;
    else if (i == j) {
#pragma unroll
      for (int k = 0; k < 12; ++k) if (k > j) F[k] *= ri;
    }
  }
}
// x_i of (L L^T) x = rhs, rhs_i in lane i
DEVFN double solve12_rows(const double (&F)[12], double rinv, double rhs, int i) {
  double y = rhs;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const double yk = bcast_lane(y * rinv, k);
    if (i == k) y = yk;
    else if (i > k) y -= F[k] * yk;
  }
  double x = y;
#pragma unroll
  for (int k = 11; k >= 0; --k) {
    const double xk = bcast_lane(x * rinv, k);
    if (i == k) x = xk;
    else if (i < k) x -= F[k] * xk;
  }
  return x;
}
DEVFN double dot3_lanes(const double* u, double v, int o) { return u[0] * bcast_lane(v, o) + u[1] * bcast_lane(v, o + 1) + u[2] * bcast_lane(v, o + 2); }
// One wave.  Rigid solve, unilateral release, then the cone check on the feet that still push; a foot outside the cone slides:
//   mode 3:  (Pi (C + soft I) Pi + (I - Pi)) lambda = Pi b,  Pi = blockdiag(I3, u u^T) on a sliding foot (u: up axis in link coordinates)
//   mode 4:  (K + sum_f c_f u_f^T) y = Pi b,  c_f = mu Pi C t_f,  lambda = y on the rotation rows, (u + mu t) (u . y) on the translation rows
// Leaves lambda, the factor of K (Cc.F), the active set, and in Zs / the aliased arrays what the multiplier tangents need.
template <bool KIN>
DEVFN void lin_contact_solve_fr(LinContact& Cc, LinSlide& Zs, const int* stance, double soft, double mu, int lane) {
  const int i = lane < 12 ? lane : 11;
  int actL = stance[0] == 1, actR = stance[1] == 1;
  double F[12], rinv = 1.0, x = 0.0;
  auto masked_row = [&](double (&R)[12]) {
    const bool ai = (i < 6) ? actL : actR;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const bool aj = (j < 6) ? actL : actR;
      const double c = Cc.C[i > j ? i : j][i > j ? j : i];
      R[j] = (ai && aj) ? c + (i == j ? soft : 0.0) : (i == j ? 1.0 : 0.0);
    }
  };
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    masked_row(F);
    chol12_rows(F, rinv, i);
    x = solve12_rows(F, rinv, ((i < 6) ? actL : actR) ? Cc.b[i] : 0.0, i);
    if (pass == 1) break;
    const double fzL = dot3_lanes(Cc.zl[0], x, 3), fzR = dot3_lanes(Cc.zl[1], x, 9);
    int again = 0;
    if (actL && fzL < 0.0) { actL = 0; again = 1; }
    if (actR && fzR < 0.0) { actR = 0; again = 1; }
    if (!again) break;
  }
  const double uL[3] = {Cc.zl[0][0], Cc.zl[0][1], Cc.zl[0][2]}, uR[3] = {Cc.zl[1][0], Cc.zl[1][1], Cc.zl[1][2]};
  double fL[3], fR[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { fL[k] = bcast_lane(x, 3 + k); fR[k] = bcast_lane(x, 9 + k); }
  const double fnL = uL[0] * fL[0] + uL[1] * fL[1] + uL[2] * fL[2], fnR = uR[0] * fR[0] + uR[1] * fR[1] + uR[2] * fR[2];
  const double ft2L = fL[0] * fL[0] + fL[1] * fL[1] + fL[2] * fL[2] - fnL * fnL, ft2R = fR[0] * fR[0] + fR[1] * fR[1] + fR[2] * fR[2] - fnR * fnR;
  const bool slL = actL && ft2L > mu * mu * fnL * fnL, slR = actR && ft2R > mu * mu * fnR * fnR;
  double nL = 0.0, nR = 0.0;
  if (slL || slR) {
    const bool ai = (i < 6) ? actL : actR;
    const bool inL = slL && i >= 3 && i < 6, inR = slR && i >= 9;
    const int a = inL ? i - 3 : (inR ? i - 9 : 0);
    double tL[3] = {0.0, 0.0, 0.0}, tR[3] = {0.0, 0.0, 0.0};
    if (KIN) {
      const double ntL = sqrt(ft2L > 0.0 ? ft2L : 1.0), ntR = sqrt(ft2R > 0.0 ? ft2R : 1.0);
#pragma unroll
      for (int k = 0; k < 3; ++k) { tL[k] = slL ? (fL[k] - fnL * uL[k]) / ntL : 0.0; tR[k] = slR ? (fR[k] - fnR * uR[k]) / ntR : 0.0; }
      if (lane < 12) {
        // the sticking system's factor: transposed into the strict upper triangle of Cc.F
#pragma unroll
        for (int j = 0; j < 12; ++j) if (j < i) Cc.F[j][i] = F[j];
        Zs.Fsd[i] = rinv; Zs.ls[i] = x;
      }
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { Zs.t[0][k] = tL[k]; Zs.t[1][k] = tR[k]; Zs.fs[0][k] = fL[k]; Zs.fs[1][k] = fR[k]; }
        Zs.fn[0] = fnL; Zs.fn[1] = fnR; Zs.nt[0] = ntL; Zs.nt[1] = ntR;
      }
    }
    double R[12]; masked_row(R);
    double bi = ai ? Cc.b[i] : 0.0, cL = 0.0, cR = 0.0;
    if (KIN) {
      double sl_ = 0.0, sr_ = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) { sl_ += Cc.C[i > 3 + k ? i : 3 + k][i > 3 + k ? 3 + k : i] * tL[k]; sr_ += Cc.C[i > 9 + k ? i : 9 + k][i > 9 + k ? 9 + k : i] * tR[k]; }
      cL = (ai && slL) ? mu * sl_ : 0.0; cR = (ai && slR) ? mu * sr_ : 0.0;
    }
    // Pi from the right (columns of the sliding translation blocks)
    if (slL) { const double w = R[3] * uL[0] + R[4] * uL[1] + R[5] * uL[2]; R[3] = w * uL[0]; R[4] = w * uL[1]; R[5] = w * uL[2]; }
    if (slR) { const double w = R[9] * uR[0] + R[10] * uR[1] + R[11] * uR[2]; R[9] = w * uR[0]; R[10] = w * uR[1]; R[11] = w * uR[2]; }
    // Pi from the left (rows), the identity on the removed directions, Pi on the right-hand sides
    if (slL) {
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const double v = dot3_lanes(uL, R[j], 3);
        if (inL) R[j] = uL[a] * v + (j == i ? 1.0 : 0.0) - ((j >= 3 && j < 6) ? uL[a] * uL[j - 3 < 0 ? 0 : (j - 3 > 2 ? 2 : j - 3)] : 0.0);
      }
      const double vb = dot3_lanes(uL, bi, 3), vl = dot3_lanes(uL, cL, 3), vr = dot3_lanes(uL, cR, 3);
      if (inL) { bi = uL[a] * vb; cL = uL[a] * vl; cR = uL[a] * vr; }
    }
    if (slR) {
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const double v = dot3_lanes(uR, R[j], 9);
        if (inR) R[j] = uR[a] * v + (j == i ? 1.0 : 0.0) - ((j >= 9) ? uR[a] * uR[j - 9 < 0 ? 0 : j - 9] : 0.0);
      }
      const double vb = dot3_lanes(uR, bi, 9), vl = dot3_lanes(uR, cL, 9), vr = dot3_lanes(uR, cR, 9);
      if (inR) { bi = uR[a] * vb; cL = uR[a] * vl; cR = uR[a] * vr; }
    }
    chol12_rows(R, rinv, i);
    x = solve12_rows(R, rinv, bi, i);
    if (!KIN) {
      nL = dot3_lanes(uL, x, 3); nR = dot3_lanes(uR, x, 9);
      if (inL) x = uL[a] * nL;        // (the removed directions carry zero up to rounding: exact, as the step does)
      if (inR) x = uR[a] * nR;
    } else {
      const double zL = solve12_rows(R, rinv, cL, i), zR = solve12_rows(R, rinv, cR, i);
      const double aLL = 1.0 + dot3_lanes(uL, zL, 3), aLR = dot3_lanes(uL, zR, 3), aRL = dot3_lanes(uR, zL, 9), aRR = 1.0 + dot3_lanes(uR, zR, 9);
      const double rL = dot3_lanes(uL, x, 3), rR = dot3_lanes(uR, x, 9);
      const double det = aLL * aRR - aLR * aRL;
      const double alL = (rL * aRR - aLR * rR) / det, alR = (aLL * rR - aRL * rL) / det;
      x = x - zL * alL - zR * alR;
      nL = dot3_lanes(uL, x, 3); nR = dot3_lanes(uR, x, 9);
      if (inL) x = (uL[a] + mu * tL[a]) * nL;
      if (inR) x = (uR[a] + mu * tR[a]) * nR;
      if (lane < 12) { Zs.Z[0][i] = zL; Zs.Z[1][i] = zR; }
      if (lane == 0) { Zs.Mi[0] = aRR / det; Zs.Mi[1] = -aLR / det; Zs.Mi[2] = -aRL / det; Zs.Mi[3] = aLL / det; }
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) F[j] = R[j];
    // rho = C lambda - b on the translation rows of the sliding feet (over b, which nobody reads any more)
    double rho = -(ai ? Cc.b[i] : 0.0);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const bool aj = (j < 6) ? actL : actR;
      const double xj = bcast_lane(x, j);
      rho += (ai && aj) ? Cc.C[i > j ? i : j][i > j ? j : i] * xj : 0.0;
    }
    if (lane < 12 && (inL || inR)) Cc.b[i] = rho;
  }
  if (lane < 12) {
    Cc.lam[i] = x;
#pragma unroll
    for (int j = 0; j < 12; ++j) if (j < i) Cc.F[i][j] = F[j];
    Cc.F[i][i] = rinv;
  }
  if (lane == 0) { Cc.act[0] = actL; Cc.act[1] = actR; Zs.sl[0] = slL; Zs.sl[1] = slR; Zs.n[0] = nL; Zs.n[1] = nR; }
}
// x <- (sticking system)^-1 x: its factor is the strict upper triangle of Cc.F (transposed), reciprocal pivots Zs.Fsd
DEVFN void lin_contact_backsolve_s(const LinContact& Cc, const LinSlide& Zs, double* x) {
#pragma unroll
  for (int i = 0; i < 12; ++i) { double t = x[i];
#pragma unroll
    for (int k = 0; k < 12; ++k) if (k < i) t -= Cc.F[k][i] * x[k];
    x[i] = t * Zs.Fsd[i];
    __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
  for (int i = 11; i >= 0; --i) { double t = x[i];
#pragma unroll
    for (int k = 0; k < 12; ++k) if (k > i) t -= Cc.F[i][k] * x[k];
    x[i] = t * Zs.Fsd[i];
    __builtin_amdgcn_sched_barrier(0); }
}
// One multiplier tangent with sliding feet.  On entry w = G^T dT - dR on the active rows (zero elsewhere), dz[g] = tangent of foot g's up
// axis in link coordinates for this direction (zero for a control column); on exit w = dlambda.  With lambda_lin = d n on a sliding foot
// (d = u in mode 3, u + mu t in mode 4) and the constraint rows Pi r = 0, r = C lambda - b (+ soft terms along u):
//   dlambda = Gm dy + e,  e = [0; n dd],   (K + sum_f c_f u_f^T) dy = Pi (w - C e) - [0; u (du . rho)]
// and in mode 4  dt = (I - t t^T) (P_perp df - f_n du - u (du . f)) / |f_t|  from the tangent df of the STICKING solve.
// (df_in: the force parts of the sticking solve's tangent for this direction, [foot][3] at stride df_stride, from the first pass; null
// for a control column, whose right-hand side does not depend on the multipliers: solved here)
template <bool KIN>
DEVFN void lin_slide_tangent(const LinContact& Cc, const LinSlide& Zs, double mu, double* w, const double (*dz)[3], const double* df_in = nullptr, int df_stride = 0) {
  double e[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  if (KIN) {
    double ws[12];
    if (df_in) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int k = 0; k < 3; ++k) ws[6 * g + 3 + k] = df_in[(3 * g + k) * df_stride];
    } else {
#pragma unroll
      for (int j = 0; j < 12; ++j) ws[j] = w[j];
      lin_contact_backsolve_s(Cc, Zs, ws);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) if (Zs.sl[g]) {
      const double* df = ws + 6 * g + 3; const double* z = Cc.zl[g]; const double* t = Zs.t[g]; const double* f = Zs.fs[g];
      const double zdf = z[0] * df[0] + z[1] * df[1] + z[2] * df[2], dzf = dz[g][0] * f[0] + dz[g][1] * f[1] + dz[g][2] * f[2];
      double dtau[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) dtau[k] = df[k] - z[k] * zdf - Zs.fn[g] * dz[g][k] - z[k] * dzf;
      const double tdt = t[0] * dtau[0] + t[1] * dtau[1] + t[2] * dtau[2];
#pragma unroll
      for (int k = 0; k < 3; ++k) e[g][k] = Zs.n[g] * (dz[g][k] + mu * (dtau[k] - t[k] * tdt) / Zs.nt[g]);
    }
  } else {
#pragma unroll
    for (int g = 0; g < 2; ++g) if (Zs.sl[g]) {
#pragma unroll
      for (int k = 0; k < 3; ++k) e[g][k] = Zs.n[g] * dz[g][k];
    }
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) if (Cc.act[i / 6]) {
    double t = w[i];
#pragma unroll
    for (int g = 0; g < 2; ++g) if (Zs.sl[g]) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { const int j = 6 * g + 3 + k; t -= Cc.C[i > j ? i : j][i > j ? j : i] * e[g][k]; }
    }
    w[i] = t;
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) if (Zs.sl[g]) {
    const double* z = Cc.zl[g]; double* wl = w + 6 * g + 3; const double* rho = Cc.b + 6 * g + 3;
    const double s = z[0] * wl[0] + z[1] * wl[1] + z[2] * wl[2] - (dz[g][0] * rho[0] + dz[g][1] * rho[1] + dz[g][2] * rho[2]);
    wl[0] = z[0] * s; wl[1] = z[1] * s; wl[2] = z[2] * s;
  }
  lin_contact_backsolve(Cc, w);
  if (KIN) {
    const double aL = Cc.zl[0][0] * w[3] + Cc.zl[0][1] * w[4] + Cc.zl[0][2] * w[5], aR = Cc.zl[1][0] * w[9] + Cc.zl[1][1] * w[10] + Cc.zl[1][2] * w[11];
    const double alL = Zs.Mi[0] * aL + Zs.Mi[1] * aR, alR = Zs.Mi[2] * aL + Zs.Mi[3] * aR;
#pragma unroll
    for (int i = 0; i < 12; ++i) w[i] -= Zs.Z[0][i] * alL + Zs.Z[1][i] * alR;
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) if (Zs.sl[g]) {
    const double* z = Cc.zl[g]; double* wl = w + 6 * g + 3;
    if (KIN) {
      const double nn = z[0] * wl[0] + z[1] * wl[1] + z[2] * wl[2];
#pragma unroll
      for (int k = 0; k < 3; ++k) wl[k] = (z[k] + mu * Zs.t[g][k]) * nn + e[g][k];
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) wl[k] += e[g][k];
    }
  }
}
// wave 0: constrained accelerations.  dq = G lambda, qacc += dq, aL += R0^T dq_lin, body accelerations a += da (level by
// level), then X a_parent and the body forces with the contact wrenches as external forces.  Wave-local ordering only.
// (mode 4 runs the tangent sweeps twice on a knot with a sliding foot, first about the sticking solution: ldq - lsub = the multipliers
// whose accelerations are added, lf = the contact wrench the body forces carry; defaults: Cc.lam)
template <bool X = false>
DEVFN void lin_contact_correct(LinShared& L, LinContact& Cc, int lane, const double* ldq = nullptr, const double* lsub = nullptr, const double* lf_ = nullptr) {
  LinDump& D = L.D;
  const double* lf = X ? lf_ : Cc.lam;
  if (lane < H1_NV) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 12; ++j) s += Cc.G[lane][j] * (X ? ldq[j] - (lsub ? lsub[j] : 0.0) : Cc.lam[j]);
    Cc.dq[lane] = s;
    D.qacc[lane] += s;
  }
  wave_sync();
  if (lane == 0) {
    double dl[3]; mtv3(D.R0, Cc.dq, dl);
    for (int k = 0; k < 3; ++k) { Cc.da[0][k] = Cc.dq[3 + k]; Cc.da[0][3 + k] = dl[k]; D.aL[k] += dl[k]; }
  }
  wave_sync();
  const int i = lane;
  const int dep = (i >= 1 && i < H1_NB) ? H1_DEPTH[i] : -1;
  for (int d = 1; d <= 5; ++d) {
    if (dep == d) {
      double a[6]; xf_motion(D.Rj[i], H1_POS[i], Cc.da[H1_PARENT[i]], a);
      a[H1_AXIS[i]] += Cc.dq[5 + i];
      for (int k = 0; k < 6; ++k) Cc.da[i][k] = a[k];
    }
    wave_sync();
  }
  if (i < H1_NB) { for (int k = 0; k < 6; ++k) Cc.a[i][k] += Cc.da[i][k]; }
  wave_sync();
  if (i >= 1 && i < H1_NB) {
    double xa[6]; xf_motion(D.Rj[i], H1_POS[i], Cc.a[H1_PARENT[i]], xa);
    for (int k = 0; k < 6; ++k) L.xa[i][k] = xa[k];
  }
  if (i < H1_NB) {
    double Ia[6], vIv[6], Iv[6];
    for (int k = 0; k < 6; ++k) Iv[k] = L.Iv[i][k];
    inertia_mul(i, Cc.a[i], Ia); crf(D.v[i], Iv, vIv);
    const int g = (i == 5) ? 0 : ((i == 10) ? 1 : -1);
    for (int k = 0; k < 6; ++k) D.F[i][k] = Ia[k] + vIv[k] - (g >= 0 ? lf[6 * (g < 0 ? 0 : g) + k] : 0.0);
  }
}

// leg sweeps of the tangent RNEA with the foot's constraint-row tangent: lin_tangent_legs + the foot body's dv, da, and the
// tangent of the rotated gravity offset carried down the leg (a 3-vector: d(R_i^T u) = R_i^T du + [own hinge] (R_i^T u) x e);
// dR = da_foot - (0, dgl) + dv_foot / h per group slot
template <int K> DEVFN void tan_leg_gravity(const LinShared& L, const LinContact& Cc, bool side, int kind, int idx, const double* pg, double* dg_f) {
  constexpr int IL = 1 + K, IR = 6 + K, ax = h1c::C_AXIS[IL];
  const int i = side ? IR : IL;
  double ng[3]; mtv3(L.D.Rj[i], pg, ng);
  const double mt = (kind == DIR_THETA && idx == i) ? 1.0 : 0.0;
  double t[3]; h1r::cross_axis<ax>(Cc.offb[i], t);
  ng[0] += mt * t[0]; ng[1] += mt * t[1]; ng[2] += mt * t[2];
  if constexpr (K + 1 < 5) tan_leg_gravity<K + 1>(L, Cc, side, kind, idx, ng, dg_f);
  else { dg_f[0] = ng[0]; dg_f[1] = ng[1]; dg_f[2] = ng[2]; }
}
// the same recursion for the world up axis (sliding feet: the normal row and the normal force turn with the foot)
template <int K> DEVFN void tan_leg_up(const LinShared& L, const LinSlide& Zs, bool side, int kind, int idx, const double* pz, double* dz_f) {
  constexpr int IL = 1 + K, IR = 6 + K, ax = h1c::C_AXIS[IL];
  const int i = side ? IR : IL;
  double nz[3]; mtv3(L.D.Rj[i], pz, nz);
  const double mt = (kind == DIR_THETA && idx == i) ? 1.0 : 0.0;
  double t[3]; h1r::cross_axis<ax>(Zs.zlb[i - 1], t);
  nz[0] += mt * t[0]; nz[1] += mt * t[1]; nz[2] += mt * t[2];
  if constexpr (K + 1 < 5) tan_leg_up<K + 1>(L, Zs, side, kind, idx, nz, dz_f);
  else { dz_f[0] = nz[0]; dz_f[1] = nz[1]; dz_f[2] = nz[2]; }
}
DEVFN void lin_tangent_legs_c(LinShared& L, LinContact& Cc, int lane) {
  const int grp = lane / 19, q = lane - 19 * grp;
  const bool side = grp == 1;
  if (grp < 2) {
    int kind, idx; slot_direction(false, side, q, kind, idx);
    const int col = dir_lane(kind, idx);
    double dv0[6], da0[6]; tan_base(L, kind, idx, dv0, da0);
    {
      double dg0[3] = {0.0, 0.0, 0.0}, dgf[3];
      if (kind == DIR_PHI) cross_axis(Cc.offb[0], idx, dg0);        // d(R0^T u) = (R0^T u) x dphi
      tan_leg_gravity<0>(L, Cc, side, kind, idx, dg0, dgf);
#pragma unroll
      for (int k = 0; k < 3; ++k) Cc.dR[grp][3 + k][q] = -dgf[k];
    }
    __builtin_amdgcn_sched_barrier(0);
    double df[5][6], dvf[6], daf[6];
    TanChain2<1, 6, 5>::fwd<0>(L, side, kind, idx, dv0, da0, df, dvf, daf);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      Cc.dR[grp][k][q] = daf[k] + dvf[k] / L.h;
      Cc.dR[grp][3 + k][q] += daf[3 + k] + dvf[3 + k] / L.h;
    }
    double dFj[6] = {0, 0, 0, 0, 0, 0};
    double acc[6] = {0, 0, 0, 0, 0, 0};
    TanChain2<1, 6, 5>::bwd<4>(L, side, kind, idx, df, acc, dFj, col);
#pragma unroll
    for (int k = 0; k < 6; ++k) L.u.t.part[grp][k][q] = dFj[k];
  }
}
// multiplier tangents.  wave 0: W = G^T dT (12 x 25 times 25 x 47) on the MFMA -- 7 k-steps x 3 column tiles, A operand lane
// (lr = j, lk) = G[4 s + lk][j], B operand = dT[4 s + lk][16 J + lr] -- written to Cc.W, then lane = direction subtracts the
// constraint-row tangents and back-substitutes; wave 1: lane = control column (19)
// (TWO: the 16-slot chain groups of the two-knot kernel, base-linear-velocity directions in the extra slots 16..18)
DEVFN int slot_in_group2c(int c, int kind, int idx) { return kind == DIR_VLIN ? 16 + idx : slot_in_group2(c, kind, idx); }
// (FRIC = 2: `stash` = 6 x 48 doubles of global scratch per knot; sticking_pass: the sweeps ran about the sticking solution -- solve with
// its factor and leave the force parts of dlambda_s there, nothing else)
template <bool TWO = false, int FRIC = 0>
DEVFN void lin_contact_multipliers(const LinShared& L, LinContact& Cc, int wv, int lane, const LinSlide* Zs = nullptr, double mu = 1.0, double* stash = nullptr, bool sticking_pass = false) {
  typedef double v4d_c __attribute__((ext_vector_type(4)));
  if (wv == 0) {
    const int lr = lane & 15, lk = lane >> 4;
    v4d_c acc[3];
#pragma unroll
    for (int J = 0; J < 3; ++J) acc[J] = (v4d_c){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int sx = 0; sx < 7; ++sx) {
      const int k = 4 * sx + lk, kc = k < H1_NV ? k : H1_NV - 1;
      const double gv = Cc.G[kc][lr < 12 ? lr : 0];
      const double am = (k < H1_NV && lr < 12) ? gv : 0.0;
#pragma unroll
      for (int J = 0; J < 3; ++J) {
        const double tv = L.dT[kc][16 * J + lr];
        acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(am, (k < H1_NV) ? tv : 0.0, acc[J], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {                 // rows j = 4 r + lk < 12
#pragma unroll
      for (int J = 0; J < 3; ++J) Cc.W[4 * r + lk][16 * J + lr] = acc[J][r];
    }
    wave_sync();
    if (lane < LIN_NDIR) {
      int kind, idx; lane_direction(lane, kind, idx);
      double w[12];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int sl = TWO ? slot_in_group2c(g, kind, idx) : slot_in_group(g, kind, idx);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const double dr = sl >= 0 ? Cc.dR[g][k][sl < 0 ? 0 : sl] : 0.0;
          w[6 * g + k] = Cc.act[g] ? Cc.W[6 * g + k][lane] - dr : 0.0;
        }
      }
      if (FRIC == 2 && sticking_pass) {
        lin_contact_backsolve_s(Cc, *Zs, w);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int k = 0; k < 3; ++k) stash[(3 * g + k) * LIN_LD + lane] = w[6 * g + 3 + k];
      } else
      if (FRIC != 0 && (Zs->sl[0] || Zs->sl[1])) {
        double dz[2][3];
        const double (*dZ)[3][16] = slide_dZ(Cc);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int sl = slot_in_group2c(g, kind, idx);
          const bool on = Zs->sl[g] && sl >= 0 && sl < 16;
#pragma unroll
          for (int k = 0; k < 3; ++k) dz[g][k] = on ? dZ[g][k][on ? sl : 0] : 0.0;
        }
        lin_slide_tangent<FRIC == 2>(Cc, *Zs, mu, w, dz, FRIC == 2 ? stash + lane : nullptr, LIN_LD);
      } else
      lin_contact_backsolve(Cc, w);
#pragma unroll
      for (int j = 0; j < 12; ++j) Cc.W[j][lane] = w[j];
    } else if (lane < LIN_LD) {
#pragma unroll
      for (int j = 0; j < 12; ++j) Cc.W[j][lane] = 0.0;            // padding column of the MFMA operand
    }
  }
  if (wv == 1 && lane < H1_NU) {
    double w[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) w[j] = Cc.act[j / 6] ? -Cc.G[6 + lane][j] : 0.0;
    if (FRIC != 0 && (Zs->sl[0] || Zs->sl[1])) {
      const double dz[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
      lin_slide_tangent<FRIC == 2>(Cc, *Zs, mu, w, dz);
    } else
    lin_contact_backsolve(Cc, w);
#pragma unroll
    for (int j = 0; j < 12; ++j) Cc.WU[j][lane] = w[j];
  }
}
// ---- two knots per four-wave workgroup (k_lin_tangent2c, round 4) -------------------------------------------------------------
// As k_lin_tangent2: 16-slot chain groups, so the leg sweeps of two knots fill one wave.  The inverse-dynamics tangent does not
// depend on the base's linear velocity here either (the contact wrench is an external force fixed in link coordinates), but the
// constraint row does -- through the foot's velocity v_f / h and the velocity-product part of its acceleration: pure kinematics.
// The three v_lin directions therefore run a kinematics-only forward chain (lin2_leg_vlin_dR: dv, da down the leg, no inertia,
// no forces, no backward pass) on 2 knots x 2 feet x 3 = 12 lanes of an otherwise idle wave, into the slots 16..18 of dR.
template <bool LIM = false>
DEVFN void lin2_tangent_legs_c(LinShared* L2, LinContact* C2, int lane, const LinSlide* Z2 = nullptr, const double (*lockc2)[H1_NB] = nullptr) {
  LinShared& L = L2[lane >> 5]; LinContact& Cc = C2[lane >> 5];
  const double* lockc = LIM ? lockc2[lane >> 5] : nullptr;
  const int grp = (lane >> 4) & 1, q = lane & 15;
  const bool side = grp == 1;
  int kind, idx; slot_direction2(false, side, q, kind, idx);
  const int col = dir_lane(kind, idx);
  double dv0[6], da0[6]; tan_base(L, kind, idx, dv0, da0);
  {
    double dg0[3] = {0.0, 0.0, 0.0}, dgf[3];
    if (kind == DIR_PHI) cross_axis(Cc.offb[0], idx, dg0);        // d(R0^T u) = (R0^T u) x dphi
    tan_leg_gravity<0>(L, Cc, side, kind, idx, dg0, dgf);
#pragma unroll
    for (int k = 0; k < 3; ++k) Cc.dR[grp][3 + k][q] = -dgf[k];
  }
  if (Z2) {
    const double zb[3] = {L.D.R0[6], L.D.R0[7], L.D.R0[8]};
    double dz0[3] = {0.0, 0.0, 0.0}, dzf[3];
    if (kind == DIR_PHI) cross_axis(zb, idx, dz0);
    tan_leg_up<0>(L, Z2[lane >> 5], side, kind, idx, dz0, dzf);
    double (*dZ)[3][16] = slide_dZ(Cc);
#pragma unroll
    for (int k = 0; k < 3; ++k) dZ[grp][k][q] = dzf[k];
  }
  __builtin_amdgcn_sched_barrier(0);
  double df[5][6], dvf[6], daf[6];
  TanChain2<1, 6, 5>::template fwd<0, LIM>(L, side, kind, idx, dv0, da0, df, dvf, daf, lockc);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    Cc.dR[grp][k][q] = daf[k] + dvf[k] / L.h;
    Cc.dR[grp][3 + k][q] += daf[3 + k] + dvf[3 + k] / L.h;
  }
  double dFj[6] = {0, 0, 0, 0, 0, 0};
  double acc[6] = {0, 0, 0, 0, 0, 0};
  TanChain2<1, 6, 5>::bwd<4>(L, side, kind, idx, df, acc, dFj, col);
#pragma unroll
  for (int k = 0; k < 6; ++k) L.u.t.part[grp][k][q] = dFj[k];
}
// kinematics-only tangent of (v_i, a_i) down a leg for a direction that moves no hinge (tan_body_fwd2 without its own-hinge terms)
template <int K> DEVFN void tan_leg_kin(const LinShared& L, bool side, const double* pv, const double* pa, double* dvf, double* daf) {
  constexpr int IL = 1 + K, IR = 6 + K, ax = h1c::C_AXIS[IL];
  const int i = side ? IR : IL;
  const double qd = L.x[H1_NQ + 6 + i - 1];
  const double r[3] = {side ? h1c::C_POS[IR][0] : h1c::C_POS[IL][0], side ? h1c::C_POS[IR][1] : h1c::C_POS[IL][1], side ? h1c::C_POS[IR][2] : h1c::C_POS[IL][2]};
  double dv[6], da[6];
  xf_motion(L.D.Rj[i], r, pv, dv);
  xf_motion(L.D.Rj[i], r, pa, da);
  double t[3];
  h1r::cross_axis<ax>(dv, t);     da[0] += qd * t[0]; da[1] += qd * t[1]; da[2] += qd * t[2];
  h1r::cross_axis<ax>(dv + 3, t); da[3] += qd * t[0]; da[4] += qd * t[1]; da[5] += qd * t[2];
  if constexpr (K + 1 < 5) tan_leg_kin<K + 1>(L, side, dv, da, dvf, daf);
  else {
#pragma unroll
    for (int k = 0; k < 6; ++k) { dvf[k] = dv[k]; daf[k] = da[k]; }
  }
}
// lanes 0..11: (knot slot, foot, base-linear-velocity direction) -> dR slots 16..18 (no gravity-offset part: v_lin turns nothing)
DEVFN void lin2_leg_vlin_dR(LinShared* L2, LinContact* C2, int lane) {
  if (lane >= 12) return;
  const int ks = lane / 6, g = (lane / 3) & 1, k3 = lane % 3;
  LinShared& L = L2[ks]; LinContact& Cc = C2[ks];
  double dv0[6], da0[6]; tan_base(L, DIR_VLIN, k3, dv0, da0);
  double dvf[6], daf[6];
  tan_leg_kin<0>(L, g == 1, dv0, da0, dvf, daf);
#pragma unroll
  for (int k = 0; k < 6; ++k) Cc.dR[g][k][16 + k3] = daf[k] + dvf[k] / L.h;
}

// dT <- -Minv dT + G W on the MFMA (row tile I = wave index): the 7 k-steps of the Minv product + 3 of the G product
DEVFN void lin_apply_minv_2c(LinShared& L, const LinContact& Cc, int tid) {
  typedef double v4d_l __attribute__((ext_vector_type(4)));
  const int I = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lk = lane >> 4;
  double am[10], bd[3][10];
  const int r = 16 * I + lr, rc = r < H1_NV ? r : H1_NV - 1;
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const int k = 4 * s + lk, kc = k < H1_NV ? k : H1_NV - 1;
    const double v = L.Minv[MINV_IDX(rc, kc)];
    am[s] = (r < H1_NV && k < H1_NV) ? -v : 0.0;
#pragma unroll
    for (int J = 0; J < 3; ++J) {
      const double w = L.dT[kc][16 * J + lr];
      bd[J][s] = (k < H1_NV) ? w : 0.0;
    }
  }
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int k = 4 * s + lk;
    const double v = Cc.G[rc][k];
    am[7 + s] = (r < H1_NV) ? v : 0.0;
#pragma unroll
    for (int J = 0; J < 3; ++J) bd[J][7 + s] = Cc.W[k][16 * J + lr];
  }
  __syncthreads();
  v4d_l acc[3];
#pragma unroll
  for (int J = 0; J < 3; ++J) acc[J] = (v4d_l){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 10; ++s)
#pragma unroll
    for (int J = 0; J < 3; ++J) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[s], bd[J][s], acc[J], 0, 0, 0);
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int row = 16 * I + 4 * rr + lk;
    if (row < H1_NV) {
#pragma unroll
      for (int J = 0; J < 3; ++J) L.dT[row][16 * J + lr] = acc[J][rr];
    }
  }
}

}  // namespace h1
