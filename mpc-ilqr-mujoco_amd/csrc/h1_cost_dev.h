// Device-side cost terms of the H1 iLQR (fp64, gfx950).
//
// Replaces, without a symbolic engine on the GPU,
//   iLQR::computeTotalCost          reference src/ilqr/ilqr.cpp:363-518
//   iLQR::computeCostQuadratics     reference src/ilqr/ilqr.cpp:133-244
//   iLQR::add*CostDerivatives       reference src/ilqr/ilqr.cpp:662-800
//   symDerivatives::sym*            reference src/common/derivatives.cpp:525-707 (CasADi gradients and
//                                   exact Hessians of Pinocchio quantities)
//   RobotUtils::constraint*         reference src/common/robot_utils.cpp:615-778
//
// The six task terms are functions of a mass-normalised point set S attached to the URDF tree:
//   c(q)     = mfrac p + R(quat) beta(theta)          (whole-body CoM, or an ankle origin)
//   cdot(q,v)= R(quat) gamma(theta, v)                (its velocity, Pinocchio velocity convention)
// with, per hinge j (all in the pelvis frame): axis z_j, origin p_j, r_j = hsub_j - msub_j p_j,
//   w_j = z_j x r_j = d beta/d theta_j,  Om_j = body angular velocity,  s_j = sum_{k below j} qd_k w_k.
// Every gradient/Hessian entry is then a closed form in these vectors (derivation in DESIGN.md):
//   d2 beta/d theta_a d theta_b = z_lo x w_hi for related joints (lo = ancestor-or-self), else 0
//   d gamma/d theta_k           = Om_k x w_k + z_k x s_k
//   d2 gamma/d theta_a d theta_b= Om_lo x (z_lo x w_hi) + z_lo x ((Om_hi - Om_lo) x w_hi) + z_lo x (z_hi x s_hi)
// Quaternion slots use R = Eigen's toRotationMatrix polynomial on the raw (x,y,z,w) coefficients and
// the reference's slot conventions (permuted state in, un-permuted slots out).
#pragma once
#include "h1_dynamics_dev.h"

namespace h1 {

struct ProblemDev {
  int N;
  DynParams dyn;
  double Q[H1_NX], R[H1_NU], Qf[H1_NX];
  double w_com, w_com_vel, w_ee_pos, w_ee_vel, w_upright, w_balance, w_joint, w_ctrl;
  // reference sets: stride 0 when shared by all rollouts
  const double* x_ref;       long x_ref_stride;        // [(N+1)*51]
  const double* u_ref;       long u_ref_stride;        // [N*19]
  const double* com_ref;     long com_ref_stride;      // [(N+1)*3]
  const int* stance;         long stance_stride;       // [(N+1)*2]
  const double* ee_ref;      long ee_ref_stride;       // [(N+1)*2*3]
  const double* com_vel_ref; long com_vel_ref_stride;  // [(N+1)*3]
};

DEVFN void limit_bounds(const double* range, double& lo, double& hi) {
  const double margin = 0.1 * (range[1] - range[0]);
  lo = range[0] + margin; hi = range[1] - margin;
}
DEVFN double joint_penalty(const ProblemDev& P, const double* x) {
  double c = 0.0;
  for (int i = 0; i < H1_NJ; ++i) {
    double lo, hi; limit_bounds(H1_JRANGE[i], lo, hi);
    const double q = x[7 + i];
    if (q > hi) { const double v = q - hi; c += P.w_joint * v * v; }
    if (q < lo) { const double v = lo - q; c += P.w_joint * v * v; }
  }
  return c;
}
DEVFN double ctrl_penalty(const ProblemDev& P, const double* u) {
  double c = 0.0;
  for (int i = 0; i < H1_NU; ++i) {
    double lo, hi; limit_bounds(H1_CTRLRANGE[i], lo, hi);
    if (u[i] > hi) { const double v = u[i] - hi; c += P.w_ctrl * v * v; }
    if (u[i] < lo) { const double v = lo - u[i]; c += P.w_ctrl * v * v; }
  }
  return c;
}
DEVFN bool support_point(const ProblemDev& P, int b, int t, double* ps) {
  const int* st = P.stance + b * P.stance_stride + 2 * t;
  const double* ee = P.ee_ref + b * P.ee_ref_stride + t * 6;
  const bool L = st[0] == 1, Rt = st[1] == 1;
  if (L && Rt) { ps[0] = 0.5 * (ee[0] + ee[3]); ps[1] = 0.5 * (ee[1] + ee[4]); return true; }
  if (L) { ps[0] = ee[0]; ps[1] = ee[1]; return true; }
  if (Rt) { ps[0] = ee[3]; ps[1] = ee[4]; return true; }
  return false;
}
// one knot of computeTotalCost (ilqr.cpp:370-443 / 447-510) including its share of the penalties (512-515)
template <class ComFn>
DEVFN double knot_cost_t(const ProblemDev& P, int b, int t, const double* x, const double* u /*null at t==N*/, ComFn com_fn) {
  const bool term = (t == P.N);
  const double* xr = P.x_ref + b * P.x_ref_stride + t * H1_NX;
  const double* Qd = term ? P.Qf : P.Q;
  double a = 0.0;
  for (int i = 0; i < H1_NX; ++i) { const double e = x[i] - xr[i]; a += e * Qd[i] * e; }
  double c = 0.5 * a;
  if (!term) {
    const double* ur = P.u_ref + b * P.u_ref_stride + t * H1_NU;
    double s = 0.0;
    for (int i = 0; i < H1_NU; ++i) { const double e = u[i] - ur[i]; s += e * P.R[i] * e; }
    c += 0.5 * s;
  }
  if (P.w_upright > 0.0) {
    const double qw = x[3], qx = x[4], qy = x[5], qz = x[6];
    const double zx = 2.0 * (qx * qz + qw * qy), zy = 2.0 * (qy * qz - qw * qx), zz = 1.0 - 2.0 * (qx * qx + qy * qy);
    c += 0.5 * P.w_upright * (zx * zx + zy * zy + (zz - 1.0) * (zz - 1.0));
  }
  if (P.w_balance > 0.0) {
    double ps[2];
    if (support_point(P, b, t, ps)) {
      double com[3]; com_fn(x, com);
      const double om = sqrt(com[2] / 9.81);
      const double rx = com[0] + x[H1_NQ] * om - ps[0], ry = com[1] + x[H1_NQ + 1] * om - ps[1];
      c += 0.5 * P.w_balance * (rx * rx + ry * ry);
    }
  }
  c += joint_penalty(P, x);
  if (!term) c += ctrl_penalty(P, u);
  return c;
}
struct ComLoop { DEVFN void operator()(const double* x, double* com) const { com_mj(x, com); } };
__device__ inline double knot_cost(const ProblemDev& P, int b, int t, const double* x, const double* u) { return knot_cost_t(P, b, t, x, u, ComLoop()); }

// ------------------------------------------------------------------ closed-form task-term quadratics
DEVFN double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
DEVFN void cross(const double* a, const double* b, double* c) { cross3(a, b, c); }

// dR/dquat_k (k over x,y,z,w) of Eigen's toRotationMatrix polynomial, linear in q
DEVFN void dR_dquat(int k, const double* q, double* D) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  if (k == 0) { D[0] = 0; D[1] = 2 * y; D[2] = 2 * z; D[3] = 2 * y; D[4] = -4 * x; D[5] = -2 * w; D[6] = 2 * z; D[7] = 2 * w; D[8] = -4 * x; }
  else if (k == 1) { D[0] = -4 * y; D[1] = 2 * x; D[2] = 2 * w; D[3] = 2 * x; D[4] = 0; D[5] = 2 * z; D[6] = -2 * w; D[7] = 2 * z; D[8] = -4 * y; }
  else if (k == 2) { D[0] = -4 * z; D[1] = -2 * w; D[2] = 2 * x; D[3] = 2 * w; D[4] = -4 * z; D[5] = 2 * y; D[6] = 2 * x; D[7] = 2 * y; D[8] = 0; }
  else { D[0] = 0; D[1] = -2 * z; D[2] = 2 * y; D[3] = 2 * z; D[4] = 0; D[5] = -2 * x; D[6] = -2 * y; D[7] = 2 * x; D[8] = 0; }
}
DEVFN void d2R_dquat2(int k, int l, double* D) { double e[4] = {0, 0, 0, 0}; e[l] = 1.0; dR_dquat(k, e, D); }

struct PointSetDev {
  double beta[3], gamma[3], mfrac;
  double w[H1_NB][3], s[H1_NB][3], dgam[H1_NB][3];  // dgam = d gamma / d theta_j
  int on[H1_NB];
};
struct KnotKin {
  double xp[H1_NX];
  double R0[9], D[4][9];
  double Rh[H1_NB][9], ph[H1_NB][3], zh[H1_NB][3], Om[H1_NB][3];
  PointSetDev S[3];          // 0 = whole-body CoM (URDF masses), 1 = left ankle origin, 2 = right ankle origin
  double Jc[3][3][H1_NX];    // d c / d x_p
  double Jv[3][3][H1_NX];    // d cdot / d x_p
};
// one weighted linear functional of c or cdot whose Hessian is needed: phi = vec^T c  or  vec^T cdot
struct HessCtx {
  int set, is_vel;
  double scale;
  double vec[3], til[3], Dv[4][3];
};

// serial part: pelvis-frame kinematics of the URDF tree (Pinocchio conventions)
__device__ inline void knot_base_kin(const double* x, KnotKin& K) {
  for (int i = 0; i < H1_NX; ++i) K.xp[i] = x[i];
  K.xp[3] = x[4]; K.xp[4] = x[5]; K.xp[5] = x[6]; K.xp[6] = x[3];   // derivatives.cpp:12-24
  const double qx = K.xp[3], qy = K.xp[4], qz = K.xp[5], qw = K.xp[6];
  {
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    K.R0[0] = 1 - (tyy + tzz); K.R0[1] = txy - twz; K.R0[2] = txz + twy;
    K.R0[3] = txy + twz; K.R0[4] = 1 - (txx + tzz); K.R0[5] = tyz - twx;
    K.R0[6] = txz - twy; K.R0[7] = tyz + twx; K.R0[8] = 1 - (txx + tyy);
  }
  for (int k = 0; k < 4; ++k) dR_dquat(k, K.xp + 3, K.D[k]);
  for (int k = 0; k < 9; ++k) K.Rh[0][k] = (k % 4 == 0) ? 1.0 : 0.0;
  for (int k = 0; k < 3; ++k) { K.ph[0][k] = 0.0; K.zh[0][k] = 0.0; K.Om[0][k] = K.xp[H1_NQ + 3 + k]; }
  for (int i = 1; i < H1_NB; ++i) {
    const int p = H1_PARENT[i], a = H1_AXIS[i];
    double Rj[9]; joint_rot(i, K.xp[7 + i - 1], H1U_RFIX, Rj);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) K.Rh[i][3 * r + c] = K.Rh[p][3 * r] * Rj[c] + K.Rh[p][3 * r + 1] * Rj[3 + c] + K.Rh[p][3 * r + 2] * Rj[6 + c];
    double t[3]; mv3(K.Rh[p], H1U_POS[i], t);
    const double qd = K.xp[H1_NQ + 6 + i - 1];
    for (int k = 0; k < 3; ++k) {
      K.ph[i][k] = K.ph[p][k] + t[k];
      K.zh[i][k] = K.Rh[i][3 * k + a];
      K.Om[i][k] = K.Om[p][k] + K.zh[i][k] * qd;
    }
  }
}
// point set `which`: subtree aggregates, beta, gamma, w, s, dgam
__device__ inline void knot_point_set(KnotKin& K, int which) {
  PointSetDev& S = K.S[which];
  double msub[H1_NB], hsub[H1_NB][3], sw[H1_NB][3];
  double mtot = 0.0;
  if (which == 0) for (int i = 0; i < H1_NB; ++i) mtot += H1U_MASS[i]; else mtot = 1.0;
  const int eb = (which == 1) ? H1_EE_LEFT : H1_EE_RIGHT;
  for (int i = 0; i < H1_NB; ++i) { msub[i] = 0.0; for (int k = 0; k < 3; ++k) { hsub[i][k] = 0.0; sw[i][k] = 0.0; } }
  double mall = 0.0;
  for (int i = H1_NB - 1; i >= 0; --i) {
    const double mu = (which == 0) ? H1U_MASS[i] / mtot : ((i == eb) ? 1.0 : 0.0);
    double ch[3] = {0.0, 0.0, 0.0};
    if (which == 0) mv3(K.Rh[i], H1U_COM[i], ch);
    msub[i] += mu; mall += mu;
    for (int k = 0; k < 3; ++k) hsub[i][k] += mu * (K.ph[i][k] + ch[k]);
    if (i > 0) { const int p = H1_PARENT[i]; msub[p] += msub[i]; for (int k = 0; k < 3; ++k) hsub[p][k] += hsub[i][k]; }
  }
  S.mfrac = mall;
  for (int k = 0; k < 3; ++k) S.beta[k] = hsub[0][k];
  const double* vb = K.xp + H1_NQ; const double* wb = K.xp + H1_NQ + 3;
  double wxb[3]; cross(wb, S.beta, wxb);
  for (int k = 0; k < 3; ++k) S.gamma[k] = S.mfrac * vb[k] + wxb[k];
  S.on[0] = 1;
  for (int i = 1; i < H1_NB; ++i) {
    S.on[i] = msub[i] > 0.0 ? 1 : 0;
    double r[3]; for (int k = 0; k < 3; ++k) r[k] = hsub[i][k] - msub[i] * K.ph[i][k];
    cross(K.zh[i], r, S.w[i]);
    const double qd = K.xp[H1_NQ + 6 + i - 1];
    for (int k = 0; k < 3; ++k) S.gamma[k] += qd * S.w[i][k];
  }
  for (int i = H1_NB - 1; i >= 1; --i) {
    const double qd = K.xp[H1_NQ + 6 + i - 1];
    const int p = H1_PARENT[i];
    for (int k = 0; k < 3; ++k) { S.s[i][k] = sw[i][k]; sw[p][k] += sw[i][k] + qd * S.w[i][k]; }
    double a[3], b[3]; cross(K.Om[i], S.w[i], a); cross(K.zh[i], S.s[i], b);
    for (int k = 0; k < 3; ++k) S.dgam[i][k] = a[k] + b[k];
  }
}
// column `c` (0..50) of the Jacobians of c and cdot for point set `which`
__device__ inline void knot_jac_column(KnotKin& K, int which, int c) {
  const PointSetDev& S = K.S[which];
  double jc[3] = {0, 0, 0}, jv[3] = {0, 0, 0};
  if (c < 3) { jc[c] = S.mfrac; }
  else if (c < 7) { mv3(K.D[c - 3], S.beta, jc); mv3(K.D[c - 3], S.gamma, jv); }
  else if (c < H1_NQ) { const int j = c - 7 + 1; if (S.on[j]) { mv3(K.R0, S.w[j], jc); mv3(K.R0, S.dgam[j], jv); } }
  else {
    const int cv = c - H1_NQ;
    double col[3] = {0, 0, 0};
    if (cv < 3) col[cv] = S.mfrac;
    else if (cv < 6) {  // -[beta]x column
      const int k = cv - 3; const double* b = S.beta;
      if (k == 0) { col[1] = -b[2]; col[2] = b[1]; } else if (k == 1) { col[0] = b[2]; col[2] = -b[0]; } else { col[0] = -b[1]; col[1] = b[0]; }
    } else { const int j = cv - 6 + 1; if (S.on[j]) { col[0] = S.w[j][0]; col[1] = S.w[j][1]; col[2] = S.w[j][2]; } }
    mv3(K.R0, col, jv);
  }
  for (int r = 0; r < 3; ++r) { K.Jc[which][r][c] = jc[r]; K.Jv[which][r][c] = jv[r]; }
}
DEVFN void make_ctx(const KnotKin& K, HessCtx& C, int set, int is_vel, double scale, const double* vec) {
  C.set = set; C.is_vel = is_vel; C.scale = scale;
  for (int k = 0; k < 3; ++k) C.vec[k] = vec[k];
  mtv3(K.R0, vec, C.til);
  for (int k = 0; k < 4; ++k) mtv3(K.D[k], vec, C.Dv[k]);
}
DEVFN bool related(int a, int b, int& lo, int& hi) {
  if (H1_ANC[a - 1][b - 1]) { lo = a; hi = b; return true; }
  if (H1_ANC[b - 1][a - 1]) { lo = b; hi = a; return true; }
  return false;
}
// coordinate classes of x_p: 0 p(0-2) | 1 quat(3-6) | 2 theta(7-25) | 3 v_b(26-28) | 4 omega_b(29-31) | 5 thetadot(32-50)
DEVFN int coord_class(int a) { return a < 3 ? 0 : (a < 7 ? 1 : (a < H1_NQ ? 2 : (a < H1_NQ + 3 ? 3 : (a < H1_NQ + 6 ? 4 : 5)))); }

// d2/dx_a dx_b of phi = vec^T (mfrac p + R0 beta)
DEVFN double hess_pos_entry(const KnotKin& K, const HessCtx& C, int a, int b) {
  const PointSetDev& S = K.S[C.set];
  int ca = coord_class(a), cb = coord_class(b);
  if (ca > cb) { int t = a; a = b; b = t; t = ca; ca = cb; cb = t; }
  if (ca == 1 && cb == 1) { double D2[9], t[3]; d2R_dquat2(a - 3, b - 3, D2); mv3(D2, S.beta, t); return dot3(C.vec, t); }
  if (ca == 1 && cb == 2) { const int j = b - 7 + 1; return S.on[j] ? dot3(C.Dv[a - 3], S.w[j]) : 0.0; }
  if (ca == 2 && cb == 2) {
    const int ja = a - 7 + 1, jb = b - 7 + 1; int lo, hi;
    if (!S.on[ja] || !S.on[jb] || !related(ja, jb, lo, hi)) return 0.0;
    double t[3]; cross(K.zh[lo], S.w[hi], t); return dot3(C.til, t);
  }
  return 0.0;
}
// d2/dx_a dx_b of phi = vec^T R0 gamma
DEVFN double hess_vel_entry(const KnotKin& K, const HessCtx& C, int a, int b) {
  const PointSetDev& S = K.S[C.set];
  int ca = coord_class(a), cb = coord_class(b);
  if (ca > cb) { int t = a; a = b; b = t; t = ca; ca = cb; cb = t; }
  if (ca == 0 || ca >= 3) return 0.0;   // p rows and the v-v block vanish
  if (ca == 1) {
    const int k = a - 3;
    if (cb == 1) { double D2[9], t[3]; d2R_dquat2(k, b - 3, D2); mv3(D2, S.gamma, t); return dot3(C.vec, t); }
    if (cb == 2) { const int j = b - 7 + 1; return S.on[j] ? dot3(C.Dv[k], S.dgam[j]) : 0.0; }
    if (cb == 3) return C.Dv[k][b - H1_NQ] * S.mfrac;
    if (cb == 4) {  // Dv . (-[beta]x e_c) = (beta x Dv)_c
      double t[3]; cross(S.beta, C.Dv[k], t); return t[b - H1_NQ - 3];
    }
    const int j = b - H1_NQ - 6 + 1; return S.on[j] ? dot3(C.Dv[k], S.w[j]) : 0.0;
  }
  // ca == 2
  const int ja = a - 7 + 1;
  if (!S.on[ja]) return 0.0;
  if (cb == 3) return 0.0;
  if (cb == 4) { double t[3]; cross(S.w[ja], C.til, t); return t[b - H1_NQ - 3]; }
  if (cb == 5) {
    const int jb = b - H1_NQ - 6 + 1; int lo, hi;
    if (!S.on[jb] || !related(ja, jb, lo, hi)) return 0.0;
    double t[3]; cross(K.zh[lo], S.w[hi], t); return dot3(C.til, t);
  }
  // theta-theta
  const int jb = b - 7 + 1; int lo, hi;
  if (!S.on[jb] || !related(ja, jb, lo, hi)) return 0.0;
  double zw[3], t1[3], dO[3], t2a[3], t2[3], t3a[3], t3[3];
  cross(K.zh[lo], S.w[hi], zw);
  cross(K.Om[lo], zw, t1);
  for (int r = 0; r < 3; ++r) dO[r] = K.Om[hi][r] - K.Om[lo][r];
  cross(dO, S.w[hi], t2a); cross(K.zh[lo], t2a, t2);
  cross(K.zh[hi], S.s[hi], t3a); cross(K.zh[lo], t3a, t3);
  return C.til[0] * (t1[0] + t2[0] + t3[0]) + C.til[1] * (t1[1] + t2[1] + t3[1]) + C.til[2] * (t1[2] + t2[2] + t3[2]);
}

}  // namespace h1
