// TEST INFRASTRUCTURE (oracle) -- CPU restatement of the reference's cost path; not shipped.
//
// Restates
//   * iLQR::computeTotalCost            /root/reference/src/ilqr/ilqr.cpp:363-518  (line-search cost)
//   * iLQR::computeCostQuadratics       /root/reference/src/ilqr/ilqr.cpp:133-244
//   * iLQR::add*CostDerivatives         /root/reference/src/ilqr/ilqr.cpp:662-800
//   * symDerivatives::sym{CoMPos,CoMVel,EEPos,EEVel,Upright,Balance}
//                                       /root/reference/src/common/derivatives.cpp:525-707
//   * RobotUtils::constraint{Cost,Gradients,Hessians}  /root/reference/src/common/robot_utils.cpp:615-778
//
// The six task terms are differentiated by the reference with CasADi over a Pinocchio<SX> model
// (third-party, not in /root/reference => "parity unpinned").  The semantics restated here
// (SURVEY.md Appendix F): the input is the PERMUTED state x_p (quat stored x,y,z,w at slots 3..6,
// derivatives.cpp:12-24), derivatives are plain partials w.r.t. the 51 raw coordinates with the
// free-flyer rotation taken as the quaternion polynomial R = I + 2[...] (Eigen toRotationMatrix),
// velocities are interpreted in Pinocchio's convention (base linear velocity in the BODY frame),
// masses/CoM offsets come from the URDF, and the resulting gradient/Hessian slots are added to
// lx_/lxx_ WITHOUT un-permuting (ilqr.cpp:670-671 etc.).  All of these quirks are reproduced.
//
// Two implementations live here: templated scalar costs (differentiated exactly with ad.hpp) and
// hand-derived closed forms (fast; used for the CPU baseline).  tests/ check that they agree.
#pragma once
#include <cmath>
#include <vector>

#include "h1_dynamics.hpp"

namespace orc {

struct Problem {
  int N = 25;
  DynParams dyn{0.02, {0.0, 0.0, -1.0}};
  double Q[H1_NX], R[H1_NU], Qf[H1_NX];
  double w_com = 0, w_com_vel = 0, w_ee_pos = 0, w_ee_vel = 0, w_upright = 0, w_balance = 0;
  double w_joint = 500.0, w_ctrl = 1000.0;  // robot_utils.cpp:10 defaults
  std::vector<double> x_ref, u_ref, com_ref;       // [(N+1)*51], [N*19], [(N+1)*3]
  std::vector<int> stance;                         // [(N+1)*2], horizon-local index
  std::vector<double> ee_ref, com_vel_ref;         // [(N+1)*2*3], [(N+1)*3]
};

// ---------------- RobotUtils::constraintCost / Gradients / Hessians ----------------
template <class R> inline void limit_bounds(const R* range, double& lo, double& hi) {   // (R: the model table's scalar)
  double margin = 0.1 * (range[1] - range[0]);
  lo = range[0] + margin; hi = range[1] - margin;
}
inline double constraint_cost(const Problem& P, const double* x, const double* u) {
  double c = 0.0;
  for (int i = 0; i < H1_NU; ++i) {
    double lo, hi; limit_bounds(H1_CTRLRANGE[i], lo, hi);
    if (u[i] > hi) { double v = u[i] - hi; c += P.w_ctrl * v * v; }
    if (u[i] < lo) { double v = lo - u[i]; c += P.w_ctrl * v * v; }
  }
  for (int i = 0; i < H1_NJ; ++i) {
    double lo, hi; limit_bounds(H1_JRANGE[i], lo, hi);
    double q = x[7 + i];
    if (q > hi) { double v = q - hi; c += P.w_joint * v * v; }
    if (q < lo) { double v = lo - q; c += P.w_joint * v * v; }
  }
  return c;
}
// adds to lx (51), lu (19), diag of lxx (51x51 row-major), diag of luu (19)
inline void constraint_derivs(const Problem& P, const double* x, const double* u, double* lx, double* lu,
                              double* lxx, double* luu_diag) {
  if (u) for (int i = 0; i < H1_NU; ++i) {
    double lo, hi; limit_bounds(H1_CTRLRANGE[i], lo, hi);
    if (u[i] > hi) lu[i] += 2.0 * P.w_ctrl * (u[i] - hi);
    if (u[i] < lo) lu[i] += -2.0 * P.w_ctrl * (lo - u[i]);
    if (u[i] > hi || u[i] < lo) luu_diag[i] += 2.0 * P.w_ctrl;
  }
  for (int i = 0; i < H1_NJ; ++i) {
    double lo, hi; limit_bounds(H1_JRANGE[i], lo, hi);
    double q = x[7 + i];
    if (q > hi) lx[7 + i] += 2.0 * P.w_joint * (q - hi);
    if (q < lo) lx[7 + i] += -2.0 * P.w_joint * (lo - q);
    if (q > hi || q < lo) lxx[(7 + i) * H1_NX + (7 + i)] += 2.0 * P.w_joint;
  }
}

// support point of the capture-point term (ilqr.cpp:403-437, 767-791); returns false when no stance
inline bool support_point(const Problem& P, int t, double* ps) {
  bool L = P.stance[2 * t] == 1, Rt = P.stance[2 * t + 1] == 1;
  const double* l = &P.ee_ref[(t * 2 + 0) * 3];
  const double* r = &P.ee_ref[(t * 2 + 1) * 3];
  if (L && Rt) { ps[0] = 0.5 * (l[0] + r[0]); ps[1] = 0.5 * (l[1] + r[1]); return true; }
  if (L) { ps[0] = l[0]; ps[1] = l[1]; return true; }
  if (Rt) { ps[0] = r[0]; ps[1] = r[1]; return true; }
  return false;
}

// ---------------- iLQR::computeTotalCost (ilqr.cpp:363-518) ----------------
inline double stage_extra_cost(const Problem& P, const double* x, int t) {
  double c = 0.0;
  if (P.w_upright > 0.0) {
    double qw = x[3], qx = x[4], qy = x[5], qz = x[6];
    double zx = 2.0 * (qx * qz + qw * qy), zy = 2.0 * (qy * qz - qw * qx), zz = 1.0 - 2.0 * (qx * qx + qy * qy);
    c += 0.5 * P.w_upright * (zx * zx + zy * zy + (zz - 1.0) * (zz - 1.0));
  }
  if (P.w_balance > 0.0) {
    double ps[2];
    if (support_point(P, t, ps)) {
      double com[3]; com_mj(x, com);
      double om = std::sqrt(com[2] / 9.81);
      double rx = com[0] + x[H1_NQ + 0] * om - ps[0], ry = com[1] + x[H1_NQ + 1] * om - ps[1];
      c += 0.5 * P.w_balance * (rx * rx + ry * ry);
    }
  }
  return c;
}
inline double total_cost(const Problem& P, const double* xs, const double* us) {
  double c = 0.0;
  const int N = P.N;
  for (int t = 0; t < N; ++t) {
    const double* x = xs + t * H1_NX; const double* u = us + t * H1_NU;
    double a = 0.0, b = 0.0;
    for (int i = 0; i < H1_NX; ++i) { double e = x[i] - P.x_ref[t * H1_NX + i]; a += e * P.Q[i] * e; }
    for (int i = 0; i < H1_NU; ++i) { double e = u[i] - P.u_ref[t * H1_NU + i]; b += e * P.R[i] * e; }
    c += 0.5 * a; c += 0.5 * b;
    c += stage_extra_cost(P, x, t);
  }
  {
    const double* x = xs + N * H1_NX; double a = 0.0;
    for (int i = 0; i < H1_NX; ++i) { double e = x[i] - P.x_ref[N * H1_NX + i]; a += e * P.Qf[i] * e; }
    c += 0.5 * a;
    c += stage_extra_cost(P, x, N);
  }
  double zero_u[H1_NU] = {0};
  for (int t = 0; t < N; ++t) c += constraint_cost(P, xs + t * H1_NX, us + t * H1_NU);
  c += constraint_cost(P, xs + N * H1_NX, zero_u);
  return c;
}

// ---------------- Pinocchio-side kinematics, templated (for AD) ----------------
// Eigen::Quaternion::toRotationMatrix polynomial on the raw (x,y,z,w) coefficients
template <class T> inline void quat_xyzw_poly_R(const T& x, const T& y, const T& z, const T& w, T* R) {
  T tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  T twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0 - (txx + tyy);
}

// base-frame kinematics of the URDF tree: Rh/ph = placement of body i in the pelvis frame,
// vh = velocity (pelvis-frame coordinates, relative to world) of a point attached to body i
template <class T> struct PinKin {
  T R0[9];
  T Rh[H1_NB][9], ph[H1_NB][3], zh[H1_NB][3];
  T Om[H1_NB][3];  // angular velocity of body i, pelvis-frame coordinates
  T vo[H1_NB][3];  // velocity of body i's origin, pelvis-frame coordinates
};
template <class T> inline void pin_kin(const T* xp, PinKin<T>& K, bool with_vel) {
  quat_xyzw_poly_R(xp[3], xp[4], xp[5], xp[6], K.R0);
  for (int k = 0; k < 9; ++k) K.Rh[0][k] = T((k % 4 == 0) ? 1.0 : 0.0);
  for (int k = 0; k < 3; ++k) { K.ph[0][k] = T(0.0); K.zh[0][k] = T(0.0); }
  if (with_vel) for (int k = 0; k < 3; ++k) { K.vo[0][k] = xp[H1_NQ + k]; K.Om[0][k] = xp[H1_NQ + 3 + k]; }
  for (int i = 1; i < H1_NB; ++i) {
    const int p = H1_PARENT[i], a = H1_AXIS[i];
    T Rj[9]; joint_rot(i, xp[7 + i - 1], H1U_RFIX, Rj);
    mat3_mul(K.Rh[p], Rj, K.Rh[i]);
    T t[3]; mat3_vec(K.Rh[p], H1U_POS[i], t);
    for (int k = 0; k < 3; ++k) { K.ph[i][k] = K.ph[p][k] + t[k]; K.zh[i][k] = K.Rh[i][3 * k + a]; }
    if (with_vel) {
      T wx[3]; cross3(K.Om[p], t, wx);
      for (int k = 0; k < 3; ++k) { K.vo[i][k] = K.vo[p][k] + wx[k]; K.Om[i][k] = K.Om[p][k] + K.zh[i][k] * xp[H1_NQ + 6 + i - 1]; }
    }
  }
}
// whole-body CoM (URDF masses) and its velocity, world frame: data.com[0], data.vcom[0]
template <class T> inline void pin_com(const T* xp, const PinKin<T>& K, T* com, T* vcom) {
  T b[3] = {T(0.0), T(0.0), T(0.0)}, gm[3] = {T(0.0), T(0.0), T(0.0)};
  double M = 0.0;
  for (int i = 0; i < H1_NB; ++i) {
    T c[3]; mat3_vec(K.Rh[i], H1U_COM[i], c);
    for (int k = 0; k < 3; ++k) b[k] += H1U_MASS[i] * (K.ph[i][k] + c[k]);
    if (vcom) { T wc[3]; cross3(K.Om[i], c, wc); for (int k = 0; k < 3; ++k) gm[k] += H1U_MASS[i] * (K.vo[i][k] + wc[k]); }
    M += H1U_MASS[i];
  }
  for (int k = 0; k < 3; ++k) { b[k] = b[k] / M; gm[k] = gm[k] / M; }
  T rb[3]; mat3_vec(K.R0, b, rb);
  for (int k = 0; k < 3; ++k) com[k] = xp[k] + rb[k];
  if (vcom) mat3_vec(K.R0, gm, vcom);
}

// scalar costs exactly as symDerivatives builds them (derivatives.cpp:525-707)
template <class T> inline T sym_com_pos(const T* xp, const double* ref, double w) {
  PinKin<T> K; pin_kin(xp, K, false); T c[3]; pin_com(xp, K, c, (T*)nullptr);
  T e0 = c[0] - ref[0], e1 = c[1] - ref[1], e2 = c[2] - ref[2];
  return w * (e0 * e0 + e1 * e1 + e2 * e2);
}
template <class T> inline T sym_com_vel(const T* xp, const double* ref, double w) {
  PinKin<T> K; pin_kin(xp, K, true); T c[3], vc[3]; pin_com(xp, K, c, vc);
  T e0 = vc[0] - ref[0], e1 = vc[1] - ref[1], e2 = vc[2] - ref[2];
  return w * (e0 * e0 + e1 * e1 + e2 * e2);
}
template <class T> inline T sym_ee_pos(const T* xp, int body, const double* ref, double w) {
  PinKin<T> K; pin_kin(xp, K, false);
  T r[3]; mat3_vec(K.R0, K.ph[body], r);
  T e0 = xp[0] + r[0] - ref[0], e1 = xp[1] + r[1] - ref[1], e2 = xp[2] + r[2] - ref[2];
  return w * (e0 * e0 + e1 * e1 + e2 * e2);
}
template <class T> inline T sym_ee_vel(const T* xp, int body, const double* ref, double w) {
  PinKin<T> K; pin_kin(xp, K, true);
  T v[3]; mat3_vec(K.R0, K.vo[body], v);
  T e0 = v[0] - ref[0], e1 = v[1] - ref[1], e2 = v[2] - ref[2];
  return w * (e0 * e0 + e1 * e1 + e2 * e2);
}
template <class T> inline T sym_upright(const T* xp, double w) {
  // derivatives.cpp:646-666 reads slots 3..6 as (qw,qx,qy,qz) although x_p stores (qx,qy,qz,qw)
  T qw = xp[3], qx = xp[4], qy = xp[5], qz = xp[6];
  T rx = 2.0 * (qx * qz + qw * qy), ry = 2.0 * (qy * qz - qw * qx), rz = (1.0 - 2.0 * (qx * qx + qy * qy)) - 1.0;
  return 0.5 * w * (rx * rx + ry * ry + rz * rz);
}
template <class T> inline T sym_balance(const T* xp, const double* ps, double w) {
  PinKin<T> K; pin_kin(xp, K, true); T c[3], vc[3]; pin_com(xp, K, c, vc);
  T om = sqrt(c[2] / 9.81);
  T r0 = c[0] + vc[0] * om - ps[0], r1 = c[1] + vc[1] * om - ps[1];
  return 0.5 * w * (r0 * r0 + r1 * r1);
}

inline void to_pin_order(const double* x, double* xp) {  // derivatives.cpp:12-24
  for (int i = 0; i < H1_NX; ++i) xp[i] = x[i];
  xp[3] = x[4]; xp[4] = x[5]; xp[5] = x[6]; xp[6] = x[3];
}

#ifndef ORC_COUNTING   // (the op-counting build, opcount.cpp, re-reads this file with the scalar type replaced: closed forms only)
// exact gradient + Hessian of a templated scalar by forward-over-forward AD; ADDS into g/H
template <class F> inline void ad_grad_hess(const double* xp, F f, double* g, double* H) {
  typedef D1<H1_NX> In; typedef DD<In> Out;
  for (int k = 0; k < H1_NX; ++k) {
    Out xs[H1_NX];
    for (int i = 0; i < H1_NX; ++i) { xs[i].v = In::var(xp[i], i); xs[i].d = In(i == k ? 1.0 : 0.0); }
    Out r = f(xs);
    if (k == 0) for (int i = 0; i < H1_NX; ++i) g[i] += r.v.g[i];
    for (int l = 0; l < H1_NX; ++l) H[k * H1_NX + l] += r.d.g[l];
  }
}
#endif

// ---------------- closed forms ----------------
struct BaseKin {
  double xp[H1_NX];
  double R0[9], D[4][9], DD_[4][4][9];
  double Rh[H1_NB][9], ph[H1_NB][3], zh[H1_NB][3], Om[H1_NB][3];
};
struct PointSet {  // mass-normalised point set attached to the tree
  double beta[3], gamma[3], mfrac;
  double w[H1_NB][3], s[H1_NB][3];
  bool on[H1_NB];
};
inline void dR_dquat(int k, const double* q /*x,y,z,w*/, double* D) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double d[4][9] = {{0, y, z, y, -2 * x, -w, z, w, -2 * x},
                          {-2 * y, x, w, x, 0, z, -w, z, -2 * y},
                          {-2 * z, -w, x, w, -2 * z, y, x, y, 0},
                          {0, -z, y, z, 0, -x, -y, x, 0}};
  for (int i = 0; i < 9; ++i) D[i] = 2.0 * d[k][i];
}
inline void base_kin(const double* xp, BaseKin& B) {
  for (int i = 0; i < H1_NX; ++i) B.xp[i] = xp[i];
  PinKin<double> K; pin_kin(xp, K, true);
  for (int k = 0; k < 9; ++k) B.R0[k] = K.R0[k];
  for (int k = 0; k < 4; ++k) {
    dR_dquat(k, xp + 3, B.D[k]);
    for (int l = 0; l < 4; ++l) { double e[4] = {0, 0, 0, 0}; e[l] = 1.0; dR_dquat(k, e, B.DD_[k][l]); }
  }
  for (int i = 0; i < H1_NB; ++i) for (int k = 0; k < 3; ++k) { B.ph[i][k] = K.ph[i][k]; B.zh[i][k] = K.zh[i][k]; B.Om[i][k] = K.Om[i][k]; }
  for (int i = 0; i < H1_NB; ++i) for (int k = 0; k < 9; ++k) B.Rh[i][k] = K.Rh[i][k];
}
// mu[i]: mass of the point attached to body i at local offset c[i]; Mtot: normalisation
template <class MU, class CC> inline void point_set(const BaseKin& B, const MU* mu, const CC (*c)[3], double Mtot, PointSet& S) {
  double msub[H1_NB], hsub[H1_NB][3], sw[H1_NB][3];
  for (int i = 0; i < H1_NB; ++i) { msub[i] = 0; for (int k = 0; k < 3; ++k) { hsub[i][k] = 0; sw[i][k] = 0; } }
  double mall = 0;
  for (int i = H1_NB - 1; i >= 0; --i) {
    double ch[3]; mat3_vec(B.Rh[i], c[i], ch);
    msub[i] += mu[i] / Mtot; mall += mu[i] / Mtot;
    for (int k = 0; k < 3; ++k) hsub[i][k] += mu[i] / Mtot * (B.ph[i][k] + ch[k]);
    if (i > 0) { int p = H1_PARENT[i]; msub[p] += msub[i]; for (int k = 0; k < 3; ++k) hsub[p][k] += hsub[i][k]; }
  }
  S.mfrac = mall;
  for (int k = 0; k < 3; ++k) S.beta[k] = hsub[0][k];
  const double* vb = B.xp + H1_NQ; const double* wb = B.xp + H1_NQ + 3;
  double wxb[3]; cross3(wb, S.beta, wxb);
  for (int k = 0; k < 3; ++k) S.gamma[k] = S.mfrac * vb[k] + wxb[k];
  for (int i = 1; i < H1_NB; ++i) {
    S.on[i] = msub[i] > 0.0;
    double r[3]; for (int k = 0; k < 3; ++k) r[k] = hsub[i][k] - msub[i] * B.ph[i][k];
    cross3(B.zh[i], r, S.w[i]);
    const double qd = B.xp[H1_NQ + 6 + i - 1];
    for (int k = 0; k < 3; ++k) S.gamma[k] += qd * S.w[i][k];
  }
  S.on[0] = true;
  // s_j = sum over strict descendants k of qd_k w_k
  for (int i = H1_NB - 1; i >= 1; --i) {
    const double qd = B.xp[H1_NQ + 6 + i - 1];
    for (int k = 0; k < 3; ++k) S.s[i][k] = sw[i][k];
    int p = H1_PARENT[i];
    for (int k = 0; k < 3; ++k) sw[p][k] += sw[i][k] + qd * S.w[i][k];
  }
}
inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// Jacobian of c = mfrac*p + R0*beta w.r.t. x_p (3 x 51, only q columns non-zero)
inline void jac_pos(const BaseKin& B, const PointSet& S, double (*J)[H1_NX]) {
  for (int r = 0; r < 3; ++r) for (int c = 0; c < H1_NX; ++c) J[r][c] = 0.0;
  for (int r = 0; r < 3; ++r) J[r][r] = S.mfrac;
  for (int k = 0; k < 4; ++k) { double t[3]; mat3_vec(B.D[k], S.beta, t); for (int r = 0; r < 3; ++r) J[r][3 + k] = t[r]; }
  for (int j = 1; j < H1_NB; ++j) if (S.on[j]) { double t[3]; mat3_vec(B.R0, S.w[j], t); for (int r = 0; r < 3; ++r) J[r][7 + j - 1] = t[r]; }
}
// d gamma / d theta_k
inline void dgamma_dtheta(const BaseKin& B, const PointSet& S, int k, double* out) {
  double a[3], b[3]; cross3(B.Om[k], S.w[k], a); cross3(B.zh[k], S.s[k], b);
  for (int r = 0; r < 3; ++r) out[r] = a[r] + b[r];
}
// d gamma / d v (3 x 25)
inline void jac_gamma_v(const BaseKin& B, const PointSet& S, double (*Jv)[H1_NV]) {
  for (int r = 0; r < 3; ++r) for (int c = 0; c < H1_NV; ++c) Jv[r][c] = 0.0;
  for (int r = 0; r < 3; ++r) Jv[r][r] = S.mfrac;
  const double* b = S.beta;
  const double mbx[9] = {0, b[2], -b[1], -b[2], 0, b[0], b[1], -b[0], 0};  // -[beta]x
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Jv[r][3 + c] = mbx[3 * r + c];
  for (int j = 1; j < H1_NB; ++j) if (S.on[j]) for (int r = 0; r < 3; ++r) Jv[r][6 + j - 1] = S.w[j][r];
}
// Jacobian of cdot = R0*gamma w.r.t. x_p (3 x 51)
inline void jac_vel(const BaseKin& B, const PointSet& S, double (*J)[H1_NX]) {
  for (int r = 0; r < 3; ++r) for (int c = 0; c < H1_NX; ++c) J[r][c] = 0.0;
  for (int k = 0; k < 4; ++k) { double t[3]; mat3_vec(B.D[k], S.gamma, t); for (int r = 0; r < 3; ++r) J[r][3 + k] = t[r]; }
  for (int j = 1; j < H1_NB; ++j) if (S.on[j]) { double d[3], t[3]; dgamma_dtheta(B, S, j, d); mat3_vec(B.R0, d, t); for (int r = 0; r < 3; ++r) J[r][7 + j - 1] = t[r]; }
  double Jv[3][H1_NV]; jac_gamma_v(B, S, Jv);
  for (int c = 0; c < H1_NV; ++c) { double col[3] = {Jv[0][c], Jv[1][c], Jv[2][c]}, t[3]; mat3_vec(B.R0, col, t); for (int r = 0; r < 3; ++r) J[r][H1_NQ + c] = t[r]; }
}
inline bool related(int a, int b, int& lo, int& hi) {  // bodies >=1; lo = ancestor-or-self
  if (H1_ANC[a - 1][b - 1]) { lo = a; hi = b; return true; }
  if (H1_ANC[b - 1][a - 1]) { lo = b; hi = a; return true; }
  return false;
}
// adds scale * Hessian of phi1 = mu^T (mfrac p + R0 beta) into H (51x51)
inline void hess_pos(const BaseKin& B, const PointSet& S, const double* mu, double scale, double* H) {
  double mt[3]; mat3T_vec(B.R0, mu, mt);
  double Dm[4][3]; for (int k = 0; k < 4; ++k) mat3T_vec(B.D[k], mu, Dm[k]);
  for (int k = 0; k < 4; ++k) for (int l = 0; l < 4; ++l) { double t[3]; mat3_vec(B.DD_[k][l], S.beta, t); H[(3 + k) * H1_NX + 3 + l] += scale * dot3(mu, t); }
  for (int k = 0; k < 4; ++k) for (int j = 1; j < H1_NB; ++j) if (S.on[j]) {
    double v = scale * dot3(Dm[k], S.w[j]);
    H[(3 + k) * H1_NX + 7 + j - 1] += v; H[(7 + j - 1) * H1_NX + 3 + k] += v;
  }
  for (int a = 1; a < H1_NB; ++a) for (int b = 1; b < H1_NB; ++b) {
    int lo, hi; if (!S.on[a] || !S.on[b] || !related(a, b, lo, hi)) continue;
    double t[3]; cross3(B.zh[lo], S.w[hi], t);
    H[(7 + a - 1) * H1_NX + 7 + b - 1] += scale * dot3(mt, t);
  }
}
// adds scale * Hessian of phi2 = nu^T R0 gamma into H (51x51)
inline void hess_vel(const BaseKin& B, const PointSet& S, const double* nu, double scale, double* H) {
  double nt[3]; mat3T_vec(B.R0, nu, nt);
  double Dn[4][3]; for (int k = 0; k < 4; ++k) mat3T_vec(B.D[k], nu, Dn[k]);
  double Jv[3][H1_NV]; jac_gamma_v(B, S, Jv);
  auto add = [&](int r, int c, double v) { H[r * H1_NX + c] += scale * v; if (r != c) H[c * H1_NX + r] += scale * v; };
  for (int k = 0; k < 4; ++k) for (int l = 0; l < 4; ++l) { double t[3]; mat3_vec(B.DD_[k][l], S.gamma, t); H[(3 + k) * H1_NX + 3 + l] += scale * dot3(nu, t); }
  for (int k = 0; k < 4; ++k) {
    for (int j = 1; j < H1_NB; ++j) if (S.on[j]) { double d[3]; dgamma_dtheta(B, S, j, d); add(3 + k, 7 + j - 1, dot3(Dn[k], d)); }
    for (int c = 0; c < H1_NV; ++c) { double col[3] = {Jv[0][c], Jv[1][c], Jv[2][c]}; add(3 + k, H1_NQ + c, dot3(Dn[k], col)); }
  }
  for (int a = 1; a < H1_NB; ++a) if (S.on[a]) {
    // theta_a - omega_b : w_a x ntilde
    double t[3]; cross3(S.w[a], nt, t);
    for (int c = 0; c < 3; ++c) add(7 + a - 1, H1_NQ + 3 + c, t[c]);
    for (int b = 1; b < H1_NB; ++b) if (S.on[b]) {
      int lo, hi; if (!related(a, b, lo, hi)) continue;
      double zw[3]; cross3(B.zh[lo], S.w[hi], zw);
      // theta_a - thetadot_b (every ordered pair once; add() mirrors it)
      add(7 + a - 1, H1_NQ + 6 + b - 1, dot3(nt, zw));
      if (a > b) continue;  // theta-theta: once per unordered pair (a<=b), add() mirrors it
      double t1[3], dO[3], t2a[3], t2[3], t3a[3], t3[3];
      cross3(B.Om[lo], zw, t1);
      for (int r = 0; r < 3; ++r) dO[r] = B.Om[hi][r] - B.Om[lo][r];
      cross3(dO, S.w[hi], t2a); cross3(B.zh[lo], t2a, t2);
      cross3(B.zh[hi], S.s[hi], t3a); cross3(B.zh[lo], t3a, t3);
      double v = nt[0] * (t1[0] + t2[0] + t3[0]) + nt[1] * (t1[1] + t2[1] + t3[1]) + nt[2] * (t1[2] + t2[2] + t3[2]);
      add(7 + a - 1, 7 + b - 1, v);
    }
  }
}

struct TermScratch { BaseKin B; PointSet com, ee[2]; bool ready = false; };
inline void prepare_terms(const double* x, TermScratch& Z) {
  double xp[H1_NX]; to_pin_order(x, xp);
  base_kin(xp, Z.B);
  double zero3[H1_NB][3] = {{0}};
  double mtot = 0.0; for (int i = 0; i < H1_NB; ++i) mtot += H1U_MASS[i];
  point_set(Z.B, H1U_MASS, H1U_COM, mtot, Z.com);
  for (int e = 0; e < 2; ++e) {
    double mu[H1_NB] = {0}; mu[e == 0 ? H1_EE_LEFT : H1_EE_RIGHT] = 1.0;
    point_set(Z.B, mu, zero3, 1.0, Z.ee[e]);
  }
  Z.ready = true;
}
// w*||c - ref||^2 (no 1/2): adds gradient and Hessian
inline void add_pos_term(const BaseKin& B, const PointSet& S, const double* ref, double w, double* g, double* H) {
  double J[3][H1_NX]; jac_pos(B, S, J);
  double rb[3]; mat3_vec(B.R0, S.beta, rb);
  double e[3]; for (int k = 0; k < 3; ++k) e[k] = S.mfrac * B.xp[k] + rb[k] - ref[k];
  for (int i = 0; i < H1_NQ; ++i) g[i] += 2.0 * w * (J[0][i] * e[0] + J[1][i] * e[1] + J[2][i] * e[2]);
  for (int i = 0; i < H1_NQ; ++i) for (int j = 0; j < H1_NQ; ++j) H[i * H1_NX + j] += 2.0 * w * (J[0][i] * J[0][j] + J[1][i] * J[1][j] + J[2][i] * J[2][j]);
  hess_pos(B, S, e, 2.0 * w, H);
}
inline void add_vel_term(const BaseKin& B, const PointSet& S, const double* ref, double w, double* g, double* H) {
  double J[3][H1_NX]; jac_vel(B, S, J);
  double v[3]; mat3_vec(B.R0, S.gamma, v);
  double e[3]; for (int k = 0; k < 3; ++k) e[k] = v[k] - ref[k];
  for (int i = 0; i < H1_NX; ++i) g[i] += 2.0 * w * (J[0][i] * e[0] + J[1][i] * e[1] + J[2][i] * e[2]);
  for (int i = 0; i < H1_NX; ++i) for (int j = 0; j < H1_NX; ++j) H[i * H1_NX + j] += 2.0 * w * (J[0][i] * J[0][j] + J[1][i] * J[1][j] + J[2][i] * J[2][j]);
  hess_vel(B, S, e, 2.0 * w, H);
}
inline void add_upright_term(const double* xp, double w, double* g, double* H) {
  const double a = xp[3], b = xp[4], c = xp[5], d = xp[6];
  const double r[3] = {2.0 * (b * d + a * c), 2.0 * (c * d - a * b), -2.0 * (b * b + c * c)};
  const double J[3][4] = {{2 * c, 2 * d, 2 * a, 2 * b}, {-2 * b, -2 * a, 2 * d, 2 * c}, {0, -4 * b, -4 * c, 0}};
  for (int i = 0; i < 4; ++i) g[3 + i] += w * (J[0][i] * r[0] + J[1][i] * r[1] + J[2][i] * r[2]);
  double Hs[4][4] = {{0}};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Hs[i][j] = J[0][i] * J[0][j] + J[1][i] * J[1][j] + J[2][i] * J[2][j];
  Hs[0][2] += 2 * r[0]; Hs[2][0] += 2 * r[0]; Hs[1][3] += 2 * r[0]; Hs[3][1] += 2 * r[0];
  Hs[2][3] += 2 * r[1]; Hs[3][2] += 2 * r[1]; Hs[0][1] += -2 * r[1]; Hs[1][0] += -2 * r[1];
  Hs[1][1] += -4 * r[2]; Hs[2][2] += -4 * r[2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) H[(3 + i) * H1_NX + 3 + j] += w * Hs[i][j];
}
inline void add_balance_term(const BaseKin& B, const PointSet& S, const double* ps, double w, double* g, double* H) {
  double Jc[3][H1_NX], Jv[3][H1_NX]; jac_pos(B, S, Jc); jac_vel(B, S, Jv);
  double rb[3], vc[3]; mat3_vec(B.R0, S.beta, rb); mat3_vec(B.R0, S.gamma, vc);
  double com[3]; for (int k = 0; k < 3; ++k) com[k] = S.mfrac * B.xp[k] + rb[k];
  const double gg = 9.81;
  const double om = std::sqrt(com[2] / gg), om1 = 1.0 / (2.0 * gg * om), om2 = -1.0 / (4.0 * gg * gg * om * om * om);
  double r[2] = {com[0] + vc[0] * om - ps[0], com[1] + vc[1] * om - ps[1]};
  double Jr[2][H1_NX];
  for (int i = 0; i < 2; ++i) for (int c = 0; c < H1_NX; ++c) Jr[i][c] = Jc[i][c] + om * Jv[i][c] + vc[i] * om1 * Jc[2][c];
  for (int c = 0; c < H1_NX; ++c) g[c] += w * (Jr[0][c] * r[0] + Jr[1][c] * r[1]);
  const double rv = r[0] * vc[0] + r[1] * vc[1];
  for (int a = 0; a < H1_NX; ++a) for (int b = 0; b < H1_NX; ++b) {
    double v = Jr[0][a] * Jr[0][b] + Jr[1][a] * Jr[1][b];
    for (int i = 0; i < 2; ++i) v += om1 * r[i] * (Jv[i][a] * Jc[2][b] + Jc[2][a] * Jv[i][b]);
    v += rv * om2 * Jc[2][a] * Jc[2][b];
    H[a * H1_NX + b] += w * v;
  }
  double mu[3] = {r[0], r[1], om1 * rv}, nu[3] = {om * r[0], om * r[1], 0.0};
  hess_pos(B, S, mu, w, H);
  hess_vel(B, S, nu, w, H);
}

enum QuadMode { QUAD_CLOSED = 0, QUAD_AD = 1 };

// computeCostQuadratics for one knot t (ilqr.cpp:133-244). lxx dense 51x51 row-major, luu diagonal (19).
inline void cost_quadratics_knot(const Problem& P, int t, const double* x, const double* u, double* lx, double* lu,
                                 double* lxx, double* luu_diag, QuadMode mode) {
  const int N = P.N; const bool term = (t == N);
  const double* Qd = term ? P.Qf : P.Q;
  for (int i = 0; i < H1_NX * H1_NX; ++i) lxx[i] = 0.0;
  for (int i = 0; i < H1_NX; ++i) { lx[i] = Qd[i] * (x[i] - P.x_ref[t * H1_NX + i]); lxx[i * H1_NX + i] = Qd[i]; }
  if (!term) for (int i = 0; i < H1_NU; ++i) { lu[i] = P.R[i] * (u[i] - P.u_ref[t * H1_NU + i]); luu_diag[i] = P.R[i]; }
  double xp[H1_NX]; to_pin_order(x, xp);
  double zero3[3] = {0, 0, 0};
  double ps[2]; const bool has_support = support_point(P, t, ps);
  if (mode == QUAD_CLOSED) {
    TermScratch Z; prepare_terms(x, Z);
    if (P.w_com > 0.0) add_pos_term(Z.B, Z.com, &P.com_ref[t * 3], P.w_com, lx, lxx);
    if (!term && P.w_com_vel > 0.0) add_vel_term(Z.B, Z.com, &P.com_vel_ref[t * 3], P.w_com_vel, lx, lxx);
    if (P.w_ee_pos > 0.0) for (int e = 0; e < 2; ++e) if (P.stance[2 * t + e] != 1) add_pos_term(Z.B, Z.ee[e], &P.ee_ref[(t * 2 + e) * 3], P.w_ee_pos, lx, lxx);
    if (P.w_ee_vel > 0.0) for (int e = 0; e < 2; ++e) if (P.stance[2 * t + e] == 1) add_vel_term(Z.B, Z.ee[e], zero3, P.w_ee_vel, lx, lxx);
    if (P.w_upright > 0.0) add_upright_term(xp, P.w_upright, lx, lxx);
    if (P.w_balance > 0.0 && has_support) add_balance_term(Z.B, Z.com, ps, P.w_balance, lx, lxx);
  }
#ifndef ORC_COUNTING
  else {
    if (P.w_com > 0.0) { const double* ref = &P.com_ref[t * 3]; double w = P.w_com; ad_grad_hess(xp, [&](const DD<D1<H1_NX>>* z) { return sym_com_pos(z, ref, w); }, lx, lxx); }
    if (!term && P.w_com_vel > 0.0) { const double* ref = &P.com_vel_ref[t * 3]; double w = P.w_com_vel; ad_grad_hess(xp, [&](const DD<D1<H1_NX>>* z) { return sym_com_vel(z, ref, w); }, lx, lxx); }
    if (P.w_ee_pos > 0.0) for (int e = 0; e < 2; ++e) if (P.stance[2 * t + e] != 1) { const double* ref = &P.ee_ref[(t * 2 + e) * 3]; double w = P.w_ee_pos; int body = e == 0 ? H1_EE_LEFT : H1_EE_RIGHT; ad_grad_hess(xp, [&](const DD<D1<H1_NX>>* z) { return sym_ee_pos(z, body, ref, w); }, lx, lxx); }
    if (P.w_ee_vel > 0.0) for (int e = 0; e < 2; ++e) if (P.stance[2 * t + e] == 1) { double w = P.w_ee_vel; int body = e == 0 ? H1_EE_LEFT : H1_EE_RIGHT; ad_grad_hess(xp, [&](const DD<D1<H1_NX>>* z) { return sym_ee_vel(z, body, zero3, w); }, lx, lxx); }
    if (P.w_upright > 0.0) {
      double w = P.w_upright; double g[H1_NX] = {0}; std::vector<double> Hh(H1_NX * H1_NX, 0.0);
      ad_grad_hess(xp, [&](const DD<D1<H1_NX>>* z) { return sym_upright(z, w); }, g, Hh.data());
      for (int i = 0; i < H1_NX; ++i) { lx[i] += g[i]; for (int j = 0; j < H1_NX; ++j) lxx[i * H1_NX + j] += 0.5 * (Hh[i * H1_NX + j] + Hh[j * H1_NX + i]); }  // derivatives.cpp:521
    }
    if (P.w_balance > 0.0 && has_support) {
      double w = P.w_balance; double g[H1_NX] = {0}; std::vector<double> Hh(H1_NX * H1_NX, 0.0);
      ad_grad_hess(xp, [&](const DD<D1<H1_NX>>* z) { return sym_balance(z, ps, w); }, g, Hh.data());
      for (int i = 0; i < H1_NX; ++i) { lx[i] += g[i]; for (int j = 0; j < H1_NX; ++j) lxx[i * H1_NX + j] += 0.5 * (Hh[i * H1_NX + j] + Hh[j * H1_NX + i]); }  // derivatives.cpp:796
    }
  }
#endif
  double zero_u[H1_NU] = {0}, dummy_lu[H1_NU] = {0}, dummy_luu[H1_NU] = {0};
  if (!term) constraint_derivs(P, x, u, lx, lu, lxx, luu_diag);
  else constraint_derivs(P, x, zero_u, lx, dummy_lu, lxx, dummy_luu);
}

}  // namespace orc
