// TEST INFRASTRUCTURE (oracle) -- CPU restatement of the reference's dynamics path; not shipped.
//
// Restates RobotUtils::rolloutOneStep (/root/reference/src/common/robot_utils.cpp:106-117), i.e. one
// MuJoCo mj_step of the Unitree H1 (h1.xml) in the constraint-free regime (no contact, no joint-limit
// constraint) with MuJoCo's semantics:
//   * qpos = [p(3), quat wxyz(4), hinge(19)], qvel = [v_lin WORLD(3), omega BODY(3), hinge rate(19)]
//   * quaternion normalised at the start of kinematics
//   * ctrl clamped to ctrlrange, motor gear 1, passive hinge damping -d*qvel
//   * semi-implicit Euler with implicit joint damping: (M + arm + h*D) qacc = tau - D v - bias
//   * qvel += h*qacc; p += h*v_lin; quat <- normalize(quat (x) exp(h*omega)); hinge += h*rate
// MuJoCo itself is not in /root/reference and not installed here => "parity unpinned" for this part;
// it is pinned instead by physics identities and an independent Kane-method residual (tests/).
// Everything is templated on the scalar so the same restatement yields exact Jacobians through
// forward-mode AD (ad.hpp) -- the analytic counterpart of linearizeDynamicsFD (robot_utils.cpp:120-160),
// which is also restated verbatim as a forward finite difference in ilqr_oracle.cpp.
#pragma once
#include <cmath>

#include "ad.hpp"
#include "h1_model_data.h"

namespace orc {
using std::cos;
using std::sin;
using std::sqrt;

struct DynParams {
  double h;     // timestep
  double g[3];  // world gravity vector
  int contact = 0;       // 0: constraint-free step; 1: rigid stance constraints on the scheduled feet;
                         // 2: the same, unilateral: a stance foot the floor would have to pull on is released
                         // 3: unilateral + Coulomb limit: a foot whose force leaves the cone |f_t| <= mu f_n slides (its two
                         //    tangential translation rows are dropped, rotation + normal rows stay; solved again, once)
                         // 4: as 3, and the sliding foot keeps kinetic friction mu lambda_n along the direction the sticking force had
  double mu = 1.0;       // sliding friction coefficient of mode 3 (MuJoCo's default geom friction; the H1 model file sets none)
  double soft = 1e-5;    // diagonal softness of the stance constraint (1 / kg), keeps J Minv J^T invertible with straight knees
  int limits = 0;        // 1: joint-limit rows (SURVEY Appendix C #7, h1.xml jnt_range enforced by mj_step): a hinge past its range that the
                         //    step would still move outward is stopped (h1_step)
  double lim_k = 0.0;    // restoring stiffness of those rows (1 / s^2): the row prescribes qacc_i = -v_i / h - lim_k r_i, r_i the violation
                         //    (MuJoCo's solref reference acceleration in its hard limit; 0: the pure stop of round 5)
};

// ---------- small dense helpers (row-major) ----------
template <class T> inline void mat3_mul(const T* A, const T* B, T* C) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { T s = A[3 * i] * B[j]; s += A[3 * i + 1] * B[3 + j]; s += A[3 * i + 2] * B[6 + j]; C[3 * i + j] = s; }
}
template <class T, class U> inline void mat3_vec(const T* A, const U* x, T* y) { for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2]; }
template <class T, class U> inline void mat3T_vec(const T* A, const U* x, T* y) { for (int i = 0; i < 3; ++i) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2]; }
template <class T, class U> inline void cross3(const T* a, const U* b, T* c) { T c0 = a[1] * b[2] - a[2] * b[1]; T c1 = a[2] * b[0] - a[0] * b[2]; T c2 = a[0] * b[1] - a[1] * b[0]; c[0] = c0; c[1] = c1; c[2] = c2; }

// unit quaternion (w,x,y,z) -> rotation matrix (body->world)
template <class T> inline void quat_wxyz_to_R(const T& w, const T& x, const T& y, const T& z, T* R) {
  R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z); R[2] = 2.0 * (x * z + w * y);
  R[3] = 2.0 * (x * y + w * z); R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
  R[6] = 2.0 * (x * z - w * y); R[7] = 2.0 * (y * z + w * x); R[8] = 1.0 - 2.0 * (x * x + y * y);
}

// rotation about principal axis a by angle th
template <class T> inline void rot_axis(int a, const T& th, T* R) {
  T c = cos(th), s = sin(th);
  for (int i = 0; i < 9; ++i) R[i] = T(0.0);
  if (a == 0) { R[0] = T(1.0); R[4] = c; R[5] = -s; R[7] = s; R[8] = c; }
  else if (a == 1) { R[4] = T(1.0); R[0] = c; R[2] = s; R[6] = -s; R[8] = c; }
  else { R[8] = T(1.0); R[0] = c; R[1] = -s; R[3] = s; R[4] = c; }
}

// joint placement of body i (>=1) in its parent: Rj = Rfix * Rot(axis, theta) (child->parent coords)
template <class T> inline void joint_rot(int i, const T& th, const double (*rfix)[3][3], T* Rj) {
  T Ra[9]; rot_axis(H1_AXIS[i], th, Ra);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { T s = rfix[i][r][0] * Ra[c]; s += rfix[i][r][1] * Ra[3 + c]; s += rfix[i][r][2] * Ra[6 + c]; Rj[3 * r + c] = s; }
}

// ---------- spatial algebra, 6-vectors ordered [angular; linear] ----------
// motion transform parent->child coordinates; E = Rj^T, r = child origin in parent coordinates
template <class T> inline void xf_motion(const T* Rj, const double* r, const T* vp, T* vc) {
  T t[3]; cross3(vp, r, t);  // omega_p x r
  T lin[3] = {vp[3] + t[0], vp[4] + t[1], vp[5] + t[2]};
  mat3T_vec(Rj, vp, vc); mat3T_vec(Rj, lin, vc + 3);
}
// force transform child->parent coordinates, accumulating
template <class T> inline void xf_force_acc(const T* Rj, const double* r, const T* fc, T* fp) {
  T n[3], f[3]; mat3_vec(Rj, fc, n); mat3_vec(Rj, fc + 3, f);
  T rf[3]; T rr[3] = {T(r[0]), T(r[1]), T(r[2])}; cross3(rr, f, rf);
  for (int k = 0; k < 3; ++k) { fp[k] += n[k] + rf[k]; fp[3 + k] += f[k]; }
}
// dense Plucker motion matrix X (6x6) parent->child
template <class T> inline void plucker(const T* Rj, const double* r, T* X) {
  T E[9]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) E[3 * i + j] = Rj[3 * j + i];
  double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
  for (int i = 0; i < 36; ++i) X[i] = T(0.0);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    X[6 * i + j] = E[3 * i + j]; X[6 * (i + 3) + (j + 3)] = E[3 * i + j];
    T s = E[3 * i] * rx[j]; s += E[3 * i + 1] * rx[3 + j]; s += E[3 * i + 2] * rx[6 + j];
    X[6 * (i + 3) + j] = -s;
  }
}
template <class T> inline void crm(const T* v, const T* m, T* out) {  // v x m (motion)
  T a[3], b[3], c[3]; cross3(v, m, a); cross3(v + 3, m, b); cross3(v, m + 3, c);
  for (int k = 0; k < 3; ++k) { out[k] = a[k]; out[3 + k] = b[k] + c[k]; }
}
template <class T> inline void crf(const T* v, const T* f, T* out) {  // v x* f (force)
  T a[3], b[3], c[3]; cross3(v, f, a); cross3(v + 3, f + 3, b); cross3(v, f + 3, c);
  for (int k = 0; k < 3; ++k) { out[k] = a[k] + b[k]; out[3 + k] = c[k]; }
}
// spatial inertia about the body origin from (mass, com, inertia about com)
inline void spatial_inertia(double m, const double* c, const double (*Ic)[3], double* I) {
  double cx[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
  for (int i = 0; i < 36; ++i) I[i] = 0.0;
  double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    I[6 * i + j] = Ic[i][j] + m * ((i == j ? cc : 0.0) - c[i] * c[j]);
    I[6 * i + (3 + j)] = m * cx[3 * i + j];
    I[6 * (3 + i) + j] = m * cx[3 * j + i];
    I[6 * (3 + i) + (3 + j)] = (i == j) ? m : 0.0;
  }
}
template <class T, class U> inline void mat6_vec(const U* A, const T* x, T* y) {
  for (int i = 0; i < 6; ++i) { T s = x[0] * A[6 * i]; for (int j = 1; j < 6; ++j) s += x[j] * A[6 * i + j]; y[i] = s; }
}
// solve SPD 6x6 system A y = b (LDL^T without pivoting)
template <class T> inline void solve6_spd(const T* A, const T* b, T* y) {
  T L[36], d[6];
  for (int j = 0; j < 6; ++j) {
    T s = A[6 * j + j]; for (int k = 0; k < j; ++k) s -= L[6 * j + k] * L[6 * j + k] * d[k]; d[j] = s;
    for (int i = j + 1; i < 6; ++i) { T t = A[6 * i + j]; for (int k = 0; k < j; ++k) t -= L[6 * i + k] * L[6 * j + k] * d[k]; L[6 * i + j] = t / d[j]; }
  }
  T z[6];
  for (int i = 0; i < 6; ++i) { T s = b[i]; for (int k = 0; k < i; ++k) s -= L[6 * i + k] * z[k]; z[i] = s; }
  for (int i = 0; i < 6; ++i) z[i] = z[i] / d[i];
  for (int i = 5; i >= 0; --i) { T s = z[i]; for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * y[k]; y[i] = s; }
}

// ---------- forward dynamics (articulated-body algorithm) in MuJoCo coordinates ----------
// Input : quat_hat (unit, wxyz), theta[19], v (MuJoCo qvel, 25), tau[19] (hinge generalized forces),
//         arm_eff (effective armature on hinges), gravity.
// Output: qacc (MuJoCo convention, 25)
template <class T>
inline void forward_dynamics_mj(const T* quat_hat, const T* theta, const T* v, const T* tau, double arm_eff,
                                const double* grav, T* qacc, const int* lock = nullptr, const T* lockacc = nullptr) {
  // lock[i - 1] != 0: hinge i is acceleration-prescribed (Featherstone's hybrid dynamics): qacc_i = lockacc[i - 1], the body passes its
  // whole articulated inertia to the parent (no reduction by U U^T / D), its bias carries IA S qacc_i
  double Isp[H1_NB][36];
  for (int i = 0; i < H1_NB; ++i) spatial_inertia(H1_MASS[i], H1_COM[i], H1_INERTIA[i], Isp[i]);

  T R0[9]; quat_wxyz_to_R(quat_hat[0], quat_hat[1], quat_hat[2], quat_hat[3], R0);
  T Rj[H1_NB][9], vel[H1_NB][6], cb[H1_NB][6], IA[H1_NB][36], pA[H1_NB][6];
  // base spatial velocity in body coordinates: (omega_body, R0^T v_lin_world)
  for (int k = 0; k < 3; ++k) vel[0][k] = v[3 + k];
  mat3T_vec(R0, v, vel[0] + 3);
  for (int k = 0; k < 6; ++k) cb[0][k] = T(0.0);
  for (int i = 0; i < H1_NB; ++i) {
    if (i > 0) {
      joint_rot(i, theta[i - 1], H1_RFIX, Rj[i]);
      xf_motion(Rj[i], H1_POS[i], vel[H1_PARENT[i]], vel[i]);
      T vJ[6]; for (int k = 0; k < 6; ++k) vJ[k] = T(0.0);
      vJ[H1_AXIS[i]] = v[6 + i - 1];
      vel[i][H1_AXIS[i]] += v[6 + i - 1];
      crm(vel[i], vJ, cb[i]);
    }
    for (int k = 0; k < 36; ++k) IA[i][k] = T(Isp[i][k]);
    T Iv[6]; mat6_vec(Isp[i], vel[i], Iv);
    crf(vel[i], Iv, pA[i]);
  }
  T U[H1_NB][6], D[H1_NB], uu[H1_NB];
  for (int i = H1_NB - 1; i >= 1; --i) {
    const int a = H1_AXIS[i];
    for (int k = 0; k < 6; ++k) U[i][k] = IA[i][6 * k + a];
    D[i] = U[i][a] + arm_eff;
    uu[i] = tau[i - 1] - pA[i][a];
    T Ia[36], pa[6];
    const bool lk = lock && lock[i - 1];
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Ia[6 * r + c] = lk ? IA[i][6 * r + c] : IA[i][6 * r + c] - U[i][r] * U[i][c] / D[i];
    T Iac[6]; mat6_vec(Ia, cb[i], Iac);
    for (int k = 0; k < 6; ++k) pa[k] = pA[i][k] + Iac[k] + U[i][k] * (lk ? lockacc[i - 1] : uu[i] / D[i]);
    // propagate to parent: IA_p += X^T Ia X ; pA_p += X^T pa
    T X[36]; plucker(Rj[i], H1_POS[i], X);
    T tmp[36];
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { T s = Ia[6 * r] * X[c]; for (int k = 1; k < 6; ++k) s += Ia[6 * r + k] * X[6 * k + c]; tmp[6 * r + c] = s; }
    const int p = H1_PARENT[i];
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { T s = X[r] * tmp[c]; for (int k = 1; k < 6; ++k) s += X[6 * k + r] * tmp[6 * k + c]; IA[p][6 * r + c] += s; }
    xf_force_acc(Rj[i], H1_POS[i], pa, pA[p]);
  }
  // base: a0' = X0 * (0, -g) = (0, -R0^T g); coordinate acceleration nu_dot = -IA^-1 pA - a0'
  T a0p[6]; for (int k = 0; k < 3; ++k) a0p[k] = T(0.0);
  T mg[3] = {T(-grav[0]), T(-grav[1]), T(-grav[2])};
  mat3T_vec(R0, mg, a0p + 3);
  T acc[H1_NB][6];
  T rhs[6]; for (int k = 0; k < 6; ++k) rhs[k] = -pA[0][k];
  solve6_spd(IA[0], rhs, acc[0]);  // gravity-offset spatial acceleration of the base
  T nudot[6]; for (int k = 0; k < 6; ++k) nudot[k] = acc[0][k] - a0p[k];
  for (int i = 1; i < H1_NB; ++i) {
    const int a = H1_AXIS[i];
    T ap[6]; xf_motion(Rj[i], H1_POS[i], acc[H1_PARENT[i]], ap);
    for (int k = 0; k < 6; ++k) ap[k] += cb[i][k];
    T s = uu[i]; for (int k = 0; k < 6; ++k) s -= U[i][k] * ap[k];
    T qdd = (lock && lock[i - 1]) ? lockacc[i - 1] : s / D[i];
    for (int k = 0; k < 6; ++k) acc[i][k] = ap[k];
    acc[i][a] += qdd;
    qacc[6 + i - 1] = qdd;
  }
  // MuJoCo convention: linear = world acceleration of the base origin = R0 (vdot_O + omega x v_O)
  T wxv[3]; cross3(vel[0], vel[0] + 3, wxv);
  T lin[3] = {nudot[3] + wxv[0], nudot[4] + wxv[1], nudot[5] + wxv[2]};
  mat3_vec(R0, lin, qacc);
  for (int k = 0; k < 3; ++k) qacc[3 + k] = nudot[k];
}

// ---------- inverse dynamics (RNEA) in MuJoCo coordinates: tau = M(q) qacc + bias(q,v) ----------
// armature enters as arm * qacc on hinge rows. Used for qfrc_bias (gravity compensation,
// robot_utils.cpp:844-866) and for the ABA o RNEA = id identity test.
template <class T>
inline void inverse_dynamics_mj(const T* quat_hat, const T* theta, const T* v, const T* qacc, double arm,
                                const double* grav, T* tau) {
  double Isp[H1_NB][36];
  for (int i = 0; i < H1_NB; ++i) spatial_inertia(H1_MASS[i], H1_COM[i], H1_INERTIA[i], Isp[i]);
  T R0[9]; quat_wxyz_to_R(quat_hat[0], quat_hat[1], quat_hat[2], quat_hat[3], R0);
  T Rj[H1_NB][9], vel[H1_NB][6], acc[H1_NB][6], f[H1_NB][6];
  for (int k = 0; k < 3; ++k) vel[0][k] = v[3 + k];
  mat3T_vec(R0, v, vel[0] + 3);
  // nu_dot: angular = qacc_ang ; linear = R0^T qacc_lin - omega x v_O ; plus gravity offset (0, -R0^T g)
  T wxv[3]; cross3(vel[0], vel[0] + 3, wxv);
  T t3[3]; mat3T_vec(R0, qacc, t3);
  T mg[3] = {T(-grav[0]), T(-grav[1]), T(-grav[2])}; T g3[3]; mat3T_vec(R0, mg, g3);
  for (int k = 0; k < 3; ++k) { acc[0][k] = qacc[3 + k]; acc[0][3 + k] = t3[k] - wxv[k] + g3[k]; }
  for (int i = 0; i < H1_NB; ++i) {
    if (i > 0) {
      joint_rot(i, theta[i - 1], H1_RFIX, Rj[i]);
      const int a = H1_AXIS[i], p = H1_PARENT[i];
      xf_motion(Rj[i], H1_POS[i], vel[p], vel[i]);
      T vJ[6]; for (int k = 0; k < 6; ++k) vJ[k] = T(0.0);
      vJ[a] = v[6 + i - 1];
      vel[i][a] += v[6 + i - 1];
      T c[6]; crm(vel[i], vJ, c);
      xf_motion(Rj[i], H1_POS[i], acc[p], acc[i]);
      for (int k = 0; k < 6; ++k) acc[i][k] += c[k];
      acc[i][a] += qacc[6 + i - 1];
    }
    T Iv[6], Ia[6], vIv[6]; mat6_vec(Isp[i], vel[i], Iv); mat6_vec(Isp[i], acc[i], Ia); crf(vel[i], Iv, vIv);
    for (int k = 0; k < 6; ++k) f[i][k] = Ia[k] + vIv[k];
  }
  for (int i = H1_NB - 1; i >= 1; --i) {
    tau[6 + i - 1] = f[i][H1_AXIS[i]] + arm * qacc[6 + i - 1];
    xf_force_acc(Rj[i], H1_POS[i], f[i], f[H1_PARENT[i]]);
  }
  mat3_vec(R0, f[0] + 3, tau);  // force conjugate to world linear velocity
  for (int k = 0; k < 3; ++k) tau[3 + k] = f[0][k];
}


// ---------- forward dynamics with schedule-driven rigid stance constraints ----------
// The reference's plant is MuJoCo with soft floor contacts (robot_utils.cpp:106-117 -> mj_step); MuJoCo is not
// available here, so the contact row is restated as the limit the reference's standing / walking scenarios operate in:
// a foot the contact schedule marks as stance (RobotUtils::isStance, robot_utils.cpp:494-504) does not move.  The
// constraint is imposed at velocity level over one step, consistently with the semi-implicit Euler integrator:
//     v_f + h a_f = 0        (v_f, a_f: spatial velocity / acceleration of the ankle link, link coordinates)
// with a_f = a_f,free + C lambda, C = J Mhat^-1 J^T (Mhat = M + armature + h D), lambda = constraint wrench on the link.
// C is built column by column by propagating unit wrenches through the articulated-body quantities of the free
// solve (inward along the leg, pelvis solve, outward), i.e. without forming M or J; (C + soft I) lambda = rhs is solved
// by Cholesky and the wrench is propagated once more for the joint accelerations.  Bilateral on purpose (a scheduled
// stance foot sticks): unilateral / friction-cone effects are outside this row.
template <class T> inline bool chol_solve_inplace(T* A, T* b, int n) {  // A (n x n, row-major, SPD) b -> x; false if not PD
  for (int j = 0; j < n; ++j) {
    T d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(val(d) > 0.0)) return false;
    d = sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      T s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; ++i) { T s = b[i]; for (int k = 0; k < i; ++k) s -= A[i * n + k] * b[k]; b[i] = s / A[i * n + i]; }
  for (int i = n - 1; i >= 0; --i) { T s = b[i]; for (int k = i + 1; k < n; ++k) s -= A[k * n + i] * b[k]; b[i] = s / A[i * n + i]; }
  return true;
}
constexpr int H1_FOOT_BODY[2] = {5, 10};   // left / right ankle link (leaf of each leg chain)

template <class T>
inline void forward_dynamics_mj_stance(const T* quat_hat, const T* theta, const T* v, const T* tau, double arm_eff,
                                       const double* grav, double h, double soft, const int* stance, T* qacc, int contact_mode = 1,
                                       double mu = 1.0, const int* lock = nullptr, const T* lockacc = nullptr) {
  // (lock / lockacc: as forward_dynamics_mj; the stance rows are then solved on the system with those hinges prescribed -- a locked
  // hinge passes a unit wrench's share inward unreduced and does not accelerate in response)
  double Isp[H1_NB][36];
  for (int i = 0; i < H1_NB; ++i) spatial_inertia(H1_MASS[i], H1_COM[i], H1_INERTIA[i], Isp[i]);
  T R0[9]; quat_wxyz_to_R(quat_hat[0], quat_hat[1], quat_hat[2], quat_hat[3], R0);
  T Rj[H1_NB][9], vel[H1_NB][6], cb[H1_NB][6], IA[H1_NB][36], pA[H1_NB][6];
  for (int k = 0; k < 3; ++k) vel[0][k] = v[3 + k];
  mat3T_vec(R0, v, vel[0] + 3);
  for (int k = 0; k < 6; ++k) cb[0][k] = T(0.0);
  for (int i = 0; i < H1_NB; ++i) {
    if (i > 0) {
      joint_rot(i, theta[i - 1], H1_RFIX, Rj[i]);
      xf_motion(Rj[i], H1_POS[i], vel[H1_PARENT[i]], vel[i]);
      T vJ[6]; for (int k = 0; k < 6; ++k) vJ[k] = T(0.0);
      vJ[H1_AXIS[i]] = v[6 + i - 1];
      vel[i][H1_AXIS[i]] += v[6 + i - 1];
      crm(vel[i], vJ, cb[i]);
    }
    for (int k = 0; k < 36; ++k) IA[i][k] = T(Isp[i][k]);
    T Iv[6]; mat6_vec(Isp[i], vel[i], Iv);
    crf(vel[i], Iv, pA[i]);
  }
  T U[H1_NB][6], D[H1_NB], uu[H1_NB];
  for (int i = H1_NB - 1; i >= 1; --i) {
    const int a = H1_AXIS[i];
    for (int k = 0; k < 6; ++k) U[i][k] = IA[i][6 * k + a];
    D[i] = U[i][a] + arm_eff;
    uu[i] = tau[i - 1] - pA[i][a];
    T Ia[36], pa[6];
    const bool lk = lock && lock[i - 1];
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Ia[6 * r + c] = lk ? IA[i][6 * r + c] : IA[i][6 * r + c] - U[i][r] * U[i][c] / D[i];
    T Iac[6]; mat6_vec(Ia, cb[i], Iac);
    for (int k = 0; k < 6; ++k) pa[k] = pA[i][k] + Iac[k] + U[i][k] * (lk ? lockacc[i - 1] : uu[i] / D[i]);
    T X[36]; plucker(Rj[i], H1_POS[i], X);
    T tmp[36];
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { T s = Ia[6 * r] * X[c]; for (int k = 1; k < 6; ++k) s += Ia[6 * r + k] * X[6 * k + c]; tmp[6 * r + c] = s; }
    const int p = H1_PARENT[i];
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { T s = X[r] * tmp[c]; for (int k = 1; k < 6; ++k) s += X[6 * k + r] * tmp[6 * k + c]; IA[p][6 * r + c] += s; }
    xf_force_acc(Rj[i], H1_POS[i], pa, pA[p]);
  }
  T a0p[6]; for (int k = 0; k < 3; ++k) a0p[k] = T(0.0);
  T mg[3] = {T(-grav[0]), T(-grav[1]), T(-grav[2])};
  mat3T_vec(R0, mg, a0p + 3);
  T acc[H1_NB][6];
  T rhs0[6]; for (int k = 0; k < 6; ++k) rhs0[k] = -pA[0][k];
  solve6_spd(IA[0], rhs0, acc[0]);
  T qdd[H1_NB];
  for (int i = 1; i < H1_NB; ++i) {
    const int a = H1_AXIS[i];
    T ap[6]; xf_motion(Rj[i], H1_POS[i], acc[H1_PARENT[i]], ap);
    for (int k = 0; k < 6; ++k) ap[k] += cb[i][k];
    T s = uu[i]; for (int k = 0; k < 6; ++k) s -= U[i][k] * ap[k];
    qdd[i] = (lock && lock[i - 1]) ? lockacc[i - 1] : s / D[i];
    for (int k = 0; k < 6; ++k) acc[i][k] = ap[k];
    acc[i][a] += qdd[i];
  }
  // ---- stance constraints
  int nf = 0, fb[2];
  for (int f = 0; f < 2; ++f) if (stance[f] == 1) fb[nf++] = H1_FOOT_BODY[f];
  T da0[6]; for (int k = 0; k < 6; ++k) da0[k] = T(0.0);
  if (nf > 0) {
    // response to wrenches fext[g] (link coordinates, acting on stance foot g): accelerations of the stance feet,
    // and (full = true) of the pelvis and every hinge
    auto respond = [&](const T (*fext)[6], T (*da_feet)[6], bool full, T* dqdd, T* dbase) {
      T dpA[H1_NB][6], du[H1_NB], da[H1_NB][6];
      for (int i = 0; i < H1_NB; ++i) { du[i] = T(0.0); for (int k = 0; k < 6; ++k) dpA[i][k] = T(0.0); }
      for (int g = 0; g < nf; ++g) for (int k = 0; k < 6; ++k) dpA[fb[g]][k] -= fext[g][k];
      for (int i = 10; i >= 1; --i) {                       // the two leg chains
        const int a = H1_AXIS[i];
        du[i] = -dpA[i][a];
        const bool lk = lock && lock[i - 1];
        T dpa[6]; for (int k = 0; k < 6; ++k) dpa[k] = lk ? dpA[i][k] : dpA[i][k] + U[i][k] * (du[i] / D[i]);
        xf_force_acc(Rj[i], H1_POS[i], dpa, dpA[H1_PARENT[i]]);
      }
      T r6[6]; for (int k = 0; k < 6; ++k) r6[k] = -dpA[0][k];
      solve6_spd(IA[0], r6, da[0]);
      const int last = full ? H1_NB - 1 : 10;
      for (int i = 1; i <= last; ++i) {
        const int a = H1_AXIS[i];
        T ap[6]; xf_motion(Rj[i], H1_POS[i], da[H1_PARENT[i]], ap);
        T s = du[i]; for (int k = 0; k < 6; ++k) s -= U[i][k] * ap[k];
        const T q = (lock && lock[i - 1]) ? T(0.0) : s / D[i];
        for (int k = 0; k < 6; ++k) da[i][k] = ap[k];
        da[i][a] += q;
        if (full) dqdd[i] = q;
      }
      for (int g = 0; g < nf; ++g) for (int k = 0; k < 6; ++k) da_feet[g][k] = da[fb[g]][k];
      if (full) for (int k = 0; k < 6; ++k) dbase[k] = da[0][k];
    };
    // (C + soft I) lambda = b for the current stance set fb[0..nf)
    T lam[12];
    auto solve_set = [&]() {
      const int nc = 6 * nf;
      T C[144];
      for (int g = 0; g < nf; ++g)
        for (int c = 0; c < 6; ++c) {
          T fext[2][6], daf[2][6];
          for (int g2 = 0; g2 < 2; ++g2) for (int k = 0; k < 6; ++k) fext[g2][k] = T(0.0);
          fext[g][c] = T(1.0);
          respond(fext, daf, false, nullptr, nullptr);
          for (int g2 = 0; g2 < nf; ++g2) for (int k = 0; k < 6; ++k) C[(6 * g2 + k) * nc + 6 * g + c] = daf[g2][k];
        }
      for (int i = 0; i < nc; ++i) C[i * nc + i] += soft;
      // true (not gravity-offset) spatial acceleration of a leg body: acc_i - X_{i<-0} (0, R0^T(-g))
      for (int g = 0; g < nf; ++g) {
        T off[6]; for (int k = 0; k < 6; ++k) off[k] = a0p[k];
        const int first = fb[g] - 4;
        for (int i = first; i <= fb[g]; ++i) { T o2[6]; xf_motion(Rj[i], H1_POS[i], off, o2); for (int k = 0; k < 6; ++k) off[k] = o2[k]; }
        for (int k = 0; k < 6; ++k) lam[6 * g + k] = -vel[fb[g]][k] / h - (acc[fb[g]][k] - off[k]);
      }
      chol_solve_inplace(C, lam, nc);
    };
    solve_set();
    if (contact_mode >= 2) {
      // unilateral: the floor pushes, it does not pull.  Normal force on foot g = (world up axis in link coordinates) . (force part
      // of lambda_g); feet with a negative one are released and the remaining set is solved again (once).
      int keep[2], nk = 0;
      for (int g = 0; g < nf; ++g) {
        T zl[6] = {T(0.0), T(0.0), T(0.0), R0[6], R0[7], R0[8]};       // R0^T e_z
        for (int i = fb[g] - 4; i <= fb[g]; ++i) { T o2[6]; xf_motion(Rj[i], H1_POS[i], zl, o2); for (int k = 0; k < 6; ++k) zl[k] = o2[k]; }
        const T fz = zl[3] * lam[6 * g + 3] + zl[4] * lam[6 * g + 4] + zl[5] * lam[6 * g + 5];
        if (!(val(fz) < 0.0)) keep[nk++] = fb[g];
      }
      if (nk < nf) { nf = nk; for (int g = 0; g < nk; ++g) fb[g] = keep[g]; if (nf > 0) solve_set(); }
    }
    if (contact_mode >= 3 && nf > 0) {
      // Coulomb limit on the feet that are left: f_n = up . f, |f_t|^2 = |f|^2 - f_n^2 (f: force part of lambda_g, link coordinates).
      // A foot outside the cone slides: its constraint keeps the three rotation rows and the normal translation row,
      //     S_g = [I3 0; 0 up^T]  (4 x 6),   (S C S^T + soft I) lambda_s = S b,   lambda = S^T lambda_s
      // with C, b the rigid system of the current set (built again: solve_set factorises in place).
      // contact_mode 4: the sliding foot keeps kinetic friction, mu lambda_n along the unit direction t in which the sticking solution
      // pulled: the normal multiplier's force acts along up + mu t while its constraint row stays up^T -- F = [I3 0; 0 (up + mu t)^T],
      //     (S C F^T + soft I) lambda_s = S b,   lambda = F^T lambda_s     (unsymmetric: Gaussian elimination with partial pivoting)
      T up[2][3], tdir[2][3]; bool slide[2] = {false, false}; bool any = false;
      for (int g = 0; g < nf; ++g) {
        T zl[6] = {T(0.0), T(0.0), T(0.0), R0[6], R0[7], R0[8]};
        for (int i = fb[g] - 4; i <= fb[g]; ++i) { T o2[6]; xf_motion(Rj[i], H1_POS[i], zl, o2); for (int k = 0; k < 6; ++k) zl[k] = o2[k]; }
        for (int k = 0; k < 3; ++k) up[g][k] = zl[3 + k];
        const T* f = lam + 6 * g + 3;
        const T fn = up[g][0] * f[0] + up[g][1] * f[1] + up[g][2] * f[2];
        const T ft2 = f[0] * f[0] + f[1] * f[1] + f[2] * f[2] - fn * fn;
        if (val(ft2) > mu * mu * val(fn) * val(fn)) {
          slide[g] = true; any = true;
          const T nt = sqrt(ft2);
          for (int k = 0; k < 3; ++k) tdir[g][k] = (f[k] - fn * up[g][k]) / nt;
        }
      }
      if (any) {
        const int nc = 6 * nf;
        T C[144], b[12];
        for (int g = 0; g < nf; ++g)
          for (int c = 0; c < 6; ++c) {
            T fext[2][6], daf[2][6];
            for (int g2 = 0; g2 < 2; ++g2) for (int k = 0; k < 6; ++k) fext[g2][k] = T(0.0);
            fext[g][c] = T(1.0);
            respond(fext, daf, false, nullptr, nullptr);
            for (int g2 = 0; g2 < nf; ++g2) for (int k = 0; k < 6; ++k) C[(6 * g2 + k) * nc + 6 * g + c] = daf[g2][k];
          }
        for (int g = 0; g < nf; ++g) {
          T off[6]; for (int k = 0; k < 6; ++k) off[k] = a0p[k];
          for (int i = fb[g] - 4; i <= fb[g]; ++i) { T o2[6]; xf_motion(Rj[i], H1_POS[i], off, o2); for (int k = 0; k < 6; ++k) off[k] = o2[k]; }
          for (int k = 0; k < 6; ++k) b[6 * g + k] = -vel[fb[g]][k] / h - (acc[fb[g]][k] - off[k]);
        }
        T S[12][12]; int ns = 0;
        for (int g = 0; g < nf; ++g) {
          const int rows = slide[g] ? 4 : 6;
          for (int r = 0; r < rows; ++r, ++ns) {
            for (int c = 0; c < nc; ++c) S[ns][c] = T(0.0);
            if (slide[g] && r == 3) for (int k = 0; k < 3; ++k) S[ns][6 * g + 3 + k] = up[g][k];
            else S[ns][6 * g + r] = T(1.0);
          }
        }
        // F: the force map of the multipliers (= S, except the normal row of a sliding foot in mode 4)
        T F[12][12];
        for (int i = 0; i < ns; ++i) for (int c = 0; c < nc; ++c) F[i][c] = S[i][c];
        if (contact_mode == 4) {
          int r = 0;
          for (int g = 0; g < nf; ++g) {
            if (slide[g]) for (int k = 0; k < 3; ++k) F[r + 3][6 * g + 3 + k] = up[g][k] + mu * tdir[g][k];
            r += slide[g] ? 4 : 6;
          }
        }
        T CS[144], Cs[144], bs[12];
        for (int i = 0; i < nc; ++i) for (int j = 0; j < ns; ++j) { T a = T(0.0); for (int k = 0; k < nc; ++k) a += C[i * nc + k] * F[j][k]; CS[i * ns + j] = a; }
        for (int i = 0; i < ns; ++i) {
          for (int j = 0; j < ns; ++j) { T a = T(0.0); for (int k = 0; k < nc; ++k) a += S[i][k] * CS[k * ns + j]; Cs[i * ns + j] = a; }
          Cs[i * ns + i] += soft;
          T a = T(0.0); for (int k = 0; k < nc; ++k) a += S[i][k] * b[k];
          bs[i] = a;
        }
        if (contact_mode == 4) {
          for (int j = 0; j < ns; ++j) {                      // Gaussian elimination, partial pivoting
            int piv = j; double best = std::fabs(val(Cs[j * ns + j]));
            for (int i = j + 1; i < ns; ++i) if (std::fabs(val(Cs[i * ns + j])) > best) { best = std::fabs(val(Cs[i * ns + j])); piv = i; }
            if (piv != j) { for (int k = 0; k < ns; ++k) { T t_ = Cs[j * ns + k]; Cs[j * ns + k] = Cs[piv * ns + k]; Cs[piv * ns + k] = t_; } T t_ = bs[j]; bs[j] = bs[piv]; bs[piv] = t_; }
            for (int i = j + 1; i < ns; ++i) {
              const T m_ = Cs[i * ns + j] / Cs[j * ns + j];
              for (int k = j; k < ns; ++k) Cs[i * ns + k] -= m_ * Cs[j * ns + k];
              bs[i] -= m_ * bs[j];
            }
          }
          for (int i = ns - 1; i >= 0; --i) { T a = bs[i]; for (int k = i + 1; k < ns; ++k) a -= Cs[i * ns + k] * bs[k]; bs[i] = a / Cs[i * ns + i]; }
        } else chol_solve_inplace(Cs, bs, ns);
        for (int c = 0; c < nc; ++c) { T a = T(0.0); for (int i = 0; i < ns; ++i) a += F[i][c] * bs[i]; lam[c] = a; }
      }
    }
    if (nf > 0) {
      T fext[2][6], daf[2][6], dq[H1_NB];
      for (int g = 0; g < 2; ++g) for (int k = 0; k < 6; ++k) fext[g][k] = g < nf ? lam[6 * g + k] : T(0.0);
      respond(fext, daf, true, dq, da0);
      for (int i = 1; i < H1_NB; ++i) qdd[i] += dq[i];
    }
  }
  for (int i = 1; i < H1_NB; ++i) qacc[6 + i - 1] = qdd[i];
  T nudot[6]; for (int k = 0; k < 6; ++k) nudot[k] = acc[0][k] - a0p[k] + da0[k];
  T wxv[3]; cross3(vel[0], vel[0] + 3, wxv);
  T lin[3] = {nudot[3] + wxv[0], nudot[4] + wxv[1], nudot[5] + wxv[2]};
  mat3_vec(R0, lin, qacc);
  for (int k = 0; k < 3; ++k) qacc[3 + k] = nudot[k];
}

// smooth cos(a/2), sin(a/2)/a as functions of s = a^2 (AD-safe at a = 0)
template <class T> inline void half_angle_cs(const T& s, T& c, T& sn_over_a) {
  if (val(s) < 1e-6) {
    c = 1.0 - s / 8.0 + s * s / 384.0 - s * s * s / 46080.0;
    sn_over_a = 0.5 - s / 48.0 + s * s / 3840.0 - s * s * s / 645120.0;
  } else {
    T a = sqrt(s); c = cos(a * 0.5); sn_over_a = sin(a * 0.5) / a;
  }
}

// One dynamics step x_next = f(x, u): restates rolloutOneStep (robot_utils.cpp:106-117).
// `stance` (two flags, left / right, of the knot being stepped) is used when P.contact != 0.
template <class T>
inline void h1_step(const T* x, const T* u, const DynParams& P, T* xn, const int* stance = nullptr) {
  const double h = P.h;
  T qn = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  T qh[4] = {x[3] / qn, x[4] / qn, x[5] / qn, x[6] / qn};
  T tau[H1_NU];
  for (int i = 0; i < H1_NU; ++i) {
    T ui = u[i];
    if (val(ui) < H1_CTRLRANGE[i][0]) ui = T(H1_CTRLRANGE[i][0]);
    if (val(ui) > H1_CTRLRANGE[i][1]) ui = T(H1_CTRLRANGE[i][1]);
    tau[i] = ui - H1_DAMPING * x[H1_NQ + 6 + i];
  }
  T qacc[H1_NV];
  if (P.contact && stance) forward_dynamics_mj_stance(qh, x + 7, x + H1_NQ, tau, H1_ARMATURE + h * H1_DAMPING, P.g, h, P.soft, stance, qacc, P.contact, P.mu);
  else forward_dynamics_mj(qh, x + 7, x + H1_NQ, tau, H1_ARMATURE + h * H1_DAMPING, P.g, qacc);
  if (P.limits) {
    // Joint-limit rows (h1.xml jnt_range, enforced inside mj_step, robot_utils.cpp:113-114) restated as the rigid,
    // velocity-level limit of that constraint, like the stance rows: a hinge past its range that the step above would still move
    // outward (v_i + h qacc_i points out of the range) is stopped, v_i+ = 0, i.e. its acceleration is prescribed, qacc_i = -v_i / h,
    //     Mhat qacc + bias - J^T lambda - E^T mu = tau,   E qacc = -v_L / h   (+ the stance rows, solved on the system with these hinges prescribed)
    // -- Featherstone's hybrid dynamics through the same articulated-body recursion; the set is decided once, from the unlimited step.
    int lock[H1_NJ], any = 0; T lockacc[H1_NJ];
    // lim_k > 0 (round 6): the restoring term of MuJoCo's constraint reference a_ref = -b v - k r in its hard limit (b = 1 / h for the
    // clamped time constant 2 h, k = 1 / (2 h)^2 = 625 at h = 0.02): the row prescribes qacc_i = -v_i / h - k r_i, r_i = q_i - hi_i > 0 or
    // q_i - lo_i < 0, and is active when the unlimited step falls short of that on the outward side (k = 0: exactly the stop above).
    for (int i = 0; i < H1_NJ; ++i) {
      const double q = val(x[7 + i]);
      const double r = q > H1_JRANGE[i][1] ? q - H1_JRANGE[i][1] : (q < H1_JRANGE[i][0] ? q - H1_JRANGE[i][0] : 0.0);
      const double vnext = val(x[H1_NQ + 6 + i]) + h * (val(qacc[6 + i]) + P.lim_k * r);
      lock[i] = ((q > H1_JRANGE[i][1] && vnext > 0.0) || (q < H1_JRANGE[i][0] && vnext < 0.0)) ? 1 : 0;
      const T rT = q > H1_JRANGE[i][1] ? x[7 + i] - H1_JRANGE[i][1] : (q < H1_JRANGE[i][0] ? x[7 + i] - H1_JRANGE[i][0] : x[7 + i] * 0.0);
      lockacc[i] = -x[H1_NQ + 6 + i] / h - P.lim_k * rT;
      any |= lock[i];
    }
    if (any) {
      if (P.contact && stance) forward_dynamics_mj_stance(qh, x + 7, x + H1_NQ, tau, H1_ARMATURE + h * H1_DAMPING, P.g, h, P.soft, stance, qacc, P.contact, P.mu, lock, lockacc);
      else forward_dynamics_mj(qh, x + 7, x + H1_NQ, tau, H1_ARMATURE + h * H1_DAMPING, P.g, qacc, lock, lockacc);
    }
  }
  T vn[H1_NV];
  for (int i = 0; i < H1_NV; ++i) { vn[i] = x[H1_NQ + i] + h * qacc[i]; xn[H1_NQ + i] = vn[i]; }
  for (int k = 0; k < 3; ++k) xn[k] = x[k] + h * vn[k];
  for (int i = 0; i < H1_NJ; ++i) xn[7 + i] = x[7 + i] + h * vn[6 + i];
  // quat <- normalize(qh (x) exp(h * omega_body))
  T s = (vn[3] * vn[3] + vn[4] * vn[4] + vn[5] * vn[5]) * (h * h);
  T c, so; half_angle_cs(s, c, so);
  T ew = c, ex = so * h * vn[3], ey = so * h * vn[4], ez = so * h * vn[5];
  T rw = qh[0] * ew - qh[1] * ex - qh[2] * ey - qh[3] * ez;
  T rx = qh[0] * ex + qh[1] * ew + qh[2] * ez - qh[3] * ey;
  T ry = qh[0] * ey - qh[1] * ez + qh[2] * ew + qh[3] * ex;
  T rz = qh[0] * ez + qh[1] * ey - qh[2] * ex + qh[3] * ew;
  T rn = sqrt(rw * rw + rx * rx + ry * ry + rz * rz);
  xn[3] = rw / rn; xn[4] = rx / rn; xn[5] = ry / rn; xn[6] = rz / rn;
}

// MuJoCo-side forward kinematics (MJCF constants): world rotation/position of every body frame.
template <class T>
inline void fk_mj(const T* x, T (*Rw)[9], T (*pw)[3]) {
  T qn = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  quat_wxyz_to_R(x[3] / qn, x[4] / qn, x[5] / qn, x[6] / qn, Rw[0]);
  for (int k = 0; k < 3; ++k) pw[0][k] = x[k];
  for (int i = 1; i < H1_NB; ++i) {
    const int p = H1_PARENT[i];
    T Rj[9]; joint_rot(i, x[7 + i - 1], H1_RFIX, Rj);
    mat3_mul(Rw[p], Rj, Rw[i]);
    T t[3]; mat3_vec(Rw[p], H1_POS[i], t);
    for (int k = 0; k < 3; ++k) pw[i][k] = pw[p][k] + t[k];
  }
}

// RobotUtils::computeCoM (robot_utils.cpp:810-833): mass-weighted mean of MuJoCo xipos (MJCF masses)
template <class T>
inline void com_mj(const T* x, T* com) {
  T Rw[H1_NB][9], pw[H1_NB][3];
  fk_mj(x, Rw, pw);
  double mtot = 0.0; for (int k = 0; k < 3; ++k) com[k] = T(0.0);
  for (int i = 0; i < H1_NB; ++i) {
    T c[3]; mat3_vec(Rw[i], H1_COM[i], c);
    for (int k = 0; k < 3; ++k) com[k] += H1_MASS[i] * (pw[i][k] + c[k]);
    mtot += H1_MASS[i];
  }
  for (int k = 0; k < 3; ++k) com[k] = com[k] / mtot;
}

}  // namespace orc
