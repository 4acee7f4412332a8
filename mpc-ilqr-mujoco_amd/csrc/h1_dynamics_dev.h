// Device-side rigid-body dynamics of the Unitree H1 for gfx950 (fp64).
//
// Replaces RobotUtils::rolloutOneStep (reference src/common/robot_utils.cpp:106-117): one MuJoCo
// step in the constraint-free regime with MuJoCo's coordinates and integrator:
//   (M + armature + h D) qacc = clamp(u) - D qvel - bias ; qvel += h qacc ; qpos integrated with the
//   new velocity (world-frame linear, body-frame angular, quat (x) exp(h w), normalised).
// Forward dynamics is a chain-wise articulated-body sweep over the H1 tree
// (pelvis -> {left leg(5), right leg(5), torso -> {left arm(4), right arm(4)}}): only three 6x6
// articulated-inertia accumulators are live at any time (running chain, torso, pelvis).
// Templated on the scalar: double for rollouts, Dual for exact directional derivatives
// (one Jacobian column per thread in the linearisation kernel).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#define H1_CONST __constant__ static const
#include "h1_model_data.h"

#define DEVFN __device__ __forceinline__

namespace h1 {

// ---------- dual number (value + one tangent) ----------
struct Dual {
  double v, d;
  DEVFN Dual() : v(0.0), d(0.0) {}
  DEVFN Dual(double a) : v(a), d(0.0) {}
  DEVFN Dual(double a, double b) : v(a), d(b) {}
};
DEVFN Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
DEVFN Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
DEVFN Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
DEVFN Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, fma(a.d, b.v, a.v * b.d)); }
DEVFN Dual operator/(Dual a, Dual b) { double ib = 1.0 / b.v; double q = a.v * ib; return Dual(q, (a.d - q * b.d) * ib); }
DEVFN Dual operator+(Dual a, double b) { return Dual(a.v + b, a.d); }
DEVFN Dual operator+(double b, Dual a) { return Dual(a.v + b, a.d); }
DEVFN Dual operator-(Dual a, double b) { return Dual(a.v - b, a.d); }
DEVFN Dual operator-(double b, Dual a) { return Dual(b - a.v, -a.d); }
DEVFN Dual operator*(Dual a, double b) { return Dual(a.v * b, a.d * b); }
DEVFN Dual operator*(double b, Dual a) { return Dual(a.v * b, a.d * b); }
DEVFN Dual operator/(Dual a, double b) { double ib = 1.0 / b; return Dual(a.v * ib, a.d * ib); }
DEVFN Dual& operator+=(Dual& a, Dual b) { a.v += b.v; a.d += b.d; return a; }
DEVFN Dual& operator-=(Dual& a, Dual b) { a.v -= b.v; a.d -= b.d; return a; }
DEVFN Dual& operator+=(Dual& a, double b) { a.v += b; return a; }
DEVFN Dual dsin(Dual a) { double s, c; sincos(a.v, &s, &c); return Dual(s, c * a.d); }
DEVFN Dual dcos(Dual a) { double s, c; sincos(a.v, &s, &c); return Dual(c, -s * a.d); }
DEVFN Dual dsqrt(Dual a) { double r = sqrt(a.v); return Dual(r, 0.5 * a.d / r); }
DEVFN void dsincos(Dual a, Dual& s, Dual& c) { double sv, cv; sincos(a.v, &sv, &cv); s = Dual(sv, cv * a.d); c = Dual(cv, -sv * a.d); }
DEVFN double dsqrt(double a) { return sqrt(a); }
DEVFN void dsincos(double a, double& s, double& c) { sincos(a, &s, &c); }
DEVFN double val(double a) { return a; }
DEVFN double val(Dual a) { return a.v; }

struct DynParams {
  double h;
  double g[3];
  int contact;      // 0: constraint-free step; 1: rigid stance constraints on the scheduled feet;
                    // 2: the same, unilateral (a stance foot the floor would have to pull on is released)
                    // 3: unilateral + Coulomb limit (a foot outside the cone |f_t| <= mu f_n slides: tangential rows dropped)
  int limits;       // 1: joint-limit rows -- a hinge past its range that the step would still move outward is stopped (h1_aba_split.h
                    // step_stance<., true>); sits in what was padding: the layout of the rest (kernel arguments) does not move
  double soft;      // diagonal softness of the stance constraint (1 / kg)
  double mu;        // sliding friction coefficient of mode 3
  double lim_k;     // restoring stiffness of the joint-limit rows (1 / s^2): the row prescribes qacc_i = -v_i / h - lim_k r_i, r_i the violation
                    // (MuJoCo's solref reference acceleration in its hard limit; 0: the pure stop)
};
// the plant needs the constrained step (stance rows and / or joint-limit rows): the two-lane kernels with the shared constrained step
__host__ __device__ inline bool constrained(const DynParams& d) { return d.contact != 0 || d.limits != 0; }

// ---------- 3-vectors / 3x3 (row-major) ----------
template <class T, class U> DEVFN void cross3(const T* a, const U* b, T* c) {
  T c0 = a[1] * b[2] - a[2] * b[1], c1 = a[2] * b[0] - a[0] * b[2], c2 = a[0] * b[1] - a[1] * b[0];
  c[0] = c0; c[1] = c1; c[2] = c2;
}
template <class T, class U> DEVFN void mv3(const T* A, const U* x, T* y) {
  T y0 = A[0] * x[0] + A[1] * x[1] + A[2] * x[2], y1 = A[3] * x[0] + A[4] * x[1] + A[5] * x[2], y2 = A[6] * x[0] + A[7] * x[1] + A[8] * x[2];
  y[0] = y0; y[1] = y1; y[2] = y2;
}
template <class T, class U> DEVFN void mtv3(const T* A, const U* x, T* y) {
  T y0 = A[0] * x[0] + A[3] * x[1] + A[6] * x[2], y1 = A[1] * x[0] + A[4] * x[1] + A[7] * x[2], y2 = A[2] * x[0] + A[5] * x[1] + A[8] * x[2];
  y[0] = y0; y[1] = y1; y[2] = y2;
}
template <class T> DEVFN void quat_wxyz_R(const T& w, const T& x, const T& y, const T& z, T* R) {
  R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z); R[2] = 2.0 * (x * z + w * y);
  R[3] = 2.0 * (x * y + w * z); R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
  R[6] = 2.0 * (x * z - w * y); R[7] = 2.0 * (y * z + w * x); R[8] = 1.0 - 2.0 * (x * x + y * y);
}
// Rj = Rfix[i] * Rot(axis_i, theta) (child -> parent coordinates)
template <class T> DEVFN void joint_rot(int i, const T& th, const double (*rfix)[3][3], T* Rj) {
  T s, c; dsincos(th, s, c);
  const int a = H1_AXIS[i];
  const int b = (a + 1) % 3, d = (a + 2) % 3;  // Rot(a): [b][b]=c [b][d]=-s [d][b]=s [d][d]=c [a][a]=1
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double fa = rfix[i][r][a], fb = rfix[i][r][b], fd = rfix[i][r][d];
    Rj[3 * r + a] = T(fa);
    Rj[3 * r + b] = fb * c + fd * s;
    Rj[3 * r + d] = fd * c - fb * s;
  }
}

// ---------- spatial algebra, 6-vectors [angular; linear] ----------
template <class T> DEVFN void xf_motion(const T* Rj, const double* r, const T* vp, T* vc) {
  T t[3]; cross3(vp, r, t);
  T lin[3] = {vp[3] + t[0], vp[4] + t[1], vp[5] + t[2]};
  mtv3(Rj, vp, vc); mtv3(Rj, lin, vc + 3);
}
template <class T> DEVFN void xf_force_acc(const T* Rj, const double* r, const T* fc, T* fp) {
  T n[3], f[3]; mv3(Rj, fc, n); mv3(Rj, fc + 3, f);
  fp[0] += n[0] + (r[1] * f[2] - r[2] * f[1]);
  fp[1] += n[1] + (r[2] * f[0] - r[0] * f[2]);
  fp[2] += n[2] + (r[0] * f[1] - r[1] * f[0]);
  fp[3] += f[0]; fp[4] += f[1]; fp[5] += f[2];
}
template <class T> DEVFN void crm(const T* v, const T* m, T* out) {
  T a[3], b[3], c[3]; cross3(v, m, a); cross3(v + 3, m, b); cross3(v, m + 3, c);
  out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; out[3] = b[0] + c[0]; out[4] = b[1] + c[1]; out[5] = b[2] + c[2];
}
template <class T> DEVFN void crf(const T* v, const T* f, T* out) {
  T a[3], b[3], c[3]; cross3(v, f, a); cross3(v + 3, f + 3, b); cross3(v, f + 3, c);
  out[0] = a[0] + b[0]; out[1] = a[1] + b[1]; out[2] = a[2] + b[2]; out[3] = c[0]; out[4] = c[1]; out[5] = c[2];
}
// spatial inertia of body i about its frame origin times a motion vector: (I_o w + m c x v ; m v - m c x w)
template <class T> DEVFN void inertia_mul(int i, const T* a, T* f) {
  const double m = H1_MASS[i];
  const double c[3] = {H1_COM[i][0], H1_COM[i][1], H1_COM[i][2]};
  T Iw[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) Iw[r] = a[0] * H1_INERTIA[i][r][0] + a[1] * H1_INERTIA[i][r][1] + a[2] * H1_INERTIA[i][r][2];
  // linear acceleration of the CoM: v + w x c
  T wc[3] = {a[1] * c[2] - a[2] * c[1], a[2] * c[0] - a[0] * c[2], a[0] * c[1] - a[1] * c[0]};
  T fl[3] = {m * (a[3] + wc[0]), m * (a[4] + wc[1]), m * (a[5] + wc[2])};
  f[0] = Iw[0] + (c[1] * fl[2] - c[2] * fl[1]);
  f[1] = Iw[1] + (c[2] * fl[0] - c[0] * fl[2]);
  f[2] = Iw[2] + (c[0] * fl[1] - c[1] * fl[0]);
  f[3] = fl[0]; f[4] = fl[1]; f[5] = fl[2];
}
// dense 6x6 spatial inertia of body i about its frame origin (symmetric)
template <class T> DEVFN void inertia_dense(int i, T* I) {
  const double m = H1_MASS[i];
  const double c[3] = {H1_COM[i][0], H1_COM[i][1], H1_COM[i][2]};
  const double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
  const double cx[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      I[6 * r + s] = T(H1_INERTIA[i][r][s] + m * ((r == s ? cc : 0.0) - c[r] * c[s]));
      I[6 * r + 3 + s] = T(m * cx[3 * r + s]);
      I[6 * (3 + r) + s] = T(m * cx[3 * s + r]);
      I[6 * (3 + r) + 3 + s] = T(r == s ? m : 0.0);
    }
}
// Y += X^T Ia X with X = Plucker motion transform (parent -> child) built from (Rj, r)
template <class T> DEVFN void fold_inertia(const T* Rj, const double* r, const T* Ia, T* Y) {
  // X = [[E, 0], [-E rx, E]],  E = Rj^T
  T X[36];
  const double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const T e = Rj[3 * j + i];
      X[6 * i + j] = e; X[6 * (i + 3) + (j + 3)] = e; X[6 * i + (j + 3)] = T(0.0);
      X[6 * (i + 3) + j] = -(Rj[0 + i] * rx[j] + Rj[3 + i] * rx[3 + j] + Rj[6 + i] * rx[6 + j]);
    }
  T tmp[36];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      T s = Ia[6 * a] * X[c];
#pragma unroll
      for (int k = 1; k < 6; ++k) s += Ia[6 * a + k] * X[6 * k + c];
      tmp[6 * a + c] = s;
    }
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      T s = X[a] * tmp[c];
#pragma unroll
      for (int k = 1; k < 6; ++k) s += X[6 * k + a] * tmp[6 * k + c];
      Y[6 * a + c] += s;
    }
}
// solve the SPD 6x6 system A y = b (LDL^T, no pivoting)
template <class T> DEVFN void solve6(const T* A, const T* b, T* y) {
  T L[36], d[6], z[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    T s = A[6 * j + j];
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k < j) s -= L[6 * j + k] * L[6 * j + k] * d[k];
    d[j] = s;
#pragma unroll
    for (int i = 0; i < 6; ++i) if (i > j) {
      T t = A[6 * i + j];
#pragma unroll
      for (int k = 0; k < 6; ++k) if (k < j) t -= L[6 * i + k] * L[6 * j + k] * d[k];
      L[6 * i + j] = t / d[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    T s = b[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k < i) s -= L[6 * i + k] * z[k];
    z[i] = s;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) z[i] = z[i] / d[i];
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    T s = z[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k > i) s -= L[6 * k + i] * y[k];
    y[i] = s;
  }
}

// ---------- forward dynamics in MuJoCo coordinates ----------
// quat_hat unit (wxyz); theta[19]; v = qvel[25]; tau[19]; returns qacc[25]
// Optional extras: tau_base = generalized force on the free joint in MuJoCo coordinates (world-frame
// linear force, body-frame torque); dump = primal per-body quantities for the linearisation kernel.
struct KnotDump {
  double R0[9];                 // base rotation, body -> world
  double aL[3];                 // R0^T (qacc_lin - g)
  double qacc[H1_NV];           // MuJoCo-coordinate acceleration of the knot
  double Rj[H1_NB][9];          // joint rotations child -> parent ([0] unused)
  double v[H1_NB][6];           // body spatial velocities (body coordinates, [ang; lin])
  double a[H1_NB][6];           // gravity-offset spatial accelerations
  double F[H1_NB][6];           // accumulated inverse-dynamics forces (body i and its subtree)
  double U[H1_NB][6];           // articulated-body U_i = IA_i S_i ([0] unused)
  double Dinv[H1_NB];           // 1 / (S_i^T U_i + effective armature)
  double IA0inv[36];            // inverse of the pelvis articulated inertia
};
template <class T, bool DUMP = false>
DEVFN void forward_dynamics(const T* quat_hat, const T* theta, const T* v, const T* tau, double arm_eff,
                            const double* grav, T* qacc, const T* tau_base = nullptr, KnotDump* dump = nullptr) {
  T R0[9]; quat_wxyz_R(quat_hat[0], quat_hat[1], quat_hat[2], quat_hat[3], R0);
  // per-body state kept for the outward acceleration sweep
  T vel[H1_NB][6];                   // body spatial velocity (body coordinates)
  T U[H1_NB][6], Dinv[H1_NB], uu[H1_NB];
  T pA[H1_NB][6];                    // bias force accumulators
#pragma unroll
  for (int k = 0; k < 3; ++k) vel[0][k] = v[3 + k];
  mtv3(R0, v, vel[0] + 3);
  // outward sweep: velocities, velocity-product bias forces
  for (int i = 0; i < H1_NB; ++i) {
    if (i > 0) {
      T Rj[9]; joint_rot(i, theta[i - 1], H1_RFIX, Rj);
      xf_motion(Rj, H1_POS[i], vel[H1_PARENT[i]], vel[i]);
      vel[i][H1_AXIS[i]] += v[6 + i - 1];
    }
    T Iv[6]; inertia_mul(i, vel[i], Iv);
    crf(vel[i], Iv, pA[i]);
  }
  // inward sweep, chain-wise accumulators
  T IAchain[36], IAtorso[36], IApelvis[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) { IAchain[k] = T(0.0); IAtorso[k] = T(0.0); IApelvis[k] = T(0.0); }
  for (int i = H1_NB - 1; i >= 1; --i) {
    const int a = H1_AXIS[i], p = H1_PARENT[i];
    const bool leaf = (i == 5 || i == 10 || i == 15 || i == 19);
    T IA[36]; inertia_dense(i, IA);
    if (i == 11) {
#pragma unroll
      for (int k = 0; k < 36; ++k) IA[k] += IAtorso[k];
    } else if (!leaf) {
#pragma unroll
      for (int k = 0; k < 36; ++k) IA[k] += IAchain[k];
    }
    T Rj[9]; joint_rot(i, theta[i - 1], H1_RFIX, Rj);
    // velocity-product acceleration c = v x (S qd): only S = e_a (angular)
    T cb[6];
    {
      T vJ[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) vJ[k] = T(0.0);
      vJ[a] = v[6 + i - 1];
      crm(vel[i], vJ, cb);
    }
    T Ui[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { Ui[k] = IA[6 * k + a]; U[i][k] = Ui[k]; }
    const T D = Ui[a] + arm_eff;
    const T di = T(1.0) / D;
    Dinv[i] = di;
    const T ui = tau[i - 1] - pA[i][a];
    uu[i] = ui;
    T Ia[36];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) Ia[6 * r + c] = IA[6 * r + c] - Ui[r] * Ui[c] * di;
    T pa[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      T s = pA[i][r] + Ui[r] * (ui * di);
#pragma unroll
      for (int k = 0; k < 6; ++k) s += Ia[6 * r + k] * cb[k];
      pa[r] = s;
    }
    T* dst = (p == 0) ? IApelvis : ((p == 11) ? IAtorso : IAchain);
    if (!(p == 0 || p == 11)) {
#pragma unroll
      for (int k = 0; k < 36; ++k) IAchain[k] = T(0.0);
    }
    fold_inertia(Rj, H1_POS[i], Ia, dst);
    xf_force_acc(Rj, H1_POS[i], pa, pA[p]);
  }
  // pelvis (free joint): IA0 = I0 + accumulated; gravity enters as base acceleration offset
  T IA0[36]; inertia_dense(0, IA0);
#pragma unroll
  for (int k = 0; k < 36; ++k) IA0[k] += IApelvis[k];
  T rhs[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) rhs[k] = -pA[0][k];
  if (tau_base) {  // external generalized force on the free joint: (body torque, R0^T world force)
    T fl[3]; mtv3(R0, tau_base, fl);
#pragma unroll
    for (int k = 0; k < 3; ++k) { rhs[k] += tau_base[3 + k]; rhs[3 + k] += fl[k]; }
  }
  T acc[H1_NB][6];
  solve6(IA0, rhs, acc[0]);
  T mg[3] = {T(-grav[0]), T(-grav[1]), T(-grav[2])};
  T a0p[3]; mtv3(R0, mg, a0p);
  T nudot[6] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3] - a0p[0], acc[0][4] - a0p[1], acc[0][5] - a0p[2]};
  for (int i = 1; i < H1_NB; ++i) {
    const int a = H1_AXIS[i];
    T Rj[9]; joint_rot(i, theta[i - 1], H1_RFIX, Rj);
    T ap[6]; xf_motion(Rj, H1_POS[i], acc[H1_PARENT[i]], ap);
    T vJ[6], cb[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) vJ[k] = T(0.0);
    vJ[a] = v[6 + i - 1];
    crm(vel[i], vJ, cb);
    T s = uu[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) { ap[k] += cb[k]; s -= U[i][k] * ap[k]; }
    const T qdd = s * Dinv[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[i][k] = ap[k];
    acc[i][a] += qdd;
    qacc[6 + i - 1] = qdd;
  }
  T wxv[3]; cross3(vel[0], vel[0] + 3, wxv);
  T lin[3] = {nudot[3] + wxv[0], nudot[4] + wxv[1], nudot[5] + wxv[2]};
  mv3(R0, lin, qacc);
  qacc[3] = nudot[0]; qacc[4] = nudot[1]; qacc[5] = nudot[2];
  if constexpr (DUMP) {
    KnotDump& Dm = *dump;
#pragma unroll
    for (int k = 0; k < 9; ++k) Dm.R0[k] = R0[k];
    for (int k = 0; k < H1_NV; ++k) Dm.qacc[k] = qacc[k];
    // aL = R0^T (qacc_lin - g) = a0_lin(gravity-offset) + omega x v_O
    for (int k = 0; k < 3; ++k) Dm.aL[k] = acc[0][3 + k] + wxv[k];
    for (int i = 0; i < H1_NB; ++i) {
      for (int k = 0; k < 6; ++k) { Dm.v[i][k] = vel[i][k]; Dm.a[i][k] = acc[i][k]; }
      T Iv[6], Ia_[6], vIv[6]; inertia_mul(i, vel[i], Iv); inertia_mul(i, acc[i], Ia_); crf(vel[i], Iv, vIv);
      for (int k = 0; k < 6; ++k) Dm.F[i][k] = Ia_[k] + vIv[k];
    }
    for (int i = H1_NB - 1; i >= 1; --i) {
      T Rj[9]; joint_rot(i, theta[i - 1], H1_RFIX, Rj);
      for (int k = 0; k < 9; ++k) Dm.Rj[i][k] = Rj[k];
      xf_force_acc(Rj, H1_POS[i], Dm.F[i], Dm.F[H1_PARENT[i]]);
      for (int k = 0; k < 6; ++k) Dm.U[i][k] = U[i][k];
      Dm.Dinv[i] = Dinv[i];
    }
    for (int c = 0; c < 6; ++c) {   // explicit inverse of the SPD 6x6 pelvis articulated inertia
      T e[6] = {T(0.0), T(0.0), T(0.0), T(0.0), T(0.0), T(0.0)}, col[6];
      e[c] = T(1.0);
      solve6(IA0, e, col);
      for (int r = 0; r < 6; ++r) Dm.IA0inv[6 * r + c] = col[r];
    }
  }
}

// ---------- forward dynamics with schedule-driven rigid stance constraints ----------
// The reference's plant is MuJoCo with floor contacts (RobotUtils::rolloutOneStep, src/common/robot_utils.cpp:106-117).
// Restated as the regime its standing / walking scenarios run in: a foot the contact schedule marks as stance
// (RobotUtils::isStance, robot_utils.cpp:494-504) does not move.  Velocity-level constraint over one step, consistent
// with the semi-implicit Euler integrator:  v_f + h a_f = 0  (spatial velocity / acceleration of the ankle link in link
// coordinates),  a_f = a_f,free + C lambda,  C = J Mhat^-1 J^T.  C is built by propagating unit wrenches through the
// articulated-body quantities of the free solve (inward along the leg, pelvis solve, outward) -- neither M nor J is
// formed; (C + soft I) lambda = rhs by Cholesky; the wrench is propagated once more for the joint accelerations.
// Scalar (scratch-resident) path, one lane per rollout; Jacobians of this step come from k_linearize_fd
// (the reference's own forward differences, robot_utils.cpp:120-160).
__device__ inline bool chol_solve_small(double* A, double* b, int n) {
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0)) return false;
    d = sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[i * n + k] * b[k]; b[i] = s / A[i * n + i]; }
  for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= A[k * n + i] * b[k]; b[i] = s / A[i * n + i]; }
  return true;
}
// response to wrenches fext[g] (link coordinates) on the stance feet fb[0..nf): accelerations of those feet and, if
// `full`, of the pelvis (dbase) and every hinge (dqdd[1..19])
__device__ inline void stance_respond(const KnotDump& Dm, int nf, const int* fb, const double (*fext)[6], double (*da_feet)[6],
                                      bool full, double* dqdd, double* dbase) {
  double dpA[H1_NB][6], du[H1_NB], da[H1_NB][6];
  for (int i = 0; i < H1_NB; ++i) { du[i] = 0.0; for (int k = 0; k < 6; ++k) dpA[i][k] = 0.0; }
  for (int g = 0; g < nf; ++g) for (int k = 0; k < 6; ++k) dpA[fb[g]][k] -= fext[g][k];
  for (int i = 10; i >= 1; --i) {                       // the two leg chains
    const int a = H1_AXIS[i];
    du[i] = -dpA[i][a];
    double dpa[6]; for (int k = 0; k < 6; ++k) dpa[k] = dpA[i][k] + Dm.U[i][k] * (du[i] * Dm.Dinv[i]);
    xf_force_acc(Dm.Rj[i], H1_POS[i], dpa, dpA[H1_PARENT[i]]);
  }
  for (int r = 0; r < 6; ++r) { double s = 0.0; for (int c = 0; c < 6; ++c) s -= Dm.IA0inv[6 * r + c] * dpA[0][c]; da[0][r] = s; }
  const int last = full ? H1_NB - 1 : 10;
  for (int i = 1; i <= last; ++i) {
    const int a = H1_AXIS[i];
    double ap[6]; xf_motion(Dm.Rj[i], H1_POS[i], da[H1_PARENT[i]], ap);
    double s = du[i]; for (int k = 0; k < 6; ++k) s -= Dm.U[i][k] * ap[k];
    const double q = s * Dm.Dinv[i];
    for (int k = 0; k < 6; ++k) da[i][k] = ap[k];
    da[i][a] += q;
    if (full) dqdd[i] = q;
  }
  for (int g = 0; g < nf; ++g) for (int k = 0; k < 6; ++k) da_feet[g][k] = da[fb[g]][k];
  if (full) for (int k = 0; k < 6; ++k) dbase[k] = da[0][k];
}
__device__ inline void forward_dynamics_stance(const double* quat_hat, const double* theta, const double* v, const double* tau, double arm_eff,
                                               const double* grav, double h, double soft, const int* stance, double* qacc, int contact_mode = 1) {
  KnotDump Dm;
  forward_dynamics<double, true>(quat_hat, theta, v, tau, arm_eff, grav, qacc, nullptr, &Dm);
  int nf = 0, fb[2];
  if (stance[0] == 1) fb[nf++] = 5;      // left ankle link
  if (stance[1] == 1) fb[nf++] = 10;     // right ankle link
  if (nf == 0) return;
  // true (not gravity-offset) spatial acceleration of the foot: a_f - X_{f<-0} (0, R0^T(-g))
  const double mg[3] = {-grav[0], -grav[1], -grav[2]};
  double a0p[3]; mtv3(Dm.R0, mg, a0p);
  double b[12];
  auto solve_set = [&]() {          // (C + soft I) lambda = b for the current stance set fb[0..nf)
    const int nc = 6 * nf;
    double C[144];
    for (int g = 0; g < nf; ++g)
      for (int c = 0; c < 6; ++c) {
        double fext[2][6], daf[2][6];
        for (int g2 = 0; g2 < 2; ++g2) for (int k = 0; k < 6; ++k) fext[g2][k] = 0.0;
        fext[g][c] = 1.0;
        stance_respond(Dm, nf, fb, fext, daf, false, nullptr, nullptr);
        for (int g2 = 0; g2 < nf; ++g2) for (int k = 0; k < 6; ++k) C[(6 * g2 + k) * nc + 6 * g + c] = daf[g2][k];
      }
    for (int i = 0; i < nc; ++i) C[i * nc + i] += soft;
    for (int g = 0; g < nf; ++g) {
      double off[6] = {0.0, 0.0, 0.0, a0p[0], a0p[1], a0p[2]};
      for (int i = fb[g] - 4; i <= fb[g]; ++i) { double o2[6]; xf_motion(Dm.Rj[i], H1_POS[i], off, o2); for (int k = 0; k < 6; ++k) off[k] = o2[k]; }
      for (int k = 0; k < 6; ++k) b[6 * g + k] = -Dm.v[fb[g]][k] / h - (Dm.a[fb[g]][k] - off[k]);
    }
    chol_solve_small(C, b, nc);
  };
  solve_set();
  if (contact_mode == 2) {
    // unilateral: normal force on foot g = (world up axis in link coordinates) . (force part of lambda_g); feet with a
    // negative one are released and the remaining set is solved again (once)
    int keep[2], nk = 0;
    for (int g = 0; g < nf; ++g) {
      double zl[6] = {0.0, 0.0, 0.0, Dm.R0[6], Dm.R0[7], Dm.R0[8]};
      for (int i = fb[g] - 4; i <= fb[g]; ++i) { double o2[6]; xf_motion(Dm.Rj[i], H1_POS[i], zl, o2); for (int k = 0; k < 6; ++k) zl[k] = o2[k]; }
      const double fz = zl[3] * b[6 * g + 3] + zl[4] * b[6 * g + 4] + zl[5] * b[6 * g + 5];
      if (!(fz < 0.0)) keep[nk++] = fb[g];
    }
    if (nk < nf) { nf = nk; for (int g = 0; g < nk; ++g) fb[g] = keep[g]; if (nf > 0) solve_set(); }
  }
  if (nf == 0) return;
  double fext[2][6], daf[2][6], dq[H1_NB], da0[6];
  for (int g = 0; g < 2; ++g) for (int k = 0; k < 6; ++k) fext[g][k] = g < nf ? b[6 * g + k] : 0.0;
  stance_respond(Dm, nf, fb, fext, daf, true, dq, da0);
  for (int i = 1; i < H1_NB; ++i) qacc[6 + i - 1] += dq[i];
  double dl[3]; mv3(Dm.R0, da0 + 3, dl);
  for (int k = 0; k < 3; ++k) { qacc[k] += dl[k]; qacc[3 + k] += da0[k]; }
}

// cos(a/2) and sin(a/2)/a as smooth functions of s = a^2
template <class T> DEVFN void half_angle_cs(const T& s, T& c, T& so) {
  if (val(s) < 1e-6) {
    c = 1.0 - s / 8.0 + s * s / 384.0 - s * s * s / 46080.0;
    so = 0.5 - s / 48.0 + s * s / 3840.0 - s * s * s / 645120.0;
  } else {
    T a = dsqrt(s), sn, cn; dsincos(a * 0.5, sn, cn);
    c = cn; so = sn / a;
  }
}

// x_next = f(x, u); `stance` (left / right flag of the knot being stepped) is used when P.contact != 0 (double only)
template <class T>
DEVFN void step(const T* x, const T* u, const DynParams& P, T* xn, const int* stance = nullptr) {
  const double h = P.h;
  const T qn = dsqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  const T qh[4] = {x[3] / qn, x[4] / qn, x[5] / qn, x[6] / qn};
  T tau[H1_NU];
#pragma unroll
  for (int i = 0; i < H1_NU; ++i) {
    T ui = u[i];
    if (val(ui) < H1_CTRLRANGE[i][0]) ui = T(H1_CTRLRANGE[i][0]);
    if (val(ui) > H1_CTRLRANGE[i][1]) ui = T(H1_CTRLRANGE[i][1]);
    tau[i] = ui - H1_DAMPING * x[H1_NQ + 6 + i];
  }
  T qacc[H1_NV];
  bool done = false;
  if constexpr (std::is_same<T, double>::value) {
    if (P.contact && stance) { forward_dynamics_stance(qh, x + 7, x + H1_NQ, tau, H1_ARMATURE + h * H1_DAMPING, P.g, h, P.soft, stance, qacc, P.contact); done = true; }
  }
  if (!done) forward_dynamics(qh, x + 7, x + H1_NQ, tau, H1_ARMATURE + h * H1_DAMPING, P.g, qacc);
  T vn[H1_NV];
#pragma unroll
  for (int i = 0; i < H1_NV; ++i) { vn[i] = x[H1_NQ + i] + h * qacc[i]; xn[H1_NQ + i] = vn[i]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) xn[k] = x[k] + h * vn[k];
#pragma unroll
  for (int i = 0; i < H1_NJ; ++i) xn[7 + i] = x[7 + i] + h * vn[6 + i];
  const T s = (vn[3] * vn[3] + vn[4] * vn[4] + vn[5] * vn[5]) * (h * h);
  T c, so; half_angle_cs(s, c, so);
  const T ew = c, ex = so * h * vn[3], ey = so * h * vn[4], ez = so * h * vn[5];
  const T rw = qh[0] * ew - qh[1] * ex - qh[2] * ey - qh[3] * ez;
  const T rx = qh[0] * ex + qh[1] * ew + qh[2] * ez - qh[3] * ey;
  const T ry = qh[0] * ey - qh[1] * ez + qh[2] * ew + qh[3] * ex;
  const T rz = qh[0] * ez + qh[1] * ey - qh[2] * ex + qh[3] * ew;
  const T rn = dsqrt(rw * rw + rx * rx + ry * ry + rz * rz);
  xn[3] = rw / rn; xn[4] = rx / rn; xn[5] = ry / rn; xn[6] = rz / rn;
}

// whole-body CoM with MuJoCo masses (RobotUtils::computeCoM, robot_utils.cpp:810-833)
__device__ inline void com_mj(const double* x, double* com) {
  double Rw[H1_NB][9], pw[H1_NB][3];
  const double qn = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  quat_wxyz_R(x[3] / qn, x[4] / qn, x[5] / qn, x[6] / qn, Rw[0]);
  pw[0][0] = x[0]; pw[0][1] = x[1]; pw[0][2] = x[2];
  double acc[3] = {0.0, 0.0, 0.0}, mtot = 0.0;
  for (int i = 0; i < H1_NB; ++i) {
    if (i > 0) {
      const int p = H1_PARENT[i];
      double Rj[9]; joint_rot(i, x[7 + i - 1], H1_RFIX, Rj);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Rw[i][3 * r + c] = Rw[p][3 * r] * Rj[c] + Rw[p][3 * r + 1] * Rj[3 + c] + Rw[p][3 * r + 2] * Rj[6 + c];
      double t[3]; mv3(Rw[p], H1_POS[i], t);
      pw[i][0] = pw[p][0] + t[0]; pw[i][1] = pw[p][1] + t[1]; pw[i][2] = pw[p][2] + t[2];
    }
    double c[3]; mv3(Rw[i], H1_COM[i], c);
    const double m = H1_MASS[i];
    acc[0] += m * (pw[i][0] + c[0]); acc[1] += m * (pw[i][1] + c[1]); acc[2] += m * (pw[i][2] + c[2]);
    mtot += m;
  }
  com[0] = acc[0] / mtot; com[1] = acc[1] / mtot; com[2] = acc[2] / mtot;
}

}  // namespace h1
