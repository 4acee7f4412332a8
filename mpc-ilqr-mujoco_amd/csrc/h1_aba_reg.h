// Register/LDS-resident articulated-body forward dynamics of the H1, one rollout per lane.
//
// Same algorithm and MuJoCo semantics as h1_dynamics_dev.h (which stays as the templated scalar/Dual
// implementation used by the FD Jacobian mode and the CPU probes) but laid out for the SIMD: the tree is
// fully unrolled with compile-time body indices (h1_model_constexpr.h), so joint axes, fixed rotations,
// offsets and inertias are immediates; articulated inertias are symmetric 3x3 blocks {A, B, C} (21 numbers)
// transformed by a Givens rotation about the joint axis plus a translation instead of a dense 6x6 congruence;
// the only quantities that must survive from the inward to the outward sweep (U_i, 1/D_i, u_i: 8 per hinge)
// live in LDS, one conflict-free column per lane.  Nothing spills to scratch by design.
#pragma once
#include <hip/hip_runtime.h>

#include "h1_fast_math.h"
#include "h1_model_constexpr.h"

#ifndef DEVFN
#define DEVFN __device__ __forceinline__
#endif

namespace h1r {
using namespace h1c;

constexpr int NB = 20, NJ = 19, NQ = 26, NV = 25, NX = 51, NU = 19;
constexpr double DAMPING = 1.0, ARMATURE = 0.1;
constexpr int LDS_SLOTS = 8 * NJ;   // U(6), Dinv, u per hinge

struct LaneLds {           // per-lane column of the LDS scratch: slot s lives at base[s * stride + lane]
  double* base; int stride, lane;
  DEVFN double& operator[](int s) const { return base[s * stride + lane]; }
};

struct Art { double A[6], B[9], C[6]; };   // [[A, B], [B^T, C]], A and C symmetric (xx, xy, xz, yy, yz, zz)
// index of (r, c) in the packed symmetric storage (xx, xy, xz, yy, yz, zz).  NOT recursive on purpose: the recursive
// form survives to the backend as a run-time loop, its result indexes the arrays dynamically and they end up in scratch.
DEVFN constexpr int sidx(int r, int c) { const int lo = r < c ? r : c, hi = r < c ? c : r; return lo * 3 - lo * (lo + 1) / 2 + hi; }

DEVFN void cross(const double* a, const double* b, double* c) {
  const double c0 = a[1] * b[2] - a[2] * b[1], c1 = a[2] * b[0] - a[0] * b[2], c2 = a[0] * b[1] - a[1] * b[0];
  c[0] = c0; c[1] = c1; c[2] = c2;
}
template <int AX> DEVFN void cross_axis(const double* a, double* o) {   // a x e_AX
  if constexpr (AX == 0) { o[0] = 0.0; o[1] = a[2]; o[2] = -a[1]; }
  else if constexpr (AX == 1) { o[0] = -a[2]; o[1] = 0.0; o[2] = a[0]; }
  else { o[0] = a[1]; o[1] = -a[0]; o[2] = 0.0; }
}
template <int I> constexpr bool rfix_identity() {
  return C_RFIX[I][0][0] == 1.0 && C_RFIX[I][1][1] == 1.0 && C_RFIX[I][2][2] == 1.0;
}
// y = Rj x with Rj = Rfix_I * Rot(AX_I, theta) (child -> parent coordinates)
template <int I> DEVFN void rot(const double* x, double s, double c, double* y) {
  constexpr int a = C_AXIS[I], b = (a + 1) % 3, d = (a + 2) % 3;
  double t[3];
  t[a] = x[a]; t[b] = c * x[b] - s * x[d]; t[d] = s * x[b] + c * x[d];
  if constexpr (rfix_identity<I>()) { y[0] = t[0]; y[1] = t[1]; y[2] = t[2]; }
  else {
    y[0] = C_RFIX[I][0][0] * t[0] + C_RFIX[I][0][1] * t[1] + C_RFIX[I][0][2] * t[2];
    y[1] = C_RFIX[I][1][0] * t[0] + C_RFIX[I][1][1] * t[1] + C_RFIX[I][1][2] * t[2];
    y[2] = C_RFIX[I][2][0] * t[0] + C_RFIX[I][2][1] * t[1] + C_RFIX[I][2][2] * t[2];
  }
}
// y = Rj^T x
template <int I> DEVFN void rotT(const double* x, double s, double c, double* y) {
  constexpr int a = C_AXIS[I], b = (a + 1) % 3, d = (a + 2) % 3;
  double t[3];
  if constexpr (rfix_identity<I>()) { t[0] = x[0]; t[1] = x[1]; t[2] = x[2]; }
  else {
    t[0] = C_RFIX[I][0][0] * x[0] + C_RFIX[I][1][0] * x[1] + C_RFIX[I][2][0] * x[2];
    t[1] = C_RFIX[I][0][1] * x[0] + C_RFIX[I][1][1] * x[1] + C_RFIX[I][2][1] * x[2];
    t[2] = C_RFIX[I][0][2] * x[0] + C_RFIX[I][1][2] * x[1] + C_RFIX[I][2][2] * x[2];
  }
  y[a] = t[a]; y[b] = c * t[b] + s * t[d]; y[d] = -s * t[b] + c * t[d];
}
// motion transform parent -> child of body I
template <int I> DEVFN void xf_motion(const double* vp, double s, double c, double* vc) {
  const double r[3] = {C_POS[I][0], C_POS[I][1], C_POS[I][2]};
  double t[3]; cross(vp, r, t);
  const double lin[3] = {vp[3] + t[0], vp[4] + t[1], vp[5] + t[2]};
  rotT<I>(vp, s, c, vc); rotT<I>(lin, s, c, vc + 3);
}
// force transform child -> parent of body I, accumulating
template <int I> DEVFN void xf_force_acc(const double* fc, double s, double c, double* fp) {
  const double r[3] = {C_POS[I][0], C_POS[I][1], C_POS[I][2]};
  double n[3], f[3], rf[3]; rot<I>(fc, s, c, n); rot<I>(fc + 3, s, c, f); cross(r, f, rf);
  fp[0] += n[0] + rf[0]; fp[1] += n[1] + rf[1]; fp[2] += n[2] + rf[2];
  fp[3] += f[0]; fp[4] += f[1]; fp[5] += f[2];
}
// spatial inertia of body I (about its frame origin) times a motion vector
template <int I> DEVFN void inertia_mul(const double* a, double* f) {
  constexpr double m = C_MASS[I];
  const double c[3] = {C_COM[I][0], C_COM[I][1], C_COM[I][2]};
  const double Iw0 = C_INERTIA[I][0][0] * a[0] + C_INERTIA[I][0][1] * a[1] + C_INERTIA[I][0][2] * a[2];
  const double Iw1 = C_INERTIA[I][1][0] * a[0] + C_INERTIA[I][1][1] * a[1] + C_INERTIA[I][1][2] * a[2];
  const double Iw2 = C_INERTIA[I][2][0] * a[0] + C_INERTIA[I][2][1] * a[1] + C_INERTIA[I][2][2] * a[2];
  const double wc[3] = {a[1] * c[2] - a[2] * c[1], a[2] * c[0] - a[0] * c[2], a[0] * c[1] - a[1] * c[0]};
  const double fl[3] = {m * (a[3] + wc[0]), m * (a[4] + wc[1]), m * (a[5] + wc[2])};
  f[0] = Iw0 + (c[1] * fl[2] - c[2] * fl[1]);
  f[1] = Iw1 + (c[2] * fl[0] - c[0] * fl[2]);
  f[2] = Iw2 + (c[0] * fl[1] - c[1] * fl[0]);
  f[3] = fl[0]; f[4] = fl[1]; f[5] = fl[2];
}
// v x* f
DEVFN void crf(const double* v, const double* f, double* out) {
  double a[3], b[3], c[3]; cross(v, f, a); cross(v + 3, f + 3, b); cross(v, f + 3, c);
  out[0] = a[0] + b[0]; out[1] = a[1] + b[1]; out[2] = a[2] + b[2]; out[3] = c[0]; out[4] = c[1]; out[5] = c[2];
}
// rigid-body inertia of body I as blocks
template <int I> DEVFN void body_inertia(Art& Y) {
  constexpr double m = C_MASS[I];
  constexpr double cx = C_COM[I][0], cy = C_COM[I][1], cz = C_COM[I][2], cc = cx * cx + cy * cy + cz * cz;
  Y.A[0] = C_INERTIA[I][0][0] + m * (cc - cx * cx); Y.A[1] = C_INERTIA[I][0][1] - m * cx * cy; Y.A[2] = C_INERTIA[I][0][2] - m * cx * cz;
  Y.A[3] = C_INERTIA[I][1][1] + m * (cc - cy * cy); Y.A[4] = C_INERTIA[I][1][2] - m * cy * cz; Y.A[5] = C_INERTIA[I][2][2] + m * (cc - cz * cz);
  // B = m [c]x
  Y.B[0] = 0.0; Y.B[1] = -m * cz; Y.B[2] = m * cy; Y.B[3] = m * cz; Y.B[4] = 0.0; Y.B[5] = -m * cx; Y.B[6] = -m * cy; Y.B[7] = m * cx; Y.B[8] = 0.0;
  Y.C[0] = m; Y.C[1] = 0.0; Y.C[2] = 0.0; Y.C[3] = m; Y.C[4] = 0.0; Y.C[5] = m;
}
DEVFN void art_add(Art& Y, const Art& X) {
#pragma unroll
  for (int k = 0; k < 6; ++k) { Y.A[k] += X.A[k]; Y.C[k] += X.C[k]; }
#pragma unroll
  for (int k = 0; k < 9; ++k) Y.B[k] += X.B[k];
}
DEVFN void art_zero(Art& Y) {
#pragma unroll
  for (int k = 0; k < 6; ++k) { Y.A[k] = 0.0; Y.C[k] = 0.0; }
#pragma unroll
  for (int k = 0; k < 9; ++k) Y.B[k] = 0.0;
}
// M (general 3x3, row-major) <- Rj M Rj^T
template <int I> DEVFN void rot_congruence(double* M, double s, double c) {
  double T[9];
#pragma unroll
  for (int col = 0; col < 3; ++col) { const double x[3] = {M[col], M[3 + col], M[6 + col]}; double y[3]; rot<I>(x, s, c, y); T[col] = y[0]; T[3 + col] = y[1]; T[6 + col] = y[2]; }
#pragma unroll
  for (int row = 0; row < 3; ++row) { double y[3]; rot<I>(T + 3 * row, s, c, y); M[3 * row] = y[0]; M[3 * row + 1] = y[1]; M[3 * row + 2] = y[2]; }
}
DEVFN void sym_to_full(const double* S, double* M) { M[0] = S[0]; M[1] = S[1]; M[2] = S[2]; M[3] = S[1]; M[4] = S[3]; M[5] = S[4]; M[6] = S[2]; M[7] = S[4]; M[8] = S[5]; }
// Yp += X^T Ya X for the joint transform of body I (rotation Rj, then translation by r = pos_I)
template <int I> DEVFN void fold_art(const Art& Ya, double s, double c, Art& Yp) {
  double A[9], B[9], C[9];
  sym_to_full(Ya.A, A); sym_to_full(Ya.C, C);
#pragma unroll
  for (int k = 0; k < 9; ++k) B[k] = Ya.B[k];
  rot_congruence<I>(A, s, c); rot_congruence<I>(B, s, c); rot_congruence<I>(C, s, c);
  constexpr double rx = C_POS[I][0], ry = C_POS[I][1], rz = C_POS[I][2];
  // R = [r]x
  const double R[9] = {0.0, -rz, ry, rz, 0.0, -rx, -ry, rx, 0.0};
  // RC = R C (C symmetric), Bp = B + RC
  double RC[9], Bp[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) { RC[3 * i + j] = R[3 * i] * C[j] + R[3 * i + 1] * C[3 + j] + R[3 * i + 2] * C[6 + j]; Bp[3 * i + j] = B[3 * i + j] + RC[3 * i + j]; }
  // Ap = A + R B^T + B R^T + R C R^T = A + R B^T + (B + R C) R^T
  double Ap[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double rbt = R[3 * i] * B[3 * j] + R[3 * i + 1] * B[3 * j + 1] + R[3 * i + 2] * B[3 * j + 2];          // (R B^T)[i][j]
      const double bprt = Bp[3 * i] * R[3 * j] + Bp[3 * i + 1] * R[3 * j + 1] + Bp[3 * i + 2] * R[3 * j + 2];     // (Bp R^T)[i][j]
      Ap[3 * i + j] = A[3 * i + j] + rbt + bprt;
    }
  Yp.A[0] += Ap[0]; Yp.A[1] += 0.5 * (Ap[1] + Ap[3]); Yp.A[2] += 0.5 * (Ap[2] + Ap[6]); Yp.A[3] += Ap[4]; Yp.A[4] += 0.5 * (Ap[5] + Ap[7]); Yp.A[5] += Ap[8];
#pragma unroll
  for (int k = 0; k < 9; ++k) Yp.B[k] += Bp[k];
  Yp.C[0] += C[0]; Yp.C[1] += 0.5 * (C[1] + C[3]); Yp.C[2] += 0.5 * (C[2] + C[6]); Yp.C[3] += C[4]; Yp.C[4] += 0.5 * (C[5] + C[7]); Yp.C[5] += C[8];
}

// ---- per-body sweeps -------------------------------------------------------------------------------
struct BodyState { double v[6], pA[6], s, c; };   // chain-local between the outward and inward sweeps

template <int I> DEVFN void body_out(const double* vp, double theta, double qd, BodyState& S) {
  constexpr int AX = C_AXIS[I];
  h1f::sincos_fast(theta, &S.s, &S.c);
  xf_motion<I>(vp, S.s, S.c, S.v);
  S.v[AX] += qd;
  double Iv[6]; inertia_mul<I>(S.v, Iv);
  crf(S.v, Iv, S.pA);
}
// inward step of hinge body I: Y = its articulated inertia (own + children), S.pA = bias (own + children).
// Writes U, 1/D, u to LDS, folds the projected inertia / bias into the parent's accumulators.
template <int I> DEVFN void body_in(Art& Y, BodyState& S, double tau, double qd, double arm_eff, const LaneLds& L, Art& Yp, double* pAp) {
  constexpr int AX = C_AXIS[I];
  double Ua[3], Ul[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { Ua[k] = Y.A[sidx(k, AX)]; Ul[k] = Y.B[3 * AX + k]; }
  const double D = Ua[AX] + arm_eff, di = 1.0 / D;
  const double u = tau - S.pA[AX];
  const int base = 8 * (I - 1);
  L[base + 0] = Ua[0]; L[base + 1] = Ua[1]; L[base + 2] = Ua[2]; L[base + 3] = Ul[0]; L[base + 4] = Ul[1]; L[base + 5] = Ul[2];
  L[base + 6] = di; L[base + 7] = u;
  // Ia = Y - U U^T / D
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = r; c < 3; ++c) { Y.A[sidx(r, c)] -= Ua[r] * Ua[c] * di; Y.C[sidx(r, c)] -= Ul[r] * Ul[c] * di; }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Y.B[3 * r + c] -= Ua[r] * Ul[c] * di;
  // velocity-product acceleration cb = v x (S qd)
  double ca[3], cl[3]; cross_axis<AX>(S.v, ca); cross_axis<AX>(S.v + 3, cl);
#pragma unroll
  for (int k = 0; k < 3; ++k) { ca[k] *= qd; cl[k] *= qd; }
  // pa = pA + Ia cb + U u / D
  double pa[6];
  const double ud = u * di;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    pa[r] = S.pA[r] + Ua[r] * ud + Y.A[sidx(r, 0)] * ca[0] + Y.A[sidx(r, 1)] * ca[1] + Y.A[sidx(r, 2)] * ca[2] + Y.B[3 * r] * cl[0] + Y.B[3 * r + 1] * cl[1] + Y.B[3 * r + 2] * cl[2];
    pa[3 + r] = S.pA[3 + r] + Ul[r] * ud + Y.B[r] * ca[0] + Y.B[3 + r] * ca[1] + Y.B[6 + r] * ca[2] + Y.C[sidx(r, 0)] * cl[0] + Y.C[sidx(r, 1)] * cl[1] + Y.C[sidx(r, 2)] * cl[2];
  }
  fold_art<I>(Y, S.s, S.c, Yp);
  xf_force_acc<I>(pa, S.s, S.c, pAp);
}
// outward acceleration step: in (vp, ap) of the parent, out (v, a) of body I and its joint acceleration
template <int I> DEVFN double body_acc(const double* vp, const double* ap, double theta, double qd, const LaneLds& L, double* v, double* a, double* sc = nullptr) {
  constexpr int AX = C_AXIS[I];
  double s, c; h1f::sincos_fast(theta, &s, &c);
  if (sc) { sc[0] = s; sc[1] = c; }
  xf_motion<I>(vp, s, c, v); v[AX] += qd;
  xf_motion<I>(ap, s, c, a);
  double ca[3], cl[3]; cross_axis<AX>(v, ca); cross_axis<AX>(v + 3, cl);
#pragma unroll
  for (int k = 0; k < 3; ++k) { a[k] += qd * ca[k]; a[3 + k] += qd * cl[k]; }
  const int base = 8 * (I - 1);
  double sum = L[base + 7];
#pragma unroll
  for (int k = 0; k < 6; ++k) sum -= L[base + k] * a[k];
  const double qdd = sum * L[base + 6];
  a[AX] += qdd;
  return qdd;
}

// ---- chains (compile-time index ranges) -------------------------------------------------------------
template <int FIRST, int LEN> struct Chain {
  // outward + inward sweep of bodies FIRST .. FIRST+LEN-1 hanging off a junction with velocity vj;
  // extra = articulated inertia / bias already accumulated at the deepest body (torso only), may be null
  static DEVFN void in(const double* vj, const double* theta, const double* qv, const double* tau, double arm_eff, const LaneLds& L,
                       const Art* extraY, const double* extraP, Art& Yj, double* pAj) {
    BodyState S[LEN];
    step_out<0>(vj, theta, qv, S);
    Art carry; double pc[6];
    step_in<LEN - 1>(theta, qv, tau, arm_eff, L, extraY, extraP, S, carry, pc, Yj, pAj);
  }
  template <int K> static DEVFN void step_out(const double* vp, const double* theta, const double* qv, BodyState* S) {
    constexpr int I = FIRST + K;
    body_out<I>(vp, theta[I - 1], qv[6 + I - 1], S[K]);
    if constexpr (K + 1 < LEN) step_out<K + 1>(S[K].v, theta, qv, S);
  }
  template <int K> static DEVFN void step_in(const double* theta, const double* qv, const double* tau, double arm_eff, const LaneLds& L,
                                             const Art* extraY, const double* extraP, BodyState* S, Art& carry, double* pc, Art& Yj, double* pAj) {
    constexpr int I = FIRST + K;
    Art Y; body_inertia<I>(Y);
    if constexpr (K == LEN - 1) {
      if (extraY) { art_add(Y, *extraY);
#pragma unroll
        for (int k = 0; k < 6; ++k) S[K].pA[k] += extraP[k]; }
    } else {
      art_add(Y, carry);
#pragma unroll
      for (int k = 0; k < 6; ++k) S[K].pA[k] += pc[k];
    }
    if constexpr (K > 0) {
      art_zero(carry);
#pragma unroll
      for (int k = 0; k < 6; ++k) pc[k] = 0.0;
      body_in<I>(Y, S[K], tau[I - 1], qv[6 + I - 1], arm_eff, L, carry, pc);
      step_in<K - 1>(theta, qv, tau, arm_eff, L, extraY, extraP, S, carry, pc, Yj, pAj);
    } else {
      body_in<I>(Y, S[K], tau[I - 1], qv[6 + I - 1], arm_eff, L, Yj, pAj);
    }
  }
  // outward acceleration sweep; sink(I, v, a, s, c) is called per body when DUMP
  template <int K, class Sink> static DEVFN void acc(const double* vp, const double* ap, const double* theta, const double* qv, const LaneLds& L, double* qacc,
                                                     double* vlast, double* alast, Sink& sink) {
    constexpr int I = FIRST + K;
    double v[6], a[6], sc[2];
    qacc[6 + I - 1] = body_acc<I>(vp, ap, theta[I - 1], qv[6 + I - 1], L, v, a, sc);
    sink(I, v, a, sc[0], sc[1]);
    if constexpr (K + 1 < LEN) acc<K + 1>(v, a, theta, qv, L, qacc, vlast, alast, sink);
    else if (vlast) {
#pragma unroll
      for (int k = 0; k < 6; ++k) { vlast[k] = v[k]; alast[k] = a[k]; }
    }
  }
};

struct NoSink { DEVFN void operator()(int, const double*, const double*, double, double) const {} };

// solve the SPD 6x6 system (blocks of Art) Y a = rhs by LDL^T; optionally return the explicit inverse
DEVFN void solve6(const Art& Y, const double* rhs, double* out, double* inv36 = nullptr) {
  double M[36];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) { M[6 * r + c] = Y.A[sidx(r, c)]; M[6 * r + 3 + c] = Y.B[3 * r + c]; M[6 * (3 + r) + c] = Y.B[3 * c + r]; M[6 * (3 + r) + 3 + c] = Y.C[sidx(r, c)]; }
  double Lm[36], d[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double s = M[6 * j + j];
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k < j) s -= Lm[6 * j + k] * Lm[6 * j + k] * d[k];
    d[j] = s;
    const double inv = 1.0 / s;
#pragma unroll
    for (int i = 0; i < 6; ++i) if (i > j) {
      double t = M[6 * i + j];
#pragma unroll
      for (int k = 0; k < 6; ++k) if (k < j) t -= Lm[6 * i + k] * Lm[6 * j + k] * d[k];
      Lm[6 * i + j] = t * inv;
    }
  }
  auto solve = [&](const double* b, double* y) {
    double z[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { double s = b[i];
#pragma unroll
      for (int k = 0; k < 6; ++k) if (k < i) s -= Lm[6 * i + k] * z[k];
      z[i] = s; }
#pragma unroll
    for (int i = 0; i < 6; ++i) z[i] = z[i] / d[i];
#pragma unroll
    for (int i = 5; i >= 0; --i) { double s = z[i];
#pragma unroll
      for (int k = 0; k < 6; ++k) if (k > i) s -= Lm[6 * k + i] * y[k];
      y[i] = s; }
  };
  solve(rhs, out);
  if (inv36) {
#pragma unroll
    for (int c = 0; c < 6; ++c) { double e[6] = {0, 0, 0, 0, 0, 0}, col[6]; e[c] = 1.0; solve(e, col);
#pragma unroll
      for (int r = 0; r < 6; ++r) inv36[6 * r + c] = col[r]; }
  }
}

// Forward dynamics in MuJoCo coordinates (same contract as h1::forward_dynamics<double>).
// R0: base rotation from the unit quaternion; theta[19]; v = qvel[25]; tau[19]; qacc[25] out.
template <class Sink>
DEVFN void forward_dynamics(const double* R0, const double* theta, const double* v, const double* tau, double arm_eff, const double* grav,
                            const LaneLds& L, double* qacc, Sink& sink, double* IA0inv = nullptr, double* aL_out = nullptr) {
  double v0[6] = {v[3], v[4], v[5], 0, 0, 0};
  v0[3] = R0[0] * v[0] + R0[3] * v[1] + R0[6] * v[2];
  v0[4] = R0[1] * v[0] + R0[4] * v[1] + R0[7] * v[2];
  v0[5] = R0[2] * v[0] + R0[5] * v[1] + R0[8] * v[2];
  // torso velocity is needed by the arms before the torso's own inward step
  BodyState T11; body_out<11>(v0, theta[10], v[6 + 10], T11);
  Art Yt; art_zero(Yt); double pt[6] = {0, 0, 0, 0, 0, 0};
  Chain<12, 4>::in(T11.v, theta, v, tau, arm_eff, L, nullptr, nullptr, Yt, pt);
#ifdef ABA_FENCE   // scheduling fence between the independent sweeps: left free, the scheduler interleaves the chains and
  __builtin_amdgcn_sched_barrier(0);   // many more temporaries live (and spill) at once
#endif
  Chain<16, 4>::in(T11.v, theta, v, tau, arm_eff, L, nullptr, nullptr, Yt, pt);
#ifdef ABA_FENCE
  __builtin_amdgcn_sched_barrier(0);
#endif
  Art Y0; body_inertia<0>(Y0);
  double p0[6];
  { double Iv[6]; inertia_mul<0>(v0, Iv); crf(v0, Iv, p0); }
  {  // torso inward
    Art Y; body_inertia<11>(Y); art_add(Y, Yt);
#pragma unroll
    for (int k = 0; k < 6; ++k) T11.pA[k] += pt[k];
    body_in<11>(Y, T11, tau[10], v[6 + 10], arm_eff, L, Y0, p0);
  }
#ifdef ABA_FENCE
  __builtin_amdgcn_sched_barrier(0);
#endif
  Chain<1, 5>::in(v0, theta, v, tau, arm_eff, L, nullptr, nullptr, Y0, p0);
#ifdef ABA_FENCE
  __builtin_amdgcn_sched_barrier(0);
#endif
  Chain<6, 5>::in(v0, theta, v, tau, arm_eff, L, nullptr, nullptr, Y0, p0);
#ifdef ABA_FENCE
  __builtin_amdgcn_sched_barrier(0);
#endif
  // pelvis
  double rhs[6] = {-p0[0], -p0[1], -p0[2], -p0[3], -p0[4], -p0[5]}, a0[6];
  solve6(Y0, rhs, a0, IA0inv);
  const double mg[3] = {-grav[0], -grav[1], -grav[2]};
  const double a0p[3] = {R0[0] * mg[0] + R0[3] * mg[1] + R0[6] * mg[2], R0[1] * mg[0] + R0[4] * mg[1] + R0[7] * mg[2], R0[2] * mg[0] + R0[5] * mg[1] + R0[8] * mg[2]};
  double wxv[3]; cross(v0, v0 + 3, wxv);
  const double lin[3] = {a0[3] - a0p[0] + wxv[0], a0[4] - a0p[1] + wxv[1], a0[5] - a0p[2] + wxv[2]};
  qacc[0] = R0[0] * lin[0] + R0[1] * lin[1] + R0[2] * lin[2];
  qacc[1] = R0[3] * lin[0] + R0[4] * lin[1] + R0[5] * lin[2];
  qacc[2] = R0[6] * lin[0] + R0[7] * lin[1] + R0[8] * lin[2];
  qacc[3] = a0[0]; qacc[4] = a0[1]; qacc[5] = a0[2];
  if (aL_out) { aL_out[0] = a0[3] + wxv[0]; aL_out[1] = a0[4] + wxv[1]; aL_out[2] = a0[5] + wxv[2]; }
  sink(0, v0, a0, 0.0, 1.0);
  // outward accelerations
  double v11[6], a11[6];
  Chain<11, 1>::template acc<0>(v0, a0, theta, v, L, qacc, v11, a11, sink);
#ifdef ABA_FENCE
  __builtin_amdgcn_sched_barrier(0);
#endif
  Chain<12, 4>::template acc<0>(v11, a11, theta, v, L, qacc, nullptr, nullptr, sink);
#ifdef ABA_FENCE
  __builtin_amdgcn_sched_barrier(0);
#endif
  Chain<16, 4>::template acc<0>(v11, a11, theta, v, L, qacc, nullptr, nullptr, sink);
#ifdef ABA_FENCE
  __builtin_amdgcn_sched_barrier(0);
#endif
  Chain<1, 5>::template acc<0>(v0, a0, theta, v, L, qacc, nullptr, nullptr, sink);
#ifdef ABA_FENCE
  __builtin_amdgcn_sched_barrier(0);
#endif
  Chain<6, 5>::template acc<0>(v0, a0, theta, v, L, qacc, nullptr, nullptr, sink);
}

DEVFN void quat_R(double w, double x, double y, double z, double* R) {
  R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z); R[2] = 2.0 * (x * z + w * y);
  R[3] = 2.0 * (x * y + w * z); R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
  R[6] = 2.0 * (x * z - w * y); R[7] = 2.0 * (y * z + w * x); R[8] = 1.0 - 2.0 * (x * x + y * y);
}

// x_next = f(x, u): RobotUtils::rolloutOneStep (reference src/common/robot_utils.cpp:106-117), smooth regime
DEVFN void step(const double* x, const double* u, double h, const double* grav, const LaneLds& L, double* xn) {
  const double qn = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  const double qh[4] = {x[3] / qn, x[4] / qn, x[5] / qn, x[6] / qn};
  double R0[9]; quat_R(qh[0], qh[1], qh[2], qh[3], R0);
  double tau[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    double ui = u[i];
    if (ui < C_CTRLRANGE[i][0]) ui = C_CTRLRANGE[i][0];
    if (ui > C_CTRLRANGE[i][1]) ui = C_CTRLRANGE[i][1];
    tau[i] = ui - DAMPING * x[NQ + 6 + i];
  }
  double qacc[NV];
  NoSink ns;
  forward_dynamics(R0, x + 7, x + NQ, tau, ARMATURE + h * DAMPING, grav, L, qacc, ns);
  double vn[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { vn[i] = x[NQ + i] + h * qacc[i]; xn[NQ + i] = vn[i]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) xn[k] = x[k] + h * vn[k];
#pragma unroll
  for (int i = 0; i < NJ; ++i) xn[7 + i] = x[7 + i] + h * vn[6 + i];
  const double s = (vn[3] * vn[3] + vn[4] * vn[4] + vn[5] * vn[5]) * (h * h);
  double c, so;
  if (s < 1e-6) { c = 1.0 - s / 8.0 + s * s / 384.0 - s * s * s / 46080.0; so = 0.5 - s / 48.0 + s * s / 3840.0 - s * s * s / 645120.0; }
  else { const double a = sqrt(s); double sn, cn; h1f::sincos_fast(0.5 * a, &sn, &cn); c = cn; so = sn / a; }
  const double ew = c, ex = so * h * vn[3], ey = so * h * vn[4], ez = so * h * vn[5];
  const double rw = qh[0] * ew - qh[1] * ex - qh[2] * ey - qh[3] * ez;
  const double rx = qh[0] * ex + qh[1] * ew + qh[2] * ez - qh[3] * ey;
  const double ry = qh[0] * ey - qh[1] * ez + qh[2] * ew + qh[3] * ex;
  const double rz = qh[0] * ez + qh[1] * ey - qh[2] * ex + qh[3] * ew;
  const double rn = sqrt(rw * rw + rx * rx + ry * ry + rz * rz);
  xn[3] = rw / rn; xn[4] = rx / rn; xn[5] = ry / rn; xn[6] = rz / rn;
}

// whole-body CoM with MuJoCo masses (RobotUtils::computeCoM, reference src/common/robot_utils.cpp:810-833)
template <int I> DEVFN void com_body(const double* Rp, const double* pp, const double* x, double* acc) {
  constexpr int a = C_AXIS[I], b = (a + 1) % 3, d = (a + 2) % 3;
  double s, c; h1f::sincos_fast(x[7 + I - 1], &s, &c);
  // R = Rp * Rfix * Rot
  double R[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double f[3];
    if constexpr (rfix_identity<I>()) { f[0] = Rp[3 * r]; f[1] = Rp[3 * r + 1]; f[2] = Rp[3 * r + 2]; }
    else {
#pragma unroll
      for (int k = 0; k < 3; ++k) f[k] = Rp[3 * r] * C_RFIX[I][0][k] + Rp[3 * r + 1] * C_RFIX[I][1][k] + Rp[3 * r + 2] * C_RFIX[I][2][k];
    }
    R[3 * r + a] = f[a]; R[3 * r + b] = f[b] * c + f[d] * s; R[3 * r + d] = f[d] * c - f[b] * s;
  }
  double p[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) p[r] = pp[r] + Rp[3 * r] * C_POS[I][0] + Rp[3 * r + 1] * C_POS[I][1] + Rp[3 * r + 2] * C_POS[I][2];
#pragma unroll
  for (int r = 0; r < 3; ++r) acc[r] += C_MASS[I] * (p[r] + R[3 * r] * C_COM[I][0] + R[3 * r + 1] * C_COM[I][1] + R[3 * r + 2] * C_COM[I][2]);
  constexpr bool leaf = (I == 5 || I == 10 || I == 15 || I == 19);
  if constexpr (I == 11) { com_body<12>(R, p, x, acc); com_body<16>(R, p, x, acc); }
  else if constexpr (!leaf) com_body<I + 1>(R, p, x, acc);
}
DEVFN void com_mj(const double* x, double* com) {
  const double qn = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  double R0[9]; quat_R(x[3] / qn, x[4] / qn, x[5] / qn, x[6] / qn, R0);
  const double p0[3] = {x[0], x[1], x[2]};
  double acc[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) acc[r] = C_MASS[0] * (p0[r] + R0[3 * r] * C_COM[0][0] + R0[3 * r + 1] * C_COM[0][1] + R0[3 * r + 2] * C_COM[0][2]);
  com_body<1>(R0, p0, x, acc); com_body<6>(R0, p0, x, acc); com_body<11>(R0, p0, x, acc);
  double mtot = 0.0;
#pragma unroll
  for (int i = 0; i < NB; ++i) mtot += C_MASS[i];
  com[0] = acc[0] / mtot; com[1] = acc[1] / mtot; com[2] = acc[2] / mtot;
}

}  // namespace h1r
