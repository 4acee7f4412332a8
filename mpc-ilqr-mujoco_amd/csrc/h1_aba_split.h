// Articulated-body forward dynamics of the H1 on TWO lanes per rollout: the even lane owns the left leg and
// the left arm, the odd lane the right ones; pelvis and torso are computed redundantly by both.
//
// Same algorithm, MuJoCo semantics and block representation as h1_aba_reg.h (one lane per rollout), but half
// the hinge state and half the serial chain length per lane, and twice the waves (the line search has only
// 8 x batch candidates: with one lane each, half of the chip's SIMDs would have no wave at all).  Left / right bodies are mirror images with identical joint axes, so both lanes run the
// same instruction stream; a body constant is `side ? right : left`, which folds to an immediate wherever the
// two values coincide (masses, x/z offsets, ...).  The two lanes meet three times per dynamics evaluation:
// the arms' articulated inertia / bias at the torso, the legs' at the pelvis (DPP quad-permute exchange + add,
// bitwise identical on both lanes), and nowhere in the outward acceleration sweep.
// U_i, 1/D_i, u_i of the lane's 10 hinges live in LDS between the sweeps (80 slots per lane).
#pragma once
#include <hip/hip_runtime.h>

#include "h1_fast_math.h"
#include "h1_model_constexpr.h"

#ifndef DEVFN
#define DEVFN __device__ __forceinline__
#endif

namespace h1s {
using namespace h1c;

constexpr int NB = 20, NJ = 19, NQ = 26, NV = 25, NX = 51, NU = 19;
constexpr double DAMPING = 1.0, ARMATURE = 0.1;
constexpr int LDS_SLOTS = 8 * 10;   // U(6), Dinv, u for torso + 5 leg + 4 arm hinges of this lane's side

struct LaneLds {           // per-lane column of the LDS scratch: slot s lives at base[s * stride + lane]
  double* base; int stride, lane;
  DEVFN double& operator[](int s) const { return base[s * stride + lane]; }
};

// value of the partner lane (lane ^ 1)
DEVFN double xch(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
DEVFN bool xch_flag(bool f) { return __builtin_amdgcn_update_dpp(0, f ? 1 : 0, 0xB1, 0xF, 0xF, true) != 0; }
DEVFN double pair_sum(double v) { return v + xch(v); }   // a + b == b + a: identical on both lanes

#define SD(IL, IR, expr_l, expr_r) ((IL) == (IR) ? (expr_l) : (side ? (expr_r) : (expr_l)))

struct Art { double A[6], B[9], C[6]; };   // [[A, B], [B^T, C]], A and C symmetric (xx, xy, xz, yy, yz, zz)
// packed symmetric index (xx, xy, xz, yy, yz, zz); not recursive on purpose (see h1_aba_reg.h)
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
#define RF(r, c) SD(IL, IR, C_RFIX[IL][r][c], C_RFIX[IR][r][c])
#define PS(k) SD(IL, IR, C_POS[IL][k], C_POS[IR][k])
#define CM(k) SD(IL, IR, C_COM[IL][k], C_COM[IR][k])
#define IN(r, c) SD(IL, IR, C_INERTIA[IL][r][c], C_INERTIA[IR][r][c])
#define MS SD(IL, IR, C_MASS[IL], C_MASS[IR])
// The model's constant matrices are full of exact zeros (axis-aligned frame rotations, offsets along one axis, diagonal
// inertia tensors) and IEEE semantics forbid the compiler to drop `0 * x` or `x + 0` (x could be NaN / Inf, the zero signed):
// 15 % of the fp64 instructions of a dynamics step carried a literal 0.  The primitives below leave those terms out at compile
// time (a term is dropped only where the constant is zero on BOTH sides of the mirror pair); what remains is evaluated in the
// same order as before.
#define ZP(k) (C_POS[IL][k] == 0.0 && C_POS[IR][k] == 0.0)
#define ZC(k) (C_COM[IL][k] == 0.0 && C_COM[IR][k] == 0.0)
#define ZR(r, c) (C_RFIX[IL][r][c] == 0.0 && C_RFIX[IR][r][c] == 0.0)
#define ZI(r, c) (C_INERTIA[IL][r][c] == 0.0 && C_INERTIA[IR][r][c] == 0.0)
// c0 x0 + c1 x1 + c2 x2 without the terms whose constant is zero
template <bool Z0, bool Z1, bool Z2> DEVFN double dot3z(double c0, double c1, double c2, double x0, double x1, double x2) {
  if constexpr (Z0 && Z1 && Z2) return 0.0;
  else if constexpr (Z1 && Z2) return c0 * x0;
  else if constexpr (Z0 && Z2) return c1 * x1;
  else if constexpr (Z0 && Z1) return c2 * x2;
  else if constexpr (Z2) return c0 * x0 + c1 * x1;
  else if constexpr (Z1) return c0 * x0 + c2 * x2;
  else if constexpr (Z0) return c1 * x1 + c2 * x2;
  else return c0 * x0 + c1 * x1 + c2 * x2;
}
// a x - b y without the products whose constant (a, b) is zero
template <bool ZA, bool ZB> DEVFN double diffz(double a, double x, double b, double y) {
  if constexpr (ZA && ZB) return 0.0;
  else if constexpr (ZB) return a * x;
  else if constexpr (ZA) return -(b * y);
  else return a * x - b * y;
}
template <bool Z> DEVFN double addz(double x, double t) { if constexpr (Z) return x; else return x + t; }
template <bool Z> DEVFN double subz(double x, double t) { if constexpr (Z) return x; else return x - t; }
// o = r x v and o = v x r for a constant r whose zero components are known (ZX, ZY, ZZ); component k of either is identically
// zero when the two OTHER components of r are
template <bool ZX, bool ZY, bool ZZ> DEVFN void cross_cv(double rx, double ry, double rz, const double* v, double* o) {
  const double o0 = diffz<ZY, ZZ>(ry, v[2], rz, v[1]), o1 = diffz<ZZ, ZX>(rz, v[0], rx, v[2]), o2 = diffz<ZX, ZY>(rx, v[1], ry, v[0]);
  o[0] = o0; o[1] = o1; o[2] = o2;
}
template <bool ZX, bool ZY, bool ZZ> DEVFN void cross_vc(double rx, double ry, double rz, const double* v, double* o) {
  const double o0 = diffz<ZZ, ZY>(rz, v[1], ry, v[2]), o1 = diffz<ZX, ZZ>(rx, v[2], rz, v[0]), o2 = diffz<ZY, ZX>(ry, v[0], rx, v[1]);
  o[0] = o0; o[1] = o1; o[2] = o2;
}

// y = Rj x with Rj = Rfix * Rot(AX, theta) (child -> parent coordinates)
template <int IL, int IR> DEVFN void rot(bool side, const double* x, double s, double c, double* y) {
  static_assert(C_AXIS[IL] == C_AXIS[IR] && rfix_identity<IL>() == rfix_identity<IR>(), "mirror bodies");
  constexpr int a = C_AXIS[IL], b = (a + 1) % 3, d = (a + 2) % 3;
  double t[3];
  t[a] = x[a]; t[b] = c * x[b] - s * x[d]; t[d] = s * x[b] + c * x[d];
  if constexpr (rfix_identity<IL>()) { y[0] = t[0]; y[1] = t[1]; y[2] = t[2]; }
  else {
    y[0] = dot3z<ZR(0, 0), ZR(0, 1), ZR(0, 2)>(RF(0, 0), RF(0, 1), RF(0, 2), t[0], t[1], t[2]);
    y[1] = dot3z<ZR(1, 0), ZR(1, 1), ZR(1, 2)>(RF(1, 0), RF(1, 1), RF(1, 2), t[0], t[1], t[2]);
    y[2] = dot3z<ZR(2, 0), ZR(2, 1), ZR(2, 2)>(RF(2, 0), RF(2, 1), RF(2, 2), t[0], t[1], t[2]);
  }
}
// y = Rj^T x
template <int IL, int IR> DEVFN void rotT(bool side, const double* x, double s, double c, double* y) {
  constexpr int a = C_AXIS[IL], b = (a + 1) % 3, d = (a + 2) % 3;
  double t[3];
  if constexpr (rfix_identity<IL>()) { t[0] = x[0]; t[1] = x[1]; t[2] = x[2]; }
  else {
    t[0] = dot3z<ZR(0, 0), ZR(1, 0), ZR(2, 0)>(RF(0, 0), RF(1, 0), RF(2, 0), x[0], x[1], x[2]);
    t[1] = dot3z<ZR(0, 1), ZR(1, 1), ZR(2, 1)>(RF(0, 1), RF(1, 1), RF(2, 1), x[0], x[1], x[2]);
    t[2] = dot3z<ZR(0, 2), ZR(1, 2), ZR(2, 2)>(RF(0, 2), RF(1, 2), RF(2, 2), x[0], x[1], x[2]);
  }
  y[a] = t[a]; y[b] = c * t[b] + s * t[d]; y[d] = -s * t[b] + c * t[d];
}
// motion transform parent -> child
template <int IL, int IR> DEVFN void xf_motion(bool side, const double* vp, double s, double c, double* vc) {
  double t[3]; cross_vc<ZP(0), ZP(1), ZP(2)>(PS(0), PS(1), PS(2), vp, t);
  const double lin[3] = {addz<ZP(1) && ZP(2)>(vp[3], t[0]), addz<ZP(2) && ZP(0)>(vp[4], t[1]), addz<ZP(0) && ZP(1)>(vp[5], t[2])};
  rotT<IL, IR>(side, vp, s, c, vc); rotT<IL, IR>(side, lin, s, c, vc + 3);
}
// inverse motion transform child -> parent: vp = X^-1 vc
template <int IL, int IR> DEVFN void xf_motion_inv(bool side, const double* vc, double s, double c, double* vp) {
  double lin[3], t[3];
  rot<IL, IR>(side, vc, s, c, vp); rot<IL, IR>(side, vc + 3, s, c, lin);
  cross_vc<ZP(0), ZP(1), ZP(2)>(PS(0), PS(1), PS(2), vp, t);
  vp[3] = subz<ZP(1) && ZP(2)>(lin[0], t[0]); vp[4] = subz<ZP(2) && ZP(0)>(lin[1], t[1]); vp[5] = subz<ZP(0) && ZP(1)>(lin[2], t[2]);
}
// force transform child -> parent, accumulating (ACC) or assigning (the target is known to be zero)
template <int IL, int IR, bool ACC = true> DEVFN void xf_force_acc(bool side, const double* fc, double s, double c, double* fp) {
  double n[3], f[3], rf[3]; rot<IL, IR>(side, fc, s, c, n); rot<IL, IR>(side, fc + 3, s, c, f);
  cross_cv<ZP(0), ZP(1), ZP(2)>(PS(0), PS(1), PS(2), f, rf);
  const double m0 = addz<ZP(1) && ZP(2)>(n[0], rf[0]), m1 = addz<ZP(2) && ZP(0)>(n[1], rf[1]), m2 = addz<ZP(0) && ZP(1)>(n[2], rf[2]);
  if constexpr (ACC) { fp[0] += m0; fp[1] += m1; fp[2] += m2; fp[3] += f[0]; fp[4] += f[1]; fp[5] += f[2]; }
  else { fp[0] = m0; fp[1] = m1; fp[2] = m2; fp[3] = f[0]; fp[4] = f[1]; fp[5] = f[2]; }
}
// spatial inertia (about the body frame origin) times a motion vector
template <int IL, int IR> DEVFN void inertia_mul(bool side, const double* a, double* f) {
  const double m = MS;
  const double Iw0 = dot3z<ZI(0, 0), ZI(0, 1), ZI(0, 2)>(IN(0, 0), IN(0, 1), IN(0, 2), a[0], a[1], a[2]);
  const double Iw1 = dot3z<ZI(1, 0), ZI(1, 1), ZI(1, 2)>(IN(1, 0), IN(1, 1), IN(1, 2), a[0], a[1], a[2]);
  const double Iw2 = dot3z<ZI(2, 0), ZI(2, 1), ZI(2, 2)>(IN(2, 0), IN(2, 1), IN(2, 2), a[0], a[1], a[2]);
  double wc[3]; cross_vc<ZC(0), ZC(1), ZC(2)>(CM(0), CM(1), CM(2), a, wc);            // w x c
  const double fl[3] = {m * addz<ZC(1) && ZC(2)>(a[3], wc[0]), m * addz<ZC(2) && ZC(0)>(a[4], wc[1]), m * addz<ZC(0) && ZC(1)>(a[5], wc[2])};
  double cf[3]; cross_cv<ZC(0), ZC(1), ZC(2)>(CM(0), CM(1), CM(2), fl, cf);           // c x f_lin
  f[0] = addz<ZC(1) && ZC(2)>(Iw0, cf[0]);
  f[1] = addz<ZC(2) && ZC(0)>(Iw1, cf[1]);
  f[2] = addz<ZC(0) && ZC(1)>(Iw2, cf[2]);
  f[3] = fl[0]; f[4] = fl[1]; f[5] = fl[2];
}
// v x* f
DEVFN void crf(const double* v, const double* f, double* out) {
  double a[3], b[3], c[3]; cross(v, f, a); cross(v + 3, f + 3, b); cross(v, f + 3, c);
  out[0] = a[0] + b[0]; out[1] = a[1] + b[1]; out[2] = a[2] + b[2]; out[3] = c[0]; out[4] = c[1]; out[5] = c[2];
}
// rigid-body inertia as blocks
template <int IL, int IR> DEVFN void body_inertia(bool side, Art& Y) {
  const double m = MS;
  const double cx = CM(0), cy = CM(1), cz = CM(2), cc = cx * cx + cy * cy + cz * cz;
  Y.A[0] = IN(0, 0) + m * (cc - cx * cx); Y.A[1] = IN(0, 1) - m * cx * cy; Y.A[2] = IN(0, 2) - m * cx * cz;
  Y.A[3] = IN(1, 1) + m * (cc - cy * cy); Y.A[4] = IN(1, 2) - m * cy * cz; Y.A[5] = IN(2, 2) + m * (cc - cz * cz);
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
DEVFN void art_pair_sum(Art& Y) {
#pragma unroll
  for (int k = 0; k < 6; ++k) { Y.A[k] = pair_sum(Y.A[k]); Y.C[k] = pair_sum(Y.C[k]); }
#pragma unroll
  for (int k = 0; k < 9; ++k) Y.B[k] = pair_sum(Y.B[k]);
}
// M (general 3x3, row-major) <- Rj M Rj^T
template <int IL, int IR> DEVFN void rot_congruence(bool side, double* M, double s, double c) {
  double T[9];
#pragma unroll
  for (int col = 0; col < 3; ++col) { const double x[3] = {M[col], M[3 + col], M[6 + col]}; double y[3]; rot<IL, IR>(side, x, s, c, y); T[col] = y[0]; T[3 + col] = y[1]; T[6 + col] = y[2]; }
#pragma unroll
  for (int row = 0; row < 3; ++row) { double y[3]; rot<IL, IR>(side, T + 3 * row, s, c, y); M[3 * row] = y[0]; M[3 * row + 1] = y[1]; M[3 * row + 2] = y[2]; }
}
DEVFN void sym_to_full(const double* S, double* M) { M[0] = S[0]; M[1] = S[1]; M[2] = S[2]; M[3] = S[1]; M[4] = S[3]; M[5] = S[4]; M[6] = S[2]; M[7] = S[4]; M[8] = S[5]; }
// Yp (+)= X^T Ya X for the joint transform (rotation Rj, then translation by r = pos); ACC = false assigns (Yp known to be zero).
// With [r]x the cross-product matrix: C' = C, B' = B + [r]x C, A' = A + [r]x B^T + B' [r]x^T -- column j of [r]x C is r x C[:, j],
// column j of [r]x B^T is r x (row j of B), row i of B' [r]x^T is r x (row i of B'); components of r that are zero drop out.
template <int IL, int IR, bool ACC = true> DEVFN void fold_art(bool side, const Art& Ya, double s, double c, Art& Yp) {
  double A[9], B[9], C[9];
  sym_to_full(Ya.A, A); sym_to_full(Ya.C, C);
#pragma unroll
  for (int k = 0; k < 9; ++k) B[k] = Ya.B[k];
  rot_congruence<IL, IR>(side, A, s, c); rot_congruence<IL, IR>(side, B, s, c); rot_congruence<IL, IR>(side, C, s, c);
  constexpr bool zx = ZP(0), zy = ZP(1), zz = ZP(2);
  constexpr bool z0 = zy && zz, z1 = zz && zx, z2 = zx && zy;      // component k of r x (.) vanishes identically
  const double rx = PS(0), ry = PS(1), rz = PS(2);
  double Bp[9], Ap[9];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double cj[3] = {C[j], C[3 + j], C[6 + j]};
    double rc[3]; cross_cv<zx, zy, zz>(rx, ry, rz, cj, rc);
    Bp[j] = addz<z0>(B[j], rc[0]); Bp[3 + j] = addz<z1>(B[3 + j], rc[1]); Bp[6 + j] = addz<z2>(B[6 + j], rc[2]);
  }
  double rbt[9], bprt[9];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double o[3]; cross_cv<zx, zy, zz>(rx, ry, rz, B + 3 * j, o);      // column j of [r]x B^T
    rbt[j] = o[0]; rbt[3 + j] = o[1]; rbt[6 + j] = o[2];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) cross_cv<zx, zy, zz>(rx, ry, rz, Bp + 3 * i, bprt + 3 * i);   // row i of B' [r]x^T
  // Ap[i][j] = A[i][j] + rbt[i][j] + bprt[i][j]; rbt[i][j] vanishes with z_i, bprt[i][j] with z_j
  Ap[0] = addz<z0>(addz<z0>(A[0], rbt[0]), bprt[0]); Ap[1] = addz<z1>(addz<z0>(A[1], rbt[1]), bprt[1]); Ap[2] = addz<z2>(addz<z0>(A[2], rbt[2]), bprt[2]);
  Ap[3] = addz<z0>(addz<z1>(A[3], rbt[3]), bprt[3]); Ap[4] = addz<z1>(addz<z1>(A[4], rbt[4]), bprt[4]); Ap[5] = addz<z2>(addz<z1>(A[5], rbt[5]), bprt[5]);
  Ap[6] = addz<z0>(addz<z2>(A[6], rbt[6]), bprt[6]); Ap[7] = addz<z1>(addz<z2>(A[7], rbt[7]), bprt[7]); Ap[8] = addz<z2>(addz<z2>(A[8], rbt[8]), bprt[8]);
  const double a1 = 0.5 * (Ap[1] + Ap[3]), a2 = 0.5 * (Ap[2] + Ap[6]), a4 = 0.5 * (Ap[5] + Ap[7]);
  const double c1 = 0.5 * (C[1] + C[3]), c2 = 0.5 * (C[2] + C[6]), c4 = 0.5 * (C[5] + C[7]);
  if constexpr (ACC) {
    Yp.A[0] += Ap[0]; Yp.A[1] += a1; Yp.A[2] += a2; Yp.A[3] += Ap[4]; Yp.A[4] += a4; Yp.A[5] += Ap[8];
#pragma unroll
    for (int k = 0; k < 9; ++k) Yp.B[k] += Bp[k];
    Yp.C[0] += C[0]; Yp.C[1] += c1; Yp.C[2] += c2; Yp.C[3] += C[4]; Yp.C[4] += c4; Yp.C[5] += C[8];
  } else {
    Yp.A[0] = Ap[0]; Yp.A[1] = a1; Yp.A[2] = a2; Yp.A[3] = Ap[4]; Yp.A[4] = a4; Yp.A[5] = Ap[8];
#pragma unroll
    for (int k = 0; k < 9; ++k) Yp.B[k] = Bp[k];
    Yp.C[0] = C[0]; Yp.C[1] = c1; Yp.C[2] = c2; Yp.C[3] = C[4]; Yp.C[4] = c4; Yp.C[5] = C[8];
  }
}

// ---- per-body sweeps -------------------------------------------------------------------------------
struct BodyState { double v[6], pA[6], s, c; };   // chain-local between the outward and inward sweeps

template <int IL, int IR> DEVFN void body_out(bool side, const double* vp, double theta, double qd, BodyState& S) {
  constexpr int AX = C_AXIS[IL];
  h1f::sincos_fast(theta, &S.s, &S.c);
  xf_motion<IL, IR>(side, vp, S.s, S.c, S.v);
  S.v[AX] += qd;
  double Iv[6]; inertia_mul<IL, IR>(side, S.v, Iv);
  crf(S.v, Iv, S.pA);
}
// inward step of a hinge body: Y = its articulated inertia (own + children), S.pA = bias (own + children).
// Writes U, 1/D, u to LDS slot block `slot`, folds the projected inertia / bias into the parent's accumulators.
template <int IL, int IR, bool ACC = true> DEVFN void body_in(bool side, Art& Y, BodyState& S, double tau, double qd, double arm_eff, const LaneLds& L, int slot, Art& Yp, double* pAp) {
  constexpr int AX = C_AXIS[IL];
  double Ua[3], Ul[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { Ua[k] = Y.A[sidx(k, AX)]; Ul[k] = Y.B[3 * AX + k]; }
  const double D = Ua[AX] + arm_eff, di = 1.0 / D;
  const double u = tau - S.pA[AX];
  L[slot + 0] = Ua[0]; L[slot + 1] = Ua[1]; L[slot + 2] = Ua[2]; L[slot + 3] = Ul[0]; L[slot + 4] = Ul[1]; L[slot + 5] = Ul[2];
  L[slot + 6] = di; L[slot + 7] = u;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = r; c < 3; ++c) { Y.A[sidx(r, c)] -= Ua[r] * Ua[c] * di; Y.C[sidx(r, c)] -= Ul[r] * Ul[c] * di; }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Y.B[3 * r + c] -= Ua[r] * Ul[c] * di;
  double ca[3], cl[3]; cross_axis<AX>(S.v, ca); cross_axis<AX>(S.v + 3, cl);
#pragma unroll
  for (int k = 0; k < 3; ++k) { ca[k] *= qd; cl[k] *= qd; }
  double pa[6];
  const double ud = u * di;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    pa[r] = S.pA[r] + Ua[r] * ud + Y.A[sidx(r, 0)] * ca[0] + Y.A[sidx(r, 1)] * ca[1] + Y.A[sidx(r, 2)] * ca[2] + Y.B[3 * r] * cl[0] + Y.B[3 * r + 1] * cl[1] + Y.B[3 * r + 2] * cl[2];
    pa[3 + r] = S.pA[3 + r] + Ul[r] * ud + Y.B[r] * ca[0] + Y.B[3 + r] * ca[1] + Y.B[6 + r] * ca[2] + Y.C[sidx(r, 0)] * cl[0] + Y.C[sidx(r, 1)] * cl[1] + Y.C[sidx(r, 2)] * cl[2];
  }
  fold_art<IL, IR, ACC>(side, Y, S.s, S.c, Yp);
  xf_force_acc<IL, IR, ACC>(side, pa, S.s, S.c, pAp);
}
// outward acceleration step: in (vp, ap) of the parent, out (v, a) of the body and its joint acceleration
template <int IL, int IR> DEVFN double body_acc(bool side, const double* vp, const double* ap, double theta, double qd, const LaneLds& L, int slot, double* v, double* a,
                                                double* sc = nullptr) {
  constexpr int AX = C_AXIS[IL];
  double s, c; h1f::sincos_fast(theta, &s, &c);
  if (sc) { sc[0] = s; sc[1] = c; }
  xf_motion<IL, IR>(side, vp, s, c, v); v[AX] += qd;
  xf_motion<IL, IR>(side, ap, s, c, a);
  double ca[3], cl[3]; cross_axis<AX>(v, ca); cross_axis<AX>(v + 3, cl);
#pragma unroll
  for (int k = 0; k < 3; ++k) { a[k] += qd * ca[k]; a[3 + k] += qd * cl[k]; }
  double sum = L[slot + 7];
#pragma unroll
  for (int k = 0; k < 6; ++k) sum -= L[slot + k] * a[k];
  const double qdd = sum * L[slot + 6];
  a[AX] += qdd;
  return qdd;
}

struct ArmPer { double base; const double* add; };      // effective armature of hinge K of a chain: base + add[K]
DEVFN double arm_at(double a, int) { return a; }
DEVFN double arm_at(const ArmPer& a, int k) { return a.base + a.add[k]; }
// ---- mirrored chains: bodies FL.. (even lane) / FR.. (odd lane), LEN hinges, LDS slot blocks from SLOT0 ------
// Register-lean sweeps: the outward pass keeps only the running velocity and the joint sines / cosines; the inward
// pass recovers each parent velocity with the inverse joint transform (v_parent = X^-1 (v - S qd)) and recomputes
// the velocity-product force from it, instead of holding 12 numbers per body across the two passes.
template <int FL, int FR, int LEN, int SLOT0> struct Chain {
  // th, qd, tau: this lane's LEN hinge values.  Yj / pAj accumulate this lane's chain only.
  // ARM: double (the same effective armature on every hinge) or ArmPer (joint-limit rows, step_stance<., true>: per-hinge armature on top)
  template <class ARM>
  static DEVFN void in(bool side, const double* vj, const double* th, const double* qd, const double* tau, ARM arm_eff, const LaneLds& L,
                       Art& Yj, double* pAj) {
    double sn[LEN], cs[LEN], v[6];
    step_out<0>(side, vj, th, qd, sn, cs, v);          // v = velocity of the chain's last body
    Art carry; double pc[6];
    step_in<LEN - 1>(side, qd, tau, arm_eff, L, sn, cs, v, carry, pc, Yj, pAj);
  }
  template <int K> static DEVFN void step_out(bool side, const double* vp, const double* th, const double* qd, double* sn, double* cs, double* vlast) {
    constexpr int AX = C_AXIS[FL + K];
    h1f::sincos_fast(th[K], &sn[K], &cs[K]);
    double v[6];
    xf_motion<FL + K, FR + K>(side, vp, sn[K], cs[K], v);
    v[AX] += qd[K];
    if constexpr (K + 1 < LEN) step_out<K + 1>(side, v, th, qd, sn, cs, vlast);
    else {
#pragma unroll
      for (int k = 0; k < 6; ++k) vlast[k] = v[k];
    }
  }
  template <int K, class ARM> static DEVFN void step_in(bool side, const double* qd, const double* tau, ARM arm_eff, const LaneLds& L,
                                             const double* sn, const double* cs, double* v, Art& carry, double* pc, Art& Yj, double* pAj) {
    constexpr int AX = C_AXIS[FL + K];
    BodyState S;
#pragma unroll
    for (int k = 0; k < 6; ++k) S.v[k] = v[k];
    S.s = sn[K]; S.c = cs[K];
    { double Iv[6]; inertia_mul<FL + K, FR + K>(side, S.v, Iv); crf(S.v, Iv, S.pA); }
    Art Y; body_inertia<FL + K, FR + K>(side, Y);
    if constexpr (K < LEN - 1) {
      art_add(Y, carry);
#pragma unroll
      for (int k = 0; k < 6; ++k) S.pA[k] += pc[k];
    }
    if constexpr (K > 0) {
      body_in<FL + K, FR + K, false>(side, Y, S, tau[K], qd[K], arm_at(arm_eff, K), L, SLOT0 + 8 * K, carry, pc);     // carry, pc <- (assigned)
      // parent's velocity for the next inward step
      double vc[6] = {v[0], v[1], v[2], v[3], v[4], v[5]};
      vc[AX] -= qd[K];
      xf_motion_inv<FL + K, FR + K>(side, vc, sn[K], cs[K], v);
      step_in<K - 1>(side, qd, tau, arm_eff, L, sn, cs, v, carry, pc, Yj, pAj);
    } else {
      body_in<FL + K, FR + K>(side, Y, S, tau[K], qd[K], arm_at(arm_eff, K), L, SLOT0 + 8 * K, Yj, pAj);
    }
  }
  // the same sweep reporting every body: sink(body index of this lane's side, v, a, sin, cos)
  template <int K, class Sink> static DEVFN void acc_dump(bool side, const double* vp, const double* ap, const double* th, const double* qd, const LaneLds& L, double* qdd, Sink& sink) {
    double v[6], a[6], sc[2];
    qdd[K] = body_acc<FL + K, FR + K>(side, vp, ap, th[K], qd[K], L, SLOT0 + 8 * K, v, a, sc);
    sink(side ? FR + K : FL + K, v, a, sc[0], sc[1]);
    if constexpr (K + 1 < LEN) acc_dump<K + 1>(side, v, a, th, qd, L, qdd, sink);
  }
  template <int K> static DEVFN void acc(bool side, const double* vp, const double* ap, const double* th, const double* qd, const LaneLds& L, double* qdd) {
    double v[6], a[6];
    qdd[K] = body_acc<FL + K, FR + K>(side, vp, ap, th[K], qd[K], L, SLOT0 + 8 * K, v, a);
    if constexpr (K + 1 < LEN) acc<K + 1>(side, v, a, th, qd, L, qdd);
  }
};

// solve the SPD 6x6 system (blocks of Art) Y a = rhs by LDL^T
DEVFN void solve6(const Art& Y, const double* rhs, double* out) {
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
  double z[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { double s = rhs[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k < i) s -= Lm[6 * i + k] * z[k];
    z[i] = s; }
#pragma unroll
  for (int i = 0; i < 6; ++i) z[i] = z[i] / d[i];
#pragma unroll
  for (int i = 5; i >= 0; --i) { double s = z[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k > i) s -= Lm[6 * k + i] * out[k];
    out[i] = s; }
}

// this lane's share of the hinge state: torso + own leg (5) + own arm (4)
struct HalfState {
  double th11, qd11, thL[5], qdL[5], thA[4], qdA[4];
};
struct HalfTau { double t11, tL[5], tA[4]; };
struct HalfAcc { double q11, qL[5], qA[4]; };

typedef Chain<1, 6, 5, 8> LegChain;      // bodies 1..5 / 6..10, LDS slots 8..47
typedef Chain<12, 16, 4, 48> ArmChain;   // bodies 12..15 / 16..19, LDS slots 48..79

// Forward dynamics in MuJoCo coordinates.  R0: base rotation from the unit quaternion; vb = qvel[0..5];
// qbase[6] (identical on both lanes) and this lane's hinge accelerations out.
// arm_add: per-hinge armature on top of arm_eff (HalfTau-shaped), null for none.  A hinge with armature 2^1000 and torque c 2^1000 IS an
// acceleration-prescribed joint, qacc = c, exactly: D = 2^1000, 1 / D = 2^-1000 and u / D = c to the last bit, the reduction of the
// articulated inertia U U^T / D vanishes below rounding -- Featherstone's hybrid dynamics through the unmodified recursion, and every
// later reader of U_i, 1 / D_i (the stance rows' unit-wrench responses, the linearisation's Minv columns) sees that hinge locked.
template <bool PER = false>
DEVFN void forward_dynamics(bool side, const double* R0, const double* vb, const HalfState& q, const HalfTau& tau, double arm_eff,
                            const double* grav, const LaneLds& L, double* qbase, HalfAcc& qacc, Art* Y0_out = nullptr, double* a0_out = nullptr,
                            const HalfTau* arm_add = nullptr) {
  double v0[6] = {vb[3], vb[4], vb[5], 0, 0, 0};
  v0[3] = R0[0] * vb[0] + R0[3] * vb[1] + R0[6] * vb[2];
  v0[4] = R0[1] * vb[0] + R0[4] * vb[1] + R0[7] * vb[2];
  v0[5] = R0[2] * vb[0] + R0[5] * vb[1] + R0[8] * vb[2];
  // torso velocity is needed by the arm before the torso's own inward step
  BodyState T11; body_out<11, 11>(side, v0, q.th11, q.qd11, T11);
  Art Y0; double p0[6];
  {
    Art Yt; art_zero(Yt); double pt[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (PER) ArmChain::in(side, T11.v, q.thA, q.qdA, tau.tA, ArmPer{arm_eff, arm_add->tA}, L, Yt, pt);
    else ArmChain::in(side, T11.v, q.thA, q.qdA, tau.tA, arm_eff, L, Yt, pt);
    art_pair_sum(Yt);                         // left + right arm
#pragma unroll
    for (int k = 0; k < 6; ++k) pt[k] = pair_sum(pt[k]);
    Art Y; body_inertia<11, 11>(side, Y); art_add(Y, Yt);
#pragma unroll
    for (int k = 0; k < 6; ++k) T11.pA[k] += pt[k];
    art_zero(Y0);
#pragma unroll
    for (int k = 0; k < 6; ++k) p0[k] = 0.0;
    body_in<11, 11>(side, Y, T11, tau.t11, q.qd11, PER ? arm_eff + arm_add->t11 : arm_eff, L, 0, Y0, p0);   // torso's share of the pelvis (both lanes)
  }
#ifdef ABA_FENCE   // scheduling fence between the independent sweeps (arms | legs | pelvis solve | outward passes): left free, the
  __builtin_amdgcn_sched_barrier(0);   // scheduler interleaves them and a third more temporaries live (and spill) at once
#endif
  {
    Art Yl; art_zero(Yl); double pl[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (PER) LegChain::in(side, v0, q.thL, q.qdL, tau.tL, ArmPer{arm_eff, arm_add->tL}, L, Yl, pl);
    else LegChain::in(side, v0, q.thL, q.qdL, tau.tL, arm_eff, L, Yl, pl);
    art_pair_sum(Yl);                         // left + right leg
#pragma unroll
    for (int k = 0; k < 6; ++k) pl[k] = pair_sum(pl[k]);
    art_add(Y0, Yl);
#pragma unroll
    for (int k = 0; k < 6; ++k) p0[k] += pl[k];
    Art Yb; body_inertia<0, 0>(side, Yb); art_add(Y0, Yb);
    double Iv[6], pv[6]; inertia_mul<0, 0>(side, v0, Iv); crf(v0, Iv, pv);
#pragma unroll
    for (int k = 0; k < 6; ++k) p0[k] += pv[k];
  }
#ifdef ABA_FENCE
  __builtin_amdgcn_sched_barrier(0);
#endif
  // pelvis
  double rhs[6] = {-p0[0], -p0[1], -p0[2], -p0[3], -p0[4], -p0[5]}, a0[6];
  solve6(Y0, rhs, a0);
  if (Y0_out) *Y0_out = Y0;
  if (a0_out) {
#pragma unroll
    for (int k = 0; k < 6; ++k) a0_out[k] = a0[k];
  }
  const double mg[3] = {-grav[0], -grav[1], -grav[2]};
  const double a0p[3] = {R0[0] * mg[0] + R0[3] * mg[1] + R0[6] * mg[2], R0[1] * mg[0] + R0[4] * mg[1] + R0[7] * mg[2], R0[2] * mg[0] + R0[5] * mg[1] + R0[8] * mg[2]};
  double wxv[3]; cross(v0, v0 + 3, wxv);
  const double lin[3] = {a0[3] - a0p[0] + wxv[0], a0[4] - a0p[1] + wxv[1], a0[5] - a0p[2] + wxv[2]};
  qbase[0] = R0[0] * lin[0] + R0[1] * lin[1] + R0[2] * lin[2];
  qbase[1] = R0[3] * lin[0] + R0[4] * lin[1] + R0[5] * lin[2];
  qbase[2] = R0[6] * lin[0] + R0[7] * lin[1] + R0[8] * lin[2];
  qbase[3] = a0[0]; qbase[4] = a0[1]; qbase[5] = a0[2];
  // outward accelerations
  double v11[6], a11[6];
  qacc.q11 = body_acc<11, 11>(side, v0, a0, q.th11, q.qd11, L, 0, v11, a11);
  ArmChain::acc<0>(side, v11, a11, q.thA, q.qdA, L, qacc.qA);
#ifdef ABA_FENCE
  __builtin_amdgcn_sched_barrier(0);
#endif
  LegChain::acc<0>(side, v0, a0, q.thL, q.qdL, L, qacc.qL);
}

DEVFN void quat_R(double w, double x, double y, double z, double* R) {
  R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z); R[2] = 2.0 * (x * z + w * y);
  R[3] = 2.0 * (x * y + w * z); R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
  R[6] = 2.0 * (x * z - w * y); R[7] = 2.0 * (y * z + w * x); R[8] = 1.0 - 2.0 * (x * x + y * y);
}

// ---- schedule-driven rigid stance constraints on two lanes per rollout -------------------------------
// Same algorithm as h1_dynamics_dev.h forward_dynamics_stance / oracle forward_dynamics_mj_stance (reference plant:
// RobotUtils::rolloutOneStep with MuJoCo's floor contacts, robot_utils.cpp:106-117): velocity-level constraint v_f + h a_f = 0
// on the ankle link of every scheduled stance foot, a_f = a_f,free + C lambda, C = J Mhat^-1 J^T built by propagating unit
// wrenches through the articulated-body quantities of the free solve (U_i, 1/D_i in LDS, the pelvis inverse).  Each lane
// owns its side's foot: it propagates the six unit wrenches of its own foot inward, the pelvis shares are exchanged with the
// partner (DPP), and every lane sweeps its own leg outward for all twelve columns -- i.e. it computes the six rows of C of
// its own foot; the two 6 x 12 row blocks are exchanged, so both lanes factor the same 12 x 12 matrix and obtain bitwise
// identical multipliers.  Feet that are not in stance keep their rows / columns as identity (no lane-divergent sizes).
struct LegTrig { double sn[5], cs[5]; };
// LDL^T of the (pair-identical) pelvis articulated inertia, kept for the 13 pelvis solves of a constrained step
struct Ldl6 { double Lm[15], d[6]; };   // strict lower triangle row by row: (1,0) (2,0) (2,1) (3,0) ...
DEVFN constexpr int lidx(int i, int j) { return i * (i - 1) / 2 + j; }
DEVFN void ldl6_factor(const Art& Y, Ldl6& F) {
  double M[36];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) { M[6 * r + c] = Y.A[sidx(r, c)]; M[6 * r + 3 + c] = Y.B[3 * r + c]; M[6 * (3 + r) + c] = Y.B[3 * c + r]; M[6 * (3 + r) + 3 + c] = Y.C[sidx(r, c)]; }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double s = M[6 * j + j];
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k < j) s -= F.Lm[lidx(j, k)] * F.Lm[lidx(j, k)] * F.d[k];
    F.d[j] = s;
    const double inv = 1.0 / s;
#pragma unroll
    for (int i = 0; i < 6; ++i) if (i > j) {
      double t = M[6 * i + j];
#pragma unroll
      for (int k = 0; k < 6; ++k) if (k < j) t -= F.Lm[lidx(i, k)] * F.Lm[lidx(j, k)] * F.d[k];
      F.Lm[lidx(i, j)] = t * inv;
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) F.d[j] = 1.0 / F.d[j];
}
DEVFN void ldl6_solve(const Ldl6& F, const double* rhs, double* out) {
  double z[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { double s = rhs[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k < i) s -= F.Lm[lidx(i, k)] * z[k];
    z[i] = s; }
#pragma unroll
  for (int i = 0; i < 6; ++i) z[i] *= F.d[i];
#pragma unroll
  for (int i = 5; i >= 0; --i) { double s = z[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k > i) s -= F.Lm[lidx(k, i)] * out[k];
    out[i] = s; }
}

// The same forward dynamics reporting what the analytic linearisation needs of the nominal knot (LinDumpG, h1_linearize_dev.h):
// sink(body, v, a, sin, cos) for every body of this lane's side (pelvis and torso: both lanes report them), the explicit
// inverse of the pelvis' articulated inertia and the pelvis' linear acceleration without the gravity term.  U, 1/D stay in
// this lane's LDS slots (torso: slot block 0, leg hinge K: 8 + 8 K, arm hinge K: 48 + 8 K).
template <class Sink, bool PER = false>
DEVFN void forward_dynamics_dump(bool side, const double* R0, const double* vb, const HalfState& q, const HalfTau& tau, double arm_eff, const double* grav,
                                 const LaneLds& L, double* qbase, HalfAcc& qacc, Sink& sink, double* inv36, double* aL, const HalfTau* arm_add = nullptr) {
  double v0[6] = {vb[3], vb[4], vb[5], 0, 0, 0};
  v0[3] = R0[0] * vb[0] + R0[3] * vb[1] + R0[6] * vb[2];
  v0[4] = R0[1] * vb[0] + R0[4] * vb[1] + R0[7] * vb[2];
  v0[5] = R0[2] * vb[0] + R0[5] * vb[1] + R0[8] * vb[2];
  BodyState T11; body_out<11, 11>(side, v0, q.th11, q.qd11, T11);
  Art Y0; double p0[6];
  {
    Art Yt; art_zero(Yt); double pt[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (PER) ArmChain::in(side, T11.v, q.thA, q.qdA, tau.tA, ArmPer{arm_eff, arm_add->tA}, L, Yt, pt);
    else ArmChain::in(side, T11.v, q.thA, q.qdA, tau.tA, arm_eff, L, Yt, pt);
    art_pair_sum(Yt);
#pragma unroll
    for (int k = 0; k < 6; ++k) pt[k] = pair_sum(pt[k]);
    Art Y; body_inertia<11, 11>(side, Y); art_add(Y, Yt);
#pragma unroll
    for (int k = 0; k < 6; ++k) T11.pA[k] += pt[k];
    art_zero(Y0);
#pragma unroll
    for (int k = 0; k < 6; ++k) p0[k] = 0.0;
    body_in<11, 11>(side, Y, T11, tau.t11, q.qd11, PER ? arm_eff + arm_add->t11 : arm_eff, L, 0, Y0, p0);
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    Art Yl; art_zero(Yl); double pl[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (PER) LegChain::in(side, v0, q.thL, q.qdL, tau.tL, ArmPer{arm_eff, arm_add->tL}, L, Yl, pl);
    else LegChain::in(side, v0, q.thL, q.qdL, tau.tL, arm_eff, L, Yl, pl);
    art_pair_sum(Yl);
#pragma unroll
    for (int k = 0; k < 6; ++k) pl[k] = pair_sum(pl[k]);
    art_add(Y0, Yl);
#pragma unroll
    for (int k = 0; k < 6; ++k) p0[k] += pl[k];
    Art Yb; body_inertia<0, 0>(side, Yb); art_add(Y0, Yb);
    double Iv[6], pv[6]; inertia_mul<0, 0>(side, v0, Iv); crf(v0, Iv, pv);
#pragma unroll
    for (int k = 0; k < 6; ++k) p0[k] += pv[k];
  }
  __builtin_amdgcn_sched_barrier(0);
  // pelvis: one factorisation, the solve and (left lane's job to store) the explicit inverse column by column
  double rhs[6] = {-p0[0], -p0[1], -p0[2], -p0[3], -p0[4], -p0[5]}, a0[6];
  Ldl6 F; ldl6_factor(Y0, F);
  ldl6_solve(F, rhs, a0);
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double e[6] = {0, 0, 0, 0, 0, 0}, col[6]; e[c] = 1.0;
    ldl6_solve(F, e, col);
#pragma unroll
    for (int r = 0; r < 6; ++r) inv36[6 * r + c] = col[r];
  }
  const double mg[3] = {-grav[0], -grav[1], -grav[2]};
  const double a0p[3] = {R0[0] * mg[0] + R0[3] * mg[1] + R0[6] * mg[2], R0[1] * mg[0] + R0[4] * mg[1] + R0[7] * mg[2], R0[2] * mg[0] + R0[5] * mg[1] + R0[8] * mg[2]};
  double wxv[3]; cross(v0, v0 + 3, wxv);
  const double lin[3] = {a0[3] - a0p[0] + wxv[0], a0[4] - a0p[1] + wxv[1], a0[5] - a0p[2] + wxv[2]};
  qbase[0] = R0[0] * lin[0] + R0[1] * lin[1] + R0[2] * lin[2];
  qbase[1] = R0[3] * lin[0] + R0[4] * lin[1] + R0[5] * lin[2];
  qbase[2] = R0[6] * lin[0] + R0[7] * lin[1] + R0[8] * lin[2];
  qbase[3] = a0[0]; qbase[4] = a0[1]; qbase[5] = a0[2];
  aL[0] = a0[3] + wxv[0]; aL[1] = a0[4] + wxv[1]; aL[2] = a0[5] + wxv[2];
  sink(0, v0, a0, 0.0, 1.0);
  __builtin_amdgcn_sched_barrier(0);
  double v11[6], a11[6], sc11[2];
  qacc.q11 = body_acc<11, 11>(side, v0, a0, q.th11, q.qd11, L, 0, v11, a11, sc11);
  sink(11, v11, a11, sc11[0], sc11[1]);
  ArmChain::acc_dump<0>(side, v11, a11, q.thA, q.qdA, L, qacc.qA, sink);
  __builtin_amdgcn_sched_barrier(0);
  LegChain::acc_dump<0>(side, v0, a0, q.thL, q.qdL, L, qacc.qL, sink);
}

// sweeps of a force / acceleration increment along a mirrored chain whose U_i, 1/D_i sit in LDS (slot blocks from SLOT0)
template <int FL, int FR, int LEN, int SLOT0> struct ChainResp {
  // inward from the chain's last body: dp = bias-force increment there; du[K] = joint-force increments; returns the root share
  template <int K> static DEVFN void in(bool side, const double* sn, const double* cs, const LaneLds& L, double* dp, double* du, double* root) {
    constexpr int AX = C_AXIS[FL + K];
    du[K] = -dp[AX];
    const double sc = du[K] * L[SLOT0 + 8 * K + 6];
    double dpa[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) dpa[k] = dp[k] + L[SLOT0 + 8 * K + k] * sc;
    double par[6] = {0, 0, 0, 0, 0, 0};
    xf_force_acc<FL + K, FR + K>(side, dpa, sn[K], cs[K], par);
    if constexpr (K > 0) in<K - 1>(side, sn, cs, L, par, du, root);
    else {
#pragma unroll
      for (int k = 0; k < 6; ++k) root[k] = par[k];
    }
  }
  // outward from the root acceleration increment ap; du may be nullptr (no joint-force increments on this chain)
  template <int K> static DEVFN void out(bool side, const double* sn, const double* cs, const LaneLds& L, const double* ap, const double* du, double* dq, double* alast) {
    constexpr int AX = C_AXIS[FL + K];
    double a[6]; xf_motion<FL + K, FR + K>(side, ap, sn[K], cs[K], a);
    double s = du ? du[K] : 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s -= L[SLOT0 + 8 * K + k] * a[k];
    const double qdd = s * L[SLOT0 + 8 * K + 6];
    a[AX] += qdd;
    if (dq) dq[K] = qdd;
    if constexpr (K + 1 < LEN) out<K + 1>(side, sn, cs, L, a, du, dq, alast);
    else if (alast) {
#pragma unroll
      for (int k = 0; k < 6; ++k) alast[k] = a[k];
    }
  }
  // velocity, gravity-offset acceleration and two pure-rotation images (linear 3-vectors) at the chain's last body
  template <int K> static DEVFN void kin(bool side, const double* sn, const double* cs, const double* th_unused, const double* qd, const double* qdd,
                                         const double* vp, const double* ap, const double* o1p, const double* o2p, double* v, double* a, double* o1, double* o2) {
    constexpr int AX = C_AXIS[FL + K];
    double vv[6], aa[6];
    xf_motion<FL + K, FR + K>(side, vp, sn[K], cs[K], vv); vv[AX] += qd[K];
    xf_motion<FL + K, FR + K>(side, ap, sn[K], cs[K], aa);
    double ca[3], cl[3]; cross_axis<AX>(vv, ca); cross_axis<AX>(vv + 3, cl);
#pragma unroll
    for (int k = 0; k < 3; ++k) { aa[k] += qd[K] * ca[k]; aa[3 + k] += qd[K] * cl[k]; }
    aa[AX] += qdd[K];
    double q1[3], q2[3];
    rotT<FL + K, FR + K>(side, o1p, sn[K], cs[K], q1); rotT<FL + K, FR + K>(side, o2p, sn[K], cs[K], q2);
    if constexpr (K + 1 < LEN) kin<K + 1>(side, sn, cs, th_unused, qd, qdd, vv, aa, q1, q2, v, a, o1, o2);
    else {
#pragma unroll
      for (int k = 0; k < 6; ++k) { v[k] = vv[k]; a[k] = aa[k]; }
#pragma unroll
      for (int k = 0; k < 3; ++k) { o1[k] = q1[k]; o2[k] = q2[k]; }
    }
  }
};
typedef ChainResp<1, 6, 5, 8> LegResp;
typedef ChainResp<12, 16, 4, 48> ArmResp;

// symmetric n x n in packed lower storage (row i: i (i + 1) / 2 + j, j <= i)
DEVFN constexpr int pidx(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
// in-place Cholesky of the packed SPD 12 x 12 (chol12_factor; the reciprocals of the pivots on the diagonal) and solution of C x = b with
// the factor (chol12_apply, b overwritten); chol12_solve = both
DEVFN void chol12_apply(const double* C, double* b) {
#pragma unroll
  for (int i = 0; i < 12; ++i) { double t = b[i];
#pragma unroll
    for (int k = 0; k < 12; ++k) if (k < i) t -= C[pidx(i, k)] * b[k];
    b[i] = t * C[pidx(i, i)]; }
#pragma unroll
  for (int i = 11; i >= 0; --i) { double t = b[i];
#pragma unroll
    for (int k = 0; k < 12; ++k) if (k > i) t -= C[pidx(k, i)] * b[k];
    b[i] = t * C[pidx(i, i)]; }
}
DEVFN void chol12_factor(double* C) {
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    double d = C[pidx(j, j)];
#pragma unroll
    for (int k = 0; k < 12; ++k) if (k < j) d -= C[pidx(j, k)] * C[pidx(j, k)];
    const double r = sqrt(d), ri = 1.0 / r;
    C[pidx(j, j)] = ri;                       // the reciprocal of the pivot is kept
#pragma unroll
    for (int i = 0; i < 12; ++i) if (i > j) {
      double t = C[pidx(i, j)];
#pragma unroll
      for (int k = 0; k < 12; ++k) if (k < j) t -= C[pidx(i, k)] * C[pidx(j, k)];
      C[pidx(i, j)] = t * ri;
    }
  }
}
DEVFN void chol12_solve(double* C, double* b) {
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    double d = C[pidx(j, j)];
#pragma unroll
    for (int k = 0; k < 12; ++k) if (k < j) d -= C[pidx(j, k)] * C[pidx(j, k)];
    const double r = sqrt(d), ri = 1.0 / r;
    C[pidx(j, j)] = ri;                       // the reciprocal of the pivot is kept
#pragma unroll
    for (int i = 0; i < 12; ++i) if (i > j) {
      double t = C[pidx(i, j)];
#pragma unroll
      for (int k = 0; k < 12; ++k) if (k < j) t -= C[pidx(i, k)] * C[pidx(j, k)];
      C[pidx(i, j)] = t * ri;
    }
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) { double t = b[i];
#pragma unroll
    for (int k = 0; k < 12; ++k) if (k < i) t -= C[pidx(i, k)] * b[k];
    b[i] = t * C[pidx(i, i)]; }
#pragma unroll
  for (int i = 11; i >= 0; --i) { double t = b[i];
#pragma unroll
    for (int k = 0; k < 12; ++k) if (k > i) t -= C[pidx(k, i)] * b[k];
    b[i] = t * C[pidx(i, i)]; }
}

// Coulomb limit (contact mode 3): a sliding foot keeps the rotation rows and the normal translation row of its constraint.  With
// Pi = blockdiag(I3, u u^T) (u: world up axis in link coordinates) the reduced system S (C + soft I) S^T lambda_s = S b, S = [I3 0; 0 u^T],
// is the range-of-Pi part of  (Pi (C + soft I) Pi + (I - Pi)) lambda = Pi b  -- same 12 x 12 shape as the rigid system, so the
// masked solve below serves it: this routine applies Pi from both sides to the packed matrix in place (translation block at
// rows / columns o..o+2) and puts the identity on the removed directions.
DEVFN void project_sliding_foot(double* Cw, int o, const double* u) {
#pragma unroll
  for (int j = 0; j < 12; ++j) if (j < o || j > o + 2) {
    const double w = u[0] * Cw[pidx(o, j)] + u[1] * Cw[pidx(o + 1, j)] + u[2] * Cw[pidx(o + 2, j)];
#pragma unroll
    for (int a = 0; a < 3; ++a) Cw[pidx(o + a, j)] = u[a] * w;
  }
  double s = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) s += u[a] * u[c] * Cw[pidx(o + a, o + c)];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c <= a; ++c) Cw[pidx(o + a, o + c)] = u[a] * u[c] * (s - 1.0) + (a == c ? 1.0 : 0.0);
}

// Corrects the accelerations of the free solve (qbase, qacc; U_i, 1/D_i of it still in LDS) for the stance constraints.
// st_own / st_par: stance flags of this lane's / the partner's foot; mode 2 = unilateral, 3 = unilateral + Coulomb limit mu, 4 = 3 with kinetic friction on the sliding feet.
// Y0, a0: pelvis articulated inertia and raw (gravity-offset, body-frame) pelvis acceleration of the free solve.
// KIN: the copy that knows kinetic friction (mode 4); the other one is compiled without that branch (the constrained step is register- and
// scratch-critical: the mere presence of the branch costs the common modes 2.5 %, measured on the contact bench)
template <bool KIN>
DEVFN void stance_correct(bool side, const double* R0, const double* vb, const HalfState& q, double h, double soft, int mode, bool st_own, bool st_par,
                          const double* grav, const LaneLds& L, const Art& Y0, const double* a0, double* qbase, HalfAcc& qacc, double mu = 1.0) {
  LegTrig T;
#pragma unroll
  for (int k = 0; k < 5; ++k) h1f::sincos_fast(q.thL[k], &T.sn[k], &T.cs[k]);
  Ldl6 F; ldl6_factor(Y0, F);
  // own foot: velocity, acceleration of the free solve, gravity offset and world up axis in link coordinates
  double v0[6] = {vb[3], vb[4], vb[5], 0, 0, 0};
  v0[3] = R0[0] * vb[0] + R0[3] * vb[1] + R0[6] * vb[2];
  v0[4] = R0[1] * vb[0] + R0[4] * vb[1] + R0[7] * vb[2];
  v0[5] = R0[2] * vb[0] + R0[5] * vb[1] + R0[8] * vb[2];
  const double mg[3] = {-grav[0], -grav[1], -grav[2]};
  const double a0p[3] = {R0[0] * mg[0] + R0[3] * mg[1] + R0[6] * mg[2], R0[1] * mg[0] + R0[4] * mg[1] + R0[7] * mg[2], R0[2] * mg[0] + R0[5] * mg[1] + R0[8] * mg[2]};
  const double zb[3] = {R0[6], R0[7], R0[8]};        // R0^T e_z
  double vf[6], af[6], off[3], zl[3];
  LegResp::kin<0>(side, T.sn, T.cs, nullptr, q.qdL, qacc.qL, v0, a0, a0p, zb, vf, af, off, zl);
  double bown[6];
#pragma unroll
  for (int k = 0; k < 3; ++k) { bown[k] = -vf[k] / h - af[k]; bown[3 + k] = -vf[3 + k] / h - (af[3 + k] - off[k]); }
  // unit wrenches on the own foot: inward shares at the pelvis and joint-force increments
  double p0[6][6], du[6][5];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double dp[6] = {0, 0, 0, 0, 0, 0}; dp[c] = -1.0;
    LegResp::in<4>(side, T.sn, T.cs, L, dp, du[c], p0[c]);
  }
  // rows of C that belong to the own foot.  Own columns (wrench c on the own foot): the pelvis' response and the outward sweep with the
  // joint-force increments.  The partner's columns reach this foot through the pelvis alone, a_f = T a0 with T the leg's acceleration
  // transmission -- and the inward sweeps above have already computed it: force transmission is its transpose, p0[c] = -T^T e_c, so
  // a_f[r] = -sum_k p0[r][k] a0[k] with the partner's pelvis response a0: a 6 x 6 product instead of an outward sweep per column.
  double Cown[6][6], Ccross[6][6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double pj[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) pj[k] = -p0[c][k];
    double a0m[6]; ldl6_solve(F, pj, a0m);
    double afo[6];
    LegResp::out<0>(side, T.sn, T.cs, L, a0m, du[c], nullptr, afo);
    double a0t[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) a0t[k] = xch(a0m[k]);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      Cown[r][c] = afo[r];
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) t -= p0[r][k] * a0t[k];
      Ccross[r][c] = t;
    }
  }
  // both lanes assemble the same 12 x 12 system: rows 0..5 from the left lane, 6..11 from the right lane
  // (columns 0..5 = left foot's wrench components: the left lane's own, the right lane's cross block; 6..11 the other way round)
  double C[78], b[12];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const double bo = bown[r], bp = xch(bo);
    b[r] = side ? bp : bo; b[6 + r] = side ? bo : bp;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const double mine = (j < 6) ? (side ? Ccross[r][j % 6] : Cown[r][j % 6]) : (side ? Cown[r][j % 6] : Ccross[r][j % 6]), theirs = xch(mine);
      const double left = side ? theirs : mine, right = side ? mine : theirs;
      if (j <= r) C[pidx(r, j)] = left;             // lower triangle of the left foot's rows
      if (j <= 6 + r) C[pidx(6 + r, j)] = right;    // lower triangle of the right foot's rows
    }
  }
  bool actL = side ? st_par : st_own, actR = side ? st_own : st_par;
  double lam[12];
  double Cw[78];
  auto solve_masked = [&]() {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const bool ai = i < 6 ? actL : actR;
      lam[i] = ai ? b[i] : 0.0;
#pragma unroll
      for (int j = 0; j < 12; ++j) if (j <= i) {
        const bool aj = j < 6 ? actL : actR;
        Cw[pidx(i, j)] = (ai && aj) ? C[pidx(i, j)] + (i == j ? soft : 0.0) : (i == j ? 1.0 : 0.0);
      }
    }
    chol12_solve(Cw, lam);
  };
  solve_masked();
  if (mode >= 2) {
    // unilateral: normal force on a foot = (world up axis in link coordinates) . (force part of its multiplier)
    const double* lo = lam + (side ? 6 : 0);
    const double fz_own = zl[0] * lo[3] + zl[1] * lo[4] + zl[2] * lo[5], fz_par = xch(fz_own);
    const double fzL = side ? fz_par : fz_own, fzR = side ? fz_own : fz_par;
    const bool relL = actL && fzL < 0.0, relR = actR && fzR < 0.0;
    if (relL || relR) { actL = actL && !relL; actR = actR && !relR; solve_masked(); }
  }
  if (mode >= 3) {
    // Coulomb limit on the feet that still push: |f_t|^2 = |f|^2 - f_n^2 > mu^2 f_n^2 -> the foot slides (oracle
    // forward_dynamics_mj_stance, mode 3): project its translation block on the up axis and solve once more
    const double* lo = lam + (side ? 6 : 0);
    const double fn = zl[0] * lo[3] + zl[1] * lo[4] + zl[2] * lo[5];
    const double ft2 = lo[3] * lo[3] + lo[4] * lo[4] + lo[5] * lo[5] - fn * fn;
    const bool act_own = side ? actR : actL;
    const bool sl_own = act_own && ft2 > mu * mu * fn * fn, sl_par = xch_flag(sl_own);
    const bool slL = side ? sl_par : sl_own, slR = side ? sl_own : sl_par;
    if (slL || slR) {
      double uL[3], uR[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { const double mine = zl[k], theirs = xch(mine); uL[k] = side ? theirs : mine; uR[k] = side ? mine : theirs; }
      if (KIN && mode == 4) {
        // Kinetic friction: the sliding foot's normal multiplier pushes along up + mu t (t: the unit direction in which the sticking solution
        // pulled), its constraint row stays up^T.  With G = blockdiag(I3, I3 + mu t up^T) the system (oracle, mode 4: S C F^T + soft I) is
        //     (K + sum_f c_f u_f^T) y = Pi b ,   K = Pi (C + soft I) Pi + (I - Pi)  (the matrix of mode 3, SPD),   c_f = mu Pi C t_f ,   lambda = G y
        // -- a rank-one update of K per sliding foot: one Cholesky factorisation, three substitutions and a 2 x 2 system (Woodbury),
        // all on the packed matrix in registers, same bits on both lanes.
        const double nt = sqrt(ft2 > 0.0 ? ft2 : 1.0);
        double tL[3], tR[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double mine = sl_own ? (lo[3 + k] - fn * zl[k]) / nt : 0.0, theirs = xch(mine); tL[k] = side ? theirs : mine; tR[k] = side ? mine : theirs; }
        double z0[12], zL[12], zR[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const bool ai = i < 6 ? actL : actR;
          z0[i] = ai ? b[i] : 0.0;
          // c_f = mu C t_f on the active rows (t_f lives in the translation block of foot f; zero unless that foot slides)
          double cl = 0.0, cr = 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) { cl += C[pidx(i, 3 + k)] * tL[k]; cr += C[pidx(i, 9 + k)] * tR[k]; }
          zL[i] = (ai && slL) ? mu * cl : 0.0; zR[i] = (ai && slR) ? mu * cr : 0.0;
#pragma unroll
          for (int j = 0; j < 12; ++j) if (j <= i) {
            const bool aj = j < 6 ? actL : actR;
            Cw[pidx(i, j)] = (ai && aj) ? C[pidx(i, j)] + (i == j ? soft : 0.0) : (i == j ? 1.0 : 0.0);
          }
        }
        // Pi on the right-hand sides and on the matrix
        auto proj3 = [](double* v, const double* u) { const double w = u[0] * v[0] + u[1] * v[1] + u[2] * v[2]; v[0] = u[0] * w; v[1] = u[1] * w; v[2] = u[2] * w; };
        if (slL) { project_sliding_foot(Cw, 3, uL); proj3(z0 + 3, uL); proj3(zL + 3, uL); proj3(zR + 3, uL); }
        if (slR) { project_sliding_foot(Cw, 9, uR); proj3(z0 + 9, uR); proj3(zL + 9, uR); proj3(zR + 9, uR); }
        chol12_factor(Cw);
        chol12_apply(Cw, z0); chol12_apply(Cw, zL); chol12_apply(Cw, zR);
        // (I + U^T Z) alpha = U^T z0 with U = [u_L, u_R] embedded: 2 x 2 (a foot that does not slide has c = 0: its alpha is u^T z0, unused)
        const double aLL = 1.0 + uL[0] * zL[3] + uL[1] * zL[4] + uL[2] * zL[5], aLR = uL[0] * zR[3] + uL[1] * zR[4] + uL[2] * zR[5];
        const double aRL = uR[0] * zL[9] + uR[1] * zL[10] + uR[2] * zL[11], aRR = 1.0 + uR[0] * zR[9] + uR[1] * zR[10] + uR[2] * zR[11];
        const double rL = uL[0] * z0[3] + uL[1] * z0[4] + uL[2] * z0[5], rR = uR[0] * z0[9] + uR[1] * z0[10] + uR[2] * z0[11];
        const double det = aLL * aRR - aLR * aRL;
        const double alL = (rL * aRR - aLR * rR) / det, alR = (aLL * rR - aRL * rL) / det;
#pragma unroll
        for (int i = 0; i < 12; ++i) lam[i] = z0[i] - zL[i] * alL - zR[i] * alR;
        // lambda = G y (y made exactly a multiple of up on the sliding feet first)
        if (slL) { const double w = uL[0] * lam[3] + uL[1] * lam[4] + uL[2] * lam[5]; lam[3] = (uL[0] + mu * tL[0]) * w; lam[4] = (uL[1] + mu * tL[1]) * w; lam[5] = (uL[2] + mu * tL[2]) * w; }
        if (slR) { const double w = uR[0] * lam[9] + uR[1] * lam[10] + uR[2] * lam[11]; lam[9] = (uR[0] + mu * tR[0]) * w; lam[10] = (uR[1] + mu * tR[1]) * w; lam[11] = (uR[2] + mu * tR[2]) * w; }
      } else {
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const bool ai = i < 6 ? actL : actR;
        lam[i] = ai ? b[i] : 0.0;
#pragma unroll
        for (int j = 0; j < 12; ++j) if (j <= i) {
          const bool aj = j < 6 ? actL : actR;
          Cw[pidx(i, j)] = (ai && aj) ? C[pidx(i, j)] + (i == j ? soft : 0.0) : (i == j ? 1.0 : 0.0);
        }
      }
      if (slL) { project_sliding_foot(Cw, 3, uL); const double w = uL[0] * lam[3] + uL[1] * lam[4] + uL[2] * lam[5]; lam[3] = uL[0] * w; lam[4] = uL[1] * w; lam[5] = uL[2] * w; }
      if (slR) { project_sliding_foot(Cw, 9, uR); const double w = uR[0] * lam[9] + uR[1] * lam[10] + uR[2] * lam[11]; lam[9] = uR[0] * w; lam[10] = uR[1] * w; lam[11] = uR[2] * w; }
      chol12_solve(Cw, lam);
      // (the removed directions carry zero up to rounding: make it exact, both lanes alike)
      if (slL) { const double w = uL[0] * lam[3] + uL[1] * lam[4] + uL[2] * lam[5]; lam[3] = uL[0] * w; lam[4] = uL[1] * w; lam[5] = uL[2] * w; }
      if (slR) { const double w = uR[0] * lam[9] + uR[1] * lam[10] + uR[2] * lam[11]; lam[9] = uR[0] * w; lam[10] = uR[1] * w; lam[11] = uR[2] * w; }
      }
    }
  }
  // propagate the multipliers: own wrench inward, pelvis, outward along every chain of this lane
  {
    const double* lo = lam + (side ? 6 : 0);
    double dp[6], duo[5], pown[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) dp[k] = -lo[k];
    LegResp::in<4>(side, T.sn, T.cs, L, dp, duo, pown);
    double ptot[6], da0[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) ptot[k] = -pair_sum(pown[k]);
    ldl6_solve(F, ptot, da0);
    double dqL[5];
    LegResp::out<0>(side, T.sn, T.cs, L, da0, duo, dqL, nullptr);
#pragma unroll
    for (int k = 0; k < 5; ++k) qacc.qL[k] += dqL[k];
    // torso (slot block 0) and arm: no joint-force increments, only the pelvis acceleration travels outward
    double s11, c11; h1f::sincos_fast(q.th11, &s11, &c11);
    double a11[6]; xf_motion<11, 11>(side, da0, s11, c11, a11);
    double sacc = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) sacc -= L[k] * a11[k];
    const double dq11 = sacc * L[6];
    a11[C_AXIS[11]] += dq11;
    qacc.q11 += dq11;
    double snA[4], csA[4], dqA[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) h1f::sincos_fast(q.thA[k], &snA[k], &csA[k]);
    ArmResp::out<0>(side, snA, csA, L, a11, nullptr, dqA, nullptr);
#pragma unroll
    for (int k = 0; k < 4; ++k) qacc.qA[k] += dqA[k];
    qbase[0] += R0[0] * da0[3] + R0[1] * da0[4] + R0[2] * da0[5];
    qbase[1] += R0[3] * da0[3] + R0[4] * da0[4] + R0[5] * da0[5];
    qbase[2] += R0[6] * da0[3] + R0[7] * da0[4] + R0[8] * da0[5];
    qbase[3] += da0[0]; qbase[4] += da0[1]; qbase[5] += da0[2];
  }
}

// this lane's view of one state x = [qpos(26), qvel(25)]: the floating base (both lanes) + its HalfState
struct HalfX { double p[3], quat[4], vb[6]; HalfState q; };
struct HalfU { double u11, uL[5], uA[4]; };
// hinge j of this lane: leg k -> 5 side + k, arm k -> 11 + 4 side + k, torso -> 10
DEVFN int jleg(bool side, int k) { return (side ? 5 : 0) + k; }
DEVFN int jarm(bool side, int k) { return (side ? 15 : 11) + k; }

DEVFN void load_half(bool side, const double* x, HalfX& h) {
#pragma unroll
  for (int k = 0; k < 3; ++k) h.p[k] = x[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) h.quat[k] = x[3 + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) h.vb[k] = x[NQ + k];
  h.q.th11 = x[7 + 10]; h.q.qd11 = x[NQ + 6 + 10];
#pragma unroll
  for (int k = 0; k < 5; ++k) { h.q.thL[k] = x[7 + jleg(side, k)]; h.q.qdL[k] = x[NQ + 6 + jleg(side, k)]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { h.q.thA[k] = x[7 + jarm(side, k)]; h.q.qdA[k] = x[NQ + 6 + jarm(side, k)]; }
}
// the even lane stores the shared coordinates, every lane its own hinges
DEVFN void store_half(bool side, const HalfX& h, double* x) {
  if (!side) {
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = h.p[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[3 + k] = h.quat[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) x[NQ + k] = h.vb[k];
    x[7 + 10] = h.q.th11; x[NQ + 6 + 10] = h.q.qd11;
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) { x[7 + jleg(side, k)] = h.q.thL[k]; x[NQ + 6 + jleg(side, k)] = h.q.qdL[k]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { x[7 + jarm(side, k)] = h.q.thA[k]; x[NQ + 6 + jarm(side, k)] = h.q.qdA[k]; }
}
DEVFN double clampu(double u, const double* range) { return u < range[0] ? range[0] : (u > range[1] ? range[1] : u); }

// x <- f(x, u) in place: RobotUtils::rolloutOneStep (reference src/common/robot_utils.cpp:106-117), smooth regime
DEVFN void step(bool side, HalfX& h, const HalfU& u, double dt, const double* grav, const LaneLds& L) {
  const double qn = sqrt(h.quat[0] * h.quat[0] + h.quat[1] * h.quat[1] + h.quat[2] * h.quat[2] + h.quat[3] * h.quat[3]);
  const double qh[4] = {h.quat[0] / qn, h.quat[1] / qn, h.quat[2] / qn, h.quat[3] / qn};
  double R0[9]; quat_R(qh[0], qh[1], qh[2], qh[3], R0);
  HalfTau tau;
  tau.t11 = clampu(u.u11, C_CTRLRANGE[10]) - DAMPING * h.q.qd11;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double lo = side ? C_CTRLRANGE[5 + k][0] : C_CTRLRANGE[k][0], hi = side ? C_CTRLRANGE[5 + k][1] : C_CTRLRANGE[k][1];
    const double uc = u.uL[k] < lo ? lo : (u.uL[k] > hi ? hi : u.uL[k]);
    tau.tL[k] = uc - DAMPING * h.q.qdL[k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double lo = side ? C_CTRLRANGE[15 + k][0] : C_CTRLRANGE[11 + k][0], hi = side ? C_CTRLRANGE[15 + k][1] : C_CTRLRANGE[11 + k][1];
    const double uc = u.uA[k] < lo ? lo : (u.uA[k] > hi ? hi : u.uA[k]);
    tau.tA[k] = uc - DAMPING * h.q.qdA[k];
  }
  double qb[6]; HalfAcc qa;
  forward_dynamics(side, R0, h.vb, h.q, tau, ARMATURE + dt * DAMPING, grav, L, qb, qa);
  // semi-implicit Euler: v' = v + h qacc, q' = q (+) h v'
#pragma unroll
  for (int k = 0; k < 6; ++k) h.vb[k] += dt * qb[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) h.p[k] += dt * h.vb[k];
  h.q.qd11 += dt * qa.q11; h.q.th11 += dt * h.q.qd11;
#pragma unroll
  for (int k = 0; k < 5; ++k) { h.q.qdL[k] += dt * qa.qL[k]; h.q.thL[k] += dt * h.q.qdL[k]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { h.q.qdA[k] += dt * qa.qA[k]; h.q.thA[k] += dt * h.q.qdA[k]; }
  const double s = (h.vb[3] * h.vb[3] + h.vb[4] * h.vb[4] + h.vb[5] * h.vb[5]) * (dt * dt);
  double c, so;
  if (s < 1e-6) { c = 1.0 - s / 8.0 + s * s / 384.0 - s * s * s / 46080.0; so = 0.5 - s / 48.0 + s * s / 3840.0 - s * s * s / 645120.0; }
  else { const double a = sqrt(s); double sn, cn; h1f::sincos_fast(0.5 * a, &sn, &cn); c = cn; so = sn / a; }
  const double ew = c, ex = so * dt * h.vb[3], ey = so * dt * h.vb[4], ez = so * dt * h.vb[5];
  const double rw = qh[0] * ew - qh[1] * ex - qh[2] * ey - qh[3] * ez;
  const double rx = qh[0] * ex + qh[1] * ew + qh[2] * ez - qh[3] * ey;
  const double ry = qh[0] * ey - qh[1] * ez + qh[2] * ew + qh[3] * ex;
  const double rz = qh[0] * ez + qh[1] * ey - qh[2] * ex + qh[3] * ew;
  const double rn = sqrt(rw * rw + rx * rx + ry * ry + rz * rz);
  h.quat[0] = rw / rn; h.quat[1] = rx / rn; h.quat[2] = ry / rn; h.quat[3] = rz / rn;
}

// x <- f(x, u) with the stance constraints of the scheduled feet (contact mode 1 / 2 / 3)
// joint-limit rows (h1.xml jnt_range, enforced inside mj_step; oracle h1_step): a hinge past its range that the
// step would still move outward is stopped -- the dynamics run once more with those hinges acceleration-prescribed, qacc_i = -v_i / h
// (forward_dynamics: armature 2^1000), the stance rows solved on that system.
constexpr double LOCK_ARM = 0x1p1000;
// The hinges of this lane that the step stops, decided on the accelerations qa of the step without the rows, as a bit mask (bit 0 the
// torso hinge, 1..5 the leg's, 6..9 the arm's); and what the mask means for the second pass: torque c 2^1000 and armature 2^1000 there.
// kr: the restoring stiffness lim_k (DynParams; 0: the pure stop).  The row prescribes qacc_i = -v_i / h - kr r_i with r_i the violation, and is
// active when the step without the rows falls short of that on the outward side: v_i + h (qacc_i + kr r_i) points out of the range
// (kr = 0: v_i + h qacc_i, the round-5 expression bit for bit -- 0 * r is 0 and x + 0.0 is x).
DEVFN unsigned limit_lock_mask(bool side, const HalfState& q, const HalfAcc& qa, double dt, double kr) {
  unsigned mask = 0u;
  auto lim = [&](int bit, double lo, double hi, double th, double qd, double qdd) {
    const double r = th > hi ? th - hi : th - lo;
    const double vn = qd + dt * (qdd + kr * r);
    if ((th > hi && vn > 0.0) || (th < lo && vn < 0.0)) mask |= 1u << bit;
  };
  lim(0, C_JRANGE[10][0], C_JRANGE[10][1], q.th11, q.qd11, qa.q11);
#pragma unroll
  for (int k = 0; k < 5; ++k) lim(1 + k, side ? C_JRANGE[5 + k][0] : C_JRANGE[k][0], side ? C_JRANGE[5 + k][1] : C_JRANGE[k][1], q.thL[k], q.qdL[k], qa.qL[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) lim(6 + k, side ? C_JRANGE[15 + k][0] : C_JRANGE[11 + k][0], side ? C_JRANGE[15 + k][1] : C_JRANGE[11 + k][1], q.thA[k], q.qdA[k], qa.qA[k]);
  return mask;
}
DEVFN void apply_lock_mask(bool side, unsigned mask, const HalfState& q, double dt, double kr, HalfTau& tau, HalfTau& add) {
  auto ap = [&](int bit, double lo, double hi, double th, double qd, double& tq, double& ad) {
    const bool lk = ((mask >> bit) & 1u) != 0u;
    const double r = th > hi ? th - hi : th - lo;
    ad = lk ? LOCK_ARM : 0.0;
    tq = lk ? (-qd / dt - kr * r) * LOCK_ARM : tq;
  };
  ap(0, C_JRANGE[10][0], C_JRANGE[10][1], q.th11, q.qd11, tau.t11, add.t11);
#pragma unroll
  for (int k = 0; k < 5; ++k) ap(1 + k, side ? C_JRANGE[5 + k][0] : C_JRANGE[k][0], side ? C_JRANGE[5 + k][1] : C_JRANGE[k][1], q.thL[k], q.qdL[k], tau.tL[k], add.tL[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) ap(6 + k, side ? C_JRANGE[15 + k][0] : C_JRANGE[11 + k][0], side ? C_JRANGE[15 + k][1] : C_JRANGE[11 + k][1], q.thA[k], q.qdA[k], tau.tA[k], add.tA[k]);
}
DEVFN bool limit_locks(bool side, const HalfState& q, const HalfAcc& qa, double dt, double kr, HalfTau& tau, HalfTau& add) {
  const unsigned mask = limit_lock_mask(side, q, qa, dt, kr);
  apply_lock_mask(side, mask, q, dt, kr, tau, add);
  return mask != 0u;
}
// The step with the rows: dyn_split_kernels.hip step_stance_shared_lim -- the accelerations are ONE function there, called once without
// extra armature (x + 0.0 is x: the step without the rows, bit for bit) and, by the lane pairs that have a hinge to stop, once more with
// it.  (Two inlined copies doubled the step's private segment, 1.3 -> 3.3 KB per lane, and with a wave per SIMD of these kernels in
// flight that crosses the runtime's scratch budget: the launch is throttled and every step, stopped hinges or not, paid 2.4 x.)

// the pieces of a constrained step: unit quaternion / base rotation / hinge torques; the accelerations (free recursion + stance rows; PER: with
// the per-hinge extra armature of the joint-limit rows); the semi-implicit Euler update
DEVFN void stance_prepare(bool side, const HalfX& h, const HalfU& u, double* qh, double* R0, HalfTau& tau) {
  const double qn = sqrt(h.quat[0] * h.quat[0] + h.quat[1] * h.quat[1] + h.quat[2] * h.quat[2] + h.quat[3] * h.quat[3]);
  qh[0] = h.quat[0] / qn; qh[1] = h.quat[1] / qn; qh[2] = h.quat[2] / qn; qh[3] = h.quat[3] / qn;
  quat_R(qh[0], qh[1], qh[2], qh[3], R0);
  tau.t11 = clampu(u.u11, C_CTRLRANGE[10]) - DAMPING * h.q.qd11;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double lo = side ? C_CTRLRANGE[5 + k][0] : C_CTRLRANGE[k][0], hi = side ? C_CTRLRANGE[5 + k][1] : C_CTRLRANGE[k][1];
    const double uc = u.uL[k] < lo ? lo : (u.uL[k] > hi ? hi : u.uL[k]);
    tau.tL[k] = uc - DAMPING * h.q.qdL[k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double lo = side ? C_CTRLRANGE[15 + k][0] : C_CTRLRANGE[11 + k][0], hi = side ? C_CTRLRANGE[15 + k][1] : C_CTRLRANGE[11 + k][1];
    const double uc = u.uA[k] < lo ? lo : (u.uA[k] > hi ? hi : u.uA[k]);
    tau.tA[k] = uc - DAMPING * h.q.qdA[k];
  }
}
template <bool KIN, bool PER>
DEVFN void stance_accelerations(bool side, const double* R0, const HalfX& h, const HalfTau& tau, double dt, const double* grav, const LaneLds& L, double soft, int mode,
                                bool st_own, bool st_par, double mu, double* qb, HalfAcc& qa, const HalfTau* add = nullptr) {
  Art Y0; double a0[6];
  forward_dynamics<PER>(side, R0, h.vb, h.q, tau, ARMATURE + dt * DAMPING, grav, L, qb, qa, &Y0, a0, add);
  if (st_own || st_par) stance_correct<KIN>(side, R0, h.vb, h.q, dt, soft, mode, st_own, st_par, grav, L, Y0, a0, qb, qa, mu);
}
DEVFN void integrate_half(HalfX& h, const double* qh, const double* qb, const HalfAcc& qa, double dt) {
#pragma unroll
  for (int k = 0; k < 6; ++k) h.vb[k] += dt * qb[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) h.p[k] += dt * h.vb[k];
  h.q.qd11 += dt * qa.q11; h.q.th11 += dt * h.q.qd11;
#pragma unroll
  for (int k = 0; k < 5; ++k) { h.q.qdL[k] += dt * qa.qL[k]; h.q.thL[k] += dt * h.q.qdL[k]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { h.q.qdA[k] += dt * qa.qA[k]; h.q.thA[k] += dt * h.q.qdA[k]; }
  const double s = (h.vb[3] * h.vb[3] + h.vb[4] * h.vb[4] + h.vb[5] * h.vb[5]) * (dt * dt);
  double c, so;
  if (s < 1e-6) { c = 1.0 - s / 8.0 + s * s / 384.0 - s * s * s / 46080.0; so = 0.5 - s / 48.0 + s * s / 3840.0 - s * s * s / 645120.0; }
  else { const double a = sqrt(s); double sn, cn; h1f::sincos_fast(0.5 * a, &sn, &cn); c = cn; so = sn / a; }
  const double ew = c, ex = so * dt * h.vb[3], ey = so * dt * h.vb[4], ez = so * dt * h.vb[5];
  const double rw = qh[0] * ew - qh[1] * ex - qh[2] * ey - qh[3] * ez;
  const double rx = qh[0] * ex + qh[1] * ew + qh[2] * ez - qh[3] * ey;
  const double ry = qh[0] * ey - qh[1] * ez + qh[2] * ew + qh[3] * ex;
  const double rz = qh[0] * ez + qh[1] * ey - qh[2] * ex + qh[3] * ew;
  const double rn = sqrt(rw * rw + rx * rx + ry * ry + rz * rz);
  h.quat[0] = rw / rn; h.quat[1] = rx / rn; h.quat[2] = ry / rn; h.quat[3] = rz / rn;
}
// x <- f(x, u) with the stance constraints of the scheduled feet (contact mode 1 / 2 / 3 / 4)
template <bool KIN = false>
DEVFN void step_stance(bool side, HalfX& h, const HalfU& u, double dt, const double* grav, const LaneLds& L, double soft, int mode, bool st_own, bool st_par, double mu = 1.0) {
  double qh[4], R0[9]; HalfTau tau;
  stance_prepare(side, h, u, qh, R0, tau);
  double qb[6]; HalfAcc qa;
  stance_accelerations<KIN, false>(side, R0, h, tau, dt, grav, L, soft, mode, st_own, st_par, mu, qb, qa);
  integrate_half(h, qh, qb, qa, dt);
}

// ---- whole-body CoM with MuJoCo masses (RobotUtils::computeCoM, reference src/common/robot_utils.cpp:810-833):
// each lane sums the bodies of its side, pelvis and torso are counted by the even lane, then the pair adds up
template <int IL, int IR, int LEN, int K> DEVFN void com_chain(bool side, const double* Rp, const double* pp, const double* th, double* acc) {
  constexpr int a = C_AXIS[IL], b = (a + 1) % 3, d = (a + 2) % 3;
  double s, c; h1f::sincos_fast(th[K], &s, &c);
  double R[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double f[3];
    if constexpr (rfix_identity<IL>()) { f[0] = Rp[3 * r]; f[1] = Rp[3 * r + 1]; f[2] = Rp[3 * r + 2]; }
    else {
#pragma unroll
      for (int k = 0; k < 3; ++k) f[k] = Rp[3 * r] * RF(0, k) + Rp[3 * r + 1] * RF(1, k) + Rp[3 * r + 2] * RF(2, k);
    }
    R[3 * r + a] = f[a]; R[3 * r + b] = f[b] * c + f[d] * s; R[3 * r + d] = f[d] * c - f[b] * s;
  }
  double p[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) p[r] = pp[r] + Rp[3 * r] * PS(0) + Rp[3 * r + 1] * PS(1) + Rp[3 * r + 2] * PS(2);
#pragma unroll
  for (int r = 0; r < 3; ++r) acc[r] += MS * (p[r] + R[3 * r] * CM(0) + R[3 * r + 1] * CM(1) + R[3 * r + 2] * CM(2));
  if constexpr (K + 1 < LEN) com_chain<IL + 1, IR + 1, LEN, K + 1>(side, R, p, th, acc);
}
DEVFN void com_mj(bool side, const HalfX& h, double* com) {
  const double qn = sqrt(h.quat[0] * h.quat[0] + h.quat[1] * h.quat[1] + h.quat[2] * h.quat[2] + h.quat[3] * h.quat[3]);
  double R0[9]; quat_R(h.quat[0] / qn, h.quat[1] / qn, h.quat[2] / qn, h.quat[3] / qn, R0);
  double acc[3] = {0.0, 0.0, 0.0};
  // torso frame (both lanes need it for the arm)
  double R11[9], p11[3];
  {
    constexpr int IL = 11, IR = 11;
    constexpr int a = C_AXIS[11], b = (a + 1) % 3, d = (a + 2) % 3;
    double s, c; h1f::sincos_fast(h.q.th11, &s, &c);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      double f[3];
      if constexpr (rfix_identity<11>()) { f[0] = R0[3 * r]; f[1] = R0[3 * r + 1]; f[2] = R0[3 * r + 2]; }
      else {
#pragma unroll
        for (int k = 0; k < 3; ++k) f[k] = R0[3 * r] * C_RFIX[11][0][k] + R0[3 * r + 1] * C_RFIX[11][1][k] + R0[3 * r + 2] * C_RFIX[11][2][k];
      }
      R11[3 * r + a] = f[a]; R11[3 * r + b] = f[b] * c + f[d] * s; R11[3 * r + d] = f[d] * c - f[b] * s;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) p11[r] = h.p[r] + R0[3 * r] * C_POS[11][0] + R0[3 * r + 1] * C_POS[11][1] + R0[3 * r + 2] * C_POS[11][2];
    (void)IL; (void)IR;
  }
  if (!side) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      acc[r] += C_MASS[0] * (h.p[r] + R0[3 * r] * C_COM[0][0] + R0[3 * r + 1] * C_COM[0][1] + R0[3 * r + 2] * C_COM[0][2]);
      acc[r] += C_MASS[11] * (p11[r] + R11[3 * r] * C_COM[11][0] + R11[3 * r + 1] * C_COM[11][1] + R11[3 * r + 2] * C_COM[11][2]);
    }
  }
  com_chain<1, 6, 5, 0>(side, R0, h.p, h.q.thL, acc);
  com_chain<12, 16, 4, 0>(side, R11, p11, h.q.thA, acc);
  double mtot = 0.0;
#pragma unroll
  for (int i = 0; i < NB; ++i) mtot += C_MASS[i];
#pragma unroll
  for (int r = 0; r < 3; ++r) com[r] = pair_sum(acc[r]) / mtot;
}

#undef RF
#undef PS
#undef CM
#undef IN
#undef MS
#undef SD

}  // namespace h1s
