// Analytic dynamics Jacobians A_t = df/dx (51x51), B_t = df/du (51x19) of the H1 step, one wave per knot.
//
// Replaces iLQR::computeLinearization -> RobotUtils::linearizeDynamicsFD (reference
// src/ilqr/ilqr.cpp:126-131, src/common/robot_utils.cpp:120-160: 71 finite-difference steps per knot)
// with exact derivatives in raw coordinates (the quaternion block contains the normalisation
// projector):
//   qacc = Minv (tau - D v - ID(q, v, 0))  =>  d qacc/d z = -Minv ( d ID(q, v, qacc)/d z + D dv/dz )
// * every lane owns one tangent direction z (3 base-rotation, 19 hinge angles, 3 + 3 base velocity,
//   19 hinge rates) and runs a tangent recursive-Newton-Euler sweep down and up the H1 tree with the
//   primal per-body quantities (KnotDump) broadcast from LDS; chains are processed one at a time so
//   only <= 5 tangent body forces are live per lane,
// * the 25x47 tangent generalized forces are multiplied by Minv (25x25) from LDS,
// * each lane then assembles one column of A / B through the integrator
//   (v' = v + h qacc, p' = p + h v'_lin, theta' = theta + h thetadot', quat' = qhat (x) exp(h w')).
#pragma once
#include "h1_aba_reg.h"
#include "h1_aba_split.h"
#include "h1_dynamics_dev.h"

namespace h1 {

// global (HBM) layout of the primal dump of one knot, written by k_lin_primal_r (doubles)
// Every block starts at an even offset and the record has an even size: the primal kernels' pieces (pairs of sin / cos, the
// 6-vectors of a body, the rows of the pelvis inverse) are then 16-byte aligned and leave as dwordx4 stores -- half the store
// instructions of a kernel that spends 78 % of its cycles stalled on store issue (round 4; the record was 493 doubles with odd offsets).
// (Measured and not kept: 16 knots interleaved per line as for the cost quadratics' record -- the primal kernel halves, 304 -> 143 us,
// but the tangent kernels' load turns into a 493-line gather per knot: 1.53 -> 1.82 ms.)
// (round 4, second layout) everything of one body in one 176-byte block -- velocity, acceleration, U_i, 1 / D_i, sin, cos -- written by
// its lane in ONE burst of 16-byte stores when the outward sweep reaches the body: the block's lines complete while they are still in L2
// and leave as full lines (with the fields in arrays of their own every line was touched by up to 20 stores spread over the whole kernel,
// and left L2 partially written: 2.2 x write amplification).  Readers go through the accessors below.
enum { LinDumpG_R0 = 0, LinDumpG_aL = 10, LinDumpG_qacc = 14, LinDumpG_IA0inv = 40, LinDumpG_BODY = 76, LinDumpG_STRIDE = 22, LinDumpG_SIZE = 516 };
DEVFN constexpr int ldg_v(int i, int k = 0) { return LinDumpG_BODY + LinDumpG_STRIDE * i + k; }
DEVFN constexpr int ldg_a(int i, int k = 0) { return LinDumpG_BODY + LinDumpG_STRIDE * i + 6 + k; }
DEVFN constexpr int ldg_U(int i, int k = 0) { return LinDumpG_BODY + LinDumpG_STRIDE * i + 12 + k; }
DEVFN constexpr int ldg_Dinv(int i) { return LinDumpG_BODY + LinDumpG_STRIDE * i + 18; }
DEVFN constexpr int ldg_s(int i) { return LinDumpG_BODY + LinDumpG_STRIDE * i + 19; }
DEVFN constexpr int ldg_c(int i) { return LinDumpG_BODY + LinDumpG_STRIDE * i + 20; }
// e = 6 i + k
DEVFN int ldg_v_lin(int e) { return ldg_v(e / 6, e % 6); }
DEVFN int ldg_a_lin(int e) { return ldg_a(e / 6, e % 6); }
DEVFN int ldg_U_lin(int e) { return ldg_U(e / 6, e % 6); }

#define LIN_NDIR 47   // tangent directions: phi(3) theta(19) v_lin(3) omega(3) thetadot(19)
#define LIN_LD 48     // padded lane stride of the direction arrays
// packed index of the symmetric Minv (row r, column c in either order)
#define MINV_IDX(r, c) ((r) >= (c) ? ((r) * ((r) + 1) / 2 + (c)) : ((c) * ((c) + 1) / 2 + (r)))

// primal quantities of the knot every phase needs (subset of KnotDump)
struct LinDump {
  double R0[9], aL[3], qacc[H1_NV];
  double Rj[H1_NB][9];          // joint rotations child -> parent ([0] unused)
  double v[H1_NB][6];           // body spatial velocities
  double F[H1_NB][6];           // accumulated inverse-dynamics forces (body i and its subtree)
};
struct LinShared {
  LinDump D;
  struct {                       // (a union until round 4: k_lin_tangent2 runs the outward Minv sweeps BESIDE the tangent sweeps)
    struct { double U[H1_NB][6], Dinv[H1_NB], IA0inv[36]; } m;      // Minv sweeps: articulated-body U_i, 1 / D_i, pelvis inverse
    struct { double part[4][6][19], part11[2][19]; } t;             // tangent sweeps: pelvis / torso-hinge shares of the four
  } u;                                                              // chain groups (LL, RL, torso+LA, torso+RA) per group slot
  double Minv[H1_NV * (H1_NV + 1) / 2];   // d qacc / d tau in MuJoCo coordinates: symmetric, lower triangle packed (MINV)
  double dT[H1_NV][LIN_LD];      // tangent generalized forces, then d qacc / d direction
  double Iv[H1_NB][6];           // I_i v_i (momentum of body i), shared by every tangent direction
  double xa[H1_NB][6];           // X_i a_parent(i): the parent's acceleration in body i's frame
  double x[H1_NX], u_[H1_NU];
  double qh[4], qn, e[4], dE[4][3], Hq[3][4];
  double free_u[H1_NU];
  double h;
};

// a x e_ax and e_ax x a for a principal axis
DEVFN void cross_axis(const double* a, int ax, double* o) {   // a x e_ax
  if (ax == 0) { o[0] = 0.0; o[1] = a[2]; o[2] = -a[1]; }
  else if (ax == 1) { o[0] = -a[2]; o[1] = 0.0; o[2] = a[0]; }
  else { o[0] = a[1]; o[1] = -a[0]; o[2] = 0.0; }
}

enum { DIR_PHI = 0, DIR_THETA = 1, DIR_VLIN = 2, DIR_OMEGA = 3, DIR_THETADOT = 4, DIR_NONE = 5 };
DEVFN void lane_direction(int lane, int& kind, int& idx) {
  if (lane < 3) { kind = DIR_PHI; idx = lane; }
  else if (lane < 22) { kind = DIR_THETA; idx = lane - 3 + 1; }       // body index 1..19
  else if (lane < 25) { kind = DIR_VLIN; idx = lane - 22; }
  else if (lane < 28) { kind = DIR_OMEGA; idx = lane - 25; }
  else if (lane < LIN_NDIR) { kind = DIR_THETADOT; idx = lane - 28 + 1; }
  else { kind = DIR_NONE; idx = 0; }
}

// tangent of (v_i, a_i, f_i) of body I given its parent's tangent (pv, pa).  The body index is a template parameter:
// joint axis, offset and inertia are immediates (h1_model_constexpr.h) and no array is indexed at run time.
// (LIM, joint-limit rows: lockc[i] = -1 / h for a hinge the step stops -- its acceleration is prescribed, qacc_i = -v_i / h - lim_k r_i, so the
// direction of its own rate carries d qacc_i = -1 / h through the inverse-dynamics tangent, the direction of its own angle -lim_k = lockc[i] *
// lockc[0] with lockc[0] = h lim_k riding in the pelvis' unused entry -- and 0 for the others)
template <int I, bool LIM = false>
DEVFN void tan_body_fwd(const LinShared& L, int kind, int idx, const double* pv, const double* pa,
                        double* dv, double* da, double* df, const double* lockc = nullptr) {
  const LinDump& D = L.D;
  constexpr int ax = h1c::C_AXIS[I];
  const double* Rj = D.Rj[I];
  const double qd = L.x[H1_NQ + 6 + I - 1];
  const double r[3] = {h1c::C_POS[I][0], h1c::C_POS[I][1], h1c::C_POS[I][2]};
  xf_motion(Rj, r, pv, dv);
  xf_motion(Rj, r, pa, da);
  // the direction's own hinge adds a few terms; they are applied through 0/1 factors instead of lane-divergent
  // branches (one lane per body would take them: 3 exec-masked branches per body break the instruction stream)
  const double mt = (kind == DIR_THETA && idx == I) ? 1.0 : 0.0;
  const double md = (kind == DIR_THETADOT && idx == I) ? 1.0 : 0.0;
  {
    // d(X u)/d theta = -S x (X u)
    double t[3], xa[6];
    h1r::cross_axis<ax>(D.v[I], t);      dv[0] += mt * t[0]; dv[1] += mt * t[1]; dv[2] += mt * t[2];        // -(e x w) = w x e
    h1r::cross_axis<ax>(D.v[I] + 3, t);  dv[3] += mt * t[0]; dv[4] += mt * t[1]; dv[5] += mt * t[2];
#pragma unroll
    for (int k = 0; k < 6; ++k) xa[k] = L.xa[I][k];
    h1r::cross_axis<ax>(xa, t);          da[0] += mt * t[0]; da[1] += mt * t[1]; da[2] += mt * t[2];
    h1r::cross_axis<ax>(xa + 3, t);      da[3] += mt * t[0]; da[4] += mt * t[1]; da[5] += mt * t[2];
  }
  dv[ax] += md;
  if constexpr (LIM) da[ax] += (md + mt * lockc[0]) * lockc[I];      // (lockc[0] = h * lim_k: the angle's direction carries d qacc_i = -lim_k)
  {  // + dv x (S qd)
    double t[3];
    h1r::cross_axis<ax>(dv, t);     da[0] += qd * t[0]; da[1] += qd * t[1]; da[2] += qd * t[2];
    h1r::cross_axis<ax>(dv + 3, t); da[3] += qd * t[0]; da[4] += qd * t[1]; da[5] += qd * t[2];
  }
  {  // + v_i x S
    double t[3];
    h1r::cross_axis<ax>(D.v[I], t);     da[0] += md * t[0]; da[1] += md * t[1]; da[2] += md * t[2];
    h1r::cross_axis<ax>(D.v[I] + 3, t); da[3] += md * t[0]; da[4] += md * t[1]; da[5] += md * t[2];
  }
  double Ida[6], Idv[6], h[6], t1[6], t2[6];
  h1r::inertia_mul<I>(da, Ida); h1r::inertia_mul<I>(dv, Idv);
#pragma unroll
  for (int k = 0; k < 6; ++k) h[k] = L.Iv[I][k];
  h1r::crf(dv, h, t1); h1r::crf(D.v[I], Idv, t2);
#pragma unroll
  for (int k = 0; k < 6; ++k) df[k] = Ida[k] + t1[k] + t2[k];
}
// body I's total tangent force tot -> its generalized force row and the contribution to the parent
template <int I>
DEVFN double tan_body_bwd(const LinShared& L, int kind, int idx, const double* tot, double* parent_acc) {
  const LinDump& D = L.D;
  constexpr int ax = h1c::C_AXIS[I];
  double g[6] = {tot[0], tot[1], tot[2], tot[3], tot[4], tot[5]};
  {   // d(X^T f)/d theta = X^T (S x* F) = X^T (e x n ; e x f), own hinge only (0/1 factor instead of a branch)
    const double mt = (kind == DIR_THETA && idx == I) ? 1.0 : 0.0;
    double t[3];
    h1r::cross_axis<ax>(D.F[I], t);     g[0] -= mt * t[0]; g[1] -= mt * t[1]; g[2] -= mt * t[2];
    h1r::cross_axis<ax>(D.F[I] + 3, t); g[3] -= mt * t[0]; g[4] -= mt * t[1]; g[5] -= mt * t[2];
  }
  const double r[3] = {h1c::C_POS[I][0], h1c::C_POS[I][1], h1c::C_POS[I][2]};
  xf_force_acc(D.Rj[I], r, g, parent_acc);
  return tot[ax];
}

// ---- tangent sweeps grouped by chain ------------------------------------------------------------------------------
// A direction only moves the bodies below its own hinge, and the inverse-dynamics force of a hinge only sees its own
// subtree: the rows of a chain's hinges are nonzero for 19 directions only (9 base directions + the chain's own hinge
// angles and rates; for an arm also the torso's).  So instead of 47 lanes sweeping all 20 bodies, 2 x 19 lanes sweep
// the two legs side by side (5 bodies), then 2 x 19 lanes the torso + the two arms (5 bodies): half the body steps.
// Left / right bodies are mirror images with the same axes; a lane's body constants are `side ? right : left`
// (h1_aba_split.h).  The chains meet at the pelvis (and the arms at the torso hinge): their shares go through
// L.u.t.part / part11 and are added per direction in a fixed order.
DEVFN int dir_lane(int kind, int idx) {   // inverse of lane_direction
  return kind == DIR_PHI ? idx : (kind == DIR_THETA ? 3 + idx - 1 : (kind == DIR_VLIN ? 22 + idx : (kind == DIR_OMEGA ? 25 + idx : 28 + idx - 1)));
}
// direction of slot q (0..18) of a chain group: 9 base directions, then the chain's own hinges
DEVFN void slot_direction(bool arms, bool side, int q, int& kind, int& idx) {
  if (q < 3) { kind = DIR_PHI; idx = q; }
  else if (q < 6) { kind = DIR_VLIN; idx = q - 3; }
  else if (q < 9) { kind = DIR_OMEGA; idx = q - 6; }
  else if (!arms) { const int first = side ? 6 : 1; if (q < 14) { kind = DIR_THETA; idx = first + q - 9; } else { kind = DIR_THETADOT; idx = first + q - 14; } }
  else {
    const int first = side ? 16 : 12;
    if (q == 9) { kind = DIR_THETA; idx = 11; } else if (q == 10) { kind = DIR_THETADOT; idx = 11; }
    else if (q < 15) { kind = DIR_THETA; idx = first + q - 11; } else { kind = DIR_THETADOT; idx = first + q - 15; }
  }
}
// slot of direction (kind, idx) in chain group c (0 LL, 1 RL, 2 torso+LA, 3 torso+RA), -1 if the group does not sweep it
DEVFN int slot_in_group(int c, int kind, int idx) {
  if (kind == DIR_PHI) return idx;
  if (kind == DIR_VLIN) return 3 + idx;
  if (kind == DIR_OMEGA) return 6 + idx;
  const int off = kind == DIR_THETA ? 0 : 1;
  if (c < 2) { const int first = c ? 6 : 1; return (idx >= first && idx < first + 5) ? 9 + 5 * off + (idx - first) : -1; }
  if (idx == 11) return 9 + off;
  const int first = c == 2 ? 12 : 16;
  return (idx >= first && idx < first + 4) ? 11 + 4 * off + (idx - first) : -1;
}
// tangent of the pelvis velocity / acceleration and of the pelvis body's own force for one direction
DEVFN void tan_base(const LinShared& L, int kind, int idx, double* dv0, double* da0) {
  const LinDump& D = L.D;
#pragma unroll
  for (int k = 0; k < 6; ++k) { dv0[k] = 0.0; da0[k] = 0.0; }
  const double* w = D.v[0];       // omega_body
  const double* vO = D.v[0] + 3;  // R0^T v_lin
  if (kind == DIR_PHI) {          // R0 -> R0 (I + [dphi]x): d(R0^T u) = (R0^T u) x dphi
    double t[3];
    cross_axis(vO, idx, t); dv0[3] = t[0]; dv0[4] = t[1]; dv0[5] = t[2];
    cross_axis(D.aL, idx, t);
    double wx[3]; cross3(w, dv0 + 3, wx);
    da0[3] = t[0] - wx[0]; da0[4] = t[1] - wx[1]; da0[5] = t[2] - wx[2];
  } else if (kind == DIR_VLIN) {  // dv_O = R0^T e_k
    dv0[3] = D.R0[3 * idx]; dv0[4] = D.R0[3 * idx + 1]; dv0[5] = D.R0[3 * idx + 2];
    double wx[3]; cross3(w, dv0 + 3, wx);
    da0[3] = -wx[0]; da0[4] = -wx[1]; da0[5] = -wx[2];
  } else if (kind == DIR_OMEGA) { // a0_lin = aL - w x v_O
    dv0[0] = idx == 0 ? 1.0 : 0.0; dv0[1] = idx == 1 ? 1.0 : 0.0; dv0[2] = idx == 2 ? 1.0 : 0.0;
    double t[3]; cross_axis(vO, idx, t);   // v_O x e_k = -(e_k x v_O)
    da0[3] = t[0]; da0[4] = t[1]; da0[5] = t[2];
  }
}
// mirrored-pair versions of tan_body_fwd / tan_body_bwd: body IL on the even group, IR on the odd one
template <int IL, int IR, bool LIM = false>
DEVFN void tan_body_fwd2(const LinShared& L, bool side, int kind, int idx, const double* pv, const double* pa, double* dv, double* da, double* df, const double* lockc = nullptr) {
  const LinDump& D = L.D;
  constexpr int ax = h1c::C_AXIS[IL];
  static_assert(h1c::C_AXIS[IL] == h1c::C_AXIS[IR], "mirror bodies");
  const int i = side ? IR : IL;
  const double* Rj = D.Rj[i];
  const double qd = L.x[H1_NQ + 6 + i - 1];
  const double r[3] = {side ? h1c::C_POS[IR][0] : h1c::C_POS[IL][0], side ? h1c::C_POS[IR][1] : h1c::C_POS[IL][1], side ? h1c::C_POS[IR][2] : h1c::C_POS[IL][2]};
  xf_motion(Rj, r, pv, dv);
  xf_motion(Rj, r, pa, da);
  const double mt = (kind == DIR_THETA && idx == i) ? 1.0 : 0.0;
  const double md = (kind == DIR_THETADOT && idx == i) ? 1.0 : 0.0;
  {
    double t[3], xa[6];
    h1r::cross_axis<ax>(D.v[i], t);      dv[0] += mt * t[0]; dv[1] += mt * t[1]; dv[2] += mt * t[2];
    h1r::cross_axis<ax>(D.v[i] + 3, t);  dv[3] += mt * t[0]; dv[4] += mt * t[1]; dv[5] += mt * t[2];
#pragma unroll
    for (int k = 0; k < 6; ++k) xa[k] = L.xa[i][k];
    h1r::cross_axis<ax>(xa, t);          da[0] += mt * t[0]; da[1] += mt * t[1]; da[2] += mt * t[2];
    h1r::cross_axis<ax>(xa + 3, t);      da[3] += mt * t[0]; da[4] += mt * t[1]; da[5] += mt * t[2];
  }
  dv[ax] += md;
  if constexpr (LIM) da[ax] += (md + mt * lockc[0]) * lockc[i];
  {
    double t[3];
    h1r::cross_axis<ax>(dv, t);     da[0] += qd * t[0]; da[1] += qd * t[1]; da[2] += qd * t[2];
    h1r::cross_axis<ax>(dv + 3, t); da[3] += qd * t[0]; da[4] += qd * t[1]; da[5] += qd * t[2];
  }
  {
    double t[3];
    h1r::cross_axis<ax>(D.v[i], t);     da[0] += md * t[0]; da[1] += md * t[1]; da[2] += md * t[2];
    h1r::cross_axis<ax>(D.v[i] + 3, t); da[3] += md * t[0]; da[4] += md * t[1]; da[5] += md * t[2];
  }
  double Ida[6], Idv[6], h[6], t1[6], t2[6];
  h1s::inertia_mul<IL, IR>(side, da, Ida); h1s::inertia_mul<IL, IR>(side, dv, Idv);
#pragma unroll
  for (int k = 0; k < 6; ++k) h[k] = L.Iv[i][k];
  h1r::crf(dv, h, t1); h1r::crf(D.v[i], Idv, t2);
#pragma unroll
  for (int k = 0; k < 6; ++k) df[k] = Ida[k] + t1[k] + t2[k];
}
template <int IL, int IR>
DEVFN double tan_body_bwd2(const LinShared& L, bool side, int kind, int idx, const double* tot, double* parent_acc) {
  const LinDump& D = L.D;
  constexpr int ax = h1c::C_AXIS[IL];
  const int i = side ? IR : IL;
  double g[6] = {tot[0], tot[1], tot[2], tot[3], tot[4], tot[5]};
  {
    const double mt = (kind == DIR_THETA && idx == i) ? 1.0 : 0.0;
    double t[3];
    h1r::cross_axis<ax>(D.F[i], t);     g[0] -= mt * t[0]; g[1] -= mt * t[1]; g[2] -= mt * t[2];
    h1r::cross_axis<ax>(D.F[i] + 3, t); g[3] -= mt * t[0]; g[4] -= mt * t[1]; g[5] -= mt * t[2];
  }
  const double r[3] = {side ? h1c::C_POS[IR][0] : h1c::C_POS[IL][0], side ? h1c::C_POS[IR][1] : h1c::C_POS[IL][1], side ? h1c::C_POS[IR][2] : h1c::C_POS[IL][2]};
  xf_force_acc(D.Rj[i], r, g, parent_acc);
  return tot[ax];
}
template <int FL, int FR, int LEN> struct TanChain2 {
  // (dv_last / da_last: optional copy of the LAST body's velocity / acceleration tangents -- the foot's, for the contact row)
  template <int K, bool LIM = false> static DEVFN void fwd(const LinShared& L, bool side, int kind, int idx, const double* pv, const double* pa, double (*df)[6],
                                         double* dv_last = nullptr, double* da_last = nullptr, const double* lockc = nullptr) {
    double nv[6], na[6];
    tan_body_fwd2<FL + K, FR + K, LIM>(L, side, kind, idx, pv, pa, nv, na, df[K], lockc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (K + 1 < LEN) fwd<K + 1, LIM>(L, side, kind, idx, nv, na, df, dv_last, da_last, lockc);
    else if (dv_last) {
#pragma unroll
      for (int k = 0; k < 6; ++k) { dv_last[k] = nv[k]; da_last[k] = na[k]; }
    }
  }
  template <int K> static DEVFN void bwd(LinShared& L, bool side, int kind, int idx, double (*df)[6], double* acc, double* dFj, int col) {
    double tot[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { tot[c] = df[K][c] + acc[c]; acc[c] = 0.0; }
    L.dT[5 + (side ? FR : FL) + K][col] = tan_body_bwd2<FL + K, FR + K>(L, side, kind, idx, tot, (K == 0) ? dFj : acc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (K > 0) bwd<K - 1>(L, side, kind, idx, df, acc, dFj, col);
  }
  // sweeps the chain for one direction; adds the chain's force tangent at its root to dFj; rows -> L.dT[.][col]
  template <bool LIM = false>
  static DEVFN void run(LinShared& L, bool side, int kind, int idx, const double* jv, const double* ja, double* dFj, int col, const double* lockc = nullptr) {
    double df[LEN][6];
    fwd<0, LIM>(L, side, kind, idx, jv, ja, df, nullptr, nullptr, lockc);
    double acc[6] = {0, 0, 0, 0, 0, 0};
    bwd<LEN - 1>(L, side, kind, idx, df, acc, dFj, col);
  }
};

// Tangent generalized forces d ID_mj / d direction (+ damping) -> L.dT[:, 0..46], in three steps with a barrier
// between them (lin_tangent_all; the host probe runs each step over all lanes instead).
// step 1: rows of the hinges are written only for the directions of their own chain group: zero the rest
DEVFN void lin_tangent_zero(LinShared& L, int lane) {
  for (int e = lane; e < (H1_NV - 6) * LIN_LD; e += 64) (&L.dT[6][0])[e] = 0.0;
}
// step 2: 2 x 19 lanes sweep the two legs, then torso + the two arms; shares of the pelvis / torso hinge -> L.u.t
DEVFN void lin_tangent_legs(LinShared& L, int lane) {
  const int grp = lane / 19, q = lane - 19 * grp;
  const bool side = grp == 1;
  if (grp < 2) {
    int kind, idx; slot_direction(false, side, q, kind, idx);
    const int col = dir_lane(kind, idx);
    double dv0[6], da0[6]; tan_base(L, kind, idx, dv0, da0);
    double dFj[6] = {0, 0, 0, 0, 0, 0};
    TanChain2<1, 6, 5>::run(L, side, kind, idx, dv0, da0, dFj, col);
#pragma unroll
    for (int k = 0; k < 6; ++k) L.u.t.part[grp][k][q] = dFj[k];
  }
}
DEVFN void lin_tangent_arms(LinShared& L, int lane) {
  const int grp = lane / 19, q = lane - 19 * grp;
  const bool side = grp == 1;
  if (grp < 2) {
    int kind, idx; slot_direction(true, side, q, kind, idx);
    const int col = dir_lane(kind, idx);
    double dv0[6], da0[6]; tan_base(L, kind, idx, dv0, da0);
    double tv[6], ta[6], dF11[6];
    tan_body_fwd<11>(L, kind, idx, dv0, da0, tv, ta, dF11);     // the torso's own force tangent: counted by the left group only
    if (side) {
#pragma unroll
      for (int k = 0; k < 6; ++k) dF11[k] = 0.0;
    }
    TanChain2<12, 16, 4>::run(L, side, kind, idx, tv, ta, dF11, col);
    // torso hinge: its row is the sum of the two groups' shares; the own-hinge correction is linear and applied once (left)
    double dFj[6] = {0, 0, 0, 0, 0, 0};
    L.u.t.part11[grp][q] = tan_body_bwd<11>(L, side ? DIR_NONE : kind, idx, dF11, dFj);
#pragma unroll
    for (int k = 0; k < 6; ++k) L.u.t.part[2 + grp][k][q] = dFj[k];
  }
}
// step 2: 2 x 19 lanes sweep the two legs, then torso + the two arms; shares of the pelvis / torso hinge -> L.u.t
// (k_lin_tangent runs the two passes on its two waves at the same time)
DEVFN void lin_tangent_chains(LinShared& L, int lane) {
  lin_tangent_legs(L, lane);
  lin_tangent_arms(L, lane);
}
// step 3: pelvis: own force tangent + the four chain shares, per direction (lane = direction, fixed order)
DEVFN void lin_tangent_pelvis(LinShared& L, int lane) {
  const LinDump& D = L.D;
  int kind, idx; lane_direction(lane, kind, idx);
  if (kind != DIR_NONE) {
    double dv0[6], da0[6]; tan_base(L, kind, idx, dv0, da0);
    double dF0[6];
    {
      double Ida[6], Idv[6], h[6], t1[6], t2[6];
      h1r::inertia_mul<0>(da0, Ida); h1r::inertia_mul<0>(dv0, Idv);
#pragma unroll
      for (int k = 0; k < 6; ++k) h[k] = L.Iv[0][k];
      h1r::crf(dv0, h, t1); h1r::crf(D.v[0], Idv, t2);
#pragma unroll
      for (int k = 0; k < 6; ++k) dF0[k] = Ida[k] + t1[k] + t2[k];
    }
    // this direction's slot in each chain group (-1: the group does not contain it, its share is zero)
    int qs[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) qs[c] = slot_in_group(c, kind, idx);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double p2 = qs[2] >= 0 ? L.u.t.part[2][k][qs[2] < 0 ? 0 : qs[2]] : 0.0, p3 = qs[3] >= 0 ? L.u.t.part[3][k][qs[3] < 0 ? 0 : qs[3]] : 0.0;
      const double p0 = qs[0] >= 0 ? L.u.t.part[0][k][qs[0] < 0 ? 0 : qs[0]] : 0.0, p1 = qs[1] >= 0 ? L.u.t.part[1][k][qs[1] < 0 ? 0 : qs[1]] : 0.0;
      dF0[k] += ((p2 + p3) + p0) + p1;
    }
    L.dT[5 + 11][lane] = (qs[2] >= 0 ? L.u.t.part11[0][qs[2] < 0 ? 0 : qs[2]] : 0.0) + (qs[3] >= 0 ? L.u.t.part11[1][qs[3] < 0 ? 0 : qs[3]] : 0.0);
    // free joint rows: torque in the body frame, force in the world frame
    double fl[3] = {dF0[3], dF0[4], dF0[5]};
    if (kind == DIR_PHI) {  // d(R0 f) = R0 (dphi x f + df)
      double t[3]; cross_axis(D.F[0] + 3, idx, t);   // F x e_k = -(e_k x F)
      fl[0] -= t[0]; fl[1] -= t[1]; fl[2] -= t[2];
    }
    double fw[3]; mv3(D.R0, fl, fw);
    L.dT[0][lane] = fw[0]; L.dT[1][lane] = fw[1]; L.dT[2][lane] = fw[2];
    L.dT[3][lane] = dF0[0]; L.dT[4][lane] = dF0[1]; L.dT[5][lane] = dF0[2];
    if (kind == DIR_THETADOT) L.dT[5 + idx][lane] += H1_DAMPING;
  }
}

DEVFN void lin_tangent_all(LinShared& L, int lane) {
  lin_tangent_zero(L, lane);
  __syncthreads();
  lin_tangent_chains(L, lane);
  __syncthreads();
  lin_tangent_pelvis(L, lane);
}

// Column c (= lane, 0..24) of Minv = d qacc / d tau in MuJoCo coordinates: response of the articulated-body
// recursion to a unit generalized force on dof c with zero velocity and gravity (the recursion is linear in
// the force): inward sweep of the bias-force increments, pelvis solve, outward sweep of the accelerations.
// The joint-force increments du are nonzero only on the path from the forced hinge to the pelvis (one body per
// tree level): they stay in registers as (body, value) per level.
struct MinvPath { int body[6]; double du[6]; };   // index = tree depth 1..5
template <int FIRST, int LEN> struct MinvChainOut {
  template <int K> static DEVFN void step(LinShared& L, const MinvPath& P, const double* ap, int lane) {
    constexpr int I = FIRST + K, ax = h1c::C_AXIS[I], dep = h1c::C_DEPTH[I];
    const double r[3] = {h1c::C_POS[I][0], h1c::C_POS[I][1], h1c::C_POS[I][2]};
    double a[6]; xf_motion(L.D.Rj[I], r, ap, a);
    double s = (P.body[dep] == I) ? P.du[dep] : 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) s -= L.u.m.U[I][q] * a[q];
    const double qdd = s * L.u.m.Dinv[I];
    a[ax] += qdd;
    if (5 + I >= lane) L.Minv[MINV_IDX(5 + I, lane)] = qdd;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (K + 1 < LEN) step<K + 1>(L, P, a, lane);
  }
};
DEVFN void lin_minv_lane(LinShared& L, int lane) {
  if (lane >= H1_NV) return;
  const int c = lane;
  // inward sweep: a unit force on hinge dof c only loads the bodies on the path from that hinge to the pelvis,
  // so every lane walks its own path (<= 5 bodies, lane-dependent body index) instead of all 19 bodies
  double p0[6] = {0, 0, 0, 0, 0, 0};
  MinvPath P;
#pragma unroll
  for (int d = 0; d < 6; ++d) { P.body[d] = -1; P.du[d] = 0.0; }
  {
    int i = (c >= 6) ? c - 5 : 0;        // current body on the path (0: done)
    double acc[6] = {0, 0, 0, 0, 0, 0};
    bool first = true;
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
  // pelvis: IA0 a0 = f_ext - p0 ; f_ext = (body torque e_{c-3}, R0^T e_c) for the free-joint dofs
  double rhs[6] = {-p0[0], -p0[1], -p0[2], -p0[3], -p0[4], -p0[5]};
  if (c < 3) { rhs[3] += L.D.R0[3 * c]; rhs[4] += L.D.R0[3 * c + 1]; rhs[5] += L.D.R0[3 * c + 2]; }
  else if (c < 6) rhs[c - 3] += 1.0;
  double a0[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) { double s = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s += L.u.m.IA0inv[6 * r + k] * rhs[k];
    a0[r] = s; }
  double lw[3]; mv3(L.D.R0, a0 + 3, lw);
  // column `lane` of Minv: only the entries on and below the diagonal are kept (row r < lane is column r's entry)
#pragma unroll
  for (int r = 0; r < 3; ++r) if (r >= lane) L.Minv[MINV_IDX(r, lane)] = lw[r];
#pragma unroll
  for (int r = 0; r < 3; ++r) if (3 + r >= lane) L.Minv[MINV_IDX(3 + r, lane)] = a0[r];
  double a11[6];
  {  // torso outward
    constexpr int I = 11, ax = h1c::C_AXIS[11];
    const double r[3] = {h1c::C_POS[I][0], h1c::C_POS[I][1], h1c::C_POS[I][2]};
    xf_motion(L.D.Rj[I], r, a0, a11);
    double s = (P.body[1] == I) ? P.du[1] : 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) s -= L.u.m.U[I][q] * a11[q];
    const double qdd = s * L.u.m.Dinv[I];
    a11[ax] += qdd;
    if (5 + I >= lane) L.Minv[MINV_IDX(5 + I, lane)] = qdd;
  }
  MinvChainOut<12, 4>::step<0>(L, P, a11, lane);
  MinvChainOut<16, 4>::step<0>(L, P, a11, lane);
  MinvChainOut<1, 5>::step<0>(L, P, a0, lane);
  MinvChainOut<6, 5>::step<0>(L, P, a0, lane);
}

// d qacc / d direction = -Minv dT  (in place, one lane per direction)
// dT <- -Minv dT (25 x 25 times 25 x 47) on v_mfma_f64_16x16x4_f64: 2 row tiles x 3 column tiles x 7 k-steps = 42 MFMA
// (A operand lane (lr, lk) = Minv[16 I + lr][4 s + lk], B operand lane (lk, lr) = dT[4 s + lk][16 J + lr]; rows / k beyond
// 25 are zero padding), instead of 625 broadcast LDS reads + 625 FMAs per lane.  All 64 lanes take part; every operand is
// read before the first result is written back.
DEVFN void lin_apply_minv_lane(LinShared& L, int lane) {
  typedef double v4d_l __attribute__((ext_vector_type(4)));
  const int lr = lane & 15, lk = lane >> 4;
  double am[2][7], bd[3][7];
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const int k = 4 * s + lk, kc = k < H1_NV ? k : H1_NV - 1;
#pragma unroll
    for (int I = 0; I < 2; ++I) {
      const int r = 16 * I + lr, rc = r < H1_NV ? r : H1_NV - 1;
      const double v = L.Minv[MINV_IDX(rc, kc)];
      am[I][s] = (r < H1_NV && k < H1_NV) ? -v : 0.0;
    }
#pragma unroll
    for (int J = 0; J < 3; ++J) {
      const double v = L.dT[kc][16 * J + lr];
      bd[J][s] = (k < H1_NV) ? v : 0.0;
    }
  }
  __syncthreads();
  v4d_l acc[2][3];
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int J = 0; J < 3; ++J) acc[I][J] = (v4d_l){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 7; ++s)
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = 0; J < 3; ++J) acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[I][s], bd[J][s], acc[I][J], 0, 0, 0);
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * I + 4 * r + lk;
      if (row < H1_NV) {
#pragma unroll
        for (int J = 0; J < 3; ++J) L.dT[row][16 * J + lr] = acc[I][J][r];
      }
    }
}

// cooperative load of the global dump into LDS (all lanes of one wave); rebuilds the joint rotations from
// their sine/cosine and the accumulated inverse-dynamics forces F_i
DEVFN void lin_load_dump(LinShared& L, const double* g, int lane) {
  LinDump& D = L.D;
  for (int e = lane; e < 9; e += 64) D.R0[e] = g[LinDumpG_R0 + e];
  for (int e = lane; e < 3; e += 64) D.aL[e] = g[LinDumpG_aL + e];
  for (int e = lane; e < H1_NV; e += 64) D.qacc[e] = g[LinDumpG_qacc + e];
  for (int e = lane; e < H1_NB * 6; e += 64) { (&D.v[0][0])[e] = g[ldg_v_lin(e)]; (&L.u.m.U[0][0])[e] = g[ldg_U_lin(e)]; }
  for (int e = lane; e < H1_NB; e += 64) L.u.m.Dinv[e] = g[ldg_Dinv(e)];
  for (int e = lane; e < 36; e += 64) L.u.m.IA0inv[e] = g[LinDumpG_IA0inv + e];
  if (lane >= 1 && lane < H1_NB) {
    const int i = lane, a = H1_AXIS[i], b = (a + 1) % 3, d = (a + 2) % 3;
    const double s = g[ldg_s(i)], c = g[ldg_c(i)];
    for (int r = 0; r < 3; ++r) {
      const double fa = H1_RFIX[i][r][a], fb = H1_RFIX[i][r][b], fd = H1_RFIX[i][r][d];
      D.Rj[i][3 * r + a] = fa; D.Rj[i][3 * r + b] = fb * c + fd * s; D.Rj[i][3 * r + d] = fd * c - fb * s;
    }
    double ap[6], xa[6];
    for (int k = 0; k < 6; ++k) ap[k] = g[ldg_a(H1_PARENT[i], k)];
    xf_motion(D.Rj[i], H1_POS[i], ap, xa);
    for (int k = 0; k < 6; ++k) L.xa[i][k] = xa[k];
  }
  if (lane < H1_NB) {
    const int i = lane;
    double v[6], a[6], Iv[6], Ia[6], vIv[6];
    for (int k = 0; k < 6; ++k) { v[k] = g[ldg_v(i, k)]; a[k] = g[ldg_a(i, k)]; }
    inertia_mul(i, v, Iv); inertia_mul(i, a, Ia); crf(v, Iv, vIv);
    for (int k = 0; k < 6; ++k) { D.F[i][k] = Ia[k] + vIv[k]; L.Iv[i][k] = Iv[k]; }
  }
}
// F_parent += X_i^T F_i, leaves first.  Lane = parent body, one pass per tree level (4 .. 0); a parent gathers its
// children in a fixed order (pelvis: legs then torso, torso: left then right arm), so the sums are deterministic.
// Call with all lanes; contains the level barriers.
DEVFN void lin_accumulate_forces(LinShared& L, int lane) {
  const int i = lane;
  int dep = -1, c0 = 0, c1 = 0, c2 = 0;
  if (i < H1_NB) {
    dep = H1_DEPTH[i];
    if (i == 0) { c0 = 1; c1 = 6; c2 = 11; }
    else if (i == 11) { c0 = 12; c1 = 16; }
    else if (i != 5 && i != 10 && i != 15 && i != 19) c0 = i + 1;
  }
  for (int d = 4; d >= 0; --d) {
    if (dep == d && c0) {
      double acc[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] = L.D.F[i][k];
      xf_force_acc(L.D.Rj[c0], H1_POS[c0], L.D.F[c0], acc);
      if (c1) xf_force_acc(L.D.Rj[c1], H1_POS[c1], L.D.F[c1], acc);
      if (c2) xf_force_acc(L.D.Rj[c2], H1_POS[c2], L.D.F[c2], acc);
#pragma unroll
      for (int k = 0; k < 6; ++k) L.D.F[i][k] = acc[k];
    }
    __syncthreads();
  }
}

// ---- two waves per knot (k_lin_tangent): the knot's LDS record is shared, the phases are split between the waves ----------
// wave-local ordering of LDS accesses (one wave executes its LDS instructions in order; this only stops the compiler)
DEVFN void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
// cooperative load by 128 threads; wave 0 rebuilds the joint rotations / parent accelerations, wave 1 the body forces
// xg / ug: the knot's state and control (into L.x, L.u_); returns false, before anything is written, when *flag (if given) is 0
DEVFN bool lin_load_dump2(LinShared& L, const double* g, int tid, const double* xg = nullptr, const double* ug = nullptr, const int* flag = nullptr,
                          const int* flag2 = nullptr) {
  LinDump& D = L.D;
  const int wv = tid >> 6, lane = tid & 63;
  const int f1 = flag ? *flag : 1, f2 = flag2 ? *flag2 : 1;       // (requested together with everything else)
  // Every value this thread needs from the knot's record is requested before the first one is used (indices clamped into the
  // record instead of predicated): written group by group -- load, LDS store, next group -- each group waits out its own HBM
  // round trip, seven of them in a row (this phase was 14 k of the kernel's 68 k cycles).
  const double r0 = g[LinDumpG_R0 + (tid < 9 ? tid : 0)];
  const double al = g[LinDumpG_aL + (tid < 3 ? tid : 0)];
  const double qa = g[LinDumpG_qacc + (tid < H1_NV ? tid : 0)];
  const int ev = tid < H1_NB * 6 ? tid : 0;
  const double vv = g[ldg_v_lin(ev)], uu = g[ldg_U_lin(ev)];
  const double di = g[ldg_Dinv(tid < H1_NB ? tid : 0)];
  const double ia = g[LinDumpG_IA0inv + (tid < 36 ? tid : 0)];
  // wave 0, lanes 1..19: sin / cos of the body's joint and the parent's acceleration; wave 1, lanes 0..19: the body's v and a
  const int i0 = (lane >= 1 && lane < H1_NB) ? lane : 1, i1 = lane < H1_NB ? lane : 0;
  const int ib = wv == 0 ? i0 : i1;
  const int par0 = (i0 == 1 || i0 == 6 || i0 == 11) ? 0 : ((i0 == 12 || i0 == 16) ? 11 : i0 - 1);     // H1_PARENT without the table's round trip
  const double* p6 = g + (wv == 0 ? ldg_a(par0) : ldg_v(i1));
  double w6[6], a6[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { w6[k] = p6[k]; a6[k] = g[ldg_a(ib, k)]; }
  const double s = g[ldg_s(i0)], c = g[ldg_c(i0)];
  double xu = 0.0;
  if (xg) xu = (tid < 64) ? xg[tid < H1_NX ? tid : 0] : ug[(tid - 64) < H1_NU ? tid - 64 : 0];
  if (!(f1 && f2)) return false;
  if (xg) { if (tid < H1_NX) L.x[tid] = xu; if (tid >= 64 && tid < 64 + H1_NU) L.u_[tid - 64] = xu; }
  if (tid < 9) D.R0[tid] = r0;
  if (tid < 3) D.aL[tid] = al;
  if (tid < H1_NV) D.qacc[tid] = qa;
  if (tid < H1_NB * 6) { (&D.v[0][0])[tid] = vv; (&L.u.m.U[0][0])[tid] = uu; }
  if (tid < H1_NB) L.u.m.Dinv[tid] = di;
  if (tid < 36) L.u.m.IA0inv[tid] = ia;
  if (wv == 0 && lane >= 1 && lane < H1_NB) {
    const int i = lane, a = H1_AXIS[i], b = (a + 1) % 3, d = (a + 2) % 3;
    for (int r = 0; r < 3; ++r) {
      const double fa = H1_RFIX[i][r][a], fb = H1_RFIX[i][r][b], fd = H1_RFIX[i][r][d];
      D.Rj[i][3 * r + a] = fa; D.Rj[i][3 * r + b] = fb * c + fd * s; D.Rj[i][3 * r + d] = fd * c - fb * s;
    }
    double xa[6];
    xf_motion(D.Rj[i], H1_POS[i], w6, xa);
    for (int k = 0; k < 6; ++k) L.xa[i][k] = xa[k];
  }
  if (wv == 1 && lane < H1_NB) {
    const int i = lane;
    double Iv[6], Ia[6], vIv[6];
    inertia_mul(i, w6, Iv); inertia_mul(i, a6, Ia); crf(w6, Iv, vIv);
    for (int k = 0; k < 6; ++k) { D.F[i][k] = Ia[k] + vIv[k]; L.Iv[i][k] = Iv[k]; }
  }
  return true;
}
// lin_accumulate_forces on ONE wave (the other one runs the Minv sweeps meanwhile): wave-local ordering, no workgroup barrier
DEVFN void lin_accumulate_forces_w(LinShared& L, int lane) {
  const int i = lane;
  int dep = -1, c0 = 0, c1 = 0, c2 = 0;
  if (i < H1_NB) {
    dep = H1_DEPTH[i];
    if (i == 0) { c0 = 1; c1 = 6; c2 = 11; }
    else if (i == 11) { c0 = 12; c1 = 16; }
    else if (i != 5 && i != 10 && i != 15 && i != 19) c0 = i + 1;
  }
  for (int d = 4; d >= 0; --d) {
    if (dep == d && c0) {
      double acc[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] = L.D.F[i][k];
      xf_force_acc(L.D.Rj[c0], H1_POS[c0], L.D.F[c0], acc);
      if (c1) xf_force_acc(L.D.Rj[c1], H1_POS[c1], L.D.F[c1], acc);
      if (c2) xf_force_acc(L.D.Rj[c2], H1_POS[c2], L.D.F[c2], acc);
#pragma unroll
      for (int k = 0; k < 6; ++k) L.D.F[i][k] = acc[k];
    }
    wave_sync();
  }
}
// dT <- -Minv dT on the MFMA, row tile I = wave index (21 MFMA per wave); contains a workgroup barrier between the operand
// reads and the write-back (in place)
DEVFN void lin_apply_minv_2(LinShared& L, int tid) {
  typedef double v4d_l __attribute__((ext_vector_type(4)));
  const int I = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lk = lane >> 4;
  double am[7], bd[3][7];
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const int k = 4 * s + lk, kc = k < H1_NV ? k : H1_NV - 1;
    const int r = 16 * I + lr, rc = r < H1_NV ? r : H1_NV - 1;
    const double v = L.Minv[MINV_IDX(rc, kc)];
    am[s] = (r < H1_NV && k < H1_NV) ? -v : 0.0;
#pragma unroll
    for (int J = 0; J < 3; ++J) {
      const double w = L.dT[kc][16 * J + lr];
      bd[J][s] = (k < H1_NV) ? w : 0.0;
    }
  }
  __syncthreads();
  v4d_l acc[3];
#pragma unroll
  for (int J = 0; J < 3; ++J) acc[J] = (v4d_l){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 7; ++s)
#pragma unroll
    for (int J = 0; J < 3; ++J) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[s], bd[J][s], acc[J], 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * I + 4 * r + lk;
    if (row < H1_NV) {
#pragma unroll
      for (int J = 0; J < 3; ++J) L.dT[row][16 * J + lr] = acc[J][r];
    }
  }
}

// ---- two knots per four-wave workgroup (k_lin_tangent2, round 4) ---------------------------------------------------------------
// The constraint-free step does not depend on the base's linear velocity (Galilean invariance: a uniform world-frame translation
// velocity moves no force; d f / d v_lin = [h I; 0; I; 0] to 7e-17 on the oracle's forward-mode AD): the three v_lin directions
// are not swept, their columns of dT stay zero and lin_column writes the constant columns.  A chain group then has 16 direction
// slots instead of 19 -- 6 base directions (phi, omega) + its own hinges' angles and rates --, so 2 sides x 16 slots x TWO KNOTS
// fill a wave exactly: wave 0 sweeps the legs of both knots, wave 1 torso + arms of both, with the instruction count one knot
// needed before (38 of 64 lanes active).  The same packing for the Minv columns (2 x 25 lanes), the force accumulation
// (2 x 20), the integrator prologue (2 x 1).  (The stance-constrained step DOES depend on v_lin through the foot's velocity
// constraint: k_lin_tangent_c keeps the 19-slot groups.)
DEVFN void slot_direction2(bool arms, bool side, int q, int& kind, int& idx) {
  if (q < 3) { kind = DIR_PHI; idx = q; }
  else if (q < 6) { kind = DIR_OMEGA; idx = q - 3; }
  else if (!arms) { const int first = side ? 6 : 1; if (q < 11) { kind = DIR_THETA; idx = first + q - 6; } else { kind = DIR_THETADOT; idx = first + q - 11; } }
  else {
    const int first = side ? 16 : 12;
    if (q == 6) { kind = DIR_THETA; idx = 11; } else if (q == 7) { kind = DIR_THETADOT; idx = 11; }
    else if (q < 12) { kind = DIR_THETA; idx = first + q - 8; } else { kind = DIR_THETADOT; idx = first + q - 12; }
  }
}
DEVFN int slot_in_group2(int c, int kind, int idx) {
  if (kind == DIR_PHI) return idx;
  if (kind == DIR_OMEGA) return 3 + idx;
  const int off = kind == DIR_THETA ? 0 : 1;
  if (c < 2) { const int first = c ? 6 : 1; return (idx >= first && idx < first + 5) ? 6 + 5 * off + (idx - first) : -1; }
  if (idx == 11) return 6 + off;
  const int first = c == 2 ? 12 : 16;
  return (idx >= first && idx < first + 4) ? 8 + 4 * off + (idx - first) : -1;
}
// lane = (knot slot, side, direction slot): all 64 lanes sweep
template <bool LIM = false>
DEVFN void lin2_tangent_legs(LinShared* L2, int lane, const double (*lockc2)[H1_NB] = nullptr) {
  LinShared& L = L2[lane >> 5];
  const double* lockc = LIM ? lockc2[lane >> 5] : nullptr;
  const int grp = (lane >> 4) & 1, q = lane & 15;
  const bool side = grp == 1;
  int kind, idx; slot_direction2(false, side, q, kind, idx);
  const int col = dir_lane(kind, idx);
  double dv0[6], da0[6]; tan_base(L, kind, idx, dv0, da0);
  double dFj[6] = {0, 0, 0, 0, 0, 0};
  TanChain2<1, 6, 5>::template run<LIM>(L, side, kind, idx, dv0, da0, dFj, col, lockc);
#pragma unroll
  for (int k = 0; k < 6; ++k) L.u.t.part[grp][k][q] = dFj[k];
}
template <bool LIM = false>
DEVFN void lin2_tangent_arms(LinShared* L2, int lane, const double (*lockc2)[H1_NB] = nullptr) {
  LinShared& L = L2[lane >> 5];
  const double* lockc = LIM ? lockc2[lane >> 5] : nullptr;
  const int grp = (lane >> 4) & 1, q = lane & 15;
  const bool side = grp == 1;
  int kind, idx; slot_direction2(true, side, q, kind, idx);
  const int col = dir_lane(kind, idx);
  double dv0[6], da0[6]; tan_base(L, kind, idx, dv0, da0);
  double tv[6], ta[6], dF11[6];
  tan_body_fwd<11, LIM>(L, kind, idx, dv0, da0, tv, ta, dF11, lockc);     // the torso's own force tangent: counted by the left group only
  if (side) {
#pragma unroll
    for (int k = 0; k < 6; ++k) dF11[k] = 0.0;
  }
  TanChain2<12, 16, 4>::template run<LIM>(L, side, kind, idx, tv, ta, dF11, col, lockc);
  double dFj[6] = {0, 0, 0, 0, 0, 0};
  L.u.t.part11[grp][q] = tan_body_bwd<11>(L, side ? DIR_NONE : kind, idx, dF11, dFj);
#pragma unroll
  for (int k = 0; k < 6; ++k) L.u.t.part[2 + grp][k][q] = dFj[k];
}
// pelvis: own force tangent + the four chain shares, lane = direction (lane_direction numbering; the v_lin lanes 22..24 sit out)
DEVFN void lin2_tangent_pelvis(LinShared& L, int lane) {
  const LinDump& D = L.D;
  int kind, idx; lane_direction(lane, kind, idx);
  if (kind != DIR_NONE && kind != DIR_VLIN) {
    double dv0[6], da0[6]; tan_base(L, kind, idx, dv0, da0);
    double dF0[6];
    {
      double Ida[6], Idv[6], h[6], t1[6], t2[6];
      h1r::inertia_mul<0>(da0, Ida); h1r::inertia_mul<0>(dv0, Idv);
#pragma unroll
      for (int k = 0; k < 6; ++k) h[k] = L.Iv[0][k];
      h1r::crf(dv0, h, t1); h1r::crf(D.v[0], Idv, t2);
#pragma unroll
      for (int k = 0; k < 6; ++k) dF0[k] = Ida[k] + t1[k] + t2[k];
    }
    int qs[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) qs[c] = slot_in_group2(c, kind, idx);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double p2 = qs[2] >= 0 ? L.u.t.part[2][k][qs[2] < 0 ? 0 : qs[2]] : 0.0, p3 = qs[3] >= 0 ? L.u.t.part[3][k][qs[3] < 0 ? 0 : qs[3]] : 0.0;
      const double p0 = qs[0] >= 0 ? L.u.t.part[0][k][qs[0] < 0 ? 0 : qs[0]] : 0.0, p1 = qs[1] >= 0 ? L.u.t.part[1][k][qs[1] < 0 ? 0 : qs[1]] : 0.0;
      dF0[k] += ((p2 + p3) + p0) + p1;
    }
    L.dT[5 + 11][lane] = (qs[2] >= 0 ? L.u.t.part11[0][qs[2] < 0 ? 0 : qs[2]] : 0.0) + (qs[3] >= 0 ? L.u.t.part11[1][qs[3] < 0 ? 0 : qs[3]] : 0.0);
    double fl[3] = {dF0[3], dF0[4], dF0[5]};
    if (kind == DIR_PHI) {  // d(R0 f) = R0 (dphi x f + df)
      double t[3]; cross_axis(D.F[0] + 3, idx, t);   // F x e_k = -(e_k x F)
      fl[0] -= t[0]; fl[1] -= t[1]; fl[2] -= t[2];
    }
    double fw[3]; mv3(D.R0, fl, fw);
    L.dT[0][lane] = fw[0]; L.dT[1][lane] = fw[1]; L.dT[2][lane] = fw[2];
    L.dT[3][lane] = dF0[0]; L.dT[4][lane] = dF0[1]; L.dT[5][lane] = dF0[2];
    if (kind == DIR_THETADOT) L.dT[5 + idx][lane] += H1_DAMPING;
  }
}
// lin_load_dump2 for both knots of a four-wave workgroup (tid 0..255): the plain copies by knot slot tid >> 7 as before; the two
// rebuild roles -- joint rotations + parent accelerations, body forces + momenta: lane = body, 19 / 20 lanes -- packed for both knots
// on waves 0 and 1 (lane = (knot slot, body)) instead of running on all four waves with a third of their lanes.
DEVFN void lin2_load_dump(LinShared* L2, const double* g0, const double* g1, int tid, const double* xg0, const double* ug0, const double* xg1, const double* ug1) {
  const int ks = tid >> 7, t7 = tid & 127, wv = tid >> 6, lane = tid & 63;
  LinShared& L = L2[ks];
  LinDump& D = L.D;
  const double* g = ks ? g1 : g0;
  const double r0 = g[LinDumpG_R0 + (t7 < 9 ? t7 : 0)];
  const double al = g[LinDumpG_aL + (t7 < 3 ? t7 : 0)];
  const double qa = g[LinDumpG_qacc + (t7 < H1_NV ? t7 : 0)];
  const int ev = t7 < H1_NB * 6 ? t7 : 0;
  const double vv = g[ldg_v_lin(ev)], uu = g[ldg_U_lin(ev)];
  const double di = g[ldg_Dinv(t7 < H1_NB ? t7 : 0)];
  const double ia = g[LinDumpG_IA0inv + (t7 < 36 ? t7 : 0)];
  const double* xg = ks ? xg1 : xg0; const double* ug = ks ? ug1 : ug0;
  const double xu = (t7 < 64) ? xg[t7 < H1_NX ? t7 : 0] : ug[(t7 - 64) < H1_NU ? t7 - 64 : 0];
  // rebuild roles on waves 0 / 1: knot slot lane >> 5, body lane & 31
  const int kr = lane >> 5, bi = lane & 31;
  const double* gr = kr ? g1 : g0;
  const int i0 = (bi >= 1 && bi < H1_NB) ? bi : 1, i1 = bi < H1_NB ? bi : 0;
  const int ib = wv == 0 ? i0 : i1;
  const int par0 = (i0 == 1 || i0 == 6 || i0 == 11) ? 0 : ((i0 == 12 || i0 == 16) ? 11 : i0 - 1);
  double w6[6], a6[6], s = 0.0, c = 0.0;
  if (wv < 2) {
    const double* p6 = gr + (wv == 0 ? ldg_a(par0) : ldg_v(i1));
#pragma unroll
    for (int k = 0; k < 6; ++k) { w6[k] = p6[k]; a6[k] = gr[ldg_a(ib, k)]; }
    s = gr[ldg_s(i0)]; c = gr[ldg_c(i0)];
  }
  if (t7 < H1_NX) L.x[t7] = xu;
  if (t7 >= 64 && t7 < 64 + H1_NU) L.u_[t7 - 64] = xu;
  if (t7 < 9) D.R0[t7] = r0;
  if (t7 < 3) D.aL[t7] = al;
  if (t7 < H1_NV) D.qacc[t7] = qa;
  if (t7 < H1_NB * 6) { (&D.v[0][0])[t7] = vv; (&L.u.m.U[0][0])[t7] = uu; }
  if (t7 < H1_NB) L.u.m.Dinv[t7] = di;
  if (t7 < 36) L.u.m.IA0inv[t7] = ia;
  LinShared& Lr = L2[kr];
  if (wv == 0 && bi >= 1 && bi < H1_NB) {
    const int i = bi, a = H1_AXIS[i], b = (a + 1) % 3, d = (a + 2) % 3;
    for (int r = 0; r < 3; ++r) {
      const double fa = H1_RFIX[i][r][a], fb = H1_RFIX[i][r][b], fd = H1_RFIX[i][r][d];
      Lr.D.Rj[i][3 * r + a] = fa; Lr.D.Rj[i][3 * r + b] = fb * c + fd * s; Lr.D.Rj[i][3 * r + d] = fd * c - fb * s;
    }
    double xa[6];
    xf_motion(Lr.D.Rj[i], H1_POS[i], w6, xa);
    for (int k = 0; k < 6; ++k) Lr.xa[i][k] = xa[k];
  }
  if (wv == 1 && bi < H1_NB) {
    const int i = bi;
    double Iv[6], Ia[6], vIv[6];
    inertia_mul(i, w6, Iv); inertia_mul(i, a6, Ia); crf(w6, Iv, vIv);
    for (int k = 0; k < 6; ++k) { Lr.D.F[i][k] = Ia[k] + vIv[k]; Lr.Iv[i][k] = Iv[k]; }
  }
}
// lin_accumulate_forces_w for two knots on one wave: lane = (knot slot, body)
DEVFN void lin2_accumulate_forces_w(LinShared* L2, int lane) {
  LinShared& L = L2[lane >> 5];
  const int i = lane & 31;
  int dep = -1, c0 = 0, c1 = 0, c2 = 0;
  if (i < H1_NB) {
    dep = H1_DEPTH[i];
    if (i == 0) { c0 = 1; c1 = 6; c2 = 11; }
    else if (i == 11) { c0 = 12; c1 = 16; }
    else if (i != 5 && i != 10 && i != 15 && i != 19) c0 = i + 1;
  }
  for (int d = 4; d >= 0; --d) {
    if (dep == d && c0) {
      double acc[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] = L.D.F[i][k];
      xf_force_acc(L.D.Rj[c0], H1_POS[c0], L.D.F[c0], acc);
      if (c1) xf_force_acc(L.D.Rj[c1], H1_POS[c1], L.D.F[c1], acc);
      if (c2) xf_force_acc(L.D.Rj[c2], H1_POS[c2], L.D.F[c2], acc);
#pragma unroll
      for (int k = 0; k < 6; ++k) L.D.F[i][k] = acc[k];
    }
    wave_sync();
  }
}
// lin_minv_lane in two halves (lane = (knot slot, column)): the inward sweep and the pelvis solve, then -- behind the workgroup
// barrier that releases the tangent sweeps of the other waves, so beside them -- the outward sweeps
struct MinvCarry { MinvPath P; double a0[6]; };
DEVFN void lin2_minv_in(LinShared& L, int c, MinvCarry& C) {
  double p0[6] = {0, 0, 0, 0, 0, 0};
  MinvPath& P = C.P;
#pragma unroll
  for (int d = 0; d < 6; ++d) { P.body[d] = -1; P.du[d] = 0.0; }
  {
    int i = (c >= 6) ? c - 5 : 0;        // current body on the path (0: done)
    double acc[6] = {0, 0, 0, 0, 0, 0};
    bool first = true;
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
  if (c < 3) { rhs[3] += L.D.R0[3 * c]; rhs[4] += L.D.R0[3 * c + 1]; rhs[5] += L.D.R0[3 * c + 2]; }
  else if (c < 6) rhs[c - 3] += 1.0;
#pragma unroll
  for (int r = 0; r < 6; ++r) { double s = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s += L.u.m.IA0inv[6 * r + k] * rhs[k];
    C.a0[r] = s; }
  double lw[3]; mv3(L.D.R0, C.a0 + 3, lw);
#pragma unroll
  for (int r = 0; r < 3; ++r) if (r >= c) L.Minv[MINV_IDX(r, c)] = lw[r];
#pragma unroll
  for (int r = 0; r < 3; ++r) if (3 + r >= c) L.Minv[MINV_IDX(3 + r, c)] = C.a0[r];
}
DEVFN void lin2_minv_out(LinShared& L, int c, const MinvCarry& C) {
  const MinvPath& P = C.P;
  double a11[6];
  {  // torso outward
    constexpr int I = 11, ax = h1c::C_AXIS[11];
    const double r[3] = {h1c::C_POS[I][0], h1c::C_POS[I][1], h1c::C_POS[I][2]};
    xf_motion(L.D.Rj[I], r, C.a0, a11);
    double s = (P.body[1] == I) ? P.du[1] : 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) s -= L.u.m.U[I][q] * a11[q];
    const double qdd = s * L.u.m.Dinv[I];
    a11[ax] += qdd;
    if (5 + I >= c) L.Minv[MINV_IDX(5 + I, c)] = qdd;
  }
  MinvChainOut<12, 4>::step<0>(L, P, a11, c);
  MinvChainOut<16, 4>::step<0>(L, P, a11, c);
  MinvChainOut<1, 5>::step<0>(L, P, C.a0, c);
  MinvChainOut<6, 5>::step<0>(L, P, C.a0, c);
}

// uniform integrator quantities (one lane)
DEVFN void lin_prologue(LinShared& L) {
  const double* x = L.x; const double h = L.h;
  const double qn = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  L.qn = qn;
  const double qw = x[3] / qn, qx = x[4] / qn, qy = x[5] / qn, qz = x[6] / qn;
  L.qh[0] = qw; L.qh[1] = qx; L.qh[2] = qy; L.qh[3] = qz;
  const double s2 = 2.0 / qn;
  // dphi = Hq dquat_raw (body-frame rotation tangent of a raw quaternion perturbation)
  L.Hq[0][0] = -qx * s2; L.Hq[0][1] = qw * s2;  L.Hq[0][2] = qz * s2;  L.Hq[0][3] = -qy * s2;
  L.Hq[1][0] = -qy * s2; L.Hq[1][1] = -qz * s2; L.Hq[1][2] = qw * s2;  L.Hq[1][3] = qx * s2;
  L.Hq[2][0] = -qz * s2; L.Hq[2][1] = qy * s2;  L.Hq[2][2] = -qx * s2; L.Hq[2][3] = qw * s2;
  double wn[3];
  for (int k = 0; k < 3; ++k) wn[k] = x[H1_NQ + 3 + k] + h * L.D.qacc[3 + k];
  const double s = (wn[0] * wn[0] + wn[1] * wn[1] + wn[2] * wn[2]) * h * h;
  double c, so, dso;
  if (s < 1e-6) {
    c = 1.0 - s / 8.0 + s * s / 384.0 - s * s * s / 46080.0;
    so = 0.5 - s / 48.0 + s * s / 3840.0 - s * s * s / 645120.0;
    dso = -1.0 / 48.0 + s / 1920.0 - s * s / 215040.0;
  } else {
    const double a = sqrt(s); double sn, cn; sincos(0.5 * a, &sn, &cn);
    c = cn; so = sn / a; dso = (0.25 * c - 0.5 * so) / s;
  }
  L.e[0] = c; L.e[1] = so * h * wn[0]; L.e[2] = so * h * wn[1]; L.e[3] = so * h * wn[2];
  for (int j = 0; j < 3; ++j) {
    L.dE[0][j] = -0.5 * so * h * h * wn[j];
    for (int i = 0; i < 3; ++i) L.dE[1 + i][j] = (i == j ? so * h : 0.0) + h * wn[i] * dso * 2.0 * h * h * wn[j];
  }
  for (int i = 0; i < H1_NU; ++i) L.free_u[i] = (L.u_[i] < H1_CTRLRANGE[i][0] || L.u_[i] > H1_CTRLRANGE[i][1]) ? 0.0 : 1.0;
}

DEVFN void quat_mul(const double* a, const double* b, double* r) {
  r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  r[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  r[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

// column k of A (is_u = 0, k in 0..50) or of B (is_u = 1, k in 0..18), streamed row by row into out(row, value):
// nothing but the three angular rows of d v' survives between rows
// (contact mode: Gc = Mhat^-1 J^T and WUc = the multiplier tangents of the control columns, h1_linearize_contact_dev.h)
template <class Out>
DEVFN void lin_column(const LinShared& L, int is_u, int k, Out&& out, const double (*Gc)[12] = nullptr, const double (*WUc)[20] = nullptr) {
  const double h = L.h;
  double dphi[3] = {0.0, 0.0, 0.0};
  int src = -1;                 // lane of dT holding d qacc / d direction for this column (theta, velocity columns)
  double fu = 0.0;
  if (is_u) fu = L.free_u[k];
  else if (k >= 3 && k < 7) { dphi[0] = L.Hq[0][k - 3]; dphi[1] = L.Hq[1][k - 3]; dphi[2] = L.Hq[2][k - 3]; }
  else if (k >= 7) src = (k < H1_NQ) ? (3 + k - 7) : (22 + k - H1_NQ);
  const int kv = (!is_u && k >= H1_NQ) ? k - H1_NQ : -1;   // velocity column: identity entry of d v'/d v
  double dw[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < H1_NV; ++r) {
    double dq;
    if (is_u) {
      dq = L.Minv[MINV_IDX(r, 6 + k)];
      if (Gc) {
#pragma unroll
        for (int j = 0; j < 12; ++j) dq += Gc[r][j] * WUc[j][k];
      }
      dq *= fu;
    }
    else if (k < 3) dq = 0.0;
    else if (k < 7) dq = L.dT[r][0] * dphi[0] + L.dT[r][1] * dphi[1] + L.dT[r][2] * dphi[2];
    else dq = L.dT[r][src];
    const double dvn = h * dq + ((r == kv) ? 1.0 : 0.0);
    out(H1_NQ + r, dvn);
    if (r < 3) out(r, ((!is_u && k == r) ? 1.0 : 0.0) + h * dvn);
    else if (r < 6) dw[r - 3] = dvn;
    else out(7 + r - 6, ((!is_u && k == 7 + r - 6) ? 1.0 : 0.0) + h * dvn);
  }
  // quaternion rows: d(qhat (x) e) = dqhat (x) e + qhat (x) de
  double dqh[4] = {0, 0, 0, 0}, de[4], t1[4], t2[4];
  if (!is_u && k >= 3 && k < 7) { const double hp[4] = {0.0, 0.5 * dphi[0], 0.5 * dphi[1], 0.5 * dphi[2]}; quat_mul(L.qh, hp, dqh); }
#pragma unroll
  for (int i = 0; i < 4; ++i) de[i] = L.dE[i][0] * dw[0] + L.dE[i][1] * dw[1] + L.dE[i][2] * dw[2];
  quat_mul(dqh, L.e, t1); quat_mul(L.qh, de, t2);
#pragma unroll
  for (int i = 0; i < 4; ++i) out(3 + i, t1[i] + t2[i]);
}

}  // namespace h1
