// K3: cost quadratics -- iLQR::computeCostQuadratics (reference src/ilqr/ilqr.cpp:133-244) with the exact task-term
// Hessians of add{CoM,CoMVel,EEPos,EEVel,Upright,Balance}CostDerivatives (ilqr.cpp:662-800; closed forms of
// derivatives.cpp:525-707, see h1_cost_dev.h).  Two kernels since round 4:
//
//   k_quad_kin         ONE LANE PER KNOT.  Everything of a knot that is a chain of small dependent steps -- the pelvis-frame
//                      kinematics of the URDF tree, the three point sets (whole-body CoM, left / right ankle origin) with their
//                      subtree aggregates, the weighted functionals of the active cost terms -- runs serially in one lane with
//                      the tree unrolled on compile-time body indices: ~50 wave instructions per knot instead of the ~2100 the
//                      lane = body / lane = (set, body) / lane-0 phases of the one-kernel version issued (five tree levels with a
//                      barrier each, 60 lanes looping over ancestor bitmasks, one lane pushing contexts while 127 wait: 37 k of
//                      that kernel's 76 k cycles per knot).  It leaves a 468-double record per knot in HBM, runs of FOUR
//                      consecutive fields of a knot together and four knots per 128-byte line (QREC_RUN below): the reader's gather
//                      is 117 line requests of 32 bytes per knot.  (Until round 6: one field per knot and 16 knots per line -- every
//                      store instruction of the producer wrote whole lines, but the reader asked for 468 lines per knot, and that
//                      work in the memory pipeline was 0.09 ms of its 1.1 per launch; runs of 8 or 16 lose it again on the
//                      producer's side, whose store instructions then touch 32 / 64 lines each.)
//   k_cost_quadratics  ONE TWO-WAVE WORKGROUP PER KNOT, the wide part: record -> LDS, then
//                        1  wave 0, lane = coordinate: Jacobian columns of c and cdot of the three point sets, balance rows,
//                           gradient lx (all per-column: no exchange);  wave 1, lane = (functional, joint): til x z_j, P'_j; lu, luu
//                        2  Hessian.  The first-order (Gauss-Newton) part -- sum_i scale_i J_i^T J_i over the four gradient-carrying
//                           functionals and the four dyads of the balance term -- is one 51 x 16 x 51 product on
//                           v_mfma_f64_16x16x4_f64 (operands straight from the Jacobian rows in LDS); the second-order part exists
//                           in 8 of the 21 pairs of coordinate classes p | quat | theta | v_b | omega_b | thetadot only: those
//                           entries are evaluated block by block into an LDS patch that the owner lanes of the accumulator tiles
//                           add before they store lxx row by row (128-byte runs).
//                      Workgroups are numbered so that the knots sharing the lines of a record group run on one XCD (one L2).
#include <hip/hip_runtime.h>

#include "h1_cost_dev.h"
#include "h1_fast_math.h"
#include "h1_model_constexpr.h"
#include "ilqr_kernels.h"
#include "riccati_pack.h"

using namespace h1;

namespace ilqr {

#ifndef QUAD_WAVES
#define QUAD_WAVES 4   // waves per SIMD requested from the register allocator: 128 registers, no spills (at 5 the operands of the product, live across the patch phase, spill: 1.44 vs 1.19 ms per launch)
#endif
// -DQUAD_STAMP: diagnostic build only -- per-phase cycle counts of the workgroup of knot (0, 0) land in S.J[0..7]
#ifdef QUAD_STAMP
#define QSTAMP(k) { const long long tn_ = clock64(); if (t == 0 && b == 0 && lane == 0) S.J[k] = (double)(tn_ - qlast); qlast = tn_; }
#else
#define QSTAMP(k)
#endif

// bit j of QANC.m[i]: body i is an ancestor of (or is) body j -- built at compile time from the model table
struct QAncTable { unsigned m[H1_NB]; };
constexpr QAncTable make_anc_table() {
  QAncTable T{};
  T.m[0] = (1u << H1_NB) - 1u;
  for (int i = 1; i < H1_NB; ++i) { unsigned v = 0; for (int j = 1; j < H1_NB; ++j) v |= (h1c::C_ANC[i - 1][j - 1] ? 1u : 0u) << j; T.m[i] = v; }
  return T;
}

// ---- the knot record (doubles; written by k_quad_kin, read by k_cost_quadratics) ---------------------------------------------
// Four functionals in fixed slots: 0 = (CoM, position) [+ the position part of the balance term], 1 = (CoM, velocity) [+ its
// velocity part], 2 / 3 = left / right foot (position in swing, velocity in stance).  A slot whose term is off has scale 0 and
// vec = 0: every contribution is linear in them and vanishes (the one-kernel version kept a variable-length list and merged it).
enum {
  QR_R0 = 0,        // [9]  base rotation (Eigen toRotationMatrix polynomial on the raw coefficients)
  QR_D = 9,         // [4][9] dR/dquat_k
  QR_UR = 45,       // [3]  upright residual pieces
  QR_UJ = 48,       // [3][4]
  QR_MFRAC = 60,    // [3]
  QR_HASBAL = 63,   // 1.0 when the balance term is active at this knot
  QR_BETA = 64,     // [3][3]
  QR_GAMMA = 73,    // [3][3]
  QR_VEC = 82,      // [4][3] merged weighted directions (second-order part)
  QR_TIL = 94,      // [4][3] R0^T vec
  QR_DV = 106,      // [4][4][3] D_k^T vec
  QR_GSUM = 154,    // [4][3] scale_i e_i (gradient)
  QR_GSCALE = 166,  // [4]
  QR_BAL = 170,     // [8]
  QR_ZH = 180,      // [19][3] joint axes, pelvis frame (joint j = 1..19 at 3 (j - 1))
  QR_OM = 237,      // [19][3] body angular velocities
  QR_W0 = 294,      // [19][3] w_j of the CoM set
  QR_DG0 = 351,     // [19][3] d gamma / d theta_j of the CoM set
  QR_WF = 408,      // [2][5][3] w_j of the foot sets (their leg's five joints)
  QR_DGF = 438,     // [2][5][3]
  QREC_SIZE = 468,
#ifndef QREC_G
#define QREC_G 4
#endif
  QREC_RUN = QREC_G,                        // consecutive fields of one knot that lie together (a divisor of 16 and of 468)
  QREC_KNOTS = 16 / QREC_RUN,               // knots sharing a 128-byte line
  QREC_LINES = (QREC_SIZE + QREC_RUN - 1) / QREC_RUN    // lines of a group of QREC_KNOTS knots
  // field f of knot k: double (k / QREC_KNOTS) * QREC_LINES * 16 + (f / QREC_RUN) * 16 + (k % QREC_KNOTS) * QREC_RUN + f % QREC_RUN
};
static_assert(16 % QREC_RUN == 0 && QREC_RUN % 2 == 0, "record runs: whole lines, 16-byte pieces");
size_t quad_rec_doubles(size_t knots) { return ((knots + QREC_KNOTS - 1) / QREC_KNOTS) * (size_t)QREC_LINES * 16; }

// ---- k_quad_kin: compile-time model tables ------------------------------------------------------------------------------------
struct QMassTab { double mu[H1_NB], msub[H1_NB], mtot; };
constexpr bool q_anc_or_self(int i, int j) { return i == 0 ? true : (j == 0 ? false : h1c::C_ANC[i - 1][j - 1] != 0); }
constexpr QMassTab make_mass_tab() {
  QMassTab T{};
  double mtot = 0.0;
  for (int k = 0; k < H1_NB; ++k) mtot += h1c::CU_MASS[k];
  T.mtot = mtot;
  for (int i = 0; i < H1_NB; ++i) T.mu[i] = h1c::CU_MASS[i] / mtot;
  for (int i = 0; i < H1_NB; ++i) { double s = 0.0; for (int j = H1_NB - 1; j >= 0; --j) if (q_anc_or_self(i, j)) s += T.mu[j]; T.msub[i] = s; }
  return T;
}
constexpr QMassTab QMASS = make_mass_tab();

struct QB { double R[9], p[3], z[3], Om[3]; };   // body frame in the pelvis frame: rotation, origin, joint axis, angular velocity

template <int I, int r> DEVFN void qk_rj_row(double cs, double sn, double* o) {   // row r of Rfix_I Rot(axis_I, theta)
  constexpr double f0 = h1c::CU_RFIX[I][r][0], f1 = h1c::CU_RFIX[I][r][1], f2 = h1c::CU_RFIX[I][r][2];
  constexpr int ax = h1c::C_AXIS[I];
  if constexpr (ax == 0) { o[0] = f0; o[1] = f1 * cs + f2 * sn; o[2] = f2 * cs - f1 * sn; }
  else if constexpr (ax == 1) { o[0] = f0 * cs - f2 * sn; o[1] = f1; o[2] = f2 * cs + f0 * sn; }
  else { o[0] = f0 * cs + f1 * sn; o[1] = f1 * cs - f0 * sn; o[2] = f2; }
}
// body I from its parent P (h1_cost_dev.h knot_base_kin); q = mu_I (p_I + R_I com_I), its share of the whole-body CoM
template <int I> DEVFN void qk_step(const QB& P, double th, double qd, QB& B, double* q) {
  constexpr int ax = h1c::C_AXIS[I];
  double sn, cs; h1f::sincos_fast(th, &sn, &cs);
  double Rj[9];
  qk_rj_row<I, 0>(cs, sn, Rj); qk_rj_row<I, 1>(cs, sn, Rj + 3); qk_rj_row<I, 2>(cs, sn, Rj + 6);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) B.R[3 * r + c] = P.R[3 * r] * Rj[c] + P.R[3 * r + 1] * Rj[3 + c] + P.R[3 * r + 2] * Rj[6 + c];
  constexpr double px = h1c::CU_POS[I][0], py = h1c::CU_POS[I][1], pz = h1c::CU_POS[I][2];
  constexpr double c0 = h1c::CU_COM[I][0], c1 = h1c::CU_COM[I][1], c2 = h1c::CU_COM[I][2];
  constexpr double mu = QMASS.mu[I];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    B.p[k] = P.p[k] + (P.R[3 * k] * px + P.R[3 * k + 1] * py + P.R[3 * k + 2] * pz);
    B.z[k] = B.R[3 * k + ax];
    B.Om[k] = P.Om[k] + B.z[k] * qd;
    q[k] = mu * (B.p[k] + (B.R[3 * k] * c0 + B.R[3 * k + 1] * c1 + B.R[3 * k + 2] * c2));
  }
}
// One chain of the tree (FIRST .. FIRST + LEN - 1, each body the parent of the next), processed outward then inward by one lane.
//   FOOT = 0: CoM set only; 1 / 2: the chain is a leg and also carries the left / right ankle set (unit mass at the last body)
// h   : in = sum of q over the subtrees hanging off the chain's LAST body (zero for a leaf), out = subtree sum of the chain's root
// sv0 : in = sum of qd_j w_j over the strict descendants of the last body, out = the same over the root's subtree incl. the root
template <int FIRST, int LEN, int FOOT, class Put> struct QChain {
  QB B[LEN]; double q[LEN][3];
  template <int K> DEVFN void out(const QB& par, const double* xg) {
    constexpr int I = FIRST + K;
    qk_step<I>(par, xg[7 + I - 1], xg[H1_NQ + 6 + I - 1], B[K], q[K]);
    if constexpr (K + 1 < LEN) out<K + 1>(B[K], xg);
  }
  template <int K> DEVFN void in(const double* xg, double* h, double* sv0, const double* pee, double* svf, Put& put) {
    constexpr int I = FIRST + K;
    constexpr double msub = QMASS.msub[I];
    const QB& b = B[K];
    double r[3], w[3], a3[3], b3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { h[k] = h[k] + q[K][k]; r[k] = h[k] - msub * b.p[k]; }
    cross(b.z, r, w);
    cross(b.Om, w, a3); cross(b.z, sv0, b3);
    const double qd = xg[H1_NQ + 6 + I - 1];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      put(QR_ZH + 3 * (I - 1) + k, b.z[k]); put(QR_OM + 3 * (I - 1) + k, b.Om[k]);
      put(QR_W0 + 3 * (I - 1) + k, w[k]); put(QR_DG0 + 3 * (I - 1) + k, a3[k] + b3[k]);
      sv0[k] += qd * w[k];
    }
    if constexpr (FOOT != 0) {      // unit mass at the ankle: subtree sum = p_ee, subtree mass 1 along the leg
      double rf[3], wf[3], af[3], bf[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) rf[k] = pee[k] - 1.0 * b.p[k];
      cross(b.z, rf, wf);
      cross(b.Om, wf, af); cross(b.z, svf, bf);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        put(QR_WF + 15 * (FOOT - 1) + 3 * K + k, wf[k]); put(QR_DGF + 15 * (FOOT - 1) + 3 * K + k, af[k] + bf[k]);
        svf[k] += qd * wf[k];
      }
    }
    if constexpr (K > 0) in<K - 1>(xg, h, sv0, pee, svf, put);
  }
};

// One lane per knot.  `rec`: the record buffer (quad_rec_doubles); knot0: index of the view's first knot in it (batch slices).
__global__ void __launch_bounds__(64) k_quad_kin(DevState S, ProblemDev P, const int* list, const int* count, double* rec, long knot0) {
  const int N1 = S.N + 1;
  const unsigned g = blockIdx.x * 64u + threadIdx.x;                    // (32-bit on purpose: a 64-bit division is ~150 scalar instructions)
  const unsigned total = (unsigned)(list ? *count : S.B) * (unsigned)N1;
  if (blockIdx.x * 64u >= total) return;                               // (whole wave past the end)
  // a lane past the end repeats the last knot (same values to the same record: harmless) so that the wave loads cooperatively
  const unsigned gc = g < total ? g : total - 1u;
  const int bs = (int)(gc / (unsigned)N1), t = (int)(gc - (unsigned)bs * (unsigned)N1);
  const int b = list ? list[bs] : bs;
  const bool term = (t == S.N);
  // The wave's 64 states through LDS: read knot-major by one lane each they were 51 loads x 64 different lines; the flattened
  // [64][51] block is fetched 64 consecutive doubles per instruction (consecutive knots are contiguous in xbar) and each lane then
  // reads its own row (odd pitch: no bank conflicts).
  __shared__ double xs[64 * H1_NX];
  __shared__ const double* xrow[64];
  xrow[threadIdx.x] = S.xbar + ((size_t)b * N1 + t) * H1_NX;
  __syncthreads();
  {
    double tmp[H1_NX];
#pragma unroll
    for (int it = 0; it < H1_NX; ++it) {
      const unsigned e = 64u * it + threadIdx.x, r = (e * 5141u) >> 18, c = e - r * (unsigned)H1_NX;       // r = e / 51 for e < 3264
      tmp[it] = xrow[r][c];
    }
#pragma unroll
    for (int it = 0; it < H1_NX; ++it) xs[64 * it + threadIdx.x] = tmp[it];
  }
  __syncthreads();
  const double* xg = xs + threadIdx.x * H1_NX;
  const long kn = knot0 + (long)b * N1 + t;
  double* out = rec + (size_t)(kn / QREC_KNOTS) * ((size_t)QREC_LINES * 16) + (kn % QREC_KNOTS) * QREC_RUN;
  auto put = [&](int f, double v) { out[(f / QREC_RUN) * 16 + f % QREC_RUN] = v; };
  typedef decltype(put) PutT;

  // base: Pinocchio slot order of the quaternion (derivatives.cpp:12-24): xp[3..6] = (qx, qy, qz, qw)
  const double qx = xg[4], qy = xg[5], qz = xg[6], qw = xg[3];
  double R0[9], D[4][9];
  {
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R0[0] = 1 - (tyy + tzz); R0[1] = txy - twz; R0[2] = txz + twy;
    R0[3] = txy + twz; R0[4] = 1 - (txx + tzz); R0[5] = tyz - twx;
    R0[6] = txz - twy; R0[7] = tyz + twx; R0[8] = 1 - (txx + tyy);
    const double qp[4] = {qx, qy, qz, qw};
#pragma unroll
    for (int k = 0; k < 4; ++k) dR_dquat(k, qp, D[k]);
#pragma unroll
    for (int k = 0; k < 9; ++k) put(QR_R0 + k, R0[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 9; ++e) put(QR_D + 9 * k + e, D[k][e]);
    // upright pieces (4 quaternion slots, derivatives.cpp:646-666 labelling)
    const double ua = qx, ub = qy, uc = qz, ud = qw;
    put(QR_UR + 0, 2.0 * (ub * ud + ua * uc)); put(QR_UR + 1, 2.0 * (uc * ud - ua * ub)); put(QR_UR + 2, -2.0 * (ub * ub + uc * uc));
    put(QR_UJ + 0, 2 * uc); put(QR_UJ + 1, 2 * ud); put(QR_UJ + 2, 2 * ua); put(QR_UJ + 3, 2 * ub);
    put(QR_UJ + 4, -2 * ub); put(QR_UJ + 5, -2 * ua); put(QR_UJ + 6, 2 * ud); put(QR_UJ + 7, 2 * uc);
    put(QR_UJ + 8, 0.0); put(QR_UJ + 9, -4 * ub); put(QR_UJ + 10, -4 * uc); put(QR_UJ + 11, 0.0);
  }
  const double vb[3] = {xg[H1_NQ], xg[H1_NQ + 1], xg[H1_NQ + 2]}, wb[3] = {xg[H1_NQ + 3], xg[H1_NQ + 4], xg[H1_NQ + 5]};
  QB pel;
#pragma unroll
  for (int k = 0; k < 9; ++k) pel.R[k] = (k % 4 == 0) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) { pel.p[k] = 0.0; pel.z[k] = 0.0; pel.Om[k] = wb[k]; }

  double h0[3], sw0[3] = {0.0, 0.0, 0.0};          // whole-body: sum of q (-> beta_0), sum of qd_j w_j (-> gamma_0)
  double betaf[2][3], swf[2][3];                 // feet
  {  // pelvis' own share: q_0 = mu_0 (p_0 + R_0 com_0) with R_0 = 1, p_0 = 0
    constexpr double mu = QMASS.mu[0];
    h0[0] = mu * (0.0 + h1c::CU_COM[0][0]); h0[1] = mu * (0.0 + h1c::CU_COM[0][1]); h0[2] = mu * (0.0 + h1c::CU_COM[0][2]);
  }
  {  // torso and arms
    QB tor; double qt[3];
    qk_step<11>(pel, xg[7 + 10], xg[H1_NQ + 6 + 10], tor, qt);
    double ht[3] = {0.0, 0.0, 0.0}, svt[3] = {0.0, 0.0, 0.0};
    {
      QChain<16, 4, 0, PutT> c; c.template out<0>(tor, xg);
      double h[3] = {0.0, 0.0, 0.0}, sv[3] = {0.0, 0.0, 0.0};
      c.template in<3>(xg, h, sv, nullptr, nullptr, put);
#pragma unroll
      for (int k = 0; k < 3; ++k) { ht[k] += h[k]; svt[k] += sv[k]; }
    }
    {
      QChain<12, 4, 0, PutT> c; c.template out<0>(tor, xg);
      double h[3] = {0.0, 0.0, 0.0}, sv[3] = {0.0, 0.0, 0.0};
      c.template in<3>(xg, h, sv, nullptr, nullptr, put);
#pragma unroll
      for (int k = 0; k < 3; ++k) { ht[k] += h[k]; svt[k] += sv[k]; }
    }
    // the torso itself
    constexpr double msub = QMASS.msub[11];
    double r[3], w[3], a3[3], b3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { ht[k] += qt[k]; r[k] = ht[k] - msub * tor.p[k]; }
    cross(tor.z, r, w); cross(tor.Om, w, a3); cross(tor.z, svt, b3);
    const double qd = xg[H1_NQ + 6 + 10];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      put(QR_ZH + 30 + k, tor.z[k]); put(QR_OM + 30 + k, tor.Om[k]); put(QR_W0 + 30 + k, w[k]); put(QR_DG0 + 30 + k, a3[k] + b3[k]);
      svt[k] += qd * w[k];
      h0[k] += ht[k]; sw0[k] += svt[k];
    }
  }
  {  // right leg (ankle set 2), then left leg (ankle set 1)
    QChain<6, 5, 2, PutT> c; c.template out<0>(pel, xg);
    double h[3] = {0.0, 0.0, 0.0}, sv[3] = {0.0, 0.0, 0.0}, svf[3] = {0.0, 0.0, 0.0};
    const double pee[3] = {c.B[4].p[0], c.B[4].p[1], c.B[4].p[2]};
    c.template in<4>(xg, h, sv, pee, svf, put);
#pragma unroll
    for (int k = 0; k < 3; ++k) { h0[k] += h[k]; sw0[k] += sv[k]; betaf[1][k] = pee[k]; swf[1][k] = svf[k]; }
  }
  {
    QChain<1, 5, 1, PutT> c; c.template out<0>(pel, xg);
    double h[3] = {0.0, 0.0, 0.0}, sv[3] = {0.0, 0.0, 0.0}, svf[3] = {0.0, 0.0, 0.0};
    const double pee[3] = {c.B[4].p[0], c.B[4].p[1], c.B[4].p[2]};
    c.template in<4>(xg, h, sv, pee, svf, put);
#pragma unroll
    for (int k = 0; k < 3; ++k) { h0[k] += h[k]; sw0[k] += sv[k]; betaf[0][k] = pee[k]; swf[0][k] = svf[k]; }
  }
  // beta, gamma of the three sets (gamma = mfrac v_b + omega_b x beta + sum_j qd_j w_j)
  constexpr double mfrac0 = QMASS.msub[0];
  double beta[3][3], gamma[3][3];
  const double mfr[3] = {mfrac0, 1.0, 1.0};
#pragma unroll
  for (int s = 0; s < 3; ++s) {
#pragma unroll
    for (int k = 0; k < 3; ++k) beta[s][k] = s == 0 ? h0[k] : betaf[s - 1][k];
    double wxb[3]; cross(wb, beta[s], wxb);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      gamma[s][k] = (mfr[s] * vb[k] + wxb[k]) + (s == 0 ? sw0[k] : swf[s - 1][k]);
      put(QR_BETA + 3 * s + k, beta[s][k]); put(QR_GAMMA + 3 * s + k, gamma[s][k]);
    }
    put(QR_MFRAC + s, mfr[s]);
  }
  // the weighted functionals of the active terms (order of ilqr.cpp:154-181), fixed slots
  double vec[4][3], gsum[4][3], gscale[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { gscale[c] = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { vec[c][k] = 0.0; gsum[c][k] = 0.0; } }
  const int* st = P.stance + b * P.stance_stride + 2 * t;
  const int st0 = st[0], st1 = st[1];
  if (P.w_com > 0.0) {   // CoM position: w ||com - ref||^2
    const double* ref = P.com_ref + b * P.com_ref_stride + t * 3;
    double rb[3]; mv3(R0, beta[0], rb);
    gscale[0] = 2.0 * P.w_com;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double e = mfr[0] * xg[k] + rb[k] - ref[k]; gsum[0][k] = gscale[0] * e; vec[0][k] = gscale[0] * e; }
  }
  if (!term && P.w_com_vel > 0.0) {
    const double* ref = P.com_vel_ref + b * P.com_vel_ref_stride + t * 3;
    double v[3]; mv3(R0, gamma[0], v);
    gscale[1] = 2.0 * P.w_com_vel;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double e = v[k] - ref[k]; gsum[1][k] = gscale[1] * e; vec[1][k] = gscale[1] * e; }
  }
#pragma unroll
  for (int ee = 0; ee < 2; ++ee) {
    const int set = 1 + ee, ste = ee == 0 ? st0 : st1;
    if (P.w_ee_pos > 0.0 && ste != 1) {
      const double* ref = P.ee_ref + b * P.ee_ref_stride + (t * 2 + ee) * 3;
      double rb[3]; mv3(R0, beta[set], rb);
      gscale[2 + ee] = 2.0 * P.w_ee_pos;
#pragma unroll
      for (int k = 0; k < 3; ++k) { const double e = mfr[set] * xg[k] + rb[k] - ref[k]; gsum[2 + ee][k] = gscale[2 + ee] * e; vec[2 + ee][k] = gscale[2 + ee] * e; }
    }
    if (P.w_ee_vel > 0.0 && ste == 1) {
      double e[3]; mv3(R0, gamma[set], e);   // zero target (ilqr.cpp:734)
      gscale[2 + ee] = 2.0 * P.w_ee_vel;
#pragma unroll
      for (int k = 0; k < 3; ++k) { gsum[2 + ee][k] = gscale[2 + ee] * e[k]; vec[2 + ee][k] = gscale[2 + ee] * e[k]; }
    }
  }
  double ps[2];
  double hasbal = 0.0, bal[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (P.w_balance > 0.0 && support_point(P, b, t, ps)) {
    double rb[3], vc[3]; mv3(R0, beta[0], rb); mv3(R0, gamma[0], vc);
    double com[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) com[k] = mfr[0] * xg[k] + rb[k];
    const double gg = 9.81;
    const double om = sqrt(com[2] / gg), om1 = 1.0 / (2.0 * gg * om), om2 = -1.0 / (4.0 * gg * gg * om * om * om);
    const double r0 = com[0] + vc[0] * om - ps[0], r1 = com[1] + vc[1] * om - ps[1];
    const double rv = r0 * vc[0] + r1 * vc[1];
    bal[0] = r0; bal[1] = r1; bal[2] = om; bal[3] = om1; bal[4] = om2; bal[5] = vc[0]; bal[6] = vc[1]; bal[7] = rv;
    hasbal = 1.0;
    const double mu[3] = {r0, r1, om1 * rv}, nu[3] = {om * r0, om * r1, 0.0};
#pragma unroll
    for (int k = 0; k < 3; ++k) { vec[0][k] += P.w_balance * mu[k]; vec[1][k] += P.w_balance * nu[k]; }
  }
  put(QR_HASBAL, hasbal);
#pragma unroll
  for (int k = 0; k < 8; ++k) put(QR_BAL + k, bal[k]);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double til[3]; mtv3(R0, vec[c], til);
    put(QR_GSCALE + c, gscale[c]);
#pragma unroll
    for (int k = 0; k < 3; ++k) { put(QR_VEC + 3 * c + k, vec[c][k]); put(QR_TIL + 3 * c + k, til[k]); put(QR_GSUM + 3 * c + k, gsum[c][k]); }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double dv[3]; mtv3(D[q], vec[c], dv);
#pragma unroll
      for (int k = 0; k < 3; ++k) put(QR_DV + 12 * c + 3 * q + k, dv[k]);
    }
  }
}

// ---- k_cost_quadratics ---------------------------------------------------------------------------------------------------------
#define QS2_R0 3            // second-order patch: rows quat | theta (3..25), columns 3..50, entry (a, b) with a <= b
#define QS2_NR 23
#define QS2_NC 48
#define QS2_SIZE (QS2_NR * QS2_NC - QS2_NR * (QS2_NR - 1) / 2)
#define QS2_IDX(a, bb) (((a) - QS2_R0) * QS2_NC - ((a) - QS2_R0) * ((a) - QS2_R0 - 1) / 2 + ((bb) - (a)))
enum { QJ_C = 0, QJ_V = 3, QJ_F0 = 6, QJ_F1 = 9, QJ_R0 = 12, QJ_R1 = 13, QJ_M = 14, QJ_ROWS = 15 };
enum { QA_I9 = 0, QA_ZERO = 9, QA_UE = 12, QA_NBX = 39, QAUX_SIZE = 66 };   // LDS-only constants of phase 1a, right behind the record
struct QuadLds {
  alignas(16) double rec[QREC_SIZE];         // the knot record
  double aux[QAUX_SIZE];                     // identity, zero vector, mfrac_s e_k, e_k x beta_s: operands of the uniform column formula
  union {
    struct { double J[QJ_ROWS][H1_NX]; };    // Jacobian rows: d c, d cdot of the CoM; the feet's; balance rows jr0, jr1, m
    double S2[QS2_SIZE];                     // second-order part of the entries that have one (packed, QS2_IDX)
  };
  double tz[4][H1_NJ][3];                    // [0]: til_0 x z_j + P'_1,j (the CoM's two functionals merged); [1..3]: til_c x z_j
  double Pp[2][H1_NJ][3];                    // P'_j of the feet's functionals (slots 2, 3)
  double dg[H1_NX];                          // diagonal additions: Q (or Qf) + soft joint-limit penalty
  double lxs[H1_NX];                         // operand layout (riccati_pack.h): the gradient lx, which rides in row / column "aug" of lxx~
  unsigned pk[64];                           // ... and the slot table: state of slot s | row base of that state in the patch << 8
};
static_assert(sizeof(QuadLds) <= 16384, "QuadLds must fit ten two-wave workgroups per CU");
static_assert(offsetof(QuadLds, aux) == sizeof(double) * QREC_SIZE, "aux directly behind rec: one base pointer for the column table");

// Phase 1a as ONE formula: column c of the Jacobian rows of functional (type, point set) = M(c) u(c), M a 3 x 3 block and u a
// 3-vector of the record (or of the LDS constants behind it): offsets from the record base, built at compile time.
//   position type: p -> I . mfrac e_c | quat -> D_k beta | theta_j (on the set) -> R0 w_j | velocity columns -> 0
//   velocity type: p -> 0 | quat -> D_k gamma | theta_j -> R0 dgamma_j | v_b -> R0 . mfrac e_k | omega_b -> R0 (e_k x beta) | thetadot_j -> R0 w_j
// (the divergent four-way class switch this replaces was ~350 instructions per pass, half of them scalar branch bookkeeping)
struct QColTable { unsigned e[2 * 3 * H1_NX]; };
constexpr bool qc_on(int s, int j) { return s == 0 ? true : (s == 1 ? (j >= 1 && j <= 5) : (j >= 6 && j <= 10)); }
constexpr int qc_w(int s, int j) { return s == 0 ? QR_W0 + 3 * (j - 1) : QR_WF + 15 * (s - 1) + 3 * (j - (s == 1 ? 1 : 6)); }
constexpr int qc_dg(int s, int j) { return s == 0 ? QR_DG0 + 3 * (j - 1) : QR_DGF + 15 * (s - 1) + 3 * (j - (s == 1 ? 1 : 6)); }
constexpr QColTable make_col_table() {
  QColTable T{};
  const int I9 = QREC_SIZE + QA_I9, Z = QREC_SIZE + QA_ZERO;
  for (int v = 0; v < 2; ++v)
    for (int s = 0; s < 3; ++s)
      for (int c = 0; c < H1_NX; ++c) {
        int m = I9, u = Z;
        if (c < 3) { if (!v) { m = I9; u = QREC_SIZE + QA_UE + 9 * s + 3 * c; } }
        else if (c < 7) { m = QR_D + 9 * (c - 3); u = (v ? QR_GAMMA : QR_BETA) + 3 * s; }
        else if (c < H1_NQ) { const int j = c - 7 + 1; if (qc_on(s, j)) { m = QR_R0; u = v ? qc_dg(s, j) : qc_w(s, j); } }
        else if (v) {
          const int cv = c - H1_NQ;
          if (cv < 3) { m = QR_R0; u = QREC_SIZE + QA_UE + 9 * s + 3 * cv; }
          else if (cv < 6) { m = QR_R0; u = QREC_SIZE + QA_NBX + 9 * s + 3 * (cv - 3); }
          else { const int j = cv - 6 + 1; if (qc_on(s, j)) { m = QR_R0; u = qc_w(s, j); } }
        }
        T.e[(v * 3 + s) * H1_NX + c] = (unsigned)m | ((unsigned)u << 16);
      }
  return T;
}
__constant__ static const QColTable QCOL = make_col_table();

// Second-order patch, entry lists (compile time).  A: the related joint pairs of (theta, theta) / (theta, thetadot); B: (quat_k, theta_j) /
// (quat_k, thetadot_j).  Each word: everything the lane needs -- joint indices, block, which foot's set also carries the entry, the
// entry's position in the packed patch.  Entries a foot set contributes to come first (A: 80 of 158, B: 80 of 152), so the passes
// past them skip the feet's code altogether.
//   A word: (lo - 1) | (hi - 1) << 5 | isd << 10 | foot << 11 | patch index << 16      (foot: 0 none, 1 left leg, 2 right leg)
//   B word: k | (j - 1) << 5 | isd << 10 | foot << 11 | patch index << 16
struct QPatchTable { unsigned a[160], b[160]; int na, nb, nfa, nfb; };
constexpr QPatchTable make_patch_table() {
  QPatchTable T{};
  const QAncTable A = make_anc_table();
  const int T0 = 7, D0 = H1_NQ + 6, Q0 = 3;
  int n = 0;
  for (int pass = 0; pass < 3; ++pass)            // foot = 1, 2, then 0
    for (int isd = 0; isd < 2; ++isd)
      for (int ja = 1; ja < H1_NB; ++ja)
        for (int jb = (isd ? 1 : ja); jb < H1_NB; ++jb) {
          const bool relat = ((A.m[ja] >> jb) & 1u) || ((A.m[jb] >> ja) & 1u);
          if (!relat) continue;
          const int foot = (ja <= 5 && jb <= 5) ? 1 : ((ja >= 6 && ja <= 10 && jb >= 6 && jb <= 10) ? 2 : 0);
          if (foot != (pass == 0 ? 1 : (pass == 1 ? 2 : 0))) continue;
          const int lo = ja < jb ? ja : jb, hi = ja < jb ? jb : ja;
          const int a = T0 + ja - 1, bb = (isd ? D0 : T0) + jb - 1;
          T.a[n++] = (unsigned)(lo - 1) | ((unsigned)(hi - 1) << 5) | ((unsigned)isd << 10) | ((unsigned)foot << 11) | ((unsigned)QS2_IDX(a, bb) << 16);
        }
  T.na = n; T.nfa = 0;
  for (int i = 0; i < n; ++i) if ((T.a[i] >> 11) & 3u) T.nfa = i + 1;
  n = 0;
  for (int pass = 0; pass < 3; ++pass)
    for (int isd = 0; isd < 2; ++isd)
      for (int k = 0; k < 4; ++k)
        for (int j = 1; j < H1_NB; ++j) {
          const int foot = j <= 5 ? 1 : (j <= 10 ? 2 : 0);
          if (foot != (pass == 0 ? 1 : (pass == 1 ? 2 : 0))) continue;
          T.b[n++] = (unsigned)k | ((unsigned)(j - 1) << 5) | ((unsigned)isd << 10) | ((unsigned)foot << 11) | ((unsigned)QS2_IDX(Q0 + k, (isd ? D0 : T0) + j - 1) << 16);
        }
  T.nb = n; T.nfb = 0;
  for (int i = 0; i < n; ++i) if ((T.b[i] >> 11) & 3u) T.nfb = i + 1;
  return T;
}
__constant__ static const QPatchTable QPATCH = make_patch_table();
static_assert(make_patch_table().na == 158 && make_patch_table().nb == 152 && make_patch_table().nfa == 80 && make_patch_table().nfb == 80, "patch entry lists");

DEVFN double sel3(const double* v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); }   // no dynamic register index
// d2R/dquat_k dquat_l: dR_dquat is linear in q, so this is dR_dquat(k, e_l)
DEVFN void d2R_sel(int k, int l, double* D) {
  const double q[4] = {l == 0 ? 1.0 : 0.0, l == 1 ? 1.0 : 0.0, l == 2 ? 1.0 : 0.0, l == 3 ? 1.0 : 0.0};
  dR_dquat(k, q, D);
}
// joint j (1..19) of point set st carries it: the CoM set everywhere, a foot set along its own leg
DEVFN bool q_on(int st, int j) { return st == 0 ? true : (st == 1 ? (j >= 1 && j <= 5) : (j >= 6 && j <= 10)); }
// w_j / d gamma / d theta_j of point set st in the record (index clamped into the set's range: callers mask with q_on)
DEVFN const double* q_w(const double* rec, int st, int j) {
  if (st == 0) return rec + QR_W0 + 3 * (j - 1);
  const int f = st == 1 ? 1 : 6; int k = j - f; k = k < 0 ? 0 : (k > 4 ? 4 : k);
  return rec + QR_WF + 15 * (st - 1) + 3 * k;
}
DEVFN const double* q_dg(const double* rec, int st, int j) {
  if (st == 0) return rec + QR_DG0 + 3 * (j - 1);
  const int f = st == 1 ? 1 : 6; int k = j - f; k = k < 0 ? 0 : (k > 4 ? 4 : k);
  return rec + QR_DGF + 15 * (st - 1) + 3 * k;
}

// 2c of k_cost_quadratics: the first-order product of this wave's two row tiles (I = WV, WV + 2) and the write-out.  The owner
// lane of an accumulator element (row a = 16 I + 4 r + lk, column bb = 16 J + lr) adds the diagonal term and the second-order
// patch entry of its (unordered) index pair and stores it: for a fixed register the wave writes four rows x 16 consecutive columns.
// Everything that depends on the tile only -- is it on the diagonal, can it hold a patched pair (the smaller index must be a row
// 3..25 of the patch), do all its rows / columns exist -- is decided at compile time, the lane-dependent rest by selects: written
// with an `if` per element the 32 elements of a wave compiled to ~100 exec-masked branches (1140 instructions, 21 k of the
// kernel's 51 k cycles per knot).
typedef double v4d_q __attribute__((ext_vector_type(4)));
DEVFN int q_rowbase(int x) { const int d = x - QS2_R0; return d * QS2_NC - (d * (d - 1)) / 2; }
template <int I, int J>
DEVFN void quad_tile_out(const v4d_q& acc, const double* S2, const double* dg, double* Hg, int lr, int lk, int rbc0, int rbc1) {
  const int bb = 16 * J + lr;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (16 * I + 4 * r >= H1_NX) continue;                     // (compile-time after unrolling: rows beyond 50 do not exist)
    const int a = 16 * I + 4 * r + lk;
    double h = acc[r];
    if (I == J) { const double d = dg[a < H1_NX ? a : 0]; h += (a == bb) ? d : 0.0; }
    if (I < 2 || J < 2) {                                      // the pair's smaller index can be a patched row
      bool cond; int idx;
      if (I > J) { cond = J == 0 ? (lr >= QS2_R0) : (lr < QS2_R0 + QS2_NR - 16); idx = (J == 0 ? rbc0 : rbc1) + (a - bb); }
      else if (I < J) { cond = (I == 0 ? (a >= QS2_R0) : (a < QS2_R0 + QS2_NR)) && bb < H1_NX; idx = q_rowbase(a) + (bb - a); }
      else {
        const bool a_lo = a < bb; const int lo = a_lo ? a : bb, df = a_lo ? bb - a : a - bb;
        cond = I == 0 ? (lo >= QS2_R0) : (lo < QS2_R0 + QS2_NR);
        idx = (a_lo ? q_rowbase(a) : (J == 0 ? rbc0 : rbc1)) + df;
      }
      const double pv = S2[cond ? idx : 0];
      h += cond ? pv : 0.0;
    }
    const bool rows_ok = 16 * I + 4 * r + 3 < H1_NX, cols_ok = 16 * J + 15 < H1_NX;      // (compile-time)
    if (rows_ok && cols_ok) Hg[a * H1_NX + bb] = h;
    else if ((rows_ok || a < H1_NX) && (cols_ok || bb < H1_NX)) Hg[a * H1_NX + bb] = h;
  }
}
// Operand layout (riccati_pack.h; inside a solve with analytic Jacobians): the first-order product runs in SLOT order (the operands
// are picked through the slot table), and tile (I, J), I >= J, is stored as the Riccati kernel's accumulator image: lane-major, the
// lane's four registers contiguous (two 16-byte stores per lane, 2 KB per tile and wave).  The owner lane looks up the states (a, bb)
// of its element; second-order entries exist where the smaller state is a quat | theta coordinate, i.e. anywhere but in tile (2, 2);
// row and column "aug" carry the gradient; padding slots and the control-column slots are written as true zeros.
struct QPkTable { unsigned w[64]; };
constexpr QPkTable make_pk_table() {
  QPkTable T{};
  for (int s_ = 0; s_ < 64; ++s_) {
    const int st = pk_slot_state(s_);
    const int d = st - QS2_R0;
    const int rb = (st >= QS2_R0 && st < QS2_R0 + QS2_NR) ? d * QS2_NC - (d * (d - 1)) / 2 : 0;
    T.w[s_] = (unsigned)st | ((unsigned)rb << 8);
  }
  return T;
}
__constant__ static const QPkTable QPK = make_pk_table();
template <int I, int J>
DEVFN void quad_tile_out_pk(const v4d_q& acc, const double* S2, const double* dg, const double* lxs, const unsigned* pkt, double* Hp, int lane64, int lr, int lk) {
  const unsigned wc = pkt[16 * J + lr];
  const int bb = (int)(wc & 63u), rbc = (int)(wc >> 8);
  v4d_q out;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (I == 1 && r >= 2) { out[r] = 0.0; continue; }         // slots 24..31: padding
    const unsigned wr = pkt[16 * I + 4 * r + lk];
    const int a = (int)(wr & 63u), rbr = (int)(wr >> 8);
    double h = acc[r];
    if (I == J) { const double d = dg[a < H1_NX ? a : 0]; h += (a == bb) ? d : 0.0; }
    if (!(I == 2 && J == 2)) {
      const bool a_lo = a < bb; const int lo = a_lo ? a : bb, hi = a_lo ? bb : a;
      const bool cond = lo >= QS2_R0 && lo < QS2_R0 + QS2_NR && hi < H1_NX;
      const double pv = S2[cond ? (a_lo ? rbr : rbc) + (hi - lo) : 0];
      h += cond ? pv : 0.0;
    }
    constexpr bool mixed_rows = (I == 1) || (I == 3), mixed_cols = (J == 1) || (J == 3);
    if (mixed_rows || mixed_cols) {
      const bool real = a < H1_NX && bb < H1_NX;
      double alt = 0.0;
      if ((I == 1 && r == 1) || J == 1) {                        // row or column "aug"
        const bool ra = a == PK_AUG && bb < H1_NX, ca = bb == PK_AUG && a < H1_NX;
        const double lv = lxs[ra ? bb : (ca ? a : 0)];
        alt = (ra || ca) ? lv : 0.0;
      }
      h = real ? h : alt;
    }
    out[r] = h;
  }
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  v2d_t* tp = reinterpret_cast<v2d_t*>(Hp + pk_l_tile(I, J) * 256) + lane64;        // (pk_l_elem: registers 0, 1 | registers 2, 3)
  tp[0] = (v2d_t){out[0], out[1]}; tp[64] = (v2d_t){out[2], out[3]};
}
// low: 0 whole matrix, 1 the tiles I >= J in the standard layout, 2 the tiles I >= J in the operand layout
template <int WV>
DEVFN void quad_hessian_tiles(const double (&av)[4][2], const double (&bv)[4][4], const double* S2, const double* dg, double* Hg, int low, int lr, int lk,
                              const double* lxs, const unsigned* pkt, int lane64) {
  // row base of this lane's column as the smaller index of a patched pair (clamped into the patch: masked where it is not one)
  const int rbc0 = q_rowbase(lr < QS2_R0 ? QS2_R0 : lr), rbc1 = q_rowbase(16 + (lr < QS2_R0 + QS2_NR - 16 ? lr : QS2_R0 + QS2_NR - 17));
  double* Hp = pk_align(Hg);
#pragma unroll
  for (int Ii = 0; Ii < 2; ++Ii) {
    constexpr int I0 = WV;                                       // row tiles WV and WV + 2
    v4d_q acc[4];
#pragma unroll
    for (int J = 0; J < 4; ++J) acc[J] = (v4d_q){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int J = 0; J < 4; ++J)
        if (!(low && J > I0 + 2 * Ii)) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks][Ii], bv[ks][J], acc[J], 0, 0, 0);      // (wave-uniform)
    if (low == 2) {
      if (Ii == 0) {
        quad_tile_out_pk<I0, 0>(acc[0], S2, dg, lxs, pkt, Hp, lane64, lr, lk);
        if (1 <= I0) quad_tile_out_pk<I0, I0 >= 1 ? 1 : 0>(acc[1], S2, dg, lxs, pkt, Hp, lane64, lr, lk);
      } else {
        quad_tile_out_pk<I0 + 2, 0>(acc[0], S2, dg, lxs, pkt, Hp, lane64, lr, lk);
        quad_tile_out_pk<I0 + 2, 1>(acc[1], S2, dg, lxs, pkt, Hp, lane64, lr, lk);
        quad_tile_out_pk<I0 + 2, 2>(acc[2], S2, dg, lxs, pkt, Hp, lane64, lr, lk);
        if (3 <= I0 + 2) quad_tile_out_pk<I0 + 2, I0 + 2 >= 3 ? 3 : 0>(acc[3], S2, dg, lxs, pkt, Hp, lane64, lr, lk);
      }
      continue;
    }
    if (Ii == 0) {
      quad_tile_out<I0, 0>(acc[0], S2, dg, Hg, lr, lk, rbc0, rbc1);
      if (!(low && 1 > I0)) quad_tile_out<I0, 1>(acc[1], S2, dg, Hg, lr, lk, rbc0, rbc1);
      if (!low) { quad_tile_out<I0, 2>(acc[2], S2, dg, Hg, lr, lk, rbc0, rbc1); quad_tile_out<I0, 3>(acc[3], S2, dg, Hg, lr, lk, rbc0, rbc1); }
    } else {
      quad_tile_out<I0 + 2, 0>(acc[0], S2, dg, Hg, lr, lk, rbc0, rbc1);
      quad_tile_out<I0 + 2, 1>(acc[1], S2, dg, Hg, lr, lk, rbc0, rbc1);
      quad_tile_out<I0 + 2, 2>(acc[2], S2, dg, Hg, lr, lk, rbc0, rbc1);
      if (!(low && 3 > I0 + 2)) quad_tile_out<I0 + 2, 3>(acc[3], S2, dg, Hg, lr, lk, rbc0, rbc1);
    }
  }
}

// Two waves per knot share the knot's LDS record.  `lane` runs over 0..127.
// `lower` (inside a solve whose backward pass is the one-wave Riccati kernel): for the knots t < N only the tiles I >= J of lxx are
// computed and stored -- exactly the ones k_backward_wave loads (load_aug<true>, riccati_wave.hip); lxx is symmetric, the six
// strictly upper 16 x 16 tiles (35 % of its entries, 0.85 GB per launch at B = 4096) were written for nobody.  The terminal knot,
// which that kernel loads whole, and every stage-API call keep the full matrix (ilqr_hip_get_quadratics mirrors the tiles back).
// Workgroup numbering: workgroups are dealt to the 8 XCDs round-robin by index, and 16 consecutive knots share the 128-byte
// lines of a record group; workgroup L therefore takes knot item (L % 8) * ceil(total / 8) + L / 8 -- consecutive items on one XCD.
__global__ void __launch_bounds__(128, QUAD_WAVES) k_cost_quadratics(DevState S, ProblemDev P, int mode, const int* list, const int* count, int lower,
                                                                     const double* recg, long knot0) {
  const int lane = threadIdx.x, wv = lane >> 6;
  const int N = S.N, N1 = N + 1;
  const unsigned total = (unsigned)(list ? *count : S.B) * (unsigned)N1;   // (32-bit on purpose: a 64-bit division is ~150 scalar instructions)
  const unsigned per = (total + 7u) >> 3;
  const unsigned item = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per || item >= total) return;
  const int bs = (int)(item / (unsigned)N1), t = (int)(item - (unsigned)bs * (unsigned)N1);
  int b = bs;
  if (list) { b = list[bs]; mode = MASK_ALL; }     // compacted selection (DevState::order): no per-rollout flags to fetch
  const bool term = (t == N);
  const bool pk = lower == 2;            // operand layout (every knot, the terminal one included)
  __shared__ QuadLds L;
#ifdef QUAD_STAMP
  long long qlast = clock64();
#endif
  const double* xg = S.xbar + ((size_t)b * N1 + t) * H1_NX;
  const double* rec = L.rec;

  // ---- phase 0: the knot's record -> LDS.  The rollout's selection flags are requested together with the knot's data (indices
  // clamped instead of predicated) and tested before anything is written; so is everything phase 1 wants from HBM
  const double* Qd = term ? P.Qf : P.Q;
  const int* stq = P.stance + b * P.stance_stride + 2 * t;      // (wave-uniform: scalar loads)
  const int fvel[2] = {stq[0] == 1, stq[1] == 1};               // stance foot: the velocity functional; swing foot: the position one
  double xv, xrv, qdv, uv = 0.0, urv = 0.0;
  unsigned colw[2], patw[3];
  {
    const int f1 = mode == MASK_ALL ? 1 : S.active[b], f2 = mode == MASK_RETRY ? S.need_retry[b] : 1;
    const long kn = knot0 + (long)b * N1 + t;
    const double* rg = recg + (size_t)(kn / QREC_KNOTS) * ((size_t)QREC_LINES * 16) + (kn % QREC_KNOTS) * QREC_RUN;
    typedef double v2d_q __attribute__((ext_vector_type(2)));
    v2d_q rv[2];                                                              // 16-byte pieces lane and lane + 128 of the record's 234
#pragma unroll
    for (int k = 0; k < 2; ++k) { int f = 2 * (lane + 128 * k); f = f < QREC_SIZE ? f : QREC_SIZE - 2; rv[k] = *reinterpret_cast<const v2d_q*>(rg + (f / QREC_RUN) * 16 + f % QREC_RUN); }
    const int a = lane < H1_NX ? lane : 0;
    xv = xg[a]; xrv = (P.x_ref + b * P.x_ref_stride + t * H1_NX)[a]; qdv = Qd[a];
    const int l1 = lane - 64, iu = (l1 >= 0 && l1 < H1_NU) ? l1 : 0, tu = term ? N - 1 : t;
    if (wv == 1) { uv = S.ubar[((size_t)b * N + tu) * H1_NU + iu]; urv = (P.u_ref + b * P.u_ref_stride + tu * H1_NU)[iu]; }
    // the table words of phases 1a and 2b travel with the knot's data as well (each was a dependent round trip to the constant
    // segment in the middle of its phase: 8 k cycles for a phase of ~40 instructions)
    {
      const int c = (lane & 63) < H1_NX ? (lane & 63) : 0;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int f = 2 * p + wv, s_ = f < 2 ? 0 : f - 1;
        const int isv = (f == 1 || (f >= 2 && fvel[f >= 2 ? f - 2 : 0])) ? 1 : 0;
        colw[p] = QCOL.e[(isv * 3 + s_) * H1_NX + c];
      }
#pragma unroll
      for (int pass = 0; pass < 3; ++pass) {
        const int e = 64 * pass + (lane & 63);
        patw[pass] = wv == 0 ? QPATCH.a[e < 158 ? e : 0] : QPATCH.b[e < 152 ? e : 0];
      }
    }
    const unsigned pkw = QPK.w[lane & 63];
    if (!(f1 && f2)) return;
#pragma unroll
    for (int k = 0; k < 2; ++k) { const int f = 2 * (lane + 128 * k); if (f < QREC_SIZE) *reinterpret_cast<v2d_q*>(&L.rec[f]) = rv[k]; }
    if (lane < 64) L.pk[lane] = pkw;
    // constants of the uniform column formula (QColTable): identity and zero vector; mfrac_s e_k and e_k x beta_s from the lanes that
    // hold mfrac_s (field 60 + s: lane 60 + s, k = 0) and beta_s[j] (field 64 + 3 s + j: lane 64 + 3 s + j, k = 0)
    if (lane < QAUX_SIZE) L.aux[lane] = (lane < 9 && lane % 4 == 0) ? 1.0 : 0.0;     // (everything else starts as zero; the writes below land behind the wave barrier)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane >= QR_MFRAC && lane < QR_MFRAC + 3) { const int s_ = lane - QR_MFRAC; const double mf = L.rec[lane]; L.aux[QA_UE + 9 * s_] = mf; L.aux[QA_UE + 9 * s_ + 4] = mf; L.aux[QA_UE + 9 * s_ + 8] = mf; }
    if (lane >= QR_BETA && lane < QR_BETA + 9) {
      const int s_ = (lane - QR_BETA) / 3, j = (lane - QR_BETA) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      // (e_k x beta)_i = eps_ikj beta_j: (i, k) = (j + 1, j + 2) -> +beta_j, (j + 2, j + 1) -> -beta_j
      const double bj = L.rec[lane];
      L.aux[QA_NBX + 9 * s_ + 3 * j2 + j1] = bj;
      L.aux[QA_NBX + 9 * s_ + 3 * j1 + j2] = -bj;
    }
  }
  __syncthreads();
  QSTAMP(0)

  const int has_bal = __builtin_amdgcn_readfirstlane(rec[QR_HASBAL] != 0.0 ? 1 : 0);

  // ---- phase 1a: the Jacobian rows of the four functionals (h1_cost_dev.h knot_jac_column), lane = coordinate, one functional per
  // wave and pass: wave 0 the CoM's d c then the left foot's, wave 1 the CoM's d cdot then the right foot's (a foot's rows are
  // d cdot in stance, d c in swing).  One 3 x 3 block times one 3-vector per lane and pass, both picked by QColTable.
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int f = 2 * p + wv;                       // (wave-uniform)
    const int s = f < 2 ? 0 : f - 1;
    const int c = lane & 63;
    if (c < H1_NX) {
      const unsigned e = colw[p];
      const double* M = rec + (e & 0xffffu);
      const double* u = rec + (e >> 16);
      double v3[3]; mv3(M, u, v3);
#pragma unroll
      for (int r = 0; r < 3; ++r) L.J[3 * f + r][c] = v3[r];
    }
  }
  __syncthreads();
  QSTAMP(1)
  // ---- phase 1b.  wave 0, lane = coordinate: balance rows, gradient lx, diagonal;  wave 1, lane = (functional, joint): til x z_j,
  //                 P'_j; then lu / luu
  if (wv == 0) {
    if (lane < H1_NX) {
      const int c = lane, a = lane;
      double jj[4][3];                 // this column of the four gradient-carrying Jacobians (slots 0..3)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 3; ++r) jj[i][r] = L.J[3 * i + r][c];
      // balance rows jr0, jr1 and m = om1 (r0 Jv0 + r1 Jv1) + rv om2 jz / 2 (zero rows when the term is off)
      const double* bal = rec + QR_BAL;
      double jr0 = 0.0, jr1 = 0.0, jm = 0.0;
      if (has_bal) {
        const double om = bal[2], om1 = bal[3], jz = jj[0][2];
        jr0 = jj[0][0] + om * jj[1][0] + bal[5] * om1 * jz;
        jr1 = jj[0][1] + om * jj[1][1] + bal[6] * om1 * jz;
        jm = bal[3] * (bal[0] * jj[1][0] + bal[1] * jj[1][1]) + 0.5 * bal[7] * bal[4] * jz;
      }
      L.J[QJ_R0][c] = jr0; L.J[QJ_R1][c] = jr1; L.J[QJ_M][c] = jm;
      // gradient lx
      double g = qdv * (xv - xrv);   // Q acts on the MuJoCo-ordered state
#pragma unroll
      for (int i = 0; i < 4; ++i) { const double* gs = rec + QR_GSUM + 3 * i; g += jj[i][0] * gs[0] + jj[i][1] * gs[1] + jj[i][2] * gs[2]; }
      if (P.w_upright > 0.0 && a >= 3 && a < 7) {
        const double* uJ = rec + QR_UJ; const double* ur = rec + QR_UR;
        g += P.w_upright * (uJ[a - 3] * ur[0] + uJ[4 + a - 3] * ur[1] + uJ[8 + a - 3] * ur[2]);
      }
      if (has_bal) g += P.w_balance * (jr0 * bal[0] + jr1 * bal[1]);
      double dgl = qdv;
      if (a >= 7 && a < H1_NQ) {
        double lo, hi; limit_bounds(H1_JRANGE[a - 7], lo, hi);
        const double q = xv;                           // (hinge slots are not permuted)
        if (q > hi) g += 2.0 * P.w_joint * (q - hi);
        if (q < lo) g += -2.0 * P.w_joint * (lo - q);
        if (q > hi || q < lo) dgl += 2.0 * P.w_joint;
      }
      S.lx[((size_t)b * N1 + t) * H1_NX + a] = g;
      L.dg[a] = dgl;
      L.lxs[a] = g;
    }
  } else {
    const int l1 = lane - 64;
    {
      // lanes 0..18: the CoM's two functionals of joint j (tz[0] = til_0 x z_j + P'_1,j: what the (theta, theta) entries contract with
      // w_hi; tz[1] = til_1 x z_j); lanes 32..50: the feet's (tz[2], tz[3], and P' where the foot's functional is a velocity)
      const int half = l1 >> 5, j = 1 + (l1 & 31);
      if (j <= H1_NJ) {
        const double* zj = rec + QR_ZH + 3 * (j - 1); const double* Oj = rec + QR_OM + 3 * (j - 1);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int c = 2 * half + q;
          const double* til = rec + QR_TIL + 3 * c;
          double tz[3]; cross(til, zj, tz);
          const bool isv = half == 0 ? (q == 1) : (fvel[q] != 0);
          double pp[3] = {0.0, 0.0, 0.0};
          if (isv) { double tO[3], p1[3], p2[3]; cross(til, Oj, tO); cross(tO, zj, p1); cross(tz, Oj, p2); pp[0] = p1[0] - p2[0]; pp[1] = p1[1] - p2[1]; pp[2] = p1[2] - p2[2]; }
          if (half == 0 && q == 0) {
            // slot 0 is a position functional: its tz is only ever added to slot 1's P' -> store the sum once both are known
#pragma unroll
            for (int k = 0; k < 3; ++k) L.tz[0][j - 1][k] = tz[k];
          } else if (half == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { L.tz[1][j - 1][k] = tz[k]; L.tz[0][j - 1][k] += pp[k]; }      // (same lane wrote tz[0] just above)
          } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) { L.tz[c][j - 1][k] = tz[k]; L.Pp[q][j - 1][k] = pp[k]; }
          }
        }
      }
    }
    if (!term && l1 < H1_NU) {
      const double u = uv;
      double g = P.R[l1] * (u - urv), h = P.R[l1];
      double lo, hi; limit_bounds(H1_CTRLRANGE[l1], lo, hi);
      if (u > hi) g += 2.0 * P.w_ctrl * (u - hi);
      if (u < lo) g += -2.0 * P.w_ctrl * (lo - u);
      if (u > hi || u < lo) h += 2.0 * P.w_ctrl;
      S.lu[((size_t)b * N + t) * H1_NU + l1] = g;
      S.luu[((size_t)b * N + t) * H1_NU + l1] = h;
    }
  }
  __syncthreads();
  QSTAMP(2)

  // ---- phase 2: Hessian lxx
  double* Hg = S.lxx + ((size_t)b * N1 + t) * H1_NX * H1_NX;
  const int lr = lane & 15, lk = (lane >> 4) & 3;
  // 2a: operands of the first-order product H1[a][b] = sum_k sA_k RA_k[a] RB_k[b], k = 4 ks + lk:
  //   lk < 3 : row lk of the Jacobian of gradient-carrying functional ks (RA = RB, sA = its scale; 0 when its term is off)
  //   lk = 3 : the balance dyads w (jr0 jr0' + jr1 jr1' + jz m' + m jz'): ks = 0 jr0, 1 jr1, 2 (jz, m), 3 (m, jz)
  // operand of row tile I / column tile J = entry 16 I + lr of the row; everything beyond column 50 and every unused row is
  // a true zero (junk operands slow the fp64 MFMA down tenfold)
  double av[4][2], bv[4][4];            // this wave's two row tiles wv, wv + 2 (balanced when only the tiles I >= J are wanted); all four column tiles
  {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const double* rowJ = L.J[3 * ks + (lk < 3 ? lk : 0)];
      const double* rowA = lk < 3 ? rowJ : (ks == 0 ? L.J[QJ_R0] : (ks == 1 ? L.J[QJ_R1] : (ks == 2 ? L.J[QJ_C + 2] : L.J[QJ_M])));
      const double* rowB = lk < 3 ? rowJ : (ks == 0 ? L.J[QJ_R0] : (ks == 1 ? L.J[QJ_R1] : (ks == 2 ? L.J[QJ_M] : L.J[QJ_C + 2])));
      const double sA = lk < 3 ? rec[QR_GSCALE + ks] : P.w_balance;
      const bool used = lk < 3 ? (sA != 0.0) : (has_bal != 0);
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        const int e = pk ? (int)(L.pk[16 * T + lr] & 63u) : 16 * T + lr, ec = e < H1_NX ? e : H1_NX - 1;      // operand layout: the state of slot 16 T + lr
        const double ra = rowA[ec], rb = rowB[ec];
        const bool ok = used && e < H1_NX;
        bv[ks][T] = ok ? rb : 0.0;
        const double avv = ok ? sA * ra : 0.0;
        if ((T & 1) == wv) av[ks][T >> 1] = avv;          // row tile T belongs to wave T & 1, slot T >> 1
      }
    }
  }
  QSTAMP(3)
  __syncthreads();   // the Jacobian rows are in registers: their storage becomes the second-order patch
  for (int e = lane; e < QS2_SIZE; e += 128) L.S2[e] = 0.0;
  __syncthreads();
  // 2b: second-order part.  Wave 0: the related joint pairs of (theta, theta) / (theta, thetadot) (QPatchTable A, three passes of 64);
  // wave 1: (quat, theta) / (quat, thetadot) (table B, three passes) and the small blocks.  The CoM's two functionals (slot 0 position,
  // slot 1 velocity) are merged in closed form; the feet's (slots 2, 3; type by the stance flag) only touch their own leg's entries,
  // which the tables list first.
  auto patch = [&](int a, int bb, double h) { L.S2[QS2_IDX(a, bb)] = h; };
  const int Q0 = 3, T0 = 7, V0 = H1_NQ, W0 = H1_NQ + 3, NJ = H1_NJ;
  const int cset[4] = {0, 0, 1, 2};
  const int cvel[4] = {0, 1, fvel[0], fvel[1]};
  if (wv == 0) {
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      const int e = 64 * pass + lane;
      if (e < 158) {
        const unsigned pk = patw[pass];
        const int lo = pk & 31, hi = (pk >> 5) & 31;                  // (joint index - 1)
        const bool isd = (pk >> 10) & 1u;
        const double* t0 = L.tz[0][lo]; const double* t1 = L.tz[1][lo];
        const double* w = rec + QR_W0 + 3 * hi; const double* dg = rec + QR_DG0 + 3 * hi;
        // CoM: (theta, theta): (til_0 x z_lo + P'_1,lo) . w_hi + (til_1 x z_lo) . dgamma_hi;  (theta, thetadot): (til_1 x z_lo) . w_hi
        double h = isd ? dot3(t1, w) : dot3(t0, w) + dot3(t1, dg);
        if (64 * pass < 80) {                                          // (compile-time: the entries a foot set carries are the first 80)
          const int foot = (pk >> 11) & 3;
          const int fi = foot == 2 ? 1 : 0, first = fi ? 5 : 0;
          int hk = hi - first; hk = hk < 0 ? 0 : (hk > 4 ? 4 : hk);
          const double* tf = L.tz[2 + fi][lo]; const double* pf = L.Pp[fi][lo];
          const double* wf = rec + QR_WF + 15 * fi + 3 * hk; const double* dgf = rec + QR_DGF + 15 * fi + 3 * hk;
          const bool fv = fi ? (fvel[1] != 0) : (fvel[0] != 0);
          const double d1 = dot3(tf, wf), d2 = dot3(pf, wf) + dot3(tf, dgf);
          const double v = fv ? (isd ? d1 : d2) : (isd ? 0.0 : d1);
          h += foot ? v : 0.0;
        }
        L.S2[pk >> 16] = h;
      }
    }
  } else {
    const int l1 = lane - 64;
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      const int e = 64 * pass + l1;
      if (e < 152) {
        const unsigned pk = patw[pass];
        const int k = pk & 3, j = (pk >> 5) & 31;                     // (joint index - 1)
        const bool isd = (pk >> 10) & 1u;
        const double* w = rec + QR_W0 + 3 * j; const double* dg = rec + QR_DG0 + 3 * j;
        const double* D0v = rec + QR_DV + 3 * k; const double* D1v = rec + QR_DV + 12 + 3 * k;
        // CoM: (quat, theta): Dv_0 . w_j + Dv_1 . dgamma_j;  (quat, thetadot): Dv_1 . w_j
        double h = isd ? dot3(D1v, w) : dot3(D0v, w) + dot3(D1v, dg);
        if (64 * pass < 80) {
          const int foot = (pk >> 11) & 3;
          const int fi = foot == 2 ? 1 : 0, first = fi ? 5 : 0;
          int jk = j - first; jk = jk < 0 ? 0 : (jk > 4 ? 4 : jk);
          const double* Dfv = rec + QR_DV + 12 * (2 + fi) + 3 * k;
          const double* wf = rec + QR_WF + 15 * fi + 3 * jk; const double* dgf = rec + QR_DGF + 15 * fi + 3 * jk;
          const bool fv = fi ? (fvel[1] != 0) : (fvel[0] != 0);
          const double d1 = dot3(Dfv, wf), d2 = dot3(Dfv, dgf);
          const double v = fv ? (isd ? d1 : d2) : (isd ? 0.0 : d1);
          h += foot ? v : 0.0;
        }
        L.S2[pk >> 16] = h;
      }
    }
    // (theta, omega_b): (w_j x til)_c, velocity functionals only
    if (l1 < 3 * NJ) {
      const int ja = 1 + l1 / 3, cc = l1 % 3;
      double h = 0.0;
#pragma unroll
      for (int c = 1; c < 4; ++c) {
        if (!cvel[c]) continue;
        const int st_ = cset[c];
        double tv[3]; cross(q_w(rec, st_, ja), rec + QR_TIL + 3 * c, tv);
        h += q_on(st_, ja) ? sel3(tv, cc) : 0.0;
      }
      patch(T0 + ja - 1, W0 + cc, h);
    }
    // (quat, v_b), (quat, omega_b) on lanes 0..23, (quat, quat) on lanes 32..41
    if (l1 < 24) {
      const bool isw = l1 >= 12;
      const int e = isw ? l1 - 12 : l1;
      const int k = e / 3, cc = e % 3;
      double h = 0.0;
#pragma unroll
      for (int c = 1; c < 4; ++c) {
        if (!cvel[c]) continue;
        const int st_ = cset[c];
        const double* Dv = rec + QR_DV + 12 * c + 3 * k;
        if (!isw) h += Dv[cc] * rec[QR_MFRAC + st_];
        else { double tv[3]; cross(rec + QR_BETA + 3 * st_, Dv, tv); h += sel3(tv, cc); }   // Dv . (-[beta]x e_c) = (beta x Dv)_c
      }
      patch(Q0 + k, (isw ? W0 : V0) + cc, h);
    } else if (l1 >= 32 && l1 < 42) {
      // (quat, quat): d2R/dq2 terms + upright; triangular index -> (ka, kb), ka <= kb < 4
      const int idx = l1 - 32;
      const int ka = idx < 4 ? 0 : (idx < 7 ? 1 : (idx < 9 ? 2 : 3));
      const int kb = idx < 4 ? idx : (idx < 7 ? idx - 3 : (idx < 9 ? idx - 5 : 3));
      double h = 0.0;
      double D2[9]; d2R_sel(ka, kb, D2);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int st_ = cset[c];
        double tv[3]; mv3(D2, cvel[c] ? rec + QR_GAMMA + 3 * st_ : rec + QR_BETA + 3 * st_, tv);
        h += dot3(rec + QR_VEC + 3 * c, tv);
      }
      if (P.w_upright > 0.0) {
        const int i = ka, j = kb;
        const double* uJ = rec + QR_UJ; const double* ur = rec + QR_UR;
        double v = uJ[i] * uJ[j] + uJ[4 + i] * uJ[4 + j] + uJ[8 + i] * uJ[8 + j];
        if ((i == 0 && j == 2) || (i == 1 && j == 3)) v += 2.0 * ur[0];
        if (i == 2 && j == 3) v += 2.0 * ur[1];
        if (i == 0 && j == 1) v += -2.0 * ur[1];
        if ((i == 1 && j == 1) || (i == 2 && j == 2)) v += -4.0 * ur[2];
        h += P.w_upright * v;
      }
      patch(Q0 + ka, Q0 + kb, h);
    }
  }
  __syncthreads();
  QSTAMP(4)
  // 2c: first-order product row tile by row tile; the owner lane of an accumulator element (row 16 I + 4 r + lk, column
  // 16 J + lr) adds the diagonal terms and the patch entry of its (unordered) index pair and stores it: for a fixed
  // register the wave writes four rows x 16 consecutive columns
  const int low = pk ? 2 : ((lower && !term) ? 1 : 0);
  if (wv == 0) quad_hessian_tiles<0>(av, bv, L.S2, L.dg, Hg, low, lr, lk, L.lxs, L.pk, lane & 63);
  else quad_hessian_tiles<1>(av, bv, L.S2, L.dg, Hg, low, lr, lk, L.lxs, L.pk, lane & 63);
  QSTAMP(5)
}

void launch_cost_quadratics(const DevState& S, const ProblemDev& P, int mode, hipStream_t st, int iter, int lower, const WorkList* wl) {
  const WorkList w = wl ? *wl : work_list(S, mode, iter);
  const long knots = (long)S.B * (S.N + 1);
  // the per-knot kinematics run for every rollout a masked (stage-API) launch might select; inside a solve for the compacted list
  hipLaunchKernelGGL(k_quad_kin, dim3((unsigned)((knots + 63) / 64)), dim3(64), 0, st, S, P, w.list, w.count, S.quad_rec, S.quad_knot0);
  // one workgroup per knot; the grid is padded to a multiple of 8 so that every XCD's share (L % 8) has ceil(total / 8) slots
  const long per = (knots + 7) / 8;
  hipLaunchKernelGGL(k_cost_quadratics, dim3((unsigned)(per * 8)), dim3(128), 0, st, S, P, mode, w.list, w.count, lower, (const double*)S.quad_rec, S.quad_knot0);
}

}  // namespace ilqr
