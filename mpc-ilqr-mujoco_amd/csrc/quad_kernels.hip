// K3: cost quadratics -- iLQR::computeCostQuadratics (reference src/ilqr/ilqr.cpp:133-244) with the exact task-term
// Hessians of add{CoM,CoMVel,EEPos,EEVel,Upright,Balance}CostDerivatives (ilqr.cpp:662-800; closed forms of
// derivatives.cpp:525-707, see h1_cost_dev.h).  One two-wave workgroup per (knot, rollout); every phase is lane-parallel:
//   1  pelvis-frame kinematics of the URDF tree, lane = body, one barrier per tree level (5 levels)
//   2  point sets (whole-body CoM, left / right ankle origin), lane = (set, body): subtree aggregates are direct
//      sums over the ancestor bitmask instead of an inward sweep, so there is no sequential dependency
//   3  Jacobian columns of c and cdot, lane = coordinate
//   4  weighted functionals (one per active cost term) and their per-joint vectors til x z_j, P'_j, lane = (term, joint)
//   5  gradient, lane = coordinate
//   6  Hessian.  The first-order (Gauss-Newton) part -- sum_i scale_i J_i^T J_i over the <= 4 gradient-carrying functionals and
//      the four dyads of the balance term -- is one 51 x 16 x 51 product on v_mfma_f64_16x16x4_f64 (4 k-steps x 16 tiles,
//      operands straight from the Jacobian rows in LDS); the second-order part exists in 8 of the 21 pairs of coordinate
//      classes p | quat | theta | v_b | omega_b | thetadot only: those entries are evaluated block by block (every lane of
//      an iteration runs the same formula) into an LDS patch that the owner lanes of the accumulator tiles add before they
//      store lxx row by row (128-byte runs; the version before this one evaluated 12 + 4 products per entry on the vector
//      ALU behind 32 LDS reads and stored every entry twice at scattered addresses: 60 k of the kernel's 107 k cycles).
#include <hip/hip_runtime.h>

#include "h1_cost_dev.h"
#include "h1_fast_math.h"
#include "h1_model_constexpr.h"
#include "ilqr_kernels.h"

using namespace h1;

namespace ilqr {

#ifndef QUAD_WAVES
#define QUAD_WAVES 4   // waves per SIMD requested from the register allocator (8 two-wave workgroups per CU)
#endif
// -DQUAD_STAMP: diagnostic build only -- per-phase cycle counts of workgroup (0, 0) land in S.J[0..7]
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
__constant__ static const QAncTable QANC = make_anc_table();

// (theta, theta) upper-triangle and (theta, thetadot) full-block entries with a second-order part: the pairs of joints one of
// which is an ancestor of (or is) the other -- 59 + 99 of the 190 + 361 entries.  Packed: ja | jb << 5 | isd << 10.
struct QRelTable { unsigned short e[192]; int n; };
constexpr QRelTable make_rel_table() {
  QRelTable T{};
  int n = 0;
  const QAncTable A = make_anc_table();
  for (int ja = 1; ja < H1_NB; ++ja)
    for (int jb = ja; jb < H1_NB; ++jb)
      if ((A.m[ja] >> jb) & 1u) T.e[n++] = (unsigned short)(ja | (jb << 5));
  for (int ja = 1; ja < H1_NB; ++ja)
    for (int jb = 1; jb < H1_NB; ++jb)
      if (((A.m[ja] >> jb) & 1u) || ((A.m[jb] >> ja) & 1u)) T.e[n++] = (unsigned short)(ja | (jb << 5) | (1 << 10));
  T.n = n;
  return T;
}
__constant__ static const QRelTable QREL = make_rel_table();
static_assert(make_rel_table().n == 158, "related joint pairs of the H1 tree");

#define QMAXC 6   // at most: CoM pos, CoM vel, one functional per foot (swing: position / stance: velocity), 2 balance
#define QMAXM 4   // after the merge of contexts on the same (point set, type): (CoM, pos), (CoM, vel), one per foot
struct QuadCtx { int set, is_vel; double scale; double vec[3], til[3], Dv[4][3]; };
// 16,2xx B of LDS (round 3; 20,4xx before): ten two-wave workgroups per CU instead of eight -- the kernel is a chain of
// dependent lane-parallel phases (62 % of its wave cycles parked on a wait), so what it gains is workgroups in flight.  Three
// cuts: (i) a foot's point set carries ONE gradient-carrying functional per knot -- its position in swing, its velocity in
// stance -- so the feet keep one Jacobian (Jf) instead of d c and d cdot both; (ii) at most four contexts survive the merge;
// (iii) the second-order patch is packed (entry (a, b), a <= b: row a holds its columns a..50 only).
// Phase-1/2 temporaries share storage with the per-joint vectors of phase 4.
#define QS2_R0 3            // second-order patch: rows quat | theta (3..25), columns 3..50, entry (a, b) with a <= b
#define QS2_NR 23
#define QS2_NC 48
#define QS2_SIZE (QS2_NR * QS2_NC - QS2_NR * (QS2_NR - 1) / 2)
#define QS2_IDX(a, bb) (((a) - QS2_R0) * QS2_NC - ((a) - QS2_R0) * ((a) - QS2_R0 - 1) / 2 + ((bb) - (a)))
struct QuadLds {
  union {
    struct {                               // phases 0..5 and the operand fetch of phase 6
      double Jc[3][H1_NX], Jv[3][H1_NX];     // d c / d x_p, d cdot / d x_p of the whole-body CoM (point set 0)
      double Jf[2][3][H1_NX];                // the feet (point sets 1, 2): d cdot / d x_p in stance, d c / d x_p in swing
      double jr[2][H1_NX];                   // balance rows jr0, jr1 (jz, Jv0, Jv1 are rows of Jc[0] / Jv[0])
      double xp[H1_NX];                      // Pinocchio-ordered state (derivatives.cpp:12-24)
      double R0[9], D[4][9];                 // base rotation (Eigen toRotationMatrix polynomial), dR/dquat_k
      double zh[H1_NB][3], Om[H1_NB][3];     // joint axes and body angular velocities, pelvis frame
      double us[H1_NU];
      double gsum[QMAXC][3];
    };
    double S2[QS2_SIZE];                   // phase 6: second-order part of the entries that have one (packed, QS2_IDX)
  };
  double beta[3][3], gamma[3][3], mfrac[3];
  double w[3][H1_NB][3], dgam[3][H1_NB][3];
  unsigned char on[3][H1_NB];
  QuadCtx ctx[QMAXC];
  double dg[H1_NX];                      // diagonal additions: Q (or Qf) + soft joint-limit penalty
  double gscale[QMAXM];                  // scale of the gradient-carrying functionals (first-order product): CoM position, CoM velocity, one per foot
  unsigned char gset[8], gvel[8]; int nctx, ng, has_bal; double bal[8];
  double uJ[3][4], ur[3];                // upright pieces
  unsigned anc[H1_NB];                   // bit j of anc[i]: body i is an ancestor of (or is) body j
  union {
    struct { double Rh[H1_NB][9], ph[H1_NB][3], mu[3][H1_NB], q[3][H1_NB][3]; } k;   // phases 1-2
    struct { double tz[QMAXM][H1_NJ][3], Pp[QMAXM][H1_NJ][3]; } j;                    // phases 4-6: til_c x z_j, P'_j
  } u;
};
static_assert(sizeof(double) * QS2_SIZE <= sizeof(double) * (12 * H1_NX + 2 * H1_NX + H1_NX + 9 + 36 + 6 * H1_NB + H1_NU + 3 * QMAXC), "the patch fits the storage it aliases");
static_assert(H1_NB * 3 >= H1_NX, "balance row m aliases Om");
static_assert(sizeof(QuadLds) <= 16384, "QuadLds must fit ten two-wave workgroups per CU");

__device__ __forceinline__ bool quad_selected(const DevState& S, int b, int mode) {
  if (mode == MASK_ALL) return true;
  if (mode == MASK_ACTIVE) return S.active[b] != 0;
  return S.active[b] != 0 && S.need_retry[b] != 0;
}
DEVFN bool related_mask(const unsigned* anc, int ja, int jb, int& lo, int& hi) {
  if ((anc[ja] >> jb) & 1u) { lo = ja; hi = jb; return true; }
  if ((anc[jb] >> ja) & 1u) { lo = jb; hi = ja; return true; }
  return false;
}
DEVFN double sel3(const double* v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); }   // no dynamic register index
// d2R/dquat_k dquat_l: dR_dquat is linear in q, so this is dR_dquat(k, e_l)
DEVFN void d2R_sel(int k, int l, double* D) {
  const double q[4] = {l == 0 ? 1.0 : 0.0, l == 1 ? 1.0 : 0.0, l == 2 ? 1.0 : 0.0, l == 3 ? 1.0 : 0.0};
  dR_dquat(k, q, D);
}

// Two waves per knot share the knot's LDS record (20.4 KB bound the occupancy at 8 one-wave workgroups per CU).  `lane` runs
// over 0..127: the lane-parallel phases 1-5 have at most 64 work items and stay on wave 0 (wave 1 waits at the barriers),
// the Hessian -- half of the kernel -- is split: patch entries over 128 lanes, two accumulator row tiles per wave.
// `lower` (inside a solve whose backward pass is the one-wave Riccati kernel): for the knots t < N only the tiles I >= J of lxx are
// computed and stored -- exactly the ones k_backward_wave loads (load_aug<true>, riccati_wave.hip); lxx is symmetric, the six
// strictly upper 16 x 16 tiles (35 % of its entries, 0.85 GB per launch at B = 4096) were written for nobody.  The terminal knot,
// which that kernel loads whole, and every stage-API call keep the full matrix (ilqr_hip_get_quadratics mirrors the tiles back).
__global__ void __launch_bounds__(128, QUAD_WAVES) k_cost_quadratics(DevState S, ProblemDev P, int mode, const int* list, const int* count, int lower) {
  const int t = blockIdx.x, lane = threadIdx.x, wv = lane >> 6;
  int b = blockIdx.y;
  if (list) {                        // compacted selection (DevState::order): no per-rollout flags to fetch, unselected workgroups leave on one cached scalar
    if (b >= *count) return;
    b = list[b];
    mode = MASK_ALL;
  }
  const int N = S.N;
  const bool term = (t == N);
  __shared__ QuadLds L;
#ifdef QUAD_STAMP
  long long qlast = clock64();
#endif
  const double* xg = S.xbar + ((size_t)b * (N + 1) + t) * H1_NX;

  // ---- phase 0: state (Pinocchio slot order of the quaternion), ancestor masks.  The rollout's selection flags are requested
  // together with the knot's data (indices clamped instead of predicated) and tested before anything is written: flag, state,
  // control and table one after the other were four serial HBM round trips at the top of every workgroup
  {
    const int f1 = mode == MASK_ALL ? 1 : S.active[b], f2 = mode == MASK_RETRY ? S.need_retry[b] : 1;
    const int lx = lane < H1_NX ? lane : 0;
    const int src = (lx == 3) ? 4 : (lx == 4) ? 5 : (lx == 5) ? 6 : (lx == 6) ? 3 : lx;
    const double xv = xg[src];
    const int tu = term ? N - 1 : t;
    const double uv = S.ubar[((size_t)b * N + tu) * H1_NU + (lane < H1_NU ? lane : 0)];
    const unsigned an = QANC.m[lane < H1_NB ? lane : 0];
    if (!(f1 && f2)) return;
    if (lane < H1_NX) L.xp[lane] = xv;
    if (!term && lane < H1_NU) L.us[lane] = uv;
    if (lane < H1_NB) L.anc[lane] = an;
  }
  __syncthreads();
  QSTAMP(0)

  // ---- phase 1: base kinematics, lane = body; Rj = Rfix * Rot(axis, theta) (child -> parent)
  {
    double Rj[9];
    int par = 0, ax = 0, dep = -1;
    if (lane >= 1 && lane < H1_NB) {
      par = H1_PARENT[lane]; ax = H1_AXIS[lane]; dep = H1_DEPTH[lane];
      double sn, cs; h1f::sincos_fast(L.xp[7 + lane - 1], &sn, &cs);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double f0 = H1U_RFIX[lane][r][0], f1 = H1U_RFIX[lane][r][1], f2 = H1U_RFIX[lane][r][2];
        Rj[3 * r + 0] = ax == 0 ? f0 : (ax == 1 ? f0 * cs - f2 * sn : f0 * cs + f1 * sn);
        Rj[3 * r + 1] = ax == 0 ? f1 * cs + f2 * sn : (ax == 1 ? f1 : f1 * cs - f0 * sn);
        Rj[3 * r + 2] = ax == 0 ? f2 * cs - f1 * sn : (ax == 1 ? f2 * cs + f0 * sn : f2);
      }
    } else if (lane == 0) {
      const double qx = L.xp[3], qy = L.xp[4], qz = L.xp[5], qw = L.xp[6];
      const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
      const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
      L.R0[0] = 1 - (tyy + tzz); L.R0[1] = txy - twz; L.R0[2] = txz + twy;
      L.R0[3] = txy + twz; L.R0[4] = 1 - (txx + tzz); L.R0[5] = tyz - twx;
      L.R0[6] = txz - twy; L.R0[7] = tyz + twx; L.R0[8] = 1 - (txx + tyy);
#pragma unroll
      for (int k = 0; k < 9; ++k) L.u.k.Rh[0][k] = (k % 4 == 0) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) { L.u.k.ph[0][k] = 0.0; L.zh[0][k] = 0.0; L.Om[0][k] = L.xp[H1_NQ + 3 + k]; }
      // upright pieces (4 quaternion slots, derivatives.cpp:646-666 labelling)
      const double ua = qx, ub = qy, uc = qz, ud = qw;
      L.ur[0] = 2.0 * (ub * ud + ua * uc); L.ur[1] = 2.0 * (uc * ud - ua * ub); L.ur[2] = -2.0 * (ub * ub + uc * uc);
      L.uJ[0][0] = 2 * uc; L.uJ[0][1] = 2 * ud; L.uJ[0][2] = 2 * ua; L.uJ[0][3] = 2 * ub;
      L.uJ[1][0] = -2 * ub; L.uJ[1][1] = -2 * ua; L.uJ[1][2] = 2 * ud; L.uJ[1][3] = 2 * uc;
      L.uJ[2][0] = 0.0; L.uJ[2][1] = -4 * ub; L.uJ[2][2] = -4 * uc; L.uJ[2][3] = 0.0;
    } else if (lane >= 32 && lane < 36) {
      dR_dquat(lane - 32, L.xp + 3, L.D[lane - 32]);
    }
    __syncthreads();
    for (int d = 1; d <= 5; ++d) {
      if (dep == d) {
        const int i = lane, p = par;
        double Rp[9], Ri[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rp[k] = L.u.k.Rh[p][k];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) Ri[3 * r + c] = Rp[3 * r] * Rj[c] + Rp[3 * r + 1] * Rj[3 + c] + Rp[3 * r + 2] * Rj[6 + c];
#pragma unroll
        for (int k = 0; k < 9; ++k) L.u.k.Rh[i][k] = Ri[k];
        const double px = H1U_POS[i][0], py = H1U_POS[i][1], pz = H1U_POS[i][2];
        const double qd = L.xp[H1_NQ + 6 + i - 1];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double z = sel3(Ri + 3 * k, ax);
          L.u.k.ph[i][k] = L.u.k.ph[p][k] + (Rp[3 * k] * px + Rp[3 * k + 1] * py + Rp[3 * k + 2] * pz);
          L.zh[i][k] = z;
          L.Om[i][k] = L.Om[p][k] + z * qd;
        }
      }
      __syncthreads();
    }
  }
  QSTAMP(1)

  // ---- phase 2: point sets, lane = (set, body)
  {
    const int s = lane / H1_NB, i = lane - s * H1_NB;
    const bool act = lane < 3 * H1_NB;
    if (act) {
      double mu;
      if (s == 0) { double mtot = 0.0; for (int k = 0; k < H1_NB; ++k) mtot += H1U_MASS[k]; mu = H1U_MASS[i] / mtot; }
      else mu = (i == ((s == 1) ? H1_EE_LEFT : H1_EE_RIGHT)) ? 1.0 : 0.0;
      double ch[3] = {0.0, 0.0, 0.0};
      if (s == 0) mv3(L.u.k.Rh[i], H1U_COM[i], ch);
      L.u.k.mu[s][i] = mu;
#pragma unroll
      for (int k = 0; k < 3; ++k) L.u.k.q[s][i][k] = mu * (L.u.k.ph[i][k] + ch[k]);
    }
    __syncthreads();
    if (act) {
      const unsigned m = L.anc[i];
      double msub = 0.0, h[3] = {0.0, 0.0, 0.0};
      for (int j = H1_NB - 1; j >= 0; --j)
        if ((m >> j) & 1u) { msub += L.u.k.mu[s][j]; h[0] += L.u.k.q[s][j][0]; h[1] += L.u.k.q[s][j][1]; h[2] += L.u.k.q[s][j][2]; }
      if (i == 0) {
        L.mfrac[s] = msub; L.on[s][0] = 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) { L.beta[s][k] = h[k]; L.w[s][0][k] = 0.0; L.dgam[s][0][k] = 0.0; }
      } else {
        L.on[s][i] = msub > 0.0 ? 1 : 0;
        double r[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) r[k] = h[k] - msub * L.u.k.ph[i][k];
        cross(L.zh[i], r, L.w[s][i]);
      }
    }
    __syncthreads();
    if (act) {
      if (i == 0) {
        const double* vb = L.xp + H1_NQ; const double* wb = L.xp + H1_NQ + 3;
        double wxb[3]; cross(wb, L.beta[s], wxb);
        double g[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) g[k] = L.mfrac[s] * vb[k] + wxb[k];
        for (int j = 1; j < H1_NB; ++j) { const double qd = L.xp[H1_NQ + 6 + j - 1]; g[0] += qd * L.w[s][j][0]; g[1] += qd * L.w[s][j][1]; g[2] += qd * L.w[s][j][2]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) L.gamma[s][k] = g[k];
      } else {
        const unsigned m = L.anc[i] & ~(1u << i);   // strict descendants
        double sv[3] = {0.0, 0.0, 0.0};
        for (int j = H1_NB - 1; j >= 1; --j)
          if ((m >> j) & 1u) { const double qd = L.xp[H1_NQ + 6 + j - 1]; sv[0] += qd * L.w[s][j][0]; sv[1] += qd * L.w[s][j][1]; sv[2] += qd * L.w[s][j][2]; }
        double a3[3], b3[3]; cross(L.Om[i], L.w[s][i], a3); cross(L.zh[i], sv, b3);
#pragma unroll
        for (int k = 0; k < 3; ++k) L.dgam[s][i][k] = a3[k] + b3[k];
      }
    }
    __syncthreads();
  }
  QSTAMP(2)

  // ---- phase 3: Jacobian columns of c and cdot, lane = coordinate (h1_cost_dev.h knot_jac_column)
  const int* stq = P.stance + b * P.stance_stride + 2 * t;      // (wave-uniform: scalar loads)
  const int fvel[2] = {stq[0] == 1, stq[1] == 1};               // stance foot: the velocity functional; swing foot: the position one
  if (lane < H1_NX) {
    const int c = lane;
    for (int s = 0; s < 3; ++s) {
      double jc[3] = {0, 0, 0}, jv[3] = {0, 0, 0};
      if (c < 3) { jc[0] = c == 0 ? L.mfrac[s] : 0.0; jc[1] = c == 1 ? L.mfrac[s] : 0.0; jc[2] = c == 2 ? L.mfrac[s] : 0.0; }
      else if (c < 7) { mv3(L.D[c - 3], L.beta[s], jc); mv3(L.D[c - 3], L.gamma[s], jv); }
      else if (c < H1_NQ) { const int j = c - 7 + 1; if (L.on[s][j]) { mv3(L.R0, L.w[s][j], jc); mv3(L.R0, L.dgam[s][j], jv); } }
      else {
        const int cv = c - H1_NQ;
        double col[3] = {0, 0, 0};
        if (cv < 3) { col[0] = cv == 0 ? L.mfrac[s] : 0.0; col[1] = cv == 1 ? L.mfrac[s] : 0.0; col[2] = cv == 2 ? L.mfrac[s] : 0.0; }
        else if (cv < 6) {  // -[beta]x column
          const int k = cv - 3; const double* bt = L.beta[s];
          if (k == 0) { col[1] = -bt[2]; col[2] = bt[1]; } else if (k == 1) { col[0] = bt[2]; col[2] = -bt[0]; } else { col[0] = -bt[1]; col[1] = bt[0]; }
        } else { const int j = cv - 6 + 1; if (L.on[s][j]) { col[0] = L.w[s][j][0]; col[1] = L.w[s][j][1]; col[2] = L.w[s][j][2]; } }
        mv3(L.R0, col, jv);
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (s == 0) { L.Jc[r][c] = jc[r]; L.Jv[r][c] = jv[r]; }
        else L.Jf[s - 1][r][c] = fvel[s - 1] ? jv[r] : jc[r];
      }
    }
  }
  __syncthreads();
  QSTAMP(3)

  // ---- phase 4: the weighted functionals of the active terms (order = ilqr.cpp:154-181)
  if (lane == 0) {
    int n = 0, g = 0; L.has_bal = 0;
    const int* st = P.stance + b * P.stance_stride + 2 * t;
    auto push = [&](int set, int is_vel, double scale, const double* vec, bool grad) {
      L.ctx[n].set = set; L.ctx[n].is_vel = is_vel; L.ctx[n].scale = scale;
      for (int k = 0; k < 3; ++k) L.ctx[n].vec[k] = vec[k];
      if (grad) { L.gset[g] = (unsigned char)set; L.gvel[g] = (unsigned char)is_vel; L.gscale[g] = scale; for (int k = 0; k < 3; ++k) L.gsum[g][k] = scale * vec[k]; ++g; }
      ++n;
    };
    if (P.w_com > 0.0) {   // CoM position: w ||com - ref||^2
      const double* ref = P.com_ref + b * P.com_ref_stride + t * 3;
      double rb[3], e[3]; mv3(L.R0, L.beta[0], rb);
      for (int k = 0; k < 3; ++k) e[k] = L.mfrac[0] * L.xp[k] + rb[k] - ref[k];
      push(0, 0, 2.0 * P.w_com, e, true);
    }
    if (!term && P.w_com_vel > 0.0) {
      const double* ref = P.com_vel_ref + b * P.com_vel_ref_stride + t * 3;
      double v[3], e[3]; mv3(L.R0, L.gamma[0], v);
      for (int k = 0; k < 3; ++k) e[k] = v[k] - ref[k];
      push(0, 1, 2.0 * P.w_com_vel, e, true);
    }
    for (int ee = 0; ee < 2; ++ee) {
      const int set = 1 + ee;
      if (P.w_ee_pos > 0.0 && st[ee] != 1) {
        const double* ref = P.ee_ref + b * P.ee_ref_stride + (t * 2 + ee) * 3;
        double rb[3], e[3]; mv3(L.R0, L.beta[set], rb);
        for (int k = 0; k < 3; ++k) e[k] = L.mfrac[set] * L.xp[k] + rb[k] - ref[k];
        push(set, 0, 2.0 * P.w_ee_pos, e, true);
      }
      if (P.w_ee_vel > 0.0 && st[ee] == 1) {
        double e[3]; mv3(L.R0, L.gamma[set], e);   // zero target (ilqr.cpp:734)
        push(set, 1, 2.0 * P.w_ee_vel, e, true);
      }
    }
    double ps[2];
    if (P.w_balance > 0.0 && support_point(P, b, t, ps)) {
      double rb[3], vc[3]; mv3(L.R0, L.beta[0], rb); mv3(L.R0, L.gamma[0], vc);
      double com[3]; for (int k = 0; k < 3; ++k) com[k] = L.mfrac[0] * L.xp[k] + rb[k];
      const double gg = 9.81;
      const double om = sqrt(com[2] / gg), om1 = 1.0 / (2.0 * gg * om), om2 = -1.0 / (4.0 * gg * gg * om * om * om);
      const double r0 = com[0] + vc[0] * om - ps[0], r1 = com[1] + vc[1] * om - ps[1];
      const double rv = r0 * vc[0] + r1 * vc[1];
      L.bal[0] = r0; L.bal[1] = r1; L.bal[2] = om; L.bal[3] = om1; L.bal[4] = om2; L.bal[5] = vc[0]; L.bal[6] = vc[1]; L.bal[7] = rv;
      L.has_bal = 1;
      const double mu[3] = {r0, r1, om1 * rv}, nu[3] = {om * r0, om * r1, 0.0};
      push(0, 0, P.w_balance, mu, false);
      push(0, 1, P.w_balance, nu, false);
    }
    // the second-order part is linear in scale_c vec_c: contexts acting on the same point set with the same type merge
    // (typically 6 -> 4: CoM position + balance, balance velocity part, one per foot); their scale becomes 1
    int nm = 0;
    for (int c = 0; c < n; ++c) {
      int at = -1;
      for (int q = 0; q < nm; ++q) if (L.ctx[q].set == L.ctx[c].set && L.ctx[q].is_vel == L.ctx[c].is_vel) at = q;
      const double sc = L.ctx[c].scale;
      const double v0 = sc * L.ctx[c].vec[0], v1 = sc * L.ctx[c].vec[1], v2 = sc * L.ctx[c].vec[2];
      if (at < 0) { const int st_ = L.ctx[c].set, iv = L.ctx[c].is_vel; at = nm++; L.ctx[at].set = st_; L.ctx[at].is_vel = iv; L.ctx[at].scale = 1.0; L.ctx[at].vec[0] = v0; L.ctx[at].vec[1] = v1; L.ctx[at].vec[2] = v2; }
      else { L.ctx[at].vec[0] += v0; L.ctx[at].vec[1] += v1; L.ctx[at].vec[2] += v2; }
    }
    L.nctx = nm; L.ng = g;
  }
  __syncthreads();
  const int nctx = L.nctx, ng = L.ng, has_bal = L.has_bal;
  if (lane < nctx) {
    QuadCtx& C = L.ctx[lane];
    mtv3(L.R0, C.vec, C.til);
    for (int k = 0; k < 4; ++k) mtv3(L.D[k], C.vec, C.Dv[k]);
  }
  if (has_bal && lane < H1_NX) {
    const int a = lane;
    const double om = L.bal[2], om1 = L.bal[3];
    const double jz = L.Jc[2][a];
    L.jr[0][a] = L.Jc[0][a] + om * L.Jv[0][a] + L.bal[5] * om1 * jz;
    L.jr[1][a] = L.Jc[1][a] + om * L.Jv[1][a] + L.bal[6] * om1 * jz;
  }
  __syncthreads();   // (also: the phase-1/2 temporaries are dead, their storage becomes tz / Pp)
  for (int e = lane; e < nctx * H1_NJ; e += 128) {
    const int c = e / H1_NJ, j = 1 + e - c * H1_NJ;
    const QuadCtx& C = L.ctx[c];
    double tz[3]; cross(C.til, L.zh[j], tz);
#pragma unroll
    for (int k = 0; k < 3; ++k) L.u.j.tz[c][j - 1][k] = tz[k];
    if (C.is_vel) {
      double tO[3], p1[3], p2[3]; cross(C.til, L.Om[j], tO); cross(tO, L.zh[j], p1); cross(tz, L.Om[j], p2);
#pragma unroll
      for (int k = 0; k < 3; ++k) L.u.j.Pp[c][j - 1][k] = p1[k] - p2[k];
    }
  }
  __syncthreads();
  double* const balm = &L.Om[0][0];   // Om is dead from here on: its storage takes the balance row m = om1 (r0 Jv0 + r1 Jv1) + rv om2 jz / 2
  if (has_bal && lane < H1_NX) balm[lane] = L.bal[3] * (L.bal[0] * L.Jv[0][lane] + L.bal[1] * L.Jv[1][lane]) + 0.5 * L.bal[7] * L.bal[4] * L.Jc[2][lane];
  __syncthreads();
  QSTAMP(4)

  const double* Qd = term ? P.Qf : P.Q;
  const double* xr = P.x_ref + b * P.x_ref_stride + t * H1_NX;

  // ---- phase 5: gradient lx (lane = coordinate), lu / luu
  if (lane < H1_NX) {
    const int a = lane;
    double g = Qd[a] * (xg[a] - xr[a]);   // Q acts on the MuJoCo-ordered state
    for (int i = 0; i < ng; ++i) {
      const double (*J)[H1_NX] = L.gset[i] == 0 ? (L.gvel[i] ? L.Jv : L.Jc) : L.Jf[L.gset[i] - 1];   // (a foot's Jf is of the type its functional has)
      g += J[0][a] * L.gsum[i][0] + J[1][a] * L.gsum[i][1] + J[2][a] * L.gsum[i][2];
    }
    if (P.w_upright > 0.0 && a >= 3 && a < 7) g += P.w_upright * (L.uJ[0][a - 3] * L.ur[0] + L.uJ[1][a - 3] * L.ur[1] + L.uJ[2][a - 3] * L.ur[2]);
    if (has_bal) g += P.w_balance * (L.jr[0][a] * L.bal[0] + L.jr[1][a] * L.bal[1]);
    if (a >= 7 && a < H1_NQ) {
      double lo, hi; limit_bounds(H1_JRANGE[a - 7], lo, hi);
      const double q = L.xp[a];
      if (q > hi) g += 2.0 * P.w_joint * (q - hi);
      if (q < lo) g += -2.0 * P.w_joint * (lo - q);
    }
    S.lx[((size_t)b * (N + 1) + t) * H1_NX + a] = g;
    double dgl = Qd[a];
    if (a >= 7 && a < H1_NQ) {
      double lo, hi; limit_bounds(H1_JRANGE[a - 7], lo, hi);
      const double q = L.xp[a];
      if (q > hi || q < lo) dgl += 2.0 * P.w_joint;
    }
    L.dg[a] = dgl;
  }
  if (!term && lane < H1_NU) {
    const double* ur_ = P.u_ref + b * P.u_ref_stride + t * H1_NU;
    const double u = L.us[lane];
    double g = P.R[lane] * (u - ur_[lane]), h = P.R[lane];
    double lo, hi; limit_bounds(H1_CTRLRANGE[lane], lo, hi);
    if (u > hi) g += 2.0 * P.w_ctrl * (u - hi);
    if (u < lo) g += -2.0 * P.w_ctrl * (lo - u);
    if (u > hi || u < lo) h += 2.0 * P.w_ctrl;
    S.lu[((size_t)b * N + t) * H1_NU + lane] = g;
    S.luu[((size_t)b * N + t) * H1_NU + lane] = h;
  }
  QSTAMP(5)

  // ---- phase 6: Hessian lxx
  double* Hg = S.lxx + ((size_t)b * (N + 1) + t) * H1_NX * H1_NX;
  typedef double v4d_q __attribute__((ext_vector_type(4)));
  const int lr = lane & 15, lk = (lane >> 4) & 3;
  // 6a: operands of the first-order product H1[a][b] = sum_k sA_k RA_k[a] RB_k[b], k = 4 ks + lk:
  //   lk < 3 : row lk of the Jacobian of gradient-carrying functional ks (RA = RB, sA = its scale)
  //   lk = 3 : the balance dyads w (jr0 jr0' + jr1 jr1' + jz m' + m jz'): ks = 0 jr0, 1 jr1, 2 (jz, m), 3 (m, jz)
  // operand of row tile I / column tile J = entry 16 I + lr of the row; everything beyond column 50 and every unused row is
  // a true zero (junk operands slow the fp64 MFMA down tenfold)
  double av[4][2], bv[4][4];            // this wave's two row tiles wv, wv + 2 (balanced when only the tiles I >= J are wanted); all four column tiles
  {
    double* const balm_ = &L.Om[0][0];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int ii = ks < ng ? ks : 0;
      const double* rowJ = (L.gset[ii] == 0 ? (L.gvel[ii] ? &L.Jv[0][0] : &L.Jc[0][0]) : &L.Jf[L.gset[ii] - 1][0][0]) + (lk < 3 ? lk : 0) * H1_NX;
      const double* rowA = lk < 3 ? rowJ : (ks == 0 ? L.jr[0] : (ks == 1 ? L.jr[1] : (ks == 2 ? L.Jc[2] : balm_)));
      const double* rowB = lk < 3 ? rowJ : (ks == 0 ? L.jr[0] : (ks == 1 ? L.jr[1] : (ks == 2 ? balm_ : L.Jc[2])));
      const bool used = lk < 3 ? (ks < ng) : (has_bal != 0);
      const double sA = lk < 3 ? L.gscale[ii] : P.w_balance;
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        const int e = 16 * T + lr, ec = e < H1_NX ? e : H1_NX - 1;
        const double ra = rowA[ec], rb = rowB[ec];
        const bool ok = used && e < H1_NX;
        bv[ks][T] = ok ? rb : 0.0;
        const double avv = ok ? sA * ra : 0.0;
        if ((T & 1) == wv) av[ks][T >> 1] = avv;          // row tile T belongs to wave T & 1, slot T >> 1
      }
    }
  }
  QSTAMP(6)
  __syncthreads();   // the Jacobian rows are in registers: their storage becomes the second-order patch
  for (int e = lane; e < QS2_SIZE; e += 128) L.S2[e] = 0.0;
  __syncthreads();
  // 6b: second-order part, block by block of the coordinate classes (a <= bb in every block).  The (merged) contexts'
  // set / type are wave-uniform: held in scalar registers, so the loops over them branch uniformly.
  auto patch = [&](int a, int bb, double h) { L.S2[QS2_IDX(a, bb)] = h; };
  // triangular index -> (i, j), i <= j < n
  auto tri = [](int idx, int n, int& i, int& j) {
    int a = (int)((2 * n + 1 - sqrtf((float)((2 * n + 1) * (2 * n + 1) - 8 * idx))) * 0.5f);   // exact integers in fp32; corrected below
    while ((a + 1) * n - ((a + 1) * a) / 2 <= idx) ++a;
    while (a * n - (a * (a - 1)) / 2 > idx) --a;
    i = a; j = a + (idx - (a * n - (a * (a - 1)) / 2));
  };
  const int Q0 = 3, T0 = 7, V0 = H1_NQ, W0 = H1_NQ + 3, D0 = H1_NQ + 6, NJ = H1_NJ;
  int cset[QMAXC], cvel[QMAXC];
#pragma unroll
  for (int c = 0; c < QMAXC; ++c) {
    const int cc = c < nctx ? c : 0;
    cset[c] = __builtin_amdgcn_readfirstlane(L.ctx[cc].set); cvel[c] = __builtin_amdgcn_readfirstlane(L.ctx[cc].is_vel);
  }
  // (theta, theta) and (theta, thetadot), related joints only (compile-time list): three passes of 64 entries
  for (int e = lane; e < 158; e += 128) {
    const unsigned pk = QREL.e[e];
    const int ja = pk & 31, jb = (pk >> 5) & 31;
    const bool isd = (pk >> 10) != 0;
    const int lo = ja < jb ? ja : jb, hi = ja < jb ? jb : ja;     // parents precede their children in the body numbering
    double h = 0.0;
#pragma unroll
    for (int c = 0; c < QMAXC; ++c) {
      if (c >= nctx) break;
      const int st_ = cset[c];
      const double f = (L.on[st_][ja] & L.on[st_][jb]) ? 1.0 : 0.0;
      const double* tz = L.u.j.tz[c][lo - 1];
      const double* wh = L.w[st_][hi];
      const double t1 = dot3(tz, wh);
      double v;
      if (cvel[c]) { const double t2 = dot3(L.u.j.Pp[c][lo - 1], wh) + dot3(tz, L.dgam[st_][hi]); v = isd ? t1 : t2; }
      else v = isd ? 0.0 : t1;
      h += f * v;                                  // (the merged contexts carry their weight in vec: scale = 1)
    }
    patch(T0 + ja - 1, (isd ? D0 : T0) + jb - 1, h);
  }
  // (quat, theta) and (quat, thetadot): Dv . w / Dv . dgam
  for (int idx = lane; idx < 2 * 4 * NJ; idx += 128) {
    const bool isd = idx >= 4 * NJ;
    const int e = isd ? idx - 4 * NJ : idx;
    const int k = e / NJ, j = 1 + e % NJ;
    double h = 0.0;
#pragma unroll
    for (int c = 0; c < QMAXC; ++c) {
      if (c >= nctx) break;
      const int st_ = cset[c];
      const double f = L.on[st_][j] ? 1.0 : 0.0;
      const double* Dv = L.ctx[c].Dv[k];
      double v;
      if (cvel[c]) v = dot3(Dv, isd ? L.w[st_][j] : L.dgam[st_][j]);
      else v = isd ? 0.0 : dot3(Dv, L.w[st_][j]);
      h += f * v;
    }
    patch(Q0 + k, (isd ? D0 : T0) + j - 1, h);
  }
  // (theta, omega_b): (w_j x til)_c on wave 0; (quat, v_b), (quat, omega_b) and (quat, quat) on wave 1
  if (lane < 3 * NJ) {
    const int ja = 1 + lane / 3, cc = lane % 3;
    double h = 0.0;
#pragma unroll
    for (int c = 0; c < QMAXC; ++c) {
      if (c >= nctx) break;
      if (!cvel[c]) continue;
      const int st_ = cset[c];
      const double f = L.on[st_][ja] ? 1.0 : 0.0;
      double tv[3]; cross(L.w[st_][ja], L.ctx[c].til, tv);
      h += f * sel3(tv, cc);
    }
    patch(T0 + ja - 1, W0 + cc, h);
  }
  if (lane >= 64 && lane < 64 + 24) {
    const bool isw = lane >= 64 + 12;
    const int e = isw ? lane - 64 - 12 : lane - 64;
    const int k = e / 3, cc = e % 3;
    double h = 0.0;
#pragma unroll
    for (int c = 0; c < QMAXC; ++c) {
      if (c >= nctx) break;
      if (!cvel[c]) continue;
      const QuadCtx& C = L.ctx[c];
      const int st_ = cset[c];
      if (!isw) h += C.Dv[k][cc] * L.mfrac[st_];
      else { double tv[3]; cross(L.beta[st_], C.Dv[k], tv); h += sel3(tv, cc); }   // Dv . (-[beta]x e_c) = (beta x Dv)_c
    }
    patch(Q0 + k, (isw ? W0 : V0) + cc, h);
  } else if (lane >= 96 && lane < 106) {
    // (quat, quat): d2R/dq2 terms + upright
    int ka, kb; tri(lane - 96, 4, ka, kb);
    double h = 0.0;
    double D2[9]; d2R_sel(ka, kb, D2);
#pragma unroll
    for (int c = 0; c < QMAXC; ++c) {
      if (c >= nctx) break;
      const QuadCtx& C = L.ctx[c];
      const int st_ = cset[c];
      double tv[3]; mv3(D2, cvel[c] ? L.gamma[st_] : L.beta[st_], tv);
      h += dot3(C.vec, tv);
    }
    if (P.w_upright > 0.0) {
      const int i = ka, j = kb;
      double v = L.uJ[0][i] * L.uJ[0][j] + L.uJ[1][i] * L.uJ[1][j] + L.uJ[2][i] * L.uJ[2][j];
      if ((i == 0 && j == 2) || (i == 1 && j == 3)) v += 2.0 * L.ur[0];
      if (i == 2 && j == 3) v += 2.0 * L.ur[1];
      if (i == 0 && j == 1) v += -2.0 * L.ur[1];
      if ((i == 1 && j == 1) || (i == 2 && j == 2)) v += -4.0 * L.ur[2];
      h += P.w_upright * v;
    }
    patch(Q0 + ka, Q0 + kb, h);
  }
  __syncthreads();
  QSTAMP(7)
  // 6c: first-order product row tile by row tile; the owner lane of an accumulator element (row 16 I + 4 r + lk, column
  // 16 J + lr) adds the diagonal terms and the patch entry of its (unordered) index pair and stores it: for a fixed
  // register the wave writes four rows x 16 consecutive columns
  const bool low = lower && !term;
#pragma unroll
  for (int Ii = 0; Ii < 2; ++Ii) {
    const int I = wv + 2 * Ii;
    v4d_q acc[4];
#pragma unroll
    for (int J = 0; J < 4; ++J) acc[J] = (v4d_q){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int J = 0; J < 4; ++J)
        if (!(low && J > I)) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks][Ii], bv[ks][J], acc[J], 0, 0, 0);      // (wave-uniform)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int a = 16 * I + 4 * r + lk;
      if (a >= H1_NX) continue;                        // rows beyond 50 do not exist (last row tile)
#pragma unroll
      for (int J = 0; J < 4; ++J) {
        if (low && J > I) continue;
        const int bb = 16 * J + lr;
        double h = acc[J][r];
        if (a == bb) h += L.dg[a];
        const int lo2 = a < bb ? a : bb, hi2 = a < bb ? bb : a;
        if (lo2 >= QS2_R0 && lo2 < QS2_R0 + QS2_NR && hi2 < H1_NX) h += L.S2[QS2_IDX(lo2, hi2)];
        if (bb < H1_NX) Hg[a * H1_NX + bb] = h;
      }
    }
  }
  QSTAMP(8)
}

void launch_cost_quadratics(const DevState& S, const ProblemDev& P, int mode, hipStream_t st, int iter, int lower) {
  const WorkList w = work_list(S, mode, iter);
  hipLaunchKernelGGL(k_cost_quadratics, dim3(S.N + 1, S.B), dim3(128), 0, st, S, P, mode, w.list, w.count, lower);
}

}  // namespace ilqr
