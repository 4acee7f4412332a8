// Dynamics kernels on TWO lanes per rollout / candidate (h1_aba_split.h): the even lane owns the left leg and arm,
// the odd lane the right ones.  A separate translation unit from dyn_kernels.hip on purpose: with both variants in one
// file the one-lane kernels came out 20 % slower (different inlining / register allocation).
//   k_rollout_s      iLQR::forwardRolloutNominal                      reference src/ilqr/ilqr.cpp:119-124
//   (computeTotalCost, ilqr.cpp:363-518, of the stored trajectories: k_traj_knot_cost / k_traj_cost_sum in dyn_kernels.hip)
//   k_line_search_s  iLQR::forwardPassLineSearch, 8 alphas at once    reference src/ilqr/ilqr.cpp:311-361
#include <hip/hip_runtime.h>

#include "h1_cost_dev.h"
#define ABA_FENCE          // scheduling fences between the sweeps of the articulated-body algorithm (h1_aba_split.h)
#include "h1_aba_split.h"
#include "h1_linearize_dev.h"      // LinDumpG: what k_lin_tangent reads of the nominal knot
#include "ilqr_kernels.h"

using namespace h1;

namespace ilqr {

#ifndef LS_UNROLL
#define LS_UNROLL 2
#endif
#ifdef LS_STAMP
#define LSS(k) { __builtin_amdgcn_s_waitcnt(0); const long long tn_ = clock64(); ph[k] += tn_ - tl; tl = tn_; }
#else
#define LSS(k)
#endif

__device__ __forceinline__ bool sel_s(const DevState& S, int b, int mode) {
  if (mode == MASK_ALL) return true;
  if (mode == MASK_ACTIVE) return S.active[b] != 0;
  return S.active[b] != 0 && S.need_retry[b] != 0;
}
__constant__ double ALPHAS_S[8] = {1.0, 0.8, 0.6, 0.4, 0.2, 0.1, 0.05, 0.01};
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  const int lo = __shfl_xor(__double2loint(v), mask), hi = __shfl_xor(__double2hiint(v), mask);
  return __hiloint2double(hi, lo);
}

// ---- contact row f4 on the two-lane kernels -------------------------------------------------------------------------
// One compiled copy of the stance-constrained step (h1s::step_stance) shared by every kernel of this file: the constraint
// solve (twelve unit-wrench propagations, a 12 x 12 Cholesky) wants the register file to itself, and one machine code for
// the rollout and the line search makes the nominal re-rollout reproduce the accepted candidate bit for bit.
// The constraint-free path (contact mode 0, the headline) keeps its inlined step and is not touched by this.
extern __shared__ double dyn_lds_c[];
__device__ __attribute__((noinline)) void step_stance_shared(h1s::HalfX* hp, const h1s::HalfU* up, double dt, double gx, double gy, double gz,
                                                             double soft, int mode, int st_left, int st_right, double mu) {
  const int lane = threadIdx.x;
  const bool side = (lane & 1) != 0;
  const h1s::LaneLds L{dyn_lds_c, 64, lane};
  const double grav[3] = {gx, gy, gz};
  h1s::HalfX h = *hp;
  const h1s::HalfU u = *up;
  h1s::step_stance<false>(side, h, u, dt, grav, L, soft, mode, (side ? st_right : st_left) == 1, (side ? st_left : st_right) == 1, mu);
  *hp = h;
}
// the copy with kinetic friction on sliding feet (contact mode 4), see h1s::stance_correct<KIN>
__device__ __attribute__((noinline)) void step_stance_shared_kin(h1s::HalfX* hp, const h1s::HalfU* up, double dt, double gx, double gy, double gz,
                                                                 double soft, int st_left, int st_right, double mu) {
  const int lane = threadIdx.x;
  const bool side = (lane & 1) != 0;
  const h1s::LaneLds L{dyn_lds_c, 64, lane};
  const double grav[3] = {gx, gy, gz};
  h1s::HalfX h = *hp;
  const h1s::HalfU u = *up;
  h1s::step_stance<true>(side, h, u, dt, grav, L, soft, 4, (side ? st_right : st_left) == 1, (side ? st_left : st_right) == 1, mu);
  *hp = h;
}
// the step with joint-limit rows (DynParams::limits; h1_aba_split.h "The step with the rows"): only reached when the option is on.
// lim_accelerations is the ONE copy of the constrained accelerations, mask = the hinges it treats as acceleration-prescribed.
struct LimAcc { double qb[6]; h1s::HalfAcc qa; };
template <bool KIN>
__device__ __attribute__((noinline)) void lim_accelerations(const h1s::HalfX* hp, const h1s::HalfU* up, unsigned mask, double dt, double gx, double gy, double gz,
                                                            double soft, int mode, int st_left, int st_right, double mu, double kr, LimAcc* out) {
  const int lane = threadIdx.x;
  const bool side = (lane & 1) != 0;
  const h1s::LaneLds L{dyn_lds_c, 64, lane};
  const double grav[3] = {gx, gy, gz};
  const h1s::HalfX h = *hp;
  const h1s::HalfU u = *up;
  double qh[4], R0[9]; h1s::HalfTau tau, add;
  h1s::stance_prepare(side, h, u, qh, R0, tau);
  h1s::apply_lock_mask(side, mask, h.q, dt, kr, tau, add);
  const bool st_own = mode != 0 && (side ? st_right : st_left) == 1, st_par = mode != 0 && (side ? st_left : st_right) == 1;      // (mode 0: no stance rows)
  LimAcc o;
  h1s::stance_accelerations<KIN, true>(side, R0, h, tau, dt, grav, L, soft, mode, st_own, st_par, mu, o.qb, o.qa, &add);
  *out = o;
}
template <bool KIN>
DEVFN void step_lim(h1s::HalfX* hp, const h1s::HalfU* up, double dt, double gx, double gy, double gz, double soft, int mode, int st_left, int st_right, double mu, double kr) {
  const bool side = (threadIdx.x & 1) != 0;
  LimAcc o;
  lim_accelerations<KIN>(hp, up, 0u, dt, gx, gy, gz, soft, mode, st_left, st_right, mu, kr, &o);
  h1s::HalfX h = *hp;
  const unsigned mask = h1s::limit_lock_mask(side, h.q, o.qa, dt, kr);
  const bool any = mask != 0u;
  if (h1s::xch_flag(any) || any) lim_accelerations<KIN>(hp, up, mask, dt, gx, gy, gz, soft, mode, st_left, st_right, mu, kr, &o);      // (the pair runs the recursion together)
  const double qn = sqrt(h.quat[0] * h.quat[0] + h.quat[1] * h.quat[1] + h.quat[2] * h.quat[2] + h.quat[3] * h.quat[3]);
  const double qh[4] = {h.quat[0] / qn, h.quat[1] / qn, h.quat[2] / qn, h.quat[3] / qn};
  h1s::integrate_half(h, qh, o.qb, o.qa, dt);
  *hp = h;
}
__device__ __attribute__((noinline)) void step_stance_shared_lim(h1s::HalfX* hp, const h1s::HalfU* up, double dt, double gx, double gy, double gz,
                                                                 double soft, int mode, int st_left, int st_right, double mu, double kr) {
  step_lim<false>(hp, up, dt, gx, gy, gz, soft, mode, st_left, st_right, mu, kr);
}
__device__ __attribute__((noinline)) void step_stance_shared_kin_lim(h1s::HalfX* hp, const h1s::HalfU* up, double dt, double gx, double gy, double gz,
                                                                     double soft, int st_left, int st_right, double mu, double kr) {
  step_lim<true>(hp, up, dt, gx, gy, gz, soft, 4, st_left, st_right, mu, kr);
}
// ---- joint-limit rows on the constraint-free plant (CONTACT == 5, round 6) ------------------------------------------------------
// With no stance rows the first pass of the step with the rows IS the free step's recursion: it stays inlined, as in the constraint-free
// kernels, and only a lane pair that has a hinge to constrain calls out -- state, controls and the mask cross that call through the
// lane's own LDS column (the dynamics scratch is dead at that point; 44 of its 80 slots), the accelerations come back the same way, so
// the caller keeps no address-taken state (by pointer, the mere presence of such a call cost the constraint-free line search 20 %, round 3).
// Until round 6 this plant ran on CONTACT == 3 -- the shared constrained step, stance code and all: - 19 % on the headline's batch with
// nothing to stop.
DEVFN void lds_put_state(const h1s::LaneLds& L, const h1s::HalfX& h) {
#pragma unroll
  for (int k = 0; k < 3; ++k) L[k] = h.p[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) L[3 + k] = h.quat[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) L[7 + k] = h.vb[k];
  L[13] = h.q.th11; L[14] = h.q.qd11;
#pragma unroll
  for (int k = 0; k < 5; ++k) { L[15 + k] = h.q.thL[k]; L[20 + k] = h.q.qdL[k]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { L[25 + k] = h.q.thA[k]; L[29 + k] = h.q.qdA[k]; }
}
DEVFN void lds_get_state(const h1s::LaneLds& L, h1s::HalfX& h) {
#pragma unroll
  for (int k = 0; k < 3; ++k) h.p[k] = L[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) h.quat[k] = L[3 + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) h.vb[k] = L[7 + k];
  h.q.th11 = L[13]; h.q.qd11 = L[14];
#pragma unroll
  for (int k = 0; k < 5; ++k) { h.q.thL[k] = L[15 + k]; h.q.qdL[k] = L[20 + k]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { h.q.thA[k] = L[25 + k]; h.q.qdA[k] = L[29 + k]; }
}
// second pass: the recursion with the hinges of `mask` acceleration-prescribed (no stance rows); in: L[0..32] state, L[33..42] controls,
// L[43] the mask; out: L[0..5] base accelerations, L[6] torso, L[7..11] leg, L[12..15] arm hinge accelerations
__device__ __attribute__((noinline)) void lim_second_pass_lds(double dt, double gx, double gy, double gz, double kr) {
  const int lane = threadIdx.x;
  const bool side = (lane & 1) != 0;
  const h1s::LaneLds L{dyn_lds_c, 64, lane};
  const double grav[3] = {gx, gy, gz};
  h1s::HalfX h; h1s::HalfU u;
  lds_get_state(L, h);
  u.u11 = L[33];
#pragma unroll
  for (int k = 0; k < 5; ++k) u.uL[k] = L[34 + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) u.uA[k] = L[39 + k];
  const unsigned mask = (unsigned)__double_as_longlong(L[43]);
  double qh[4], R0[9]; h1s::HalfTau tau, add;
  h1s::stance_prepare(side, h, u, qh, R0, tau);
  h1s::apply_lock_mask(side, mask, h.q, dt, kr, tau, add);
  double qb[6]; h1s::HalfAcc qa;
  h1s::forward_dynamics<true>(side, R0, h.vb, h.q, tau, h1s::ARMATURE + dt * h1s::DAMPING, grav, L, qb, qa, nullptr, nullptr, &add);
#pragma unroll
  for (int k = 0; k < 6; ++k) L[k] = qb[k];
  L[6] = qa.q11;
#pragma unroll
  for (int k = 0; k < 5; ++k) L[7 + k] = qa.qL[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) L[12 + k] = qa.qA[k];
}
// one step of either kind; `st` = stance flags (left, right) of the knot being stepped
// (compile-time switch: the constraint-free instantiation of a kernel contains no call and no address-taken state -- with a
// run-time branch the mere presence of the call cost the headline's line search 20 %)
DEVFN void pin(double& v) { asm volatile("" : "+v"(v)); }
DEVFN void pin_half(h1s::HalfX& h) {
#pragma unroll
  for (int k = 0; k < 3; ++k) pin(h.p[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) pin(h.quat[k]);
#pragma unroll
  for (int k = 0; k < 6; ++k) pin(h.vb[k]);
  pin(h.q.th11); pin(h.q.qd11);
#pragma unroll
  for (int k = 0; k < 5; ++k) { pin(h.q.thL[k]); pin(h.q.qdL[k]); }
#pragma unroll
  for (int k = 0; k < 4; ++k) { pin(h.q.thA[k]); pin(h.q.qdA[k]); }
}
DEVFN void pin_half_u(h1s::HalfU& u) {
  pin(u.u11);
#pragma unroll
  for (int k = 0; k < 5; ++k) pin(u.uL[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) pin(u.uA[k]);
}
// CONTACT: 0 constraint-free, 1 stance constraints (contact modes 1-3), 2 stance constraints with kinetic friction on sliding feet (mode 4),
// 3 / 4: as 1 / 2 with joint-limit rows (DynParams::limits; 3 also serves the constraint-free plant with them: no stance rows in mode 0).
// Mode 4 has kernels of its own: the private segment of a kernel is the largest frame it can reach, and the constrained kernels lose with
// every kilobyte of it (1.4 -> 1.8 KB per lane: -0.7 % on the contact bench, -> 4 KB: -4 %, same machine code otherwise).
template <int CONTACT>
DEVFN void step_any(bool side, h1s::HalfX& h, const h1s::HalfU& u, const DynParams& dyn, const int* st, const h1s::LaneLds& L) {
  if constexpr (CONTACT == 5) {
    double dt = dyn.h; asm volatile("" : "+s"(dt));
    h1s::HalfU uo = u;
    pin_half(h); pin_half_u(uo);
    double qh[4], R0[9]; h1s::HalfTau tau;
    h1s::stance_prepare(side, h, uo, qh, R0, tau);
    double qb[6]; h1s::HalfAcc qa;
    h1s::forward_dynamics(side, R0, h.vb, h.q, tau, h1s::ARMATURE + dt * h1s::DAMPING, dyn.g, L, qb, qa);
    const unsigned mask = h1s::limit_lock_mask(side, h.q, qa, dt, dyn.lim_k);
    const bool any = mask != 0u;
    // (inlined in this branch instead, the second pass made a 1.7 KB private segment with 526 spilled registers: measured, dropped)
    if (h1s::xch_flag(any) || any) {      // (the pair runs the recursion together)
      lds_put_state(L, h);
      L[33] = uo.u11;
#pragma unroll
      for (int k = 0; k < 5; ++k) L[34 + k] = uo.uL[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) L[39 + k] = uo.uA[k];
      L[43] = __longlong_as_double((long long)mask);
      lim_second_pass_lds(dt, dyn.g[0], dyn.g[1], dyn.g[2], dyn.lim_k);
#pragma unroll
      for (int k = 0; k < 6; ++k) qb[k] = L[k];
      qa.q11 = L[6];
#pragma unroll
      for (int k = 0; k < 5; ++k) qa.qL[k] = L[7 + k];
#pragma unroll
      for (int k = 0; k < 4; ++k) qa.qA[k] = L[12 + k];
    }
    h1s::integrate_half(h, qh, qb, qa, dt);
    pin_half(h);
  }
  else if constexpr (CONTACT == 4) step_stance_shared_kin_lim(&h, &u, dyn.h, dyn.g[0], dyn.g[1], dyn.g[2], dyn.soft, st[0], st[1], dyn.mu, dyn.lim_k);
  else if constexpr (CONTACT == 3) step_stance_shared_lim(&h, &u, dyn.h, dyn.g[0], dyn.g[1], dyn.g[2], dyn.soft, dyn.contact, st[0], st[1], dyn.mu, dyn.lim_k);
  else if constexpr (CONTACT == 2) step_stance_shared_kin(&h, &u, dyn.h, dyn.g[0], dyn.g[1], dyn.g[2], dyn.soft, st[0], st[1], dyn.mu);
  else if constexpr (CONTACT == 1) step_stance_shared(&h, &u, dyn.h, dyn.g[0], dyn.g[1], dyn.g[2], dyn.soft, dyn.contact, st[0], st[1], dyn.mu);
  else {
    // (the step size behind an opaque barrier as well: with h a loop invariant the articulated quantities of the chains' leaf
    // bodies -- constants plus the armature term h * damping -- are hoisted out of the knot loop, spilled and reloaded per step)
    double dt = dyn.h; asm volatile("" : "+s"(dt));
    // Opaque boundary around the step: inlined into different kernels the same source is otherwise fused / scheduled together
    // with whatever surrounds it (the feedback law in the line search, plain loads in the rollout), and the re-rollout of an
    // accepted candidate can differ from it in the last bit of a few entries.  With every input and output pinned the step is
    // the same expression graph in every kernel (ilqr_hip_get_adopt_mismatches stays 0; GPU tests).
    h1s::HalfU uo = u;
    pin_half(h); pin_half_u(uo);
    h1s::step(side, h, uo, dt, dyn.g, L);
    pin_half(h);
  }
}

// ---- line search on two lanes per candidate (h1_aba_split.h): thread per (rollout, alpha, side), the 16 lanes of
// a rollout adjacent (lane = 16 r + 2 alpha + side).  With one lane per candidate the 8 x B candidates fill only 2
// waves per CU; two lanes each give every SIMD a wave.  Same cooperative feedback as k_line_search_r, over 16 lanes.
#define DYN_LDS_BYTES_S (h1s::LDS_SLOTS * 64 * sizeof(double))
DEVFN void load_half_u(bool side, const double* u, h1s::HalfU& o) {
  o.u11 = u[10];
#pragma unroll
  for (int k = 0; k < 5; ++k) o.uL[k] = u[h1s::jleg(side, k)];
#pragma unroll
  for (int k = 0; k < 4; ++k) o.uA[k] = u[h1s::jarm(side, k)];
}
// RPW = rollouts per wave: 4 (the 64 lanes are four rollouts' 16) or 1 -- for batches of at most 1024 rollouts, where four per wave
// would leave SIMDs empty: the wave's lanes 16..63 then mirror lanes 0..15 (same rollout, same candidates, no stores), a step
// fetches one rollout's K_t and the four blocks of the feedback product take four row groups of that rollout instead of four rollouts,
// and every rollout has a SIMD to itself.  Every output accumulates the same thirteen k-steps in the same order in both: results are
// bit-identical (batch invariance, GPU tests).
template <int CONTACT, int RPW>
__global__ void __launch_bounds__(64) k_line_search_s(DevState S, ProblemDev P, int mode, const int* list, const int* count) {
  extern __shared__ double lds[];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int ai = (gid >> 1) & 7;
  const bool side = (gid & 1) != 0;
  // No lane leaves early: the feedback below is an MFMA product over the whole wave (every lane supplies operands of every
  // rollout of the wave), so the 16 lanes of a rollout that is not selected run along on a valid rollout and skip their stores.
  int b = RPW == 4 ? gid >> 4 : (int)blockIdx.x;
  const bool owner = RPW == 4 || threadIdx.x < 16;          // RPW = 1: lanes 16..63 are mirrors
  bool live;
  if (list) {                          // compacted selection (DevState::order): the selected rollouts fill the first waves
    const int cnt = *count;
    live = b < cnt;
    b = list[live ? b : 0];
  } else {
    live = b < S.B;
    b = live ? b : S.B - 1;
    live = live && sel_s(S, b, mode);
  }
  if (!__any(live)) return;
  live = live && owner;
  const int lane0 = threadIdx.x;
  const int N = S.N, n = H1_NX, m = H1_NU;
  const double alpha = ALPHAS_S[ai];
  h1s::HalfX h; h1s::load_half(side, S.x0 + (size_t)b * n, h);
#ifdef LS_STAMP
  long long ph[8] = {0}; long long tl = clock64();
#endif
  for (int t = 0; t < N; ++t) {
    // rollout and lane indices re-derived behind an opaque barrier every step: the dozens of per-lane addresses and LDS
    // offsets derived from them are then recomputed (a few integer operations) instead of being hoisted out of the knot loop
    // as loop invariants, spilled, and fetched back from scratch
    int bt = b, lane = lane0;
    asm volatile("" : "+v"(bt), "+v"(lane));
    const int col = lane & ~1, a8 = lane & ~15;     // the pair's LDS column; first lane of this rollout's 16
    const h1s::LaneLds L{lds, 64, lane};
    const double* xb = S.xbar + (size_t)bt * (N + 1) * n;
    const double* ub = S.ubar + (size_t)bt * N * m;
    const double* Kg = S.K + (size_t)bt * N * m * n;
    const double* kg = S.kff + (size_t)bt * N * m;
    double* xc = S.xcand + ((size_t)bt * 8 + ai) * (N + 1) * n;
    double* uc = S.ucand + ((size_t)bt * 8 + ai) * N * m;
    const double* xbt = xb + t * n;
    // ---- everything this step reads from HBM, requested in ONE batch at the very top: this lane's half of the nominal state (for
    // x - xbar), nominal control and feedforward, then the K_t operands of the wave's rollouts (MFMA feedback below; left to itself the
    // scheduler sinks these loads down to their first use, behind the state exchange and its barrier -- a second exposed HBM round
    // trip per step).  The scheduling fences keep the order of issue.
    typedef double v2d_s __attribute__((ext_vector_type(2), aligned(8)));      // rows of K_t have an odd pitch (51 doubles): the pairs are 8-byte aligned only
    // ---- U_r = K_t,r dX_r on v_mfma_f64_4x4x4_4b_f64: FOUR independent 4 x 4 x 4 products per instruction (16 cycles).  Lane layout
    // (tools/probes/mfma_f64_4x4x4_layout.hip): block = (lane >> 2) & 3;  A[i][k] at i = lane & 3, k = lane >> 4;  B[k][n] at k = lane >> 4,
    // n = lane & 3;  D[i][n] at i = lane >> 4, n = lane & 3.
    //   RPW = 4: block = rollout of the wave; per (row group g = 0..4, candidate half c, k-step s): A = K_t[4 g + i][kappa(s, k)],
    //            B = dx_{kappa(s, k)} of candidate 4 c + n, D = U[4 g + i][4 c + n]: 5 x 2 x 13 = 130 products of which 93 % are real
    //            work -- the 16 x 16 x 4 form ran 104 of four times the size, 29 % real (the 8 candidates and 19 rows fill a quarter of
    //            its 16 x 32 tile).
    //   RPW = 1: one rollout; block = row group (0..3, then 4 on block 0): 2 x 2 x 13 = 52 products.
    // kappa: the dealing of the 51 contraction indices to the slots -- k-steps 2 j + h (j = 0..5, h = 0, 1) take 8 j + 2 k + h, so a
    // lane's two operands of a step pair come as ONE 16-byte load and the four k lanes of a row fetch a contiguous 64-byte run; k-step
    // 12 takes 48 + k, and column 51 does not exist: BOTH operands of that slot are zeroed (the A operand fetched there is the first
    // entry of the next row of K -- of the NEXT rollout's gains for the last row of the last knot --, and 0 * NaN = NaN: a diverged
    // neighbour would otherwise poison a healthy rollout, GPU test).  Every operand of the step is requested up front.
    const int mb = (lane >> 2) & 3, mi = lane & 3, mk = lane >> 4;
    constexpr int NG = RPW == 4 ? 5 : 2;                                     // row groups a lane works on
    const int bk = RPW == 4 ? __shfl(bt, 16 * mb) : bt;                      // rollout of this lane's block
    const double* Kt = S.K + ((size_t)bk * N + t) * m * n;
    // (the nominal state / control / feedforward FIRST: vector-memory results return in issue order, and the state exchange below only
    // needs these -- requested behind the 35 K_t loads it waited for all of them)
    h1s::HalfX xh; h1s::HalfU ubh, kfh;
    h1s::load_half(side, xbt, xh);
    load_half_u(side, ub + t * m, ubh);
    load_half_u(side, kg + t * m, kfh);
#ifndef LS_NO_HOIST
    __builtin_amdgcn_sched_barrier(0);
#endif
    v2d_s ka[NG][6];
    double kt[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int row = RPW == 4 ? 4 * g + mi : 4 * (mb + 4 * g) + mi;
      const unsigned off = (unsigned)((row < m ? row : m - 1) * n);          // (rows past 18: a valid row, its product discarded)
#pragma unroll
      for (int j = 0; j < 6; ++j) ka[g][j] = *reinterpret_cast<const v2d_s*>(Kt + off + 8 * j + 2 * mk);
      kt[g] = Kt[off + 48 + (mk < 3 ? mk : 2)];                               // (column 51 does not exist: a valid load, zeroed below)
    }
#ifndef LS_NO_HOIST
    __builtin_amdgcn_sched_barrier(0);
#endif
    // ---- u = ubar + alpha k + K (x - xbar)   (ilqr.cpp:332-333)
    // state deviations of this candidate -> LDS slot j of the pair's column (even lane: shared coordinates + left)
    if (!side) {
#pragma unroll
      for (int k = 0; k < 3; ++k) lds[k * 64 + col] = h.p[k] - xh.p[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) lds[(3 + k) * 64 + col] = h.quat[k] - xh.quat[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) lds[(H1_NQ + k) * 64 + col] = h.vb[k] - xh.vb[k];
      lds[(7 + 10) * 64 + col] = h.q.th11 - xh.q.th11; lds[(H1_NQ + 6 + 10) * 64 + col] = h.q.qd11 - xh.q.qd11;
    }
    const int so = side ? 5 * 64 : 0, sa = side ? 4 * 64 : 0;   // LDS slot offset of this side's leg / arm hinges
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      lds[(7 + k) * 64 + so + col] = h.q.thL[k] - xh.q.thL[k];
      lds[(H1_NQ + 6 + k) * 64 + so + col] = h.q.qdL[k] - xh.q.qdL[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      lds[(7 + 11 + k) * 64 + sa + col] = h.q.thA[k] - xh.q.thA[k];
      lds[(H1_NQ + 6 + 11 + k) * 64 + sa + col] = h.q.qdA[k] - xh.q.qdA[k];
    }
    __syncthreads();
    {
      constexpr int UROW = 52;                        // LDS rows 52..70: the 19 feedback terms, column = the pair's
      const int colB = (RPW == 4 ? 16 * mb : 0) + 2 * mi;                    // LDS column of candidate n of this lane's rollout (+ 8 for the second half)
      double acc[NG][2];
#pragma unroll
      for (int g = 0; g < NG; ++g) { acc[g][0] = 0.0; acc[g][1] = 0.0; }
#pragma unroll
      for (int s13 = 0; s13 < 13; ++s13) {
        const int xrow = s13 < 12 ? 8 * (s13 >> 1) + 2 * mk + (s13 & 1) : 48 + mk;
        double b0 = lds[xrow * 64 + colB], b1 = lds[xrow * 64 + colB + 8];
        if (s13 == 12) { b0 = mk == 3 ? 0.0 : b0; b1 = mk == 3 ? 0.0 : b1; }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          double a = s13 < 12 ? ka[g][s13 >> 1][s13 & 1] : kt[g];
          if (s13 == 12) a = mk == 3 ? 0.0 : a;
          acc[g][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b0, acc[g][0], 0, 0, 0);
          acc[g][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b1, acc[g][1], 0, 0, 0);
        }
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int row = RPW == 4 ? 4 * g + mk : 4 * (mb + 4 * g) + mk;       // D: row index in the lane's upper field
        if (row < m) {
          lds[(UROW + row) * 64 + colB] = acc[g][0];
          lds[(UROW + row) * 64 + colB + 8] = acc[g][1];
        }
      }
      __syncthreads();
      LSS(0)
    }
    h1s::HalfU u;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      // k = 0..4 leg, 5..8 arm, 9 torso (left side only; the right side repeats its last arm row and discards it)
      const int row = k < 5 ? h1s::jleg(side, k) : (k < 9 ? h1s::jarm(side, k - 5) : (side ? 18 : 10));
      const double sfb = lds[(52 + row) * 64 + (RPW == 4 ? col : (col & 15))];      // (RPW = 1: the mirrors read the owners' columns)
      const double ubase = k < 5 ? ubh.uL[k < 5 ? k : 0] : (k < 9 ? ubh.uA[k < 9 && k >= 5 ? k - 5 : 0] : ubh.u11);
      const double kbase = k < 5 ? kfh.uL[k < 5 ? k : 0] : (k < 9 ? kfh.uA[k < 9 && k >= 5 ? k - 5 : 0] : kfh.u11);
      const double ui = ubase + alpha * kbase + sfb;
      if (k < 5) u.uL[k < 5 ? k : 0] = ui;
      else if (k < 9) u.uA[(k >= 5 && k < 9) ? k - 5 : 0] = ui;
      else u.u11 = side ? 0.0 : ui;
      if (live && (k < 9 || !side)) uc[t * m + row] = ui;
    }
    __syncthreads();   // the dynamics step below reuses these LDS rows
    u.u11 = h1s::pair_sum(u.u11);     // torso control: from the left lane to both
    LSS(1)
    // x_t leaves HERE, behind the step's loads and beside its controls -- not at the end of the step that produced it: vector-memory
    // operations retire in order, so the next step's loads (requested a few instructions later) waited for these 33 scattered stores
    // (each touches 32 lines) to be acknowledged: 0.11 of the kernel's 0.70 ms.  From here the dynamics step covers them.
    if constexpr (RPW == 4) {
      // ... and as whole rows: a lane's own stores put 8 bytes into each of 32 candidates' rows per instruction (33 instructions x 32
      // lines through the CU's one address unit, for each of its four waves: 0.1 of the kernel's 0.70 ms).  The wave parks its 32 rows
      // in LDS (free between the feedback and the dynamics step) and writes each out with lanes 0..50 on consecutive doubles; which
      // rollout a row belongs to, and whether it is selected, are wave-uniform per row (scalar address arithmetic).
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      h1s::store_half(side, h, lds + (lane >> 1) * n);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int lv = live ? 1 : 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int br = __builtin_amdgcn_readlane(bt, 16 * r), lr = __builtin_amdgcn_readlane(lv, 16 * r);
        if (lr) {
          double* row0 = S.xcand + ((size_t)br * 8 * (N + 1) + t) * n;
#pragma unroll
          for (int a = 0; a < 8; ++a) {
            const double v = lds[(8 * r + a) * n + (lane < n ? lane : 0)];
            if (lane < n) (row0 + (size_t)a * (N + 1) * n)[lane] = v;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
      if (live) h1s::store_half(side, h, xc + t * n);
    }
    LSS(2)
    int sd = side; asm volatile("" : "+v"(sd));          // (see k_rollout_s)
    const bool side_t = sd != 0;
    step_any<CONTACT>(side_t, h, u, P.dyn, P.stance + b * P.stance_stride + 2 * t, L);
    LSS(3)
    LSS(4)
  }
  if (live) h1s::store_half(side, h, S.xcand + (((size_t)b * 8 + ai) * (N + 1) + N) * n);
  // the candidates' costs are evaluated afterwards, all knots in parallel (launch_cand_costs, dyn_kernels.hip)
#ifdef LS_STAMP
  if (gid == 0) for (int q = 0; q < 8; ++q) S.J[q] = (double)ph[q];
#endif
}

// thread per (rollout, side): nominal rollout
// (the cost of the trajectory is evaluated afterwards, all knots in parallel: launch_nominal_costs, dyn_kernels.hip)
template <int CONTACT>
__global__ void __launch_bounds__(64) k_rollout_s(DevState S, ProblemDev P, int mode, int count_iter) {
  extern __shared__ double lds[];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid >> 1;
  const bool side = (gid & 1) != 0;
  if (b >= S.B || !sel_s(S, b, mode)) return;
  const h1s::LaneLds L{lds, 64, (int)threadIdx.x};
  const int N = S.N;
  double* xb = S.xbar + (size_t)b * (N + 1) * H1_NX;
  const double* ub = S.ubar + (size_t)b * N * H1_NU;
  if (count_iter && !side) S.iters[b] += 1;
  h1s::HalfX h;
  h1s::load_half(side, S.x0 + (size_t)b * H1_NX, h);
  for (int t = 0; t < N; ++t) {
    h1s::HalfU u;
    u.u11 = ub[t * H1_NU + 10];
#pragma unroll
    for (int k = 0; k < 5; ++k) u.uL[k] = ub[t * H1_NU + h1s::jleg(side, k)];
#pragma unroll
    for (int k = 0; k < 4; ++k) u.uA[k] = ub[t * H1_NU + h1s::jarm(side, k)];
    h1s::store_half(side, h, xb + t * H1_NX);       // x_t: behind the step's loads (see k_line_search_s)
    // the side is re-derived behind an opaque barrier every step: otherwise the ~120 per-lane body constants `side ? right : left`
    // are hoisted out of the knot loop as loop invariants, spilled, and fetched back from scratch every step
    int sd = side; asm volatile("" : "+v"(sd));
    const bool side_t = sd != 0;
    step_any<CONTACT>(side_t, h, u, P.dyn, P.stance + b * P.stance_stride + 2 * t, L);
  }
  h1s::store_half(side, h, xb + (size_t)N * H1_NX);
}
// ---- primal dump of the analytic linearisation on two lanes per knot (replaces k_lin_primal_r, dyn_kernels.hip, wherever the
// two-lane kernels run): forward dynamics of the nominal knot, every body's velocity / acceleration / sin, cos / U / 1/D, the base
// rotation, the joint accelerations and the inverse of the pelvis' articulated inertia -> LinDumpG (h1_linearize_dev.h)
// Transposing store of the primal dump.  A lane owns a knot's side, so a plain store instruction puts 64 pieces of 64 different records
// on the CU's store path (3200 lone waves: 0.28 ms, bound by exactly that).  Every lane parks 22 doubles and the offset of their home
// in the dump in its row of a staging area in LDS; the wave then writes the 64 rows out with consecutive lanes on consecutive doubles of
// ONE row: eleven coalesced 16-byte store instructions instead of eleven scattered ones.  The staging area (13 KB per wave) brings the kernel from four
// to three waves per CU -- each of them several times shorter.  Every lane of the wave must call flush() at the same program point.
#define DUMP_STG_LD 26      // row pitch in doubles: 11 chunks of 16 bytes + the offset word; 52 dwords -> 16-byte accesses of 16 lanes hit 64 distinct banks
#define DUMP_STG_BYTES (64 * DUMP_STG_LD * sizeof(double))
struct DumpStage {
  double* stg; double* dump; int lane; bool direct;
  DEVFN void flush(const double* vals, long off) const {      // off < 0: nothing of this lane's is written
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    if (direct) {      // small launches (every wave resident at once, nothing to contend with): the lane stores its own 176 bytes, 13 us less latency
      if (off >= 0) {
        v2d_t* out = reinterpret_cast<v2d_t*>(dump + off);
#pragma unroll
        for (int c = 0; c < 11; ++c) { v2d_t w; w.x = vals[2 * c]; w.y = vals[2 * c + 1]; out[c] = w; }
      }
      return;
    }
    v2d_t* my = reinterpret_cast<v2d_t*>(stg + lane * DUMP_STG_LD);
#pragma unroll
    for (int c = 0; c < 11; ++c) { v2d_t w; w.x = vals[2 * c]; w.y = vals[2 * c + 1]; my[c] = w; }
    stg[lane * DUMP_STG_LD + 22] = __longlong_as_double(off);
    // (one wave per workgroup, LDS operations of a wave execute in order: the reads below see the rows, the next flush's writes come
    // after these reads -- only the compiler has to be kept from moving them across)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // chunk e = 64 r + lane of the 64 x 11 chunks: row e / 11, chunk e % 11 (advanced by 64 = 5 x 11 + 9 per instruction)
    int src = lane / 11, c = lane - 11 * (lane / 11);
#pragma unroll
    for (int r = 0; r < 11; ++r) {
      const double* row = stg + src * DUMP_STG_LD;
      const v2d_t x = reinterpret_cast<const v2d_t*>(row)[c];
      const long o = __double_as_longlong(row[22]);
      if (o >= 0) *reinterpret_cast<v2d_t*>(dump + o + 2 * c) = x;
      c += 9; src += 5;
      if (c >= 11) { c -= 11; src += 1; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
};
// One body's block (see LinDumpG): U_i, 1 / D_i come from this lane's LDS slots, where the inward sweeps left them before the outward
// sweep calls the sink (torso: slot block 0, leg hinge K: 8 + 8 K, arm hinge K: 48 + 8 K; the pelvis has none).
struct DumpSinkS {
  const DumpStage* st; long rec; const h1s::LaneLds* L; bool side, live;
  DEVFN void operator()(int i, const double* v, const double* a, double s, double c) const {
    const int slot = i == 11 ? 0 : (i >= 12 ? 48 + 8 * (i - (side ? 16 : 12)) : 8 + 8 * (i - (side ? 6 : 1)));
    double blk[22];
#pragma unroll
    for (int k = 0; k < 6; ++k) { blk[k] = v[k]; blk[6 + k] = a[k]; blk[12 + k] = i == 0 ? 0.0 : (*L)[(i == 0 ? 0 : slot) + k]; }
    blk[18] = i == 0 ? 0.0 : (*L)[(i == 0 ? 0 : slot) + 6]; blk[19] = s; blk[20] = c; blk[21] = 0.0;
    // pelvis and torso are computed on both lanes: only the left one stores them
    const bool write = live && (!side || (i != 0 && i != 11));
    st->flush(blk, write ? rec + ldg_v(i) : -1L);
  }
};
// LIM (joint-limit rows): the set of hinges the step stops is decided as the step decides it -- on the accelerations of the step without
// the rows, stance rows included --, then the recursion that is dumped runs with those hinges acceleration-prescribed.
template <bool LIM>
__global__ void __launch_bounds__(64) k_lin_primal_s(DevState S, ProblemDev P, int mode, const int* list, const int* count, int direct) {
  extern __shared__ double lds[];
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long knot = gid >> 1;
  const bool side = (gid & 1) != 0;
  // (no lane leaves alone: the dump is written cooperatively; lanes without a knot run along on the last one and store nothing)
  const long total = (long)(list ? *count : S.B) * S.N;
  bool live = knot < total;
  if (!__any(live)) return;
  knot = live ? knot : total - 1;
  const int t = (int)(knot % S.N);
  int b = (int)(knot / S.N);
  if (list) { b = list[b]; mode = MASK_ALL; }      // compacted selection (DevState::order): position -> rollout
  const int f1 = mode == MASK_ALL ? 1 : S.active[b], f2 = mode == MASK_RETRY ? S.need_retry[b] : 1;   // requested with the knot's data, tested after
  const h1s::LaneLds L{lds, 64, (int)threadIdx.x};
  h1s::HalfX h; h1s::load_half(side, S.xbar + ((size_t)b * (S.N + 1) + t) * H1_NX, h);
  h1s::HalfU u; load_half_u(side, S.ubar + ((size_t)b * S.N + t) * H1_NU, u);
  live = live && f1 && f2;
  if (!__any(live)) return;
  const double dt = P.dyn.h;
  const double qn = sqrt(h.quat[0] * h.quat[0] + h.quat[1] * h.quat[1] + h.quat[2] * h.quat[2] + h.quat[3] * h.quat[3]);
  double R0[9]; h1s::quat_R(h.quat[0] / qn, h.quat[1] / qn, h.quat[2] / qn, h.quat[3] / qn, R0);
  h1s::HalfTau tau;
  tau.t11 = h1s::clampu(u.u11, h1s::C_CTRLRANGE[10]) - h1s::DAMPING * h.q.qd11;
#pragma unroll
  for (int k = 0; k < 5; ++k) tau.tL[k] = h1s::clampu(u.uL[k], h1s::C_CTRLRANGE[h1s::jleg(side, k)]) - h1s::DAMPING * h.q.qdL[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) tau.tA[k] = h1s::clampu(u.uA[k], h1s::C_CTRLRANGE[h1s::jarm(side, k)]) - h1s::DAMPING * h.q.qdA[k];
  const long rec = (long)(((size_t)b * S.N + t) * LinDumpG_SIZE);
  // (direct: the launch has no staging area; or the pass is small -- at most one wave per SIMD --, whatever the grid was sized for)
  const DumpStage stage{lds + h1s::LDS_SLOTS * 64, S.lin_dump, (int)threadIdx.x, direct != 0 || total <= 1024 * 32};
  const DumpSinkS sink{&stage, rec, &L, side, live};
  double qb[6], inv36[36], aL[3]; h1s::HalfAcc qa;
  if constexpr (LIM) {
    // The decision pass is the step kernels' own non-inlined lim_accelerations (mask 0: the step without the rows, stance rows included):
    // inlined here beside the reporting recursion it made a 3.4 KB private segment with 854 spilled registers -- past the scratch budget of
    // a wave per SIMD, every launch throttled.  The kernel's frame is now the callee's.
    h1s::HalfTau add;
    {
      const int* stn = P.stance + b * P.stance_stride + 2 * t;
      LimAcc o;
      lim_accelerations<true>(&h, &u, 0u, dt, P.dyn.g[0], P.dyn.g[1], P.dyn.g[2], P.dyn.soft, P.dyn.contact, stn[0], stn[1], P.dyn.mu, P.dyn.lim_k, &o);
      h1s::limit_locks(side, h.q, o.qa, dt, P.dyn.lim_k, tau, add);
    }
    h1s::forward_dynamics_dump<const DumpSinkS, true>(side, R0, h.vb, h.q, tau, h1s::ARMATURE + dt * h1s::DAMPING, P.dyn.g, L, qb, qa, sink, inv36, aL, &add);
  } else
  h1s::forward_dynamics_dump(side, R0, h.vb, h.q, tau, h1s::ARMATURE + dt * h1s::DAMPING, P.dyn.g, L, qb, qa, sink, inv36, aL);
  // the record's header, all of it from the left lane (the right lane's hinge accelerations cross over): four chunks of 22, the last
  // two overlapping (same values) so that nothing beyond the 76 doubles is touched
  double qLr[5], qAr[4];
#pragma unroll
  for (int k = 0; k < 5; ++k) qLr[k] = h1s::xch(qa.qL[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) qAr[k] = h1s::xch(qa.qA[k]);
  const bool wh = live && !side;
  {
    const double c0[22] = {R0[0], R0[1], R0[2], R0[3], R0[4], R0[5], R0[6], R0[7], R0[8], 0.0, aL[0], aL[1], aL[2], 0.0, qb[0], qb[1], qb[2], qb[3], qb[4], qb[5], qa.qL[0], qa.qL[1]};
    stage.flush(c0, wh ? rec + 0 : -1L);
    const double c1[22] = {qa.qL[2], qa.qL[3], qa.qL[4], qLr[0], qLr[1], qLr[2], qLr[3], qLr[4], qa.q11, qa.qA[0], qa.qA[1], qa.qA[2], qa.qA[3], qAr[0], qAr[1], qAr[2], qAr[3], 0.0,
                           inv36[0], inv36[1], inv36[2], inv36[3]};
    stage.flush(c1, wh ? rec + 22 : -1L);
    stage.flush(inv36 + 4, wh ? rec + 44 : -1L);
    stage.flush(inv36 + 14, wh ? rec + 54 : -1L);
  }
}
__global__ void __launch_bounds__(64) k_count_iter(DevState S, int mode) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < S.B && sel_s(S, b, mode)) S.iters[b] += 1;
}

// two lanes per item: plain batched step with explicit stance flags (stage API / plant of the closed loop)
template <int CK>     // CK: the CONTACT value of step_any (0: the constraint-free plant -- an instantiation that cannot reach the constrained step keeps its private segment small; 1..4)
__global__ void __launch_bounds__(64) k_step_s(int count, const double* x, const double* u, DynParams dyn, double* xn, int st_l, int st_r) {
  extern __shared__ double lds[];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = gid >> 1;
  const bool side = (gid & 1) != 0;
  if (i >= count) return;
  const h1s::LaneLds L{lds, 64, (int)threadIdx.x};
  h1s::HalfX h; h1s::load_half(side, x + (size_t)i * H1_NX, h);
  h1s::HalfU uu; load_half_u(side, u + (size_t)i * H1_NU, uu);
  const int st[2] = {st_l, st_r};
  step_any<CK>(side, h, uu, dyn, st, L);
  h1s::store_half(side, h, xn + (size_t)i * H1_NX);
}
// last knot of the warm start: xbar[N] = f(xbar[N-1], ubar[N-1])  (ilqr.cpp:72-80)
template <int CK>
__global__ void __launch_bounds__(64) k_last_step_s(DevState S, ProblemDev P) {
  extern __shared__ double lds[];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid >> 1;
  const bool side = (gid & 1) != 0;
  if (b >= S.B) return;
  const h1s::LaneLds L{lds, 64, (int)threadIdx.x};
  const int N = S.N;
  h1s::HalfX h; h1s::load_half(side, S.xbar + ((size_t)b * (N + 1) + N - 1) * H1_NX, h);
  h1s::HalfU uu; load_half_u(side, S.ubar + ((size_t)b * N + N - 1) * H1_NU, uu);
  step_any<CK>(side, h, uu, P.dyn, P.stance + b * P.stance_stride + 2 * (N - 1), L);
  h1s::store_half(side, h, S.xbar + ((size_t)b * (N + 1) + N) * H1_NX);
}
// Reference-style forward differences (RobotUtils::linearizeDynamicsFD, robot_utils.cpp:120-160) on the two-lane step:
// a lane pair per (knot, column); columns 0..50 perturb x, 51..69 perturb u, column 70 is the unperturbed step (evaluated
// once per knot, as the reference does).  The stepped states land in A / B / the knot's lin_dump record; k_fd_finish turns
// them into difference quotients.
DEVFN void perturb_half(bool side, h1s::HalfX& h, h1s::HalfU& u, int col, double eps) {
#pragma unroll
  for (int k = 0; k < 3; ++k) h.p[k] += (col == k) ? eps : 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) h.quat[k] += (col == 3 + k) ? eps : 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) h.vb[k] += (col == H1_NQ + k) ? eps : 0.0;
  h.q.th11 += (col == 7 + 10) ? eps : 0.0; h.q.qd11 += (col == H1_NQ + 6 + 10) ? eps : 0.0;
#pragma unroll
  for (int k = 0; k < 5; ++k) { const int j = h1s::jleg(side, k); h.q.thL[k] += (col == 7 + j) ? eps : 0.0; h.q.qdL[k] += (col == H1_NQ + 6 + j) ? eps : 0.0; u.uL[k] += (col == H1_NX + j) ? eps : 0.0; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int j = h1s::jarm(side, k); h.q.thA[k] += (col == 7 + j) ? eps : 0.0; h.q.qdA[k] += (col == H1_NQ + 6 + j) ? eps : 0.0; u.uA[k] += (col == H1_NX + j) ? eps : 0.0; }
  u.u11 += (col == H1_NX + 10) ? eps : 0.0;
}
// store this lane's half of a state as column `col` (stride ld) of a row-major matrix
DEVFN void store_half_col(bool side, const h1s::HalfX& h, double* M, int ld, int col) {
  if (!side) {
#pragma unroll
    for (int k = 0; k < 3; ++k) M[k * ld + col] = h.p[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) M[(3 + k) * ld + col] = h.quat[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) M[(H1_NQ + k) * ld + col] = h.vb[k];
    M[(7 + 10) * ld + col] = h.q.th11; M[(H1_NQ + 6 + 10) * ld + col] = h.q.qd11;
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) { M[(7 + h1s::jleg(side, k)) * ld + col] = h.q.thL[k]; M[(H1_NQ + 6 + h1s::jleg(side, k)) * ld + col] = h.q.qdL[k]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { M[(7 + h1s::jarm(side, k)) * ld + col] = h.q.thA[k]; M[(H1_NQ + 6 + h1s::jarm(side, k)) * ld + col] = h.q.qdA[k]; }
}
#define FD_NCOL (H1_NX + H1_NU + 1)
template <int CK>
__global__ void __launch_bounds__(64) k_fd_steps_s(DevState S, ProblemDev P, int mode, double eps, int dump_doubles) {
  extern __shared__ double lds[];
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long pr = gid >> 1;
  const bool side = (gid & 1) != 0;
  const long total = (long)S.B * S.N * FD_NCOL;
  const long prc = pr < total ? pr : total - 1;
  const int col = (int)(prc % FD_NCOL);
  const long item = prc / FD_NCOL;
  const int t = (int)(item % S.N), b = (int)(item / S.N);
  const bool act = pr < total && sel_s(S, b, mode);
  if (!__any(act)) return;
  const h1s::LaneLds L{lds, 64, (int)threadIdx.x};
  h1s::HalfX h; h1s::load_half(side, S.xbar + ((size_t)b * (S.N + 1) + t) * H1_NX, h);
  h1s::HalfU uu; load_half_u(side, S.ubar + ((size_t)b * S.N + t) * H1_NU, uu);
  perturb_half(side, h, uu, col, eps);
  step_any<CK>(side, h, uu, P.dyn, P.stance + b * P.stance_stride + 2 * t, L);
  if (!act) return;
  if (col < H1_NX) store_half_col(side, h, S.A + (size_t)item * H1_NX * H1_NX, H1_NX, col);
  else if (col < H1_NX + H1_NU) store_half_col(side, h, S.Bm + (size_t)item * H1_NX * H1_NU, H1_NU, col - H1_NX);
  else h1s::store_half(side, h, S.lin_dump + (size_t)item * dump_doubles);
}
__global__ void __launch_bounds__(256) k_fd_finish(DevState S, int mode, double eps, int dump_doubles) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int NC = H1_NX + H1_NU;
  const long total = (long)S.B * S.N * H1_NX * NC;
  if (gid >= total) return;
  const int c = (int)(gid % NC);
  const int i = (int)((gid / NC) % H1_NX);
  const long item = gid / ((long)NC * H1_NX);
  const int b = (int)(item / S.N);
  if (!sel_s(S, b, mode)) return;
  const double base = S.lin_dump[(size_t)item * dump_doubles + i];
  double* e = c < H1_NX ? S.A + ((size_t)item * H1_NX + i) * H1_NX + c : S.Bm + ((size_t)item * H1_NX + i) * H1_NU + (c - H1_NX);
  *e = (*e - base) / eps;
}

static inline int cdiv_s(long a, long b) { return (int)((a + b - 1) / b); }
int dyn_split_kernels_set_attr() {
  int rc = 0;
  const int lds = (int)DYN_LDS_BYTES_S;
  rc |= hipFuncSetAttribute((const void*)k_line_search_s<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_line_search_s<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_line_search_s<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_line_search_s<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_line_search_s<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_line_search_s<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_line_search_s<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_line_search_s<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_line_search_s<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_line_search_s<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_line_search_s<5, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_line_search_s<5, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_rollout_s<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_rollout_s<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_rollout_s<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_rollout_s<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_rollout_s<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_rollout_s<5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_step_s<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_step_s<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_step_s<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_step_s<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_step_s<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_step_s<5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_last_step_s<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_last_step_s<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_last_step_s<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_last_step_s<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_last_step_s<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_last_step_s<5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_fd_steps_s<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_fd_steps_s<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_fd_steps_s<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_fd_steps_s<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_fd_steps_s<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_fd_steps_s<5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_lin_primal_s<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DYN_LDS_BYTES_S + DUMP_STG_BYTES)) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_lin_primal_s<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DYN_LDS_BYTES_S + DUMP_STG_BYTES)) != hipSuccess;
  return rc;
}
// CONTACT / CK of the kernels: 0 constraint-free; 1 / 2 stance rows (2: kinetic friction, mode 4); 3 / 4 the same with joint-limit rows;
// 5 joint-limit rows on the constraint-free plant
static int step_kind(const DynParams& d) { return !constrained(d) ? 0 : (d.contact == 0 ? 5 : (d.contact == 4 ? 2 : 1) + (d.limits ? 2 : 0)); }
#ifndef LS_RPW1_MAX_BATCH
#define LS_RPW1_MAX_BATCH 1024      // one wave per SIMD on the 1024 SIMDs of an MI355X
#endif
void launch_line_search_s(const DevState& S, const ProblemDev& P, int mode, hipStream_t st, const int* list, const int* count, int max_rollouts) {
  // a rollout per wave (see the kernel) for small batches -- and for the tail of an early-exit solve of a large one, where the
  // host knows an upper bound of the compacted list's length (the selected rollouts are its first entries: blocks past the
  // list's true count leave at once); the results do not depend on the choice
  const int nsel = (list && max_rollouts >= 0 && max_rollouts < S.B) ? max_rollouts : S.B;
  const int ck = step_kind(P.dyn);      // the CONTACT value of step_any
  if (nsel <= LS_RPW1_MAX_BATCH) {
    const int blocks = nsel > 0 ? nsel : 1;
    if (ck == 5) hipLaunchKernelGGL((k_line_search_s<5, 1>), dim3(blocks), dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, list, count);
    else if (ck == 4) hipLaunchKernelGGL((k_line_search_s<4, 1>), dim3(blocks), dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, list, count);
    else if (ck == 3) hipLaunchKernelGGL((k_line_search_s<3, 1>), dim3(blocks), dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, list, count);
    else if (ck == 2) hipLaunchKernelGGL((k_line_search_s<2, 1>), dim3(blocks), dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, list, count);
    else if (ck == 1) hipLaunchKernelGGL((k_line_search_s<1, 1>), dim3(blocks), dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, list, count);
    else hipLaunchKernelGGL((k_line_search_s<0, 1>), dim3(blocks), dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, list, count);
    return;
  }
  const dim3 grid(cdiv_s((long)S.B * 16, 64));
  if (ck == 5) hipLaunchKernelGGL((k_line_search_s<5, 4>), grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, list, count);
  else if (ck == 4) hipLaunchKernelGGL((k_line_search_s<4, 4>), grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, list, count);
  else if (ck == 3) hipLaunchKernelGGL((k_line_search_s<3, 4>), grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, list, count);
  else if (ck == 2) hipLaunchKernelGGL((k_line_search_s<2, 4>), grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, list, count);
  else if (ck == 1) hipLaunchKernelGGL((k_line_search_s<1, 4>), grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, list, count);
  else hipLaunchKernelGGL((k_line_search_s<0, 4>), grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, list, count);
}
void launch_lin_primal_s(const DevState& S, const ProblemDev& P, int mode, hipStream_t st, const int* list, const int* count) {
  // up to one wave per SIMD (32 knots per wave): the lanes store their blocks themselves, without the staging area (and its LDS: four waves per CU)
  const long waves = cdiv_s((long)S.B * S.N * 2, 64);
  const int direct = waves <= 1024 ? 1 : 0;
  if (P.dyn.limits) hipLaunchKernelGGL(k_lin_primal_s<true>, dim3(waves), dim3(64), DYN_LDS_BYTES_S + (direct ? 0 : DUMP_STG_BYTES), st, S, P, mode, list, count, direct);
  else hipLaunchKernelGGL(k_lin_primal_s<false>, dim3(waves), dim3(64), DYN_LDS_BYTES_S + (direct ? 0 : DUMP_STG_BYTES), st, S, P, mode, list, count, direct);
}
void launch_step_s(int count, const double* x, const double* u, const DynParams& dyn, double* xn, hipStream_t st, int stance_l, int stance_r) {
  const dim3 grid(cdiv_s((long)count * 2, 64));
  switch (step_kind(dyn)) {
    case 5: hipLaunchKernelGGL(k_step_s<5>, grid, dim3(64), DYN_LDS_BYTES_S, st, count, x, u, dyn, xn, stance_l, stance_r); break;
    case 4: hipLaunchKernelGGL(k_step_s<4>, grid, dim3(64), DYN_LDS_BYTES_S, st, count, x, u, dyn, xn, stance_l, stance_r); break;
    case 3: hipLaunchKernelGGL(k_step_s<3>, grid, dim3(64), DYN_LDS_BYTES_S, st, count, x, u, dyn, xn, stance_l, stance_r); break;
    case 2: hipLaunchKernelGGL(k_step_s<2>, grid, dim3(64), DYN_LDS_BYTES_S, st, count, x, u, dyn, xn, stance_l, stance_r); break;
    case 1: hipLaunchKernelGGL(k_step_s<1>, grid, dim3(64), DYN_LDS_BYTES_S, st, count, x, u, dyn, xn, stance_l, stance_r); break;
    default: hipLaunchKernelGGL(k_step_s<0>, grid, dim3(64), DYN_LDS_BYTES_S, st, count, x, u, dyn, xn, stance_l, stance_r);
  }
}
void launch_last_step_s(const DevState& S, const ProblemDev& P, hipStream_t st) {
  const dim3 grid(cdiv_s((long)S.B * 2, 64));
  switch (step_kind(P.dyn)) {
    case 5: hipLaunchKernelGGL(k_last_step_s<5>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P); break;
    case 4: hipLaunchKernelGGL(k_last_step_s<4>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P); break;
    case 3: hipLaunchKernelGGL(k_last_step_s<3>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P); break;
    case 2: hipLaunchKernelGGL(k_last_step_s<2>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P); break;
    case 1: hipLaunchKernelGGL(k_last_step_s<1>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P); break;
    default: hipLaunchKernelGGL(k_last_step_s<0>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P);
  }
}
void launch_linearize_fd_s(const DevState& S, const ProblemDev& P, int mode, double eps, hipStream_t st) {
  const int dd = (int)lin_dump_doubles();
  const dim3 grid(cdiv_s((long)S.B * S.N * FD_NCOL * 2, 64));
  switch (step_kind(P.dyn)) {
    case 5: hipLaunchKernelGGL(k_fd_steps_s<5>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, eps, dd); break;
    case 4: hipLaunchKernelGGL(k_fd_steps_s<4>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, eps, dd); break;
    case 3: hipLaunchKernelGGL(k_fd_steps_s<3>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, eps, dd); break;
    case 2: hipLaunchKernelGGL(k_fd_steps_s<2>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, eps, dd); break;
    case 1: hipLaunchKernelGGL(k_fd_steps_s<1>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, eps, dd); break;
    default: hipLaunchKernelGGL(k_fd_steps_s<0>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, eps, dd);
  }
  hipLaunchKernelGGL(k_fd_finish, dim3(cdiv_s((long)S.B * S.N * H1_NX * (H1_NX + H1_NU), 256)), dim3(256), 0, st, S, mode, eps, dd);
}
void launch_rollout_s(const DevState& S, const ProblemDev& P, int mode, int do_roll, int count_iter, double* cost_out, hipStream_t st) {
  const dim3 grid(cdiv_s((long)S.B * 2, 64));
  const int ck = step_kind(P.dyn);
  if (do_roll && ck == 5) hipLaunchKernelGGL(k_rollout_s<5>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, count_iter);
  else if (do_roll && ck == 4) hipLaunchKernelGGL(k_rollout_s<4>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, count_iter);
  else if (do_roll && ck == 3) hipLaunchKernelGGL(k_rollout_s<3>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, count_iter);
  else if (do_roll && ck == 2) hipLaunchKernelGGL(k_rollout_s<2>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, count_iter);
  else if (do_roll && ck == 1) hipLaunchKernelGGL(k_rollout_s<1>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, count_iter);
  else if (do_roll) hipLaunchKernelGGL(k_rollout_s<0>, grid, dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, count_iter);
  else if (count_iter) hipLaunchKernelGGL(k_count_iter, dim3(cdiv_s(S.B, 64)), dim3(64), 0, st, S, mode);
  launch_nominal_costs(S, P, mode, cost_out, st);
}

}  // namespace ilqr
