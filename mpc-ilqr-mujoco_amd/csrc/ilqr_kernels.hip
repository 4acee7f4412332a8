// HIP kernels of the batched H1 iLQR for gfx950 (fp64) -- the hot path behind include/ilqr_hip.h.
//
//   k_rollout          iLQR::forwardRolloutNominal + computeTotalCost   reference src/ilqr/ilqr.cpp:119-124, 363-518
//   k_linearize        iLQR::computeLinearization                       reference src/ilqr/ilqr.cpp:126-131
//   k_cost_quadratics  iLQR::computeCostQuadratics                      reference src/ilqr/ilqr.cpp:133-244
//   k_backward         iLQR::backwardPass                               reference src/ilqr/ilqr.cpp:250-309
//   k_line_search      iLQR::forwardPassLineSearch (8 alphas at once)   reference src/ilqr/ilqr.cpp:311-361
//   k_control          iLQR::solve bookkeeping (lambda, accept, exit)   reference src/ilqr/ilqr.cpp:547-656
//   k_compute_control  MPC::stepOnce control law                        reference src/ilqr/mpc.cpp:97-101
//
// HBM layout: every per-rollout array is rollout-major, row-major inside (b, t, i, j); one work
// item (thread, wave or workgroup, per kernel) owns one rollout / knot and streams its own slab.
#include <hip/hip_runtime.h>

#include <cstring>

#include <cstdlib>

#include "h1_cost_dev.h"
#include "h1_linearize_dev.h"
#include "h1_linearize_contact_dev.h"
#include <type_traits>
#include "ilqr_kernels.h"
#include "riccati_pack.h"

using namespace h1;

namespace ilqr {

#ifndef LINT_WAVES
#define LINT_WAVES 1
#endif

__device__ __forceinline__ bool selected(const DevState& S, int b, int mode) {
  if (mode == MASK_ALL) return true;
  if (mode == MASK_ACTIVE) return S.active[b] != 0;
  return S.active[b] != 0 && S.need_retry[b] != 0;
}

// ------------------------------------------------------------------ K1: nominal rollout + cost
// thread per rollout. mode MASK_ALL also used by the stage API. do_roll = 0 -> cost only.
#ifdef ILQR_LEGACY_KERNELS
__global__ void __launch_bounds__(64) k_rollout(DevState S, ProblemDev P, int mode, int do_roll, int count_iter, double* cost_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= S.B || !selected(S, b, mode)) return;
  const int N = S.N;
  double* xb = S.xbar + (size_t)b * (N + 1) * H1_NX;
  const double* ub = S.ubar + (size_t)b * N * H1_NU;
  if (count_iter) S.iters[b] += 1;
  double x[H1_NX], xn[H1_NX], u[H1_NU];
  if (do_roll) { for (int i = 0; i < H1_NX; ++i) { x[i] = S.x0[(size_t)b * H1_NX + i]; xb[i] = x[i]; } }
  else { for (int i = 0; i < H1_NX; ++i) x[i] = xb[i]; }
  double c = 0.0;
  for (int t = 0; t < N; ++t) {
    for (int i = 0; i < H1_NU; ++i) u[i] = ub[t * H1_NU + i];
    c += knot_cost(P, b, t, x, u);
    if (do_roll) {
      step<double>(x, u, P.dyn, xn, P.stance + b * P.stance_stride + 2 * t);
      for (int i = 0; i < H1_NX; ++i) { x[i] = xn[i]; xb[(t + 1) * H1_NX + i] = xn[i]; }
    } else {
      for (int i = 0; i < H1_NX; ++i) x[i] = xb[(t + 1) * H1_NX + i];
    }
  }
  c += knot_cost(P, b, N, x, nullptr);
  cost_out[b] = c;
}
#endif

// plain batched step for the stage API
#ifdef ILQR_LEGACY_KERNELS
__global__ void __launch_bounds__(64) k_step(int count, const double* x, const double* u, DynParams dyn, double* xn, int stance_l, int stance_r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double xl[H1_NX], ul[H1_NU], out[H1_NX];
  for (int k = 0; k < H1_NX; ++k) xl[k] = x[(size_t)i * H1_NX + k];
  for (int k = 0; k < H1_NU; ++k) ul[k] = u[(size_t)i * H1_NU + k];
  const int stance[2] = {stance_l, stance_r};
  step<double>(xl, ul, dyn, out, stance);
  for (int k = 0; k < H1_NX; ++k) xn[(size_t)i * H1_NX + k] = out[k];
}
#endif

// ------------------------------------------------------------------ K2: dynamics Jacobians
// Analytic mode = two kernels (h1_linearize_dev.h):
//   k_lin_primal : thread per knot; forward dynamics of the nominal knot, dumping the primal per-body
//                  quantities (velocities, accelerations, forces, articulated-body U / D / pelvis inverse).
//   k_lin_tangent: one wave per knot; lanes = columns of Minv (unit-force sweeps) and tangent directions
//                  (tangent RNEA sweeps), then lanes = Jacobian columns.
// -DLIN_STAMP: diagnostic build only -- per-phase cycle counts of workgroup (0, 0) land in S.J[0..7]
#ifdef LIN_STAMP
#define LSTAMP(k) { const long long tn_ = clock64(); if (t == 0 && b == 0 && tid == 0) S.J[k] = (double)(tn_ - qlast); qlast = tn_; }
#else
#define LSTAMP(k)
#endif
// Two waves per knot share the knot's LDS record (22.7 KB: the LDS, not the registers, bounds the occupancy -- 7 one-wave
// workgroups per CU before, now 6 x 2 waves): wave 0 accumulates the inverse-dynamics forces and sweeps the legs, wave 1 runs
// the Minv sweeps and sweeps torso + arms; the Minv product is split by row tile, the 51 columns of A go to wave 0 and the 19
// of B to wave 1.  A knot's critical path drops from ~107 k to ~65 k cycles and twice as many waves hide each other's latencies.
#ifdef ILQR_LEGACY_KERNELS
__global__ void __launch_bounds__(128, 3) k_lin_tangent(DevState S, ProblemDev P, int mode, const int* list, const int* count) {
  const int t = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  int b = blockIdx.y;
  if (list) {                        // compacted selection (DevState::order)
    if (b >= *count) return;
    b = list[b];
    mode = MASK_ALL;
  }
  __shared__ LinShared L;
#ifdef LIN_STAMP
  long long qlast = clock64();
#endif
  const size_t knot = (size_t)b * S.N + t;
  // the rollout's selection flags travel with the knot's data instead of ahead of it (one HBM round trip less per workgroup)
  if (!lin_load_dump2(L, S.lin_dump + knot * LinDumpG_SIZE, tid, S.xbar + ((size_t)b * (S.N + 1) + t) * H1_NX, S.ubar + ((size_t)b * S.N + t) * H1_NU,
                      mode == MASK_ALL ? nullptr : S.active + b, mode == MASK_RETRY ? S.need_retry + b : nullptr)) return;
  if (tid == 127) L.h = P.dyn.h;
  for (int e = tid; e < (H1_NV - 6) * LIN_LD; e += 128) (&L.dT[6][0])[e] = 0.0;     // lin_tangent_zero
  __syncthreads();
  LSTAMP(0)
  if (wv == 0) {
    lin_accumulate_forces_w(L, lane);
    LSTAMP(1)
    if (lane == 32) lin_prologue(L);
    LSTAMP(2)
  } else {
    lin_minv_lane(L, lane);       // lanes 0..24: columns of Minv
  }
  __syncthreads();
  LSTAMP(3)
  if (wv == 0) lin_tangent_legs(L, lane); else lin_tangent_arms(L, lane);
  __syncthreads();
  if (wv == 0) lin_tangent_pelvis(L, lane);
  __syncthreads();
  LSTAMP(4)
  lin_apply_minv_2(L, tid);
  __syncthreads();
  LSTAMP(5)
  double* Ag = S.A + knot * H1_NX * H1_NX;
  double* Bg = S.Bm + knot * H1_NX * H1_NU;
  // each lane streams one column; for a fixed row the lanes write consecutive addresses
  if (wv == 0 && lane < H1_NX) lin_column(L, 0, lane, [&](int r, double v) { Ag[r * H1_NX + lane] = v; });
  if (wv == 1 && lane < H1_NU) lin_column(L, 1, lane, [&](int r, double v) { Bg[r * H1_NU + lane] = v; });
  LSTAMP(6)
}
#endif

// Column stores of the two-knot tangent kernels.  PACK (inside a solve whose backward pass is the operand-layout Riccati kernel,
// riccati_pack.hip): column c of A_t goes to column slot pk_state_slot(c) of the packed image A~, row r to row slot pk_state_slot(r)
// -- the 22 position rows (copies of their velocity rows through the integrator, lin_column) have no slot and are not written;
// columns 0..15 of B_t go to the packed B0, columns 16..18 to column slots 60..62 of A~.  For a fixed row the lanes of a column tile
// write consecutive addresses, as in the standard layout.  (Padding slots are zeroed once by k_pack_zero_pads when a handle's
// buffers change over to this layout: nothing ever writes them afterwards.)
template <bool PACK, class Cfn>
DEVFN void lin2_store_A(const LinShared& L, int c, double* Ag, Cfn&& call) {
  if (PACK) {
    const int C = pk_state_slot(c);
    double* col = pk_align(Ag) + (C >> 4) * 512 + (C & 15);
    call([&](int r, double v) { const int R = pk_state_slot(r); if (R >= 32) col[(R - 32) * 16] = v; });
  } else {
    call([&](int r, double v) { Ag[r * H1_NX + c] = v; });
  }
}
template <bool PACK, class Cfn>
DEVFN void lin2_store_B(const LinShared& L, int c, double* Ag, double* Bg, Cfn&& call) {
  if (PACK) {
    double* col = c < 16 ? pk_align(Bg) + c : pk_align(Ag) + 3 * 512 + 12 + (c - 16);
    call([&](int r, double v) { const int R = pk_state_slot(r); if (R >= 32) col[(R - 32) * 16] = v; });
  } else {
    call([&](int r, double v) { Bg[r * H1_NU + c] = v; });
  }
}

// Round 4: TWO knots per four-wave workgroup (h1_linearize_dev.h "two knots per four-wave workgroup").  The one-knot kernel
// above issued ~6600 vector instructions per knot with 47 / 38 / 25 of 64 lanes active and sat at 70 % VALU busy: it was bound by
// its instruction count.  Without the three base-linear-velocity directions a chain group has 16 slots, so the lane-parallel
// phases of two knots pack into one wave each:
//   wave 0  force accumulation of both knots (2 x 20 lanes) | BAR | tangent sweeps of the legs of both knots (64 lanes)
//   wave 1                                                  | BAR | tangent sweeps of torso + arms of both knots (64 lanes)
//   wave 2  Minv columns of both knots, inward + pelvis     | BAR | ... outward sweeps (2 x 25 lanes), beside the tangent sweeps
//   wave 3  integrator prologue of both knots (2 lanes)     | BAR |
// then pelvis rows (wave = knot), the Minv product on the MFMA (wave = knot x row tile), the columns: A of knot 0 on wave 0,
// A of knot 1 on wave 1, B of both knots on wave 2.  Work items are (selected rollout, knot) pairs in rollout-major order.
// PACK: operand layout of riccati_pack.h (lin2_store_A / lin2_store_B above); getters and the other kernel families get the standard
// layout back from k_unpack_ab.
// LIM: joint-limit rows (DynParams::limits).  The dump then comes from the recursion with the stopped hinges acceleration-prescribed
// (1 / D_i = 2^-1000: their Minv rows and columns vanish by themselves); what is left to carry is d qacc_i = -1 / h in the direction of
// a stopped hinge's own rate -- seeded into the inverse-dynamics tangent (tan_body_fwd*) and written into its row of d qacc.  An
// instantiation of its own: the default kernel keeps its machine code.
template <bool LIM> struct LinLockOpt { double c[2][H1_NB]; DEVFN double (*get())[H1_NB] { return c; } };
template <> struct LinLockOpt<false> { DEVFN double (*get())[H1_NB] { return nullptr; } };
// (lim_k, the restoring stiffness: qacc_i = -v_i / h - lim_k r_i on a constrained hinge, so its own angle carries d qacc_i = -lim_k as well;
// c[0] -- the pelvis has no hinge -- carries h lim_k to the seeds of tan_body_fwd*)
template <bool LIM>
DEVFN void lin_lock_flags(LinShared& L, double* c, int tid7, double h, double lim_k) {
  if constexpr (LIM) { if (tid7 < H1_NB) c[tid7] = tid7 == 0 ? h * lim_k : ((L.u.m.Dinv[tid7] < 1e-200) ? -1.0 / h : 0.0); }
}
template <bool LIM>
DEVFN void lin_lock_rows(LinShared& L, const double* c, int tid7) {
  if constexpr (LIM) {
    if (tid7 >= 1 && tid7 < H1_NB && c[tid7] != 0.0) {
      L.dT[5 + tid7][dir_lane(DIR_THETADOT, tid7)] = c[tid7];
      if (c[0] != 0.0) L.dT[5 + tid7][dir_lane(DIR_THETA, tid7)] = c[tid7] * c[0];
    }
  }
}
template <bool PACK, bool LIM = false>
__global__ void __launch_bounds__(256, 3) k_lin_tangent2(DevState S, ProblemDev P, int mode, const int* list, const int* count) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int ks = tid >> 7, tid7 = tid & 127;                   // knot slot of this thread in the 128-thread phases
  const unsigned N = (unsigned)S.N;
  const unsigned total = (unsigned)(list ? *count : S.B) * N;
  const unsigned it0 = 2u * blockIdx.x;
  if (it0 >= total) return;
  __shared__ LinShared L2[2];
  __shared__ LinLockOpt<LIM> LKO;
  double (*lockc2)[H1_NB] = LKO.get();
#ifdef LIN_STAMP
  long long qlast = clock64();
  const int t = (int)(it0 % N), b = (int)(it0 / N);
#endif
  // both slots always compute (an odd item count: slot 1 repeats slot 0's knot); `valid` gates the stores only
  size_t knot[2]; bool valid[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const unsigned it = it0 + k < total ? it0 + k : it0;
    const unsigned bs = it / N, tt = it - bs * N;
    const int bb = list ? list[bs] : (int)bs;
    knot[k] = (size_t)bb * N + tt;
    bool ok = (it0 + k < total);
    if (!list && mode != MASK_ALL) ok = ok && S.active[bb] != 0 && (mode != MASK_RETRY || S.need_retry[bb] != 0);
    valid[k] = ok;
  }
  if (!valid[0] && !valid[1]) return;
  {
    const size_t b0 = knot[0] / N, t0 = knot[0] - b0 * N, b1 = knot[1] / N, t1 = knot[1] - b1 * N;
    lin2_load_dump(L2, S.lin_dump + knot[0] * LinDumpG_SIZE, S.lin_dump + knot[1] * LinDumpG_SIZE, tid, S.xbar + (b0 * (N + 1) + t0) * H1_NX, S.ubar + knot[0] * H1_NU,
                   S.xbar + (b1 * (N + 1) + t1) * H1_NX, S.ubar + knot[1] * H1_NU);
    if (tid7 == 127) L2[ks].h = P.dyn.h;
    for (int e = tid7; e < H1_NV * LIN_LD; e += 128) (&L2[ks].dT[0][0])[e] = 0.0;     // all rows: the v_lin columns are written by nobody
  }
  __syncthreads();
  LSTAMP(0)
  if constexpr (LIM) lin_lock_flags<LIM>(L2[ks], lockc2[ks], tid7, P.dyn.h, P.dyn.lim_k);
  // (one barrier for all four waves between the two halves: a __syncthreads() inside each wave's own branch happens to work --
  // s_barrier counts waves -- but matching barriers across divergent code paths is not something to lean on)
  MinvCarry Cm;
  const int cm = lane & 31;
  if (wv == 0) {
    lin2_accumulate_forces_w(L2, lane);
    LSTAMP(1)
  } else if (wv == 2) {
    if (cm < H1_NV) lin2_minv_in(L2[lane >> 5], cm, Cm);
  } else if (wv == 3) {
    if (cm == 0) lin_prologue(L2[lane >> 5]);
  }
  __syncthreads();
  if (wv == 0) lin2_tangent_legs<LIM>(L2, lane, lockc2);
  else if (wv == 1) lin2_tangent_arms<LIM>(L2, lane, lockc2);
  else if (wv == 2) { if (cm < H1_NV) lin2_minv_out(L2[lane >> 5], cm, Cm); }
  __syncthreads();
  LSTAMP(2)
  if (wv < 2) lin2_tangent_pelvis(L2[wv], lane);
  __syncthreads();
  LSTAMP(3)
  lin_apply_minv_2(L2[ks], tid7);
  __syncthreads();
  if constexpr (LIM) { lin_lock_rows<LIM>(L2[ks], lockc2[ks], tid7); __syncthreads(); }
  LSTAMP(4)
  // each lane streams one column; for a fixed row the lanes write consecutive addresses
  if (wv < 2) {
    if (lane < H1_NX && valid[wv]) lin2_store_A<PACK>(L2[wv], lane, S.A + knot[wv] * H1_NX * H1_NX, [&](auto&& out) { lin_column(L2[wv], 0, lane, out); });
  } else if (wv == 2) {
    const int k = lane >> 5, c = lane & 31;
    if (c < H1_NU && valid[k]) lin2_store_B<PACK>(L2[k], c, S.A + knot[k] * H1_NX * H1_NX, S.Bm + knot[k] * H1_NX * H1_NU, [&](auto&& out) { lin_column(L2[k], 1, c, out); });
  }
  LSTAMP(5)
}

// Contact row f4: analytic Jacobians of the stance-constrained step (h1_linearize_contact_dev.h).  Same two-wave layout; the
// primal dump is the free solve, the multipliers and the constrained accelerations are rebuilt here (twelve unit-wrench
// lanes beside the 25 Minv lanes), the tangent sweeps carry the contact wrench as an external force and collect the
// constraint-row tangents, and the final product is -Minv dT + G dlambda.
#ifdef ILQR_LEGACY_KERNELS
__global__ void __launch_bounds__(128, 2) k_lin_tangent_c(DevState S, ProblemDev P, int mode, const int* list, const int* count) {
  const int t = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  int b = blockIdx.y;
  if (list) {                        // compacted selection (DevState::order)
    if (b >= *count) return;
    b = list[b];
    mode = MASK_ALL;
  }
  __shared__ LinShared L;
  __shared__ LinContact Cc;
#ifdef LIN_STAMP
  long long qlast = clock64();
#endif
  const size_t knot = (size_t)b * S.N + t;
  if (!lin_load_dump2c(L, Cc, S.lin_dump + knot * LinDumpG_SIZE, tid, S.xbar + ((size_t)b * (S.N + 1) + t) * H1_NX, S.ubar + ((size_t)b * S.N + t) * H1_NU,
                       mode == MASK_ALL ? nullptr : S.active + b, mode == MASK_RETRY ? S.need_retry + b : nullptr)) return;
  if (tid == 127) L.h = P.dyn.h;
  for (int e = tid; e < (H1_NV - 6) * LIN_LD; e += 128) (&L.dT[6][0])[e] = 0.0;
  __syncthreads();
  LSTAMP(0)
  if (wv == 1) lin_minv_lane_c(L, Cc, lane);                 // lanes 0..24: Minv, 25..36: G and C
  else if (lane < 2) lin_contact_rhs(L, Cc, P.dyn.g, lane);
  __syncthreads();
  LSTAMP(1)
  // wave 0: the constraint solve; wave 1, idle here otherwise: the integrator's uniform quantities except what depends on the
  // constrained accelerations -- lin_prologue reads qacc, so it follows the correction below, but on wave 1 (one lane, 7 k
  // cycles) beside wave 0's force accumulation instead of after it
  if (wv == 0) lin_contact_solve_w(Cc, P.stance + b * P.stance_stride + 2 * t, P.dyn.soft, P.dyn.contact, lane);
  __syncthreads();
  LSTAMP(2)
  if (wv == 0) lin_contact_correct(L, Cc, lane);
  __syncthreads();
  if (wv == 0) lin_accumulate_forces_w(L, lane);
  else if (lane == 32) lin_prologue(L);
  __syncthreads();
  LSTAMP(3)
  if (wv == 0) lin_tangent_legs_c(L, Cc, lane); else lin_tangent_arms(L, lane);
  __syncthreads();
  if (wv == 0) lin_tangent_pelvis(L, lane);
  __syncthreads();
  LSTAMP(4)
  lin_contact_multipliers(L, Cc, wv, lane);
  __syncthreads();
  LSTAMP(5)
  lin_apply_minv_2c(L, Cc, tid);
  __syncthreads();
  LSTAMP(6)
  double* Ag = S.A + knot * H1_NX * H1_NX;
  double* Bg = S.Bm + knot * H1_NX * H1_NU;
  if (wv == 0 && lane < H1_NX) lin_column(L, 0, lane, [&](int r, double v) { Ag[r * H1_NX + lane] = v; });
  if (wv == 1 && lane < H1_NU) lin_column(L, 1, lane, [&](int r, double v) { Bg[r * H1_NU + lane] = v; }, Cc.G, Cc.WU);
  LSTAMP(7)
}
#endif

// Round 4: the stance-constrained tangent kernel with two knots per four-wave workgroup (h1_linearize_contact_dev.h "two knots per
// four-wave workgroup"): the leg and arm sweeps of both knots on one wave each (64 lanes), the Minv columns of both knots on wave
// 2 and the twelve unit-wrench columns of both on wave 3, one constraint solve per wave, the three base-linear-velocity
// directions as kinematics-only lanes.
// FRIC: 0 contact modes 1 / 2; 1 mode 3 (Coulomb limit: a foot outside the cone slides), 2 mode 4 (kinetic friction on the sliding
// foot) -- instantiations of their own (LinSlide, the cone check and the sliding branch of the multiplier tangents), modes 1 / 2 keep
// their machine code.
template <int FRIC> struct LinSlideOpt { LinSlide z[2]; DEVFN LinSlide* get() { return z; } };
template <> struct LinSlideOpt<0> { DEVFN LinSlide* get() { return nullptr; } };
template <bool PACK, int FRIC = 0, bool LIM = false>
__global__ void __launch_bounds__(256, 2) k_lin_tangent2c(DevState S, ProblemDev P, int mode, const int* list, const int* count) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int ks = tid >> 7, tid7 = tid & 127;
  const unsigned N = (unsigned)S.N;
  const unsigned total = (unsigned)(list ? *count : S.B) * N;
  const unsigned it0 = 2u * blockIdx.x;
  if (it0 >= total) return;
  __shared__ LinShared L2[2];
  __shared__ LinContact C2[2];
  __shared__ LinSlideOpt<FRIC> ZO;
  LinSlide* Z2 = ZO.get();
  __shared__ LinLockOpt<LIM> LKO;
  double (*lockc2)[H1_NB] = LKO.get();
#ifdef LIN_STAMP
  long long qlast = clock64();
  const int t = (int)(it0 % N), b = (int)(it0 / N);
#endif
  size_t knot[2]; bool valid[2]; const int* stance[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const unsigned it = it0 + k < total ? it0 + k : it0;
    const unsigned bs = it / N, tt = it - bs * N;
    const int bb = list ? list[bs] : (int)bs;
    knot[k] = (size_t)bb * N + tt;
    stance[k] = P.stance + bb * P.stance_stride + 2 * tt;
    bool ok = (it0 + k < total);
    if (!list && mode != MASK_ALL) ok = ok && S.active[bb] != 0 && (mode != MASK_RETRY || S.need_retry[bb] != 0);
    valid[k] = ok;
  }
  if (!valid[0] && !valid[1]) return;
  {
    const size_t kn = knot[ks];
    const size_t bb = kn / N, tt = kn - bb * N;
    lin_load_dump2c(L2[ks], C2[ks], S.lin_dump + kn * LinDumpG_SIZE, tid7, S.xbar + (bb * (N + 1) + tt) * H1_NX, S.ubar + kn * H1_NU, nullptr, nullptr);
    if (tid7 == 127) L2[ks].h = P.dyn.h;
    for (int e = tid7; e < H1_NV * LIN_LD; e += 128) (&L2[ks].dT[0][0])[e] = 0.0;
  }
  __syncthreads();
  LSTAMP(0)
  if constexpr (LIM) lin_lock_flags<LIM>(L2[ks], lockc2[ks], tid7, P.dyn.h, P.dyn.lim_k);
  if (wv == 2) { const int c = lane & 31; if (c < H1_NV) lin_minv_lane_c(L2[lane >> 5], C2[lane >> 5], c); }          // Minv columns of both knots
  else if (wv == 3) { const int c = lane & 31; if (c < 12) lin_minv_lane_c(L2[lane >> 5], C2[lane >> 5], H1_NV + c); }   // unit-wrench columns (G, C) of both
  else if (wv == 0 && (lane & 31) < 2) lin_contact_rhs(L2[lane >> 5], C2[lane >> 5], P.dyn.g, lane & 31, FRIC ? Z2 + (lane >> 5) : nullptr);
  __syncthreads();
  LSTAMP(1)
  if (wv < 2) {                                                                              // one constraint solve per wave
    if constexpr (FRIC != 0) lin_contact_solve_fr<FRIC == 2>(C2[wv], Z2[wv], stance[wv], P.dyn.soft, P.dyn.mu, lane);
    else lin_contact_solve_w(C2[wv], stance[wv], P.dyn.soft, P.dyn.contact, lane);
  }
  __syncthreads();
  LSTAMP(2)
  // Mode 4 (FRIC = 2): the friction direction of a sliding foot follows the STICKING solution, whose tangent needs the sweeps about that
  // solution (inverse-dynamics and constraint-row tangents are affine in the multipliers and the accelerations they cause): on a
  // workgroup with a sliding foot the sweeps run twice -- first about the sticking solution, leaving the force parts of dlambda_s in
  // the knot's dump record (consumed by now), then about the final one.
  if constexpr (FRIC == 2) {
  const bool two_pass = Z2[0].sl[0] || Z2[0].sl[1] || Z2[1].sl[0] || Z2[1].sl[1];
  for (int pass = two_pass ? 0 : 1; pass < 2; ++pass) {
    const bool sticking = pass == 0;
    if (wv < 2) {
      {
        const bool slides = Z2[wv].sl[0] || Z2[wv].sl[1];
        // (a knot without a sliding foot beside one with: corrected in the first pass, nothing to add in the second)
        if (sticking) lin_contact_correct<true>(L2[wv], C2[wv], lane, slides ? Z2[wv].ls : C2[wv].lam, nullptr, slides ? Z2[wv].ls : C2[wv].lam);
        else if (two_pass) lin_contact_correct<true>(L2[wv], C2[wv], lane, C2[wv].lam, slides ? Z2[wv].ls : C2[wv].lam, C2[wv].lam);
        else lin_contact_correct(L2[wv], C2[wv], lane);
      }
    }
    __syncthreads();
    if (wv == 0) lin2_accumulate_forces_w(L2, lane);
    else if (wv == 3 && (lane & 31) == 0 && !sticking) lin_prologue(L2[lane >> 5]);
    __syncthreads();
    LSTAMP(3)
    if (wv == 0) lin2_tangent_legs_c<LIM>(L2, C2, lane, Z2, lockc2);
    else if (wv == 1) lin2_tangent_arms<LIM>(L2, lane, lockc2);
    else if (wv == 2) lin2_leg_vlin_dR(L2, C2, lane);
    __syncthreads();
    if (wv < 2) lin2_tangent_pelvis(L2[wv], lane);
    __syncthreads();
    LSTAMP(4)
    double* stash = S.lin_dump + knot[wv < 2 ? wv : (lane >> 5)] * LinDumpG_SIZE;
    if (wv < 2) lin_contact_multipliers<true, FRIC>(L2[wv], C2[wv], 0, lane, FRIC ? Z2 + wv : nullptr, P.dyn.mu, stash, sticking);      // W = G^T dT, then dlambda per direction
    else if (wv == 2 && !sticking) { const int c = lane & 31; lin_contact_multipliers<true, FRIC>(L2[lane >> 5], C2[lane >> 5], 1, c < H1_NU ? c : 63, FRIC ? Z2 + (lane >> 5) : nullptr, P.dyn.mu); }   // dlambda per control column
    __syncthreads();
  }
  } else {
    if (wv < 2) lin_contact_correct(L2[wv], C2[wv], lane);
    __syncthreads();
    if (wv == 0) lin2_accumulate_forces_w(L2, lane);
    else if (wv == 3 && (lane & 31) == 0) lin_prologue(L2[lane >> 5]);
    __syncthreads();
    LSTAMP(3)
    if (wv == 0) lin2_tangent_legs_c<LIM>(L2, C2, lane, Z2, lockc2);
    else if (wv == 1) lin2_tangent_arms<LIM>(L2, lane, lockc2);
    else if (wv == 2) lin2_leg_vlin_dR(L2, C2, lane);
    __syncthreads();
    if (wv < 2) lin2_tangent_pelvis(L2[wv], lane);
    __syncthreads();
    LSTAMP(4)
    if (wv < 2) lin_contact_multipliers<true, FRIC>(L2[wv], C2[wv], 0, lane, FRIC ? Z2 + wv : nullptr, P.dyn.mu);      // W = G^T dT, then dlambda per direction
    else if (wv == 2) { const int c = lane & 31; lin_contact_multipliers<true, FRIC>(L2[lane >> 5], C2[lane >> 5], 1, c < H1_NU ? c : 63, FRIC ? Z2 + (lane >> 5) : nullptr, P.dyn.mu); }   // dlambda per control column
    __syncthreads();
  }
  LSTAMP(5)
  lin_apply_minv_2c(L2[ks], C2[ks], tid7);
  __syncthreads();
  if constexpr (LIM) { lin_lock_rows<LIM>(L2[ks], lockc2[ks], tid7); __syncthreads(); }
  LSTAMP(6)
  if (wv < 2) {
    if (lane < H1_NX && valid[wv]) lin2_store_A<PACK>(L2[wv], lane, S.A + knot[wv] * H1_NX * H1_NX, [&](auto&& out) { lin_column(L2[wv], 0, lane, out); });
  } else if (wv == 2) {
    const int k = lane >> 5, c = lane & 31;
    if (c < H1_NU && valid[k]) lin2_store_B<PACK>(L2[k], c, S.A + knot[k] * H1_NX * H1_NX, S.Bm + knot[k] * H1_NX * H1_NU, [&](auto&& out) { lin_column(L2[k], 1, c, out, C2[k].G, C2[k].WU); });
  }
  LSTAMP(7)
}

// Reference-style forward differences (RobotUtils::linearizeDynamicsFD, robot_utils.cpp:120-160):
// thread per (rollout, knot, column); columns 0..50 = d/dx, 51..69 = d/du.
// The unperturbed step f(x_t, u_t) is evaluated once per knot (k_fd_base, thread per knot, into the first 51 slots of the
// knot's lin_dump record, unused in this mode) -- as the reference does (robot_utils.cpp:126) -- not once per column.
#ifdef ILQR_LEGACY_KERNELS
__global__ void __launch_bounds__(64) k_fd_base(DevState S, ProblemDev P, int mode) {
  const long item = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= (long)S.B * S.N) return;
  const int t = (int)(item % S.N);
  const int b = (int)(item / S.N);
  if (!selected(S, b, mode)) return;
  const double* xg = S.xbar + ((size_t)b * (S.N + 1) + t) * H1_NX;
  const double* ug = S.ubar + ((size_t)b * S.N + t) * H1_NU;
  double x[H1_NX], u[H1_NU], base[H1_NX];
  for (int i = 0; i < H1_NX; ++i) x[i] = xg[i];
  for (int i = 0; i < H1_NU; ++i) u[i] = ug[i];
  step<double>(x, u, P.dyn, base, P.stance + b * P.stance_stride + 2 * t);
  double* out = S.lin_dump + (size_t)item * LinDumpG_SIZE;
  for (int i = 0; i < H1_NX; ++i) out[i] = base[i];
}
#endif
#ifdef ILQR_LEGACY_KERNELS
__global__ void __launch_bounds__(256) k_linearize_fd(DevState S, ProblemDev P, int mode, double eps) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int NC = H1_NX + H1_NU;
  const long total = (long)S.B * S.N * NC;
  if (gid >= total) return;
  const int col = (int)(gid % NC);
  const long item = gid / NC;
  const int t = (int)(item % S.N);
  const int b = (int)(item / S.N);
  if (!selected(S, b, mode)) return;
  const double* xg = S.xbar + ((size_t)b * (S.N + 1) + t) * H1_NX;
  const double* ug = S.ubar + ((size_t)b * S.N + t) * H1_NU;
  const double* base = S.lin_dump + (size_t)item * LinDumpG_SIZE;
  double* Ag = S.A + ((size_t)b * S.N + t) * H1_NX * H1_NX;
  double* Bg = S.Bm + ((size_t)b * S.N + t) * H1_NX * H1_NU;
  double x[H1_NX], u[H1_NU], pert[H1_NX];
  for (int i = 0; i < H1_NX; ++i) x[i] = xg[i];
  for (int i = 0; i < H1_NU; ++i) u[i] = ug[i];
  const int* stance = P.stance + b * P.stance_stride + 2 * t;
  if (col < H1_NX) x[col] += eps; else u[col - H1_NX] += eps;
  step<double>(x, u, P.dyn, pert, stance);
  if (col < H1_NX) { for (int i = 0; i < H1_NX; ++i) Ag[i * H1_NX + col] = (pert[i] - base[i]) / eps; }
  else { const int c = col - H1_NX; for (int i = 0; i < H1_NX; ++i) Bg[i * H1_NU + c] = (pert[i] - base[i]) / eps; }
}
#endif

// K3 (cost quadratics) lives in quad_kernels.hip

// ------------------------------------------------------------------ K4: Riccati backward pass
// one 256-thread workgroup per rollout; Vxx, A_t, W, B_t, G, Qxu, K_t staged in LDS.
#define LDN 52   // padded leading dimension of 51-wide LDS matrices
#define LDM 20   // padded leading dimension of 19-wide LDS matrices
#ifdef ILQR_LEGACY_KERNELS
__global__ void __launch_bounds__(256) k_backward(DevState S, int mode) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (!selected(S, b, mode)) return;
  const int N = S.N, n = H1_NX, m = H1_NU;
  extern __shared__ double sm[];
  double* Vxx = sm;                 // n x LDN  (also holds Qxx)
  double* At = Vxx + n * LDN;       // n x LDN
  double* W = At + n * LDN;         // n x LDN  (also T1)
  double* Bt = W + n * LDN;         // n x LDM
  double* G = Bt + n * LDM;         // n x LDM
  double* Qxu = G + n * LDM;        // n x LDM
  double* Kt = Qxu + n * LDM;       // m x LDN
  double* Mq = Kt + m * LDN;        // m x LDN  (Quu K)
  double* Quu = Mq + m * LDN;       // m x LDM
  double* Lc = Quu + m * LDM;       // m x LDM  (Cholesky factor or inverse)
  double* Vx = Lc + m * LDM;        // n
  double* Qx = Vx + n;              // n
  double* Qu = Qx + n;              // m (+pad)
  double* kt = Qu + LDM;            // m (+pad)
  double* Quuk = kt + LDM;          // m (+pad)
  __shared__ int chol_fail, use_inv;

  const double lam = S.lambda[b];
  const double* lxg = S.lx + (size_t)b * (N + 1) * n;
  const double* lxxg = S.lxx + (size_t)b * (N + 1) * n * n;
  for (int e = tid; e < n * n; e += 256) Vxx[(e / n) * LDN + (e % n)] = lxxg[(size_t)N * n * n + e];
  if (tid < n) Vx[tid] = lxg[N * n + tid];
  __syncthreads();

  for (int t = N - 1; t >= 0; --t) {
    const double* Ag = S.A + ((size_t)b * N + t) * n * n;
    const double* Bg = S.Bm + ((size_t)b * N + t) * n * m;
    for (int e = tid; e < n * n; e += 256) At[(e / n) * LDN + (e % n)] = Ag[e];
    for (int e = tid; e < n * m; e += 256) Bt[(e / m) * LDM + (e % m)] = Bg[e];
    __syncthreads();
    // W = Vxx A ; G = Vxx B ; Qx = lx + A^T Vx ; Qu = lu + B^T Vx
    for (int e = tid; e < n * n; e += 256) {
      const int i = e / n, j = e % n; double s = 0.0;
      for (int k = 0; k < n; ++k) s += Vxx[i * LDN + k] * At[k * LDN + j];
      W[i * LDN + j] = s;
    }
    for (int e = tid; e < n * m; e += 256) {
      const int i = e / m, j = e % m; double s = 0.0;
      for (int k = 0; k < n; ++k) s += Vxx[i * LDN + k] * Bt[k * LDM + j];
      G[i * LDM + j] = s;
    }
    if (tid < n) { double s = 0.0; for (int k = 0; k < n; ++k) s += At[k * LDN + tid] * Vx[k]; Qx[tid] = lxg[t * n + tid] + s; }
    else if (tid >= 64 && tid < 64 + m) { const int i = tid - 64; double s = 0.0; for (int k = 0; k < n; ++k) s += Bt[k * LDM + i] * Vx[k]; Qu[i] = S.lu[((size_t)b * N + t) * m + i] + s; }
    __syncthreads();
    // Qxx = lxx + A^T W (into Vxx) ; Quu = luu + B^T G + lam I ; Qxu = A^T G
    for (int e = tid; e < n * n; e += 256) {
      const int i = e / n, j = e % n; double s = 0.0;
      for (int k = 0; k < n; ++k) s += At[k * LDN + i] * W[k * LDN + j];
      Vxx[i * LDN + j] = lxxg[(size_t)t * n * n + e] + s;
    }
    for (int e = tid; e < n * m; e += 256) {
      const int i = e / m, j = e % m; double s = 0.0;
      for (int k = 0; k < n; ++k) s += At[k * LDN + i] * G[k * LDM + j];
      Qxu[i * LDM + j] = s;
    }
    for (int e = tid; e < m * m; e += 256) {
      const int i = e / m, j = e % m; double s = 0.0;
      for (int k = 0; k < n; ++k) s += Bt[k * LDM + i] * G[k * LDM + j];
      if (i == j) s += S.luu[((size_t)b * N + t) * m + i] + lam;
      Quu[i * LDM + j] = s;
    }
    __syncthreads();
    // Cholesky (LLT check, +1e-4 I once, ilqr.cpp:278-281); single thread keeps the branchy part simple
    if (tid == 0) {
      use_inv = 0;
      for (int attempt = 0; attempt < 2; ++attempt) {
        int fail = 0;
        for (int j = 0; j < m && !fail; ++j) {
          double s = Quu[j * LDM + j];
          for (int k = 0; k < j; ++k) s -= Lc[j * LDM + k] * Lc[j * LDM + k];
          if (!(s > 0.0)) { fail = 1; break; }
          const double d = sqrt(s); Lc[j * LDM + j] = d;
          for (int i = j + 1; i < m; ++i) { double v = Quu[i * LDM + j]; for (int k = 0; k < j; ++k) v -= Lc[i * LDM + k] * Lc[j * LDM + k]; Lc[i * LDM + j] = v / d; }
        }
        chol_fail = fail;
        if (!fail) break;
        if (attempt == 0) for (int i = 0; i < m; ++i) Quu[i * LDM + i] += 1e-4;
      }
      if (chol_fail) {
        // indefinite Quu: the reference's pivoted LDLT still solves; use Gauss-Jordan inverse with partial pivoting
        double Mx[H1_NU][2 * H1_NU];
        for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { Mx[i][j] = Quu[i * LDM + j]; Mx[i][m + j] = (i == j) ? 1.0 : 0.0; }
        for (int c = 0; c < m; ++c) {
          int p = c; double best = fabs(Mx[c][c]);
          for (int r = c + 1; r < m; ++r) if (fabs(Mx[r][c]) > best) { best = fabs(Mx[r][c]); p = r; }
          if (p != c) for (int k = 0; k < 2 * m; ++k) { const double tmp = Mx[c][k]; Mx[c][k] = Mx[p][k]; Mx[p][k] = tmp; }
          const double ip = 1.0 / Mx[c][c];
          for (int k = 0; k < 2 * m; ++k) Mx[c][k] *= ip;
          for (int r = 0; r < m; ++r) if (r != c) { const double f = Mx[r][c]; for (int k = 0; k < 2 * m; ++k) Mx[r][k] -= f * Mx[c][k]; }
        }
        for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Lc[i * LDM + j] = Mx[i][m + j];
        use_inv = 1;
      }
    }
    __syncthreads();
    // K = -Quu^-1 Qxu^T (one thread per column), k = -Quu^-1 Qu
    if (tid <= n) {
      double rhs[H1_NU], y[H1_NU], xsol[H1_NU];
      if (tid < n) { for (int i = 0; i < m; ++i) rhs[i] = Qxu[tid * LDM + i]; } else { for (int i = 0; i < m; ++i) rhs[i] = Qu[i]; }
      if (!use_inv) {
        for (int i = 0; i < m; ++i) { double s = rhs[i]; for (int k = 0; k < i; ++k) s -= Lc[i * LDM + k] * y[k]; y[i] = s / Lc[i * LDM + i]; }
        for (int i = m - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < m; ++k) s -= Lc[k * LDM + i] * xsol[k]; xsol[i] = s / Lc[i * LDM + i]; }
      } else {
        for (int i = 0; i < m; ++i) { double s = 0.0; for (int k = 0; k < m; ++k) s += Lc[i * LDM + k] * rhs[k]; xsol[i] = s; }
      }
      if (tid < n) { for (int i = 0; i < m; ++i) Kt[i * LDN + tid] = -xsol[i]; } else { for (int i = 0; i < m; ++i) kt[i] = -xsol[i]; }
    }
    __syncthreads();
    // store gains; Mq = Quu K ; Quuk = Quu k
    double* Kg = S.K + ((size_t)b * N + t) * m * n;
    for (int e = tid; e < m * n; e += 256) Kg[e] = Kt[(e / n) * LDN + (e % n)];
    if (tid < m) S.kff[((size_t)b * N + t) * m + tid] = kt[tid];
    for (int e = tid; e < m * n; e += 256) {
      const int a = e / n, j = e % n; double s = 0.0;
      for (int c = 0; c < m; ++c) s += Quu[a * LDM + c] * Kt[c * LDN + j];
      Mq[a * LDN + j] = s;
    }
    if (tid < m) { double s = 0.0; for (int c = 0; c < m; ++c) s += Quu[tid * LDM + c] * kt[c]; Quuk[tid] = s; }
    __syncthreads();
    // Vx = Qx + K^T Quu k + K^T Qu + Qxu k ; T1 = Qxx + K^T Quu K + K^T Qxu^T + Qxu K (into W)
    double nvx = 0.0;
    if (tid < n) {
      double s1 = 0.0, s2 = 0.0, s3 = 0.0;
      for (int a = 0; a < m; ++a) { s1 += Kt[a * LDN + tid] * Quuk[a]; s2 += Kt[a * LDN + tid] * Qu[a]; s3 += Qxu[tid * LDM + a] * kt[a]; }
      nvx = Qx[tid] + s1 + s2 + s3;
    }
    for (int e = tid; e < n * n; e += 256) {
      const int i = e / n, j = e % n; double s1 = 0.0, s2 = 0.0, s3 = 0.0;
      for (int a = 0; a < m; ++a) { const double kai = Kt[a * LDN + i]; s1 += kai * Mq[a * LDN + j]; s2 += kai * Qxu[j * LDM + a]; s3 += Qxu[i * LDM + a] * Kt[a * LDN + j]; }
      W[i * LDN + j] = Vxx[i * LDN + j] + s1 + s2 + s3;
    }
    __syncthreads();
    if (tid < n) Vx[tid] = nvx;
    for (int e = tid; e < n * n; e += 256) { const int i = e / n, j = e % n; Vxx[i * LDN + j] = 0.5 * (W[i * LDN + j] + W[j * LDN + i]); }
    __syncthreads();
  }
  // value function at knot 0 (parity artefact)
  for (int e = tid; e < n * n; e += 256) S.Vxx[(size_t)b * n * n + e] = Vxx[(e / n) * LDN + (e % n)];
  if (tid < n) S.Vx[(size_t)b * n + tid] = Vx[tid];
}
#endif
size_t backward_lds_bytes() {
  const int n = H1_NX, m = H1_NU;
  return sizeof(double) * (size_t)(3 * n * LDN + 3 * n * LDM + 2 * m * LDN + 2 * m * LDM + 2 * n + 3 * LDM);
}

// ------------------------------------------------------------------ K5: line search, all 8 alphas at once
// thread per (rollout, alpha); candidates kept in HBM, k_control copies the accepted one.
__constant__ double ALPHAS[8] = {1.0, 0.8, 0.6, 0.4, 0.2, 0.1, 0.05, 0.01};
#ifdef ILQR_LEGACY_KERNELS
__global__ void __launch_bounds__(64) k_line_search(DevState S, ProblemDev P, int mode) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid >> 3, ai = gid & 7;
  if (b >= S.B || !selected(S, b, mode)) return;
  const int N = S.N, n = H1_NX, m = H1_NU;
  const double alpha = ALPHAS[ai];
  const double* xb = S.xbar + (size_t)b * (N + 1) * n;
  const double* ub = S.ubar + (size_t)b * N * m;
  const double* Kg = S.K + (size_t)b * N * m * n;
  const double* kg = S.kff + (size_t)b * N * m;
  double* xc = S.xcand + ((size_t)b * 8 + ai) * (N + 1) * n;
  double* uc = S.ucand + ((size_t)b * 8 + ai) * N * m;
  double x[H1_NX], xn[H1_NX], u[H1_NU], dx[H1_NX];
  for (int i = 0; i < n; ++i) { x[i] = S.x0[(size_t)b * n + i]; xc[i] = x[i]; }
  double c = 0.0;
  for (int t = 0; t < N; ++t) {
    for (int j = 0; j < n; ++j) dx[j] = x[j] - xb[t * n + j];
    for (int i = 0; i < m; ++i) {
      double s = 0.0;
      const double* Kr = Kg + ((size_t)t * m + i) * n;
      for (int j = 0; j < n; ++j) s += Kr[j] * dx[j];
      u[i] = ub[t * m + i] + alpha * kg[t * m + i] + s;
      uc[t * m + i] = u[i];
    }
    c += knot_cost(P, b, t, x, u);
    step<double>(x, u, P.dyn, xn, P.stance + b * P.stance_stride + 2 * t);
    for (int i = 0; i < n; ++i) { x[i] = xn[i]; xc[(t + 1) * n + i] = xn[i]; }
  }
  c += knot_cost(P, b, N, x, nullptr);
  S.cand_cost[(size_t)b * 8 + ai] = c;
}
#endif

// ------------------------------------------------------------------ K6: iteration control
// one wave per rollout, 16 rollouts per workgroup: lane 0 decides (ilqr.cpp:619-655), all lanes copy the accepted candidate.
// phase 0: after the first line search of an iteration; phase 1: after the retry line search;
// phase 2: stage API (report only, accept if improved, no lambda / activity bookkeeping).
// dst[0..len) = src[0..len) by the 64 lanes of a wave, eight loads in flight per lane (a plain copy loop waits for every load
// before it issues the next one: source and destination might alias as far as the compiler knows)
__device__ __forceinline__ void wave_copy(double* dst, const double* src, int len, int lane) {
  for (int base = 0; base < len; base += 64 * 8) {
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int e = base + 64 * j + lane; v[j] = src[e < len ? e : 0]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int e = base + 64 * j + lane; if (e < len) dst[e] = v[j]; }
  }
}
#define CTRL_WAVES 16
__global__ void __launch_bounds__(64 * CTRL_WAVES) k_control(DevState S, int phase, int iter, double tol, int ee_flags, int sum_knots, const int* gate) {
  if (gate && *gate == 0) return;      // (device-side choice between two enqueued launch orders: launch_spec_gate)
  // bit 0: the reference's convergence exit; bit 1 (phase 0 only, ilqr_hip_set_dedup_saturated_retry): a lambda retry whose lambda is
  // already saturated -- min(10 lambda, 1e-3) == lambda -- would repeat the pass that has just failed bit for bit (same Jacobians,
  // quadratics and lambda give the same gains, the same eight candidates and the same costs): its bookkeeping (ilqr.cpp:640-655, the
  // failing branch) is played now and the rollout does not join the retry pass
  const int early_exit = ee_flags & 1, dedup = (ee_flags >> 1) & 1;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * CTRL_WAVES + wv;
  __shared__ int s_accept[CTRL_WAVES], s_slot[CTRL_WAVES], s_base[3];
  const int N = S.N;
  bool run = b < S.B;
  if (run && phase == 0 && !S.active[b]) run = false;
  if (run && phase == 1 && !(S.active[b] && S.need_retry[b])) run = false;
  if (lane == 0) { s_accept[wv] = -1; s_slot[wv] = -1; }
  // sum_knots: the candidates' costs arrive as per-knot costs (k_traj_knot_cost); lane a adds candidate a's in knot order -- the
  // order of a sequential accumulation along the rollout, as k_traj_cost_sum does (one launch and one launch gap less per pass)
  double csum = 0.0;
  if (run && sum_knots && lane < 8) {
    const double* ck = S.cand_knot + ((size_t)b * 8 + lane) * (N + 1);
    for (int t = 0; t <= N; ++t) csum += ck[t];
    S.cand_cost[(size_t)b * 8 + lane] = csum;
  }
  double cc[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const int lo = __shfl(__double2loint(csum), a), hi = __shfl(__double2hiint(csum), a);
    cc[a] = __hiloint2double(hi, lo);
  }
  if (run && lane == 0) {
    if (phase == 0) S.iters[b] += 1;
    const double base = S.Jbase[b];
    int acc = -1;
    if (!sum_knots) {
#pragma unroll
      for (int a = 0; a < 8; ++a) cc[a] = S.cand_cost[(size_t)b * 8 + a];     // (one batch; inside the scan each load would wait for the previous test)
    }
#pragma unroll
    for (int a = 7; a >= 0; --a) if (cc[a] < base - 1e-6) acc = a;            // the first (largest) alpha that improves
    s_accept[wv] = acc;
    S.improved[b] = acc >= 0;
    S.alpha_idx[b] = acc;
    if (phase == 2) {
      S.ls_cost[b] = acc >= 0 ? cc[acc < 0 ? 0 : acc] : base;
    } else {
      const double lam_used = S.lambda[b];
      const int tr = iter;  // trace slot
      if (acc >= 0) {
        const double Jprev = S.J[b];
        const double Jn = cc[acc];
        S.J[b] = Jn;
        S.Jbase[b] = Jn;   // the accepted candidate is the next nominal trajectory: its cost is the next baseline
        S.lambda[b] = fmax(lam_used / 2.0, 1e-6);
        S.need_retry[b] = 0;
        S.trace_cost[(size_t)b * (S.max_iter + 1) + tr + 1] = Jn;
        S.trace_alpha[(size_t)b * S.max_iter + tr] = ALPHAS[acc];
        S.trace_lambda[(size_t)b * S.max_iter + tr] = lam_used;
        if (early_exit && (fabs(Jn - Jprev) < tol || Jn > 1e6)) S.active[b] = 0;
      } else if (phase == 0 && !(dedup && fmin(lam_used * 10.0, 1e-3) == lam_used)) {
        S.lambda[b] = fmin(lam_used * 10.0, 1e-3);
        S.need_retry[b] = 1;
      } else {
        S.need_retry[b] = 0;
        S.trace_cost[(size_t)b * (S.max_iter + 1) + tr + 1] = S.J[b];
        S.trace_alpha[(size_t)b * S.max_iter + tr] = 0.0;
        S.trace_lambda[(size_t)b * S.max_iter + tr] = lam_used;
        if (early_exit && iter > 1) S.active[b] = 0;
      }
      // compacted work lists (DevState::order): this rollout goes into the retry pass of this iteration (kind 1) or, its
      // iteration being over, among the rollouts still active in the next one (kind 0)
      if (S.order) {
        const int retry = phase == 0 && S.need_retry[b];
        if (retry || S.active[b]) s_slot[wv] = retry ? 1 : 0;
      }
      // early continuation: who starts iteration iter + 1 now (group A, decided by the first search) and who after the retry (R)
      if (S.grp_a) {
        if (phase == 0) { S.grp_a[b] = (!S.need_retry[b] && S.active[b]) ? 1 : 0; S.grp_r[b] = 0; }
        else S.grp_r[b] = S.active[b] ? 1 : 0;
      }
    }
  }
  __syncthreads();
  if (S.order && phase != 2) {
    // one atomic per list and workgroup (16 rollouts), positions within the workgroup in rollout order
    if (threadIdx.x < 2) {
      int cnt = 0;
      for (int w = 0; w < CTRL_WAVES; ++w) cnt += s_slot[w] == (int)threadIdx.x;
      const int slot = threadIdx.x ? 2 * iter + 1 : 2 * (iter + 1);
      s_base[threadIdx.x] = cnt ? atomicAdd(&S.order_n[slot], cnt) : 0;
      if (threadIdx.x == 0 && phase == 1 && S.order_r) s_base[2] = cnt ? atomicAdd(&S.order_rn[iter + 1], cnt) : 0;
    }
    __syncthreads();
    if (lane == 0 && s_slot[wv] >= 0) {
      const int kind = s_slot[wv];
      int pos = s_base[kind], posr = s_base[2];
      for (int w = 0; w < wv; ++w) { pos += s_slot[w] == kind; posr += s_slot[w] == kind; }
      const int slot = kind ? 2 * iter + 1 : 2 * (iter + 1);
      S.order[(size_t)slot * S.B + pos] = b;
      if (kind == 0 && phase == 1 && S.order_r) S.order_r[posr] = b;
    }
  }
  const int acc = s_accept[wv];
  if (acc < 0) return;
  const double* xc = S.xcand + ((size_t)b * 8 + acc) * (N + 1) * H1_NX;
  const double* uc = S.ucand + ((size_t)b * 8 + acc) * N * H1_NU;
  double* xb = S.xbar + (size_t)b * (N + 1) * H1_NX;
  double* ub = S.ubar + (size_t)b * N * H1_NU;
  wave_copy(xb, xc, (N + 1) * H1_NX, lane);
  wave_copy(ub, uc, N * H1_NU, lane);
}

// ---- speculative lambda retry (small active sets) ---------------------------------------------------------------------------
// The retry of ilqr.cpp:619-644 -- lambda <- min(10 lambda, 1e-3), backward pass, line search -- only depends on data that exist
// before the FIRST line search of the iteration has decided anything.  While the rollouts of a pass fill at most half of the chip's
// SIMDs, both Riccati passes (lambda and 10 lambda) and both line searches run side by side on two streams, each into its own
// buffers (T = the twin view: K, kff, Vx, Vxx, candidates, lambda), and this kernel then plays ilqr.cpp:619-655 once with both
// outcomes on the table: first search accepted -> the twin is dropped; else the twin IS the retry the reference would have run
// (its gains, value function, candidates and costs replace the first pass's, accepted or not).  Same results, one pass of latency.
__global__ void k_spec_gate(const int* n_ptr, int max, int* g) { const int n = *n_ptr; g[0] = n <= max ? n : 0; g[1] = n <= max ? 0 : 1; g[2] = n <= max ? 0 : n; }
__global__ void k_spec_lambda(DevState S, double* lambda2) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < S.B) lambda2[b] = fmin(S.lambda[b] * 10.0, 1e-3);
}
__global__ void __launch_bounds__(64 * CTRL_WAVES) k_control_spec(DevState S, DevState T, int iter, double tol, int early_exit, int sum_knots, const int* gate) {
  if (gate && *gate == 0) return;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * CTRL_WAVES + wv;
  __shared__ int s_accept[CTRL_WAVES], s_retry[CTRL_WAVES], s_slot[CTRL_WAVES], s_base[1];
  const int N = S.N;
  const bool run = b < S.B && S.active[b];
  if (lane == 0) { s_accept[wv] = -1; s_retry[wv] = 0; s_slot[wv] = -1; }
  // lanes 0..7: candidates of the first search, lanes 8..15: of the twin
  double csum = 0.0;
  if (run && lane < 16) {
    const DevState& D = lane < 8 ? S : T;
    const int a = lane & 7;
    if (sum_knots) {
      const double* ck = D.cand_knot + ((size_t)b * 8 + a) * (N + 1);
      for (int t = 0; t <= N; ++t) csum += ck[t];
    } else csum = D.cand_cost[(size_t)b * 8 + a];
  }
  double cc[16];
#pragma unroll
  for (int a = 0; a < 16; ++a) {
    const int lo = __shfl(__double2loint(csum), a), hi = __shfl(__double2hiint(csum), a);
    cc[a] = __hiloint2double(hi, lo);
  }
  if (run && lane == 0) {
    S.iters[b] += 1;
    const double base = S.Jbase[b];
    int acc1 = -1, acc2 = -1;
#pragma unroll
    for (int a = 7; a >= 0; --a) { if (cc[a] < base - 1e-6) acc1 = a; if (cc[8 + a] < base - 1e-6) acc2 = a; }
    const double lam0 = S.lambda[b];
    const bool retry = acc1 < 0;
    const double lam_used = retry ? fmin(lam0 * 10.0, 1e-3) : lam0;      // (ilqr.cpp:634)
    const int acc = retry ? acc2 : acc1;
    s_accept[wv] = acc; s_retry[wv] = retry ? 1 : 0;
    S.improved[b] = acc >= 0;
    S.alpha_idx[b] = acc;
    S.need_retry[b] = 0;
#pragma unroll
    for (int a = 0; a < 8; ++a) S.cand_cost[(size_t)b * 8 + a] = retry ? cc[8 + a] : cc[a];
    const int tr = iter;
    if (acc >= 0) {
      const double Jprev = S.J[b];
      const double Jn = retry ? cc[8 + acc] : cc[acc];
      S.J[b] = Jn;
      S.Jbase[b] = Jn;
      S.lambda[b] = fmax(lam_used / 2.0, 1e-6);
      S.trace_cost[(size_t)b * (S.max_iter + 1) + tr + 1] = Jn;
      S.trace_alpha[(size_t)b * S.max_iter + tr] = ALPHAS[acc];
      S.trace_lambda[(size_t)b * S.max_iter + tr] = lam_used;
      if (early_exit && (fabs(Jn - Jprev) < tol || Jn > 1e6)) S.active[b] = 0;
    } else {
      S.lambda[b] = lam_used;
      S.trace_cost[(size_t)b * (S.max_iter + 1) + tr + 1] = S.J[b];
      S.trace_alpha[(size_t)b * S.max_iter + tr] = 0.0;
      S.trace_lambda[(size_t)b * S.max_iter + tr] = lam_used;
      if (early_exit && iter > 1) S.active[b] = 0;
    }
    if (S.order && S.active[b]) s_slot[wv] = 0;
  }
  __syncthreads();
  if (S.order) {
    if (threadIdx.x == 0) {
      int cnt = 0;
      for (int w = 0; w < CTRL_WAVES; ++w) cnt += s_slot[w] == 0;
      s_base[0] = cnt ? atomicAdd(&S.order_n[2 * (iter + 1)], cnt) : 0;
    }
    __syncthreads();
    if (lane == 0 && s_slot[wv] == 0) {
      int pos = s_base[0];
      for (int w = 0; w < wv; ++w) pos += s_slot[w] == 0;
      S.order[(size_t)(2 * (iter + 1)) * S.B + pos] = b;
    }
  }
  // The copies: the whole workgroup serves its sixteen rollouts one after the other (a rollout's retry moves 200 KB of gains: by its own
  // wave alone 46 us -- 4 % of an iteration of a single-rollout solve; sixteen waves take 5).
  __syncthreads();
  for (int w = 0; w < CTRL_WAVES; ++w) {
    const int acc = s_accept[w];
    const bool retry = s_retry[w] != 0;
    if (acc < 0 && !retry) continue;            // (rollouts that did not run have neither)
    const size_t bb = (size_t)blockIdx.x * CTRL_WAVES + w;
    auto block_copy = [&](double* dst, const double* src, int len) {
      for (int base = 0; base < len; base += 64 * CTRL_WAVES * 4) {
        double v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int e = base + 64 * CTRL_WAVES * j + (int)threadIdx.x; v[j] = src[e < len ? e : 0]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int e = base + 64 * CTRL_WAVES * j + (int)threadIdx.x; if (e < len) dst[e] = v[j]; }
      }
    };
    if (retry) {      // the last backward pass the reference would have executed is the twin's: its gains and value function stay
      block_copy(S.K + bb * N * H1_NU * H1_NX, T.K + bb * N * H1_NU * H1_NX, N * H1_NU * H1_NX);
      block_copy(S.kff + bb * N * H1_NU, T.kff + bb * N * H1_NU, N * H1_NU);
      block_copy(S.Vx + bb * H1_NX, T.Vx + bb * H1_NX, H1_NX);
      block_copy(S.Vxx + bb * H1_NX * H1_NX, T.Vxx + bb * H1_NX * H1_NX, H1_NX * H1_NX);
    }
    if (acc < 0) continue;
    const DevState& D = retry ? T : S;
    block_copy(S.xbar + bb * (N + 1) * H1_NX, D.xcand + (bb * 8 + acc) * (N + 1) * H1_NX, (N + 1) * H1_NX);
    block_copy(S.ubar + bb * N * H1_NU, D.ucand + (bb * 8 + acc) * N * H1_NU, N * H1_NU);
  }
}

// solve prologue: J = initial cost, trace[0], counters
__global__ void k_solve_begin(DevState S) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= S.B) return;
  S.active[b] = 1; S.need_retry[b] = 0; S.iters[b] = 0;
  if (S.order) {       // iteration 0: every rollout, in order; the other lists are filled by k_control
    S.order[b] = b;
    if (b == 0) { S.order_n[0] = S.B; for (int i = 1; i < 2 * (S.max_iter + 1); ++i) S.order_n[i] = 0; }
  }
  if (S.grp_a) {
    S.grp_a[b] = 0; S.grp_r[b] = 0;
    if (b == 0) for (int i = 0; i < S.max_iter + 2; ++i) { S.order_rn[i] = 0; S.order_an[i] = 0; }
  }
  const double J0 = S.Jbase[b];
  S.J[b] = J0;
  for (int i = 0; i <= S.max_iter; ++i) S.trace_cost[(size_t)b * (S.max_iter + 1) + i] = (i == 0) ? J0 : __builtin_nan("");
  for (int i = 0; i < S.max_iter; ++i) { S.trace_alpha[(size_t)b * S.max_iter + i] = __builtin_nan(""); S.trace_lambda[(size_t)b * S.max_iter + i] = __builtin_nan(""); }
}

// warm start shift (ilqr.cpp:68-80): in place on the resident solution; the last state is re-rolled by the caller
__global__ void k_warm_shift(DevState S, const double* prev_x, const double* prev_u) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int N = S.N, n = H1_NX, m = H1_NU;
  double* xb = S.xbar + (size_t)b * (N + 1) * n;
  double* ub = S.ubar + (size_t)b * N * m;
  const double* px = prev_x + (size_t)b * (N + 1) * n;
  const double* pu = prev_u + (size_t)b * N * m;
  for (int e = lane; e < n; e += blockDim.x) xb[e] = S.x0[(size_t)b * n + e];
  for (int e = lane; e < (N - 1) * n; e += blockDim.x) xb[n + e] = px[2 * n + e];
  for (int e = lane; e < (N - 1) * m; e += blockDim.x) ub[e] = pu[m + e];
  for (int e = lane; e < m; e += blockDim.x) ub[(N - 1) * m + e] = pu[(N - 1) * m + e];
}
// the nominal re-rollout of iterations >= 1 runs beside the linearisation into a shadow buffer (ilqr_capi.hip): adopt it
__global__ void k_adopt_rollout(DevState S, const double* shadow, int mode, unsigned long long* mismatches) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (!selected(S, b, mode)) return;
  const size_t len = (size_t)(S.N + 1) * H1_NX;
  double* xb = S.xbar + (size_t)b * len;
  const double* sh = shadow + (size_t)b * len;
  // The linearisation and the cost quadratics that ran beside the re-rollout used the pre-adoption trajectory; the
  // reference linearises the trajectory it has just rolled out (ilqr.cpp:563-588).  The two coincide only while the
  // re-rollout reproduces the accepted candidate bit for bit: count every element that does not (checked by the GPU tests).
  int bad = 0;
  for (int base = 0; base < (int)len; base += 64 * 8) {       // (blockDim.x == 64; eight elements in flight per lane, see wave_copy)
    double v[8], o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int e = base + 64 * j + lane, ec = e < (int)len ? e : 0; v[j] = sh[ec]; o[j] = xb[ec]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = base + 64 * j + lane;
      if (e < (int)len) { bad += (__double_as_longlong(v[j]) != __double_as_longlong(o[j])); xb[e] = v[j]; }
    }
  }
  if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}
// last knot of the warm start: xbar[N] = f(xbar[N-1], ubar[N-1])
#ifdef ILQR_LEGACY_KERNELS
__global__ void k_last_step(DevState S, ProblemDev P) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= S.B) return;
  const int N = S.N;
  double x[H1_NX], u[H1_NU], xn[H1_NX];
  for (int i = 0; i < H1_NX; ++i) x[i] = S.xbar[((size_t)b * (N + 1) + N - 1) * H1_NX + i];
  for (int i = 0; i < H1_NU; ++i) u[i] = S.ubar[((size_t)b * N + N - 1) * H1_NU + i];
  step<double>(x, u, P.dyn, xn, P.stance + b * P.stance_stride + 2 * (N - 1));
  for (int i = 0; i < H1_NX; ++i) S.xbar[((size_t)b * (N + 1) + N) * H1_NX + i] = xn[i];
}
#endif

// u = ubar[0] + K[0] (x_meas - xbar[0]); also packs the first-knot results for the per-step gather
__global__ void k_compute_control(DevState S, const double* x_meas, double* u_out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int N = S.N, n = H1_NX, m = H1_NU;
  if (lane >= m) return;
  const double* Kr = S.K + ((size_t)b * N * m + lane) * n;
  const double* xb = S.xbar + (size_t)b * (N + 1) * n;
  double s = S.ubar[(size_t)b * N * m + lane];
  for (int j = 0; j < n; ++j) s += Kr[j] * (x_meas[(size_t)b * n + j] - xb[j]);
  u_out[(size_t)b * m + lane] = s;
}
__global__ void k_pack_first_knot(DevState S, double* u0, double* K0) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int N = S.N, n = H1_NX, m = H1_NU;
  for (int e = lane; e < m; e += blockDim.x) u0[(size_t)b * m + e] = S.ubar[(size_t)b * N * m + e];
  for (int e = lane; e < m * n; e += blockDim.x) K0[(size_t)b * m * n + e] = S.K[(size_t)b * N * m * n + e];
}

// payload of the per-step gather: row b = [u0(19) | cost | K0(19 x 51) when with_gains]
__global__ void k_pack_payload(DevState S, int with_gains, double* out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int N = S.N, n = H1_NX, m = H1_NU;
  const int W = m + 1 + (with_gains ? m * n : 0);
  double* row = out + (size_t)b * W;
  for (int e = lane; e < m; e += blockDim.x) row[e] = S.ubar[(size_t)b * N * m + e];
  if (lane == 0) row[m] = S.J[b];
  if (with_gains) for (int e = lane; e < m * n; e += blockDim.x) row[m + 1 + e] = S.K[(size_t)b * N * m * n + e];
}

// Inside a solve the cost quadratics leave the strictly upper 16 x 16 tiles of lxx_t (t < N) unwritten (quad_kernels.hip: the
// one-wave Riccati kernel does not read them).  Before a consumer that reads the whole matrix from the device (the stage API's
// backward pass on another kernel family) they are mirrored back: entry (i, j), tile(i) < tile(j), <- entry (j, i).
__global__ void __launch_bounds__(256) k_mirror_lxx(DevState S) {
  const size_t knot = blockIdx.x;                  // over B * (N + 1)
  if ((int)(knot % (S.N + 1)) == S.N) return;      // the terminal knot is always written whole
  double* H = S.lxx + knot * H1_NX * H1_NX;
  for (int e = threadIdx.x; e < H1_NX * H1_NX; e += 256) {
    const int i = e / H1_NX, j = e % H1_NX;
    if ((i >> 4) < (j >> 4)) H[e] = H[j * H1_NX + i];
  }
}
void launch_mirror_lxx(const DevState& S, hipStream_t st) { hipLaunchKernelGGL(k_mirror_lxx, dim3((unsigned)((size_t)S.B * (S.N + 1))), dim3(256), 0, st, S); }

// ------------------------------------------------------------------ launchers
// defaults measured on MI355X at B = 4096 (time per full launch): rollout one lane 1.18 ms, two lanes 0.78 ms (0.4 ms
// without the in-kernel cost); line search one lane per candidate 2.35 ms, two lanes per candidate with side-owned
// feedback rows + parallel candidate costs 1.69 ms
#ifndef ROLLOUT_SPLIT_DEFAULT
#define ROLLOUT_SPLIT_DEFAULT 1
#endif
#ifndef LS_SPLIT_DEFAULT
#define LS_SPLIT_DEFAULT 1
#endif
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Kernel variants, selectable through the environment.  The switches are read ONCE when a handle is created (read_variants, called
// from ilqr_hip_create) and kept in the handle; every C-ABI call installs its handle's copy for the calling host thread (set_variants)
// and the launchers it then calls read that thread-local copy -- no getenv on the call path, and handles driven from different host
// threads (one thread and one handle per GPU, tests/cpp/cpp_multi_gpu_demo.cpp) never write to shared state.
//   ILQR_DYN=s        scratch-resident scalar ABA kernels for every dynamics stage (on-device cross-check; always in contact mode)
//   ILQR_ROLLOUT=s|r  nominal rollout on two lanes (dyn_split_kernels.hip) or one lane per rollout
//   ILQR_LS=s|r       line search on two lanes or one lane per candidate
//   ILQR_BACKWARD=wave|wave-fold|wave-generic|wg|valu  one-wave MFMA on the operand layout (riccati_pack.hip, with analytic Jacobians) /
//                     the folded one-wave kernel on the standard layout (riccati_wave.hip) / never a folded variant / four-wave MFMA
//                     (riccati_mfma.hip) / LDS + VALU cross-check
//   ILQR_LINT=1       the one-knot two-wave tangent kernel (cross-check of k_lin_tangent2)
#ifndef BACKWARD_DEFAULT
#define BACKWARD_DEFAULT 2
#endif
#define BACKWARD_DEFAULT_VALUE BACKWARD_DEFAULT
static thread_local Variants g_var = {0, ROLLOUT_SPLIT_DEFAULT, LS_SPLIT_DEFAULT, -1, 2, 0};
// The cross-check families -- scalar scratch-resident dynamics (ILQR_DYN=s), one-lane rollout / line search / primal dump
// (ILQR_ROLLOUT=r, ILQR_LS=r), the VALU and four-wave Riccati kernels (ILQR_BACKWARD=valu / wg), the folded Riccati kernel on the
// standard layout (wave-fold), the one-knot tangent kernels (ILQR_LINT=1) -- are compiled with -DILQR_LEGACY_KERNELS only: the test
// library lib/libilqr_hip_legacy.so.  The product library holds the default family alone; ilqr_hip_create refuses a handle whose
// environment selects anything else (variants_supported).
#ifdef ILQR_LEGACY_KERNELS
#define LEGACY_LAUNCH(...) __VA_ARGS__
#else
#define LEGACY_LAUNCH(...) (void)0
#endif
int variants_supported(const Variants& v) {
#ifdef ILQR_LEGACY_KERNELS
  (void)v; return 1;
#else
  const int bk = v.backward < 0 ? BACKWARD_DEFAULT_VALUE : v.backward;
  return (!v.scalar_dyn && v.rollout_split && v.ls_split && bk == 2 && v.fold != 1 && !v.lin_one_knot) ? 1 : 0;
#endif
}
static int env_split(const char* var, int dflt) { const char* e = getenv(var); return !e ? dflt : (e[0] == 's' ? 1 : 0); }
static int backward_kind() { return g_var.backward < 0 ? BACKWARD_DEFAULT : g_var.backward; }
static int use_scalar_dyn() { return g_var.scalar_dyn; }
Variants read_variants() {
  Variants v;
  const char* e = getenv("ILQR_DYN");
  v.scalar_dyn = (e && e[0] == 's') ? 1 : 0;
  v.rollout_split = env_split("ILQR_ROLLOUT", ROLLOUT_SPLIT_DEFAULT);
  v.ls_split = env_split("ILQR_LS", LS_SPLIT_DEFAULT);
  e = getenv("ILQR_BACKWARD");
  v.backward = !e ? BACKWARD_DEFAULT : (e[0] == 'v') ? 1 : (e[0] == 'w' && e[1] == 'a') ? 2 : 0;
  // fold: 0 never ("wave-generic"), 1 the folded kernel on the standard layout ("wave-fold": cross-check), 2 the operand-layout kernel
  // riccati_pack.hip (default)
  v.fold = (e && strstr(e, "generic")) ? 0 : (e && strstr(e, "fold")) ? 1 : 2;
  e = getenv("ILQR_LINT");
  v.lin_one_knot = (e && e[0] == '1') ? 1 : 0;
  return v;
}
void set_variants(const Variants& v) { g_var = v; }
int variant_ls_split() { return g_var.ls_split; }
int variant_rollout_split() { return g_var.rollout_split; }
int variant_backward() { return backward_kind(); }
int variant_scalar_dyn() { return g_var.scalar_dyn; }
int variant_lin_one_knot() { return g_var.lin_one_knot; }
int variant_pack() { return (backward_kind() == 2 && g_var.fold == 2) ? 1 : 0; }
void launch_rollout(const DevState& S, const ProblemDev& P, int mode, int do_roll, int count_iter, double* cost_out, hipStream_t st) {
  // contact mode (f4) runs on the two-lane kernels (the one-lane register kernels are constraint-free only)
  if (!use_scalar_dyn()) { if (g_var.rollout_split || constrained(P.dyn)) launch_rollout_s(S, P, mode, do_roll, count_iter, cost_out, st); else launch_rollout_r(S, P, mode, do_roll, count_iter, cost_out, st); return; }
  LEGACY_LAUNCH(hipLaunchKernelGGL(k_rollout, dim3(cdiv(S.B, 64)), dim3(64), 0, st, S, P, mode, do_roll, count_iter, cost_out));
}
void launch_step(int count, const double* x, const double* u, const DynParams& dyn, double* xn, hipStream_t st, int stance_l, int stance_r) {
  if (!use_scalar_dyn()) { if (constrained(dyn)) launch_step_s(count, x, u, dyn, xn, st, stance_l, stance_r); else launch_step_r(count, x, u, dyn, xn, st); return; }
  LEGACY_LAUNCH(hipLaunchKernelGGL(k_step, dim3(cdiv(count, 64)), dim3(64), 0, st, count, x, u, dyn, xn, stance_l, stance_r));
}
// phases: 1 = primal dump only, 2 = tangent sweeps / FD only, 3 = both
// constraint-free tangent kernel: two knots per four-wave workgroup (default) or, ILQR_LINT=1, the one-knot two-wave kernel
static void launch_lin_tangent_free(const DevState& S, const ProblemDev& P, int mode, hipStream_t st, const WorkList& w, int pack) {
  if (g_var.lin_one_knot) { LEGACY_LAUNCH(hipLaunchKernelGGL(k_lin_tangent, dim3(S.N, S.B), dim3(128), 0, st, S, P, mode, w.list, w.count)); if (pack) launch_pack_ab(S, st, mode, w.list, w.count); return; }
  const long items = (long)S.B * S.N;
  const dim3 grid2((unsigned)((items + 1) / 2));
  if (P.dyn.limits) { if (pack) hipLaunchKernelGGL((k_lin_tangent2<true, true>), grid2, dim3(256), 0, st, S, P, mode, w.list, w.count); else hipLaunchKernelGGL((k_lin_tangent2<false, true>), grid2, dim3(256), 0, st, S, P, mode, w.list, w.count); }
  else if (pack) hipLaunchKernelGGL((k_lin_tangent2<true, false>), grid2, dim3(256), 0, st, S, P, mode, w.list, w.count);
  else hipLaunchKernelGGL((k_lin_tangent2<false, false>), grid2, dim3(256), 0, st, S, P, mode, w.list, w.count);
}
void launch_linearize(const DevState& S, const ProblemDev& P, int mode, int jac_mode, double eps, hipStream_t st, int phases, int iter, int pack, const WorkList* wl) {
  const WorkList w = wl ? *wl : work_list(S, mode, iter);
  if (jac_mode == 0 && !use_scalar_dyn()) {
    // primal dump: on two lanes per knot beside the two-lane rollout kernels, one lane per knot with ILQR_ROLLOUT=r
    if (phases & 1) { if (g_var.rollout_split || constrained(P.dyn)) launch_lin_primal_s(S, P, mode, st, w.list, w.count); else launch_lin_primal_r(S, P, mode, st); }   // (contact mode: the dump is the free solve, see k_lin_tangent_c)
    if ((phases & 2) && P.dyn.contact) {
      const dim3 grid2((unsigned)(((long)S.B * S.N + 1) / 2));
      if (g_var.lin_one_knot) { LEGACY_LAUNCH(hipLaunchKernelGGL(k_lin_tangent_c, dim3(S.N, S.B), dim3(128), 0, st, S, P, mode, w.list, w.count)); if (pack) launch_pack_ab(S, st, mode, w.list, w.count); }
      else {
        // (template values: operand layout, Coulomb-limit branch of the contact mode, joint-limit rows)
#define LAUNCH_T2C(PK, FR, LM) hipLaunchKernelGGL((k_lin_tangent2c<PK, FR, LM>), grid2, dim3(256), 0, st, S, P, mode, w.list, w.count)
#define LAUNCH_T2C_L(PK, FR) do { if (P.dyn.limits) LAUNCH_T2C(PK, FR, true); else LAUNCH_T2C(PK, FR, false); } while (0)
#define LAUNCH_T2C_F(PK) do { if (P.dyn.contact == 3) LAUNCH_T2C_L(PK, 1); else if (P.dyn.contact == 4) LAUNCH_T2C_L(PK, 2); else LAUNCH_T2C_L(PK, 0); } while (0)
        if (pack) LAUNCH_T2C_F(true); else LAUNCH_T2C_F(false);
#undef LAUNCH_T2C_F
#undef LAUNCH_T2C_L
#undef LAUNCH_T2C
      }
    }
    else if (phases & 2) launch_lin_tangent_free(S, P, mode, st, w, pack);
  } else if (jac_mode == 0 && !P.dyn.contact) {               // ILQR_DYN=s: the analytic kernels are constraint-free only
    if (phases & 1) launch_lin_primal_r(S, P, mode, st);
    if (phases & 2) launch_lin_tangent_free(S, P, mode, st, w, pack);
  } else if ((phases & 2) && !use_scalar_dyn()) {
    launch_linearize_fd_s(S, P, mode, eps, st);       // forward differences on the two-lane step (any contact mode)
  } else if (phases & 2) {
    const long total = (long)S.B * S.N * (H1_NX + H1_NU);
    LEGACY_LAUNCH(hipLaunchKernelGGL(k_fd_base, dim3(cdiv((long)S.B * S.N, 64)), dim3(64), 0, st, S, P, mode));
    LEGACY_LAUNCH(hipLaunchKernelGGL(k_linearize_fd, dim3(cdiv(total, 256)), dim3(256), 0, st, S, P, mode, eps));
    (void)total;
  }
}
size_t lin_dump_doubles() { return LinDumpG_SIZE; }
// Step size h if launch_linearize(jac_mode) writes Jacobians whose hinge-position rows are exactly e_k + h * the hinge-velocity
// rows (the analytic tangent kernels, lin_column) and the backward kernel can use that (riccati_wave.hip fold_rows); else 0.
double linearize_fold_h(const ProblemDev& P, int jac_mode) {
  const bool analytic = jac_mode == 0 && (!use_scalar_dyn() || !P.dyn.contact);
  return (analytic && g_var.fold && backward_kind() == 2) ? P.dyn.h : 0.0;
}
// ILQR_BACKWARD=valu selects the LDS + VALU kernel (kept as an on-device cross-check), =wg the four-wave MFMA
// kernel (riccati_mfma.hip), =wave the one-wave-per-rollout MFMA kernel (riccati_wave.hip)
void launch_backward(const DevState& S, int mode, hipStream_t st, double fold_h, int iter) {
  const int kind = backward_kind();
  if (kind == 1) LEGACY_LAUNCH(hipLaunchKernelGGL(k_backward, dim3(S.B), dim3(256), backward_lds_bytes(), st, S, mode));
  else if (kind == 2) {
    // inside a solve (iter >= 0) the selected rollouts come from the compacted list of this pass
    const int slot = (S.order && iter >= 0 && mode != MASK_ALL) ? 2 * iter + (mode == MASK_RETRY ? 1 : 0) : -1;
    const int* list = slot >= 0 ? S.order + (size_t)slot * S.B : nullptr; const int* count = slot >= 0 ? S.order_n + slot : nullptr;
    if (g_var.fold == 2 && fold_h != 0.0) launch_backward_pack(S, mode, st, fold_h, list, count);      // (S.A, S.Bm, S.lxx in the operand layout: the caller's business)
    else launch_backward_wave(S, mode, st, g_var.fold == 1 ? fold_h : 0.0, list, count);
  }
  else LEGACY_LAUNCH(launch_backward_mfma(S, mode, st));
}
void launch_line_search(const DevState& S, const ProblemDev& P, int mode, hipStream_t st, int iter, int max_rollouts) {
  if (!use_scalar_dyn()) {
    if (g_var.ls_split || constrained(P.dyn)) {
      const int slot = (S.order && iter >= 0 && mode != MASK_ALL) ? 2 * iter + (mode == MASK_RETRY ? 1 : 0) : -1;      // as launch_backward
      launch_line_search_s(S, P, mode, st, slot >= 0 ? S.order + (size_t)slot * S.B : nullptr, slot >= 0 ? S.order_n + slot : nullptr, slot >= 0 ? max_rollouts : -1);
      launch_cand_costs(S, P, mode, st, iter < 0);      // (inside a solve k_control adds the knot costs up)
    }   // candidates' costs: all knots in parallel
    else launch_line_search_r(S, P, mode, st);                                                  // (the one-lane kernel sums its own)
    return;
  }
  LEGACY_LAUNCH(hipLaunchKernelGGL(k_line_search, dim3(cdiv((long)S.B * 8, 64)), dim3(64), 0, st, S, P, mode));
}
// inside a solve the two-lane line search leaves per-knot costs behind and k_control sums them itself (ls_costs_per_knot)
bool ls_costs_per_knot(const ProblemDev& P) { return !use_scalar_dyn() && (g_var.ls_split || constrained(P.dyn)); }
void launch_control(const DevState& S, int phase, int iter, double tol, int early_exit, hipStream_t st, int sum_knots, const int* gate) {
  hipLaunchKernelGGL(k_control, dim3(cdiv(S.B, CTRL_WAVES)), dim3(64 * CTRL_WAVES), 0, st, S, phase, iter, tol, early_exit, sum_knots, gate);
}
void launch_spec_lambda(const DevState& S, double* lambda2, hipStream_t st) { hipLaunchKernelGGL(k_spec_lambda, dim3(cdiv(S.B, 256)), dim3(256), 0, st, S, lambda2); }
void launch_control_spec(const DevState& S, const DevState& T, int iter, double tol, int early_exit, hipStream_t st, int sum_knots, const int* gate) {
  hipLaunchKernelGGL(k_control_spec, dim3(cdiv(S.B, CTRL_WAVES)), dim3(64 * CTRL_WAVES), 0, st, S, T, iter, tol, early_exit, sum_knots, gate);
}
void launch_spec_gate(const DevState& S, int iter, int max, int* g, hipStream_t st) { hipLaunchKernelGGL(k_spec_gate, dim3(1), dim3(1), 0, st, (const int*)(S.order_n + 2 * iter), max, g); }
bool spec_dual_available(const ProblemDev& P) { return !use_scalar_dyn() && backward_kind() == 2 && (g_var.ls_split || constrained(P.dyn)); }
void launch_backward_list(const DevState& S, hipStream_t st, double fold_h, const int* list, const int* count) {
  if (g_var.fold == 2 && fold_h != 0.0) launch_backward_pack(S, MASK_ACTIVE, st, fold_h, list, count);
  else launch_backward_wave(S, MASK_ACTIVE, st, g_var.fold == 1 ? fold_h : 0.0, list, count);
}
void launch_line_search_list(const DevState& S, const ProblemDev& P, hipStream_t st, const int* list, const int* count, int max_rollouts) { launch_line_search_s(S, P, MASK_ACTIVE, st, list, count, max_rollouts); }
void launch_solve_begin(const DevState& S, hipStream_t st) { hipLaunchKernelGGL(k_solve_begin, dim3(cdiv(S.B, 64)), dim3(64), 0, st, S); }
void launch_adopt_rollout(const DevState& S, const double* shadow, int mode, unsigned long long* mismatches, hipStream_t st) { hipLaunchKernelGGL(k_adopt_rollout, dim3(S.B), dim3(64), 0, st, S, shadow, mode, mismatches); }
void launch_warm_shift(const DevState& S, const double* px, const double* pu, hipStream_t st) { hipLaunchKernelGGL(k_warm_shift, dim3(S.B), dim3(64), 0, st, S, px, pu); }
void launch_last_step(const DevState& S, const ProblemDev& P, hipStream_t st) { if (!use_scalar_dyn()) { if (constrained(P.dyn)) launch_last_step_s(S, P, st); else launch_last_step_r(S, P, st); return; } LEGACY_LAUNCH(hipLaunchKernelGGL(k_last_step, dim3(cdiv(S.B, 64)), dim3(64), 0, st, S, P)); }
void launch_compute_control(const DevState& S, const double* x_meas, double* u_out, hipStream_t st) { hipLaunchKernelGGL(k_compute_control, dim3(S.B), dim3(64), 0, st, S, x_meas, u_out); }
void launch_pack_first_knot(const DevState& S, double* u0, double* K0, hipStream_t st) { hipLaunchKernelGGL(k_pack_first_knot, dim3(S.B), dim3(64), 0, st, S, u0, K0); }
void launch_pack_payload(const DevState& S, int with_gains, double* out, hipStream_t st) { hipLaunchKernelGGL(k_pack_payload, dim3(S.B), dim3(64), 0, st, S, with_gains, out); }
int backward_needs_lds_attr() {
  if (backward_mfma_set_attr() != 0) return 1;
  if (dyn_kernels_set_attr() != 0) return 1;
  if (dyn_split_kernels_set_attr() != 0) return 1;
#ifdef ILQR_LEGACY_KERNELS
  return hipFuncSetAttribute((const void*)k_backward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)backward_lds_bytes()) == hipSuccess ? 0 : 1;
#else
  return 0;
#endif
}

}  // namespace ilqr
