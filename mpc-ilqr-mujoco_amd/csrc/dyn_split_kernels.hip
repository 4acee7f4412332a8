// Dynamics kernels on TWO lanes per rollout / candidate (h1_aba_split.h): the even lane owns the left leg and arm,
// the odd lane the right ones.  A separate translation unit from dyn_kernels.hip on purpose: with both variants in one
// file the one-lane kernels came out 20 % slower (different inlining / register allocation).
//   k_rollout_s      iLQR::forwardRolloutNominal + computeTotalCost   reference src/ilqr/ilqr.cpp:119-124, 363-518
//   k_line_search_s  iLQR::forwardPassLineSearch, 8 alphas at once    reference src/ilqr/ilqr.cpp:311-361
#include <hip/hip_runtime.h>

#include "h1_cost_dev.h"
#include "h1_aba_split.h"
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

// ---- line search on two lanes per candidate (h1_aba_split.h): thread per (rollout, alpha, side), the 16 lanes of
// a rollout adjacent (lane = 16 r + 2 alpha + side).  With one lane per candidate the 8 x B candidates fill only 2
// waves per CU; two lanes each give every SIMD a wave.  Same cooperative feedback as k_line_search_r, over 16 lanes.
#define DYN_LDS_BYTES_S (h1s::LDS_SLOTS * 64 * sizeof(double))
// this lane's share of one knot of computeTotalCost (ilqr.cpp:370-443 / 447-510, penalties 512-515): its own hinges,
// plus base, torso and the base-only terms on the even lane; `com` = whole-body CoM (already summed over the pair)
// (left / right hinge indices are compile-time constants and the operands wave-uniform, so every table value is a
// scalar load or an immediate selected by the lane's side -- no per-lane indexed loads)
DEVFN double knot_cost_half(const ProblemDev& P, int b, int t, bool side, const h1s::HalfX& h, const h1s::HalfU* u, const double* com) {
  const bool term = (t == P.N);
  const double* xr = P.x_ref + b * P.x_ref_stride + t * H1_NX;
  const double* Qd = term ? P.Qf : P.Q;
  double a = 0.0, pen = 0.0;
  auto sq1 = [&](int i, double v) { const double e = v - xr[i]; a += e * Qd[i] * e; };
  // reference entries are per-rollout (one indexed load); weights and limits are wave-uniform: both sides' values are
  // loaded unconditionally and selected (`side ? p[r] : p[l]` written inline would become a branch per use)
  auto sq2 = [&](int il, int ir, double v) { const double ql = Qd[il], qr = Qd[ir]; const double e = v - xr[side ? ir : il]; a += e * (side ? qr : ql) * e; };
  auto jpen2 = [&](int jl, int jr, double q) {
    double lol, hil, lor, hir; limit_bounds(H1_JRANGE[jl], lol, hil); limit_bounds(H1_JRANGE[jr], lor, hir);
    const double lo = side ? lor : lol, hi = side ? hir : hil;
    const double vh = fmax(q - hi, 0.0), vl = fmax(lo - q, 0.0);   // branch-free: a branch per hinge would serialise the loads
    pen += P.w_joint * (vh * vh) + P.w_joint * (vl * vl);
  };
  if (!side) {
#pragma unroll
    for (int k = 0; k < 3; ++k) sq1(k, h.p[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) sq1(3 + k, h.quat[k]);
#pragma unroll
    for (int k = 0; k < 6; ++k) sq1(H1_NQ + k, h.vb[k]);
    sq1(7 + 10, h.q.th11); sq1(H1_NQ + 6 + 10, h.q.qd11); jpen2(10, 10, h.q.th11);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) { sq2(7 + k, 7 + 5 + k, h.q.thL[k]); sq2(H1_NQ + 6 + k, H1_NQ + 6 + 5 + k, h.q.qdL[k]); jpen2(k, 5 + k, h.q.thL[k]); }
#pragma unroll
  for (int k = 0; k < 4; ++k) { sq2(7 + 11 + k, 7 + 15 + k, h.q.thA[k]); sq2(H1_NQ + 6 + 11 + k, H1_NQ + 6 + 15 + k, h.q.qdA[k]); jpen2(11 + k, 15 + k, h.q.thA[k]); }
  double c = 0.5 * a;
  if (!term) {
    const double* ur = P.u_ref + b * P.u_ref_stride + t * H1_NU;
    double s = 0.0;
    auto usq2 = [&](int jl, int jr, double v) {
      const double rl = P.R[jl], rr = P.R[jr];
      const double e = v - ur[side ? jr : jl]; s += e * (side ? rr : rl) * e;
      double lol, hil, lor, hir; limit_bounds(H1_CTRLRANGE[jl], lol, hil); limit_bounds(H1_CTRLRANGE[jr], lor, hir);
      const double lo = side ? lor : lol, hi = side ? hir : hil;
      const double dh = fmax(v - hi, 0.0), dl = fmax(lo - v, 0.0);
      pen += P.w_ctrl * (dh * dh) + P.w_ctrl * (dl * dl);
    };
    if (!side) usq2(10, 10, u->u11);
#pragma unroll
    for (int k = 0; k < 5; ++k) usq2(k, 5 + k, u->uL[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) usq2(11 + k, 15 + k, u->uA[k]);
    c += 0.5 * s;
  }
  if (!side) {
    if (P.w_upright > 0.0) {
      const double qw = h.quat[0], qx = h.quat[1], qy = h.quat[2], qz = h.quat[3];
      const double zx = 2.0 * (qx * qz + qw * qy), zy = 2.0 * (qy * qz - qw * qx), zz = 1.0 - 2.0 * (qx * qx + qy * qy);
      c += 0.5 * P.w_upright * (zx * zx + zy * zy + (zz - 1.0) * (zz - 1.0));
    }
    if (P.w_balance > 0.0) {
      double ps[2];
      if (support_point(P, b, t, ps)) {
        const double om = sqrt(com[2] / 9.81);
        const double rx = com[0] + h.vb[0] * om - ps[0], ry = com[1] + h.vb[1] * om - ps[1];
        c += 0.5 * P.w_balance * (rx * rx + ry * ry);
      }
    }
  }
  return c + pen;
}
__global__ void __launch_bounds__(64) k_line_search_s(DevState S, ProblemDev P, int mode) {
  extern __shared__ double lds[];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid >> 4, ai = (gid >> 1) & 7;
  const bool side = (gid & 1) != 0;
  if (b >= S.B || !sel_s(S, b, mode)) return;
  const int lane = threadIdx.x, grp = lane & ~15, c16 = lane & 15, col = lane & ~1;
  const h1s::LaneLds L{lds, 64, lane};
  const int N = S.N, n = H1_NX, m = H1_NU;
  const double alpha = ALPHAS_S[ai];
  const double* xb = S.xbar + (size_t)b * (N + 1) * n;
  const double* ub = S.ubar + (size_t)b * N * m;
  const double* Kg = S.K + (size_t)b * N * m * n;
  const double* kg = S.kff + (size_t)b * N * m;
  double* xc = S.xcand + ((size_t)b * 8 + ai) * (N + 1) * n;
  double* uc = S.ucand + ((size_t)b * 8 + ai) * N * m;
  const bool b8 = (ai & 4) != 0, b4 = (ai & 2) != 0, b2 = (ai & 1) != 0;
  h1s::HalfX h; h1s::load_half(side, S.x0 + (size_t)b * n, h);
  h1s::store_half(side, h, xc);
#ifdef LS_STAMP
  long long ph[8] = {0}; long long tl = clock64();
#endif
  double c = 0.0;
  for (int t = 0; t < N; ++t) {
    const double* xbt = xb + t * n;
    // ---- u = ubar + alpha k + K (x - xbar)   (ilqr.cpp:332-333)
    // state deviations of this candidate -> LDS slot j of the pair's column (even lane: shared coordinates + left)
    if (!side) {
#pragma unroll
      for (int k = 0; k < 3; ++k) lds[k * 64 + col] = h.p[k] - xbt[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) lds[(3 + k) * 64 + col] = h.quat[k] - xbt[3 + k];
#pragma unroll
      for (int k = 0; k < 6; ++k) lds[(H1_NQ + k) * 64 + col] = h.vb[k] - xbt[H1_NQ + k];
      lds[(7 + 10) * 64 + col] = h.q.th11 - xbt[7 + 10]; lds[(H1_NQ + 6 + 10) * 64 + col] = h.q.qd11 - xbt[H1_NQ + 6 + 10];
    }
    const int so = side ? 5 * 64 : 0, sa = side ? 4 * 64 : 0;   // LDS slot offset of this side's leg / arm hinges
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int j = h1s::jleg(side, k);
      lds[(7 + k) * 64 + so + col] = h.q.thL[k] - xbt[7 + j];
      lds[(H1_NQ + 6 + k) * 64 + so + col] = h.q.qdL[k] - xbt[H1_NQ + 6 + j];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = h1s::jarm(side, k);
      lds[(7 + 11 + k) * 64 + sa + col] = h.q.thA[k] - xbt[7 + j];
      lds[(H1_NQ + 6 + 11 + k) * 64 + sa + col] = h.q.qdA[k] - xbt[H1_NQ + 6 + j];
    }
    __syncthreads();
    double dxs[4][8];   // dx_{16 q + c16} of the 8 candidates of this rollout
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = 16 * q + c16;
#pragma unroll
      for (int a = 0; a < 8; ++a) dxs[q][a] = (j < H1_NX) ? lds[j * 64 + grp + 2 * a] : 0.0;
    }
    __syncthreads();   // the dynamics step below reuses these LDS columns
    LSS(0)
    h1s::HalfU u;
#pragma unroll LS_UNROLL
    for (int i = 0; i < H1_NU; ++i) {
      const double* Kr = Kg + ((size_t)t * m + i) * n;
      double kv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int j = 16 * q + c16; kv[q] = (j < H1_NX) ? Kr[j] : 0.0; }
      double acc[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) acc[a] = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] += kv[q] * dxs[q][a];
      // reduce over the 16 lanes, scatter by alpha: both lanes of pair a end up with the sum of acc[a]
      double r1[4], r2[2];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const double keep = b8 ? acc[4 + q] : acc[q], send = b8 ? acc[q] : acc[4 + q]; r1[q] = keep + shfl_xor_f64(send, 8); }
#pragma unroll
      for (int q = 0; q < 2; ++q) { const double keep = b4 ? r1[2 + q] : r1[q], send = b4 ? r1[q] : r1[2 + q]; r2[q] = keep + shfl_xor_f64(send, 4); }
      const double keep = b2 ? r2[1] : r2[0], send = b2 ? r2[0] : r2[1];
      const double r3 = keep + shfl_xor_f64(send, 2);
      const double s = h1s::pair_sum(r3);
      const double ui = ub[t * m + i] + alpha * kg[t * m + i] + s;
      // hand u_i to its owner: torso -> both lanes, hinge of this lane's side -> this lane
      const int il = i - (side ? 5 : 0), ia = i - (side ? 15 : 11);
      if (i == 10) u.u11 = ui;
#pragma unroll
      for (int k = 0; k < 5; ++k) if (il == k) u.uL[k] = ui;
#pragma unroll
      for (int k = 0; k < 4; ++k) if (ia == k) u.uA[k] = ui;
      const bool mine = (i == 10) ? !side : ((i < 10) ? ((i >= 5) == side) : ((i >= 15) == side));
      if (mine) uc[t * m + i] = ui;
    }
    LSS(1)
    double com[3] = {0.0, 0.0, 0.0};
    if (P.w_balance > 0.0) h1s::com_mj(side, h, com);
    c += knot_cost_half(P, b, t, side, h, &u, com);
    LSS(2)
    h1s::step(side, h, u, P.dyn.h, P.dyn.g, L);
    LSS(3)
    h1s::store_half(side, h, xc + (t + 1) * n);
    LSS(4)
  }
  double com[3] = {0.0, 0.0, 0.0};
  if (P.w_balance > 0.0) h1s::com_mj(side, h, com);
  c += knot_cost_half(P, b, N, side, h, nullptr, com);
  c = h1s::pair_sum(c);
  if (!side) S.cand_cost[(size_t)b * 8 + ai] = c;
#ifdef LS_STAMP
  if (gid == 0) for (int q = 0; q < 8; ++q) S.J[q] = (double)ph[q];
#endif
}

// thread per (rollout, side): nominal rollout + cost; mode / do_roll as k_rollout_r
__global__ void __launch_bounds__(64) k_rollout_s(DevState S, ProblemDev P, int mode, int do_roll, int count_iter, double* cost_out) {
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
  if (do_roll) { h1s::load_half(side, S.x0 + (size_t)b * H1_NX, h); h1s::store_half(side, h, xb); }
  else h1s::load_half(side, xb, h);
  double c = 0.0;
  for (int t = 0; t < N; ++t) {
    h1s::HalfU u;
    u.u11 = ub[t * H1_NU + 10];
#pragma unroll
    for (int k = 0; k < 5; ++k) u.uL[k] = ub[t * H1_NU + h1s::jleg(side, k)];
#pragma unroll
    for (int k = 0; k < 4; ++k) u.uA[k] = ub[t * H1_NU + h1s::jarm(side, k)];
    double com[3] = {0.0, 0.0, 0.0};
    if (P.w_balance > 0.0) h1s::com_mj(side, h, com);
    c += knot_cost_half(P, b, t, side, h, &u, com);
    if (do_roll) { h1s::step(side, h, u, P.dyn.h, P.dyn.g, L); h1s::store_half(side, h, xb + (t + 1) * H1_NX); }
    else h1s::load_half(side, xb + (t + 1) * H1_NX, h);
  }
  double com[3] = {0.0, 0.0, 0.0};
  if (P.w_balance > 0.0) h1s::com_mj(side, h, com);
  c += knot_cost_half(P, b, N, side, h, nullptr, com);
  c = h1s::pair_sum(c);
  if (!side) cost_out[b] = c;
}

static inline int cdiv_s(long a, long b) { return (int)((a + b - 1) / b); }
int dyn_split_kernels_set_attr() {
  int rc = 0;
  rc |= hipFuncSetAttribute((const void*)k_line_search_s, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DYN_LDS_BYTES_S) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_rollout_s, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DYN_LDS_BYTES_S) != hipSuccess;
  return rc;
}
void launch_line_search_s(const DevState& S, const ProblemDev& P, int mode, hipStream_t st) {
  hipLaunchKernelGGL(k_line_search_s, dim3(cdiv_s((long)S.B * 16, 64)), dim3(64), DYN_LDS_BYTES_S, st, S, P, mode);
}
void launch_rollout_s(const DevState& S, const ProblemDev& P, int mode, int do_roll, int count_iter, double* cost_out, hipStream_t st) {
  hipLaunchKernelGGL(k_rollout_s, dim3(cdiv_s((long)S.B * 2, 64)), dim3(64), DYN_LDS_BYTES_S, st, S, P, mode, do_roll, count_iter, cost_out);
}

}  // namespace ilqr
