// Dynamics-bound kernels on the register/LDS-resident ABA (h1_aba_reg.h): one rollout (or one
// (rollout, alpha) pair, or one knot) per lane, 64 lanes per workgroup, 152 LDS slots per lane.
//   k_rollout_r      iLQR::forwardRolloutNominal + computeTotalCost   reference src/ilqr/ilqr.cpp:119-124, 363-518
//   k_line_search_r  iLQR::forwardPassLineSearch, 8 alphas at once    reference src/ilqr/ilqr.cpp:311-361
//   k_lin_primal_r   primal quantities of every knot for k_lin_tangent
//   k_step_r, k_last_step_r
#include <hip/hip_runtime.h>

#include "h1_cost_dev.h"
#include "h1_linearize_dev.h"
#define ABA_FENCE          // scheduling fences between the sweeps of the articulated-body algorithm (h1_aba_reg.h): k_lin_primal_r spills 532 instead of 1152 B
#include "h1_aba_reg.h"
#include "ilqr_kernels.h"

using namespace h1;

namespace ilqr {

#define DYN_LDS_BYTES (h1r::LDS_SLOTS * 64 * sizeof(double))

struct ComReg { DEVFN void operator()(const double* x, double* com) const { h1r::com_mj(x, com); } };

__device__ __forceinline__ bool sel(const DevState& S, int b, int mode) {
  if (mode == MASK_ALL) return true;
  if (mode == MASK_ACTIVE) return S.active[b] != 0;
  return S.active[b] != 0 && S.need_retry[b] != 0;
}

#ifdef ILQR_LEGACY_KERNELS
__global__ void __launch_bounds__(64) k_rollout_r(DevState S, ProblemDev P, int mode, int do_roll, int count_iter, double* cost_out) {
  extern __shared__ double lds[];
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= S.B || !sel(S, b, mode)) return;
  const h1r::LaneLds L{lds, 64, (int)threadIdx.x};
  const int N = S.N;
  double* xb = S.xbar + (size_t)b * (N + 1) * H1_NX;
  const double* ub = S.ubar + (size_t)b * N * H1_NU;
  if (count_iter) S.iters[b] += 1;
  double x[H1_NX], u[H1_NU];
  if (do_roll) {
#pragma unroll
    for (int i = 0; i < H1_NX; ++i) { x[i] = S.x0[(size_t)b * H1_NX + i]; xb[i] = x[i]; }
  } else {
#pragma unroll
    for (int i = 0; i < H1_NX; ++i) x[i] = xb[i];
  }
  double c = 0.0;
  for (int t = 0; t < N; ++t) {
#pragma unroll
    for (int i = 0; i < H1_NU; ++i) u[i] = ub[t * H1_NU + i];
    c += knot_cost_t(P, b, t, x, u, ComReg());
    if (do_roll) {
      h1r::step(x, u, P.dyn.h, P.dyn.g, L, x);
#pragma unroll
      for (int i = 0; i < H1_NX; ++i) xb[(t + 1) * H1_NX + i] = x[i];
    } else {
#pragma unroll
      for (int i = 0; i < H1_NX; ++i) x[i] = xb[(t + 1) * H1_NX + i];
    }
  }
  c += knot_cost_t(P, b, N, x, (const double*)nullptr, ComReg());
  cost_out[b] = c;
}
#endif

__global__ void __launch_bounds__(64) k_step_r(int count, const double* x, const double* u, DynParams dyn, double* xn) {
  extern __shared__ double lds[];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const h1r::LaneLds L{lds, 64, (int)threadIdx.x};
  double xl[H1_NX], ul[H1_NU], out[H1_NX];
#pragma unroll
  for (int k = 0; k < H1_NX; ++k) xl[k] = x[(size_t)i * H1_NX + k];
#pragma unroll
  for (int k = 0; k < H1_NU; ++k) ul[k] = u[(size_t)i * H1_NU + k];
  h1r::step(xl, ul, dyn.h, dyn.g, L, out);
#pragma unroll
  for (int k = 0; k < H1_NX; ++k) xn[(size_t)i * H1_NX + k] = out[k];
}

__global__ void __launch_bounds__(64) k_last_step_r(DevState S, ProblemDev P) {
  extern __shared__ double lds[];
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= S.B) return;
  const h1r::LaneLds L{lds, 64, (int)threadIdx.x};
  const int N = S.N;
  double x[H1_NX], u[H1_NU], xn[H1_NX];
#pragma unroll
  for (int i = 0; i < H1_NX; ++i) x[i] = S.xbar[((size_t)b * (N + 1) + N - 1) * H1_NX + i];
#pragma unroll
  for (int i = 0; i < H1_NU; ++i) u[i] = S.ubar[((size_t)b * N + N - 1) * H1_NU + i];
  h1r::step(x, u, P.dyn.h, P.dyn.g, L, xn);
#pragma unroll
  for (int i = 0; i < H1_NX; ++i) S.xbar[((size_t)b * (N + 1) + N) * H1_NX + i] = xn[i];
}

// thread per (rollout, alpha), the 8 alphas of a rollout in 8 adjacent lanes; candidates kept in HBM, k_control
// copies the accepted one.  The feedback K (x - xbar) is evaluated cooperatively by the 8 lanes of a rollout:
// lane a owns the columns j = a, a+8, ... of K_t (one fully used 64-byte line per row and load instruction, every
// element of K fetched once per rollout instead of once per alpha), multiplies them with the state deviations
// of all 8 candidates (exchanged through the LDS columns the dynamics step is not using at that point) and the
// 8 x 8 partial sums are reduce-scattered over the lanes with three exchange steps.
#ifndef LS_UNROLL
#define LS_UNROLL 2   // rows of K_t in flight per lane (measured: 1 -> 2.92 ms, 2 -> 2.70 ms, 19 -> 2.72 ms per launch)
#endif
// -DLS_STAMP: diagnostic build only -- per-phase cycle sums of thread 0 land in S.J[0..7]
#ifdef LS_STAMP
#define LSS(k) { __builtin_amdgcn_s_waitcnt(0); const long long tn_ = clock64(); ph[k] += tn_ - tl; tl = tn_; }
#else
#define LSS(k)
#endif
__constant__ double ALPHAS_R[8] = {1.0, 0.8, 0.6, 0.4, 0.2, 0.1, 0.05, 0.01};
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  const int lo = __shfl_xor(__double2loint(v), mask), hi = __shfl_xor(__double2hiint(v), mask);
  return __hiloint2double(hi, lo);
}
#ifdef ILQR_LEGACY_KERNELS
__global__ void __launch_bounds__(64) k_line_search_r(DevState S, ProblemDev P, int mode) {
  extern __shared__ double lds[];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid >> 3, ai = gid & 7;
  if (b >= S.B || !sel(S, b, mode)) return;
  const int lane = threadIdx.x, grp = lane & ~7;
  const h1r::LaneLds L{lds, 64, lane};
  const int N = S.N, n = H1_NX, m = H1_NU;
  const double alpha = ALPHAS_R[ai];
  const double* xb = S.xbar + (size_t)b * (N + 1) * n;
  const double* ub = S.ubar + (size_t)b * N * m;
  const double* Kg = S.K + (size_t)b * N * m * n;
  const double* kg = S.kff + (size_t)b * N * m;
  double* xc = S.xcand + ((size_t)b * 8 + ai) * (N + 1) * n;
  double* uc = S.ucand + ((size_t)b * 8 + ai) * N * m;
  const bool b4 = (ai & 4) != 0, b2 = (ai & 2) != 0, b1 = (ai & 1) != 0;
  // register budget: only x[51] and u[19] stay live across the dynamics step (no dx / x_next copies)
  double x[H1_NX], u[H1_NU];
#pragma unroll
  for (int i = 0; i < H1_NX; ++i) { x[i] = S.x0[(size_t)b * n + i]; xc[i] = x[i]; }
#ifdef LS_STAMP
  long long ph[8] = {0}; long long tl = clock64();
#endif
  double c = 0.0;
  for (int t = 0; t < N; ++t) {
    const double* xbt = xb + t * n;
    // ---- u = ubar + alpha k + K (x - xbar)   (ilqr.cpp:332-333)
    // state deviations of this candidate -> LDS slot j of this lane's column
#pragma unroll
    for (int j = 0; j < H1_NX; ++j) lds[j * 64 + lane] = x[j] - xbt[j];
    __syncthreads();
    double dxs[7][8];   // dx_{8q + ai} of the 8 candidates of this rollout
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int j = 8 * q + ai;
#pragma unroll
      for (int a = 0; a < 8; ++a) dxs[q][a] = (j < H1_NX) ? lds[j * 64 + grp + a] : 0.0;
    }
    __syncthreads();   // the dynamics step below reuses these LDS columns
    LSS(0)
#pragma unroll LS_UNROLL
    for (int i = 0; i < H1_NU; ++i) {
      const double* Kr = Kg + ((size_t)t * m + i) * n;
      double kv[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) { const int j = 8 * q + ai; kv[q] = (j < H1_NX) ? Kr[j] : 0.0; }
      double acc[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) acc[a] = 0.0;
#pragma unroll
      for (int q = 0; q < 7; ++q)
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] += kv[q] * dxs[q][a];
      // reduce-scatter: lane a ends up with sum over the 8 lanes of acc[a]
      double r1[4], r2[2];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const double keep = b4 ? acc[4 + q] : acc[q], send = b4 ? acc[q] : acc[4 + q]; r1[q] = keep + shfl_xor_f64(send, 4); }
#pragma unroll
      for (int q = 0; q < 2; ++q) { const double keep = b2 ? r1[2 + q] : r1[q], send = b2 ? r1[q] : r1[2 + q]; r2[q] = keep + shfl_xor_f64(send, 2); }
      const double keep = b1 ? r2[1] : r2[0], send = b1 ? r2[0] : r2[1];
      const double s = keep + shfl_xor_f64(send, 1);
      u[i] = ub[t * m + i] + alpha * kg[t * m + i] + s;
      uc[t * m + i] = u[i];
    }
    LSS(1)
    c += knot_cost_t(P, b, t, x, u, ComReg());
    LSS(2)
    h1r::step(x, u, P.dyn.h, P.dyn.g, L, x);       // in place: every read of x precedes the integrator's writes
    LSS(3)
#pragma unroll
    for (int i = 0; i < H1_NX; ++i) xc[(t + 1) * n + i] = x[i];
    LSS(4)
  }
  c += knot_cost_t(P, b, N, x, (const double*)nullptr, ComReg());
  S.cand_cost[(size_t)b * 8 + ai] = c;
#ifdef LS_STAMP
  if (gid == 0) for (int q = 0; q < 8; ++q) S.J[q] = (double)ph[q];
#endif
}
#endif

// computeTotalCost (ilqr.cpp:363-518) of the 8 line-search candidates of every rollout, taken off the serial rollout of
// the two-lanes-per-candidate line search (k_line_search_s, where it cost 78 k of 194 k cycles per step):
// thread per (rollout, alpha, knot) evaluates one knot cost from the stored candidate (a wave's 64 state rows are
// contiguous in HBM: fetched coalesced through LDS), a second tiny kernel adds them in knot order, i.e. in exactly the
// order a sequential accumulation along the rollout would use (the accepted step size depends on cost < J - 1e-6).
#define CK_LD 51
// rows = consecutive [N+1][51] trajectories in xsrc ([B][8] candidates: cshift = 3, oshift = 0; [B] nominal
// trajectories: cshift = 0, oshift = 3 -- they use the slots of candidate 0 of their rollout in the knot buffer)
// gate (optional): the launch does nothing when *gate == 0 (device-side choice between two enqueued launch orders, ilqr_capi.hip)
__global__ void __launch_bounds__(64) k_traj_knot_cost(DevState S, ProblemDev P, int mode, const double* xsrc, const double* usrc, int cshift, int oshift, const int* gate) {
  if (gate && *gate == 0) return;
  __shared__ double xs[32 * CK_LD];     // half a wave's rows at a time: 13 KB, so that the registers (two waves per SIMD), not the LDS, set the occupancy
  const int N = S.N, lane = threadIdx.x;
  const long total = ((long)S.B << cshift) * (N + 1);
  const long first = (long)blockIdx.x * 64;
  const long idx = first + lane;
  const long idc = idx < total ? idx : total - 1;
  const int t = (int)(idc % (N + 1));
  const long cand = idc / (N + 1);
  const int b = (int)(cand >> cshift);
  const bool act = idx < total && sel(S, b, mode);
  if (!__any(act)) return;       // wave-uniform: every rollout this wave touches is unselected
  const long nrow = (first + 64 <= total) ? 64 : (total - first);
  const double* src = xsrc + first * H1_NX;
  double x[H1_NX], u[H1_NU];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int r0 = 32 * half, nr = (int)nrow - r0 < 32 ? (int)nrow - r0 : 32;      // rows r0 .. r0 + nr of this wave
    if (half) __syncthreads();
    {
      // all the loads of the half first, then the LDS writes: as a plain copy loop the compiler waits for every load before it
      // issues the next one (26 exposed HBM round trips per half; this was most of the kernel's time)
      const int cnt = nr * H1_NX;
      constexpr int NIT = (32 * H1_NX + 63) / 64;
      double tmp[NIT];
#pragma unroll
      // (lanes past the block's rows fall back to its FIRST row, which always exists: with <= 32 rows left the second half has none)
      for (int it = 0; it < NIT; ++it) { const int e = lane + 64 * it; tmp[it] = src[e < cnt ? r0 * H1_NX + e : 0]; }
#pragma unroll
      for (int it = 0; it < NIT; ++it) { const int e = lane + 64 * it; if (e < cnt) xs[e] = tmp[it]; }
    }
    __syncthreads();
    if ((lane >> 5) == half) {
      const int row = (idx < total ? lane : (int)nrow - 1) - r0;
#pragma unroll
      for (int i = 0; i < H1_NX; ++i) x[i] = xs[(row >= 0 ? row : 0) * CK_LD + i];
    }
  }
  const double* ug = usrc + (cand * N + (t < N ? t : N - 1)) * H1_NU;
#pragma unroll
  for (int i = 0; i < H1_NU; ++i) u[i] = ug[i];
  const double c = knot_cost_t(P, b, t, x, t < N ? u : (const double*)nullptr, ComReg());
  if (act) S.cand_knot[((cand << oshift) * (N + 1)) + t] = c;
}
__global__ void __launch_bounds__(64) k_traj_cost_sum(DevState S, int mode, int cshift, int oshift, double* cost_out) {
  const int cand = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = cand >> cshift;
  if (b >= S.B || !sel(S, b, mode)) return;
  const double* ck = S.cand_knot + ((size_t)cand << oshift) * (S.N + 1);
  double c = 0.0;
  for (int t = 0; t <= S.N; ++t) c += ck[t];
  cost_out[cand] = c;
}
void launch_cand_costs(const DevState& S, const ProblemDev& P, int mode, hipStream_t st, bool with_sum, const int* gate) {
  const long total = (long)S.B * 8 * (S.N + 1);
  hipLaunchKernelGGL(k_traj_knot_cost, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, st, S, P, mode, S.xcand, S.ucand, 3, 0, gate);
  if (with_sum) hipLaunchKernelGGL(k_traj_cost_sum, dim3((unsigned)(((long)S.B * 8 + 63) / 64)), dim3(64), 0, st, S, mode, 3, 0, S.cand_cost);
}
// computeTotalCost of the nominal trajectories (S.xbar, S.ubar) into cost_out[B], same kernels, same summation order
void launch_nominal_costs(const DevState& S, const ProblemDev& P, int mode, double* cost_out, hipStream_t st) {
  const long total = (long)S.B * (S.N + 1);
  hipLaunchKernelGGL(k_traj_knot_cost, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, st, S, P, mode, S.xbar, S.ubar, 0, 3, (const int*)nullptr);
  hipLaunchKernelGGL(k_traj_cost_sum, dim3((unsigned)((S.B + 63) / 64)), dim3(64), 0, st, S, mode, 0, 3, cost_out);
}

// primal dump of one knot per lane: see LinDumpG in h1_linearize_dev.h
struct DumpSink {
  double* g;   // this knot's LinDumpG as doubles
  const h1r::LaneLds* L;      // U_i, 1 / D_i of hinge i - 1 at slots 8 (i - 1) .. + 6, left there by the inward sweep
  DEVFN void operator()(int i, const double* v, const double* a, double s, double c) const {
    double* blk = (double*)__builtin_assume_aligned(g + ldg_v(i), 16);
    double U[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) U[k] = i == 0 ? 0.0 : (*L)[8 * (i == 0 ? 0 : i - 1) + k];
#pragma unroll
    for (int k = 0; k < 6; ++k) { blk[k] = v[k]; blk[6 + k] = a[k]; blk[12 + k] = U[k]; }
    blk[18] = U[6]; blk[19] = s; blk[20] = c; blk[21] = 0.0;
  }
};
#ifdef ILQR_LEGACY_KERNELS
__global__ void __launch_bounds__(64) k_lin_primal_r(DevState S, ProblemDev P, int mode) {
  extern __shared__ double lds[];
  const long knot = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (knot >= (long)S.B * S.N) return;
  const int t = (int)(knot % S.N), b = (int)(knot / S.N);
  if (!sel(S, b, mode)) return;
  const h1r::LaneLds L{lds, 64, (int)threadIdx.x};
  const double* xg = S.xbar + ((size_t)b * (S.N + 1) + t) * H1_NX;
  const double* ug = S.ubar + ((size_t)b * S.N + t) * H1_NU;
  const double h = P.dyn.h;
  double x[H1_NX], tau[H1_NU], qacc[H1_NV];
#pragma unroll
  for (int i = 0; i < H1_NX; ++i) x[i] = xg[i];
  const double qn = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  double R0[9]; h1r::quat_R(x[3] / qn, x[4] / qn, x[5] / qn, x[6] / qn, R0);
#pragma unroll
  for (int i = 0; i < H1_NU; ++i) {
    double ui = ug[i];
    if (ui < h1c::C_CTRLRANGE[i][0]) ui = h1c::C_CTRLRANGE[i][0];
    if (ui > h1c::C_CTRLRANGE[i][1]) ui = h1c::C_CTRLRANGE[i][1];
    tau[i] = ui - H1_DAMPING * x[H1_NQ + 6 + i];
  }
  double* g = (double*)__builtin_assume_aligned(S.lin_dump + (size_t)knot * LinDumpG_SIZE, 16);
  DumpSink sink{g, &L};
  double inv36[36], aL[3];
  h1r::forward_dynamics(R0, x + 7, x + H1_NQ, tau, H1_ARMATURE + h * H1_DAMPING, P.dyn.g, L, qacc, sink, inv36, aL);
#pragma unroll
  for (int k = 0; k < 9; ++k) g[LinDumpG_R0 + k] = R0[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) g[LinDumpG_aL + k] = aL[k];
#pragma unroll
  for (int k = 0; k < H1_NV; ++k) g[LinDumpG_qacc + k] = qacc[k];
#pragma unroll
  for (int k = 0; k < 36; ++k) g[LinDumpG_IA0inv + k] = inv36[k];
}
#endif

static inline int cdiv2(long a, long b) { return (int)((a + b - 1) / b); }
int dyn_kernels_set_attr() {
  int rc = 0;
  rc |= hipFuncSetAttribute((const void*)k_step_r, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DYN_LDS_BYTES) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_last_step_r, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DYN_LDS_BYTES) != hipSuccess;
#ifdef ILQR_LEGACY_KERNELS
  rc |= hipFuncSetAttribute((const void*)k_rollout_r, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DYN_LDS_BYTES) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_line_search_r, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DYN_LDS_BYTES) != hipSuccess;
  rc |= hipFuncSetAttribute((const void*)k_lin_primal_r, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DYN_LDS_BYTES) != hipSuccess;
#endif
  return rc;
}
void launch_step_r(int count, const double* x, const double* u, const DynParams& dyn, double* xn, hipStream_t st) {
  hipLaunchKernelGGL(k_step_r, dim3(cdiv2(count, 64)), dim3(64), DYN_LDS_BYTES, st, count, x, u, dyn, xn);
}
void launch_last_step_r(const DevState& S, const ProblemDev& P, hipStream_t st) {
  hipLaunchKernelGGL(k_last_step_r, dim3(cdiv2(S.B, 64)), dim3(64), DYN_LDS_BYTES, st, S, P);
}
// the one-lane cross-check family (ILQR_ROLLOUT=r / ILQR_LS=r): compiled into the test library only (-DILQR_LEGACY_KERNELS)
#ifdef ILQR_LEGACY_KERNELS
void launch_rollout_r(const DevState& S, const ProblemDev& P, int mode, int do_roll, int count_iter, double* cost_out, hipStream_t st) {
  hipLaunchKernelGGL(k_rollout_r, dim3(cdiv2(S.B, 64)), dim3(64), DYN_LDS_BYTES, st, S, P, mode, do_roll, count_iter, cost_out);
}
void launch_line_search_r(const DevState& S, const ProblemDev& P, int mode, hipStream_t st) {
  hipLaunchKernelGGL(k_line_search_r, dim3(cdiv2((long)S.B * 8, 64)), dim3(64), DYN_LDS_BYTES, st, S, P, mode);
}
void launch_lin_primal_r(const DevState& S, const ProblemDev& P, int mode, hipStream_t st) {
  hipLaunchKernelGGL(k_lin_primal_r, dim3(cdiv2((long)S.B * S.N, 64)), dim3(64), DYN_LDS_BYTES, st, S, P, mode);
}
#else
void launch_rollout_r(const DevState&, const ProblemDev&, int, int, int, double*, hipStream_t) {}
void launch_line_search_r(const DevState&, const ProblemDev&, int, hipStream_t) {}
void launch_lin_primal_r(const DevState&, const ProblemDev&, int, hipStream_t) {}
#endif

}  // namespace ilqr
