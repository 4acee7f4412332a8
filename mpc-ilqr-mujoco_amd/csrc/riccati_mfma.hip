// K4 (MFMA): Riccati backward pass, one 256-thread workgroup (4 waves) per rollout, fp64 matrix cores.
//
// Replaces iLQR::backwardPass (reference src/ilqr/ilqr.cpp:250-309).  All 51x51 / 51x19 contractions of a
// knot run on v_mfma_f64_16x16x4_f64 (16x16 output tile, K step 4; C/D layout: lane l holds
// D[(l>>4) + 4r][l & 15], r = 0..3; A operand lane l = A[l & 15][l >> 4]; B operand lane l = B[l >> 4][l & 15]):
//   P1  W   = Vxx A          wave w -> column tile w of W           (A_t column tile held in registers)
//   P2  G   = Vxx B          wave w -> row tile w of G              (B_t held in registers)
//   P3  Qxx = lxx + A^T W    wave w -> row tile w  (A^T operand == the registers of P1)
//   P4  Qxu = A^T G          wave w -> row tile w
//   P5  Quu = luu + B^T G + lambda I   (2x2 tiles over the 4 waves)
//   --  wave 0: LLT check (+1e-4 I once), K = -Quu^-1 Qxu^T, k = -Quu^-1 Qu (Cholesky in registers,
//       row per lane, cross-lane broadcast by v_readlane; 52 right-hand sides one per lane)
//   P6  Vxx = sym(Qxx + Qxu K),  Vx = Qx + Qxu k
// The value-function update uses K^T Quu K + K^T Qxu^T = K^T (Quu K + Qxu^T) = 0 for the solved gains, i.e.
// Vxx = Qxx + Qxu K and Vx = Qx + Qxu k -- algebraically identical to the reference's long form
// (ilqr.cpp:294-307), differing only in rounding (covered by the parity tests).
// A_t and B_t are never staged in LDS: each wave streams its own column tile from HBM into registers.
// LDS operands use leading dimensions chosen so that the MFMA operand reads are bank-conflict free:
// 54 for [i][k]-pattern reads of 51-wide rows, 80 / 48 for [k][j]-pattern reads of 64 / 32 columns.
#include <hip/hip_runtime.h>

#include "h1_dynamics_dev.h"
#include "ilqr_kernels.h"

namespace ilqr {

typedef double v4d __attribute__((ext_vector_type(4)));

#define RN 51
#define RM 19
#define KS 13          // k-steps of 4 over the padded inner dimension 52
#define LDV 54         // Vxx / Qxx rows [i][k]
#define LDW 80         // W rows [k][j], 64 columns
#define LDG 48         // G rows [k][j2], 32 columns
#define LDQ 22         // Qxu rows [i][a]
#define LDK 80         // K rows [a][j]
#define LDU 20         // Quu / L rows

struct RiccatiLds {
  double Vxx[52 * LDV];   // rows 0..50 valid, row 51 and column 51.. zero
  double W[52 * LDW];     // W = Vxx A, later T1 (unsymmetrised Vxx update)
  double G[52 * LDG];
  double Qxu[64 * LDQ];
  double Kt[20 * LDK];
  double Quu[RM * LDU];
  double Lc[RM * LDU];
  double Vx[64], Qx[64], Qu[32], kt[32];
  int flags[4];
};

__device__ __forceinline__ v4d mfma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double bcast(double x, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(256) k_backward_mfma(DevState S, int mode) {
  const int b = blockIdx.x;
  if (mode == MASK_ACTIVE && !S.active[b]) return;
  if (mode == MASK_RETRY && !(S.active[b] && S.need_retry[b])) return;
  extern __shared__ double smem[];
  RiccatiLds& L = *reinterpret_cast<RiccatiLds*>(smem);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int N = S.N, n = RN, m = RM;
  const double lam = S.lambda[b];
  const double* lxg = S.lx + (size_t)b * (N + 1) * n;
  const double* lxxg = S.lxx + (size_t)b * (N + 1) * n * n;

  // zero every LDS buffer once (pads must be finite zeros), then load the terminal value function
  for (int e = tid; e < (int)(sizeof(RiccatiLds) / sizeof(double)); e += 256) smem[e] = 0.0;
  __syncthreads();
  for (int e = tid; e < n * n; e += 256) L.Vxx[(e / n) * LDV + (e % n)] = lxxg[(size_t)N * n * n + e];
  if (tid < n) L.Vx[tid] = lxg[N * n + tid];
  __syncthreads();

  for (int t = N - 1; t >= 0; --t) {
    const double* Ag = S.A + ((size_t)b * N + t) * n * n;
    const double* Bg = S.Bm + ((size_t)b * N + t) * n * m;
    // ---- operands streamed from HBM into registers
    double areg[KS];          // A[4s + lk][16w + lr]
    double breg[2][KS];       // B[4s + lk][16j2 + lr]
    {
      const int col = 16 * w + lr;
#pragma unroll
      for (int s = 0; s < KS; ++s) { const int k = 4 * s + lk; areg[s] = (k < n && col < n) ? Ag[k * n + col] : 0.0; }
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        const int c2 = 16 * j2 + lr;
#pragma unroll
        for (int s = 0; s < KS; ++s) { const int k = 4 * s + lk; breg[j2][s] = (k < n && c2 < m) ? Bg[k * m + c2] : 0.0; }
      }
    }
    // ---- P1: W[:, tile w] = Vxx A[:, tile w]
    {
      v4d acc[4];
#pragma unroll
      for (int I = 0; I < 4; ++I) acc[I] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int I = 0; I < 4; ++I) {
          int row = 16 * I + lr; row = row > 51 ? 51 : row;
          acc[I] = mfma(L.Vxx[row * LDV + 4 * s + lk], areg[s], acc[I]);
        }
      }
#pragma unroll
      for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * I + lk + 4 * r; if (row < 52) L.W[row * LDW + 16 * w + lr] = acc[I][r]; }
    }
    // ---- P2: G[tile w, :] = Vxx[tile w, :] B
    {
      v4d acc[2] = {(v4d){0.0, 0.0, 0.0, 0.0}, (v4d){0.0, 0.0, 0.0, 0.0}};
      int row = 16 * w + lr; row = row > 51 ? 51 : row;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const double a = L.Vxx[row * LDV + 4 * s + lk];
        acc[0] = mfma(a, breg[0][s], acc[0]);
        acc[1] = mfma(a, breg[1][s], acc[1]);
      }
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int rr = 16 * w + lk + 4 * r; if (rr < 52) L.G[rr * LDG + 16 * j2 + lr] = acc[j2][r]; }
    }
    // Qx = lx + A^T Vx (row tile w), Qu = lu + B^T Vx (waves 0, 1)
    {
      double s1 = 0.0;
#pragma unroll
      for (int s = 0; s < KS; ++s) s1 += areg[s] * L.Vx[4 * s + lk];
      s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
      const int i = 16 * w + lr;
      if (lk == 0 && i < n) L.Qx[i] = lxg[t * n + i] + s1;
      if (w < 2) {
        double s2 = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) s2 += breg[w][s] * L.Vx[4 * s + lk];
        s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
        const int a = 16 * w + lr;
        if (lk == 0 && a < m) L.Qu[a] = S.lu[((size_t)b * N + t) * m + a] + s2;
      }
    }
    __syncthreads();   // W, G complete; every wave is done reading Vxx
    // ---- P3: Qxx[tile w, :] = lxx + A^T W  -> Vxx buffer
    {
      v4d acc[4];
#pragma unroll
      for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * w + lk + 4 * r, col = 16 * J + lr;
          acc[J][r] = (row < n && col < n) ? lxxg[(size_t)t * n * n + row * n + col] : 0.0;
        }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int J = 0; J < 4; ++J) acc[J] = mfma(areg[s], L.W[(4 * s + lk) * LDW + 16 * J + lr], acc[J]);
      }
#pragma unroll
      for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * w + lk + 4 * r, col = 16 * J + lr; if (row < n && col < n) L.Vxx[row * LDV + col] = acc[J][r]; }
    }
    // ---- P4: Qxu[tile w, :] = A^T G
    {
      v4d acc[2] = {(v4d){0.0, 0.0, 0.0, 0.0}, (v4d){0.0, 0.0, 0.0, 0.0}};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        acc[0] = mfma(areg[s], L.G[(4 * s + lk) * LDG + lr], acc[0]);
        acc[1] = mfma(areg[s], L.G[(4 * s + lk) * LDG + 16 + lr], acc[1]);
      }
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * w + lk + 4 * r, col = 16 * j2 + lr; if (row < n && col < m) L.Qxu[row * LDQ + col] = acc[j2][r]; }
    }
    // ---- P5: Quu tile (w >> 1, w & 1) = B^T G (+ luu + lambda on the diagonal)
    {
      v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
      const int Ia = w >> 1, Jb = w & 1;
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = mfma(breg[Ia][s], L.G[(4 * s + lk) * LDG + 16 * Jb + lr], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * Ia + lk + 4 * r, col = 16 * Jb + lr;
        if (row < m && col < m) L.Quu[row * LDU + col] = acc[r] + ((row == col) ? (S.luu[((size_t)b * N + t) * m + row] + lam) : 0.0);
      }
    }
    __syncthreads();
    // ---- wave 0: Cholesky of Quu (row per lane, registers) + gains
    if (w == 0) {
      double row[RM];
      int fail = 0;
      for (int attempt = 0; attempt < 2; ++attempt) {
#pragma unroll
        for (int c = 0; c < RM; ++c) row[c] = (lane < m) ? L.Quu[lane * LDU + c] : 0.0;
        fail = 0;
#pragma unroll
        for (int j = 0; j < RM; ++j) {
          double s = row[j];
#pragma unroll
          for (int k = 0; k < j; ++k) s -= row[k] * bcast(row[k], j);
          const double piv = bcast(s, j);
          if (!(piv > 0.0)) fail = 1;
          const double d = sqrt(piv > 0.0 ? piv : 1.0);
          row[j] = (lane == j) ? d : s / d;
        }
        if (!fail) break;
        if (attempt == 0 && lane < m) L.Quu[lane * LDU + lane] += 1e-4;   // ilqr.cpp:280
      }
      if (lane == 0) L.flags[0] = fail;
      if (!fail) {
#pragma unroll
        for (int c = 0; c < RM; ++c) if (lane < m) L.Lc[lane * LDU + c] = row[c];
      }
    }
    __syncthreads();
    if (L.flags[0]) {
      // indefinite Quu even after the bump: explicit inverse by Gauss-Jordan with partial pivoting (rare)
      if (tid == 0) {
        double* Mx = L.W;   // W is free here (P3/P4 done); rows of 2m
        const int ld = 2 * RM;
        for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { Mx[i * ld + j] = L.Quu[i * LDU + j]; Mx[i * ld + m + j] = (i == j) ? 1.0 : 0.0; }
        for (int c = 0; c < m; ++c) {
          int p = c; double best = fabs(Mx[c * ld + c]);
          for (int r = c + 1; r < m; ++r) if (fabs(Mx[r * ld + c]) > best) { best = fabs(Mx[r * ld + c]); p = r; }
          if (p != c) for (int k = 0; k < 2 * m; ++k) { const double tmp = Mx[c * ld + k]; Mx[c * ld + k] = Mx[p * ld + k]; Mx[p * ld + k] = tmp; }
          const double ip = 1.0 / Mx[c * ld + c];
          for (int k = 0; k < 2 * m; ++k) Mx[c * ld + k] *= ip;
          for (int r = 0; r < m; ++r) if (r != c) { const double f = Mx[r * ld + c]; for (int k = 0; k < 2 * m; ++k) Mx[r * ld + k] -= f * Mx[c * ld + k]; }
        }
        for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) L.Lc[i * LDU + j] = Mx[i * ld + m + j];
      }
      __syncthreads();
    }
    // gains: one right-hand side per lane of wave 0 (51 columns of Qxu^T, then Qu)
    if (w == 0 && lane <= n) {
      double y[RM];
      if (lane < n) {
#pragma unroll
        for (int i = 0; i < RM; ++i) y[i] = L.Qxu[lane * LDQ + i];
      } else {
#pragma unroll
        for (int i = 0; i < RM; ++i) y[i] = L.Qu[i];
      }
      if (!L.flags[0]) {
#pragma unroll
        for (int i = 0; i < RM; ++i) {
          double s = y[i];
#pragma unroll
          for (int k = 0; k < i; ++k) s -= L.Lc[i * LDU + k] * y[k];
          y[i] = s / L.Lc[i * LDU + i];
        }
#pragma unroll
        for (int i = RM - 1; i >= 0; --i) {
          double s = y[i];
#pragma unroll
          for (int k = i + 1; k < RM; ++k) s -= L.Lc[k * LDU + i] * y[k];
          y[i] = s / L.Lc[i * LDU + i];
        }
      } else {
        double z[RM];
#pragma unroll
        for (int i = 0; i < RM; ++i) { double s = 0.0; for (int k = 0; k < RM; ++k) s += L.Lc[i * LDU + k] * y[k]; z[i] = s; }
#pragma unroll
        for (int i = 0; i < RM; ++i) y[i] = z[i];
      }
      if (lane < n) {
#pragma unroll
        for (int i = 0; i < RM; ++i) L.Kt[i * LDK + lane] = -y[i];
      } else {
#pragma unroll
        for (int i = 0; i < RM; ++i) L.kt[i] = -y[i];
      }
    }
    __syncthreads();
    // ---- store gains; P6: T1 = Qxx + Qxu K (row tile w) -> W buffer; Vx
    {
      double* Kg = S.K + ((size_t)b * N + t) * m * n;
      for (int e = tid; e < m * n; e += 256) Kg[e] = L.Kt[(e / n) * LDK + (e % n)];
      if (tid < m) S.kff[((size_t)b * N + t) * m + tid] = L.kt[tid];
      v4d acc[4];
#pragma unroll
      for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * w + lk + 4 * r, col = 16 * J + lr;
          acc[J][r] = (row < n && col < n) ? L.Vxx[row * LDV + col] : 0.0;
        }
      const int qrow = 16 * w + lr;
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int a = 4 * s + lk;
        const double qa = (a < LDQ) ? L.Qxu[qrow * LDQ + a] : 0.0;   // columns 19..21 are zero pads
#pragma unroll
        for (int J = 0; J < 4; ++J) acc[J] = mfma(qa, L.Kt[a * LDK + 16 * J + lr], acc[J]);
      }
#pragma unroll
      for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * w + lk + 4 * r, col = 16 * J + lr; if (row < 52) L.W[row * LDW + col] = acc[J][r]; }
    }
    double nvx = 0.0;
    if (tid < n) { double s = L.Qx[tid]; for (int a = 0; a < m; ++a) s += L.Qxu[tid * LDQ + a] * L.kt[a]; nvx = s; }
    __syncthreads();
    if (tid < n) L.Vx[tid] = nvx;
    for (int e = tid; e < n * n; e += 256) { const int i = e / n, j = e % n; L.Vxx[i * LDV + j] = 0.5 * (L.W[i * LDW + j] + L.W[j * LDW + i]); }
    __syncthreads();
  }
  for (int e = tid; e < n * n; e += 256) S.Vxx[(size_t)b * n * n + e] = L.Vxx[(e / n) * LDV + (e % n)];
  if (tid < n) S.Vx[(size_t)b * n + tid] = L.Vx[tid];
}

size_t backward_mfma_lds_bytes() { return sizeof(RiccatiLds); }
int backward_mfma_set_attr() {
  return hipFuncSetAttribute((const void*)k_backward_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RiccatiLds)) == hipSuccess ? 0 : 1;
}
void launch_backward_mfma(const DevState& S, int mode, hipStream_t st) {
  hipLaunchKernelGGL(k_backward_mfma, dim3(S.B), dim3(256), sizeof(RiccatiLds), st, S, mode);
}

}  // namespace ilqr
