// K4 (MFMA): Riccati backward pass, one 256-thread workgroup (4 waves) per rollout, fp64 matrix cores.
//
// Replaces iLQR::backwardPass (reference src/ilqr/ilqr.cpp:250-309).  All 51x51 / 51x19 contractions AND the
// gain solves of a knot run on v_mfma_f64_16x16x4_f64 (16x16 output tile, K step 4; C/D layout: lane l holds
// D[(l>>4) + 4r][l & 15], r = 0..3; A operand lane l = A[l & 15][l >> 4]; B operand lane l = B[l >> 4][l & 15]):
//   P1  W   = Vxx A          wave w -> column tile w of W           (A_t column tile held in registers)
//   P2  G   = Vxx B          wave w -> row tile w of G              (B_t held in registers)
//   P3  Qxx = lxx + A^T W    wave w -> row tile w  (A^T operand == the registers of P1)
//   P4  Qxu = A^T G          wave w -> row tile w  (row 51 of the Qxu buffer carries Qu, so k comes with K)
//   P5  Quu = luu + B^T G + lambda I   (2x2 tiles over the 4 waves)
//   --  wave 0: Cholesky Quu = L L^T and Linv = L^-1 in registers (row / column per lane, v_readlane
//       broadcasts; LLT failure -> +1e-4 I once, ilqr.cpp:278-281);  waves 1-3 meanwhile copy the next knot's
//       A, B, lxx from HBM into LDS staging, so the operand fetch never sits on the critical path
//   P6  Y = Linv Qxu^T,  [K | k] = -Linv^T Y     (wave w -> column tile w)
//   P7  Vxx = Qxx + 1/2 (Qxu K + K^T Qxu^T),  Vx = Qx + Qxu k
// The value-function update uses K^T Quu K + K^T Qxu^T = K^T (Quu K + Qxu^T) = 0 for the solved gains, i.e.
// Vxx = Qxx + Qxu K and Vx = Qx + Qxu k -- algebraically identical to the reference's long form
// (ilqr.cpp:294-307), differing only in rounding (covered by the parity tests); the symmetrisation
// 0.5 (V + V^T) of ilqr.cpp:307 is applied to the Qxu K term by running the product in both operand orders.
// LDS operands use leading dimensions chosen so that the MFMA operand reads are bank-conflict free:
// 54 / 22 for [i][k]-pattern reads, 80 / 48 for [k][j]-pattern reads of 64 / 32 columns.
#include <hip/hip_runtime.h>

#include "h1_dynamics_dev.h"
#include "ilqr_kernels.h"

namespace ilqr {

typedef double v4d __attribute__((ext_vector_type(4)));
// -DRIC_STAMP: diagnostic build only -- per-phase cycle sums of workgroup 0 / thread 0 land in S.J[0..15]
#ifdef RIC_STAMP
#define STAMP(k) { const long long tn_ = clock64(); ph[k] += tn_ - tlast; tlast = tn_; }
#else
#define STAMP(k)
#endif

#define RN 51
#define RM 19
#define KS 13          // k-steps of 4 over the padded inner dimension 52
#define LDV 54         // Vxx / Qxx rows [i][k]
#define LDW 80         // W rows [k][j], 64 columns
#define LDG 48         // G rows [k][j2], 32 columns
#define LDQ 22         // Qxu rows [i][a]
#define LDK 80         // K / Y rows [a][j]
#define LDU 20         // Quu rows
#define LDLA 22        // Linv rows read as [i][k]
#define LDLB 48        // Linv rows read as [k][i]

struct RiccatiLds {
  double Vxx[52 * LDV];   // rows 0..50 valid, row 51 and columns 51.. zero
  double W[52 * LDW];     // W = Vxx A; afterwards staging of the next knot's A_t (51 x 51, dense)
  double G[52 * LDG];     // G = Vxx B; afterwards staging of the next knot's B_t (51 x 19, dense)
  double Qxu[64 * LDQ];   // rows 0..50 = Qxu, row 51 = Qu
  double Kt[20 * LDK];    // K[a][j], column 51 = k
  double Y[20 * LDK];     // Linv Qxu^T
  double lxxS[RN * RN + 7];
  double LinvA[32 * LDLA];
  double LinvB[20 * LDLB];
  double Quu[RM * LDU];
  double Vx[64], Qx[64], Qu[32], kt[32];
  double lxS[64], luS[32], luuS[32];   // staged lx_t, lu_t, luu_t
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

  // zero every LDS buffer once (pads must be finite zeros), load the terminal value function and stage knot N-1
  for (int e = tid; e < (int)(sizeof(RiccatiLds) / sizeof(double)); e += 256) smem[e] = 0.0;
  __syncthreads();
  for (int e = tid; e < n * n; e += 256) L.Vxx[(e / n) * LDV + (e % n)] = lxxg[(size_t)N * n * n + e];
  if (tid < n) L.Vx[tid] = lxg[N * n + tid];
  // HBM -> LDS staging of A_t, B_t, lxx_t, lx_t, lu_t, luu_t; 8 loads in flight per thread
  // (dense rows of `width` doubles are re-pitched to `ld` so that the operand reads are conflict free)
  auto copy_pipelined = [&](double* dst, const double* src, int count, int width, int ld, int first, int nthreads) {
    for (int e0 = first; e0 < count; e0 += 8 * nthreads) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + u * nthreads; v[u] = (e < count) ? src[e] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + u * nthreads; if (e < count) dst[(e / width) * ld + (e % width)] = v[u]; }
    }
  };
  auto stage_knot = [&](int t, int first, int nthreads) {
    copy_pipelined(L.W, S.A + ((size_t)b * N + t) * n * n, n * n, n, LDW, first, nthreads);
    copy_pipelined(L.lxxS, lxxg + (size_t)t * n * n, n * n, n, n, first, nthreads);
    copy_pipelined(L.G, S.Bm + ((size_t)b * N + t) * n * m, n * m, m, LDG, first, nthreads);
    if (first < n) L.lxS[first] = lxg[t * n + first];
    else if (first >= 64 && first < 64 + m) { L.luS[first - 64] = S.lu[((size_t)b * N + t) * m + first - 64]; L.luuS[first - 64] = S.luu[((size_t)b * N + t) * m + first - 64]; }
  };
  stage_knot(N - 1, tid, 256);
  __syncthreads();

#ifdef RIC_STAMP
  long long ph[16] = {0}; long long tlast = clock64();
#endif
  for (int t = N - 1; t >= 0; --t) {
    // ---- operands from the LDS staging into registers
    double areg[KS];          // A[4s + lk][16w + lr]
    double breg[2][KS];       // B[4s + lk][16j2 + lr]
    {
      const int col = 16 * w + lr;
#pragma unroll
      for (int s = 0; s < KS; ++s) { const int k = 4 * s + lk; areg[s] = (k < n && col < n) ? L.W[k * LDW + col] : 0.0; }
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        const int c2 = 16 * j2 + lr;
#pragma unroll
        for (int s = 0; s < KS; ++s) { const int k = 4 * s + lk; breg[j2][s] = (k < n && c2 < m) ? L.G[k * LDG + c2] : 0.0; }
      }
    }
    STAMP(0)
    __syncthreads();   // staging consumed: W and G may be overwritten
    STAMP(1)
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
    STAMP(2)
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
    STAMP(3)
    // Qx = lx + A^T Vx (row tile w), Qu = lu + B^T Vx (waves 0, 1) -> also row 51 of the Qxu buffer
    {
      double s1 = 0.0;
#pragma unroll
      for (int s = 0; s < KS; ++s) s1 += areg[s] * L.Vx[4 * s + lk];
      s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
      const int i = 16 * w + lr;
      if (lk == 0 && i < n) L.Qx[i] = L.lxS[i] + s1;
      if (w < 2) {
        double s2 = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) s2 += breg[w][s] * L.Vx[4 * s + lk];
        s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
        const int a = 16 * w + lr;
        if (lk == 0 && a < m) { const double qu = L.luS[a] + s2; L.Qu[a] = qu; L.Qxu[51 * LDQ + a] = qu; }
      }
    }
    STAMP(4)
    __syncthreads();   // W, G complete; every wave is done reading Vxx
    STAMP(5)
    // ---- P3: Qxx[tile w, :] = lxx + A^T W  -> Vxx buffer
    {
      v4d acc[4];
#pragma unroll
      for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * w + lk + 4 * r, col = 16 * J + lr;
          acc[J][r] = (row < n && col < n) ? L.lxxS[row * n + col] : 0.0;
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
    STAMP(6)
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
        if (row < m && col < m) L.Quu[row * LDU + col] = acc[r] + ((row == col) ? (L.luuS[row] + lam) : 0.0);
      }
    }
    STAMP(7)
    __syncthreads();   // Quu, Qxu, Qxx complete; W and G are free again
    STAMP(8)
    if (w == 0) {
      // ---- wave 0: Cholesky of Quu (row per lane) and Linv = L^-1 (column per lane), all in registers
      double row[RM], dinv[RM];
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
          const double di = rsqrt(piv > 0.0 ? piv : 1.0);   // 1 / L[j][j]
          dinv[j] = di;
          row[j] = s * di;                                  // lane j: piv / sqrt(piv) = L[j][j]
        }
        if (!fail) break;
        if (attempt == 0 && lane < m) L.Quu[lane * LDU + lane] += 1e-4;   // ilqr.cpp:280
      }
      if (lane == 0) L.flags[0] = fail;
      if (!fail) {
        double x[RM];   // column `lane` of L^-1
#pragma unroll
        for (int i = 0; i < RM; ++i) {
          double s = (i == lane) ? 1.0 : 0.0;
#pragma unroll
          for (int k = 0; k < i; ++k) s -= bcast(row[k], i) * x[k];
          x[i] = s * dinv[i];
        }
        if (lane < m) {
#pragma unroll
          for (int i = 0; i < RM; ++i) { L.LinvA[i * LDLA + lane] = x[i]; L.LinvB[i * LDLB + lane] = x[i]; }
        }
      }
    } else if (t > 0) {
      // ---- waves 1-3: fetch the next knot's operands while wave 0 factorises
      stage_knot(t - 1, tid - 64, 192);
    }
    STAMP(9)
    __syncthreads();
    STAMP(10)
    if (!L.flags[0]) {
      // ---- P6a: Y[:, tile w] = Linv Qxu^T   (Qxu^T[k][j] = Qxu[j][k])
      {
        v4d acc[2] = {(v4d){0.0, 0.0, 0.0, 0.0}, (v4d){0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          const double bq = L.Qxu[(16 * w + lr) * LDQ + 4 * s + lk];
          acc[0] = mfma(L.LinvA[lr * LDLA + 4 * s + lk], bq, acc[0]);
          acc[1] = mfma(L.LinvA[(16 + lr) * LDLA + 4 * s + lk], bq, acc[1]);
        }
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int row = 16 * I + lk + 4 * r; if (row < 20) L.Y[row * LDK + 16 * w + lr] = acc[I][r]; }
      }
      __syncthreads();
      // ---- P6b: [K | k][:, tile w] = -Linv^T Y
      {
        v4d acc[2] = {(v4d){0.0, 0.0, 0.0, 0.0}, (v4d){0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          const double by = L.Y[(4 * s + lk) * LDK + 16 * w + lr];
          acc[0] = mfma(L.LinvB[(4 * s + lk) * LDLB + lr], by, acc[0]);
          acc[1] = mfma(L.LinvB[(4 * s + lk) * LDLB + 16 + lr], by, acc[1]);
        }
        double* Kg = S.K + ((size_t)b * N + t) * m * n;
        const int col = 16 * w + lr;
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int a = 16 * I + lk + 4 * r;
            if (a < m) {
              const double v = -acc[I][r];
              L.Kt[a * LDK + col] = v;
              if (col < n) Kg[a * n + col] = v;
              else if (col == n) { L.kt[a] = v; S.kff[((size_t)b * N + t) * m + a] = v; }
            }
          }
      }
    } else {
      // indefinite Quu even after the bump (rare): explicit inverse by Gauss-Jordan with partial pivoting,
      // standing in for the reference's pivoted LDLT; the staged operands in W are not touched (Y is scratch)
      if (tid == 0) {
        double* Mx = L.Y; const int ld = 2 * RM;
        for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { Mx[i * ld + j] = L.Quu[i * LDU + j]; Mx[i * ld + m + j] = (i == j) ? 1.0 : 0.0; }
        for (int c = 0; c < m; ++c) {
          int p = c; double best = fabs(Mx[c * ld + c]);
          for (int r = c + 1; r < m; ++r) if (fabs(Mx[r * ld + c]) > best) { best = fabs(Mx[r * ld + c]); p = r; }
          if (p != c) for (int k = 0; k < 2 * m; ++k) { const double tmp = Mx[c * ld + k]; Mx[c * ld + k] = Mx[p * ld + k]; Mx[p * ld + k] = tmp; }
          const double ip = 1.0 / Mx[c * ld + c];
          for (int k = 0; k < 2 * m; ++k) Mx[c * ld + k] *= ip;
          for (int r = 0; r < m; ++r) if (r != c) { const double f = Mx[r * ld + c]; for (int k = 0; k < 2 * m; ++k) Mx[r * ld + k] -= f * Mx[c * ld + k]; }
        }
      }
      __syncthreads();
      double* Kg = S.K + ((size_t)b * N + t) * m * n;
      for (int e = tid; e < m * (n + 1); e += 256) {
        const int a = e / (n + 1), j = e % (n + 1);
        double s = 0.0;
        for (int c = 0; c < m; ++c) s += L.Y[a * 2 * RM + m + c] * L.Qxu[j * LDQ + c];
        L.Kt[a * LDK + j] = -s;
        if (j < n) Kg[a * n + j] = -s; else { L.kt[a] = -s; S.kff[((size_t)b * N + t) * m + a] = -s; }
      }
    }
    STAMP(11)
    __syncthreads();
    STAMP(12)
    // ---- P7: Vxx <- Qxx + 1/2 (Qxu K + K^T Qxu^T) (row tile w); Vx <- Qx + Qxu k
    {
      v4d acc[4];
#pragma unroll
      for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * w + lk + 4 * r, col = 16 * J + lr;
          acc[J][r] = (row < n && col < n) ? L.Vxx[row * LDV + col] : ((row < n && col == n) ? L.Qx[row] : 0.0);   // column 51 carries Vx
        }
      const int qrow = 16 * w + lr;
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int a = 4 * s + lk;
        const double qa = (qrow < n) ? 0.5 * L.Qxu[qrow * LDQ + a] : 0.0;   // row 51 holds Qu, not part of Qxu
        const double ka = (qrow < n) ? 0.5 * L.Kt[a * LDK + qrow] : 0.0;
#pragma unroll
        for (int J = 0; J < 4; ++J) {
          const int jc = 16 * J + lr;
          acc[J] = mfma(qa, (jc < n) ? L.Kt[a * LDK + jc] : ((jc == n) ? 2.0 * L.Kt[a * LDK + jc] : 0.0), acc[J]);   // col 51: Qx + Qxu k
          acc[J] = mfma(ka, (jc < n) ? L.Qxu[jc * LDQ + a] : 0.0, acc[J]);
        }
      }
#pragma unroll
      for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * w + lk + 4 * r, col = 16 * J + lr;
          if (row < n && col < n) L.Vxx[row * LDV + col] = acc[J][r];
          else if (row < n && col == n) L.Vx[row] = acc[J][r];
        }
    }
    STAMP(13)
    __syncthreads();
    STAMP(14)
  }
#ifdef RIC_STAMP
  if (b == 0 && tid == 0) for (int k = 0; k < 16; ++k) S.J[k] = (double)ph[k];
#endif
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
