// K4 (MFMA): Riccati backward pass, one 256-thread workgroup (4 waves) per rollout, fp64 matrix cores.
//
// Replaces iLQR::backwardPass (reference src/ilqr/ilqr.cpp:250-309).  All 51x51 / 51x19 contractions AND the
// gain solves of a knot run on v_mfma_f64_16x16x4_f64 (16x16 output tile, K step 4; C/D layout: lane l holds
// D[(l>>4) + 4r][l & 15], r = 0..3; A operand lane l = A[l & 15][l >> 4]; B operand lane l = B[l >> 4][l & 15]).
// The C/D layout of one product IS the B-operand layout of the next one (register r of row tile I is k-step
// 4I + r), so chained products never leave the registers:
//   P1  W[:, w]   = Vxx A[:, w]            wave w -> column tile w, kept in registers
//   P2  G[w, 0..15] = Vxx[w, :] B[:, 0..15]  wave w -> row tile w, to LDS; G[:, 16..18] comes out of P1: the padding
//       columns 52..54 of A carry B[:, 16..18], and P3 turns them into Qxu[:, 16..18] (26 MFMA per wave saved)
//   P3  Qxx[:, w] = lxx + A^T W[:, w]      B operand = the accumulators of P1; result stays in registers until P7
//   P4  Qux[0..15, w] = G[:, 0..15]^T A[:, w]   column tile w in registers (column 51 is replaced by Qu)
//   P5  Quu       = luu + B^T G + lambda I (2x2 tiles over the 4 waves), to LDS
//   --  one wave (rotating per knot): right-looking Cholesky Quu = L L^T fused with the forward substitution for Linv = L^-1
//       (row of Quu / column of Linv per lane, one v_readlane broadcast of L[k][j] feeds both updates;
//       LLT failure -> +1e-4 I once, ilqr.cpp:278-281);  waves 1-3 meanwhile copy the next knot's A, B, lx, lu, luu
//       from HBM into LDS, so the operand fetch never sits on the critical path
//   P6a Y[:, w]   = Linv [Qux | Qu][:, w]  B operand = the accumulators of P4; to LDS (A operand of P7)
//   P6b [K | k][:, w] = -Linv^T Y[:, w]    B operand = the accumulators of P6a; straight to HBM
//   P7  Vxx[:, w] = Qxx[:, w] - Y^T Y[:, w]   accumulates onto the registers of P3
// Vectors ride in the padding: row 51 of the Vxx buffer holds Vx, so row 51 of W / G is A^T Vx / B^T Vx, row 51 of
// the Qxx accumulators is Qx = lx + A^T Vx, column 51 of Qux is Qu, column 51 of Y gives k, and row 51 of P7 is
// Vx = Qx - Y^T y_u.  The value-function update uses K = -Quu^-1 Qux with Quu = L L^T (the regularised Quu,
// as in the reference, which adds lambda in place): K^T Quu K + K^T Qux + Qxu K = -Qxu Quu^-1 Qux = -Y^T Y and
// K^T Quu k + K^T Qu + Qxu k = -Y^T y_u -- algebraically identical to the reference's long form
// (ilqr.cpp:294-307), symmetric by construction (ilqr.cpp:307), differing only in rounding (parity tests).
// LDS operands use leading dimensions chosen so that the MFMA operand reads are bank-conflict free:
// 54 / 22 for [i][k]-pattern reads, 80 / 48 for [k][j]-pattern reads of 64 / 32 columns.
// Compiled into the test library only (-DILQR_LEGACY_KERNELS, lib/libilqr_hip_legacy.so): ILQR_BACKWARD=wg, a cross-check family.
#ifndef ILQR_LEGACY_KERNELS
#include <hip/hip_runtime.h>
#include "ilqr_kernels.h"
namespace ilqr {
void launch_backward_mfma(const DevState&, int, hipStream_t) {}
int backward_mfma_set_attr() { return 0; }
size_t backward_mfma_lds_bytes() { return 0; }
}  // namespace ilqr
#else
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
#define LDV 54         // Vxx rows [i][k]
#define LDA 54         // A^T rows [j][k]
#define LDLA 22        // Linv rows read as [i][k]
// [k][j]-pattern buffers (G, Y, LinvB) use a power-of-two pitch and an XOR swizzle of the column by 16 on odd
// rows: a half-wave reads rows k, k+1 x 16 columns and lands on 32 distinct 8-byte bank pairs.
#define SWZ(k, c, ld) ((k) * (ld) + ((c) ^ (((k) & 1) << 4)))
#define LDG 32
#define LDY 64
#define LDLB 32
#define LDU 20         // Quu rows (aliases the Y buffer)

// 81,296 B: two workgroups per CU (160 KB LDS), i.e. two waves per SIMD -- while one wave of one rollout runs the
// latency-bound Cholesky, the other rollout's waves keep the matrix cores busy.
struct RiccatiLds {
  double Vxx[52 * LDV];   // rows/cols 0..50 = Vxx, row 51 = Vx, column 51 = don't care (finite)
  double At[55 * LDA];    // At[j][k] = A_t[k][j]; row 51 and column 51 stay zero; rows 52..54 = columns 16..18 of B_t (see P1)
  double G[52 * LDG];     // staging of B_t ([k][c] swizzled), then G = Vxx B with row 51 = B^T Vx
  double Y[20 * LDY];     // Quu (P5 .. Cholesky), then Y = Linv [Qux | Qu]; scratch of the indefinite fallback
  double LinvA[21 * LDLA];  // row 20 stays zero (rows 20..31 of the padded operand read it)
  double LinvB[20 * LDLB];
  double QH[52 * 4];      // Qxu[:, 16..18] (row 51: Qu[16..18]) on its way from wave 3's P3 accumulators to P6a
  double lxS[64], luS[32], luuS[32];   // staged lx_t, lu_t, luu_t
  int flags[4];
};

__device__ __forceinline__ v4d mfma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double bcast(double x, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}

// HBM -> LDS staging of A_t (transposed), B_t, lx_t, lu_t, luu_t by NW waves (thread f of 64 NW); 8 loads in
// flight per thread, affine addressing (lane = column, wave-strided rows) so nothing but two bases stays live
template <int NW>
__device__ __forceinline__ void stage_knot(RiccatiLds& L, const DevState& S, int b, int t, int f) {
  const int N = S.N, n = RN, m = RM;
  {
    const double* Ag = S.A + ((size_t)b * N + t) * n * n;
    const int j = f & 63, kk = f >> 6;
    for (int k0 = kk; k0 < n; k0 += 8 * NW) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int k = k0 + u * NW; v[u] = (k < n && j < n) ? Ag[k * n + j] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int k = k0 + u * NW; if (k < n && j < n) L.At[j * LDA + k] = v[u]; }
    }
  }
  {
    const double* Bg = S.Bm + ((size_t)b * N + t) * n * m;
    const int c = f & 31, kk = f >> 5;
    for (int k0 = kk; k0 < n; k0 += 8 * 2 * NW) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int k = k0 + u * 2 * NW; v[u] = (k < n && c < m) ? Bg[k * m + c] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u * 2 * NW;
        if (k < n && c < m) { L.G[SWZ(k, c, LDG)] = v[u]; if (c >= 16) L.At[(52 + c - 16) * LDA + k] = v[u]; }
      }
    }
  }
  if (f < n) L.lxS[f] = S.lx[((size_t)b * (N + 1) + t) * n + f];
  else if (f >= 64 && f < 64 + m) { L.luS[f - 64] = S.lu[((size_t)b * N + t) * m + f - 64]; L.luuS[f - 64] = S.luu[((size_t)b * N + t) * m + f - 64]; }
}

__global__ void __launch_bounds__(256, 2) k_backward_mfma(DevState S, int mode) {
  const int b = blockIdx.x;
  if (mode == MASK_ACTIVE && !S.active[b]) return;
  if (mode == MASK_RETRY && !(S.active[b] && S.need_retry[b])) return;
  // static allocation: LDS addresses are immediates (with a dynamic array every address constant is a hoisted,
  // spilled SGPR)
  __shared__ RiccatiLds L;
  double* const smem = reinterpret_cast<double*>(&L);
  double* const Quu = L.Y;
  const int tid = threadIdx.x, w0 = __builtin_amdgcn_readfirstlane(tid >> 6), lane0 = tid & 63;
  const int N = S.N, n = RN, m = RM;
  const double lam = S.lambda[b];
  const double* lxg = S.lx + (size_t)b * (N + 1) * n;
  const double* lxxg = S.lxx + (size_t)b * (N + 1) * n * n;

  // zero every LDS buffer once (pads must be finite zeros), load the terminal value function and stage knot N-1
  for (int e = tid; e < (int)(sizeof(RiccatiLds) / sizeof(double)); e += 256) smem[e] = 0.0;
  __syncthreads();
  for (int e = tid; e < n * n; e += 256) L.Vxx[(e / n) * LDV + (e % n)] = lxxg[(size_t)N * n * n + e];
  if (tid < n) L.Vxx[51 * LDV + tid] = lxg[N * n + tid];
  stage_knot<4>(L, S, b, N - 1, tid);
  __syncthreads();

#ifdef RIC_STAMP
  long long ph[16] = {0}; long long tlast = clock64();
#endif
  for (int t = N - 1; t >= 0; --t) {
    // lane / wave indices are re-derived behind an opaque barrier every knot: otherwise LICM hoists ~100 LDS
    // addresses and predicates out of this loop and the register allocator spills them (and reloads each knot)
    int lane = lane0, w = w0;
    asm volatile("" : "+v"(lane), "+s"(w));
    const int lr = lane & 15, lk = lane >> 4;
    const int jcol = 16 * w + lr;                  // the column this lane owns in every column-tile product
    const int jrow = jcol > 54 ? 51 : jcol;        // rows 52..54: B_t columns 16..18; beyond: the zero row 51
    // ---- operands from the LDS staging into registers
    double areg[KS];          // A[4s + lk][16w + lr]
    double breg[2][KS];       // B[4s + lk][16j2 + lr]
#pragma unroll
    for (int s = 0; s < KS; ++s) areg[s] = L.At[jrow * LDA + 4 * s + lk];
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const int c2 = 16 * j2 + lr;
#pragma unroll
      for (int s = 0; s < KS; ++s) { const int k = 4 * s + lk; breg[j2][s] = (k < n && c2 < m) ? L.G[SWZ(k, c2, LDG)] : 0.0; }
    }
    STAMP(0)
    __syncthreads();   // B staging consumed: G may be overwritten
    STAMP(1)
    // ---- P1: W[:, tile w] = Vxx A[:, tile w]   (row 51 of the Vxx buffer is Vx -> row 51 of W is A^T Vx)
    v4d wacc[4];
#pragma unroll
    for (int I = 0; I < 4; ++I) wacc[I] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int I = 0; I < 4; ++I) {
        int row = 16 * I + lr; row = row > 51 ? 51 : row;
        wacc[I] = mfma(L.Vxx[row * LDV + 4 * s + lk], areg[s], wacc[I]);
      }
    }
    STAMP(2)
    // ---- lxx_t straight from HBM into the accumulators of P3 (latency hidden behind P2)
    v4d qxx[4];
    {
      const double* lg = lxxg + (size_t)t * n * n;
#pragma unroll
      for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * I + lk + 4 * r;
          qxx[I][r] = (row < n && jcol < n) ? lg[row * n + jcol] : 0.0;
        }
    }
    // columns 52..54 of wave 3's W tile are Vxx B[:, 16..18] = G[:, 16..18] (the padding columns of A carry them for
    // free): into the G buffer for P5
    if (w == 3 && lr >= 4 && lr < 7) {
#pragma unroll
      for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * I + lk + 4 * r; if (row < 52) L.G[SWZ(row, 16 + lr - 4, LDG)] = wacc[I][r]; }
    }
    // ---- P2: G[tile w, 0..15] = Vxx[tile w, :] B[:, 0..15]   (row 51: B^T Vx)
    {
      v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
      int row = 16 * w + lr; row = row > 51 ? 51 : row;
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = mfma(L.Vxx[row * LDV + 4 * s + lk], breg[0][s], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int rr = 16 * w + lk + 4 * r; if (rr < 52) L.G[SWZ(rr, lr, LDG)] = acc[r]; }
    }
    STAMP(3)
    // ---- P3: Qxx[:, tile w] = lxx + A^T W[:, tile w]; row 51 <- Qx = lx + A^T Vx (row 51 of At is zero)
    if (lk == 3) qxx[3][0] = ((jcol < n) ? L.lxS[jcol] : 0.0) + wacc[3][0];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const double wb = wacc[s >> 2][s & 3];
#pragma unroll
      for (int I = 0; I < 4; ++I) {
        int row = 16 * I + lr; row = row > 51 ? 51 : row;
        qxx[I] = mfma(L.At[row * LDA + 4 * s + lk], wb, qxx[I]);
      }
    }
    // columns 52..54 of wave 3's tile are A^T G[:, 16..18] = Qxu[:, 16..18]; row 51 carries B^T Vx -> Qu[16..18]
    if (w == 3 && lr >= 4 && lr < 7) {
#pragma unroll
      for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * I + lk + 4 * r;
          if (row < 51) L.QH[row * 4 + lr - 4] = qxx[I][r];
          else if (row == 51) L.QH[51 * 4 + lr - 4] = L.luS[16 + lr - 4] + qxx[I][r];
        }
    }
    STAMP(4)
    __syncthreads();   // G and QH complete; every wave is done reading Vxx
    STAMP(5)
    // ---- P4: Qux[0..15, tile w] = G[:, 0..15]^T A[:, tile w]; column 51 <- Qu = lu + B^T Vx; rows 16..18 from QH
    v4d qux[2] = {(v4d){0.0, 0.0, 0.0, 0.0}, (v4d){0.0, 0.0, 0.0, 0.0}};
#pragma unroll
    for (int s = 0; s < KS; ++s) qux[0] = mfma(L.G[SWZ(4 * s + lk, lr, LDG)], areg[s], qux[0]);
    if (jcol == 51) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int a = lk + 4 * r; qux[0][r] = L.luS[a] + L.G[SWZ(51, a, LDG)]; }
    }
    qux[1][0] = (lk < 3 && jcol < 52) ? L.QH[jcol * 4 + lk] : 0.0;   // row 16 + lk (row 19 is padding)
    STAMP(6)
    // ---- P5: Quu tile (w >> 1, w & 1) = B^T G (+ luu + lambda on the diagonal)
    {
      v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
      const int Ia = w >> 1, Jb = w & 1;
#pragma unroll
      for (int s = 0; s < KS; ++s) { const int k = 4 * s + lk; acc = mfma(Ia ? breg[1][s] : breg[0][s], L.G[SWZ(k, 16 * Jb + lr, LDG)], acc); }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * Ia + lk + 4 * r, col = 16 * Jb + lr;
        if (row < m && col < m) Quu[row * LDU + col] = acc[r] + ((row == col) ? (L.luuS[row] + lam) : 0.0);
      }
    }
    STAMP(7)
    __syncthreads();   // Quu complete; At, G, lxS, luS, luuS are free for the next knot's staging
    STAMP(8)
    const int cw = (t + (b >> 8)) & 3;   // the factorising wave rotates, so co-resident workgroups load all four SIMDs evenly
    if (w == cw) {
      // ---- wave cw: right-looking Cholesky of Quu (row per lane) fused with Linv = L^-1 (column per lane)
      double row[RM], x[RM];
      int fail = 0;
      for (int attempt = 0; attempt < 2; ++attempt) {
#pragma unroll
        for (int c = 0; c < RM; ++c) { row[c] = (lane < m) ? Quu[lane * LDU + c] : 0.0; x[c] = (c == lane) ? 1.0 : 0.0; }
        fail = 0;
#pragma unroll
        for (int j = 0; j < RM; ++j) {
          const double piv = bcast(row[j], j);
          if (!(piv > 0.0)) fail = 1;
          const double di = rsqrt(piv > 0.0 ? piv : 1.0);   // 1 / L[j][j]
          const double lij = row[j] * di;                    // lane i >= j: L[i][j]
          const double xj = x[j] * di;                       // lane c: Linv[j][c]
          x[j] = xj;
#pragma unroll
          for (int k = j + 1; k < RM; ++k) {
            const double lkj = bcast(lij, k);                // L[k][j]
            row[k] -= lij * lkj;
            x[k] -= lkj * xj;
            asm volatile("" : "+v"(x[k]));                   // pin the update here: LLVM otherwise sinks the whole Linv chain
            __builtin_amdgcn_sched_barrier(0);             // below the loop and keeps 171 broadcasts alive in spilled SGPRs
          }
        }
        if (!fail) break;
        if (attempt == 0 && lane < m) Quu[lane * LDU + lane] += 1e-4;   // ilqr.cpp:280
      }
      if (lane == 0) L.flags[0] = fail;
      if (!fail && lane < m) {
#pragma unroll
        for (int i = 0; i < RM; ++i) { L.LinvA[i * LDLA + lane] = x[i]; L.LinvB[SWZ(i, lane, LDLB)] = x[i]; }
      }
    } else if (t > 0) {
      // ---- the other three waves: fetch the next knot's operands while wave cw factorises
      stage_knot<3>(L, S, b, t - 1, ((w - cw - 1) & 3) * 64 + lane);
    }
    STAMP(9)
    __syncthreads();
    STAMP(10)
    double* Kg = S.K + ((size_t)b * N + t) * m * n;
    double* kg = S.kff + ((size_t)b * N + t) * m;
    if (!L.flags[0]) {
      // ---- P6a: Y[:, tile w] = Linv [Qux | Qu][:, tile w]
      v4d yacc[2] = {(v4d){0.0, 0.0, 0.0, 0.0}, (v4d){0.0, 0.0, 0.0, 0.0}};
      const int ra1 = (16 + lr) > 20 ? 20 : (16 + lr);
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const double qb = qux[s >> 2][s & 3];
        yacc[0] = mfma(L.LinvA[lr * LDLA + 4 * s + lk], qb, yacc[0]);
        yacc[1] = mfma(L.LinvA[ra1 * LDLA + 4 * s + lk], qb, yacc[1]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) L.Y[SWZ(lk + 4 * r, jcol, LDY)] = yacc[0][r];
      L.Y[SWZ(16 + lk, jcol, LDY)] = yacc[1][0];
      // ---- P6b: [K | k][:, tile w] = -Linv^T Y[:, tile w]
      {
        v4d acc[2] = {(v4d){0.0, 0.0, 0.0, 0.0}, (v4d){0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          const double yb = yacc[s >> 2][s & 3];
          const int k = 4 * s + lk;
          acc[0] = mfma(L.LinvB[SWZ(k, lr, LDLB)], yb, acc[0]);
          acc[1] = mfma(L.LinvB[SWZ(k, 16 + lr, LDLB)], yb, acc[1]);
        }
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int a = 16 * I + lk + 4 * r;
            if (a < m) {
              if (jcol < n) Kg[a * n + jcol] = -acc[I][r];
              else if (jcol == n) kg[a] = -acc[I][r];
            }
          }
      }
      STAMP(11)
      __syncthreads();   // Y complete
      STAMP(12)
      // ---- P7: Vxx[:, tile w] = Qxx[:, tile w] - Y^T Y[:, tile w]; row 51: Vx = Qx - Y^T y_u
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const double yb = -yacc[s >> 2][s & 3];
        const int k = 4 * s + lk;
#pragma unroll
        for (int I = 0; I < 4; ++I) qxx[I] = mfma(L.Y[SWZ(k, 16 * I + lr, LDY)], yb, qxx[I]);
      }
    } else {
#ifndef NO_FALLBACK
      // indefinite Quu even after the bump (rare): explicit inverse by Gauss-Jordan with partial pivoting,
      // standing in for the reference's pivoted LDLT.  Scratch: the Vxx buffer (dead between P2 and P7) takes
      // [Qux | Qu] and the augmented matrix, the Y buffer takes [K | k] once Quu has been copied out of it;
      // the update is Vxx = Qxx + sym(Qxu K), Vx = Qx + Qxu k.
      double* Qd = L.Vxx;                    // Qd[a * 52 + j], a < 20
      double* Mx = L.Vxx + 20 * 52;          // 19 x 38 augmented
      double* Kd = L.Y;                      // Kd[a * 52 + j]
      const int ld = 2 * RM;
#pragma unroll
      for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int a = 16 * I + lk + 4 * r; if (a < 20 && jcol < 52) Qd[a * 52 + jcol] = qux[I][r]; }
      if (tid == 0) {
        for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { Mx[i * ld + j] = Quu[i * LDU + j]; Mx[i * ld + m + j] = (i == j) ? 1.0 : 0.0; }
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
      for (int e = tid; e < m * 52; e += 256) {
        const int a = e / 52, j = e % 52;
        double s = 0.0;
#pragma nounroll
        for (int c = 0; c < m; ++c) s += Mx[a * ld + m + c] * Qd[c * 52 + j];
        Kd[a * 52 + j] = -s;
        if (j < n) Kg[a * n + j] = -s; else kg[a] = -s;
      }
      __syncthreads();
#pragma unroll
      for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * I + lk + 4 * r;
          if (i < n && jcol < n) {
            double s = 0.0;
#pragma nounroll
            for (int a = 0; a < m; ++a) s += Qd[a * 52 + i] * Kd[a * 52 + jcol] + Kd[a * 52 + i] * Qd[a * 52 + jcol];
            qxx[I][r] += 0.5 * s;
          } else if (i == n && jcol < n) {
            double s = 0.0;
#pragma nounroll
            for (int a = 0; a < m; ++a) s += Qd[a * 52 + jcol] * Kd[a * 52 + n];
            qxx[I][r] += s;
          }
        }
      __syncthreads();   // scratch in the Vxx buffer consumed
#endif
    }
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int row = 16 * I + lk + 4 * r; if (row < 52 && jcol < 52) L.Vxx[row * LDV + jcol] = qxx[I][r]; }
    STAMP(13)
    __syncthreads();
    STAMP(14)
  }
#ifdef RIC_STAMP
  if (b == 0 && tid == 0) for (int k = 0; k < 16; ++k) S.J[k] = (double)ph[k];
#endif
  for (int e = tid; e < n * n; e += 256) S.Vxx[(size_t)b * n * n + e] = L.Vxx[(e / n) * LDV + (e % n)];
  if (tid < n) S.Vx[(size_t)b * n + tid] = L.Vxx[51 * LDV + tid];
}

size_t backward_mfma_lds_bytes() { return sizeof(RiccatiLds); }
int backward_mfma_set_attr() { return 0; }   // LDS is allocated statically
void launch_backward_mfma(const DevState& S, int mode, hipStream_t st) {
  hipLaunchKernelGGL(k_backward_mfma, dim3(S.B), dim3(256), 0, st, S, mode);
}

}  // namespace ilqr

#endif
