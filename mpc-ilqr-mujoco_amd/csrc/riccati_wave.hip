// K4 (MFMA, one wave per rollout): Riccati backward pass with the whole value function resident in registers.
//
// Replaces iLQR::backwardPass (reference src/ilqr/ilqr.cpp:250-309).  Same algebra as riccati_mfma.hip
// (Vxx <- Qxx - Y^T Y with Y = L^-1 [Qux | Qu], Quu = L L^T after "+ lambda I" and the single "+1e-4 I" bump of
// ilqr.cpp:275-281, K = -L^-T Y), but no workgroup: a single 64-lane wave owns a rollout, so there is no barrier,
// no critical path through a factorising wave, and no LDS staging of operands.  v_mfma_f64_16x16x4_f64 layouts:
// A operand lane l = A[l & 15][l >> 4], B operand lane l = B[l >> 4][l & 15], C/D lane l register r =
// D[(l >> 4) + 4r][l & 15].  With (lk, lr) = (l >> 4, l & 15) a C/D tile (I, J) register r holds element
// (16 I + 4 r + lk, 16 J + lr), which is at once
//   * the B operand of k-step s = 4 I + r for column tile J          (C layout == B layout), and
//   * the A operand of k-step s = 4 I + r of the TRANSPOSED matrix for row tile J.
// Every product of the knot therefore takes its operands straight from the accumulators of the previous one:
//   M    augmented value function, 64 x 64 in 16 accumulator tiles: [0..50]^2 = Vxx, row 51 = column 51 = Vx
//        (symmetric, so M itself is its own transposed A operand), rows / columns 52..54 scratch
//   A~   51 x 51 A_t padded: A~[51][51] = 1 (carries the vector slot through both products), columns 52..54 =
//        B_t[:, 16..18]; loaded from HBM directly in operand layout (lane = 16 consecutive doubles of 4 rows)
//   P2   G0 = M B_t[:, 0..15]                       (52 MFMA)      A operand = M tiles, B operand = B_t from HBM
//   P4   Qux~[0..15, :] = G0^T A~                   (52)           A operand = G0 accumulators; column 51 + lu = Qu,
//                                                                  columns 52..54 = Quu[0..15, 16..18]
//   P5   Quu[0..15, 0..15] = B0^T G0                (13)
//   P1   W = M A~                                   (208)          row / column 51 of W = A^T Vx / Vx
//   P3   Q = lxx~ + A~^T W                          (130)          lxx~ = lxx with lx in row and column 51, straight
//        from HBM into the accumulators; row / column 51 of Q = Qx; rows 52..54 = B_t[:,16..18]^T W =
//        Qux[16..18, :] (column 51: Qu[16..18] - lu, columns 52..54: Quu[16..18, 16..18]) for free
//   --   Quu -> LDS (1.5 KB), Cholesky + L^-1 on the same wave (row of Quu / column of L^-1 per lane)
//   P6a  Y = L^-1 Qux~ (40), P6b [K | k] = -L^-T Y (24: rows 16..18 only need k >= 16; straight to HBM),
//   P7   M <- Q - Y^T Y (50, tiles I >= J)
// 569 MFMA per knot with P3 / P7 on the lower tiles only (732 in the four-wave kernel); 429 in the folded variant
// (template parameter FOLD, see fold_rows below) that the analytic Jacobians select.  ~400 live registers at the peak
// (one wave per SIMD).
// LDS holds Quu, L^-1 in the two operand layouts and the scratch of the indefinite-Quu fallback only.
#include <hip/hip_runtime.h>

#include "ilqr_kernels.h"

namespace ilqr {

typedef double v4d __attribute__((ext_vector_type(4)));

#define WN 51
#define WM 19
#define WKS 13
#define WLDU 20
#define WLDLA 22
#define WLDLB 32
#define WSWZ(k, c, ld) ((k) * (ld) + ((c) ^ (((k) & 1) << 4)))

// -DWAVE_STAMP: diagnostic build only -- per-phase cycle sums of rollout 0 land in S.J[0..15]
#ifdef WAVE_STAMP
#define WSTAMP(k) { const long long tn_ = clock64(); ph[k] += tn_ - tlast; tlast = tn_; }
#else
#define WSTAMP(k)
#endif

#define WLDT 17
#define WLDQ 20
// 36,864 B per wave: four waves (rollouts) per CU.
struct WaveLds {
  double QL[20 * WLDQ];          // Quu for the Cholesky, then Linv = L^-1 (row 19 and column 19 zero)
  double col[2][64];             // columns of L on their way from the row lanes to every lane (double-buffered)
  double T[3][16 * WLDT];        // transposition of the strictly lower tiles after P7, three at a time
  union {
    double Aop[53 * 64];         // A~ of the knot in operand order: tile (T, s) = 64 doubles [lk][lr] at (13 T + s) * 64
    struct {
      double Qd[20 * 52];        // fallback: [Qux | Qu]
      double Kd[20 * 52];        // fallback: [K | k]
      double Mx[WM * 40];        // fallback: augmented Gauss-Jordan matrix [Quu | I], pitch 40
    };
  };
};

__device__ __forceinline__ v4d wmfma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double wbcast(double x, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}

// Folded knots (template parameter FOLD of the kernel).  The semi-implicit Euler step makes the hinge-position rows of the
// analytic Jacobians exact copies of the hinge-velocity rows: A_t[7 + j][:] = e_(7+j)^T + h A_t[32 + j][:] and
// B_t[7 + j][:] = h B_t[32 + j][:] (h1_linearize_dev.h lin_column).  For the sixteen rows k = 8..23 -- k-steps 2..5 of
// every product that contracts over the rows of A~ or B_t -- the contribution  X[:, k] A~[k][:]  is therefore
// X[:, k] e_k^T + h X[:, k] A~[k + 25][:]: fold h X[:, k] into X[:, k + 25] once (rows 33..48 of the operand += h * rows
// 8..23, a rotation by 16 lanes plus a register shift in C layout), add the identity part X[:, 8..23] to columns 8..23
// of the result, and drop the four k-steps.  140 of the 585 MFMA of a knot go away (P1 64, P3 40, P2 16, P4 16, P5 4)
// together with 16 of the 51 rows of A_t and of B_t[:, 0..15] that no longer have to be fetched.
// wrot16: value of lane - 16 (mod 64), i.e. of the same column one lk earlier.
__device__ __forceinline__ double wrot16(double x, int raddr) {
  const int lo = __builtin_amdgcn_ds_bpermute(raddr, __double2loint(x));
  const int hi = __builtin_amdgcn_ds_bpermute(raddr, __double2hiint(x));
  return __hiloint2double(hi, lo);
}
// One column of tiles t0..t3 (rows 0..63 of 16 columns, register r of tile I = row 16 I + 4 r + lk): rows 33..48 += h * rows
// 8..23.  Row k = 4 s + lk takes row k - 25 = 4 (s - 6) + lk - 1: the lane one lk earlier, register s - 6 (lk >= 1) or,
// wrapping to lk = 3, register s - 7.
__device__ __forceinline__ void fold_rows(const v4d& t0, const v4d& t1, v4d& t2, v4d& t3, int lk, int raddr, double h) {
  const bool top = lk == 3;
  const double r8 = wrot16(top ? t0[1] : t0[2], raddr);
  const double r9 = wrot16(top ? t0[2] : t0[3], raddr);
  const double r10 = wrot16(top ? t0[3] : t1[0], raddr);
  const double r11 = wrot16(top ? t1[0] : t1[1], raddr);
  const double r12 = wrot16(t1[1], raddr);
  t2[0] = (lk >= 1) ? __builtin_fma(h, r8, t2[0]) : t2[0];      // row 32 stays
  t2[1] = __builtin_fma(h, r9, t2[1]);
  t2[2] = __builtin_fma(h, r10, t2[2]);
  t2[3] = __builtin_fma(h, r11, t2[3]);
  t3[0] = (lk == 0) ? __builtin_fma(h, r12, t3[0]) : t3[0];     // row 48 only
}
// The same fold on the COLUMNS of one register row of M (four tiles c0..c3 of the same rows): columns 33..48 += h * columns 8..23.
// Column 33 + m is (tile 2, lr = 1 + m) for m < 15 and (tile 3, lr = 0) for m = 15; its source column 8 + m is (tile 0, lr = 8 + m)
// for m < 8 and (tile 1, lr = m - 8) beyond: in every case the lane seven to the right in the 16-lane row (mod 16), so the
// sending lane picks the tile (lr >= 8: tile 0, else tile 1) and one rotation serves all sixteen columns.
// Round 3: with M folded on BOTH sides (M~ = F M F^T) every product that has M on the left delivers its result with the rows
// 33..48 already folded -- W = M~ A~, G0 = M~ B0 -- so the five per-product folds of G0 and W (accumulator tiles: a read and a
// write of the AGPRs for every touched register) are replaced by nine register rows of M (the k-steps that are not skipped).
__device__ __forceinline__ void fold_cols(const v4d& c0, const v4d& c1, v4d& c2, v4d& c3, int r, int lr, int caddr, double h) {
  const double y = wrot16(lr >= 8 ? c0[r] : c1[r], caddr);
  c2[r] = (lr >= 1) ? __builtin_fma(h, y, c2[r]) : c2[r];      // column 32 stays
  c3[r] = (lr == 0) ? __builtin_fma(h, y, c3[r]) : c3[r];      // column 48 only
}
#define WFOLD_SKIP(s) (FOLD && (s) >= 2 && (s) <= 5)

// Augmented cost Hessian of one knot in C layout: lxx inside, lx in row 51 and column 51, zeros beyond.  Every load is
// unconditional from an in-range address and masked afterwards (no branches: the 64 loads issue back to back).
// LOWER: only the tiles I >= J (the strictly upper ones are filled by transposition after P7).
template <bool LOWER>
__device__ __forceinline__ void load_aug(v4d (&Q)[4][4], const double* lxx, const double* lx, int lk, int lr) {
  const unsigned off = (unsigned)(lk * WN + lr);
#pragma unroll
  for (int I = 0; I < 3; ++I)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int J = 0; J < 3; ++J) if (!LOWER || I >= J) Q[I][J][r] = (lxx + ((16 * I + 4 * r) * WN + 16 * J))[off];
      if (!LOWER) {
        // column tile 3: columns 48..50 of lxx, lx in column 51
        const int row = 16 * I + 4 * r + lk;
        const double* p = (lr < 3) ? (lxx + row * WN + 48 + lr) : (lx + row);
        const double v = *p;
        Q[I][3][r] = (lr <= 3) ? v : 0.0;
      }
    }
  // row tile 3: rows 48..50 of lxx, lx in row 51 (register 0 only; rows 52.. are zero)
#pragma unroll
  for (int J = 0; J < 4; ++J) {
    const int col = 16 * J + lr, cc = col < WN ? col : 0;
    const double* p = (lk < 3) ? ((col < WN) ? (lxx + (48 + lk) * WN + col) : (lx + 48 + lk)) : (lx + cc);
    const double v = *p;
    const bool ok = (lk < 3) ? (col <= WN) : (col < WN);
    Q[3][J][0] = ok ? v : 0.0;
#pragma unroll
    for (int r = 1; r < 4; ++r) Q[3][J][r] = 0.0;
  }
}

// v / sqrt(x) for a positive, normal x: hardware estimate y (v_rsq_f64, ~2^-26) and one Newton correction applied to the
// product (a = v y, e = 1 - x y^2, a + a e / 2: relative error 3 e^2 / 8 ~ 1e-16), arranged for a short dependent chain
__device__ __forceinline__ double scale_rsqrt(double v, double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double a = v * y, t = x * y;
  const double e = __builtin_fma(-t, y, 1.0);
  return __builtin_fma(0.5 * a, e, a);
}

// One sweep of the fused factorisation over the 19 registers of a lane (see the kernel); returns 1 if a pivot is not
// positive (LLT failure in the sense of ilqr.cpp:278).  The sweep is a latency chain (update, broadcast pivot, rsqrt,
// scale, broadcast the column), so: the single entry of column j the next pivot needs, L[j+1][j], is broadcast with
// v_readlane; the rest of the column goes through LDS (published by the row lanes, read back by every lane at
// wave-uniform addresses, one 16-byte broadcast read per two entries) and feeds the bulk of the updates, which are
// pinned in right-looking order so that they fill the gaps of the chain instead of extending it.
struct CholSweep {
  int fail;
  double vj, c1;
  double c[WM];
  __device__ __forceinline__ void begin(double (&v)[WM], double (&col)[2][64], int lane) {
    fail = 0;
    const double piv = wbcast(v[0], 0);
    if (!(piv > 0.0)) fail = 1;
    vj = scale_rsqrt(v[0], piv > 0.0 ? piv : 1.0);
    v[0] = vj;
    col[0][lane] = vj;   // every lane stores (no exec change); entries 0..18 are the column of L
    c1 = wbcast(vj, 1);
#pragma unroll
    for (int k = 2; k < WM; ++k) c[k] = col[0][k];
  }
  // step j = 0..17 (compile-time constant after unrolling)
  __device__ __forceinline__ void step(int j, double (&v)[WM], double (&col)[2][64], int lane) {
    // critical chain: finish register j + 1, pivot, scale, publish, broadcast the entry the next step starts with
    v[j + 1] = __builtin_fma(-vj, c1, v[j + 1]);
    const double piv = wbcast(v[j + 1], j + 1);
    if (!(piv > 0.0)) fail = 1;
    const double vn = scale_rsqrt(v[j + 1], piv > 0.0 ? piv : 1.0);
    v[j + 1] = vn;
    double cn[WM];
    double c1n = 0.0;
    if (j + 2 < WM) {
      col[(j + 1) & 1][lane] = vn;
      c1n = wbcast(vn, j + 2);
#pragma unroll
      for (int k = j + 3; k < WM; ++k) cn[k] = col[(j + 1) & 1][k];
    }
    // bulk of step j (register j + 2 first: the next step needs it)
#pragma unroll
    for (int k = j + 2; k < WM; ++k) { v[k] = __builtin_fma(-vj, c[k], v[k]); asm volatile("" : "+v"(v[k])); }
#pragma unroll
    for (int k = j + 3; k < WM; ++k) c[k] = cn[k];
    vj = vn;
    c1 = c1n;
  }
};
__device__ __forceinline__ int chol_linv(double (&v)[WM], double (&col)[2][64], int lane) {
  CholSweep cs;
  cs.begin(v, col, lane);
#pragma unroll
  for (int j = 0; j < WM - 1; ++j) cs.step(j, v, col, lane);
  return cs.fail;
}

// HBM -> LDS staging of A~ of one knot without passing through registers (global_load_lds_dwordx4: every lane
// fetches 16 bytes, the wave's 1 KB lands contiguously in lane order).  One instruction fills two operand tiles
// (T, 2j) and (T, 2j + 1): lane = (h, lk, p) fetches the pair A~[4 (2j + h) + lk][16 T + 2p .. 2p + 1].  Column tile 3
// gathers columns 48..50 of A_t (+ one junk double), columns 16..18 of B_t (+ junk); fix_A masks the junk and sets
// the unit entry.  Row 51 does not exist in A_t: those lanes fetch row 50 and are masked as well.
template <bool FOLD>
__device__ __forceinline__ void stage_A(WaveLds& L, const double* Ap, const double* Bp, int lane) {
  typedef const __attribute__((address_space(1))) void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;
  const int h = lane >> 5, lk = (lane >> 3) & 3, p = lane & 7;
  const int row0 = 4 * h + lk;
  const int row6 = (48 + row0) < 50 ? (48 + row0) : 50;
  const double* b012 = Ap + row0 * WN + 2 * p;
  const double* b012_6 = Ap + row6 * WN + 2 * p;
  const bool useB = (p == 2) || (p == 3);
  const double* ptr3 = (p == 0) ? (Ap + 48) : (p == 1) ? (Ap + 50) : (p == 2) ? (Bp + 16) : (p == 3) ? (Bp + 18) : Ap;
  const int pitch3 = useB ? WM : WN;
  const double* b3 = ptr3 + row0 * pitch3;
  const double* b3_6 = ptr3 + row6 * pitch3;
  const int st3 = 8 * pitch3;
#pragma unroll
  for (int T = 0; T < 3; ++T) {
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (!(FOLD && (j == 1 || j == 2)))     // folded: rows 8..23 are never used
        __builtin_amdgcn_global_load_lds((gptr)(b012 + 8 * j * WN + 16 * T), (lptr)&L.Aop[(T * 13 + 2 * j) * 64], 16, 0, 0);
    if (lane < 32) __builtin_amdgcn_global_load_lds((gptr)(b012_6 + 16 * T), (lptr)&L.Aop[(T * 13 + 12) * 64], 16, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < 6; ++j)
    if (!(FOLD && (j == 1 || j == 2)))
      __builtin_amdgcn_global_load_lds((gptr)(b3 + j * st3), (lptr)&L.Aop[(39 + 2 * j) * 64], 16, 0, 0);
  if (lane < 32) __builtin_amdgcn_global_load_lds((gptr)b3_6, (lptr)&L.Aop[(39 + 12) * 64], 16, 0, 0);
}

// B_t[:, 0..15] in B-operand layout; row 51 does not exist (clamped to row 50, zeroed)
template <bool FOLD>
__device__ __forceinline__ void load_b0(double (&b0)[WKS], const double* Bp, int lk, int lr) {
  const unsigned off = (unsigned)(lk * WM + lr);
#pragma unroll
  for (int s = 0; s < WKS - 1; ++s) if (!WFOLD_SKIP(s)) b0[s] = (Bp + 4 * s * WM)[off];
  const int lkc = lk < 3 ? lk : 2;
  const double v = (Bp + 48 * WM)[(unsigned)(lkc * WM + lr)];
  b0[WKS - 1] = (lk < 3) ? v : 0.0;
}

// After the staged data has landed: zero row 51 (k-step 12, lk = 3) of every column tile, and in column tile 3 keep
// columns 48..50 (A_t) and 52..54 (B_t[:, 16..18]), set A~[51][51] = 1, zero the rest.
template <bool FOLD>
__device__ __forceinline__ void fix_A(WaveLds& L, int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  const bool keep = (lr < 3) || (lr >= 4 && lr < 7);
#pragma unroll
  for (int s = 0; s < WKS; ++s) {
    if (WFOLD_SKIP(s)) continue;
    const double v = L.Aop[(39 + s) * 64 + lane];
    double w = keep ? v : 0.0;
    if (s == WKS - 1 && lk == 3) w = (lr == 3) ? 1.0 : 0.0;
    L.Aop[(39 + s) * 64 + lane] = w;
  }
  if (lk == 3) {
#pragma unroll
    for (int T = 0; T < 3; ++T) L.Aop[(T * 13 + 12) * 64 + lane] = 0.0;
  }
}

// Quu^-1 by Gauss-Jordan with partial pivoting (Quu in L.QL, pitch WLDQ), result back into L.QL as an operand buffer
// (row 19 and column 19 zero).  Lane = (row r = lane % 19, column group q = lane / 19 of 13 columns); rows are not
// swapped: perm[c] remembers which row became the pivot of column c.  Scratch aliases the A~ staging area.
__device__ __forceinline__ void gauss_jordan_inverse(WaveLds& L, int lane) {
  constexpr int m = WM, ld = 40;
  double* Mx = L.Mx;            // [19][40]: [Quu | I | pad]
  int* perm = reinterpret_cast<int*>(L.Kd);
  for (int e = lane; e < m * ld; e += 64) {
    const int i = e / ld, j = e % ld;
    Mx[e] = (j < m) ? L.QL[i * WLDQ + j] : ((j - m == i) ? 1.0 : 0.0);
  }
  __syncthreads();
  const int r = lane % m, q = lane / m;
  unsigned used = 0u;
  for (int c = 0; c < m; ++c) {
    double a = (lane < m && !((used >> lane) & 1u)) ? fabs(Mx[lane * ld + c]) : -1.0;
    int idx = lane;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
      const int lo = __shfl_xor(__double2loint(a), off), hi = __shfl_xor(__double2hiint(a), off);
      const double a2 = __hiloint2double(hi, lo);
      const int i2 = __shfl_xor(idx, off);
      if (a2 > a || (a2 == a && i2 < idx)) { a = a2; idx = i2; }
    }
    const int p = __builtin_amdgcn_readfirstlane(idx);
    used |= 1u << p;
    if (lane == 0) perm[c] = p;
    const double ip = 1.0 / Mx[p * ld + c];
    const double f = Mx[r * ld + c] * ip;
    __syncthreads();
    if (q < 3) {
#pragma unroll
      for (int kk = 0; kk < 13; ++kk) {
        const int k = q * 13 + kk;
        const double pk = Mx[p * ld + k], own = Mx[r * ld + k];
        // (all lanes of the wave read before any writes: one instruction stream)
        Mx[r * ld + k] = (r == p) ? own * ip : own - f * pk;
      }
    }
    __syncthreads();
  }
  for (int e = lane; e < 20 * WLDQ; e += 64) {
    const int c = e / WLDQ, j = e % WLDQ;
    L.QL[e] = (c < m && j < m) ? Mx[perm[c] * ld + m + j] : 0.0;
  }
  __syncthreads();
}

template <bool FOLD>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) k_backward_wave(DevState S, int mode, double fh, const int* list, const int* count) {
  int b = blockIdx.x;
  if (list) {                       // compacted selection (DevState::order): the first *count blocks take the listed rollouts
    if (b >= *count) return;
    b = list[b];
  } else {
    if (mode == MASK_ACTIVE && !S.active[b]) return;
    if (mode == MASK_RETRY && !(S.active[b] && S.need_retry[b])) return;
  }
  __shared__ WaveLds L;
  const int lane0 = threadIdx.x;
  const int N = S.N;
  constexpr int n = WN, m = WM;
  const double lam = S.lambda[b];
  const double* lxg = S.lx + (size_t)b * (N + 1) * n;
  const double* lxxg = S.lxx + (size_t)b * (N + 1) * n * n;

  v4d M[4][4];
  load_aug<false>(M, lxxg + (size_t)N * n * n, lxg + N * n, lane0 >> 4, lane0 & 15);
  // knot N-1: A~ into LDS, B_t[:, 0..15] into registers (b0[s] = B[4s + lk][lr]; row 51 is masked at use)
  double b0[WKS];
  {
    const double* Ap = S.A + ((size_t)b * N + (N - 1)) * n * n;
    const double* Bp = S.Bm + ((size_t)b * N + (N - 1)) * n * m;
    stage_A<FOLD>(L, Ap, Bp, lane0);
    load_b0<FOLD>(b0, Bp, lane0 >> 4, lane0 & 15);
  }

#ifdef WAVE_STAMP
  long long ph[16] = {0}; long long tlast = clock64();
#endif
  for (int t = N - 1; t >= 0; --t) {
    // lane indices re-derived behind an opaque barrier every knot (keeps LICM from hoisting per-lane addresses and
    // predicates of the whole knot out of the loop, where they would be spilled)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int lr = lane & 15, lk = lane >> 4;
    const double* Ag = S.A + ((size_t)b * N + t) * n * n;
    const double* Bg = S.Bm + ((size_t)b * N + t) * n * m;
    const double* lug = S.lu + ((size_t)b * N + t) * m;
    const double* luug = S.luu + ((size_t)b * N + t) * m;
    // ---- A~ was staged into LDS during the previous knot: wait for it, mask the junk; B_t[:, 0..15] arrived in
    // registers (b0); lxx_t goes straight from HBM into the accumulators of P3 (issued now, needed after P1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifndef WAVE_SKIP_FIX
    fix_A<FOLD>(L, lane);
    __syncthreads();
#endif
#define AOP(T, s) L.Aop[((T) * 13 + (s)) * 64 + lane]
    v4d Q[4][4];
    load_aug<true>(Q, lxxg + (size_t)t * n * n, lxg + t * n, lk, lr);
    // lu_t, luu_t for the lanes that will need them (rows 4r + lk and 16 + lk), fetched now, used after P3
    double lu4[4], luu4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { lu4[r] = lug[4 * r + lk]; luu4[r] = luug[4 * r + lk]; }
    const int lkc3 = lk < 3 ? lk : 2;
    const double lu16 = lug[16 + lkc3], luu16 = luug[16 + lkc3];
    WSTAMP(0)
    // ---- folded: keep the identity part M[:, 8..23] of W for the rows about to change (row tiles 2, 3; rows 52.. of W are
    // never used), then rows 33..48 of M += h * rows 8..23 for every product that has M on the left (P2, P1)
    const int raddr = ((lane - 16) & 63) << 2;
    const int caddr = ((lane & 48) | ((lane + 7) & 15)) << 2;
    if (FOLD) {
      // rows 33..48 of M += h * rows 8..23 (contraction over the rows of A~ / B_t), then the same on the columns of the register
      // rows that serve as k-steps (results arrive with their rows 33..48 folded): M~ = F M F^T
#pragma unroll
      for (int I = 0; I < 4; ++I) fold_rows(M[0][I], M[1][I], M[2][I], M[3][I], lk, raddr, fh);
#pragma unroll
      for (int s = 0; s < WKS; ++s)
        if (!WFOLD_SKIP(s)) fold_cols(M[s >> 2][0], M[s >> 2][1], M[s >> 2][2], M[s >> 2][3], s & 3, lr, caddr, fh);
    }
    // ---- P2: G0 = M B0 (row 51: B0^T Vx)
    v4d g0[4];
#pragma unroll
    for (int I = 0; I < 4; ++I) g0[I] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < WKS; ++s)
#pragma unroll
      for (int I = 0; I < 4; ++I) if (!WFOLD_SKIP(s)) g0[I] = wmfma(M[s >> 2][I][s & 3], b0[s], g0[I]);
    if (FOLD) {
      // identity part of P4: Qux~[u][k] += G0[k][u] for k = 8..23, the transposes of two tiles of G0 through LDS
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) L.T[q][(4 * r + lk) * WLDT + lr] = g0[q][r];
      // (rows 33..48 of G0 arrive folded -- M~ on the left -- for the products that contract over its rows: P4, P5)
      __syncthreads();
    }
    WSTAMP(1)
    // ---- P4: Qux~[0..15, :] = G0^T A~
    v4d qux0[4];
#pragma unroll
    for (int J = 0; J < 4; ++J) qux0[J] = (v4d){0.0, 0.0, 0.0, 0.0};
    if (FOLD) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double t0 = L.T[0][lr * WLDT + 4 * r + lk], t1 = L.T[1][lr * WLDT + 4 * r + lk];
        qux0[0][r] = lr >= 8 ? t0 : 0.0;
        qux0[1][r] = lr < 8 ? t1 : 0.0;
      }
    }
#pragma unroll
    for (int s = 0; s < WKS; ++s)
#pragma unroll
      for (int J = 0; J < 4; ++J) if (!WFOLD_SKIP(s)) qux0[J] = wmfma(g0[s >> 2][s & 3], AOP(J, s), qux0[J]);
    // ---- P5: Quu[0..15, 0..15] = B0^T G0 (two interleaved accumulators)
    v4d quu0 = (v4d){0.0, 0.0, 0.0, 0.0}, quu1 = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < WKS; ++s) {
      if (WFOLD_SKIP(s)) continue;
      if (s & 1) quu1 = wmfma(b0[s], g0[s >> 2][s & 3], quu1);
      else quu0 = wmfma(b0[s], g0[s >> 2][s & 3], quu0);
    }
    WSTAMP(2)
    // ---- P1, column tile 3 first: W[:, 3] = M A~[:, 48..63] (its columns 52..54 are G[:, 16..18])
    v4d W[4][4];
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
      for (int J = 0; J < 4; ++J) W[I][J] = (v4d){0.0, 0.0, 0.0, 0.0};
    if (FOLD) {
      // identity part of P1: W[:, 8..23] starts from the row-folded M[:, 8..23] (columns 8..23 are not touched by the column fold)
#pragma unroll
      for (int J = 0; J < 2; ++J) {
        const bool in = J == 0 ? lr >= 8 : lr < 8;
#pragma unroll
        for (int r = 0; r < 4; ++r) { W[0][J][r] = in ? M[0][J][r] : 0.0; W[1][J][r] = in ? M[1][J][r] : 0.0; W[2][J][r] = in ? M[2][J][r] : 0.0; }
        W[3][J][0] = in ? M[3][J][0] : 0.0;
      }
    }
#pragma unroll
    for (int s = 0; s < WKS; ++s) {
      if (WFOLD_SKIP(s)) continue;
      const double a3 = AOP(3, s);
#pragma unroll
      for (int I = 0; I < 4; ++I) W[I][3] = wmfma(M[s >> 2][I][s & 3], a3, W[I][3]);
    }
    // ---- P3, tile (3, 3): rows / columns 52..54 = Quu[16..18, 16..18], column 51 = Qu[16..18] - lu (two accumulators)
    {
      v4d q1 = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < WKS; ++s) {
        if (WFOLD_SKIP(s)) continue;
        const double a3 = AOP(3, s);
        if (s & 1) q1 = wmfma(a3, W[s >> 2][3][s & 3], q1);
        else Q[3][3] = wmfma(a3, W[s >> 2][3][s & 3], Q[3][3]);
      }
      Q[3][3] += q1;
    }
    WSTAMP(3)
    // ---- Qu = lu + B^T Vx: column 51 of Qux~ (rows 0..15 in qux0[3], rows 16..18 in rows 52..54 of Q)
    if (lr == 3) {
#pragma unroll
      for (int r = 0; r < 4; ++r) qux0[3][r] += lu4[r];
      if (lk < 3) Q[3][3][1] += lu16;
    }
    // ---- Quu -> LDS, both triangles
    {
      const v4d quu = quu0 + quu1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ar = 4 * r + lk;
        L.QL[ar * WLDQ + lr] = quu[r] + ((ar == lr) ? (luu4[r] + lam) : 0.0);
        if (lr >= 4 && lr < 7) { const double v = qux0[3][r]; L.QL[ar * WLDQ + 12 + lr] = v; L.QL[(12 + lr) * WLDQ + ar] = v; }
      }
      if (lk < 3 && lr >= 4 && lr < 7) L.QL[(16 + lk) * WLDQ + 12 + lr] = Q[3][3][1] + ((lk == lr - 4) ? (luu16 + lam) : 0.0);
    }
    __syncthreads();
    WSTAMP(4)
    // ---- the rest of P1 (column tiles 0..2) and of P3 (tiles I >= J other than (3, 3)): the fp64 MFMA executes on the
    // vector ALU (it is not an XDL op), so nothing can be hidden behind it -- the factorisation below simply follows
#pragma unroll
    for (int s = 0; s < WKS; ++s) {
      if (WFOLD_SKIP(s)) continue;
      double aj[3];
#pragma unroll
      for (int J = 0; J < 3; ++J) aj[J] = AOP(J, s);
#pragma unroll
      for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int J = 0; J < 3; ++J) W[I][J] = wmfma(M[s >> 2][I][s & 3], aj[J], W[I][J]);
    }
    if (FOLD) {
      // identity part of P3: rows 8..23 of Q += rows 8..23 of W (rows 33..48 of W arrived folded)
#pragma unroll
      for (int r = 0; r < 2; ++r) { Q[0][0][2 + r] += W[0][0][2 + r]; Q[1][0][r] += W[1][0][r]; Q[1][1][r] += W[1][1][r]; }
    }
    // The factorisation of Quu (below) is a latency chain of 18 short steps on the same wave (pivot broadcast, rsqrt, scale,
    // column through LDS); Quu has been in LDS since before P1, so the sweep is woven into the MFMA of the rest of P3: the
    // 9 products of a k-step form a flat list, and every point of the sweep where the next instruction would wait (after
    // the rsqrt, after the column has been published and requested back, after the bulk update) is followed by the next
    // one or two of them.  Scheduling fences keep the compiler from regrouping either stream.
    int fail = 0;
    double v[WM];
    const int xl = lane - 32;
#ifndef WAVE_NO_INTERLEAVE
    {
      constexpr int NS = FOLD ? 9 : WKS, NMF = 9 * NS;
#define WACT(si) (FOLD ? ((si) < 2 ? (si) : (si) + 4) : (si))
#define WFENCE __builtin_amdgcn_sched_barrier(0)
      double ai[4], an[4];
#pragma unroll
      for (int I = 0; I < 4; ++I) an[I] = AOP(I, WACT(0));
      int nm = 0;
      auto mf = [&]() {
        if (nm < NMF) {
          const int si = nm / 9, q = nm % 9, ks = WACT(si);
          if (q == 0) {
#pragma unroll
            for (int I = 0; I < 4; ++I) ai[I] = an[I];
            if (si + 1 < NS) {
#pragma unroll
              for (int I = 0; I < 4; ++I) an[I] = AOP(I, WACT(si + 1));
            }
          }
          const int I = q < 1 ? 0 : q < 3 ? 1 : q < 6 ? 2 : 3, J = q - (I == 0 ? 0 : I == 1 ? 1 : I == 2 ? 3 : 6);
          Q[I][J] = wmfma(ai[I], W[ks >> 2][J][ks & 3], Q[I][J]);
          ++nm;
        }
      };
      CholSweep cs;
#pragma unroll
      for (int c = 0; c < WM; ++c) v[c] = (lane < m) ? L.QL[lane * WLDQ + c] : ((c == xl) ? 1.0 : 0.0);
      WFENCE; mf(); mf(); WFENCE;
      cs.begin(v, L.col, lane);
      WFENCE; mf(); mf(); WFENCE;
#pragma unroll
      for (int j = 0; j < WM - 1; ++j) {
        v[j + 1] = __builtin_fma(-cs.vj, cs.c1, v[j + 1]);
        const double piv = wbcast(v[j + 1], j + 1);
        if (!(piv > 0.0)) cs.fail = 1;
        const double x = piv > 0.0 ? piv : 1.0;
        const double y = __builtin_amdgcn_rsq(x);
        WFENCE; mf(); WFENCE;
        const double a = v[j + 1] * y, tt = x * y;
        const double e = __builtin_fma(-tt, y, 1.0);
        const double vn = __builtin_fma(0.5 * a, e, a);
        v[j + 1] = vn;
        double cn[WM];
        double c1n = 0.0;
        if (j + 2 < WM) {
          L.col[(j + 1) & 1][lane] = vn;
          c1n = wbcast(vn, j + 2);
#pragma unroll
          for (int k = j + 3; k < WM; ++k) cn[k] = L.col[(j + 1) & 1][k];
        }
        WFENCE; mf(); mf(); WFENCE;
#pragma unroll
        for (int k = j + 2; k < WM; ++k) v[k] = __builtin_fma(-cs.vj, cs.c[k], v[k]);
#pragma unroll
        for (int k = j + 3; k < WM; ++k) cs.c[k] = cn[k];
        cs.vj = vn;
        cs.c1 = c1n;
        WFENCE; mf(); if (j & 1) mf(); WFENCE;
      }
#pragma unroll
      for (int r = 0; r < NMF; ++r) mf();
      fail = cs.fail;
#undef WACT
#undef WFENCE
    }
#else
#pragma unroll
    for (int s = 0; s < WKS; ++s) {
      if (WFOLD_SKIP(s)) continue;
#pragma unroll
      for (int I = 0; I < 4; ++I) {
        const double ai = AOP(I, s);
#pragma unroll
        for (int J = 0; J <= I && J < 3; ++J) Q[I][J] = wmfma(ai, W[s >> 2][J][s & 3], Q[I][J]);
      }
    }
#endif
    WSTAMP(5)
    // ---- right-looking Cholesky Quu = L L^T fused with Linv = L^-1.  Lanes 0..18 hold a row of Quu, lanes 32..50 a
    // column of Linv, in the same 19 registers: step j scales register j by 1 / L[j][j] (-> L[i][j] on the row lanes,
    // Linv[j][c] on the column lanes) and every later register k loses v[j] * L[k][j] (one broadcast, one FMA for
    // both halves; the broadcasts of a step are independent of its FMAs, so they issue back to back).
    {
#ifdef WAVE_NO_INTERLEAVE
#pragma unroll
      for (int c = 0; c < WM; ++c) v[c] = (lane < m) ? L.QL[lane * WLDQ + c] : ((c == xl) ? 1.0 : 0.0);
      fail = chol_linv(v, L.col, lane);
#endif
#ifndef WAVE_NO_FALLBACK
      if (fail) {
        // ilqr.cpp:278-281: one retry with Quu + 1e-4 I
        if (lane < m) L.QL[lane * WLDQ + lane] += 1e-4;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < WM; ++c) v[c] = (lane < m) ? L.QL[lane * WLDQ + c] : ((c == xl) ? 1.0 : 0.0);
        fail = chol_linv(v, L.col, lane);
      }
#endif
      if (!fail) {
        // Linv over the Quu buffer (the row lanes have theirs in registers): QL[i][c] = Linv[i][c]; row 19, column 19 zero
        __syncthreads();
        if (xl >= 0 && xl < WLDQ) {
#pragma unroll
          for (int i = 0; i < WM; ++i) L.QL[i * WLDQ + xl] = (xl < m) ? v[i] : 0.0;
          L.QL[19 * WLDQ + xl] = 0.0;
        }
      }
    }
    __syncthreads();
    WSTAMP(6)
    double* Kg = S.K + ((size_t)b * N + t) * m * n;
    double* kg = S.kff + ((size_t)b * N + t) * m;
    if (fail) {
      // ---- indefinite Quu even after the bump: the reference's ldlt() is a pivoted factorisation that still solves the
      // system; here Quu^-1 explicitly, by Gauss-Jordan with partial pivoting spread over the lanes (lane = row x
      // one of three column groups; the pivot row is read at wave-uniform LDS addresses).  Quu^-1 then takes the place
      // of Linv in the buffer and the products below run in their "explicit inverse" form.
      gauss_jordan_inverse(L, lane);
    }
    // ---- every operand of this knot has left the A~ buffer: stage the next knot behind P6 / P7
#ifndef WAVE_SKIP_STAGE
    if (t > 0) {
      asm volatile("" ::: "memory");
      stage_A<FOLD>(L, Ag - n * n, Bg - n * m, lane);
      load_b0<FOLD>(b0, Bg - n * m, lk, lr);
    }
#endif
    {
      // ---- P6a: Y = Linv Qux~ (fallback: Z = Quu^-1 Qux~)   (k-steps 0..3: qux0, k-step 4: rows 52..55 of Q = Qux[16..18], 0)
      v4d y[2][4];
      double q16[4];
#pragma unroll
      for (int J = 0; J < 4; ++J) q16[J] = Q[3][J][1];
#pragma unroll
      for (int Ia = 0; Ia < 2; ++Ia)
#pragma unroll
        for (int J = 0; J < 4; ++J) y[Ia][J] = (v4d){0.0, 0.0, 0.0, 0.0};
      const int ra1 = lr < 3 ? 16 + lr : 19;      // rows 19.. of the padded Linv are zero
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const double la0 = L.QL[lr * WLDQ + 4 * s + lk];
        const double la1 = L.QL[ra1 * WLDQ + 4 * s + lk];
#pragma unroll
        for (int J = 0; J < 4; ++J) {
          const double qb = (s < 4) ? qux0[J][s & 3] : q16[J];
          y[0][J] = wmfma(la0, qb, y[0][J]);
          y[1][J] = wmfma(la1, qb, y[1][J]);
        }
      }
      WSTAMP(7)
      // ---- P6b: [K | k] = -Linv^T Y, straight to HBM (fallback: [K | k] = -Z, no second product)
      v4d kk[2][4];
      if (!fail) {
#pragma unroll
        for (int Ia = 0; Ia < 2; ++Ia)
#pragma unroll
          for (int J = 0; J < 4; ++J) kk[Ia][J] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          const int k = 4 * s + lk;
          const double lb0 = L.QL[k * WLDQ + lr];
          const double lb1 = (lr < 3) ? L.QL[k * WLDQ + 16 + lr] : 0.0;
#pragma unroll
          for (int J = 0; J < 4; ++J) {
            const double yb = y[s >> 2][J][s & 3];
            kk[0][J] = wmfma(lb0, yb, kk[0][J]);
            if (s == 4) kk[1][J] = wmfma(lb1, yb, kk[1][J]);      // Linv is lower triangular: Linv[k][16..18] = 0 for k < 16
          }
        }
      } else {
#pragma unroll
        for (int Ia = 0; Ia < 2; ++Ia)
#pragma unroll
          for (int J = 0; J < 4; ++J) kk[Ia][J] = y[Ia][J];
      }
      // rows 0..15: columns < 48 unconditionally, column tile 3 = K[:, 48..50] and k; rows 16..18: register 0
#ifdef WAVE_SKIP_KSTORE
      if (kk[0][0][0] == 123.456)
#endif
      {
        double* Kl = Kg + lk * n + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int J = 0; J < 3; ++J) Kl[4 * r * n + 16 * J] = -kk[0][J][r];
          double* p = (lr < 3) ? (Kl + 4 * r * n + 48) : (kg + 4 * r + lk);
          if (lr <= 3) *p = -kk[0][3][r];
        }
        if (lk < 3) {
#pragma unroll
          for (int J = 0; J < 3; ++J) Kl[16 * n + 16 * J] = -kk[1][J][0];
          double* p = (lr < 3) ? (Kl + 16 * n + 48) : (kg + 16 + lk);
          if (lr <= 3) *p = -kk[1][3][0];
        }
      }
      WSTAMP(8)
      // ---- P7: M = Q - Y^T Y, tiles I >= J   (fallback: M = Q - Qux~^T Z = Qxx + Qxu K, Vx = Qx + Qxu k: with the
      // explicit inverse the reference's long form ilqr.cpp:294-307 reduces to this; symmetric because Quu^-1 is)
      if (!fail) {
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
          for (int I = 0; I < 4; ++I) {
            const double ya = -y[s >> 2][I][s & 3];
#pragma unroll
            for (int J = 0; J <= I; ++J) Q[I][J] = wmfma(ya, y[s >> 2][J][s & 3], Q[I][J]);
          }
      } else {
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
          for (int I = 0; I < 4; ++I) {
            const double qa = -((s < 4) ? qux0[I][s & 3] : q16[I]);
#pragma unroll
            for (int J = 0; J <= I; ++J) Q[I][J] = wmfma(qa, y[s >> 2][J][s & 3], Q[I][J]);
          }
      }
    }
    WSTAMP(9)
    // ---- M <- Q; the strictly upper tiles are the transposes of the lower ones (through LDS: written in C layout,
    // read back with rows and columns exchanged), so M is exactly symmetric (ilqr.cpp:307)
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J) M[I][J] = Q[I][J];
    // strictly lower tiles in order: (1,0) (2,0) (2,1) | (3,0) (3,1) (3,2), three per round through L.T
#pragma unroll
    for (int round = 0; round < 2; ++round) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int I = round == 0 ? (q == 0 ? 1 : 2) : 3, J = round == 0 ? (q == 2 ? 1 : 0) : q;
#pragma unroll
        for (int r = 0; r < 4; ++r) L.T[q][(4 * r + lk) * WLDT + lr] = Q[I][J][r];
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int I = round == 0 ? (q == 0 ? 1 : 2) : 3, J = round == 0 ? (q == 2 ? 1 : 0) : q;
#pragma unroll
        for (int r = 0; r < 4; ++r) M[J][I][r] = L.T[q][lr * WLDT + 4 * r + lk];
      }
      __syncthreads();
    }
    WSTAMP(10)
  }
#ifdef WAVE_STAMP
  if (b == 0 && lane0 == 0) for (int k = 0; k < 16; ++k) S.J[k] = (double)ph[k];
#endif
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + 4 * r + (lane0 >> 4), j = 16 * J + (lane0 & 15);
        if (i < n && j < n) S.Vxx[(size_t)b * n * n + i * n + j] = M[I][J][r];
        else if (i == n && j < n) S.Vx[(size_t)b * n + j] = M[I][J][r];
      }
}

// fold_h: the step size h when A_t, B_t come from the analytic linearisation kernels (their hinge-position rows are then
// exactly e_k + h * the hinge-velocity rows, see fold_rows), 0 for Jacobians of any other origin (generic kernel)
void launch_backward_wave(const DevState& S, int mode, hipStream_t st, double fold_h, const int* list, const int* count) {
#ifdef ILQR_LEGACY_KERNELS      // (the folded variant on the standard layout, ILQR_BACKWARD=wave-fold: cross-check of riccati_pack.hip, test library only)
  if (fold_h != 0.0) { hipLaunchKernelGGL(k_backward_wave<true>, dim3(S.B), dim3(64), 0, st, S, mode, fold_h, list, count); return; }
#endif
  hipLaunchKernelGGL(k_backward_wave<false>, dim3(S.B), dim3(64), 0, st, S, mode, 0.0, list, count);
}

}  // namespace ilqr
