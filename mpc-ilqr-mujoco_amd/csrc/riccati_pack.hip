// K4 (MFMA, one wave per rollout, operand layout): Riccati backward pass on the packed images of riccati_pack.h.
//
// Replaces iLQR::backwardPass (reference src/ilqr/ilqr.cpp:250-309) inside a solve with analytic Jacobians.  Same algebra and the
// same v_mfma_f64_16x16x4_f64 register chaining as riccati_wave.hip (C/D tile (I, J) register r, lane (lk, lr) = element
// (16 I + 4 r + lk, 16 J + lr) = B operand of k-step 4 I + r for column tile J = A operand of the transposed matrix for row tile
// J), in the slot order of riccati_pack.h:
//   Q     the value function of knot t + 1, augmented (row / column "aug" = Vx), ten tiles I >= J as P7 left them
//   M~    = F M F^T, M = sym(Q), F = I + h sum_p e_v(p) e_p^T over the 22 position slots p (partner v(p) = p + 32).  Only the two
//         bottom tile rows M~[2..3][0..3] (the contracted rows, 8 k-steps) and the unfolded block M[0..1][0..1] are ever used:
//         M~[2][J] = Q[2][J] + h M[0][J] is a register FMA, the upper tiles M[0][1], M[0][2], M[1][2], M[1][3] it needs are four
//         transposes through LDS, M~[2][3] = M~[3][2]^T a fifth.
//   P2    G0 = M~ B0 (rows 32..63 contracted)                                   32 MFMA
//   P4    Qux~[0..15, :] = G0[32..63, :]^T A~ + G0[0..31, :]^T (identity part)   32     column aug = Qu - lu, columns 60..62 = Quu[0..15, 16..18]
//   P5    Quu[0..15, 0..15] = B0^T G0[32..63, :]                                  8
//   P1    W = M~ A~ + M~[:, 0..31] (identity part), tiles W[2..3][0..3], W[0][0], W[1][0], W[1][1]    88
//   P3    Q = lxx~ + A~^T W[32..63, :] (+ W[0..31, :] on rows 0..31), tiles I >= J  80     rows 60..62 = Qux[16..18, :]
//   --    Quu -> LDS, Cholesky + L^-1 on the same wave (riccati_wave.hip CholSweep), indefinite branch by Gauss-Jordan
//   P6a   Y = L^-1 Qux~ (40), P6b [K | k] = -L^-T Y (24, sign by the MFMA's neg modifier), P7 Q <- Q - Y^T Y (50, tiles I >= J)
// 354 MFMA per knot (429 in riccati_wave.hip<FOLD>, 569 unfolded); the padding slots stay exact zeros from knot to knot (every
// product that could write them multiplies a zero column of A~ or a zero row of M), so no operand is ever masked.
#include <hip/hip_runtime.h>

#include "ilqr_kernels.h"
#include "riccati_pack.h"

namespace ilqr {

typedef double v4d __attribute__((ext_vector_type(4)));

#define PN 51
#define PM 19
#define PLDQ 20
#define PLDT 17
#define QI(I, J) ((I) * ((I) + 1) / 2 + (J))

#ifdef WAVE_STAMP
#define PSTAMP(k) { const long long tn_ = clock64(); ph[k] += tn_ - tlast; tlast = tn_; }
#else
#define PSTAMP(k)
#endif

// 31,744 B per wave
struct PackLds {
  double QL[64 * PLDQ];          // rows 0..19: Quu, then Linv = L^-1 (row 19 and column 19 zero); rows 32..50: identity (columns of Linv start there)
  double Id[20 * PLDQ];          // identity (row 19 zero): takes the place of Linv^T in P6b when QL holds Quu^-1 (indefinite branch)
  double col[2][64];             // columns of L on their way from the row lanes to every lane (double-buffered)
  double T[4][16 * PLDT];        // transposition buffers
  union {
    double Aop[32 * 64];         // A~ of the knot in operand order: tile (J, s) = 64 doubles [lk][lr] at (8 J + s) * 64
    struct {
      double Mx[PM * 40];        // fallback: augmented Gauss-Jordan matrix [Quu | I], pitch 40
      int perm[32];
    };
  };
};

template <int NEG = 0>
__device__ __forceinline__ v4d pmfma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, NEG); }
__device__ __forceinline__ double pbcast(double x, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wave_sync() { __syncthreads(); }     // one wave per workgroup: a wait on the LDS counter, no s_barrier

// v / sqrt(x) for a positive, normal x (riccati_wave.hip scale_rsqrt)
__device__ __forceinline__ double pscale_rsqrt(double v, double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double a = v * y, t = x * y;
  const double e = __builtin_fma(-t, y, 1.0);
  return __builtin_fma(0.5 * a, e, a);
}

// Right-looking Cholesky Quu = L L^T fused with Linv = L^-1 (riccati_wave.hip CholSweep): lanes 0..18 hold a row of Quu, lanes
// 32..50 a column of Linv, in the same 19 registers; returns 1 if a pivot is not positive (LLT failure, ilqr.cpp:278).
__device__ __forceinline__ int pchol_linv(double (&v)[PM], double (&col)[2][64], int lane) {
  // LLT fails when a pivot is not positive: the sweep just keeps the smallest pivot (one instruction per step instead of a compare
  // and three selects) and divides blindly -- a non-positive pivot is recorded before its NaN / infinity can reach a later one, and
  // what the sweep leaves in v is then discarded by the caller
  double c[PM];
  const double piv0 = pbcast(v[0], 0);
  double pmin = piv0;
  double vj = pscale_rsqrt(v[0], piv0);
  v[0] = vj;
  col[0][lane] = vj;
  double c1 = pbcast(vj, 1);
#pragma unroll
  for (int k = 2; k < PM; ++k) c[k] = col[0][k];
#pragma unroll
  for (int j = 0; j < PM - 1; ++j) {
    v[j + 1] = __builtin_fma(-vj, c1, v[j + 1]);
    const double piv = pbcast(v[j + 1], j + 1);
    pmin = __builtin_fmin(pmin, piv);
    const double vn = pscale_rsqrt(v[j + 1], piv);
    v[j + 1] = vn;
    double cn[PM];
    double c1n = 0.0;
    if (j + 2 < PM) {
      col[(j + 1) & 1][lane] = vn;
      c1n = pbcast(vn, j + 2);
#pragma unroll
      for (int k = j + 3; k < PM; ++k) cn[k] = col[(j + 1) & 1][k];
    }
#pragma unroll
    for (int k = j + 2; k < PM; ++k) { v[k] = __builtin_fma(-vj, c[k], v[k]); asm volatile("" : "+v"(v[k])); }
#pragma unroll
    for (int k = j + 3; k < PM; ++k) c[k] = cn[k];
    vj = vn;
    c1 = c1n;
  }
  return !(pmin > 0.0);
}

// Quu^-1 by Gauss-Jordan with partial pivoting (riccati_wave.hip gauss_jordan_inverse): Quu in L.QL, result back into L.QL as an
// operand buffer (row 19 and column 19 zero).  Scratch aliases the A~ staging area.
__device__ __forceinline__ void pgauss_jordan_inverse(PackLds& L, int lane) {
  constexpr int m = PM, ld = 40;
  double* Mx = L.Mx;
  int* perm = L.perm;
  for (int e = lane; e < m * ld; e += 64) {
    const int i = e / ld, j = e % ld;
    Mx[e] = (j < m) ? L.QL[i * PLDQ + j] : ((j - m == i) ? 1.0 : 0.0);
  }
  wave_sync();
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
    wave_sync();
    if (q < 3) {
#pragma unroll
      for (int kk = 0; kk < 13; ++kk) {
        const int k = q * 13 + kk;
        const double pk = Mx[p * ld + k], own = Mx[r * ld + k];
        Mx[r * ld + k] = (r == p) ? own * ip : own - f * pk;
      }
    }
    wave_sync();
  }
  for (int e = lane; e < 20 * PLDQ; e += 64) {
    const int c = e / PLDQ, j = e % PLDQ;
    L.QL[e] = (c < m && j < m) ? Mx[perm[c] * ld + m + j] : 0.0;
  }
  wave_sync();
}

// HBM -> LDS staging of the packed A~ of one knot without passing through registers (global_load_lds_dwordx4: the wave's 1 KB
// lands contiguously in lane order); B0 in B-operand layout into registers
__device__ __forceinline__ void pstage_A(PackLds& L, const double* Ap, int lane) {
  typedef const __attribute__((address_space(1))) void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;
  // (the instruction offset moves the HBM address and the LDS address alike: four instructions per base pair)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    __builtin_amdgcn_global_load_lds((gptr)(Ap + 512 * g + 2 * lane), (lptr)&L.Aop[512 * g], 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr)(Ap + 512 * g + 2 * lane), (lptr)&L.Aop[512 * g], 16, 1024, 0);
    __builtin_amdgcn_global_load_lds((gptr)(Ap + 512 * g + 2 * lane), (lptr)&L.Aop[512 * g], 16, 2048, 0);
    __builtin_amdgcn_global_load_lds((gptr)(Ap + 512 * g + 2 * lane), (lptr)&L.Aop[512 * g], 16, 3072, 0);
  }
}
__device__ __forceinline__ void pload_b0(double (&b0)[8], const double* Bp, int lane) {
#pragma unroll
  for (int s = 0; s < 8; ++s) b0[s] = Bp[64 * s + lane];
}
__device__ __forceinline__ void pload_q(v4d (&Q)[10], const double* Lp, int lane) {
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  const v2d_t* p = reinterpret_cast<const v2d_t*>(Lp) + lane;                          // (pk_l_elem: registers 0, 1 | registers 2, 3)
#pragma unroll
  for (int t = 0; t < 10; ++t) { const v2d_t a = p[128 * t], c = p[128 * t + 64]; Q[t] = (v4d){a.x, a.y, c.x, c.y}; }
}
__device__ __forceinline__ void put_tile(double* T, const v4d& q, int lk, int lr) {
#pragma unroll
  for (int r = 0; r < 4; ++r) T[(4 * r + lk) * PLDT + lr] = q[r];
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) k_backward_pack(DevState S, int mode, double fh, const int* list, const int* count) {
  int b = blockIdx.x;
  if (list) {                       // compacted selection (DevState::order): the first *count blocks take the listed rollouts
    if (b >= *count) return;
    b = list[b];
  } else {
    if (mode == MASK_ACTIVE && !S.active[b]) return;
    if (mode == MASK_RETRY && !(S.active[b] && S.need_retry[b])) return;
  }
  __shared__ PackLds L;
  const int lane0 = threadIdx.x;
  const int N = S.N;
  constexpr int n = PN, m = PM;
  const double lam = S.lambda[b];
  const double* lpk = S.lxx + (size_t)b * (N + 1) * n * n;

  // identity rows of the factorisation's start (lanes 32..50 = columns of Linv), zero rows for the lanes that hold nothing
  for (int e = lane0; e < 45 * PLDQ; e += 64) { const int r = 19 + e / PLDQ, c = e % PLDQ; L.QL[r * PLDQ + c] = (r >= 32 && r - 32 == c && c < m) ? 1.0 : 0.0; }
  for (int e = lane0; e < 20 * PLDQ; e += 64) { const int r = e / PLDQ, c = e % PLDQ; L.Id[e] = (r == c && r < m) ? 1.0 : 0.0; }

  v4d Q[10];
  pload_q(Q, pk_align(lpk + (size_t)N * n * n), lane0);
  double b0[8];
  pstage_A(L, pk_align(S.A + ((size_t)b * N + (N - 1)) * n * n), lane0);
  pload_b0(b0, pk_align(S.Bm + ((size_t)b * N + (N - 1)) * n * m), lane0);

#ifdef WAVE_STAMP
  long long ph[16] = {0}; long long tlast = clock64();
#endif
  for (int t = N - 1; t >= 0; --t) {
    // lane indices re-derived behind an opaque barrier every knot (keeps LICM from hoisting per-lane addresses and predicates of the
    // whole knot out of the loop, where they would be spilled)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int lr = lane & 15, lk = lane >> 4;
    const double* lug = S.lu + ((size_t)b * N + t) * m;
    const double* luug = S.luu + ((size_t)b * N + t) * m;
    const double h1r = (lk == 2) ? 0.0 : fh;       // register 1 of row tile 1: slots 20, 21 fold with h, the vector slot 22 does not (23: padding)
    const double hc1 = (lr == 6) ? 0.0 : fh;       // the same for column slots 16..31 -> 48..63
#define AOP(J, s) L.Aop[((J) * 8 + (s)) * 64 + lane]
    // A~ was staged into LDS during the previous knot (behind P6 / P7), B0 arrived in registers: nothing of this knot is in flight yet
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
    // ---- lxx~_t straight from HBM into the accumulators of P3 (issued now, needed after P1); lu_t, luu_t for the lanes that need them
    v4d Qn[10];
#ifdef PK_EXP_NOLXX
#pragma unroll
    for (int q = 0; q < 10; ++q) Qn[q] = (v4d){1.0, 0.0, 0.0, 1.0};
#else
    pload_q(Qn, pk_align(lpk + (size_t)t * n * n), lane);
#endif
    const double lu_c = lug[lr], lu_16 = lug[16 + (lk < 3 ? lk : 2)];
    double luu4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) luu4[r] = luug[4 * r + lk];
    const double luu16 = luug[16 + (lk < 3 ? lk : 2)];
    // ---- M = sym(Q), M~ = F M F^T: the two bottom tile rows
    put_tile(L.T[0], Q[QI(1, 0)], lk, lr); put_tile(L.T[1], Q[QI(2, 0)], lk, lr); put_tile(L.T[2], Q[QI(2, 1)], lk, lr); put_tile(L.T[3], Q[QI(3, 1)], lk, lr);
    wave_sync();
    v4d Mb[2][4];
    {
      v4d t10, t20; double t21[2], t31[2];
#pragma unroll
      for (int r = 0; r < 4; ++r) { t10[r] = L.T[0][lr * PLDT + 4 * r + lk]; t20[r] = L.T[1][lr * PLDT + 4 * r + lk]; }
#pragma unroll
      for (int r = 0; r < 2; ++r) { t21[r] = L.T[2][lr * PLDT + 4 * r + lk]; t31[r] = L.T[3][lr * PLDT + 4 * r + lk]; }
      Mb[0][0] = Q[QI(2, 0)] + fh * Q[QI(0, 0)];
      Mb[0][1] = Q[QI(2, 1)] + fh * t10;
      Mb[1][0] = Q[QI(3, 0)]; Mb[1][0][0] += fh * Q[QI(1, 0)][0]; Mb[1][0][1] += h1r * Q[QI(1, 0)][1];
      Mb[1][1] = Q[QI(3, 1)]; Mb[1][1][0] += fh * Q[QI(1, 1)][0]; Mb[1][1][1] += h1r * Q[QI(1, 1)][1];
      Mb[0][2] = Q[QI(2, 2)] + fh * t20 + fh * Mb[0][0];
      Mb[1][2] = Q[QI(3, 2)]; Mb[1][2][0] += fh * t21[0]; Mb[1][2][1] += h1r * t21[1]; Mb[1][2] += fh * Mb[1][0];
      Mb[1][3] = Q[QI(3, 3)]; Mb[1][3][0] += fh * t31[0]; Mb[1][3][1] += h1r * t31[1]; Mb[1][3] += hc1 * Mb[1][1];
      wave_sync();
      put_tile(L.T[0], Mb[1][2], lk, lr);
      wave_sync();
#pragma unroll
      for (int r = 0; r < 4; ++r) Mb[0][3][r] = L.T[0][lr * PLDT + 4 * r + lk];
    }
    // the unfolded block M[0..1][0..1] (identity part of P1): the accumulators of W start from it
    const v4d M00 = Q[QI(0, 0)], M10 = Q[QI(1, 0)], M11 = Q[QI(1, 1)];
    PSTAMP(0)
    // ---- P2: G0 = M~ B0
    v4d g0[4];
#pragma unroll
    for (int I = 0; I < 4; ++I) g0[I] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int I = 0; I < 4; ++I) g0[I] = pmfma(Mb[s >> 2][I][s & 3], b0[s], g0[I]);
    // Qu = lu + B^T Vx: row aug of G0 is B0^T Vx
    g0[1][1] += (lk == 2) ? lu_c : 0.0;
    // identity part of P4: Qux~[u][p] = G0[p][u] for the slots p < 32: the transposes of two tiles of G0
    put_tile(L.T[0], g0[0], lk, lr); put_tile(L.T[1], g0[1], lk, lr);
    wave_sync();
    v4d qux0[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { qux0[0][r] = L.T[0][lr * PLDT + 4 * r + lk]; qux0[1][r] = L.T[1][lr * PLDT + 4 * r + lk]; }
    qux0[2] = (v4d){0.0, 0.0, 0.0, 0.0}; qux0[3] = (v4d){0.0, 0.0, 0.0, 0.0};
    PSTAMP(1)
    // ---- P4: Qux~[0..15, :] += G0[32..63, :]^T A~
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int J = 0; J < 4; ++J) qux0[J] = pmfma(g0[2 + (s >> 2)][s & 3], AOP(J, s), qux0[J]);
    // ---- P5: Quu[0..15, 0..15] = B0^T G0 (two interleaved accumulators)
    v4d quu0 = (v4d){0.0, 0.0, 0.0, 0.0}, quu1 = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s & 1) quu1 = pmfma(b0[s], g0[2 + (s >> 2)][s & 3], quu1);
      else quu0 = pmfma(b0[s], g0[2 + (s >> 2)][s & 3], quu0);
    }
    PSTAMP(2)
    // ---- P1, column tile 3 first: W[2..3][3] = M~ A~[:, 48..63] (its columns 60..62 are G[:, 16..18])
    v4d Wb[2][4];
    Wb[0][3] = (v4d){0.0, 0.0, 0.0, 0.0}; Wb[1][3] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const double a3 = AOP(3, s);
      Wb[0][3] = pmfma(Mb[s >> 2][2][s & 3], a3, Wb[0][3]);
      Wb[1][3] = pmfma(Mb[s >> 2][3][s & 3], a3, Wb[1][3]);
    }
    // ---- P3, tile (3, 3): rows / columns 60..62 = Quu[16..18, 16..18] (two accumulators)
    {
      v4d q1 = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const double a3 = AOP(3, s);
        if (s & 1) q1 = pmfma(a3, Wb[s >> 2][3][s & 3], q1);
        else Qn[QI(3, 3)] = pmfma(a3, Wb[s >> 2][3][s & 3], Qn[QI(3, 3)]);
      }
      Qn[QI(3, 3)] += q1;
    }
    PSTAMP(3)
    // ---- Quu -> LDS, both triangles
    {
      const v4d quu = quu0 + quu1;
#pragma unroll
      for (int r = 0; r < 4; ++r) L.QL[(4 * r + lk) * PLDQ + lr] = quu[r] + ((4 * r + lk == lr) ? (luu4[r] + lam) : 0.0);
      if (lr >= 12 && lr < 15) {      // columns 60..62 of Qux~[0..15, :] and of rows 60..62 of Q: Quu[:, 16..18]
#pragma unroll
        for (int r = 0; r < 4; ++r) { const double v = qux0[3][r]; L.QL[(4 * r + lk) * PLDQ + 4 + lr] = v; L.QL[(4 + lr) * PLDQ + 4 * r + lk] = v; }
        if (lk < 3) L.QL[(16 + lk) * PLDQ + 4 + lr] = Qn[QI(3, 3)][3] + ((lk == lr - 12) ? (luu16 + lam) : 0.0);
      }
    }
    wave_sync();
    PSTAMP(4)
    // ---- the rest of P1: W[2..3][0..2] and W[0][0], W[1][0], W[1][1]; accumulators start from the identity part M~[:, 0..31]
    // (W[0][0], W[1][0], W[1][1] first: they are the last readers of M~[2..3][0..1] as an operand, whose registers then become the
    // accumulators of W[2..3][0..1] -- started from a copy, each of those four tiles cost eight register moves)
    v4d W00 = M00, W10 = M10, W11 = M11;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const double a0 = AOP(0, s), a1 = AOP(1, s);
      W00 = pmfma(Mb[s >> 2][0][s & 3], a0, W00);
      W10 = pmfma(Mb[s >> 2][1][s & 3], a0, W10);
      W11 = pmfma(Mb[s >> 2][1][s & 3], a1, W11);
    }
    Wb[0][0] = Mb[0][0]; Wb[0][1] = Mb[0][1]; Wb[1][0] = Mb[1][0]; Wb[1][1] = Mb[1][1];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      double aj[3];
#pragma unroll
      for (int J = 0; J < 3; ++J) aj[J] = AOP(J, s);
#pragma unroll
      for (int J = 0; J < 3; ++J) {
        if (s == 0 && J == 2) { Wb[0][2] = pmfma(Mb[0][2][0], aj[2], (v4d){0.0, 0.0, 0.0, 0.0}); Wb[1][2] = pmfma(Mb[0][3][0], aj[2], (v4d){0.0, 0.0, 0.0, 0.0}); }
        else { Wb[0][J] = pmfma(Mb[s >> 2][2][s & 3], aj[J], Wb[0][J]); Wb[1][J] = pmfma(Mb[s >> 2][3][s & 3], aj[J], Wb[1][J]); }
      }
    }
    PSTAMP(5)
    // ---- the rest of P3 (tiles I >= J other than (3, 3)); rows 0..31 get their identity part W[0..31, :] first (an add behind the
    // last product would wait for the MFMA pipeline to drain)
    Qn[QI(0, 0)] += W00; Qn[QI(1, 0)] += W10; Qn[QI(1, 1)] += W11;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      double ai[4];
#pragma unroll
      for (int I = 0; I < 4; ++I) ai[I] = AOP(I, s);
#pragma unroll
      for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int J = 0; J <= I && J < 3; ++J) Qn[QI(I, J)] = pmfma(ai[I], Wb[s >> 2][J][s & 3], Qn[QI(I, J)]);
    }
    // Qu[16..18] = lu[16..18] + B[:, 16..18]^T Vx: rows 60..62 of column aug
    Qn[QI(3, 1)][3] += (lr == 6 && lk < 3) ? lu_16 : 0.0;
    PSTAMP(6)
    // ---- Cholesky Quu = L L^T fused with Linv = L^-1
    int fail;
    double v[PM];
    const int xl = lane - 32;
    {
#pragma unroll
      for (int c = 0; c < PM; ++c) v[c] = L.QL[lane * PLDQ + c];
#ifdef PK_EXP_NOCHOL
      fail = 0;
#else
      fail = pchol_linv(v, L.col, lane);
#endif
      if (fail) {
        // ilqr.cpp:278-281: one retry with Quu + 1e-4 I
        if (lane < m) L.QL[lane * PLDQ + lane] += 1e-4;
        wave_sync();
#pragma unroll
        for (int c = 0; c < PM; ++c) v[c] = L.QL[lane * PLDQ + c];
        fail = pchol_linv(v, L.col, lane);
      }
      if (!fail) {
        // Linv over the Quu buffer: QL[i][c] = Linv[i][c]; row 19, column 19 zero
        wave_sync();
        if (xl >= 0 && xl < PLDQ) {
#pragma unroll
          for (int i = 0; i < PM; ++i) L.QL[i * PLDQ + xl] = (xl < m) ? v[i] : 0.0;
          L.QL[19 * PLDQ + xl] = 0.0;
        }
      }
    }
    wave_sync();
    PSTAMP(7)
    double* Kg = S.K + ((size_t)b * N + t) * m * n;
    double* kg = S.kff + ((size_t)b * N + t) * m;
    if (fail) {
      // indefinite Quu even after the bump: Quu^-1 explicitly (the reference's ldlt() still solves the system); it takes the place of
      // Linv and the products below run in their "explicit inverse" form
      pgauss_jordan_inverse(L, lane);
    }
    // ---- every operand of this knot has left the A~ buffer: stage the next knot behind P6 / P7
#ifndef PK_EXP_NOSTAGE
    if (t > 0)
#else
    if (t > 100)
#endif
    {
      asm volatile("" ::: "memory");
      pstage_A(L, pk_align(S.A + ((size_t)b * N + (t - 1)) * n * n), lane);
      pload_b0(b0, pk_align(S.Bm + ((size_t)b * N + (t - 1)) * n * m), lane);
    }
    {
      // ---- P6a: Y = Linv Qux~ (indefinite branch: Z = Quu^-1 Qux~)   (k-steps 0..3: qux0, k-step 4: rows 60..63 of Q = Qux[16..18] and
      // a row that meets the zero column 19 of Linv)
      v4d y[2][4];
      double q16[4];
      q16[0] = Qn[QI(3, 0)][3]; q16[1] = Qn[QI(3, 1)][3]; q16[2] = Qn[QI(3, 2)][3]; q16[3] = Qn[QI(3, 3)][3];
      const int ra1 = lr < 3 ? 16 + lr : 19;      // rows 19.. of the padded Linv are zero
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const double la0 = L.QL[lr * PLDQ + 4 * s + lk];
        const double la1 = L.QL[ra1 * PLDQ + 4 * s + lk];
#pragma unroll
        for (int J = 0; J < 4; ++J) {
          const double qb = (s < 4) ? qux0[J][s & 3] : q16[J];
          if (s == 0) { y[0][J] = pmfma(la0, qb, (v4d){0.0, 0.0, 0.0, 0.0}); y[1][J] = pmfma(la1, qb, (v4d){0.0, 0.0, 0.0, 0.0}); }
          else { y[0][J] = pmfma(la0, qb, y[0][J]); y[1][J] = pmfma(la1, qb, y[1][J]); }
        }
      }
      PSTAMP(8)
      // ---- P6b: [K | k] = -Linv^T Y, straight to HBM.  No branch around the products (a second definition of kk or of P7's operands
      // costs the common path a copy of every accumulator at the merge): in the indefinite branch the identity stands in for Linv^T,
      // [K | k] = -I Z
      v4d kk[2][4];
      const double* lbp = fail ? L.Id : L.QL;
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int k = 4 * s + lk;
        const double lb0 = lbp[k * PLDQ + lr];
#pragma unroll
        for (int J = 0; J < 4; ++J) {
          const double yb = y[s >> 2][J][s & 3];
          if (s == 0) kk[0][J] = pmfma<1>(lb0, yb, (v4d){0.0, 0.0, 0.0, 0.0});
          else kk[0][J] = pmfma<1>(lb0, yb, kk[0][J]);
        }
        if (s == 4) {      // Linv is lower triangular: Linv[k][16..18] = 0 for k < 16
          const double lb1 = lbp[k * PLDQ + (lr < 3 ? 16 + lr : 19)];
#pragma unroll
          for (int J = 0; J < 4; ++J) kk[1][J] = pmfma<1>(lb1, y[1][J][0], (v4d){0.0, 0.0, 0.0, 0.0});
        }
      }
      // K[u][state of the column slot]: tiles 0 and 2 are whole runs of states (7 + lr, 32 + lr), tiles 1 and 3 mix; the vector slot
      // (tile 1, lr = 6) is k
#ifdef PK_EXP_NOKSTORE
      if (kk[0][0][0] == 123.456)
#endif
      {
        double* K0 = Kg + lk * n + 7 + lr;
        double* K2 = Kg + lk * n + 32 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) { K0[4 * r * n] = kk[0][0][r]; K2[4 * r * n] = kk[0][2][r]; }
        if (lk < 3) { K0[16 * n] = kk[1][0][0]; K2[16 * n] = kk[1][2][0]; }
        const int st1 = lr < 3 ? 23 + lr : lr - 3;                                     // lr < 6
        const int st3 = lr < 3 ? 48 + lr : lr < 6 ? 23 + lr : lr < 10 ? lr - 3 : lr < 12 ? 19 + lr : 31;    // lr < 12 or lr == 15
        if (lr < 6) {
          double* K1 = Kg + lk * n + st1;
#pragma unroll
          for (int r = 0; r < 4; ++r) K1[4 * r * n] = kk[0][1][r];
          if (lk < 3) K1[16 * n] = kk[1][1][0];
        }
        if (lr == 6) {
#pragma unroll
          for (int r = 0; r < 4; ++r) kg[4 * r + lk] = kk[0][1][r];
          if (lk < 3) kg[16 + lk] = kk[1][1][0];
        }
        if (lr < 12 || lr == 15) {
          double* K3 = Kg + lk * n + st3;
#pragma unroll
          for (int r = 0; r < 4; ++r) K3[4 * r * n] = kk[0][3][r];
          if (lk < 3) K3[16 * n] = kk[1][3][0];
        }
      }
      PSTAMP(9)
      // ---- P7: Q = Q + Qux~^T [K | k], tiles I >= J (= Q - Y^T Y; the reference's long form ilqr.cpp:294-307 reduces to this for the
      // gains it has just solved for; symmetric because Quu^-1 is).  Rows 16..19 of [K | k]: register 0 of kk[1] (row 19: zero)
#pragma unroll
      for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int I = 0; I < 4; ++I) {
          const double qa = (s < 4) ? qux0[I][s & 3] : q16[I];
#pragma unroll
          for (int J = 0; J <= I; ++J) Qn[QI(I, J)] = pmfma(qa, (s < 4) ? kk[0][J][s & 3] : kk[1][J][0], Qn[QI(I, J)]);
        }
    }
    PSTAMP(10)
#pragma unroll
    for (int q = 0; q < 10; ++q) Q[q] = Qn[q];
  }
#ifdef WAVE_STAMP
  if (b == 0 && lane0 == 0) for (int k = 0; k < 16; ++k) S.J[k] = (double)ph[k];
#endif
  // ---- Vxx, Vx of knot 0 in state order: the tiles I >= J and their mirror images (symmetric, ilqr.cpp:307)
  {
    const int lr = lane0 & 15, lk = lane0 >> 4;
    double* Vxx = S.Vxx + (size_t)b * n * n;
    double* Vx = S.Vx + (size_t)b * n;
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int si = pk_slot_state(16 * I + 4 * r + lk), sj = pk_slot_state(16 * J + lr);
          const double val = Q[QI(I, J)][r];
          if (si < n && sj < n) { Vxx[si * n + sj] = val; if (I != J) Vxx[sj * n + si] = val; }
          else if (si == PK_AUG && sj < n) Vx[sj] = val;
          else if (sj == PK_AUG && si < n && I > 1) Vx[si] = val;     // (column aug below the aug row: the states of tiles 2, 3)
        }
  }
}

// ------------------------------------------------------------------ layout conversions (stage API, getters; one workgroup per knot)
// standard -> packed (in place: the knot's region is read whole into LDS first).  The 22 position rows are dropped: the caller
// guarantees they are e_p + h * velocity row (analytic Jacobians).
__global__ void __launch_bounds__(256) k_pack_ab(DevState S, int mode, const int* list, const int* count) {
  __shared__ double sa[PN * PN];
  __shared__ double sb[PN * PM];
  const int bs = (int)(blockIdx.x / (unsigned)S.N), t = (int)(blockIdx.x % (unsigned)S.N);
  int b = bs;
  if (list) { if (bs >= *count) return; b = list[bs]; }
  else if ((mode == MASK_ACTIVE && !S.active[b]) || (mode == MASK_RETRY && !(S.active[b] && S.need_retry[b]))) return;
  const size_t knot = (size_t)b * S.N + t;
  double* Ag = S.A + knot * PN * PN;
  double* Bg = S.Bm + knot * PN * PM;
  for (int e = threadIdx.x; e < PN * PN; e += 256) sa[e] = Ag[e];
  for (int e = threadIdx.x; e < PN * PM; e += 256) sb[e] = Bg[e];
  __syncthreads();
  for (int e = threadIdx.x; e < PK_A_DOUBLES; e += 256) {
    const int J = e >> 9, R = 32 + ((e >> 4) & 31), C = 16 * J + (e & 15);
    const int sr = pk_slot_state(R), sc = pk_slot_state(C);
    double v = 0.0;
    if (sr < PN) { if (sc < PN) v = sa[sr * PN + sc]; else if (sc >= PK_BU && sc < PK_BU + 3) v = sb[sr * PM + 16 + (sc - PK_BU)]; }
    pk_align(Ag)[e] = v;
  }
  for (int e = threadIdx.x; e < PK_B_DOUBLES; e += 256) {
    const int R = 32 + (e >> 4), u = e & 15;
    const int sr = pk_slot_state(R);
    pk_align(Bg)[e] = sr < PN ? sb[sr * PM + u] : 0.0;
  }
}
// the entries of the packed A~ / B0 images the tangent kernels never write: column slots 22..31 (vector slot, padding), row slots 60..62
// (the control-column slots have no row)
__global__ void __launch_bounds__(256) k_pack_zero_pads(DevState S) {
  const size_t knot = blockIdx.x;
  double* Ap = pk_align(S.A + knot * PN * PN);
  double* Bp = pk_align(S.Bm + knot * PN * PM);
  for (int e = threadIdx.x; e < 32 * 10; e += 256) { const int R = 32 + e / 10, C = 22 + e % 10; Ap[pk_a_index(R, C)] = 0.0; }
  for (int e = threadIdx.x; e < 3 * 64; e += 256) { const int R = 60 + e / 64, C = e % 64; Ap[pk_a_index(R, C)] = 0.0; }
  for (int e = threadIdx.x; e < 3 * 16; e += 256) { const int R = 60 + e / 16, u = e % 16; Bp[pk_b_index(R, u)] = 0.0; }
}
// packed -> standard; position rows rebuilt as lin_column writes them: e_p + h * velocity row (B: h * velocity row)
__global__ void __launch_bounds__(256) k_unpack_ab(DevState S, double h) {
  __shared__ double pa[PK_A_DOUBLES];
  __shared__ double pb[PK_B_DOUBLES];
  const size_t knot = blockIdx.x;
  double* Ag = S.A + knot * PN * PN;
  double* Bg = S.Bm + knot * PN * PM;
  for (int e = threadIdx.x; e < PK_A_DOUBLES; e += 256) pa[e] = pk_align(Ag)[e];
  for (int e = threadIdx.x; e < PK_B_DOUBLES; e += 256) pb[e] = pk_align(Bg)[e];
  __syncthreads();
  for (int e = threadIdx.x; e < PN * PN; e += 256) {
    const int r = e / PN, c = e % PN;
    const int R = pk_state_slot(r), C = pk_state_slot(c);
    Ag[e] = R >= 32 ? pa[pk_a_index(R, C)] : ((r == c) ? 1.0 : 0.0) + h * pa[pk_a_index(R + 32, C)];
  }
  for (int e = threadIdx.x; e < PN * PM; e += 256) {
    const int r = e / PM, u = e % PM;
    const int R = pk_state_slot(r);
    const int Rr = R >= 32 ? R : R + 32;
    const double v = u < 16 ? pb[pk_b_index(Rr, u)] : pa[pk_a_index(Rr, 60 + (u - 16))];
    Bg[e] = R >= 32 ? v : 0.0 + h * v;
  }
}
// standard (whole matrix, or at least the entries on and below the diagonal of every 16 x 16 tile row -- `lower` images qualify only
// through k_mirror_lxx first) + S.lx -> packed
__global__ void __launch_bounds__(256) k_pack_lxx(DevState S) {
  __shared__ double sh[PN * PN];
  const size_t knot = blockIdx.x;
  double* Hg = S.lxx + knot * PN * PN;
  const double* lx = S.lx + knot * PN;
  for (int e = threadIdx.x; e < PN * PN; e += 256) sh[e] = Hg[e];
  __syncthreads();
  for (int e = threadIdx.x; e < PK_L_DOUBLES; e += 256) {
    const int tile = e >> 8, lane = (e >> 1) & 63, r = 2 * ((e >> 7) & 1) + (e & 1);       // (pk_l_elem)
    int I = 0; while (pk_l_tile(I + 1, 0) <= tile) ++I;
    const int J = tile - pk_l_tile(I, 0);
    const int sr = pk_slot_state(16 * I + 4 * r + (lane >> 4)), sc = pk_slot_state(16 * J + (lane & 15));
    double v = 0.0;
    if (sr < PN && sc < PN) v = sh[sr * PN + sc];
    else if (sr == PK_AUG && sc < PN) v = lx[sc];
    else if (sc == PK_AUG && sr < PN) v = lx[sr];
    pk_align(Hg)[e] = v;
  }
}
__global__ void __launch_bounds__(256) k_unpack_lxx(DevState S) {
  __shared__ double pl[PK_L_DOUBLES];
  const size_t knot = blockIdx.x;
  double* Hg = S.lxx + knot * PN * PN;
  for (int e = threadIdx.x; e < PK_L_DOUBLES; e += 256) pl[e] = pk_align(Hg)[e];
  __syncthreads();
  for (int e = threadIdx.x; e < PN * PN; e += 256) {
    const int r = e / PN, c = e % PN;
    const int R = pk_state_slot(r), C = pk_state_slot(c);
    const bool low = (R >> 4) > (C >> 4) || ((R >> 4) == (C >> 4));
    Hg[e] = low ? pl[pk_l_index(R, C)] : pl[pk_l_index(C, R)];
  }
}

void launch_backward_pack(const DevState& S, int mode, hipStream_t st, double fold_h, const int* list, const int* count) {
  hipLaunchKernelGGL(k_backward_pack, dim3(S.B), dim3(64), 0, st, S, mode, fold_h, list, count);
}
void launch_pack_ab(const DevState& S, hipStream_t st, int mode, const int* list, const int* count) { hipLaunchKernelGGL(k_pack_ab, dim3((unsigned)((size_t)S.B * S.N)), dim3(256), 0, st, S, mode, list, count); }
void launch_pack_zero_pads(const DevState& S, hipStream_t st) { hipLaunchKernelGGL(k_pack_zero_pads, dim3((unsigned)((size_t)S.B * S.N)), dim3(256), 0, st, S); }
void launch_unpack_ab(const DevState& S, double h, hipStream_t st) { hipLaunchKernelGGL(k_unpack_ab, dim3((unsigned)((size_t)S.B * S.N)), dim3(256), 0, st, S, h); }
void launch_pack_lxx(const DevState& S, hipStream_t st) { hipLaunchKernelGGL(k_pack_lxx, dim3((unsigned)((size_t)S.B * (S.N + 1))), dim3(256), 0, st, S); }
void launch_unpack_lxx(const DevState& S, hipStream_t st) { hipLaunchKernelGGL(k_unpack_lxx, dim3((unsigned)((size_t)S.B * (S.N + 1))), dim3(256), 0, st, S); }

}  // namespace ilqr
