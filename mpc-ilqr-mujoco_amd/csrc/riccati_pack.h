// Operand ("packed") layout of the Riccati inputs inside a solve with analytic Jacobians (riccati_pack.hip).
//
// The one-wave Riccati kernel works on 64 x 64 tiles of v_mfma_f64_16x16x4_f64; which state sits in which of the 64 slots is free.
// The semi-implicit Euler step makes every position row of the analytic Jacobians a copy of its velocity row through the integrator
// (h1_linearize_dev.h lin_column: A_t[p][:] = e_p^T + h A_t[v(p)][:], B_t[p][:] = h B_t[v(p)][:] for the three base positions and
// the nineteen hinge angles), so those 22 rows never have to be contracted: fold h X[:, p] into X[:, v(p)] once and add the identity
// part (riccati_wave.hip fold_rows did this for the sixteen rows that happened to fill whole k-steps of the state order).  The slot
// order below puts
//   * the 22 position slots in tiles 0 and 1 with their velocity partner exactly 32 slots later (same lane, same register, two
//     tiles down: the fold is a plain register FMA, no lane exchange),
//   * every row that IS contracted (hinge rates, base velocity, quaternion: 29 rows + the three zero rows that carry the last three
//     control columns) in tiles 2 and 3: 8 k-steps instead of 13 (9 folded), and only the two bottom tile rows of W = M A~ are ever
//     an operand -- 11 of its 16 tiles are computed,
//   * the vector slot (Vx, Qx: "aug") among the positions with fold factor 0: its row of A~ is the identity part alone.
//   slot  0..15  hinge angles 0..15        (state 7..22)      partner 32..47  hinge rates 0..15   (state 32..47)
//        16..18  hinge angles 16..18       (state 23..25)             48..50  hinge rates 16..18  (state 48..50)
//        19..21  base position             (state 0..2)               51..53  base linear velocity (state 26..28)
//        22      aug (Vx / lx / k)                                    54..57  quaternion           (state 3..6)
//        23..31  zero padding                                         58, 59  base angular velocity x, y (state 29, 30)
//                                                                     60..62  columns 16..18 of B_t (rows: zero)
//                                                                     63      base angular velocity z (state 31)
// HBM images, written by the producers in exactly the order the kernel consumes them (all loads affine, unmasked, whole 512-byte
// / 1 KB runs per wave instruction; padding is written as true zeros by the producers):
//   S.A  knot region (51 x 51 doubles): AP[J][s][lk][lr], J < 4, s < 8  = A~[32 + 4 s + lk][16 J + lr]      2048 doubles
//   S.Bm knot region (51 x 19 doubles): BP[s][lk][u],  u < 16           = B_t[state(32 + 4 s + lk)][u]        512 doubles
//   S.lxx knot region (51 x 51 doubles): LP[tile(I, J)][lane][r], I >= J = lxx~[16 I + 4 r + lk][16 J + lr]  2560 doubles
//        (lxx~: lxx in slot order with lx in row and column "aug"; lane = 16 lk + lr)
// Every image starts at the first 64-byte boundary inside its knot region (the regions are 8-byte aligned: 51 x 51 and 51 x 19 are
// odd); the spare doubles of the smallest region (2601 - 2560 = 41) cover the at most seven skipped ones.
#pragma once
#include <cstdint>

namespace ilqr {

template <class T>
__host__ __device__ inline T* pk_align(T* region) { return reinterpret_cast<T*>((reinterpret_cast<uintptr_t>(region) + 63) & ~static_cast<uintptr_t>(63)); }

constexpr int PK_AUG = 51;       // pk_slot_state of the vector slot
constexpr int PK_BU = 52;        // + j: column 16 + j of B_t
constexpr int PK_PAD = 55;
constexpr int PK_SLOT_AUG = 22;
constexpr int PK_A_DOUBLES = 2048, PK_B_DOUBLES = 512, PK_L_DOUBLES = 2560;

__host__ __device__ constexpr int pk_slot_state(int s) {
  return s < 19 ? 7 + s : s < 22 ? s - 19 : s == 22 ? PK_AUG : s < 32 ? PK_PAD : s < 51 ? s : s < 54 ? s - 25 : s < 58 ? s - 51 : s < 60 ? s - 29 : s < 63 ? PK_BU + (s - 60) : 31;
}
__host__ __device__ constexpr int pk_state_slot(int st) {
  return st < 3 ? 19 + st : st < 7 ? 51 + st : st < 26 ? st - 7 : st < 29 ? st + 25 : st < 31 ? st + 29 : st == 31 ? 63 : st;
}
// does the slot's row of A_t exist in the packed image (rows 32..63 minus the three control-column slots)?
__host__ __device__ constexpr bool pk_row_slot(int s) { return s >= 32 && !(s >= 60 && s < 63); }
__host__ __device__ constexpr int pk_a_index(int R, int C) { return (C >> 4) * 512 + (R - 32) * 16 + (C & 15); }
__host__ __device__ constexpr int pk_b_index(int R, int u) { return (R - 32) * 16 + u; }
__host__ __device__ constexpr int pk_l_tile(int I, int J) { return I * (I + 1) / 2 + J; }
// element (row 4 r + lk, column lr) of a tile: register r of lane 16 lk + lr in the MFMA accumulator layout.  Registers 0, 1 of the 64 lanes
// form the tile's first kilobyte, registers 2, 3 its second: a lane's 32 bytes leave (and arrive) as two 16-byte accesses, and with the four
// registers of a lane side by side each of the two instructions touched every other 16-byte chunk of the tile -- every 64-byte segment
// written twice, half at a time (twice the write requests of the cost-quadratics kernel, its address unit 40 % of the time waiting for L2)
__host__ __device__ constexpr int pk_l_elem(int lane, int r) { return (r >> 1) * 128 + lane * 2 + (r & 1); }
__host__ __device__ constexpr int pk_l_index(int R, int C) { return pk_l_tile(R >> 4, C >> 4) * 256 + pk_l_elem((R & 3) * 16 + (C & 15), (R >> 2) & 3); }

}  // namespace ilqr
