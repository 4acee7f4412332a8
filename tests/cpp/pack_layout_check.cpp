// CPU check of the operand layout's index maps (mpc-ilqr-mujoco_amd/csrc/riccati_pack.h): slot <-> state bijection, fold partners 32 slots
// apart, contracted rows in tiles 2 / 3, image indices inside their knot regions and disjoint.  Test harness, not product code.
#define __host__
#define __device__
#include <cstdio>
#include <set>
#include "../../mpc-ilqr-mujoco_amd/csrc/riccati_pack.h"
using namespace ilqr;
int main() {
  int fails = 0;
  auto chk = [&](bool ok, const char* what) { if (!ok) { std::printf("FAIL %s\n", what); ++fails; } };
  std::set<int> seen;
  for (int st = 0; st < 51; ++st) { const int s = pk_state_slot(st); chk(s >= 0 && s < 64 && pk_slot_state(s) == st, "state -> slot -> state"); seen.insert(s); }
  chk(seen.size() == 51, "slots of the 51 states are distinct");
  chk(pk_slot_state(PK_SLOT_AUG) == PK_AUG && !seen.count(PK_SLOT_AUG), "vector slot");
  for (int j = 0; j < 3; ++j) chk(pk_slot_state(60 + j) == PK_BU + j && !seen.count(60 + j), "control-column slots");
  int pads = 0; for (int s = 0; s < 64; ++s) pads += pk_slot_state(s) == PK_PAD;
  chk(pads == 64 - 51 - 1 - 3, "padding slots");
  // position p (base x, y, z = states 0..2; hinge angle j = state 7 + j) and its velocity (states 26..28; 32 + j): partner 32 slots later, same lane / register
  for (int p = 0; p < 3; ++p) chk(pk_state_slot(26 + p) == pk_state_slot(p) + 32 && pk_state_slot(p) < 22, "base position partners");
  for (int j = 0; j < 19; ++j) chk(pk_state_slot(32 + j) == pk_state_slot(7 + j) + 32 && pk_state_slot(7 + j) < 22, "hinge partners");
  // every contracted row (quaternion, all velocities) sits in tiles 2, 3
  for (int st = 3; st < 7; ++st) chk(pk_row_slot(pk_state_slot(st)), "quaternion rows contracted");
  for (int st = 26; st < 51; ++st) chk(pk_row_slot(pk_state_slot(st)), "velocity rows contracted");
  for (int st : {0, 1, 2}) chk(!pk_row_slot(pk_state_slot(st)), "position rows not stored");
  // images: indices distinct and inside the knot regions (+ 7 doubles of alignment slack)
  std::set<int> ia, il;
  for (int R = 32; R < 64; ++R) for (int C = 0; C < 64; ++C) { const int i = pk_a_index(R, C); chk(i >= 0 && i < PK_A_DOUBLES, "A~ index range"); ia.insert(i); }
  chk((int)ia.size() == PK_A_DOUBLES && PK_A_DOUBLES + 7 <= 51 * 51, "A~ image");
  for (int R = 0; R < 64; ++R) for (int C = 0; C <= R; ++C) if ((C >> 4) <= (R >> 4)) { const int i = pk_l_index(R, C); chk(i >= 0 && i < PK_L_DOUBLES, "lxx~ index range"); il.insert(i); }
  chk(PK_L_DOUBLES + 7 <= 51 * 51 && PK_B_DOUBLES + 7 <= 51 * 19, "images fit their regions");
  // tile (I, J) register r lane (lk, lr) <-> element (16 I + 4 r + lk, 16 J + lr)
  // (registers 0, 1 of the 64 lanes are the tile's first kilobyte, registers 2, 3 its second: pk_l_elem)
  chk(pk_l_index(16 * 3 + 4 * 2 + 1, 16 * 1 + 5) == pk_l_tile(3, 1) * 256 + 128 + (1 * 16 + 5) * 2 + 0, "accumulator image order");
  chk(pk_l_index(16 * 2 + 4 * 1 + 3, 16 * 2 + 9) == pk_l_tile(2, 2) * 256 + (3 * 16 + 9) * 2 + 1, "accumulator image order (first half)");
  {
    std::set<int> one;
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) { const int i = pk_l_elem(lane, r); chk(i >= 0 && i < 256, "tile element range"); one.insert(i); }
    chk(one.size() == 256, "tile elements distinct");
  }
  std::printf(fails ? "pack layout: %d failures\n" : "pack layout ok\n", fails);
  return fails ? 1 : 0;
}
