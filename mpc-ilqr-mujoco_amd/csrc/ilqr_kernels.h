// Device-state descriptor and kernel launchers shared by ilqr_kernels.hip and ilqr_capi.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace h1 {
struct ProblemDev;
struct DynParams;
}  // namespace h1

namespace ilqr {

enum { MASK_ALL = 0, MASK_ACTIVE = 1, MASK_RETRY = 2 };

// All pointers are device pointers; rollout-major, row-major inside.
struct DevState {
  int B, N, max_iter;
  double* x0;          // [B][51]
  double* xbar;        // [B][N+1][51]
  double* ubar;        // [B][N][19]
  double* xcand;       // [B][8][N+1][51]   line-search candidates
  double* ucand;       // [B][8][N][19]
  double* cand_cost;   // [B][8]
  double* cand_knot;   // [B][8][N+1]       per-knot costs of the line-search candidates (summed in knot order)
  double* A;           // [B][N][51][51]
  double* Bm;          // [B][N][51][19]
  double* lx;          // [B][N+1][51]
  double* lu;          // [B][N][19]
  double* lxx;         // [B][N+1][51][51]  (inside a solve with the one-wave Riccati kernel: knots t < N hold the 16 x 16 tiles I >= J only)
  double* luu;         // [B][N][19]        diagonal
  double* K;           // [B][N][19][51]
  double* kff;         // [B][N][19]
  double* lin_dump;    // [B][N][sizeof(KnotDump)/8] primal per-body quantities of every knot
  double* quad_rec;    // per-knot records of the cost quadratics (k_quad_kin -> k_cost_quadratics; quad_rec_doubles(B (N + 1)), four knots per line in runs of four fields)
  long quad_knot0;     // index of this view's first knot in quad_rec (batch slices share the handle's buffer)
  double* Vx;          // [B][51]           value gradient at knot 0
  double* Vxx;         // [B][51][51]
  double* J;           // [B] current cost
  double* Jbase;       // [B] cost of the nominal trajectory (line-search baseline)
  double* ls_cost;     // [B] stage API: cost after the line search
  double* lambda;      // [B]
  int* active;         // [B]
  int* need_retry;     // [B]
  int* iters;          // [B]
  int* improved;       // [B]
  int* alpha_idx;      // [B]
  double* trace_cost;  // [B][max_iter+1]
  double* trace_alpha; // [B][max_iter]
  double* trace_lambda;// [B][max_iter]
  // Compacted work lists, rebuilt on the device by k_control (null: not in use, e.g. batch slices): list (it, 0) = rollouts
  // active at the start of iteration it, list (it, 1) = rollouts that take the lambda retry of iteration it.  A launch whose
  // blocks are one rollout each (k_backward_wave) takes rollout order[...][blockIdx] for blockIdx < order_n[...]: the selected
  // rollouts then occupy the FIRST blocks of the grid and spread evenly over the shader engines (workgroups are dealt to them
  // round-robin by index: with a scattered selection one engine needs an extra 0.5 ms round of the one-wave-per-SIMD kernel).
  int* order;          // [2 (max_iter + 1)][B]
  int* order_n;        // [2 (max_iter + 1)]
  // Early continuation (ilqr_capi.hip enqueue_solve): the rollouts whose FIRST line search of iteration it - 1 accepted a step start
  // the linearisation / cost quadratics / re-rollout of iteration it while the others still take their lambda retry.  Group A =
  // the first order_an[it] entries of list (it, 0) (k_control phase 0 fills that list first; the count is snapshot after it), group R =
  // order_r[0 .. order_rn[it]) (filled by phase 1); grp_a / grp_r: the same two sets as per-rollout flags for the kernels that select
  // by mask (rollout, trajectory cost, adoption).  Null: not in use.
  int* grp_a;          // [B]
  int* grp_r;          // [B]
  int* order_r;        // [B]
  int* order_rn;       // [max_iter + 2]
  int* order_an;       // [max_iter + 2]
};

void launch_rollout(const DevState& S, const h1::ProblemDev& P, int mode, int do_roll, int count_iter, double* cost_out, hipStream_t st);
void launch_step(int count, const double* x, const double* u, const h1::DynParams& dyn, double* xn, hipStream_t st, int stance_l = 1, int stance_r = 1);
// pack != 0 (a solve whose backward pass is the operand-layout Riccati kernel): A_t, B_t in the layout of riccati_pack.h (the two-knot
// analytic kernels write it themselves, any other producer is followed by the conversion kernel)
struct WorkList;
void launch_linearize(const DevState& S, const h1::ProblemDev& P, int mode, int jac_mode, double eps, hipStream_t st, int phases = 3, int iter = -1, int pack = 0, const WorkList* wl = nullptr);
// lower = 1: knots t < N get only the tiles I >= J of lxx (what k_backward_wave loads); 2: every knot in the operand layout of
// riccati_pack.h (lx in row / column "aug"); the stage API always asks for the full matrix (0)
void launch_cost_quadratics(const DevState& S, const h1::ProblemDev& P, int mode, hipStream_t st, int iter = -1, int lower = 0, const WorkList* wl = nullptr);
// compacted list of the rollouts of a pass inside a solve (DevState::order), or nulls: MASK_ACTIVE at iteration iter -> list (iter, 0), MASK_RETRY -> (iter, 1)
struct WorkList { const int* list; const int* count; };
inline WorkList work_list(const DevState& S, int mode, int iter) {
  if (!S.order || iter < 0 || mode == MASK_ALL) return WorkList{nullptr, nullptr};
  const int slot = 2 * iter + (mode == MASK_RETRY ? 1 : 0);
  return WorkList{S.order + (size_t)slot * S.B, S.order_n + slot};
}
void launch_backward(const DevState& S, int mode, hipStream_t st, double fold_h = 0.0, int iter = -1);
// speculative lambda retry (ilqr_kernels.hip k_control_spec): T = the twin view whose K, kff, Vx, Vxx, candidates and lambda are its own
void launch_spec_lambda(const DevState& S, double* lambda2, hipStream_t st);
void launch_control_spec(const DevState& S, const DevState& T, int iter, double tol, int early_exit, hipStream_t st, int sum_knots, const int* gate = nullptr);
// Device-side choice between the two orders (the host's count of active rollouts is one iteration old): g[0] = n if n <= max else 0 (count of
// the speculative launches), g[1] = 0 / 1 (gate of the sequential bookkeeping), g[2] = 0 / n (count of the sequential first line search), with
// n = the length of list (iter, 0).  The launchers below take an explicit list / count (default kernel families only: spec_dual_available).
void launch_spec_gate(const DevState& S, int iter, int max, int* g, hipStream_t st);
bool spec_dual_available(const h1::ProblemDev& P);
void launch_backward_list(const DevState& S, hipStream_t st, double fold_h, const int* list, const int* count);
void launch_line_search_list(const DevState& S, const h1::ProblemDev& P, hipStream_t st, const int* list, const int* count, int max_rollouts);
double linearize_fold_h(const h1::ProblemDev& P, int jac_mode);
// max_rollouts: upper bound of the rollouts this pass can select (the batch, or -- with the early-exit gate -- the count of
// still-active rollouts the host saw two iterations ago): at most 1024 -> one rollout per wave in the two-lane line search
void launch_line_search(const DevState& S, const h1::ProblemDev& P, int mode, hipStream_t st, int iter = -1, int max_rollouts = -1);
void launch_control(const DevState& S, int phase, int iter, double tol, int early_exit, hipStream_t st, int sum_knots = 0, const int* gate = nullptr);
bool ls_costs_per_knot(const h1::ProblemDev& P);
void launch_solve_begin(const DevState& S, hipStream_t st);
void launch_adopt_rollout(const DevState& S, const double* shadow, int mode, unsigned long long* mismatches, hipStream_t st);
void launch_warm_shift(const DevState& S, const double* prev_x, const double* prev_u, hipStream_t st);
void launch_last_step(const DevState& S, const h1::ProblemDev& P, hipStream_t st);
void launch_compute_control(const DevState& S, const double* x_meas, double* u_out, hipStream_t st);
void launch_pack_first_knot(const DevState& S, double* u0, double* K0, hipStream_t st);
void launch_pack_payload(const DevState& S, int with_gains, double* out, hipStream_t st);
void launch_mirror_lxx(const DevState& S, hipStream_t st);   // fill the strictly upper tiles of lxx_t, t < N, from the lower ones
int backward_needs_lds_attr();
// kernel variants (ILQR_DYN / ILQR_ROLLOUT / ILQR_LS / ILQR_BACKWARD / ILQR_LINT): read from the environment ONCE per handle
// (ilqr_hip_create; read_variants) and installed for the calling host thread at the top of every C-ABI call (set_variants)
struct Variants { int scalar_dyn, rollout_split, ls_split, backward, fold, lin_one_knot; };
Variants read_variants();
void set_variants(const Variants& v);
int variants_supported(const Variants& v);      // 0: the environment selects a cross-check family this build does not hold (-DILQR_LEGACY_KERNELS)
int variant_ls_split();
int variant_rollout_split();
int variant_backward();
int variant_scalar_dyn();
int variant_lin_one_knot();
size_t backward_lds_bytes();
size_t lin_dump_doubles();
size_t quad_rec_doubles(size_t knots);
void launch_rollout_r(const DevState& S, const h1::ProblemDev& P, int mode, int do_roll, int count_iter, double* cost_out, hipStream_t st);
void launch_step_r(int count, const double* x, const double* u, const h1::DynParams& dyn, double* xn, hipStream_t st);
void launch_last_step_r(const DevState& S, const h1::ProblemDev& P, hipStream_t st);
void launch_cand_costs(const DevState& S, const h1::ProblemDev& P, int mode, hipStream_t st, bool with_sum = true, const int* gate = nullptr);
void launch_nominal_costs(const DevState& S, const h1::ProblemDev& P, int mode, double* cost_out, hipStream_t st);
void launch_line_search_r(const DevState& S, const h1::ProblemDev& P, int mode, hipStream_t st);
void launch_lin_primal_r(const DevState& S, const h1::ProblemDev& P, int mode, hipStream_t st);
int dyn_kernels_set_attr();
// dyn_split_kernels.hip: two lanes per rollout / candidate
void launch_rollout_s(const DevState& S, const h1::ProblemDev& P, int mode, int do_roll, int count_iter, double* cost_out, hipStream_t st);
void launch_lin_primal_s(const DevState& S, const h1::ProblemDev& P, int mode, hipStream_t st, const int* list = nullptr, const int* count = nullptr);
void launch_line_search_s(const DevState& S, const h1::ProblemDev& P, int mode, hipStream_t st, const int* list = nullptr, const int* count = nullptr, int max_rollouts = -1);
int dyn_split_kernels_set_attr();
void launch_step_s(int count, const double* x, const double* u, const h1::DynParams& dyn, double* xn, hipStream_t st, int stance_l, int stance_r);
void launch_last_step_s(const DevState& S, const h1::ProblemDev& P, hipStream_t st);
void launch_linearize_fd_s(const DevState& S, const h1::ProblemDev& P, int mode, double eps, hipStream_t st);
void launch_backward_mfma(const DevState& S, int mode, hipStream_t st);
int backward_mfma_set_attr();
void launch_backward_wave(const DevState& S, int mode, hipStream_t st, double fold_h, const int* list, const int* count);
size_t backward_mfma_lds_bytes();
// riccati_pack.hip: the one-wave kernel on the operand layout of riccati_pack.h and the conversions between that layout and the
// standard one (in place, per knot region)
int variant_pack();
void launch_backward_pack(const DevState& S, int mode, hipStream_t st, double fold_h, const int* list, const int* count);
void launch_pack_ab(const DevState& S, hipStream_t st, int mode = MASK_ALL, const int* list = nullptr, const int* count = nullptr);
void launch_pack_zero_pads(const DevState& S, hipStream_t st);
void launch_unpack_ab(const DevState& S, double h, hipStream_t st);
void launch_pack_lxx(const DevState& S, hipStream_t st);
void launch_unpack_lxx(const DevState& S, hipStream_t st);

}  // namespace ilqr
