// C-ABI of the batched HIP iLQR (include/ilqr_hip.h): context, device buffers, problem data and the
// host-side orchestration of iLQR::solve (reference src/ilqr/ilqr.cpp:521-660) as a fixed stream of
// kernel launches with per-rollout masks on the device -- no host round trip inside a solve.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is opened lazily (ilqr_hip_comm_*), never linked
#include <dlfcn.h>
#include <cstdlib>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ilqr_hip.h"
#include "h1_cost_dev.h"
#include "h1_host_model.h"
#include "ilqr_kernels.h"

using ilqr::DevState;

// Environment switches (diagnostics, tests): ONE pass over the environment when a handle is created, kept in the handle; no getenv on
// the call path.  ILQR_ENV_PER_CALL=1 (itself read at creation) restores the re-read at the top of every C-ABI call -- the test
// suite and the tools that switch kernel families within one process set it (tests/conftest.py); ilqr_hip_reload_environment
// re-reads on demand.
struct Knobs {
  ilqr::Variants var;
  int dedup_retry;      // ILQR_DEDUP_RETRY (-1: the handle's own setting, ilqr_hip_set_dedup_saturated_retry)
  int slices, stagger, overlap_rollout, reuse_rollout, ee_gate /* -1: the handle's own setting */, split /* -1: on with the convergence exit */, spec, spec_dual, spec_max;
  bool per_call;
};
static Knobs read_knobs() {
  Knobs k;
  auto geti = [](const char* n, int d) { const char* e = getenv(n); return e ? atoi(e) : d; };
  k.var = ilqr::read_variants();
#ifndef SLICES_DEFAULT
#define SLICES_DEFAULT 1
#endif
  k.slices = geti("ILQR_SLICES", SLICES_DEFAULT);
  k.stagger = geti("ILQR_STAGGER", 1);
  k.dedup_retry = geti("ILQR_DEDUP_RETRY", -1);
  k.overlap_rollout = geti("ILQR_OVERLAP_ROLLOUT", 1);
  { const char* e = getenv("ILQR_REUSE_ROLLOUT"); k.reuse_rollout = (e && e[0] == '1') ? 1 : 0; }
  k.ee_gate = geti("ILQR_EE_GATE", -1);
  k.split = geti("ILQR_SPLIT", -1);
  k.spec = geti("ILQR_SPEC", 1);
  k.spec_dual = geti("ILQR_SPEC_DUAL", 1);
  k.spec_max = geti("ILQR_SPEC_MAX", 512);
  k.per_call = geti("ILQR_ENV_PER_CALL", 0) != 0;
  return k;
}
struct ilqr_hip_ctx {
  int device = 0, B = 0, N = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;          // cost quadratics run here, concurrently with the linearisation
  hipStream_t stream3 = nullptr;          // nominal re-rollout of iterations >= 1, concurrently with both
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_roll = nullptr;
  hipEvent_t ev_lin = nullptr, ev_adopt = nullptr;   // adoption of the re-rolled trajectory beside the backward pass (enqueue_solve)
  double* d_shadowx = nullptr;            // [B][N+1][51] target of the concurrent re-rollout
  DevState S{};
  h1::ProblemDev P{};
  // reference sets on the device
  double *d_xref = nullptr, *d_uref = nullptr, *d_comref = nullptr, *d_eeref = nullptr, *d_comvelref = nullptr;
  int* d_stance = nullptr;
  int n_xref = 0, n_stance = 0, n_ee = 0;
  // scratch
  double *d_tmpx = nullptr, *d_tmpu = nullptr, *d_prevx = nullptr, *d_prevu = nullptr, *d_u0 = nullptr, *d_K0 = nullptr, *d_cost_tmp = nullptr;
  double *d_stepx = nullptr, *d_stepu = nullptr, *d_stepn = nullptr;   // ilqr_hip_step* scratch, grown on demand
  size_t step_cap = 0;
  unsigned long long* d_mismatch = nullptr;
  // multi-GPU: RCCL communicator of this handle's rank and the packed first-knot payload it sends
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
  double* d_payload = nullptr;
  size_t payload_cap = 0;   // elements in which a concurrent re-rollout differed from the trajectory it replaced
  int max_iter = 10;
  double tol = 1e-4;
  int jac_mode = ILQR_JAC_ANALYTIC;
  double fd_eps = 1e-5;
  int early_exit = 1;
  int ee_gate = 1;            // early-exit gate (ilqr_hip_set_early_exit_gate): solve_async waits for iteration i - 2 before it enqueues iteration i
  bool initialized = false, refs_set = false;
  // early-exit gate: the number of rollouts still active after each iteration travels to this pinned array behind the
  // iteration; the host stays one iteration ahead of the device and stops enqueuing once the batch has converged
  int* h_active = nullptr;
  std::vector<hipEvent_t> ev_active;
  int iterations_enqueued = 0;
  // S.xbar is a rollout of (S.x0, S.ubar) by `rolled_variant` under `rolled_dyn` (the cold start): iteration 0 of the next solve
  // may then re-roll it beside the linearisation like every later iteration (enqueue_solve); cleared by anything that changes
  // the trajectory or x0 without a rollout
  bool xbar_rolled = false, first_aside = false;
  int rolled_variant = -1;
  h1::DynParams rolled_dyn{};
  // Layouts the last writer left behind (riccati_pack.h).  lxx: 0 whole matrices, 1 the tiles I >= J of the knots t < N only (getters
  // mirror them), 2 operand layout; ab_packed: S.A / S.Bm in the operand layout; ab_pads_clean: the slots of those images that the
  // tangent kernels never write are zeros (true after k_pack_zero_pads or a whole-batch k_pack_ab, false once anything wrote the
  // standard layout over them).  The stage API and the getters convert on demand (in place, per knot region).
  int lxx_layout = 0;
  bool ab_packed = false, ab_pads_clean = false;
  int dedup_retry = 0;          // ilqr_hip_set_dedup_saturated_retry
  bool env_refused = false;     // ILQR_ENV_PER_CALL: the last re-read selected a family this library does not hold (enter)
  double packed_h = 0.0;        // step size h of the packed image while ab_packed (k_unpack_ab rebuilds the position rows from it)
  Knobs knobs = read_knobs();   // (constructed in ilqr_hip_create)
  double lin_fold_h = 0.0;   // step size h while S.A / S.Bm hold the analytic Jacobians (folded backward kernel), else 0
  std::string err;
  // profiling
  int profiling = 0;
  unsigned prof_mask = 0xFFu;             // stages that get event pairs while profiling is on (ilqr_hip_set_profiled_stages)
  struct Span { int stage; hipEvent_t a, b; };
  std::vector<Span> spans;
  std::vector<hipEvent_t> pool;
  size_t pool_next = 0;
  double stage_ms[8] = {0}, stage_launches[8] = {0};
  // batch slices: the solve of each contiguous slice of the batch is enqueued on its own pair of streams, so that the
  // latency-bound stages of one slice (line search: 8 waves per 64 rollouts, nominal rollout) overlap with the
  // throughput-bound stages of the others (ILQR_SLICES, default SLICES_DEFAULT; 1 = whole batch on one stream pair)
  struct Slice { hipStream_t st = nullptr, st2 = nullptr, st3 = nullptr; hipEvent_t fork = nullptr, join = nullptr, done = nullptr, lead = nullptr, roll = nullptr; };
  std::vector<Slice> slices;
  hipEvent_t ev_begin = nullptr;
  int n_slices = 1;
  // speculative lambda retry (ilqr_kernels.hip k_control_spec): the twin view's own buffers, allocated by the first solve that can use them
  DevState T{};
  bool twin = false, twin_failed = false;
  hipEvent_t ev_spec_fork = nullptr, ev_spec_join = nullptr;
  int spec_iterations = 0;    // iterations of the last solve that enqueued both passes side by side
  int* d_spec_gate = nullptr; // [4] device-side choice of the order (launch_spec_gate)
  // early continuation (enqueue_solve): streams / events of the group that starts the next iteration behind the first control pass
  hipStream_t a1 = nullptr;     // (its cost quadratics and re-rollout share streams 2 and 3 with the other group, see enqueue_solve)
  hipEvent_t evA_fork = nullptr, evA_join = nullptr, evA_roll = nullptr, evA_lin = nullptr, evA_adopt = nullptr;
  int split_iterations = 0;   // iterations of the last solve whose concurrent region ran in two groups
};

#define HIPCHK(ctx, call)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
      return ILQR_ERR_HIP;                                                                  \
    }                                                                                       \
  } while (0)

template <class T> static int dalloc(ilqr_hip_ctx* c, T** p, size_t count) {
  HIPCHK(c, hipMalloc((void**)p, count * sizeof(T)));
  HIPCHK(c, hipMemsetAsync(*p, 0, count * sizeof(T), c->stream));
  return ILQR_OK;
}
#define TRY(x) do { int r_ = (x); if (r_ != ILQR_OK) return r_; } while (0)

static hipEvent_t next_event(ilqr_hip_ctx* c) {
  if (c->pool_next == c->pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { c->err = "hipEventCreate failed (profiling events)"; return nullptr; }
    c->pool.push_back(e);
  }
  return c->pool[c->pool_next++];
}
struct StageTimer {
  ilqr_hip_ctx* c; int stage; hipStream_t st; hipEvent_t a{}, b{};
  StageTimer(ilqr_hip_ctx* c_, int s, hipStream_t st_ = nullptr) : c(c_), stage(s), st(st_ ? st_ : c_->stream) { if (c->profiling && ((c->prof_mask >> s) & 1u)) { a = next_event(c); b = next_event(c); if (a && b) hipEventRecord(a, st); } }
  ~StageTimer() { if (a && b) { hipEventRecord(b, st); c->spans.push_back({stage, a, b}); } }
};

static inline void enter(ilqr_hip_ctx* c) {
  hipSetDevice(c->device);
  if (c->knobs.per_call) {
    // an unsupported selection keeps the previous one AND is recorded: the calls that launch kernels refuse (ENV_REFUSED) until the
    // environment selects a family this library holds again -- a test that switches families on a product handle must not pass vacuously
    const Knobs k = read_knobs();
    if (ilqr::variants_supported(k.var)) { c->knobs = k; c->env_refused = false; }
    else { c->env_refused = true; c->err = "ILQR_ENV_PER_CALL: the environment selects a kernel family this library does not hold (see ilqr_hip_create)"; }
  }
  ilqr::set_variants(c->knobs.var);
}
#define ENV_REFUSED(c) do { if ((c)->env_refused) return ILQR_ERR_UNSUPPORTED; } while (0)

extern "C" {

int ilqr_hip_create(ilqr_hip_ctx** out, int device, int batch, int horizon, double dt) {
  if (!out || batch <= 0 || horizon <= 0 || !(dt > 0.0)) return ILQR_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return ILQR_ERR_NO_DEVICE;
  ilqr_hip_ctx* c = new ilqr_hip_ctx();
  c->device = device; c->B = batch; c->N = horizon;
  if (!ilqr::variants_supported(c->knobs.var)) {
    // (the handle is returned so that ilqr_hip_last_error can say why; the caller destroys it)
    c->err = "the environment selects a cross-check kernel family (ILQR_BACKWARD / ILQR_LS / ILQR_ROLLOUT / ILQR_DYN / ILQR_LINT) that this library does not hold: "
             "they are compiled into the test library only (make ../lib/libilqr_hip_legacy.so, -DILQR_LEGACY_KERNELS)";
    *out = c;
    return ILQR_ERR_UNSUPPORTED;
  }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess || hipStreamCreate(&c->stream2) != hipSuccess || hipStreamCreate(&c->stream3) != hipSuccess || hipEventCreateWithFlags(&c->ev_roll, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_lin, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_adopt, hipEventDisableTiming) != hipSuccess) { delete c; return ILQR_ERR_NO_DEVICE; }
  const size_t B = batch, N = horizon, n = ILQR_NX, m = ILQR_NU;
  DevState& S = c->S;
  S.B = batch; S.N = horizon; S.max_iter = c->max_iter;
  int rc = ILQR_OK;
  auto A = [&](int r) { if (rc == ILQR_OK) rc = r; };
  A(dalloc(c, &S.x0, B * n)); A(dalloc(c, &S.xbar, B * (N + 1) * n)); A(dalloc(c, &S.ubar, B * N * m));
  A(dalloc(c, &S.xcand, B * 8 * (N + 1) * n)); A(dalloc(c, &S.ucand, B * 8 * N * m)); A(dalloc(c, &S.cand_cost, B * 8)); A(dalloc(c, &S.cand_knot, B * 8 * (N + 1)));
  A(dalloc(c, &S.A, B * N * n * n + 32)); A(dalloc(c, &S.Bm, B * N * n * m + 32));   // slack: riccati_wave.hip stages 16-byte pairs that may straddle the end of the last row
  A(dalloc(c, &S.lx, B * (N + 1) * n)); A(dalloc(c, &S.lu, B * N * m)); A(dalloc(c, &S.lxx, B * (N + 1) * n * n)); A(dalloc(c, &S.luu, B * N * m));
  A(dalloc(c, &S.lin_dump, B * N * ilqr::lin_dump_doubles()));
  A(dalloc(c, &S.quad_rec, ilqr::quad_rec_doubles(B * (N + 1)))); S.quad_knot0 = 0;
  A(dalloc(c, &S.K, B * N * m * n + 32)); /* slack: the line search stages K_t in 16-byte pairs, the last one ends one double past the knot */ A(dalloc(c, &S.kff, B * N * m)); A(dalloc(c, &S.Vx, B * n)); A(dalloc(c, &S.Vxx, B * n * n));
  A(dalloc(c, &S.J, B)); A(dalloc(c, &S.Jbase, B)); A(dalloc(c, &S.ls_cost, B)); A(dalloc(c, &S.lambda, B));
  A(dalloc(c, &S.active, B)); A(dalloc(c, &S.need_retry, B)); A(dalloc(c, &S.iters, B)); A(dalloc(c, &S.improved, B)); A(dalloc(c, &S.alpha_idx, B));
  A(dalloc(c, &S.trace_cost, B * (c->max_iter + 1))); A(dalloc(c, &S.trace_alpha, B * c->max_iter)); A(dalloc(c, &S.trace_lambda, B * c->max_iter));
  A(dalloc(c, &S.order, B * 2 * (c->max_iter + 1))); A(dalloc(c, &S.order_n, 2 * (size_t)(c->max_iter + 1)));
  A(dalloc(c, &S.grp_a, B)); A(dalloc(c, &S.grp_r, B)); A(dalloc(c, &S.order_r, B)); A(dalloc(c, &S.order_rn, (size_t)c->max_iter + 2)); A(dalloc(c, &S.order_an, (size_t)c->max_iter + 2));
  if (rc == ILQR_OK && (hipStreamCreate(&c->a1) != hipSuccess ||
      hipEventCreateWithFlags(&c->evA_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->evA_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->evA_roll, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->evA_lin, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->evA_adopt, hipEventDisableTiming) != hipSuccess)) { c->err = "stream / event creation failed (early continuation)"; rc = ILQR_ERR_HIP; }
  A(dalloc(c, &c->d_tmpx, B * (N + 1) * n)); A(dalloc(c, &c->d_tmpu, B * N * m));
  A(dalloc(c, &c->d_prevx, B * (N + 1) * n)); A(dalloc(c, &c->d_prevu, B * N * m)); A(dalloc(c, &c->d_shadowx, B * (N + 1) * n));
  A(dalloc(c, &c->d_u0, B * m)); A(dalloc(c, &c->d_K0, B * m * n)); A(dalloc(c, &c->d_cost_tmp, B)); A(dalloc(c, &c->d_mismatch, 1));
  // shared reference sets sized for per-rollout use
  A(dalloc(c, &c->d_xref, B * (N + 1) * n)); A(dalloc(c, &c->d_uref, B * N * m)); A(dalloc(c, &c->d_comref, B * (N + 1) * 3));
  A(dalloc(c, &c->d_eeref, B * (N + 1) * 6)); A(dalloc(c, &c->d_comvelref, B * (N + 1) * 3)); A(dalloc(c, &c->d_stance, B * (N + 1) * 2));
  if (rc != ILQR_OK) { *out = c; return rc; }
  if (ilqr::backward_needs_lds_attr() != 0) { c->err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"; *out = c; return ILQR_ERR_HIP; }
  h1::ProblemDev& P = c->P;
  P.N = horizon; P.dyn.h = dt; P.dyn.g[0] = 0; P.dyn.g[1] = 0; P.dyn.g[2] = -9.81; P.dyn.contact = 0; P.dyn.soft = 1e-5; P.dyn.mu = 1.0; P.dyn.limits = 0; P.dyn.lim_k = 0.0;
  for (int i = 0; i < ILQR_NX; ++i) { P.Q[i] = 1.0; P.Qf[i] = 1.0; }
  for (int i = 0; i < ILQR_NU; ++i) P.R[i] = 1.0;
  P.w_com = P.w_com_vel = P.w_ee_pos = P.w_ee_vel = P.w_upright = P.w_balance = 0.0;
  P.w_joint = 500.0; P.w_ctrl = 1000.0;  // RobotUtils ctor defaults, robot_utils.cpp:10
  P.x_ref = c->d_xref; P.u_ref = c->d_uref; P.com_ref = c->d_comref; P.stance = c->d_stance; P.ee_ref = c->d_eeref; P.com_vel_ref = c->d_comvelref;
  P.x_ref_stride = P.u_ref_stride = P.com_ref_stride = P.stance_stride = P.ee_ref_stride = P.com_vel_ref_stride = 0;
  // default: stance everywhere (RobotUtils::isStance default, robot_utils.cpp:494-504), lambda = 1e-6 (ilqr.cpp:16)
  std::vector<int> ones((N + 1) * 2, 1);
  hipMemcpyAsync(c->d_stance, ones.data(), ones.size() * sizeof(int), hipMemcpyHostToDevice, c->stream);
  std::vector<double> lam(B, 1e-6);
  hipMemcpyAsync(S.lambda, lam.data(), B * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (hipStreamSynchronize(c->stream) != hipSuccess) { c->err = "stream sync failed in create"; *out = c; return ILQR_ERR_HIP; }
  if (getenv("ILQR_DEBUG_PTRS")) {   // diagnostic: device address ranges, to place the address of a reported memory access fault
    auto pr = [&](const char* name, const void* p, size_t bytes) { std::fprintf(stderr, "[ilqr_hip] %-10s %p .. %p\n", name, p, (const void*)((const char*)p + bytes)); };
    pr("x0", S.x0, B * n * 8); pr("xbar", S.xbar, B * (N + 1) * n * 8); pr("ubar", S.ubar, B * N * m * 8);
    pr("xcand", S.xcand, B * 8 * (N + 1) * n * 8); pr("ucand", S.ucand, B * 8 * N * m * 8); pr("K", S.K, (B * N * m * n + 32) * 8); pr("kff", S.kff, B * N * m * 8);
    pr("A", S.A, (B * N * n * n + 32) * 8); pr("Bm", S.Bm, (B * N * n * m + 32) * 8); pr("lxx", S.lxx, B * (N + 1) * n * n * 8); pr("lin_dump", S.lin_dump, B * N * ilqr::lin_dump_doubles() * 8);
    pr("active", S.active, B * 4); pr("cand_knot", S.cand_knot, B * 8 * (N + 1) * 8);
  }
  *out = c;
  return ILQR_OK;
}

int ilqr_hip_destroy(ilqr_hip_ctx* c) {
  if (!c) return ILQR_ERR_ARG;
  enter(c);
  DevState& S = c->S;
  void* ptrs[] = {S.cand_knot, S.lin_dump, S.quad_rec, S.x0, S.xbar, S.ubar, S.xcand, S.ucand, S.cand_cost, S.A, S.Bm, S.lx, S.lu, S.lxx, S.luu, S.K, S.kff, S.Vx, S.Vxx, S.J, S.Jbase, S.ls_cost,
                  S.lambda, S.active, S.need_retry, S.iters, S.improved, S.alpha_idx, S.trace_cost, S.trace_alpha, S.trace_lambda, S.order, S.order_n, c->d_tmpx, c->d_tmpu,
                  c->d_prevx, c->d_prevu, c->d_shadowx, c->d_u0, c->d_K0, c->d_cost_tmp, c->d_stepx, c->d_stepu, c->d_stepn, c->d_mismatch, c->d_payload, c->d_xref, c->d_uref, c->d_comref, c->d_eeref, c->d_comvelref, c->d_stance};
  for (void* p : ptrs) if (p) hipFree(p);
  if (c->twin) { void* tw[] = {c->T.K, c->T.kff, c->T.Vx, c->T.Vxx, c->T.xcand, c->T.ucand, c->T.cand_cost, c->T.cand_knot, c->T.lambda, c->d_spec_gate}; for (void* p : tw) if (p) hipFree(p); }
  { void* gp[] = {S.grp_a, S.grp_r, S.order_r, S.order_rn, S.order_an}; for (void* p : gp) if (p) hipFree(p); }
  for (hipEvent_t e : {c->evA_fork, c->evA_join, c->evA_roll, c->evA_lin, c->evA_adopt}) if (e) hipEventDestroy(e);
  if (c->a1) hipStreamDestroy(c->a1);
  if (c->ev_spec_fork) hipEventDestroy(c->ev_spec_fork);
  if (c->ev_spec_join) hipEventDestroy(c->ev_spec_join);
  if (c->comm) ilqr_hip_comm_destroy(c);
  for (hipEvent_t e : c->pool) hipEventDestroy(e);
  for (hipEvent_t e : c->ev_active) hipEventDestroy(e);
  if (c->h_active) (void)hipHostFree(c->h_active);
  for (auto& sl : c->slices) { hipEventDestroy(sl.fork); hipEventDestroy(sl.join); hipEventDestroy(sl.done); hipEventDestroy(sl.lead); hipEventDestroy(sl.roll); hipStreamDestroy(sl.st3); hipStreamDestroy(sl.st2); hipStreamDestroy(sl.st); }
  if (c->ev_begin) hipEventDestroy(c->ev_begin);
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  if (c->ev_join) hipEventDestroy(c->ev_join);
  if (c->ev_roll) hipEventDestroy(c->ev_roll);
  if (c->ev_lin) hipEventDestroy(c->ev_lin);
  if (c->ev_adopt) hipEventDestroy(c->ev_adopt);
  if (c->stream3) hipStreamDestroy(c->stream3);
  if (c->stream2) hipStreamDestroy(c->stream2);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return ILQR_OK;
}
const char* ilqr_hip_last_error(const ilqr_hip_ctx* c) { return c ? c->err.c_str() : "null context"; }
int ilqr_hip_batch(const ilqr_hip_ctx* c) { return c ? c->B : -1; }
int ilqr_hip_horizon(const ilqr_hip_ctx* c) { return c ? c->N : -1; }
void* ilqr_hip_stream(ilqr_hip_ctx* c) { return c ? (void*)c->stream : nullptr; }

int ilqr_hip_set_cost_weights(ilqr_hip_ctx* c, const double* Q, const double* R, const double* Qf) {
  if (!c || !Q || !R || !Qf) return ILQR_ERR_ARG;
  std::memcpy(c->P.Q, Q, sizeof(c->P.Q)); std::memcpy(c->P.R, R, sizeof(c->P.R)); std::memcpy(c->P.Qf, Qf, sizeof(c->P.Qf));
  return ILQR_OK;
}
int ilqr_hip_set_task_weights(ilqr_hip_ctx* c, double w_com, double w_com_vel, double w_ee_pos, double w_ee_vel, double w_upright, double w_balance) {
  if (!c) return ILQR_ERR_ARG;
  c->P.w_com = w_com; c->P.w_com_vel = w_com_vel; c->P.w_ee_pos = w_ee_pos; c->P.w_ee_vel = w_ee_vel; c->P.w_upright = w_upright; c->P.w_balance = w_balance;
  return ILQR_OK;
}
int ilqr_hip_set_constraint_weights(ilqr_hip_ctx* c, double wj, double wc) { if (!c) return ILQR_ERR_ARG; c->P.w_joint = wj; c->P.w_ctrl = wc; return ILQR_OK; }
int ilqr_hip_set_gravity(ilqr_hip_ctx* c, double gx, double gy, double gz) { if (!c) return ILQR_ERR_ARG; c->P.dyn.g[0] = gx; c->P.dyn.g[1] = gy; c->P.dyn.g[2] = gz; return ILQR_OK; }

static int check_sets(const ilqr_hip_ctx* c, int n_sets) { return (n_sets == 1 || n_sets == c->B) ? ILQR_OK : ILQR_ERR_ARG; }

int ilqr_hip_set_contact_schedule(ilqr_hip_ctx* c, const int* stance, int n_sets) {
  if (!c || !stance || check_sets(c, n_sets)) return ILQR_ERR_ARG;
  enter(c);
  const size_t per = (size_t)(c->N + 1) * 2;
  HIPCHK(c, hipMemcpyAsync(c->d_stance, stance, per * n_sets * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->P.stance_stride = n_sets == 1 ? 0 : (long)per;
  c->xbar_rolled = false;      // (contact mode: the schedule is part of the dynamics)
  return ILQR_OK;
}
int ilqr_hip_set_ee_references(ilqr_hip_ctx* c, const double* ee_ref, const double* com_vel_ref, int n_sets) {
  if (!c || !ee_ref || check_sets(c, n_sets)) return ILQR_ERR_ARG;
  enter(c);
  const size_t per = (size_t)(c->N + 1);
  HIPCHK(c, hipMemcpyAsync(c->d_eeref, ee_ref, per * 6 * n_sets * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (com_vel_ref) HIPCHK(c, hipMemcpyAsync(c->d_comvelref, com_vel_ref, per * 3 * n_sets * sizeof(double), hipMemcpyHostToDevice, c->stream));
  else HIPCHK(c, hipMemsetAsync(c->d_comvelref, 0, per * 3 * n_sets * sizeof(double), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->P.ee_ref_stride = n_sets == 1 ? 0 : (long)(per * 6);
  c->P.com_vel_ref_stride = n_sets == 1 ? 0 : (long)(per * 3);
  return ILQR_OK;
}
int ilqr_hip_set_references(ilqr_hip_ctx* c, const double* x_ref, const double* u_ref, const double* com_ref, int n_sets) {
  if (!c || !x_ref || !u_ref || !com_ref || check_sets(c, n_sets)) return ILQR_ERR_ARG;
  enter(c);
  const size_t N = c->N;
  HIPCHK(c, hipMemcpyAsync(c->d_xref, x_ref, (N + 1) * ILQR_NX * n_sets * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_uref, u_ref, N * ILQR_NU * n_sets * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_comref, com_ref, (N + 1) * 3 * n_sets * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->P.x_ref_stride = n_sets == 1 ? 0 : (long)((N + 1) * ILQR_NX);
  c->P.u_ref_stride = n_sets == 1 ? 0 : (long)(N * ILQR_NU);
  c->P.com_ref_stride = n_sets == 1 ? 0 : (long)((N + 1) * 3);
  c->refs_set = true;
  return ILQR_OK;
}

int ilqr_hip_set_regularization(ilqr_hip_ctx* c, double lambda) {
  if (!c) return ILQR_ERR_ARG;
  enter(c);
  std::vector<double> lam(c->B, lambda);
  HIPCHK(c, hipMemcpyAsync(c->S.lambda, lam.data(), c->B * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ILQR_OK;
}
int ilqr_hip_set_max_iterations(ilqr_hip_ctx* c, int max_iter) {
  if (!c || max_iter <= 0) return ILQR_ERR_ARG;
  enter(c);
  if (max_iter != c->max_iter) {
    hipFree(c->S.trace_cost); hipFree(c->S.trace_alpha); hipFree(c->S.trace_lambda); hipFree(c->S.order); hipFree(c->S.order_n);
    c->S.trace_cost = c->S.trace_alpha = c->S.trace_lambda = nullptr; c->S.order = c->S.order_n = nullptr;
    c->max_iter = max_iter; c->S.max_iter = max_iter;
    TRY(dalloc(c, &c->S.trace_cost, (size_t)c->B * (max_iter + 1))); TRY(dalloc(c, &c->S.trace_alpha, (size_t)c->B * max_iter)); TRY(dalloc(c, &c->S.trace_lambda, (size_t)c->B * max_iter));
    TRY(dalloc(c, &c->S.order, (size_t)c->B * 2 * (max_iter + 1))); TRY(dalloc(c, &c->S.order_n, 2 * (size_t)(max_iter + 1)));
    if (c->S.order_rn) hipFree(c->S.order_rn);
    if (c->S.order_an) hipFree(c->S.order_an);
    c->S.order_rn = c->S.order_an = nullptr;
    TRY(dalloc(c, &c->S.order_rn, (size_t)max_iter + 2)); TRY(dalloc(c, &c->S.order_an, (size_t)max_iter + 2));
  }
  return ILQR_OK;
}
int ilqr_hip_set_tolerance(ilqr_hip_ctx* c, double tol) { if (!c) return ILQR_ERR_ARG; c->tol = tol; return ILQR_OK; }
int ilqr_hip_set_options(ilqr_hip_ctx* c, int jacobian_mode, double fd_eps, int early_exit) {
  if (!c || (jacobian_mode != ILQR_JAC_ANALYTIC && jacobian_mode != ILQR_JAC_FD_FORWARD) || !(fd_eps > 0.0)) return ILQR_ERR_ARG;
  c->jac_mode = jacobian_mode; c->fd_eps = fd_eps; c->early_exit = early_exit ? 1 : 0;
  return ILQR_OK;
}
int ilqr_hip_set_early_exit_gate(ilqr_hip_ctx* c, int on) { if (!c) return ILQR_ERR_ARG; c->ee_gate = on ? 1 : 0; return ILQR_OK; }

// ---------------------------------------------------------------- initializeWithReference
// Which kernel rolls a trajectory out under these switches: the scalar kernels (ILQR_DYN=s), the two-lane kernels (ILQR_ROLLOUT=s,
// and always in contact mode) or the one-lane ones.  Iteration 0 of a solve may re-roll a cold start BESIDE the linearisation only
// if it is this very kernel under these very dynamics parameters (bit-identical result); compared field by field, not by memcmp
// (padding bytes).
static int rollout_kernel_identity(const h1::ProblemDev& P) {
  return ilqr::variant_scalar_dyn() ? 2 : ((ilqr::variant_rollout_split() || h1::constrained(P.dyn)) ? 1 : 0);
}
static bool same_dyn(const h1::DynParams& a, const h1::DynParams& b) {
  return a.h == b.h && a.g[0] == b.g[0] && a.g[1] == b.g[1] && a.g[2] == b.g[2] && a.contact == b.contact && a.soft == b.soft && a.mu == b.mu && a.limits == b.limits && a.lim_k == b.lim_k;
}
static int cold_start_device(ilqr_hip_ctx* c, const double* x0_dev, const double* uinit_dev) {
  const size_t B = c->B, N = c->N;
  HIPCHK(c, hipMemcpyAsync(c->S.x0, x0_dev, B * ILQR_NX * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->S.ubar, uinit_dev, B * N * ILQR_NU * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  ilqr::launch_rollout(c->S, c->P, ilqr::MASK_ALL, 1, 0, c->S.Jbase, c->stream);  // N rollouts (ilqr.cpp:113-115)
  HIPCHK(c, hipGetLastError());
  c->initialized = true;
  c->xbar_rolled = true;
  c->rolled_variant = rollout_kernel_identity(c->P); c->rolled_dyn = c->P.dyn;
  return ILQR_OK;
}
int ilqr_hip_initialize_device(ilqr_hip_ctx* c, const double* x0_device, const double* u_init_device) {
  if (!c || !x0_device || !u_init_device) return ILQR_ERR_ARG;
  enter(c);
  return cold_start_device(c, x0_device, u_init_device);
}
int ilqr_hip_initialize(ilqr_hip_ctx* c, const double* x0, const double* u_init, const double* prev_xbar, const double* prev_ubar) {
  if (!c || !x0) return ILQR_ERR_ARG;
  enter(c);
  const size_t B = c->B, N = c->N;
  if (prev_xbar && prev_ubar) {
    HIPCHK(c, hipMemcpyAsync(c->S.x0, x0, B * ILQR_NX * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_prevx, prev_xbar, B * (N + 1) * ILQR_NX * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_prevu, prev_ubar, B * N * ILQR_NU * sizeof(double), hipMemcpyHostToDevice, c->stream));
    ilqr::launch_warm_shift(c->S, c->d_prevx, c->d_prevu, c->stream);
    ilqr::launch_last_step(c->S, c->P, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->initialized = true;
    c->xbar_rolled = false;
    return ILQR_OK;
  }
  std::vector<double> ug;
  const double* uh = u_init;
  if (!uh) {  // gravity compensation evaluated at each rollout's x0 (computeGravComp uses the plant state)
    ug.resize(B * N * ILQR_NU);
    for (size_t b = 0; b < B; ++b) {
      double u1[ILQR_NU]; h1host::gravity_compensation(x0 + b * ILQR_NX, c->P.dyn.g, u1);
      for (size_t t = 0; t < N; ++t) std::memcpy(&ug[(b * N + t) * ILQR_NU], u1, sizeof(u1));
    }
    uh = ug.data();
  }
  HIPCHK(c, hipMemcpyAsync(c->d_tmpx, x0, B * ILQR_NX * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_tmpu, uh, B * N * ILQR_NU * sizeof(double), hipMemcpyHostToDevice, c->stream));
  TRY(cold_start_device(c, c->d_tmpx, c->d_tmpu));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ILQR_OK;
}
int ilqr_hip_initialize_warm_resident(ilqr_hip_ctx* c, const double* x0) {
  if (!c || !x0) return ILQR_ERR_ARG;
  if (!c->initialized) return ILQR_ERR_STATE;
  enter(c);
  const size_t B = c->B, N = c->N;
  HIPCHK(c, hipMemcpyAsync(c->S.x0, x0, B * ILQR_NX * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_prevx, c->S.xbar, B * (N + 1) * ILQR_NX * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_prevu, c->S.ubar, B * N * ILQR_NU * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  ilqr::launch_warm_shift(c->S, c->d_prevx, c->d_prevu, c->stream);
  ilqr::launch_last_step(c->S, c->P, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->xbar_rolled = false;
  return ILQR_OK;
}

// ---------------------------------------------------------------- solve
static void collect_profile(ilqr_hip_ctx* c) {
  for (int i = 0; i < 8; ++i) { c->stage_ms[i] = 0; c->stage_launches[i] = 0; }
  for (auto& s : c->spans) { float ms = 0; hipEventElapsedTime(&ms, s.a, s.b); c->stage_ms[s.stage] += ms; c->stage_launches[s.stage] += 1; }
  c->spans.clear(); c->pool_next = 0;
}

static int slices_wanted(const ilqr_hip_ctx* c, int B) {
  int k = c->knobs.slices;
  if (k < 1) k = 1;
  if (k > 32) k = 32;
  while (k > 1 && B / k < 64) --k;   // a slice is at least one wave of the widest kernels
  return k;
}
// view of rollouts [b0, b0 + Bs) of the batch (every array is rollout-major)
static DevState slice_state(const DevState& S, size_t b0, int Bs) {
  DevState T = S;
  const size_t N = S.N, n = ILQR_NX, m = ILQR_NU, mi = S.max_iter;
  T.B = Bs;
  T.x0 += b0 * n; T.xbar += b0 * (N + 1) * n; T.ubar += b0 * N * m;
  T.xcand += b0 * 8 * (N + 1) * n; T.ucand += b0 * 8 * N * m; T.cand_cost += b0 * 8; T.cand_knot += b0 * 8 * (N + 1);
  T.A += b0 * N * n * n; T.Bm += b0 * N * n * m;
  T.lx += b0 * (N + 1) * n; T.lu += b0 * N * m; T.lxx += b0 * (N + 1) * n * n; T.luu += b0 * N * m;
  T.K += b0 * N * m * n; T.kff += b0 * N * m; T.lin_dump += b0 * N * ilqr::lin_dump_doubles();
  T.quad_knot0 = S.quad_knot0 + (long)(b0 * (N + 1));      // (the record buffer is indexed by knot, 16 to a line: not a pointer offset)
  T.Vx += b0 * n; T.Vxx += b0 * n * n;
  T.J += b0; T.Jbase += b0; T.ls_cost += b0; T.lambda += b0;
  T.active += b0; T.need_retry += b0; T.iters += b0; T.improved += b0; T.alpha_idx += b0;
  T.trace_cost += b0 * (mi + 1); T.trace_alpha += b0 * mi; T.trace_lambda += b0 * mi;
  T.order = nullptr; T.order_n = nullptr;      // the compacted lists index the whole batch: not used by slices
  T.grp_a = T.grp_r = T.order_r = T.order_rn = T.order_an = nullptr;
  return T;
}
static h1::ProblemDev slice_problem(const h1::ProblemDev& P, long b0) {
  h1::ProblemDev T = P;
  T.x_ref += b0 * P.x_ref_stride; T.u_ref += b0 * P.u_ref_stride; T.com_ref += b0 * P.com_ref_stride;
  T.stance += b0 * P.stance_stride; T.ee_ref += b0 * P.ee_ref_stride; T.com_vel_ref += b0 * P.com_vel_ref_stride;
  return T;
}
static int ensure_slices(ilqr_hip_ctx* c, int k) {
  if (!c->ev_begin) HIPCHK(c, hipEventCreateWithFlags(&c->ev_begin, hipEventDisableTiming));
  while ((int)c->slices.size() < k) {
    ilqr_hip_ctx::Slice sl;
    HIPCHK(c, hipStreamCreate(&sl.st)); HIPCHK(c, hipStreamCreate(&sl.st2)); HIPCHK(c, hipStreamCreate(&sl.st3));
    HIPCHK(c, hipEventCreateWithFlags(&sl.roll, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&sl.fork, hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&sl.join, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&sl.lead, hipEventDisableTiming));
    c->slices.push_back(sl);
  }
  return ILQR_OK;
}
// the launch sequence of iLQR::solve (ilqr.cpp:521-660) for one slice on its streams; `wait_lead` (optional) delays the
// first throughput-bound stage until the previous slice has finished its first backward pass, `lead` is recorded there
// the handle's own setting (ilqr_hip_set_early_exit_gate); the environment, when set, overrides it (diagnostics, tests)
static int early_exit_gate(const ilqr_hip_ctx* c) { return c->knobs.ee_gate >= 0 ? c->knobs.ee_gate : c->ee_gate; }
static int ensure_gate(ilqr_hip_ctx* c) {
  if ((int)c->ev_active.size() >= c->max_iter + 2 && c->h_active) return ILQR_OK;
  if (c->h_active) { (void)hipHostFree(c->h_active); c->h_active = nullptr; }
  HIPCHK(c, hipHostMalloc((void**)&c->h_active, sizeof(int) * (size_t)(c->max_iter + 2), hipHostMallocDefault));
  while ((int)c->ev_active.size() < c->max_iter + 2) { hipEvent_t e; HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_active.push_back(e); }
  return ILQR_OK;
}
// Speculative lambda retry: while at most SPEC_MAX rollouts are in a pass (the whole batch, or -- convergence exit with the gate --
// the count the host has seen), two Riccati passes / line searches of that many one-wave rollouts fit on the chip's 1024 SIMDs side by
// side.  ILQR_SPEC=0 switches it off (sequential retry as for large passes), ILQR_SPEC_MAX moves the threshold (diagnostics, tests).
// early continuation (enqueue_solve): by default with the convergence exit only -- measured on one MI355X, executed iterations/s with / without:
// constraint-free B = 4096 +1.5 %, contact B = 4096 +9 %, contact B = 1024 +5 %, configs[4] +4 %; with a fixed iteration count most rollouts
// retry in most iterations, the early group is small, and its kernels only take SIMDs from the one-wave-per-SIMD kernels of the retry
// (headline -3 %, contact +3 % / -3.5 % at B = 4096 / 1024).  ILQR_SPLIT=0 / 1 forces it off / on.
static int split_enabled(const ilqr_hip_ctx* c) { return c->knobs.split >= 0 ? c->knobs.split : c->early_exit; }
static int alloc_twin(ilqr_hip_ctx* c);
// (the side-by-side order is an optimisation: a handle that cannot get the memory keeps the sequential order instead of failing the solve)
// Memory: the twin is indexed by rollout like everything else, so it is a full-batch copy of K, k, Vx, Vxx, the eight candidates and
// their knot costs -- 0.33 MB per rollout at N = 25 (1.3 GB at B = 4096, 10.7 GB at B = 32 768), allocated by the first solve that can
// take the side-by-side order: B <= ILQR_SPEC_MAX, or the convergence exit with its gate on (the late passes of any batch shrink below
// the threshold).  ILQR_SPEC=0 keeps a handle from ever allocating it.
static int ensure_twin(ilqr_hip_ctx* c) {
  if (c->twin || c->twin_failed) return ILQR_OK;
  if (alloc_twin(c) != ILQR_OK) {
    void* tw[] = {c->T.K, c->T.kff, c->T.Vx, c->T.Vxx, c->T.xcand, c->T.ucand, c->T.cand_cost, c->T.cand_knot, c->T.lambda, c->d_spec_gate};
    for (void* p : tw) if (p) (void)hipFree(p);
    c->T = DevState{}; c->d_spec_gate = nullptr; c->twin = false; c->twin_failed = true;
    (void)hipGetLastError();
    c->err.clear();      // (the failed allocation is not an error of the call: the sequential order runs instead)
  }
  return ILQR_OK;
}
static int alloc_twin(ilqr_hip_ctx* c) {
  const size_t B = c->B, N = c->N, n = ILQR_NX, m = ILQR_NU;
  DevState& T = c->T;
  T = c->S;
  T.K = T.kff = T.Vx = T.Vxx = T.xcand = T.ucand = T.cand_cost = T.cand_knot = T.lambda = nullptr;
  c->twin = true;       // (from here on destroy frees whatever was allocated)
  TRY(dalloc(c, &T.K, B * N * m * n + 32)); TRY(dalloc(c, &T.kff, B * N * m)); TRY(dalloc(c, &T.Vx, B * n)); TRY(dalloc(c, &T.Vxx, B * n * n));
  TRY(dalloc(c, &T.xcand, B * 8 * (N + 1) * n)); TRY(dalloc(c, &T.ucand, B * 8 * N * m)); TRY(dalloc(c, &T.cand_cost, B * 8)); TRY(dalloc(c, &T.cand_knot, B * 8 * (N + 1)));
  TRY(dalloc(c, &T.lambda, B));
  TRY(dalloc(c, &c->d_spec_gate, 4));
  if (!c->ev_spec_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_spec_fork, hipEventDisableTiming));
  if (!c->ev_spec_join) HIPCHK(c, hipEventCreateWithFlags(&c->ev_spec_join, hipEventDisableTiming));
  return ILQR_OK;
}
// the twin view of (a slice of) the batch: everything of S, with the twin's own gains, value function, candidates and lambda
static DevState twin_view(const ilqr_hip_ctx* c, const DevState& S) {
  DevState T = S;
  const DevState& W = c->T; const DevState& F = c->S;
  T.K = W.K + (S.K - F.K); T.kff = W.kff + (S.kff - F.kff); T.Vx = W.Vx + (S.Vx - F.Vx); T.Vxx = W.Vxx + (S.Vxx - F.Vxx);
  T.xcand = W.xcand + (S.xcand - F.xcand); T.ucand = W.ucand + (S.ucand - F.ucand); T.cand_cost = W.cand_cost + (S.cand_cost - F.cand_cost);
  T.cand_knot = W.cand_knot + (S.cand_knot - F.cand_knot); T.lambda = W.lambda + (S.lambda - F.lambda);
  return T;
}
static int enqueue_solve(ilqr_hip_ctx* c, const DevState& S, const h1::ProblemDev& P, hipStream_t st, hipStream_t st2, hipStream_t st3,
                         hipEvent_t ev_fork, hipEvent_t ev_join, hipEvent_t ev_roll, const double* shadow_base, hipEvent_t wait_lead, hipEvent_t lead,
                         hipEvent_t ev_lin = nullptr, hipEvent_t ev_adopt = nullptr) {
  // shadow target of the concurrent re-rollout: same rollouts as S.xbar, in the shadow buffer
  double* shadow = const_cast<double*>(shadow_base) + (S.xbar - c->S.xbar);
  const double fold_h = ilqr::linearize_fold_h(P, c->jac_mode);
  { StageTimer T(c, 0, st); ilqr::launch_rollout(S, P, ilqr::MASK_ALL, 0, 0, S.Jbase, st); ilqr::launch_solve_begin(S, st); }  // ilqr.cpp:540
  // With the convergence exit on, the launches of an iteration nobody needs are pure latency (17 launches that find nothing to
  // do): the count of rollouts active after iteration i (DevState::order_n, maintained by k_control) follows iteration i to the
  // host, which enqueues iteration i only after it has seen the count left by iteration i - 2 -- one full iteration stays
  // queued on the device meanwhile, so the device never waits for the host.  (ILQR_EE_GATE=0: always enqueue max_iter.)
  // without the convergence exit every rollout stays active: the per-knot kernels then skip the selection altogether
  const int sel_mode = c->early_exit ? ilqr::MASK_ACTIVE : ilqr::MASK_ALL;
  const bool gate = c->early_exit && S.order && early_exit_gate(c);
  const bool xbar_rolled = c->first_aside;
  // the operand-layout Riccati kernel (analytic Jacobians, riccati_pack.hip) has its producers write A_t, B_t and lxx~_t in its own
  // layout; the generic one-wave kernel reads only the tiles I >= J of lxx_t (t < N): the cost quadratics then leave the others unwritten
  const int pack = (fold_h != 0.0 && ilqr::variant_pack()) ? 1 : 0;
  const int lxx_lower = pack ? 2 : (ilqr::variant_backward() == 2 ? 1 : 0);
  // (iteration 0 rewrites A_t, B_t, lxx_t of every rollout: a solve leaves one layout behind; ilqr_hip_solve_async has prepared the padding)
  c->lxx_layout = lxx_lower;
  c->ab_packed = pack != 0;
  c->lin_fold_h = fold_h;        // what S.A / S.Bm hold after this solve; assigned with the layout flags (an early return above leaves all of them alone)
  if (pack) c->packed_h = fold_h;
  if (!pack) c->ab_pads_clean = false;
  if (gate) TRY(ensure_gate(c));
  c->iterations_enqueued = c->max_iter;
  // One group's share of the concurrent region of an iteration -- linearisation (:576) on G.m, cost quadratics (:588) on G.q, the
  // nominal re-rollout (:551,563) into the shadow buffer and its adoption on G.r -- behind an event recorded on fork_src.
  //   From the second iteration on the nominal trajectory already is a rollout from x0 (the accepted line-search candidate, or the
  //   unchanged previous nominal), so the re-rollout reproduces it bit for bit (ilqr_hip_get_adopt_mismatches): it runs beside the
  //   linearisation and is adopted (with its cost, the line-search baseline) once the linearisation and the cost quadratics have
  //   read the old copy; the backward pass reads neither, only the line search waits for the adoption.  ILQR_OVERLAP_ROLLOUT=0
  //   restores the sequential order; ILQR_REUSE_ROLLOUT=1 skips the re-rollout (not the default: SURVEY 8(d) counts it).
  //   (only while the line search and the rollout run the same step implementation; iteration 0 as well when the nominal trajectory
  //   is itself a rollout by this very kernel -- the cold start, initializeWithReference ilqr.cpp:113-115; a warm-shifted or
  //   caller-supplied trajectory is rolled out BEFORE the linearisation, as the reference does)
  // Sm: the view the mask-selected kernels (rollout, trajectory cost, adoption) get -- S, or S with a group's flags as `active`;
  // wl: the group's compacted list for the per-knot kernels (null: the iteration's own list / mask, as knot_mode and iter_l say).
  // ilqr_hip_set_dedup_saturated_retry: bit 1 of k_control's early_exit argument (the sequential orders; the side-by-side order has
  // both passes in flight before either outcome is known)
  const int dedup_bit = (c->knobs.dedup_retry >= 0 ? c->knobs.dedup_retry : c->dedup_retry) ? 2 : 0;
  struct Rg { hipStream_t m, q, r; hipEvent_t fork, join, roll, lin, adopt; };
  const Rg G0{st, st2, st3, ev_fork, ev_join, ev_roll, ev_lin, ev_adopt};
  // (group A's cost quadratics and re-rollout share the second and third stream with group R -- idle while the retry runs, and R's
  // work comes behind A's anyway --, only its linearisation has a stream of its own: the runtime maps streams onto four hardware
  // queues by default, and streams that share a queue serialise)
  const Rg GA{c->a1, st2, st3, c->evA_fork, c->evA_join, c->evA_roll, c->evA_lin, c->evA_adopt};
  auto region = [&](const Rg& G, hipStream_t fork_src, const DevState& Sm, int knot_mode, int iter_l, const ilqr::WorkList* wl, bool concurrent_roll) -> int {
    HIPCHK(c, hipEventRecord(G.fork, fork_src));
    if (G.m != fork_src) HIPCHK(c, hipStreamWaitEvent(G.m, G.fork, 0));
    HIPCHK(c, hipStreamWaitEvent(G.q, G.fork, 0));
    if (concurrent_roll) {
      HIPCHK(c, hipStreamWaitEvent(G.r, G.fork, 0));
      DevState Sr = Sm; Sr.xbar = shadow;
      { StageTimer T(c, 0, G.r); ilqr::launch_rollout(Sr, P, ilqr::MASK_ACTIVE, 1, 0, S.Jbase, G.r); }
      HIPCHK(c, hipEventRecord(G.roll, G.r));
    }
    { StageTimer T(c, 2, G.q); ilqr::launch_cost_quadratics(S, P, knot_mode, G.q, iter_l, lxx_lower, wl); }
    HIPCHK(c, hipEventRecord(G.join, G.q));
    { StageTimer T(c, 1, G.m); ilqr::launch_linearize(S, P, knot_mode, c->jac_mode, c->fd_eps, G.m, 3, iter_l, pack, wl); }
    if (concurrent_roll && G.lin && G.adopt) {
      HIPCHK(c, hipEventRecord(G.lin, G.m));
      HIPCHK(c, hipStreamWaitEvent(G.r, G.lin, 0)); HIPCHK(c, hipStreamWaitEvent(G.r, G.join, 0));
      ilqr::launch_adopt_rollout(Sm, shadow, ilqr::MASK_ACTIVE, c->d_mismatch, G.r);
      HIPCHK(c, hipEventRecord(G.adopt, G.r));
    }
    return ILQR_OK;
  };
  auto rolls_aside = [&](int iter) {
    const bool first_aside = iter == 0 && xbar_rolled && !ilqr::variant_scalar_dyn();
    return (iter > 0 || first_aside) && !c->knobs.reuse_rollout && c->knobs.overlap_rollout && (h1::constrained(P.dyn) || ilqr::variant_ls_split() == ilqr::variant_rollout_split());   // (contact mode: both on the two-lane kernels)
  };
  // Early continuation: the rollouts whose first line search of iteration i accepted a step are done with iteration i; their share of
  // iteration i + 1's concurrent region (group A) starts right behind the first control pass, on streams of its own, while the
  // others take their lambda retry (:619-644); those follow as group R behind the second control pass, and both groups meet again at
  // the backward pass.  The retry of an early iteration keeps a fraction of the chip busy for two latency-bound passes: the time it
  // left idle is what this recovers.  Same kernels on the same data in the same per-rollout order: results are bit-identical
  // (GPU test).  On by default with the convergence exit (split_enabled above); ILQR_SPLIT=0 / 1 forces one / two groups.
  // (analytic Jacobians only: the forward-difference launchers select by S.active, not by the group's list, and use S.A / S.Bm as
  // scratch that k_fd_finish rewrites in place -- group A's pass would rewrite the Jacobians the retry's backward pass is reading)
  const bool can_split = split_enabled(c) && S.order && S.grp_a && ev_lin && ev_adopt && c->a1 != nullptr && c->jac_mode == ILQR_JAC_ANALYTIC;
  bool prev_split = false;      // group A of the iteration at hand is already enqueued
  for (int iter = 0; iter < c->max_iter; ++iter) {
    if (gate && iter >= 2) {
      HIPCHK(c, hipEventSynchronize(c->ev_active[iter - 2]));
      if (c->h_active[iter - 1] == 0) { c->iterations_enqueued = iter; break; }     // nobody was active in iteration iter - 1
    }
    const bool concurrent_roll = rolls_aside(iter);
    if (!prev_split) {
      if ((iter == 0 || !c->knobs.reuse_rollout) && !concurrent_roll) { StageTimer T(c, 0, st); ilqr::launch_rollout(S, P, ilqr::MASK_ACTIVE, 1, 0, S.Jbase, st); }
      if (iter == 0 && wait_lead) HIPCHK(c, hipStreamWaitEvent(st, wait_lead, 0));
      TRY(region(G0, st, S, sel_mode, iter, nullptr, concurrent_roll));
    } else {
      DevState Sg = S; Sg.active = S.grp_r;
      const ilqr::WorkList wr{S.order_r, S.order_rn + iter};
      TRY(region(G0, st, Sg, ilqr::MASK_ACTIVE, -1, &wr, true));
      ++c->split_iterations;
    }
    const bool adopt_aside = concurrent_roll && ev_lin && ev_adopt;
    HIPCHK(c, hipStreamWaitEvent(st, ev_join, 0));
    if (prev_split) { HIPCHK(c, hipStreamWaitEvent(st, GA.join, 0)); HIPCHK(c, hipStreamWaitEvent(st, GA.lin, 0)); }
    if (concurrent_roll && !adopt_aside) { HIPCHK(c, hipStreamWaitEvent(st, ev_roll, 0)); ilqr::launch_adopt_rollout(S, shadow, ilqr::MASK_ACTIVE, c->d_mismatch, st); }
    auto wait_adoption = [&](hipStream_t t) -> int {
      if (adopt_aside) HIPCHK(c, hipStreamWaitEvent(t, ev_adopt, 0));
      if (prev_split) HIPCHK(c, hipStreamWaitEvent(t, GA.adopt, 0));
      return ILQR_OK;
    };
    // (with the gate the host has seen how many rollouts were still active at the start of iteration iter - 1: an upper bound
    // for this iteration's passes -- the active set only shrinks)
    const int ls_bound = (gate && iter >= 2) ? c->h_active[iter - 1] : -1;
    const int pass_bound = ls_bound >= 0 ? ls_bound : S.B;
    if (c->twin && c->knobs.spec && S.order && pass_bound <= c->knobs.spec_max) {
      // both passes of ilqr.cpp:601-646 side by side (k_control_spec): the twin on the second stream
      const DevState Tw = twin_view(c, S);
      ++c->spec_iterations;
      HIPCHK(c, hipEventRecord(c->ev_spec_fork, st));
      HIPCHK(c, hipStreamWaitEvent(st2, c->ev_spec_fork, 0));
      ilqr::launch_spec_lambda(S, Tw.lambda, st2);
      { StageTimer T(c, 3, st); ilqr::launch_backward(S, ilqr::MASK_ACTIVE, st, fold_h, iter); }
      { StageTimer T(c, 6, st2); ilqr::launch_backward(Tw, ilqr::MASK_ACTIVE, st2, fold_h, iter); }
      TRY(wait_adoption(st)); TRY(wait_adoption(st2));
      if (iter == 0 && lead) HIPCHK(c, hipEventRecord(lead, st));
      { StageTimer T(c, 4, st); ilqr::launch_line_search(S, P, ilqr::MASK_ACTIVE, st, iter, ls_bound); }
      { StageTimer T(c, 7, st2); ilqr::launch_line_search(Tw, P, ilqr::MASK_ACTIVE, st2, iter, ls_bound); }
      HIPCHK(c, hipEventRecord(c->ev_spec_join, st2));
      HIPCHK(c, hipStreamWaitEvent(st, c->ev_spec_join, 0));
      { StageTimer T(c, 5, st); ilqr::launch_control_spec(S, Tw, iter, c->tol, c->early_exit, st, ilqr::ls_costs_per_knot(P)); }
      if (gate) {
        HIPCHK(c, hipMemcpyAsync(&c->h_active[iter + 1], S.order_n + 2 * (iter + 1), sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipEventRecord(c->ev_active[iter], st));
      }
      prev_split = false;
      continue;
    }
    if (c->twin && c->knobs.spec && c->knobs.spec_dual && S.order && gate && pass_bound <= 4 * c->knobs.spec_max && ilqr::spec_dual_available(P)) {
      // The host's count is one iteration old: between spec_max and 4 spec_max the pass may or may not have shrunk below the
      // threshold by now.  Both orders are enqueued and the device takes one (launch_spec_gate): the twin's launches and the
      // one-rollout-per-wave line search see a count of zero unless the list holds <= spec_max rollouts, the sequential first line
      // search and bookkeeping see zero if it does; the lambda-retry launches find an empty retry list behind k_control_spec.
      const DevState Tw = twin_view(c, S);
      const int* list = S.order + (size_t)(2 * iter) * S.B;
      int* g = c->d_spec_gate;
      ++c->spec_iterations;
      ilqr::launch_spec_gate(S, iter, c->knobs.spec_max, g, st);
      HIPCHK(c, hipEventRecord(c->ev_spec_fork, st));
      HIPCHK(c, hipStreamWaitEvent(st2, c->ev_spec_fork, 0));
      ilqr::launch_spec_lambda(S, Tw.lambda, st2);
      { StageTimer T(c, 3, st); ilqr::launch_backward(S, ilqr::MASK_ACTIVE, st, fold_h, iter); }
      { StageTimer T(c, 6, st2); ilqr::launch_backward_list(Tw, st2, fold_h, list, g); }
      TRY(wait_adoption(st)); TRY(wait_adoption(st2));
      { StageTimer T(c, 4, st);
        ilqr::launch_line_search_list(S, P, st, list, g, c->knobs.spec_max);
        ilqr::launch_line_search_list(S, P, st, list, g + 2, ls_bound);
        ilqr::launch_cand_costs(S, P, ilqr::MASK_ACTIVE, st, false); }
      { StageTimer T(c, 7, st2); ilqr::launch_line_search_list(Tw, P, st2, list, g, c->knobs.spec_max); ilqr::launch_cand_costs(Tw, P, ilqr::MASK_ACTIVE, st2, false, g); }
      HIPCHK(c, hipEventRecord(c->ev_spec_join, st2));
      HIPCHK(c, hipStreamWaitEvent(st, c->ev_spec_join, 0));
      { StageTimer T(c, 5, st);
        ilqr::launch_control_spec(S, Tw, iter, c->tol, c->early_exit, st, ilqr::ls_costs_per_knot(P), g);
        ilqr::launch_control(S, 0, iter, c->tol, c->early_exit | dedup_bit, st, ilqr::ls_costs_per_knot(P), g + 1); }
      { StageTimer T(c, 6, st); ilqr::launch_backward(S, ilqr::MASK_RETRY, st, fold_h, iter); }
      { StageTimer T(c, 7, st); ilqr::launch_line_search(S, P, ilqr::MASK_RETRY, st, iter, ls_bound); }
      { StageTimer T(c, 5, st); ilqr::launch_control(S, 1, iter, c->tol, c->early_exit, st, ilqr::ls_costs_per_knot(P)); }
      HIPCHK(c, hipMemcpyAsync(&c->h_active[iter + 1], S.order_n + 2 * (iter + 1), sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(c, hipEventRecord(c->ev_active[iter], st));
      prev_split = false;
      continue;
    }
    { StageTimer T(c, 3, st); ilqr::launch_backward(S, ilqr::MASK_ACTIVE, st, fold_h, iter); }                                  // :601
    TRY(wait_adoption(st));
    if (iter == 0 && lead) HIPCHK(c, hipEventRecord(lead, st));
    { StageTimer T(c, 4, st); ilqr::launch_line_search(S, P, ilqr::MASK_ACTIVE, st, iter, ls_bound); }                  // :616
    { StageTimer T(c, 5, st); ilqr::launch_control(S, 0, iter, c->tol, c->early_exit | dedup_bit, st, ilqr::ls_costs_per_knot(P)); }          // :619-620,645-655
    const bool split_next = can_split && iter + 1 < c->max_iter && rolls_aside(iter + 1);
    if (split_next) {
      // group A of iteration iter + 1: the first entries of its list, as many as the first control pass has just put there
      HIPCHK(c, hipMemcpyAsync(S.order_an + iter + 1, S.order_n + 2 * (iter + 1), sizeof(int), hipMemcpyDeviceToDevice, st));
      DevState Sg = S; Sg.active = S.grp_a;
      const ilqr::WorkList wa{S.order + (size_t)(2 * (iter + 1)) * S.B, S.order_an + iter + 1};
      TRY(region(GA, st, Sg, ilqr::MASK_ACTIVE, -1, &wa, true));
    }
    { StageTimer T(c, 6, st); ilqr::launch_backward(S, ilqr::MASK_RETRY, st, fold_h, iter); }                                  // :637
    { StageTimer T(c, 7, st); ilqr::launch_line_search(S, P, ilqr::MASK_RETRY, st, iter, ls_bound); }                   // :638
    { StageTimer T(c, 5, st); ilqr::launch_control(S, 1, iter, c->tol, c->early_exit, st, ilqr::ls_costs_per_knot(P)); }                      // :640-646
    if (gate) {
      HIPCHK(c, hipMemcpyAsync(&c->h_active[iter + 1], S.order_n + 2 * (iter + 1), sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(c, hipEventRecord(c->ev_active[iter], st));
    }
    prev_split = split_next;
  }
  if (prev_split) {      // (the convergence exit ended the loop behind a group A that found nothing to do: its streams rejoin)
    HIPCHK(c, hipStreamWaitEvent(st, GA.join, 0)); HIPCHK(c, hipStreamWaitEvent(st, GA.lin, 0)); HIPCHK(c, hipStreamWaitEvent(st, GA.adopt, 0));
  }
  return ILQR_OK;
}
int ilqr_hip_get_split_iterations(const ilqr_hip_ctx* c) { return c ? c->split_iterations : -1; }
int ilqr_hip_get_speculative_iterations(const ilqr_hip_ctx* c) { return c ? c->spec_iterations : -1; }
int ilqr_hip_set_dedup_saturated_retry(ilqr_hip_ctx* c, int on) { if (!c) return ILQR_ERR_ARG; c->dedup_retry = on ? 1 : 0; return ILQR_OK; }
int ilqr_hip_reload_environment(ilqr_hip_ctx* c) {
  if (!c) return ILQR_ERR_ARG;
  const Knobs k = read_knobs();
  if (!ilqr::variants_supported(k.var)) { c->err = "the environment selects a kernel family this library does not hold (see ilqr_hip_create)"; return ILQR_ERR_UNSUPPORTED; }
  const bool pc = c->knobs.per_call; c->knobs = k; c->knobs.per_call = k.per_call || pc;
  return ILQR_OK;
}
int ilqr_hip_num_slices(const ilqr_hip_ctx* c) { return c ? slices_wanted(c, c->B) : -1; }
// The analytic Jacobians differentiate the constrained step with the active set and the sliding decisions held fixed.  The sliding
// branch (contact modes 3 / 4: the normal row and the normal force turn with the foot, kinetic friction follows the sticking
// solution) is carried by the two-knot tangent kernel only (k_lin_tangent2c<., 1 / 2>): the one-knot and scalar cross-check families
// refuse it.
static int jacobians_available(ilqr_hip_ctx* c) {
  if (c->P.dyn.limits && c->jac_mode == ILQR_JAC_ANALYTIC && (ilqr::variant_lin_one_knot() || ilqr::variant_scalar_dyn())) {
    c->err = "joint-limit rows (ilqr_hip_set_joint_limits): analytic Jacobians are not available in this kernel family (ILQR_LIN / ILQR_DYN), select ILQR_JAC_FD_FORWARD with ilqr_hip_set_options";
    return ILQR_ERR_UNSUPPORTED;
  }
  if (c->P.dyn.contact >= ILQR_CONTACT_FRICTION_STANCE && c->jac_mode == ILQR_JAC_ANALYTIC && (ilqr::variant_lin_one_knot() || ilqr::variant_scalar_dyn())) {
    c->err = "contact modes 3 / 4 (Coulomb limit): analytic Jacobians are not available in this kernel family (ILQR_LIN / ILQR_DYN), select ILQR_JAC_FD_FORWARD with ilqr_hip_set_options";
    return ILQR_ERR_UNSUPPORTED;
  }
  return ILQR_OK;
}
int ilqr_hip_solve_async(ilqr_hip_ctx* c) {
  if (!c) return ILQR_ERR_ARG;
  if (!c->initialized || !c->refs_set) { c->err = "solve before initialize/set_references"; return ILQR_ERR_STATE; }
  enter(c);      // (first: the checks below read this handle's kernel family, not the last one entered on this thread)
  ENV_REFUSED(c);
  if (int rc = jacobians_available(c)) return rc;
  hipStream_t st = c->stream;
  const DevState& S = c->S; const h1::ProblemDev& P = c->P;
  c->spans.clear(); c->pool_next = 0;
  HIPCHK(c, hipMemsetAsync(c->d_mismatch, 0, sizeof(unsigned long long), st));
  const int k = slices_wanted(c, c->B);
  c->n_slices = k;
  c->spec_iterations = 0; c->split_iterations = 0;
  if (c->knobs.spec && k <= 1 && !c->twin && (c->B <= c->knobs.spec_max || (c->early_exit && early_exit_gate(c)))) TRY(ensure_twin(c));
  if (ilqr::linearize_fold_h(P, c->jac_mode) != 0.0 && ilqr::variant_pack() && !c->ab_pads_clean) { ilqr::launch_pack_zero_pads(S, st); c->ab_pads_clean = true; }
  c->first_aside = c->xbar_rolled && c->rolled_variant == rollout_kernel_identity(P) && same_dyn(c->rolled_dyn, P.dyn);
  c->xbar_rolled = false;                                   // after this solve xbar is an accepted line-search candidate
  if (k <= 1) {
    TRY(enqueue_solve(c, S, P, st, c->stream2, c->stream3, c->ev_fork, c->ev_join, c->ev_roll, c->d_shadowx, nullptr, nullptr, c->ev_lin, c->ev_adopt));
  } else {
    TRY(ensure_slices(c, k));
    HIPCHK(c, hipEventRecord(c->ev_begin, st));
    const int per = (c->B + k - 1) / k;
    const bool stagger = c->knobs.stagger != 0;
    for (int i = 0; i < k; ++i) {
      const int b0 = i * per, Bs = (b0 + per <= c->B) ? per : (c->B - b0);
      if (Bs <= 0) continue;
      auto& sl = c->slices[i];
      HIPCHK(c, hipStreamWaitEvent(sl.st, c->ev_begin, 0));
      const DevState Ss = slice_state(S, (size_t)b0, Bs);
      const h1::ProblemDev Ps = slice_problem(P, b0);
      TRY(enqueue_solve(c, Ss, Ps, sl.st, sl.st2, sl.st3, sl.fork, sl.join, sl.roll, c->d_shadowx, (stagger && i > 0) ? c->slices[i - 1].lead : nullptr, sl.lead));
      HIPCHK(c, hipEventRecord(sl.done, sl.st));
      HIPCHK(c, hipStreamWaitEvent(st, sl.done, 0));
    }
  }
  HIPCHK(c, hipGetLastError());
  return ILQR_OK;
}
int ilqr_hip_synchronize(ilqr_hip_ctx* c) {
  if (!c) return ILQR_ERR_ARG;
  enter(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->profiling) collect_profile(c);
  return ILQR_OK;
}
int ilqr_hip_solve(ilqr_hip_ctx* c, const double* x0, double* cost_out) {
  if (!c) return ILQR_ERR_ARG;
  enter(c);
  if (x0) {      // (a new x0: the nominal trajectory is rolled out from it before the linearisation, as the reference does)
    HIPCHK(c, hipMemcpyAsync(c->S.x0, x0, (size_t)c->B * ILQR_NX * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->xbar_rolled = false;
  }
  TRY(ilqr_hip_solve_async(c));
  TRY(ilqr_hip_synchronize(c));
  if (cost_out) HIPCHK(c, hipMemcpy(cost_out, c->S.J, (size_t)c->B * sizeof(double), hipMemcpyDeviceToHost));
  return ILQR_OK;
}

// ---------------------------------------------------------------- accessors
#define GETTER(name, ptr, count, type)                                                                          \
  int name(ilqr_hip_ctx* c, type* out) {                                                                        \
    if (!c || !out) return ILQR_ERR_ARG;                                                                        \
    enter(c);                                                                                    \
    HIPCHK(c, hipStreamSynchronize(c->stream));                                                                 \
    HIPCHK(c, hipMemcpy(out, c->S.ptr, (size_t)(count) * sizeof(type), hipMemcpyDeviceToHost));                 \
    return ILQR_OK;                                                                                             \
  }
GETTER(ilqr_hip_get_xbar, xbar, (size_t)c->B*(c->N + 1) * ILQR_NX, double)
GETTER(ilqr_hip_get_ubar, ubar, (size_t)c->B* c->N* ILQR_NU, double)
GETTER(ilqr_hip_get_gains_K, K, (size_t)c->B* c->N* ILQR_NU* ILQR_NX, double)
GETTER(ilqr_hip_get_gains_kff, kff, (size_t)c->B* c->N* ILQR_NU, double)
GETTER(ilqr_hip_get_cost, J, c->B, double)
GETTER(ilqr_hip_get_iterations, iters, c->B, int)
GETTER(ilqr_hip_get_lambda, lambda, c->B, double)

int ilqr_hip_get_trace(ilqr_hip_ctx* c, double* cost, double* alpha, double* lambda) {
  if (!c) return ILQR_ERR_ARG;
  enter(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (cost) HIPCHK(c, hipMemcpy(cost, c->S.trace_cost, (size_t)c->B * (c->max_iter + 1) * sizeof(double), hipMemcpyDeviceToHost));
  if (alpha) HIPCHK(c, hipMemcpy(alpha, c->S.trace_alpha, (size_t)c->B * c->max_iter * sizeof(double), hipMemcpyDeviceToHost));
  if (lambda) HIPCHK(c, hipMemcpy(lambda, c->S.trace_lambda, (size_t)c->B * c->max_iter * sizeof(double), hipMemcpyDeviceToHost));
  return ILQR_OK;
}
int ilqr_hip_first_knot_device(ilqr_hip_ctx* c, const double** u0, const double** K0, const double** cost) {
  if (!c) return ILQR_ERR_ARG;
  enter(c);
  ilqr::launch_pack_first_knot(c->S, c->d_u0, c->d_K0, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (u0) *u0 = c->d_u0;
  if (K0) *K0 = c->d_K0;
  if (cost) *cost = c->S.J;
  return ILQR_OK;
}
int ilqr_hip_pack_first_knot_device(ilqr_hip_ctx* c, double* u0_out, double* K0_out, double* cost_out) {
  if (!c || !u0_out) return ILQR_ERR_ARG;
  enter(c);
  ilqr::launch_pack_first_knot(c->S, u0_out, K0_out ? K0_out : c->d_K0, c->stream);
  HIPCHK(c, hipGetLastError());
  if (cost_out) HIPCHK(c, hipMemcpyAsync(cost_out, c->S.J, (size_t)c->B * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ILQR_OK;
}
int ilqr_hip_compute_control(ilqr_hip_ctx* c, const double* x_measured, double* u_apply) {
  if (!c || !x_measured || !u_apply) return ILQR_ERR_ARG;
  enter(c);
  HIPCHK(c, hipMemcpyAsync(c->d_tmpx, x_measured, (size_t)c->B * ILQR_NX * sizeof(double), hipMemcpyHostToDevice, c->stream));
  ilqr::launch_compute_control(c->S, c->d_tmpx, c->d_u0, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(u_apply, c->d_u0, (size_t)c->B * ILQR_NU * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ILQR_OK;
}

// ---------------------------------------------------------------- stage entry points
int ilqr_hip_set_trajectory(ilqr_hip_ctx* c, const double* xbar, const double* ubar) {
  if (!c || !xbar || !ubar) return ILQR_ERR_ARG;
  enter(c);
  const size_t B = c->B, N = c->N;
  HIPCHK(c, hipMemcpyAsync(c->S.xbar, xbar, B * (N + 1) * ILQR_NX * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->S.ubar, ubar, B * N * ILQR_NU * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(c->S.x0, ILQR_NX * sizeof(double), c->S.xbar, (N + 1) * ILQR_NX * sizeof(double), ILQR_NX * sizeof(double), B, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->initialized = true;
  c->xbar_rolled = false;
  return ILQR_OK;
}
#define STAGE_PRE if (!c) return ILQR_ERR_ARG; if (!c->initialized) return ILQR_ERR_STATE; enter(c); ENV_REFUSED(c)
#define STAGE_POST HIPCHK(c, hipGetLastError()); HIPCHK(c, hipStreamSynchronize(c->stream)); return ILQR_OK
int ilqr_hip_stage_rollout(ilqr_hip_ctx* c) { STAGE_PRE; ilqr::launch_rollout(c->S, c->P, ilqr::MASK_ALL, 1, 0, c->S.Jbase, c->stream); STAGE_POST; }
int ilqr_hip_stage_linearize(ilqr_hip_ctx* c) { STAGE_PRE; if (int rc = jacobians_available(c)) return rc; ilqr::launch_linearize(c->S, c->P, ilqr::MASK_ALL, c->jac_mode, c->fd_eps, c->stream); c->lin_fold_h = ilqr::linearize_fold_h(c->P, c->jac_mode); c->ab_packed = false; c->ab_pads_clean = false; STAGE_POST; }
int ilqr_hip_stage_cost_quadratics(ilqr_hip_ctx* c) { STAGE_PRE; if (!c->refs_set) return ILQR_ERR_STATE; ilqr::launch_cost_quadratics(c->S, c->P, ilqr::MASK_ALL, c->stream); c->lxx_layout = 0; STAGE_POST; }
// layout conversions on demand (in place): what a consumer of the standard layout (getters, any kernel family but the operand-layout
// one) or of the operand layout (stage API on riccati_pack.hip) calls first
static void want_standard_ab(ilqr_hip_ctx* c) { if (c->ab_packed) { ilqr::launch_unpack_ab(c->S, c->packed_h, c->stream); c->ab_packed = false; c->ab_pads_clean = false; } }
static void want_standard_lxx(ilqr_hip_ctx* c, bool whole) {
  if (c->lxx_layout == 2) { ilqr::launch_unpack_lxx(c->S, c->stream); c->lxx_layout = 0; }
  if (c->lxx_layout == 1 && whole) { ilqr::launch_mirror_lxx(c->S, c->stream); c->lxx_layout = 0; }
}
int ilqr_hip_stage_backward_pass(ilqr_hip_ctx* c) {
  STAGE_PRE;
  if (c->lin_fold_h != 0.0 && ilqr::variant_pack()) {
    // analytic Jacobians and the operand-layout kernel (riccati_pack.hip): convert in place what is not yet in its layout
    if (!c->ab_packed) { ilqr::launch_pack_ab(c->S, c->stream); c->ab_packed = true; c->packed_h = c->lin_fold_h; c->ab_pads_clean = true; }
    if (c->lxx_layout != 2) { want_standard_lxx(c, true); ilqr::launch_pack_lxx(c->S, c->stream); c->lxx_layout = 2; }
  } else {
    // any other family reads the standard layout (the one-wave kernel: the tiles I >= J of lxx_t, t < N, suffice)
    want_standard_ab(c);
    want_standard_lxx(c, ilqr::variant_backward() != 2);
  }
  ilqr::launch_backward(c->S, ilqr::MASK_ALL, c->stream, c->lin_fold_h);
  STAGE_POST;
}
int ilqr_hip_stage_total_cost(ilqr_hip_ctx* c, double* cost) {
  STAGE_PRE; if (!cost || !c->refs_set) return ILQR_ERR_ARG;
  ilqr::launch_rollout(c->S, c->P, ilqr::MASK_ALL, 0, 0, c->d_cost_tmp, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(cost, c->d_cost_tmp, (size_t)c->B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ILQR_OK;
}
int ilqr_hip_stage_line_search(ilqr_hip_ctx* c, int* improved, double* new_cost, double* alpha) {
  STAGE_PRE; if (!c->refs_set) return ILQR_ERR_STATE;
  ilqr::launch_rollout(c->S, c->P, ilqr::MASK_ALL, 0, 0, c->S.Jbase, c->stream);   // baseline = computeTotalCost(xbar, ubar), ilqr.cpp:317
  ilqr::launch_line_search(c->S, c->P, ilqr::MASK_ALL, c->stream);
  ilqr::launch_control(c->S, 2, 0, c->tol, 0, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t B = c->B;
  std::vector<int> ai(B);
  HIPCHK(c, hipMemcpy(ai.data(), c->S.alpha_idx, B * sizeof(int), hipMemcpyDeviceToHost));
  static const double alphas[8] = {1.0, 0.8, 0.6, 0.4, 0.2, 0.1, 0.05, 0.01};
  for (size_t b = 0; b < B; ++b) { if (improved) improved[b] = ai[b] >= 0; if (alpha) alpha[b] = ai[b] >= 0 ? alphas[ai[b]] : 0.0; }
  if (new_cost) HIPCHK(c, hipMemcpy(new_cost, c->S.ls_cost, B * sizeof(double), hipMemcpyDeviceToHost));
  return ILQR_OK;
}
int ilqr_hip_get_linearization(ilqr_hip_ctx* c, double* A, double* Bm) {
  if (!c) return ILQR_ERR_ARG; enter(c);
  const size_t B = c->B, N = c->N;
  want_standard_ab(c); HIPCHK(c, hipGetLastError());      // (the last solve may have left the operand layout behind)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (A) HIPCHK(c, hipMemcpy(A, c->S.A, B * N * ILQR_NX * ILQR_NX * sizeof(double), hipMemcpyDeviceToHost));
  if (Bm) HIPCHK(c, hipMemcpy(Bm, c->S.Bm, B * N * ILQR_NX * ILQR_NU * sizeof(double), hipMemcpyDeviceToHost));
  return ILQR_OK;
}
int ilqr_hip_set_linearization(ilqr_hip_ctx* c, const double* A, const double* Bm) {
  if (!c || !A || !Bm) return ILQR_ERR_ARG; enter(c);
  const size_t B = c->B, N = c->N;
  HIPCHK(c, hipMemcpy(c->S.A, A, B * N * ILQR_NX * ILQR_NX * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->S.Bm, Bm, B * N * ILQR_NX * ILQR_NU * sizeof(double), hipMemcpyHostToDevice));
  c->initialized = true;
  c->lin_fold_h = 0.0;      // Jacobians of unknown origin: generic backward kernel
  c->ab_packed = false; c->ab_pads_clean = false;
  return ILQR_OK;
}
int ilqr_hip_get_quadratics(ilqr_hip_ctx* c, double* lx, double* lu, double* lxx, double* luu) {
  if (!c) return ILQR_ERR_ARG; enter(c);
  const size_t B = c->B, N = c->N;
  if (lxx) { want_standard_lxx(c, false); HIPCHK(c, hipGetLastError()); }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (lx) HIPCHK(c, hipMemcpy(lx, c->S.lx, B * (N + 1) * ILQR_NX * sizeof(double), hipMemcpyDeviceToHost));
  if (lu) HIPCHK(c, hipMemcpy(lu, c->S.lu, B * N * ILQR_NU * sizeof(double), hipMemcpyDeviceToHost));
  if (lxx) {
    HIPCHK(c, hipMemcpy(lxx, c->S.lxx, B * (N + 1) * ILQR_NX * ILQR_NX * sizeof(double), hipMemcpyDeviceToHost));
    if (c->lxx_layout == 1) {   // the last solve stored the tiles I >= J of the knots t < N only (quad_kernels.hip): lxx is symmetric
      for (size_t k = 0; k < B * (N + 1); ++k) {
        if (k % (N + 1) == N) continue;
        double* H = lxx + k * ILQR_NX * ILQR_NX;
        for (int i = 0; i < ILQR_NX; ++i) for (int j = 16 * (i / 16 + 1); j < ILQR_NX; ++j) H[i * ILQR_NX + j] = H[j * ILQR_NX + i];
      }
    }
  }
  if (luu) HIPCHK(c, hipMemcpy(luu, c->S.luu, B * N * ILQR_NU * sizeof(double), hipMemcpyDeviceToHost));
  return ILQR_OK;
}
int ilqr_hip_set_quadratics(ilqr_hip_ctx* c, const double* lx, const double* lu, const double* lxx, const double* luu) {
  if (!c || !lx || !lu || !lxx || !luu) return ILQR_ERR_ARG; enter(c);
  const size_t B = c->B, N = c->N;
  HIPCHK(c, hipMemcpy(c->S.lx, lx, B * (N + 1) * ILQR_NX * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->S.lu, lu, B * N * ILQR_NU * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->S.lxx, lxx, B * (N + 1) * ILQR_NX * ILQR_NX * sizeof(double), hipMemcpyHostToDevice));
  c->lxx_layout = 0;
  HIPCHK(c, hipMemcpy(c->S.luu, luu, B * N * ILQR_NU * sizeof(double), hipMemcpyHostToDevice));
  return ILQR_OK;
}
int ilqr_hip_get_value_function(ilqr_hip_ctx* c, double* Vx, double* Vxx) {
  if (!c) return ILQR_ERR_ARG; enter(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (Vx) HIPCHK(c, hipMemcpy(Vx, c->S.Vx, (size_t)c->B * ILQR_NX * sizeof(double), hipMemcpyDeviceToHost));
  if (Vxx) HIPCHK(c, hipMemcpy(Vxx, c->S.Vxx, (size_t)c->B * ILQR_NX * ILQR_NX * sizeof(double), hipMemcpyDeviceToHost));
  return ILQR_OK;
}
int ilqr_hip_step_stance(ilqr_hip_ctx* c, int count, const double* x, const double* u, int stance_left, int stance_right, double* x_next);
int ilqr_hip_step(ilqr_hip_ctx* c, int count, const double* x, const double* u, double* x_next) {
  return ilqr_hip_step_stance(c, count, x, u, 1, 1, x_next);
}
int ilqr_hip_set_contact_mode(ilqr_hip_ctx* c, int mode, double softness) {
  if (!c || (mode != ILQR_CONTACT_NONE && mode != ILQR_CONTACT_RIGID_STANCE && mode != ILQR_CONTACT_UNILATERAL_STANCE && mode != ILQR_CONTACT_FRICTION_STANCE && mode != ILQR_CONTACT_KINETIC_FRICTION_STANCE)) return ILQR_ERR_ARG;
  enter(c);
  if (mode >= ILQR_CONTACT_FRICTION_STANCE && ilqr::variant_scalar_dyn()) { c->err = "contact modes 3 / 4 (Coulomb limit) exist on the two-lane kernels only; unset ILQR_DYN=s"; return ILQR_ERR_UNSUPPORTED; }
  c->P.dyn.contact = mode;
  if (softness > 0.0) c->P.dyn.soft = softness;
  return ILQR_OK;
}
int ilqr_hip_set_joint_limits(ilqr_hip_ctx* c, int on) {
  if (!c) return ILQR_ERR_ARG;
  enter(c);
  if (on && ilqr::variant_scalar_dyn()) { c->err = "joint-limit rows exist on the two-lane kernels only; unset ILQR_DYN=s"; return ILQR_ERR_UNSUPPORTED; }
  c->P.dyn.limits = on ? 1 : 0;     // (a nominal rolled under the other setting is recognised by same_dyn)
  return ILQR_OK;
}
int ilqr_hip_set_joint_limit_stiffness(ilqr_hip_ctx* c, double k) {
  if (!c || !(k >= 0.0) || !std::isfinite(k)) return ILQR_ERR_ARG;
  c->P.dyn.lim_k = k;          // (a nominal rolled under another stiffness is recognised by same_dyn)
  return ILQR_OK;
}
int ilqr_hip_set_friction(ilqr_hip_ctx* c, double mu) {
  if (!c || !(mu >= 0.0)) return ILQR_ERR_ARG;
  c->P.dyn.mu = mu;            // (a nominal rolled under another mu is recognised by same_dyn)
  return ILQR_OK;
}
int ilqr_hip_step_stance(ilqr_hip_ctx* c, int count, const double* x, const double* u, int stance_left, int stance_right, double* x_next) {
  if (!c || count <= 0 || !x || !u || !x_next) return ILQR_ERR_ARG;
  enter(c);
  if ((size_t)count > c->step_cap) {   // scratch owned by the context, grown on demand (the closed loop steps the plant every MPC step)
    if (c->d_stepx) hipFree(c->d_stepx);
    if (c->d_stepu) hipFree(c->d_stepu);
    if (c->d_stepn) hipFree(c->d_stepn);
    c->d_stepx = c->d_stepu = c->d_stepn = nullptr; c->step_cap = 0;
    HIPCHK(c, hipMalloc((void**)&c->d_stepx, (size_t)count * ILQR_NX * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&c->d_stepu, (size_t)count * ILQR_NU * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&c->d_stepn, (size_t)count * ILQR_NX * sizeof(double)));
    c->step_cap = (size_t)count;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_stepx, x, (size_t)count * ILQR_NX * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_stepu, u, (size_t)count * ILQR_NU * sizeof(double), hipMemcpyHostToDevice, c->stream));
  ilqr::launch_step(count, c->d_stepx, c->d_stepu, c->P.dyn, c->d_stepn, c->stream, stance_left, stance_right);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(x_next, c->d_stepn, (size_t)count * ILQR_NX * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ILQR_OK;
}
int ilqr_hip_get_iterations_enqueued(const ilqr_hip_ctx* c) { return c ? c->iterations_enqueued : -1; }
int ilqr_hip_get_adopt_mismatches(ilqr_hip_ctx* c, unsigned long long* count) {
  if (!c || !count) return ILQR_ERR_ARG;
  enter(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(count, c->d_mismatch, sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return ILQR_OK;
}

// ---------------------------------------------------------------- multi-GPU: the ONE collective of an MPC step
// RCCL (librccl.so.1) is opened on first use; a process that never shards never loads it.
namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
bool g_rccl_ok = false;
// Opened and resolved exactly once per process, whichever host thread gets there first (the multi-GPU C++ model is one thread
// and one handle per GPU, tests/cpp/cpp_multi_gpu_demo.cpp); the table is published only after every symbol has resolved.
bool rccl_load() {
  std::call_once(g_rccl_once, [] {
    Rccl R;
    const char* names[] = {getenv("ILQR_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) { if (n && (lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break; }
    if (!lib) { const char* de = dlerror(); g_rccl.err = std::string("cannot open librccl: ") + (de ? de : "?"); return; }
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) { ok = false; g_rccl.err = std::string("librccl lacks ") + n; } return p; };
    R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
    R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
    R.Send = (decltype(R.Send))sym("ncclSend");
    R.Recv = (decltype(R.Recv))sym("ncclRecv");
    R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
    R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
    R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(lib); return; }
    R.lib = lib;
    g_rccl = R;
    g_rccl_ok = true;
  });
  return g_rccl_ok;
}
}  // namespace
#define NCCLCHK(ctx, call)                                                                          \
  do {                                                                                              \
    ncclResult_t r_ = (call);                                                                       \
    if (r_ != ncclSuccess) { (ctx)->err = std::string(#call) + ": " + g_rccl.GetErrorString(r_); return ILQR_ERR_HIP; } \
  } while (0)

int ilqr_hip_payload_width(int with_gains) { return ILQR_NU + 1 + (with_gains ? ILQR_NU * ILQR_NX : 0); }
int ilqr_hip_comm_available(void) { return rccl_load() ? 1 : 0; }
int ilqr_hip_comm_get_unique_id(char* id) {
  if (!id) return ILQR_ERR_ARG;
  if (!rccl_load()) return ILQR_ERR_UNSUPPORTED;
  ncclUniqueId u;
  if (g_rccl.GetUniqueId(&u) != ncclSuccess) return ILQR_ERR_HIP;
  std::memcpy(id, u.internal, ILQR_COMM_ID_BYTES);
  return ILQR_OK;
}
int ilqr_hip_comm_init(ilqr_hip_ctx* c, int world, int rank, const char* id) {
  if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) return ILQR_ERR_ARG;
  if (c->comm) { c->err = "communicator already initialised"; return ILQR_ERR_STATE; }
  enter(c);
  c->world = world; c->rank = rank;
  if (world == 1) return ILQR_OK;           // one GPU: the gather degenerates to a device copy, RCCL is not needed
  if (!rccl_load()) { c->err = g_rccl.err; return ILQR_ERR_UNSUPPORTED; }
  ncclUniqueId u; std::memcpy(u.internal, id, ILQR_COMM_ID_BYTES);
  NCCLCHK(c, g_rccl.CommInitRank(&c->comm, world, u, rank));
  return ILQR_OK;
}
int ilqr_hip_comm_destroy(ilqr_hip_ctx* c) {
  if (!c) return ILQR_ERR_ARG;
  if (c->comm) { hipSetDevice(c->device); hipStreamSynchronize(c->stream); g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
  c->world = 1; c->rank = 0;
  return ILQR_OK;
}
int ilqr_hip_comm_world(const ilqr_hip_ctx* c) { return c ? c->world : -1; }
int ilqr_hip_comm_rank(const ilqr_hip_ctx* c) { return c ? c->rank : -1; }
int ilqr_hip_gather_first_knot(ilqr_hip_ctx* c, int root, int with_gains, double* recv_device) {
  if (!c || root < 0 || root >= c->world || (c->rank == root && !recv_device)) return ILQR_ERR_ARG;
  if (!c->initialized) return ILQR_ERR_STATE;
  enter(c);
  const size_t W = (size_t)ilqr_hip_payload_width(with_gains), cnt = (size_t)c->B * W;
  if (cnt > c->payload_cap) {
    if (c->d_payload) hipFree(c->d_payload);
    c->d_payload = nullptr; c->payload_cap = 0;
    HIPCHK(c, hipMalloc((void**)&c->d_payload, cnt * sizeof(double)));
    c->payload_cap = cnt;
  }
  // rank r's rollouts are rows [r B, (r + 1) B) of the gathered array: contiguous shards, global rollout order
  double* mine = (c->rank == root) ? recv_device + (size_t)c->rank * cnt : c->d_payload;
  ilqr::launch_pack_payload(c->S, with_gains, mine, c->stream);
  HIPCHK(c, hipGetLastError());
  if (c->world > 1) {
    if (!c->comm) { c->err = "gather before ilqr_hip_comm_init"; return ILQR_ERR_STATE; }
    // a gather as grouped point-to-point transfers: every peer sends to the root over its own xGMI link
    NCCLCHK(c, g_rccl.GroupStart());
    ncclResult_t gr = ncclSuccess;      // an error inside the group still closes it before the call returns
    if (c->rank == root) {
      for (int r = 0; r < c->world && gr == ncclSuccess; ++r) if (r != root) gr = g_rccl.Recv(recv_device + (size_t)r * cnt, cnt, ncclDouble, r, c->comm, c->stream);
    } else {
      gr = g_rccl.Send(c->d_payload, cnt, ncclDouble, root, c->comm, c->stream);
    }
    const ncclResult_t ge = g_rccl.GroupEnd();
    if (gr != ncclSuccess || ge != ncclSuccess) { c->err = std::string("RCCL gather (grouped send/recv): ") + g_rccl.GetErrorString(gr != ncclSuccess ? gr : ge); return ILQR_ERR_HIP; }
  }
  return ILQR_OK;   // asynchronous on the handle's stream: ilqr_hip_synchronize before reading recv_device
}

int ilqr_hip_enable_profiling(ilqr_hip_ctx* c, int on) { if (!c) return ILQR_ERR_ARG; c->profiling = on ? 1 : 0; return ILQR_OK; }
int ilqr_hip_set_profiled_stages(ilqr_hip_ctx* c, unsigned mask) { if (!c) return ILQR_ERR_ARG; c->prof_mask = mask & 0xFFu; return ILQR_OK; }
int ilqr_hip_get_stage_ms(ilqr_hip_ctx* c, double* ms, double* launches) {
  if (!c || !ms) return ILQR_ERR_ARG;
  for (int i = 0; i < 8; ++i) { ms[i] = c->stage_ms[i]; if (launches) launches[i] = c->stage_launches[i]; }
  return ILQR_OK;
}

int ilqr_hip_reference_kinematics(const double* x, double* com, double* ee) {
  if (!x || !com || !ee) return ILQR_ERR_ARG;
  h1host::reference_kinematics(x, com, ee);
  return ILQR_OK;
}
int ilqr_hip_reference_com_velocity(const double* x, double* comvel) {
  if (!x || !comvel) return ILQR_ERR_ARG;
  h1host::reference_com_velocity(x, comvel);
  return ILQR_OK;
}
int ilqr_hip_foot_clearance(const double* qpos, double* clearance) {
  if (!qpos || !clearance) return ILQR_ERR_ARG;
  h1host::foot_clearance(qpos, clearance);
  return ILQR_OK;
}
int ilqr_hip_gravity_compensation(const double* x, const double* gravity, double* u) {
  if (!x || !gravity || !u) return ILQR_ERR_ARG;
  h1host::gravity_compensation(x, gravity, u);
  return ILQR_OK;
}

}  // extern "C"
