/*
 * ilqr_hip.h -- C ABI of the MI355X-native batched iLQR solver (libilqr_hip.so).
 *
 * Drop-in boundary for the reference's MPC_iLQR_solve path.  The reference has no FFI; its boundary
 * is the C++ class API of `iLQR` / `MPC` / `RobotUtils`.  Every entry point below cites the
 * reference interface it replaces (paths relative to the reference repository root).
 * A batch of B independent rollouts (same robot, same horizon) is solved per handle; B = 1
 * reproduces the reference call-for-call (see include/ilqr_hip.hpp for the C++ mirror classes).
 *
 * Conventions: row-major doubles in caller-owned HOST buffers unless a name ends in `_device`;
 * state x = [qpos(26): p, quat wxyz, hinge(19) | qvel(25): v_lin world, omega body, hinge rates];
 * nx = 51, nu = 19; N = horizon.  "set" arrays may be shared by all rollouts (n_sets == 1) or
 * given per rollout (n_sets == batch).  All functions return ILQR_OK (0) or an error code;
 * no exceptions cross the ABI; calls on one handle must be serialised by the caller; one handle
 * per GPU.  There is NO CPU fallback: creating a handle without a HIP device fails.
 */
#ifndef ILQR_HIP_H
#define ILQR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define ILQR_NX 51
#define ILQR_NU 19
#define ILQR_NQ 26
#define ILQR_NV 25

enum ilqr_status {
  ILQR_OK = 0,
  ILQR_ERR_ARG = 1,         /* null pointer / bad size (reference: size check, src/ilqr/ilqr.cpp:526-532) */
  ILQR_ERR_HIP = 2,         /* HIP runtime error, see ilqr_hip_last_error */
  ILQR_ERR_NO_DEVICE = 3,   /* no usable gfx950 device */
  ILQR_ERR_STATE = 4,       /* call order violated (e.g. solve before initialize) */
  ILQR_ERR_UNSUPPORTED = 5
};

enum ilqr_jacobian_mode {
  ILQR_JAC_ANALYTIC = 0,    /* exact derivatives of the dynamics step (north-star mode) */
  ILQR_JAC_FD_FORWARD = 1   /* reference-style forward differences, RobotUtils::linearizeDynamicsFD,
                               src/common/robot_utils.cpp:120-160 (eps default 1e-5, robot_utils.hpp:51-53) */
};

typedef struct ilqr_hip_ctx ilqr_hip_ctx;

/* iLQR::iLQR(RobotUtils&, int N, double dt, urdf) -- include/ilqr/ilqr.hpp:19, src/ilqr/ilqr.cpp:14-48.
   The H1 model constants (h1.xml / h1.urdf) are compiled in.  device = HIP device ordinal. */
int ilqr_hip_create(ilqr_hip_ctx** out, int device, int batch, int horizon, double dt);
int ilqr_hip_destroy(ilqr_hip_ctx* ctx);
/* The diagnostic environment switches (kernel families ILQR_BACKWARD / ILQR_LS / ILQR_ROLLOUT / ILQR_DYN / ILQR_LINT, launch orders
   ILQR_SLICES / ILQR_STAGGER / ILQR_OVERLAP_ROLLOUT / ILQR_REUSE_ROLLOUT / ILQR_EE_GATE / ILQR_SPLIT / ILQR_SPEC*) are read ONCE, by
   ilqr_hip_create, and kept in the handle: no getenv on the call path.  This call re-reads them for one handle (tests, profiling
   tools); ILQR_ENV_PER_CALL=1 at creation makes every call of the handle do so.  No reference counterpart. */
int ilqr_hip_reload_environment(ilqr_hip_ctx* ctx);
/* Off by default (the solve then executes every pass the reference executes).  On: a lambda retry (ilqr.cpp:619-644) whose lambda is
   already saturated -- min(10 lambda, 1e-3) == lambda, the state a rollout reaches after a few failed searches -- is not executed:
   it would repeat the backward pass and the line search that have just failed on identical inputs, bit for bit, and fail again.  Its
   bookkeeping (trace entry, iteration count, convergence-exit rule of ilqr.cpp:640-655) is played at once.  Every observable of the solve
   is unchanged (GPU test); bench.py reports the resulting rate as its own object, never as the headline.  No reference counterpart:
   the reference recomputes. */
int ilqr_hip_set_dedup_saturated_retry(ilqr_hip_ctx* ctx, int on);
const char* ilqr_hip_last_error(const ilqr_hip_ctx* ctx);
int ilqr_hip_batch(const ilqr_hip_ctx* ctx);
int ilqr_hip_horizon(const ilqr_hip_ctx* ctx);
/* Number of contiguous batch slices a solve is enqueued as (each on its own streams, so that the line search of one
   slice overlaps with the Riccati / Jacobian kernels of the others; environment ILQR_SLICES, 1 = one launch sequence
   for the whole batch).  No reference counterpart: the reference solves one trajectory at a time. */
int ilqr_hip_num_slices(const ilqr_hip_ctx* ctx);

/* RobotUtils::setCostWeights -- include/common/robot_utils.hpp:60, src/common/robot_utils.cpp:253-279.
   Q/R/Qf are diagonal by construction (Config::buildCostMatrices, src/common/config.cpp:66-122). */
int ilqr_hip_set_cost_weights(ilqr_hip_ctx* ctx, const double* Q_diag /*51*/, const double* R_diag /*19*/, const double* Qf_diag /*51*/);
/* RobotUtils::set{CoM,CoMVel,EEPos,EEVel,Upright,Balance}Weight -- include/common/robot_utils.hpp:120-129 */
int ilqr_hip_set_task_weights(ilqr_hip_ctx* ctx, double w_com, double w_com_vel, double w_ee_pos, double w_ee_vel, double w_upright, double w_balance);
/* RobotUtils::setConstraintWeights -- src/common/robot_utils.cpp:674-680 */
int ilqr_hip_set_constraint_weights(ilqr_hip_ctx* ctx, double w_joint_limits, double w_control_limits);
/* RobotUtils::setGravity -- src/common/robot_utils.cpp:782-789 */
int ilqr_hip_set_gravity(ilqr_hip_ctx* ctx, double gx, double gy, double gz);
/* RobotUtils::loadContactSchedule / isStance -- src/common/robot_utils.cpp:445-504; horizon-local rows 0..N
   (the reference indexes the schedule with the horizon-local t). stance[n_sets][N+1][2] */
int ilqr_hip_set_contact_schedule(ilqr_hip_ctx* ctx, const int* stance, int n_sets);
/* RobotUtils::getEEReference / getCoMVelReference -- src/common/robot_utils.cpp:525-549.
   ee_ref[n_sets][N+1][2][3] (left, right ankle), com_vel_ref[n_sets][N+1][3] (may be NULL -> zeros) */
int ilqr_hip_set_ee_references(ilqr_hip_ctx* ctx, const double* ee_ref, const double* com_vel_ref, int n_sets);
/* reference window handed to solve(): x_ref[n_sets][N+1][51], u_ref[n_sets][N][19], com_ref[n_sets][N+1][3]
   (MPC::extractReferenceWindow, src/ilqr/mpc.cpp:163-166) */
int ilqr_hip_set_references(ilqr_hip_ctx* ctx, const double* x_ref, const double* u_ref, const double* com_ref, int n_sets);

/* iLQR::setRegularization / setMaxIterations / setTolerance -- include/ilqr/ilqr.hpp:22-24.
   setRegularization resets every rollout's lambda (lambda persists across solves, ilqr.hpp:54). */
int ilqr_hip_set_regularization(ilqr_hip_ctx* ctx, double lambda);
int ilqr_hip_set_max_iterations(ilqr_hip_ctx* ctx, int max_iter);
int ilqr_hip_set_tolerance(ilqr_hip_ctx* ctx, double tol);
/* build-specific options: Jacobian mode, FD step, early_exit (0 = run exactly max_iter iterations) */
int ilqr_hip_set_options(ilqr_hip_ctx* ctx, int jacobian_mode, double fd_eps, int early_exit);

/* iLQR::initializeWithReference -- include/ilqr/ilqr.hpp:40-45, src/ilqr/ilqr.cpp:50-117.
   x0[B][51]; cold start: u_init[B][N][19] or NULL (gravity compensation, RobotUtils::computeGravComp,
   src/common/robot_utils.cpp:844-866 with the correct dof index) followed by N rollouts;
   warm start: prev_xbar[B][N+1][51], prev_ubar[B][N][19] shifted by one knot (ilqr.cpp:68-80). */
int ilqr_hip_initialize(ilqr_hip_ctx* ctx, const double* x0, const double* u_init, const double* prev_xbar, const double* prev_ubar);
/* warm start from the solver's own previous solution kept on the device (MPC::stepOnce, src/ilqr/mpc.cpp:58-60) */
int ilqr_hip_initialize_warm_resident(ilqr_hip_ctx* ctx, const double* x0);
/* device-resident variant of the cold start: x0_device[B][51], u_init_device[B][N][19] are HIP device pointers */
int ilqr_hip_initialize_device(ilqr_hip_ctx* ctx, const double* x0_device, const double* u_init_device);

/* iLQR::solve -- include/ilqr/ilqr.hpp:27-31, src/ilqr/ilqr.cpp:521-660.  References are the ones last set
   with ilqr_hip_set_references; x0[B][51] (host) or NULL to reuse the x0 given to initialize.
   cost_out[B] may be NULL.  Runs asynchronously on the handle's stream and synchronises before returning. */
int ilqr_hip_solve(ilqr_hip_ctx* ctx, const double* x0, double* cost_out);
/* enqueue only (nothing copied back); pair with ilqr_hip_synchronize.
   With the reference's convergence exit on (ilqr_hip_set_options early_exit = 1, the default; ilqr.cpp:645-655) AND the
   early-exit gate on (ilqr_hip_set_early_exit_gate, default on) this call BLOCKS the host while it enqueues: it launches
   iteration i only after it has seen how many rollouts were still active after iteration i - 2, and stops once that count is
   zero.  The device never waits (one full iteration stays queued), but a host thread that drives several handles, or overlaps
   its own work with the solve, should turn the gate off for a handle: the call then enqueues all max_iter iterations and
   returns at once (converged rollouts are masked out on the device, results are identical). */
int ilqr_hip_solve_async(ilqr_hip_ctx* ctx);
/* per-handle switch of the early-exit gate described above (1 = on, the default).  The environment variable ILQR_EE_GATE,
   when set, overrides it for every handle of the process (diagnostics). */
int ilqr_hip_set_early_exit_gate(ilqr_hip_ctx* ctx, int on);
int ilqr_hip_synchronize(ilqr_hip_ctx* ctx);

/* accessors: iLQR::xbar/ubar/gainsK/gainsKff -- include/ilqr/ilqr.hpp:34-37 */
int ilqr_hip_get_xbar(ilqr_hip_ctx* ctx, double* xbar /*[B][N+1][51]*/);
int ilqr_hip_get_ubar(ilqr_hip_ctx* ctx, double* ubar /*[B][N][19]*/);
int ilqr_hip_get_gains_K(ilqr_hip_ctx* ctx, double* K /*[B][N][19][51]*/);
int ilqr_hip_get_gains_kff(ilqr_hip_ctx* ctx, double* kff /*[B][N][19]*/);
int ilqr_hip_get_cost(ilqr_hip_ctx* ctx, double* cost /*[B]*/);
int ilqr_hip_get_iterations(ilqr_hip_ctx* ctx, int* iters /*[B]*/);
int ilqr_hip_get_lambda(ilqr_hip_ctx* ctx, double* lambda /*[B]*/);
/* per-iteration trace (parity artefact; the reference keeps these internal):
   cost[B][max_iter+1] (entry 0 = initial cost), alpha[B][max_iter] (0 = no step), lambda[B][max_iter] */
int ilqr_hip_get_trace(ilqr_hip_ctx* ctx, double* cost, double* alpha, double* lambda);
/* first-knot results gathered per MPC step: u0[B][19], K0[B][19][51] (either may be NULL); device pointers */
int ilqr_hip_first_knot_device(ilqr_hip_ctx* ctx, const double** u0_device, const double** K0_device, const double** cost_device);
/* same payload written into CALLER-owned device buffers (the send buffer of the per-step RCCL gather):
   u0_out[B][19], K0_out[B][19][51] (nullable), cost_out[B] (nullable); synchronises the handle's stream */
int ilqr_hip_pack_first_knot_device(ilqr_hip_ctx* ctx, double* u0_out_device, double* K0_out_device, double* cost_out_device);

/* ---- multi-GPU: one handle per GPU, one process or thread per handle.  The reference is single
   process; its consumer of the result is MPC::stepOnce (src/ilqr/mpc.cpp:97-113: u_apply from ubar[0], K[0]).  The global
   batch is cut into contiguous shards, rank r owning rollouts [r B, (r + 1) B); the only exchange is ONE gather per MPC
   step of the first-knot payload row [u0(19) | cost | K0(19 x 51) if with_gains] of every rollout to `root`, in global
   rollout order -- RCCL grouped send/recv over xGMI on the handle's stream.  librccl is opened on first use.
     rank 0: ilqr_hip_comm_get_unique_id(id) -> hand the 128 bytes to every rank (file, socket, MPI, torch.distributed ...)
     all   : ilqr_hip_comm_init(ctx, world, rank, id)        (world == 1: no RCCL, the gather is a device copy)
     step  : solve ...; ilqr_hip_gather_first_knot(ctx, root, with_gains, recv); ilqr_hip_synchronize(ctx)
   recv_device (root only, else NULL): device buffer [world * B][ilqr_hip_payload_width(with_gains)]. */
#define ILQR_COMM_ID_BYTES 128
int ilqr_hip_payload_width(int with_gains);
/* 1 if librccl can be opened and resolves every entry point the gather needs, else 0.  ncclCommInitRank is collective: a rank
   that cannot load the library would leave its peers waiting inside ilqr_hip_comm_init, so a launcher lets every rank check
   this (and agree on the outcome) BEFORE any rank calls ilqr_hip_comm_init with world > 1 (bench.py does). */
int ilqr_hip_comm_available(void);
int ilqr_hip_comm_get_unique_id(char* id /*[ILQR_COMM_ID_BYTES]*/);
int ilqr_hip_comm_init(ilqr_hip_ctx* ctx, int world, int rank, const char* id /*[ILQR_COMM_ID_BYTES], may be NULL when world == 1*/);
int ilqr_hip_comm_destroy(ilqr_hip_ctx* ctx);
int ilqr_hip_comm_world(const ilqr_hip_ctx* ctx);
int ilqr_hip_comm_rank(const ilqr_hip_ctx* ctx);
int ilqr_hip_gather_first_knot(ilqr_hip_ctx* ctx, int root, int with_gains, double* recv_device);

/* MPC::stepOnce control law u = ubar[0] + K[0](x_meas - xbar[0]) -- src/ilqr/mpc.cpp:97-101 */
int ilqr_hip_compute_control(ilqr_hip_ctx* ctx, const double* x_measured /*[B][51]*/, double* u_apply /*[B][19]*/);

/* ---- stage entry points (one reference function each; used by the parity tests and the bench breakdown) ---- */
int ilqr_hip_set_trajectory(ilqr_hip_ctx* ctx, const double* xbar, const double* ubar);   /* overwrite nominal trajectory */
int ilqr_hip_stage_rollout(ilqr_hip_ctx* ctx);          /* iLQR::forwardRolloutNominal, src/ilqr/ilqr.cpp:119-124 (+ total cost) */
int ilqr_hip_stage_linearize(ilqr_hip_ctx* ctx);        /* iLQR::computeLinearization, src/ilqr/ilqr.cpp:126-131 */
int ilqr_hip_stage_cost_quadratics(ilqr_hip_ctx* ctx);  /* iLQR::computeCostQuadratics, src/ilqr/ilqr.cpp:133-244 */
int ilqr_hip_stage_backward_pass(ilqr_hip_ctx* ctx);    /* iLQR::backwardPass, src/ilqr/ilqr.cpp:250-309 */
int ilqr_hip_stage_line_search(ilqr_hip_ctx* ctx, int* improved /*[B]*/, double* new_cost /*[B]*/, double* alpha /*[B]*/); /* ilqr.cpp:311-361 */
int ilqr_hip_stage_total_cost(ilqr_hip_ctx* ctx, double* cost /*[B]*/);  /* iLQR::computeTotalCost, src/ilqr/ilqr.cpp:363-518 */
int ilqr_hip_get_linearization(ilqr_hip_ctx* ctx, double* A /*[B][N][51][51]*/, double* B /*[B][N][51][19]*/);
int ilqr_hip_set_linearization(ilqr_hip_ctx* ctx, const double* A, const double* B);
int ilqr_hip_get_quadratics(ilqr_hip_ctx* ctx, double* lx /*[B][N+1][51]*/, double* lu /*[B][N][19]*/, double* lxx /*[B][N+1][51][51]*/, double* luu_diag /*[B][N][19]*/);
int ilqr_hip_set_quadratics(ilqr_hip_ctx* ctx, const double* lx, const double* lu, const double* lxx, const double* luu_diag);
int ilqr_hip_get_value_function(ilqr_hip_ctx* ctx, double* Vx /*[B][51]*/, double* Vxx /*[B][51][51]*/); /* at knot 0 after the backward pass */
/* one dynamics step for arbitrary (x,u) pairs: RobotUtils::rolloutOneStep, src/common/robot_utils.cpp:106-117 */
int ilqr_hip_step(ilqr_hip_ctx* ctx, int count, const double* x /*[count][51]*/, const double* u /*[count][19]*/, double* x_next /*[count][51]*/);

/* Contact row.  The reference's plant is MuJoCo with floor contacts (mj_step inside
   RobotUtils::rolloutOneStep, src/common/robot_utils.cpp:106-117).  ILQR_CONTACT_RIGID_STANCE restates the regime its
   scenarios run in: a foot the contact schedule (ilqr_hip_set_contact_schedule, horizon-local rows) marks as stance does
   not move -- velocity-level constraint on the ankle link over one step, solved through the articulated-body
   quantities; `softness` (> 0: set, <= 0: keep, default 1e-5 / kg) regularises the constraint-space inertia.  Rollout, line
   search, warm-start step and plant run on the two-lane register / LDS kernels with the constraint solve in them; the
   Jacobians are analytic (ILQR_JAC_ANALYTIC: derivative of the constrained step with the active set held fixed) or the
   reference's forward differences (ILQR_JAC_FD_FORWARD, robot_utils.cpp:120-160), as ilqr_hip_set_options selects.
   ilqr_hip_step_stance: one step with explicit stance flags (the flags only matter in contact mode).
   ILQR_CONTACT_UNILATERAL_STANCE: the same constraint, but the floor only pushes: a scheduled stance foot whose constraint
   force has a negative component along the world up axis is released for that step and the remaining set solved again.
   ILQR_CONTACT_FRICTION_STANCE: unilateral, and sticking is limited by Coulomb friction, the other half of what MuJoCo's floor
   contacts do inside mj_step: a foot whose constraint force leaves the cone |f_t| <= mu f_n (f_n along the world up axis) slides --
   its two tangential translation rows are dropped (rotation and normal rows stay, no tangential force on a sliding foot) and
   the set is solved again, once.  mu: ilqr_hip_set_friction (default 1, MuJoCo's default sliding friction; the reference's
   robots/h1_description/mjcf model sets none).  Jacobians in this mode: the reference's forward differences, or analytic
   (decisions held fixed; a sliding foot's normal row and normal force turn with the foot: that term is carried).
   ILQR_CONTACT_KINETIC_FRICTION_STANCE: the same decision, but the sliding foot keeps kinetic friction: a tangential force mu f_n along
   the direction in which the sticking solution pulled (the one that opposes the slip); its normal multiplier then acts along
   up + mu t while the constraint row stays the normal one -- an unsymmetric 12 x 12 system, Gaussian elimination with partial
   pivoting for the knots where a foot slides.  Jacobians as mode 3; the analytic ones also carry the tangent of the sticking
   solve, which the friction direction t follows. */
enum ilqr_contact_mode { ILQR_CONTACT_NONE = 0, ILQR_CONTACT_RIGID_STANCE = 1, ILQR_CONTACT_UNILATERAL_STANCE = 2, ILQR_CONTACT_FRICTION_STANCE = 3,
                         ILQR_CONTACT_KINETIC_FRICTION_STANCE = 4 };
int ilqr_hip_set_contact_mode(ilqr_hip_ctx* ctx, int mode, double softness);
int ilqr_hip_set_friction(ilqr_hip_ctx* ctx, double mu);
/* Joint-limit rows of the plant: the reference's plant is mj_step (src/common/robot_utils.cpp:106-117), which
   enforces the hinge ranges of robots/h1_description/mjcf/h1.xml (jnt_range, :55-151) as constraints that are active only when violated;
   the cost side only carries the soft penalty (RobotUtils::constraintCost, robot_utils.cpp:615-672).  Restated here as the rigid,
   velocity-level limit of that constraint, like the stance rows: a hinge past its range that the step would still move outward
   (v_i + h qacc_i points out of the range, qacc of the step without these rows) is stopped over the step, v_i+ = 0 -- its acceleration
   is prescribed, qacc_i = -v_i / h, in a second pass of the articulated-body recursion (Featherstone's hybrid dynamics; the stance
   rows of the contact modes are solved on that system).  A hinge past its range that moves back in is left alone; nothing pushes a
   hinge back unless ilqr_hip_set_joint_limit_stiffness gives the rows a restoring term.  Default off (the constraint-free restatement).
   Rollout, line search, plant step and both Jacobian schemes carry it (two-lane kernels, in instantiations of their own: with the
   option off every kernel keeps its machine code).  Analytic Jacobians: the dumped recursion has the stopped hinges
   acceleration-prescribed and d qacc_i = -1 / h rides the direction of a stopped hinge's own rate (decisions held fixed). */
int ilqr_hip_set_joint_limits(ilqr_hip_ctx* ctx, int on);
/* Restoring stiffness of the joint-limit rows (round 6; mj_step pushes a hinge back into its range, h1.xml:55-151 / robot_utils.cpp:113-114, the
   pure stop above does not).  MuJoCo drives a violated constraint towards the reference acceleration a_ref = -b v - k r (solref: b = 2 /
   (dmax timeconst), k = 1 / (dmax^2 timeconst^2 dampratio^2), timeconst clamped to 2 h); in the hard limit of its impedance the constrained
   hinge takes exactly that acceleration.  With k = stiffness the row prescribes qacc_i = -v_i / h - k r_i (r_i = q_i - hi_i > 0 or q_i - lo_i < 0:
   the violation; b = 1 / h is solref's damping at the clamped time constant), i.e. v_i+ = -h k r_i, and it is active when the step without the
   rows falls short of that acceleration on the outward side -- a hinge that drifts back in too slowly is constrained too, one that returns
   faster is left alone (the row only pushes).  k = 1 / (2 h)^2 (625 at h = 0.02) is solref's default time constant: a quarter of the violation
   per step.  0 (default): the pure stop, bit for bit.  Needs ilqr_hip_set_joint_limits(ctx, 1); carried by the step, rollout, line search and
   both Jacobian schemes (analytic: d qacc_i = -k rides the direction of the constrained hinge's own angle).  k < 0: ILQR_ERR_ARG. */
int ilqr_hip_set_joint_limit_stiffness(ilqr_hip_ctx* ctx, double stiffness);
int ilqr_hip_step_stance(ilqr_hip_ctx* ctx, int count, const double* x, const double* u, int stance_left, int stance_right, double* x_next);

/* per-stage device time of the last solve in milliseconds, keyed like the reference's profiler
   (src/ilqr/ilqr.cpp:537-639): 0 computeCost/rollout, 1 linearization, 2 costQuadratics, 3 backwardPass,
   4 lineSearch, 5 control, 6 backwardPass (lambda-retry launch), 7 lineSearch (lambda-retry launch);
   requires ilqr_hip_enable_profiling(ctx, 1) before the solve */
int ilqr_hip_enable_profiling(ilqr_hip_ctx* ctx, int on);
/* Restrict the event pairs to the stages whose bit (stage index as above) is set in `mask` (default 0xFF: all).  Every timed
   launch costs two event records on its stream: all eight stages together add 1.3 ms to a 95 ms solve at B = 4096. */
int ilqr_hip_set_profiled_stages(ilqr_hip_ctx* ctx, unsigned mask);
/* Diagnostic of the concurrent nominal re-rollout (iterations >= 1 of a solve roll the nominal trajectory beside the
   linearisation, ilqr.cpp:563 vs :576): number of trajectory elements of the last solve in which the re-rolled trajectory
   differed bit-wise from the one the linearisation saw.  0 means the launch order is equivalent to the reference's. */
int ilqr_hip_get_adopt_mismatches(ilqr_hip_ctx* ctx, unsigned long long* count);
int ilqr_hip_get_stage_ms(ilqr_hip_ctx* ctx, double* ms /*[8]*/, double* launches /*[8]*/);
/* Iterations whose kernels the last solve enqueued: max_iterations, or fewer when the convergence exit (ilqr.cpp:645-655,
   ilqr_hip_set_options early_exit) is on and every rollout of the batch had left the loop -- the host follows the device-side
   count of active rollouts one iteration behind and stops launching (ilqr_hip_set_early_exit_gate(ctx, 0) turns that off).
   Returns the count, or -1 for a null handle. */
int ilqr_hip_get_iterations_enqueued(const ilqr_hip_ctx* ctx);
/* Iterations of the last solve that ran the lambda retry of ilqr.cpp:619-644 speculatively: while a pass holds at most 512
   rollouts (the whole batch, or -- convergence exit -- the count of rollouts still active the host has seen), the Riccati pass and
   the line search for lambda and for min(10 lambda, 1e-3) run side by side on two streams and the bookkeeping of :619-655 is
   played once with both outcomes known; results (gains, value function, trajectory, lambda, trace) are those of the sequential
   order, one pass of latency sooner.  Counted: the iterations that ENQUEUED the side-by-side passes -- while the host's count (one iteration
   old) is between 512 and 2048 both orders are enqueued and the device takes one by the length of the work list (ILQR_SPEC_DUAL=0: host
   count alone).  Environment ILQR_SPEC=0 keeps the sequential order.  Returns the count, -1 for a null handle. */
int ilqr_hip_get_speculative_iterations(const ilqr_hip_ctx* ctx);
/* Iterations of the last solve whose concurrent region (linearisation, cost quadratics, nominal re-rollout: ilqr.cpp:551-588) ran in
   two groups: the rollouts whose first line search of the previous iteration accepted a step start right behind that iteration's first
   bookkeeping pass, beside the lambda retry (:619-644) of the others, which follow behind the second; both meet at the backward pass.
   Bit-identical results.  On by default when the convergence exit is enabled (where it pays; with a fixed iteration count nearly every
   rollout retries and the early group is small); environment ILQR_SPLIT=0 / 1 forces it off / on.  Returns the count, -1 for a null handle. */
int ilqr_hip_get_split_iterations(const ilqr_hip_ctx* ctx);

/* ---- host-side model helpers (no GPU needed) ---- */
/* reference construction as RobotUtils::loadReferences does it (src/common/robot_utils.cpp:369-403):
   whole-body CoM (MuJoCo masses) and world positions of the two ankle bodies */
int ilqr_hip_reference_kinematics(const double* x /*51*/, double* com /*3*/, double* ee /*[2][3]*/);
/* CoM-velocity reference of the same loader (mj_jacSubtreeCom(root) * qvel, src/common/robot_utils.cpp:383-391) */
int ilqr_hip_reference_com_velocity(const double* x /*51*/, double* comvel /*3*/);
/* offline contact-schedule tool (get_contacts.py:96-147): height of the lowest point of each foot's collision hull
   (ankle-link mesh, h1.xml:81,116) above the floor plane z = 0 for the configuration qpos; the tool's stance flag is
   clearance < 0 (MuJoCo reports a floor contact, margin 0).  clearance[0] left, [1] right */
int ilqr_hip_foot_clearance(const double* qpos /*26*/, double* clearance /*2*/);
/* RobotUtils::computeGravComp (src/common/robot_utils.cpp:844-866, correct dof index): qfrc_bias[6+i] at v = 0 */
int ilqr_hip_gravity_compensation(const double* x /*51*/, const double* gravity /*3*/, double* u /*19*/);

/* the HIP stream the handle launches on (as void*), for event timing by the caller */
void* ilqr_hip_stream(ilqr_hip_ctx* ctx);
/* name/duration of the dominant kernel of the last profiled solve are reported by bench.py via hipEvents */

#ifdef __cplusplus
}
#endif
#endif /* ILQR_HIP_H */
