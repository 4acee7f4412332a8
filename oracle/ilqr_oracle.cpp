// TEST INFRASTRUCTURE (oracle) -- CPU restatement of the reference's iLQR solver; not shipped.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// PARITY UNPINNED: the reference (/root/reference) has no tests, golden vectors or fixtures for
// this path and cannot be built here (MuJoCo, Pinocchio, CasADi, Eigen, yaml-cpp absent).  The
// restatement is pinned by (a) a NumPy restatement of the Riccati recursion, (b) torch-autograd
// derivatives of independently written cost terms, (c) physics identities and an independent
// Kane-method residual for the dynamics -- see tests/ and tests/golden/.
//
// Restates, line by line in control flow:
//   iLQR::initializeWithReference   /root/reference/src/ilqr/ilqr.cpp:50-117
//   iLQR::forwardRolloutNominal     /root/reference/src/ilqr/ilqr.cpp:119-124
//   iLQR::computeLinearization      /root/reference/src/ilqr/ilqr.cpp:126-131 (+ robot_utils.cpp:120-160)
//   iLQR::computeCostQuadratics     /root/reference/src/ilqr/ilqr.cpp:133-244 (h1_costs.hpp)
//   iLQR::backwardPass              /root/reference/src/ilqr/ilqr.cpp:250-309
//   iLQR::forwardPassLineSearch     /root/reference/src/ilqr/ilqr.cpp:311-361
//   iLQR::solve                     /root/reference/src/ilqr/ilqr.cpp:521-660
//   MPC::stepOnce control law       /root/reference/src/ilqr/mpc.cpp:97-101
//   Config::buildCostMatrices       /root/reference/src/common/config.cpp:66-122
//   RobotUtils::computeGravComp     /root/reference/src/common/robot_utils.cpp:844-866 (correct dof index,
//                                   SURVEY.md Appendix D #8)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "h1_costs.hpp"

#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

enum JacMode { JAC_ANALYTIC = 0, JAC_FD = 1 };

struct Solver {
  Problem P;
  int N;
  double lambda = 1e-6;
  int max_iter = 10;
  double tol = 1e-4;
  int jac_mode = JAC_ANALYTIC;
  double fd_eps = 1e-5;
  int quad_mode = QUAD_CLOSED;
  int early_exit = 1;  // 0: run exactly max_iter iterations (bench headline mode)
  std::vector<double> xbar, ubar, K, kff, A, B, lx, lu, lxx, luu;
  std::vector<double> Vx_last, Vxx_last;
  // trace (parity artefact): cost after each iteration, accepted alpha (0 if none), lambda used
  std::vector<double> trace_cost, trace_alpha, trace_lambda;
  int iters_done = 0;
  long backward_passes = 0;

  explicit Solver(int N_, double h) : N(N_) {
    P.N = N_; P.dyn.h = h;
    for (int i = 0; i < H1_NX; ++i) { P.Q[i] = 1.0; P.Qf[i] = 1.0; }
    for (int i = 0; i < H1_NU; ++i) P.R[i] = 1.0;
    P.x_ref.assign((N + 1) * H1_NX, 0.0); P.u_ref.assign(N * H1_NU, 0.0); P.com_ref.assign((N + 1) * 3, 0.0);
    P.stance.assign((N + 1) * 2, 1); P.ee_ref.assign((N + 1) * 6, 0.0); P.com_vel_ref.assign((N + 1) * 3, 0.0);
    xbar.assign((N + 1) * H1_NX, 0.0); ubar.assign(N * H1_NU, 0.0);
    K.assign(N * H1_NU * H1_NX, 0.0); kff.assign(N * H1_NU, 0.0);
    A.assign(N * H1_NX * H1_NX, 0.0); B.assign(N * H1_NX * H1_NU, 0.0);
    lx.assign((N + 1) * H1_NX, 0.0); lu.assign(N * H1_NU, 0.0);
    lxx.assign((N + 1) * H1_NX * H1_NX, 0.0); luu.assign(N * H1_NU, 0.0);
    Vx_last.assign(H1_NX, 0.0); Vxx_last.assign(H1_NX * H1_NX, 0.0);
  }

  // t = horizon-local knot index of the state being stepped: selects the stance flags in contact mode (f4)
  void step(const double* x, const double* u, double* xn, int t) const { h1_step<double>(x, u, P.dyn, xn, &P.stance[2 * t]); }

  void rollout_nominal() { for (int t = 0; t < N; ++t) step(&xbar[t * H1_NX], &ubar[t * H1_NU], &xbar[(t + 1) * H1_NX], t); }

  void linearize_knot(int t, const double* x, const double* u, double* At, double* Bt) const {
    if (jac_mode == JAC_ANALYTIC) {
      typedef D1<H1_NX + H1_NU> T;
      std::vector<T> xs(H1_NX), us(H1_NU), xn(H1_NX);
      for (int i = 0; i < H1_NX; ++i) xs[i] = T::var(x[i], i);
      for (int i = 0; i < H1_NU; ++i) us[i] = T::var(u[i], H1_NX + i);
      h1_step<T>(xs.data(), us.data(), P.dyn, xn.data(), &P.stance[2 * t]);
      for (int i = 0; i < H1_NX; ++i) {
        for (int j = 0; j < H1_NX; ++j) At[i * H1_NX + j] = xn[i].g[j];
        for (int j = 0; j < H1_NU; ++j) Bt[i * H1_NU + j] = xn[i].g[H1_NX + j];
      }
    } else {  // robot_utils.cpp:120-160 verbatim: forward differences on raw coordinates
      double base[H1_NX], pert[H1_NX], xp[H1_NX], up[H1_NU];
      step(x, u, base, t);
      for (int j = 0; j < H1_NX; ++j) {
        std::memcpy(xp, x, sizeof(xp)); xp[j] += fd_eps; step(xp, u, pert, t);
        for (int i = 0; i < H1_NX; ++i) At[i * H1_NX + j] = (pert[i] - base[i]) / fd_eps;
      }
      for (int j = 0; j < H1_NU; ++j) {
        std::memcpy(up, u, sizeof(up)); up[j] += fd_eps; step(x, up, pert, t);
        for (int i = 0; i < H1_NX; ++i) Bt[i * H1_NU + j] = (pert[i] - base[i]) / fd_eps;
      }
    }
  }
  void linearize() { for (int t = 0; t < N; ++t) linearize_knot(t, &xbar[t * H1_NX], &ubar[t * H1_NU], &A[t * H1_NX * H1_NX], &B[t * H1_NX * H1_NU]); }

  void cost_quadratics() {
    for (int t = 0; t <= N; ++t)
      cost_quadratics_knot(P, t, &xbar[t * H1_NX], t < N ? &ubar[t * H1_NU] : nullptr, &lx[t * H1_NX], t < N ? &lu[t * H1_NU] : nullptr,
                           &lxx[t * H1_NX * H1_NX], t < N ? &luu[t * H1_NU] : nullptr, (QuadMode)quad_mode);
  }

  // ilqr.cpp:250-309
  void backward_pass() {
    const int n = H1_NX, m = H1_NU;
    ++backward_passes;
    std::vector<double> Vx(&lx[N * n], &lx[N * n] + n), Vxx(&lxx[N * n * n], &lxx[N * n * n] + n * n);
    std::vector<double> W(n * n), G(n * m), Qx(n), Qu(m), Qxx(n * n), Quu(m * m), Qxu(n * m), L(m * m), QuuK(m * n), T1(n * n);
    for (int t = N - 1; t >= 0; --t) {
      const double* At = &A[t * n * n]; const double* Bt = &B[t * n * m];
      double* Kt = &K[t * m * n]; double* kt = &kff[t * m];
      for (int i = 0; i < n; ++i) { double s = 0; for (int k = 0; k < n; ++k) s += At[k * n + i] * Vx[k]; Qx[i] = lx[t * n + i] + s; }
      for (int i = 0; i < m; ++i) { double s = 0; for (int k = 0; k < n; ++k) s += Bt[k * m + i] * Vx[k]; Qu[i] = lu[t * m + i] + s; }
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += Vxx[i * n + k] * At[k * n + j]; W[i * n + j] = s; }
      for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += Vxx[i * n + k] * Bt[k * m + j]; G[i * m + j] = s; }
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += At[k * n + i] * W[k * n + j]; Qxx[i * n + j] = lxx[t * n * n + i * n + j] + s; }
      for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += Bt[k * m + i] * G[k * m + j]; Quu[i * m + j] = (i == j ? luu[t * m + i] : 0.0) + s; }
      for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += At[k * n + i] * G[k * m + j]; Qxu[i * m + j] = s; }  // lxu == 0
      for (int i = 0; i < m; ++i) Quu[i * m + i] += lambda;
      use_lu_ = false;
      if (!cholesky(Quu.data(), L.data(), m)) { for (int i = 0; i < m; ++i) Quu[i * m + i] += 1e-4; if (!cholesky(Quu.data(), L.data(), m)) ldl_fallback(Quu.data(), L.data(), m); }
      // K = -Quu^-1 Qxu^T ; k = -Quu^-1 Qu
      std::vector<double> rhs(m), sol(m);
      for (int j = 0; j < n; ++j) { for (int i = 0; i < m; ++i) rhs[i] = Qxu[j * m + i]; solve_factored(L.data(), rhs.data(), sol.data(), m); for (int i = 0; i < m; ++i) Kt[i * n + j] = -sol[i]; }
      solve_factored(L.data(), Qu.data(), sol.data(), m); for (int i = 0; i < m; ++i) kt[i] = -sol[i];
      // Vx = Qx + K^T Quu k + K^T Qu + Qxu k
      std::vector<double> Quuk(m), nVx(n);
      for (int i = 0; i < m; ++i) { double s = 0; for (int j = 0; j < m; ++j) s += Quu[i * m + j] * kt[j]; Quuk[i] = s; }
      for (int i = 0; i < n; ++i) { double s1 = 0, s2 = 0, s3 = 0; for (int a = 0; a < m; ++a) { s1 += Kt[a * n + i] * Quuk[a]; s2 += Kt[a * n + i] * Qu[a]; s3 += Qxu[i * m + a] * kt[a]; } nVx[i] = Qx[i] + s1 + s2 + s3; }
      // Vxx = Qxx + K^T Quu K + K^T Qxu^T + Qxu K ; symmetrise
      for (int a = 0; a < m; ++a) for (int j = 0; j < n; ++j) { double s = 0; for (int b = 0; b < m; ++b) s += Quu[a * m + b] * Kt[b * n + j]; QuuK[a * n + j] = s; }
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s1 = 0, s2 = 0, s3 = 0; for (int a = 0; a < m; ++a) { s1 += Kt[a * n + i] * QuuK[a * n + j]; s2 += Kt[a * n + i] * Qxu[j * m + a]; s3 += Qxu[i * m + a] * Kt[a * n + j]; } T1[i * n + j] = Qxx[i * n + j] + s1 + s2 + s3; }
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Vxx[i * n + j] = 0.5 * (T1[i * n + j] + T1[j * n + i]);
      Vx = nVx;
    }
    Vx_last = Vx; Vxx_last = Vxx;
  }
  // LLT; returns false if a pivot is not positive (Eigen::LLT NumericalIssue)
  static bool cholesky(const double* M, double* L, int m) {
    for (int i = 0; i < m * m; ++i) L[i] = 0.0;
    for (int j = 0; j < m; ++j) {
      double s = M[j * m + j]; for (int k = 0; k < j; ++k) s -= L[j * m + k] * L[j * m + k];
      if (!(s > 0.0)) return false;
      double d = std::sqrt(s); L[j * m + j] = d;
      for (int i = j + 1; i < m; ++i) { double t = M[i * m + j]; for (int k = 0; k < j; ++k) t -= L[i * m + k] * L[j * m + k]; L[i * m + j] = t / d; }
    }
    return true;
  }
  // Indefinite fallback (Quu + 1e-4 I still not PD): the reference's Eigen ldlt() is a pivoted
  // factorisation that solves the symmetric indefinite system; restated as LU with partial pivoting.
  // The factor is flagged by L[0] = NaN and the LU is kept in lu_/piv_.
  std::vector<double> lu_; std::vector<int> piv_; bool use_lu_ = false;
  void ldl_fallback(const double* M, double* L, int m) {
    lu_.assign(M, M + m * m); piv_.assign(m, 0); use_lu_ = true;
    for (int c = 0; c < m; ++c) {
      int p = c; double best = std::fabs(lu_[c * m + c]);
      for (int r = c + 1; r < m; ++r) if (std::fabs(lu_[r * m + c]) > best) { best = std::fabs(lu_[r * m + c]); p = r; }
      piv_[c] = p;
      if (p != c) for (int k = 0; k < m; ++k) std::swap(lu_[c * m + k], lu_[p * m + k]);
      for (int r = c + 1; r < m; ++r) { double f = lu_[r * m + c] / lu_[c * m + c]; lu_[r * m + c] = f; for (int k = c + 1; k < m; ++k) lu_[r * m + k] -= f * lu_[c * m + k]; }
    }
  }
  void solve_factored(const double* L, const double* b, double* x, int m) const {
    std::vector<double> y(m);
    if (use_lu_) {
      for (int i = 0; i < m; ++i) y[i] = b[i];
      for (int c = 0; c < m; ++c) if (piv_[c] != c) std::swap(y[c], y[piv_[c]]);
      for (int c = 0; c < m; ++c) for (int r = c + 1; r < m; ++r) y[r] -= lu_[r * m + c] * y[c];
      for (int i = m - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < m; ++k) s -= lu_[i * m + k] * x[k]; x[i] = s / lu_[i * m + i]; }
      return;
    }
    for (int i = 0; i < m; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i * m + k] * y[k]; y[i] = s / L[i * m + i]; }
    for (int i = m - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < m; ++k) s -= L[k * m + i] * x[k]; x[i] = s / L[i * m + i]; }
  }

  double total_cost_of(const double* xs, const double* us) const { return total_cost(P, xs, us); }

  // ilqr.cpp:311-361
  bool line_search(const double* x0, double& new_cost, double& alpha_out) {
    static const double alphas[8] = {1.0, 0.8, 0.6, 0.4, 0.2, 0.1, 0.05, 0.01};
    const int n = H1_NX, m = H1_NU;
    const double baseline = total_cost_of(xbar.data(), ubar.data());
    std::vector<double> xn((N + 1) * n), un(N * m);
    for (int ai = 0; ai < 8; ++ai) {
      const double alpha = alphas[ai];
      std::memcpy(&xn[0], x0, n * sizeof(double));
      for (int t = 0; t < N; ++t) {
        for (int i = 0; i < m; ++i) {
          double s = 0; for (int j = 0; j < n; ++j) s += K[(t * m + i) * n + j] * (xn[t * n + j] - xbar[t * n + j]);
          un[t * m + i] = ubar[t * m + i] + alpha * kff[t * m + i] + s;
        }
        step(&xn[t * n], &un[t * m], &xn[(t + 1) * n], t);
      }
      const double c = total_cost_of(xn.data(), un.data());
      if (c < baseline - 1e-6) { xbar = xn; ubar = un; new_cost = c; alpha_out = alpha; return true; }
    }
    new_cost = baseline; alpha_out = 0.0;
    return false;
  }

  // ilqr.cpp:521-660
  bool solve(const double* x0, double& cost_out) {
    trace_cost.clear(); trace_alpha.clear(); trace_lambda.clear();
    double J = total_cost_of(xbar.data(), ubar.data());
    trace_cost.push_back(J);
    iters_done = 0;
    for (int iter = 0; iter < max_iter; ++iter) {
      const double Jprev = J;
      ++iters_done;
      std::memcpy(&xbar[0], x0, H1_NX * sizeof(double));
      rollout_nominal();
      linearize();
      cost_quadratics();
      backward_pass();
      double Jn, alpha; double lam_used = lambda;
      bool improved = line_search(x0, Jn, alpha);
      if (!improved) {
        lambda = std::min(lambda * 10.0, 1e-3);
        lam_used = lambda;
        backward_pass();
        improved = line_search(x0, Jn, alpha);
        if (!improved) {
          trace_cost.push_back(J); trace_alpha.push_back(0.0); trace_lambda.push_back(lam_used);
          if (iter > 1 && early_exit) break;
          continue;
        }
      }
      J = Jn;
      lambda = std::max(lambda / 2.0, 1e-6);
      trace_cost.push_back(J); trace_alpha.push_back(alpha); trace_lambda.push_back(lam_used);
      if (early_exit) {
        if (std::fabs(J - Jprev) < tol) break;
        if (J > 1e6) break;
      }
    }
    cost_out = J;
    return true;
  }

  // gravity compensation torques: qfrc_bias[6+i] at (x0.q, v=0)  (robot_utils.cpp:844-866, fixed index)
  void grav_comp(const double* x, double* ug) const {
    double qn = std::sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
    double qh[4] = {x[3] / qn, x[4] / qn, x[5] / qn, x[6] / qn};
    double v[H1_NV] = {0}, a[H1_NV] = {0}, tau[H1_NV];
    inverse_dynamics_mj<double>(qh, x + 7, v, a, H1_ARMATURE, P.dyn.g, tau);
    for (int i = 0; i < H1_NU; ++i) ug[i] = tau[6 + i];
  }

  // ilqr.cpp:50-117; u_init == nullptr -> gravity-compensation cold start evaluated at x0
  void initialize(const double* x0, const double* u_init, const double* prev_xbar, const double* prev_ubar) {
    const int n = H1_NX, m = H1_NU;
    std::memcpy(&xbar[0], x0, n * sizeof(double));
    if (prev_xbar && prev_ubar) {
      for (int t = 0; t < N - 1; ++t) std::memcpy(&ubar[t * m], &prev_ubar[(t + 1) * m], m * sizeof(double));
      std::memcpy(&ubar[(N - 1) * m], &prev_ubar[(N - 1) * m], m * sizeof(double));
      for (int t = 0; t < N - 1; ++t) std::memcpy(&xbar[(t + 1) * n], &prev_xbar[(t + 2) * n], n * sizeof(double));
      step(&xbar[(N - 1) * n], &ubar[(N - 1) * m], &xbar[N * n], N - 1);
    } else {
      if (u_init) std::memcpy(&ubar[0], u_init, N * m * sizeof(double));
      else { double ug[H1_NU]; grav_comp(x0, ug); for (int t = 0; t < N; ++t) std::memcpy(&ubar[t * m], ug, m * sizeof(double)); }
      rollout_nominal();
    }
  }
};

}  // namespace orc

using orc::Solver;

extern "C" {

void* orc_create(int N, double h) { return new Solver(N, h); }
void orc_destroy(void* s) { delete (Solver*)s; }

void orc_set_cost_weights(void* s, const double* Q, const double* R, const double* Qf) {
  Solver* S = (Solver*)s; std::memcpy(S->P.Q, Q, sizeof(S->P.Q)); std::memcpy(S->P.R, R, sizeof(S->P.R)); std::memcpy(S->P.Qf, Qf, sizeof(S->P.Qf));
}
void orc_set_task_weights(void* s, double w_com, double w_com_vel, double w_ee_pos, double w_ee_vel, double w_upright, double w_balance) {
  Solver* S = (Solver*)s; S->P.w_com = w_com; S->P.w_com_vel = w_com_vel; S->P.w_ee_pos = w_ee_pos; S->P.w_ee_vel = w_ee_vel; S->P.w_upright = w_upright; S->P.w_balance = w_balance;
}
void orc_set_constraint_weights(void* s, double wj, double wc) { Solver* S = (Solver*)s; S->P.w_joint = wj; S->P.w_ctrl = wc; }
void orc_set_gravity(void* s, double gx, double gy, double gz) { Solver* S = (Solver*)s; S->P.dyn.g[0] = gx; S->P.dyn.g[1] = gy; S->P.dyn.g[2] = gz; }
void orc_set_references(void* s, const double* x_ref, const double* u_ref, const double* com_ref) {
  Solver* S = (Solver*)s; const int N = S->N;
  S->P.x_ref.assign(x_ref, x_ref + (N + 1) * H1_NX); S->P.u_ref.assign(u_ref, u_ref + N * H1_NU); S->P.com_ref.assign(com_ref, com_ref + (N + 1) * 3);
}
void orc_set_contact_schedule(void* s, const int* stance) { Solver* S = (Solver*)s; S->P.stance.assign(stance, stance + (S->N + 1) * 2); }
void orc_set_ee_references(void* s, const double* ee_ref, const double* com_vel_ref) {
  Solver* S = (Solver*)s; const int N = S->N;
  S->P.ee_ref.assign(ee_ref, ee_ref + (N + 1) * 6);
  if (com_vel_ref) S->P.com_vel_ref.assign(com_vel_ref, com_vel_ref + (N + 1) * 3);
}
void orc_set_options(void* s, double lambda, int max_iter, double tol, int jac_mode, double fd_eps, int quad_mode, int early_exit) {
  Solver* S = (Solver*)s; S->lambda = lambda; S->max_iter = max_iter; S->tol = tol; S->jac_mode = jac_mode; S->fd_eps = fd_eps; S->quad_mode = quad_mode; S->early_exit = early_exit;
}
double orc_get_lambda(void* s) { return ((Solver*)s)->lambda; }

void orc_initialize(void* s, const double* x0, const double* u_init, const double* prev_xbar, const double* prev_ubar) { ((Solver*)s)->initialize(x0, u_init, prev_xbar, prev_ubar); }
void orc_set_trajectory(void* s, const double* xbar, const double* ubar) { Solver* S = (Solver*)s; S->xbar.assign(xbar, xbar + (S->N + 1) * H1_NX); S->ubar.assign(ubar, ubar + S->N * H1_NU); }
int orc_solve(void* s, const double* x0, double* cost_out) { double c = 0; bool ok = ((Solver*)s)->solve(x0, c); *cost_out = c; return ok ? 0 : 1; }

// stage entry points (parity per kernel)
void orc_step(void* s, const double* x, const double* u, double* xn) { Solver* S = (Solver*)s; orc::h1_step<double>(x, u, S->P.dyn, xn, nullptr); }
// one step with explicit stance flags (left, right) -- contact mode only uses them when set_contact_mode(1)
void orc_step_stance(void* s, const double* x, const double* u, const int* stance, double* xn) { Solver* S = (Solver*)s; orc::h1_step<double>(x, u, S->P.dyn, xn, stance); }
void orc_set_contact_mode(void* s, int mode, double soft) { Solver* S = (Solver*)s; S->P.dyn.contact = mode; if (soft > 0.0) S->P.dyn.soft = soft; }
void orc_set_friction(void* s, double mu) { ((Solver*)s)->P.dyn.mu = mu; }   // sliding friction coefficient of contact mode 3
void orc_joint_ranges(double* out /*[19][2]*/) { for (int i = 0; i < H1_NJ; ++i) { out[2 * i] = H1_JRANGE[i][0]; out[2 * i + 1] = H1_JRANGE[i][1]; } }
void orc_set_joint_limits(void* s, int on) { ((Solver*)s)->P.dyn.limits = on ? 1 : 0; }   // joint-limit rows of the plant (h1_step)
void orc_set_joint_limit_stiffness(void* s, double k) { ((Solver*)s)->P.dyn.lim_k = k; }   // restoring stiffness of those rows (0: pure stop)
void orc_rollout(void* s) { ((Solver*)s)->rollout_nominal(); }
void orc_linearize(void* s) { ((Solver*)s)->linearize(); }
void orc_cost_quadratics(void* s) { ((Solver*)s)->cost_quadratics(); }
void orc_backward_pass(void* s) { ((Solver*)s)->backward_pass(); }
int orc_line_search(void* s, const double* x0, double* new_cost, double* alpha) { return ((Solver*)s)->line_search(x0, *new_cost, *alpha) ? 1 : 0; }
double orc_total_cost(void* s) { Solver* S = (Solver*)s; return S->total_cost_of(S->xbar.data(), S->ubar.data()); }
void orc_grav_comp(void* s, const double* x, double* ug) { ((Solver*)s)->grav_comp(x, ug); }
void orc_set_linearization(void* s, const double* A, const double* B) { Solver* S = (Solver*)s; S->A.assign(A, A + S->N * H1_NX * H1_NX); S->B.assign(B, B + S->N * H1_NX * H1_NU); }
void orc_set_quadratics(void* s, const double* lx, const double* lu, const double* lxx, const double* luu_diag) {
  Solver* S = (Solver*)s; const int N = S->N;
  S->lx.assign(lx, lx + (N + 1) * H1_NX); S->lu.assign(lu, lu + N * H1_NU); S->lxx.assign(lxx, lxx + (N + 1) * H1_NX * H1_NX); S->luu.assign(luu_diag, luu_diag + N * H1_NU);
}

#define ORC_GETTER(name, member)                                                     \
  void orc_get_##name(void* s, double* out) { Solver* S = (Solver*)s; std::memcpy(out, S->member.data(), S->member.size() * sizeof(double)); }
ORC_GETTER(xbar, xbar) ORC_GETTER(ubar, ubar) ORC_GETTER(K, K) ORC_GETTER(kff, kff) ORC_GETTER(A, A) ORC_GETTER(B, B)
ORC_GETTER(lx, lx) ORC_GETTER(lu, lu) ORC_GETTER(lxx, lxx) ORC_GETTER(luu, luu) ORC_GETTER(Vx, Vx_last) ORC_GETTER(Vxx, Vxx_last)

int orc_get_trace(void* s, double* cost /*[max_iter+1]*/, double* alpha /*[max_iter]*/, double* lambda /*[max_iter]*/) {
  Solver* S = (Solver*)s;
  std::memcpy(cost, S->trace_cost.data(), S->trace_cost.size() * sizeof(double));
  std::memcpy(alpha, S->trace_alpha.data(), S->trace_alpha.size() * sizeof(double));
  std::memcpy(lambda, S->trace_lambda.data(), S->trace_lambda.size() * sizeof(double));
  return S->iters_done;
}
// u = ubar[0] + K[0] (x_meas - xbar[0])   (mpc.cpp:97-101)
void orc_compute_control(void* s, const double* x_meas, double* u) {
  Solver* S = (Solver*)s;
  for (int i = 0; i < H1_NU; ++i) { double a = S->ubar[i]; for (int j = 0; j < H1_NX; ++j) a += S->K[i * H1_NX + j] * (x_meas[j] - S->xbar[j]); u[i] = a; }
}

// kinematic helpers used to build references the way loadReferences does (robot_utils.cpp:369-403):
// com = MuJoCo subtree_com of the root (MJCF masses), ee = xpos of the two ankle bodies
void orc_reference_kinematics(const double* x, double* com, double* ee /*[2][3]*/) {
  orc::com_mj<double>(x, com);
  double Rw[H1_NB][9], pw[H1_NB][3]; orc::fk_mj<double>(x, Rw, pw);
  for (int k = 0; k < 3; ++k) { ee[k] = pw[H1_EE_LEFT][k]; ee[3 + k] = pw[H1_EE_RIGHT][k]; }
}
void orc_forward_dynamics(const double* x, const double* tau, double arm_eff, const double* grav, double* qacc) {
  double qn = std::sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  double qh[4] = {x[3] / qn, x[4] / qn, x[5] / qn, x[6] / qn};
  orc::forward_dynamics_mj<double>(qh, x + 7, x + H1_NQ, tau, arm_eff, grav, qacc);
}
void orc_inverse_dynamics(const double* x, const double* qacc, double arm, const double* grav, double* tau) {
  double qn = std::sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  double qh[4] = {x[3] / qn, x[4] / qn, x[5] / qn, x[6] / qn};
  orc::inverse_dynamics_mj<double>(qh, x + 7, x + H1_NQ, qacc, arm, grav, tau);
}
// cost quadratics of a single knot in either mode (closed form vs AD cross-check)
void orc_knot_quadratics(void* s, int t, const double* x, const double* u, int mode, double* lx, double* lu, double* lxx, double* luu) {
  Solver* S = (Solver*)s;
  orc::cost_quadratics_knot(S->P, t, x, u, lx, lu, lxx, luu, (orc::QuadMode)mode);
}

// Batched CPU baseline: B independent rollouts, OpenMP over rollouts (one rollout per thread at a time).
// tmpl supplies problem data and options; x0 [B][51], u_init [B][N][19] (nullable -> gravity comp).
// Outputs: cost [B], iters [B], optional K0 [B][19][51], u0 [B][19]. Returns total iLQR iterations executed.
long orc_batch_solve(void* tmpl, int Bn, const double* x0, const double* u_init, double* cost, int* iters, double* u0, double* K0, int nthreads) {
  Solver* T = (Solver*)tmpl;
  long total = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
  for (int b = 0; b < Bn; ++b) {
    Solver S = *T;
    const int N = S.N;
    S.initialize(x0 + (size_t)b * H1_NX, u_init ? u_init + (size_t)b * N * H1_NU : nullptr, nullptr, nullptr);
    double c = 0; S.solve(x0 + (size_t)b * H1_NX, c);
    cost[b] = c; iters[b] = S.iters_done; total += S.iters_done;
    if (u0) std::memcpy(u0 + (size_t)b * H1_NU, S.ubar.data(), H1_NU * sizeof(double));
    if (K0) std::memcpy(K0 + (size_t)b * H1_NU * H1_NX, S.K.data(), H1_NU * H1_NX * sizeof(double));
  }
  return total;
}
// algorithmic flop counts of SURVEY.md 8(d) on the oracle's own code (opcount.cpp)
void orc_op_counts_impl(const orc::Problem& P, int t, const double* x, const double* u, double* out);
void orc_op_counts(void* s, int t, const double* x, const double* u, double* out /*[8]*/) { orc_op_counts_impl(((Solver*)s)->P, t, x, u, out); }
int orc_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
