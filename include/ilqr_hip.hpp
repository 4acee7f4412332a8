// C++ mirror of the reference's solver classes over the C ABI (include/ilqr_hip.h), for a maintainer who
// wants to swap the reference's CPU iLQR for the HIP one without touching call sites.
//
//   ilqr_hip::iLQR  mirrors  class iLQR   (reference include/ilqr/ilqr.hpp:17-45)
//   ilqr_hip::MPC   mirrors  class MPC    (reference include/ilqr/mpc.hpp:18-47, src/ilqr/mpc.cpp:40-127)
//
// Eigen is not a dependency of this build; vectors/matrices are std::vector<double> (row-major) and
// std::array<double,3>.  With Eigen available, Eigen::Map<const VectorXd>(v.data(), v.size()) adapts both ways.
// batch = 1 reproduces the reference call for call; batch > 1 solves independent rollouts (x0 = batch * 51 doubles)
// against one shared reference window.
#pragma once
#include <array>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "ilqr_hip.h"

namespace ilqr_hip {

using Vec = std::vector<double>;

// Config::CostWeights + Config::buildCostMatrices (reference include/common/config.hpp, src/common/config.cpp:66-122):
// the diagonals of Q, R, Qf from the scalar weights of config.yaml (defaults = the shipped config.yaml:24-46).
struct CostConfig {
  double Q_position_x = 200.0, Q_position_y = 50.0, Q_position_z = 200.0, Q_quat_w = 50.0;
  std::array<double, 3> Q_quat_xyz{{50.0, 50.0, 50.0}};
  double Q_joint_pos = 50.0, Q_vel_x = 150.0, Q_vel_y = 50.0, Q_vel_z = 150.0, Q_ang_vel = 75.0, Q_joint_vel = 75.0;
  double R_control = 0.001, Qf_multiplier = 2.0, Qf_position_x = 5.0, Qf_position_y = 2.0, Qf_position_z = 5.0, Qf_vel_z = 4.0;
};
inline void buildCostMatrices(const CostConfig& c, Vec& Qdiag, Vec& Rdiag, Vec& Qfdiag) {
  const int nx = ILQR_NX, nu = ILQR_NU, nq = ILQR_NQ;
  Qdiag.assign(nx, 1.0); Rdiag.assign(nu, 1.0);
  Qdiag[0] = c.Q_position_x; Qdiag[1] = c.Q_position_y; Qdiag[2] = c.Q_position_z;
  Qdiag[3] = c.Q_quat_w; Qdiag[4] = c.Q_quat_xyz[0]; Qdiag[5] = c.Q_quat_xyz[1]; Qdiag[6] = c.Q_quat_xyz[2];
  for (int i = 7; i < nq; ++i) Qdiag[i] = c.Q_joint_pos;
  Qdiag[nq + 0] = c.Q_vel_x; Qdiag[nq + 1] = c.Q_vel_y; Qdiag[nq + 2] = c.Q_vel_z;
  for (int i = nq + 3; i < nq + 6; ++i) Qdiag[i] = c.Q_ang_vel;
  for (int i = nq + 6; i < nx; ++i) Qdiag[i] = c.Q_joint_vel;
  for (double& r : Rdiag) r *= c.R_control;
  Qfdiag = Qdiag;
  for (double& q : Qfdiag) q *= c.Qf_multiplier;
  Qfdiag[0] *= c.Qf_position_x; Qfdiag[1] *= c.Qf_position_y; Qfdiag[2] *= c.Qf_position_z; Qfdiag[nq + 2] *= c.Qf_vel_z;
}

// The reference's profiler (include/common/profiler.hpp: prof_data[name].times, main/humanoid_mpc.cpp:195-226): named lists of
// milliseconds and the table printer, same keys (MPC_stepOnce / MPC_extractReference / MPC_warmStart / MPC_iLQR_solve /
// MPC_computeControl from the host clock, iLQR_* from the device events of the solve).
struct Profiler {
  std::map<std::string, std::vector<double>> times;
  void add(const std::string& key, double ms) { times[key].push_back(ms); }
  void print(std::ostream& os) const {
    os << "\n=== Performance Profiling ===" << std::endl;
    os << std::fixed << std::setprecision(2);
    os << "\n--- Timing Summary ---" << std::endl;
    os << std::left << std::setw(20) << "Function" << std::right << std::setw(8) << "Calls" << std::setw(12) << "Total(ms)" << std::setw(12) << "Avg(ms)"
       << std::setw(12) << "Min(ms)" << std::setw(12) << "Max(ms)" << std::endl;
    os << std::string(76, '-') << std::endl;
    for (const auto& e : times) {
      const auto& t = e.second;
      if (t.empty()) continue;
      double total = 0.0, mn = t[0], mx = t[0];
      for (double v : t) { total += v; if (v < mn) mn = v; if (v > mx) mx = v; }
      os << std::left << std::setw(20) << e.first << std::right << std::setw(8) << t.size() << std::setw(12) << total << std::setw(12) << total / t.size()
         << std::setw(12) << mn << std::setw(12) << mx << std::endl;
    }
  }
};

class iLQR {
 public:
  // iLQR(RobotUtils&, int N, double dt, urdf_path): the H1 model is compiled into the library
  iLQR(int N, double dt, int batch = 1, int device = 0) : N_(N), B_(batch) {
    const int rc = ilqr_hip_create(&ctx_, device, batch, N, dt);
    if (rc != ILQR_OK) { std::string m = ctx_ ? ilqr_hip_last_error(ctx_) : "no device"; if (ctx_) ilqr_hip_destroy(ctx_); throw std::runtime_error("ilqr_hip_create: " + m); }
  }
  ~iLQR() { if (ctx_) ilqr_hip_destroy(ctx_); }
  iLQR(const iLQR&) = delete;
  iLQR& operator=(const iLQR&) = delete;
  ilqr_hip_ctx* handle() { return ctx_; }

  // RobotUtils setters the reference app calls in setupSimulation (main/humanoid_mpc.cpp:94-118)
  void setCostWeights(const Vec& Qdiag, const Vec& Rdiag, const Vec& Qfdiag) { chk(ilqr_hip_set_cost_weights(ctx_, Qdiag.data(), Rdiag.data(), Qfdiag.data())); }
  void setTaskWeights(double com, double com_vel, double ee_pos, double ee_vel, double upright, double balance) { chk(ilqr_hip_set_task_weights(ctx_, com, com_vel, ee_pos, ee_vel, upright, balance)); }
  void setConstraintWeights(double joint, double ctrl) { chk(ilqr_hip_set_constraint_weights(ctx_, joint, ctrl)); }
  void setGravity(double gx, double gy, double gz) { chk(ilqr_hip_set_gravity(ctx_, gx, gy, gz)); }
  void setContactSchedule(const std::vector<int>& stance /*[N+1][2]*/) { chk(ilqr_hip_set_contact_schedule(ctx_, stance.data(), 1)); }
  // contact row: ILQR_CONTACT_NONE (default) or ILQR_CONTACT_RIGID_STANCE on the scheduled feet
  void setContactMode(int mode, double softness = 0.0) { chk(ilqr_hip_set_contact_mode(ctx_, mode, softness)); }
  // sliding friction coefficient of contact modes 3 / 4 (ilqr_hip.h: Coulomb limit on the stance feet; forward-difference Jacobians only)
  void setFriction(double mu) { chk(ilqr_hip_set_friction(ctx_, mu)); }
  void setJointLimits(bool on) { chk(ilqr_hip_set_joint_limits(ctx_, on ? 1 : 0)); }     // joint-limit rows of the plant (ilqr_hip.h)
  void setJointLimitStiffness(double k) { chk(ilqr_hip_set_joint_limit_stiffness(ctx_, k)); }      // their restoring term (0: pure stop)
  void setEEReferences(const Vec& ee_ref /*[N+1][2][3]*/, const Vec* com_vel_ref = nullptr) { chk(ilqr_hip_set_ee_references(ctx_, ee_ref.data(), com_vel_ref ? com_vel_ref->data() : nullptr, 1)); }

  // include/ilqr/ilqr.hpp:22-24
  void setRegularization(double lambda) { chk(ilqr_hip_set_regularization(ctx_, lambda)); }
  void setMaxIterations(int n) { chk(ilqr_hip_set_max_iterations(ctx_, n)); }
  void setTolerance(double tol) { chk(ilqr_hip_set_tolerance(ctx_, tol)); }
  // build-specific: analytic Jacobians (default) or the reference's forward differences (robot_utils.cpp:120-160);
  // early_exit = false runs exactly max_iterations iterations
  void setOptions(int jacobian_mode, double fd_eps = 1e-5, bool early_exit = true) { chk(ilqr_hip_set_options(ctx_, jacobian_mode, fd_eps, early_exit ? 1 : 0)); }
  // off: solveAsync enqueues every iteration at once and never blocks the host (see ilqr_hip_solve_async in ilqr_hip.h)
  void setEarlyExitGate(bool on) { chk(ilqr_hip_set_early_exit_gate(ctx_, on ? 1 : 0)); }
  void enableProfiling(bool on) { chk(ilqr_hip_enable_profiling(ctx_, on ? 1 : 0)); }
  // device time per stage of the last solve, keyed like the reference's profiler (ilqr.cpp:537-639)
  std::map<std::string, double> stageMs() {
    double ms[8], n[8]; chk(ilqr_hip_get_stage_ms(ctx_, ms, n));
    return {{"iLQR_forwardRollout", ms[0]}, {"iLQR_linearization", ms[1]}, {"iLQR_costQuadratics", ms[2]}, {"iLQR_backwardPass", ms[3] + ms[6]}, {"iLQR_lineSearch", ms[4] + ms[7]}};
  }

  // include/ilqr/ilqr.hpp:40-45 (batch 1: x0[51]; prev_* nullable)
  void initializeWithReference(const Vec& x0, const std::vector<Vec>& x_ref, const std::vector<Vec>& u_ref,
                               const std::vector<std::array<double, 3>>& com_ref, const std::vector<Vec>* prev_xbar = nullptr,
                               const std::vector<Vec>* prev_ubar = nullptr) {
    if (!setWindow(x_ref, u_ref, com_ref)) throw std::runtime_error("reference size mismatch");
    checkState(x0);
    if (prev_xbar && prev_ubar && prev_xbar->size() == (size_t)N_ + 1 && prev_ubar->size() == (size_t)N_) {
      Vec px = flatten(*prev_xbar), pu = flatten(*prev_ubar);
      chk(ilqr_hip_initialize(ctx_, x0.data(), nullptr, px.data(), pu.data()));
    } else {
      chk(ilqr_hip_initialize(ctx_, x0.data(), nullptr, nullptr, nullptr));   // gravity-compensation cold start
    }
  }
  // include/ilqr/ilqr.hpp:27-31: false on a reference-size mismatch (ilqr.cpp:526-532), true otherwise
  bool solve(const Vec& x0, const std::vector<Vec>& x_ref, const std::vector<Vec>& u_ref,
             const std::vector<std::array<double, 3>>& com_ref, double& cost_out) {
    if (!setWindow(x_ref, u_ref, com_ref)) return false;
    checkState(x0);
    Vec cost(B_);
    chk(ilqr_hip_solve(ctx_, x0.data(), cost.data()));
    cost_out = cost[0];
    return true;
  }
  // include/ilqr/ilqr.hpp:34-37 (rollout 0 of the batch; use the C ABI for all rollouts)
  std::vector<Vec> xbar() { Vec f((size_t)B_ * (N_ + 1) * ILQR_NX); chk(ilqr_hip_get_xbar(ctx_, f.data())); return split(f, N_ + 1, ILQR_NX); }
  std::vector<Vec> ubar() { Vec f((size_t)B_ * N_ * ILQR_NU); chk(ilqr_hip_get_ubar(ctx_, f.data())); return split(f, N_, ILQR_NU); }
  std::vector<Vec> gainsK() { Vec f((size_t)B_ * N_ * ILQR_NU * ILQR_NX); chk(ilqr_hip_get_gains_K(ctx_, f.data())); return split(f, N_, ILQR_NU * ILQR_NX); }
  std::vector<Vec> gainsKff() { Vec f((size_t)B_ * N_ * ILQR_NU); chk(ilqr_hip_get_gains_kff(ctx_, f.data())); return split(f, N_, ILQR_NU); }
  int horizon() const { return N_; }
  int batch() const { return B_; }

 private:
  void chk(int rc) { if (rc != ILQR_OK) throw std::runtime_error(std::string("ilqr_hip: ") + ilqr_hip_last_error(ctx_)); }
  // x0 carries one state per rollout of the handle (the C ABI copies batch * 51 doubles)
  void checkState(const Vec& x0) const { if (x0.size() != (size_t)B_ * ILQR_NX) throw std::runtime_error("x0 must hold batch * 51 doubles"); }
  static Vec flatten(const std::vector<Vec>& v) { Vec f; for (const Vec& r : v) f.insert(f.end(), r.begin(), r.end()); return f; }
  static std::vector<Vec> split(const Vec& f, int rows, int width) { std::vector<Vec> o(rows); for (int t = 0; t < rows; ++t) o[t].assign(f.begin() + (size_t)t * width, f.begin() + (size_t)(t + 1) * width); return o; }
  bool setWindow(const std::vector<Vec>& x_ref, const std::vector<Vec>& u_ref, const std::vector<std::array<double, 3>>& com_ref) {
    if (x_ref.size() != (size_t)N_ + 1 || u_ref.size() != (size_t)N_ || com_ref.size() != (size_t)N_ + 1) return false;
    Vec xr = flatten(x_ref), ur = flatten(u_ref), cr;
    for (const auto& c : com_ref) cr.insert(cr.end(), c.begin(), c.end());
    chk(ilqr_hip_set_references(ctx_, xr.data(), ur.data(), cr.data(), 1));
    return true;
  }
  ilqr_hip_ctx* ctx_ = nullptr;
  int N_, B_;
};

// MPC (include/ilqr/mpc.hpp:18-47, src/ilqr/mpc.cpp) for batch 1.  `window(t_idx, x_ref, u_ref, com_ref)` plays the role of
// RobotUtils::getReferenceWindow (src/common/robot_utils.cpp:422-443).  Logging and profiling follow the reference's file
// formats and keys (mpc.cpp:181-355, main/humanoid_mpc.cpp:195-226) so its plotting / analysis scripts keep working.
template <class WindowFn>
class MPC {
 public:
  MPC(int N, double dt, WindowFn window, int device = 0) : ilqr_(N, dt, 1, device), window_(window), N_(N), dt_(dt) {}
  ~MPC() { finalizeCSVLog(false); finalizeOptimalTrajectoryLog(false); }
  iLQR& solver() { return ilqr_; }
  Profiler& profiler() { return prof_; }
  void enableProfiling(bool on) { profiling_ = on; ilqr_.enableProfiling(on); }

  // mpc.cpp:40-127
  bool stepOnce(const Vec& x_measured, Vec& u_apply) {
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_start = clk::now();
    try {
      auto t0 = clk::now();
      window_(t_idx_, x_ref_window_, u_ref_window_, com_ref_window_);              // extractReferenceWindow
      auto t1 = clk::now();
      if (profiling_) prof_.add("MPC_extractReference", ms(t0, t1));
      if (has_prev_) ilqr_.initializeWithReference(x_measured, x_ref_window_, u_ref_window_, com_ref_window_, &prev_xbar_, &prev_ubar_);
      else ilqr_.initializeWithReference(x_measured, x_ref_window_, u_ref_window_, com_ref_window_);
      auto t2 = clk::now();
      if (profiling_) prof_.add("MPC_warmStart", ms(t1, t2));
      double cost = 0.0;
      const bool ok = ilqr_.solve(x_measured, x_ref_window_, u_ref_window_, com_ref_window_, cost);
      auto t3 = clk::now();
      if (profiling_) { prof_.add("MPC_iLQR_solve", ms(t2, t3)); for (const auto& kv : ilqr_.stageMs()) prof_.add(kv.first, kv.second); }
      if (!ok) {                                                                    // mpc.cpp:82-91
        std::cerr << "iLQR solve failed at time index " << t_idx_ << std::endl;
        u_apply = has_prev_ ? prev_ubar_[0] : Vec(ILQR_NU, 0.0);
        return false;
      }
      u_apply.assign(ILQR_NU, 0.0);
      if (ilqr_hip_compute_control(ilqr_.handle(), x_measured.data(), u_apply.data()) != ILQR_OK) throw std::runtime_error("compute_control");
      auto t4 = clk::now();
      if (profiling_) prof_.add("MPC_computeControl", ms(t3, t4));
      prev_xbar_ = ilqr_.xbar(); prev_ubar_ = ilqr_.ubar(); prev_K_ = ilqr_.gainsK();
      has_prev_ = true; last_solve_cost_ = cost;
      last_solve_time_ms_ = std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t_start).count() / 1000.0;
      ++t_idx_;
      logCurrentStep(x_measured, u_apply);
      logAppliedOptimal(x_measured, u_apply);
      if (profiling_) prof_.add("MPC_stepOnce", ms(t_start, clk::now()));          // main/humanoid_mpc.cpp:143-147
      return true;
    } catch (const std::exception& e) {                                             // mpc.cpp:122-126
      std::cerr << "Exception in MPC step: " << e.what() << std::endl;
      u_apply.assign(ILQR_NU, 0.0);
      return false;
    }
  }
  void reset() { t_idx_ = 0; has_prev_ = false; last_solve_cost_ = 0.0; last_solve_time_ms_ = 0.0; prev_xbar_.clear(); prev_ubar_.clear(); prev_K_.clear(); }
  void setTimeIndex(int t_idx) { t_idx_ = t_idx; }
  int getTimeIndex() const { return t_idx_; }
  const std::vector<Vec>& gainsK() const { return prev_K_; }                        // K_t as row-major 19 x 51
  double getLastSolveCost() const { return last_solve_cost_; }
  void getNominalTrajectory(std::vector<Vec>& x_traj, std::vector<Vec>& u_traj) const {   // mpc.cpp:150-161
    if (has_prev_) { x_traj = prev_xbar_; u_traj = prev_ubar_; } else { x_traj.clear(); u_traj.clear(); }
  }
  // MPC::computeTVLQRControl, mpc.cpp:168-179
  Vec computeTVLQRControl(const Vec& x_measured) const {
    Vec u(ILQR_NU, 0.0);
    if (!has_prev_) return u;
    for (int i = 0; i < ILQR_NU; ++i) {
      double s = prev_ubar_[0][i];
      for (int j = 0; j < ILQR_NX; ++j) s += prev_K_[0][(size_t)i * ILQR_NX + j] * (x_measured[j] - prev_xbar_[0][j]);
      u[i] = s;
    }
    return u;
  }

  // ---- CSV logging, mpc.cpp:181-262: header time_index,time_sec,solve_cost,solve_time_ms,x_*,u_*,x_ref_*,u_ref_*
  void enableCSVLogging(const std::string& filename) {
    csv_filename_ = filename;
    csv_file_.open(csv_filename_, std::ios::out | std::ios::trunc);
    if (!csv_file_.is_open()) { std::cerr << "Failed to open CSV file: " << csv_filename_ << std::endl; return; }
    csv_file_ << "time_index,time_sec,solve_cost,solve_time_ms";
    for (int i = 0; i < ILQR_NX; ++i) csv_file_ << ",x_" << i;
    for (int i = 0; i < ILQR_NU; ++i) csv_file_ << ",u_" << i;
    for (int i = 0; i < ILQR_NX; ++i) csv_file_ << ",x_ref_" << i;
    for (int i = 0; i < ILQR_NU; ++i) csv_file_ << ",u_ref_" << i;
    csv_file_ << std::endl;
  }
  void logCurrentStep(const Vec& x_measured, const Vec& u_applied) {
    if (!csv_file_.is_open()) return;
    csv_file_ << t_idx_ << "," << (t_idx_ * dt_) << "," << last_solve_cost_ << "," << last_solve_time_ms_;
    for (double v : x_measured) csv_file_ << "," << v;
    for (double v : u_applied) csv_file_ << "," << v;
    if (!x_ref_window_.empty()) { for (double v : x_ref_window_[0]) csv_file_ << "," << v; } else { for (int i = 0; i < ILQR_NX; ++i) csv_file_ << ",0.0"; }
    if (!u_ref_window_.empty()) { for (double v : u_ref_window_[0]) csv_file_ << "," << v; } else { for (int i = 0; i < ILQR_NU; ++i) csv_file_ << ",0.0"; }
    csv_file_ << std::endl;
  }
  void finalizeCSVLog(bool announce = true) {
    if (csv_file_.is_open()) { csv_file_.flush(); csv_file_.close(); if (announce) std::cout << "CSV log finalized: " << csv_filename_ << std::endl; }
  }
  // ---- optimal-trajectory logging, mpc.cpp:264-355: <base>/q_optimal.csv (step,time_sec,q_0..q_25), <base>/u_optimal.csv
  void enableOptimalTrajectoryLogging(const std::string& base_path) {
    trajectory_base_path_ = base_path;
    q_optimal_file_.open(base_path + "/q_optimal.csv", std::ios::out | std::ios::trunc);
    u_optimal_file_.open(base_path + "/u_optimal.csv", std::ios::out | std::ios::trunc);
    if (!q_optimal_file_.is_open() || !u_optimal_file_.is_open()) { std::cerr << "Failed to open optimal trajectory files in: " << base_path << std::endl; return; }
    q_optimal_file_ << "step,time_sec";
    for (int i = 0; i < ILQR_NQ; ++i) q_optimal_file_ << ",q_" << i;
    q_optimal_file_ << std::endl;
    u_optimal_file_ << "step,time_sec";
    for (int i = 0; i < ILQR_NU; ++i) u_optimal_file_ << ",u_" << i;
    u_optimal_file_ << std::endl;
  }
  void logAppliedOptimal(const Vec& x_applied, const Vec& u_applied) {
    if (!q_optimal_file_.is_open() || !u_optimal_file_.is_open()) return;
    q_optimal_file_ << t_idx_ << "," << (t_idx_ * dt_);
    const Vec& q0 = prev_xbar_.empty() ? x_applied : prev_xbar_[0];     // first knot of the optimised trajectory
    for (int i = 0; i < ILQR_NQ; ++i) q_optimal_file_ << "," << q0[i];
    q_optimal_file_ << std::endl;
    u_optimal_file_ << t_idx_ << "," << (t_idx_ * dt_);
    const Vec& u0 = prev_ubar_.empty() ? u_applied : prev_ubar_[0];
    for (double v : u0) u_optimal_file_ << "," << v;
    u_optimal_file_ << std::endl;
  }
  void finalizeOptimalTrajectoryLog(bool announce = true) {
    const bool was = q_optimal_file_.is_open() || u_optimal_file_.is_open();
    if (q_optimal_file_.is_open()) { q_optimal_file_.flush(); q_optimal_file_.close(); }
    if (u_optimal_file_.is_open()) { u_optimal_file_.flush(); u_optimal_file_.close(); }
    if (was && announce) std::cout << "Optimal trajectory logs finalized: " << trajectory_base_path_ << "/q_optimal.csv and u_optimal.csv" << std::endl;
  }

 private:
  iLQR ilqr_;
  WindowFn window_;
  int N_;
  double dt_;
  int t_idx_ = 0;
  bool has_prev_ = false, profiling_ = false;
  double last_solve_cost_ = 0.0, last_solve_time_ms_ = 0.0;
  std::vector<Vec> prev_xbar_, prev_ubar_, prev_K_;
  std::vector<Vec> x_ref_window_, u_ref_window_;
  std::vector<std::array<double, 3>> com_ref_window_;
  Profiler prof_;
  std::string csv_filename_, trajectory_base_path_;
  std::ofstream csv_file_, q_optimal_file_, u_optimal_file_;
};

}  // namespace ilqr_hip
