// C++ mirror of the reference's solver classes over the C ABI (include/ilqr_hip.h), for a maintainer who
// wants to swap the reference's CPU iLQR for the HIP one without touching call sites.
//
//   ilqr_hip::iLQR  mirrors  class iLQR   (reference include/ilqr/ilqr.hpp:17-45)
//   ilqr_hip::MPC   mirrors  class MPC    (reference include/ilqr/mpc.hpp:18-47, src/ilqr/mpc.cpp:40-127)
//
// Eigen is not a dependency of this build; vectors/matrices are std::vector<double> (row-major) and
// std::array<double,3>.  With Eigen available, Eigen::Map<const VectorXd>(v.data(), v.size()) adapts both ways.
// batch = 1 reproduces the reference call for call; batch > 1 solves independent rollouts (x0, u_init per
// rollout) against shared or per-rollout reference windows.
#pragma once
#include <array>
#include <stdexcept>
#include <string>
#include <vector>

#include "ilqr_hip.h"

namespace ilqr_hip {

using Vec = std::vector<double>;

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
  void setEEReferences(const Vec& ee_ref /*[N+1][2][3]*/, const Vec* com_vel_ref = nullptr) { chk(ilqr_hip_set_ee_references(ctx_, ee_ref.data(), com_vel_ref ? com_vel_ref->data() : nullptr, 1)); }

  // include/ilqr/ilqr.hpp:22-24
  void setRegularization(double lambda) { chk(ilqr_hip_set_regularization(ctx_, lambda)); }
  void setMaxIterations(int n) { chk(ilqr_hip_set_max_iterations(ctx_, n)); }
  void setTolerance(double tol) { chk(ilqr_hip_set_tolerance(ctx_, tol)); }

  // include/ilqr/ilqr.hpp:40-45 (batch 1: x0[51]; prev_* nullable)
  void initializeWithReference(const Vec& x0, const std::vector<Vec>& x_ref, const std::vector<Vec>& u_ref,
                               const std::vector<std::array<double, 3>>& com_ref, const std::vector<Vec>* prev_xbar = nullptr,
                               const std::vector<Vec>* prev_ubar = nullptr) {
    if (!setWindow(x_ref, u_ref, com_ref)) throw std::runtime_error("reference size mismatch");
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

// MPC::stepOnce (src/ilqr/mpc.cpp:40-127) for batch 1.  `window(t_idx, x_ref, u_ref, com_ref)` plays the role of
// RobotUtils::getReferenceWindow (src/common/robot_utils.cpp:422-443).
template <class WindowFn>
class MPC {
 public:
  MPC(int N, double dt, WindowFn window, int device = 0) : ilqr_(N, dt, 1, device), window_(window) {}
  iLQR& solver() { return ilqr_; }
  bool stepOnce(const Vec& x_measured, Vec& u_apply) {
    std::vector<Vec> x_ref, u_ref; std::vector<std::array<double, 3>> com_ref;
    window_(t_idx_, x_ref, u_ref, com_ref);
    try {
      if (has_prev_) ilqr_.initializeWithReference(x_measured, x_ref, u_ref, com_ref, &prev_xbar_, &prev_ubar_);
      else ilqr_.initializeWithReference(x_measured, x_ref, u_ref, com_ref);
      double cost = 0.0;
      if (!ilqr_.solve(x_measured, x_ref, u_ref, com_ref, cost)) {   // mpc.cpp:82-91
        u_apply = has_prev_ ? prev_ubar_[0] : Vec(ILQR_NU, 0.0);
        return false;
      }
      u_apply.assign(ILQR_NU, 0.0);
      if (ilqr_hip_compute_control(ilqr_.handle(), x_measured.data(), u_apply.data()) != ILQR_OK) throw std::runtime_error("compute_control");
      prev_xbar_ = ilqr_.xbar(); prev_ubar_ = ilqr_.ubar();
      has_prev_ = true; last_solve_cost_ = cost; ++t_idx_;
      return true;
    } catch (const std::exception&) {                                 // mpc.cpp:122-126
      u_apply.assign(ILQR_NU, 0.0);
      return false;
    }
  }
  void reset() { t_idx_ = 0; has_prev_ = false; last_solve_cost_ = 0.0; }
  double getLastSolveCost() const { return last_solve_cost_; }

 private:
  iLQR ilqr_;
  WindowFn window_;
  int t_idx_ = 0;
  bool has_prev_ = false;
  double last_solve_cost_ = 0.0;
  std::vector<Vec> prev_xbar_, prev_ubar_;
};

}  // namespace ilqr_hip
