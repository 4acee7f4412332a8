// Exact-signature drop-in for the reference's solver classes (Eigen types, RobotUtils in the constructor), over ilqr_hip.hpp / the C ABI:
//
//   ilqr_hip_eigen::iLQR<Robot>   iLQR(Robot&, int N, double dt, const std::string& urdf_path)            reference include/ilqr/ilqr.hpp:19
//                                 solve(const Eigen::VectorXd&, const std::vector<Eigen::VectorXd>&, ..., double&)           ilqr.hpp:27-31
//                                 initializeWithReference(..., prev_xbar, prev_ubar), xbar() / ubar() / gainsK() / gainsKff()  ilqr.hpp:34-45
//   ilqr_hip_eigen::MPC<Robot>    MPC(Robot&, int N, double dt, const std::string& urdf_path)              reference include/ilqr/mpc.hpp:20
//                                 stepOnce(const Eigen::VectorXd&, Eigen::VectorXd&), reset, logging, gainsK() ...           mpc.hpp:23-47
//
// `Robot` is the reference's RobotUtils (include/common/robot_utils.hpp) or anything with the same getters: the adapter pulls the
// problem data out of them exactly where the reference's iLQR reads them --
//   Q(), R(), Qf()                                           robot_utils.hpp:36-38   (ilqr.cpp:145-150, 204-205)
//   getCoMWeight() ... getBalanceWeight()                    robot_utils.hpp:60-71   (ilqr.cpp:154-179)
//   getEEReference(t, ee), getEEVelReference(t, ee), getCoMVelReference(t), isStance(ee, t)   robot_utils.hpp:97-105, with the
//       window-local knot index t = 0..N the reference passes (ilqr.cpp:403-434: not the MPC time index)
//   getReferenceWindow(t0, N, x_ref, u_ref, com_ref)         robot_utils.hpp:90-93   (mpc.cpp extractReferenceWindow)
//   model()->opt.gravity                                     robot_utils.hpp:110     (MuJoCo; detected at compile time)
// RobotUtils keeps its two constraint weights private (setConstraintWeights, robot_utils.hpp:78, has no getter): the adapter uses
// getJointLimitWeight() / getControlLimitWeight() when the robot class has them and the shipped config.yaml values otherwise
// (setConstraintWeights on the adapter overrides).  With this header, src/ilqr/mpc.cpp and main/humanoid_mpc.cpp compile against the
// HIP solver with an include swap and two aliases:
//     #include "ilqr_hip_eigen.hpp"
//     using iLQR = ilqr_hip_eigen::iLQR<RobotUtils>;   using MPC = ilqr_hip_eigen::MPC<RobotUtils>;
// Compiled only where Eigen is installed (this image has none: tests/cpp/fake_eigen holds the few members the adapter touches, for the
// compile-and-run test tests/cpp/cpp_eigen_drop_in_demo.cpp).
#pragma once
#if !__has_include(<Eigen/Dense>)
#error "ilqr_hip_eigen.hpp needs Eigen (<Eigen/Dense>); without it use ilqr_hip.hpp (std::vector interface)"
#else
#include <Eigen/Dense>

#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "ilqr_hip.hpp"

namespace ilqr_hip_eigen {

namespace detail {
template <class R, class = void> struct has_limit_getters : std::false_type {};
template <class R> struct has_limit_getters<R, std::void_t<decltype(std::declval<const R&>().getJointLimitWeight()), decltype(std::declval<const R&>().getControlLimitWeight())>> : std::true_type {};
template <class R, class = void> struct has_mj_model : std::false_type {};
template <class R> struct has_mj_model<R, std::void_t<decltype(std::declval<const R&>().model()->opt.gravity[2])>> : std::true_type {};
inline ilqr_hip::Vec to_vec(const Eigen::VectorXd& v) { ilqr_hip::Vec o((size_t)v.size()); for (int i = 0; i < (int)v.size(); ++i) o[(size_t)i] = v(i); return o; }
inline Eigen::VectorXd to_eigen(const ilqr_hip::Vec& v) { Eigen::VectorXd o((int)v.size()); for (size_t i = 0; i < v.size(); ++i) o((int)i) = v[i]; return o; }
inline std::vector<ilqr_hip::Vec> to_vecs(const std::vector<Eigen::VectorXd>& v) { std::vector<ilqr_hip::Vec> o; o.reserve(v.size()); for (const auto& e : v) o.push_back(to_vec(e)); return o; }
inline std::vector<std::array<double, 3>> to_arr3(const std::vector<Eigen::Vector3d>& v) { std::vector<std::array<double, 3>> o; o.reserve(v.size()); for (const auto& e : v) o.push_back({{e(0), e(1), e(2)}}); return o; }
}  // namespace detail

template <class Robot>
class iLQR {
 public:
  // reference include/ilqr/ilqr.hpp:19 (urdf_path: the H1 model is compiled into the library; kept for the signature)
  iLQR(Robot& robot, int N, double dt, const std::string& /*urdf_path*/, int device = 0) : robot_(robot), impl_(N, dt, 1, device), N_(N) { pullProblem(); }

  void setRegularization(double lambda) { impl_.setRegularization(lambda); }                    // ilqr.hpp:22-24
  void setMaxIterations(int max_iter) { impl_.setMaxIterations(max_iter); }
  void setTolerance(double tol) { impl_.setTolerance(tol); }
  void setConstraintWeights(double w_joint_limits, double w_control_limits) { w_joint_ = w_joint_limits; w_ctrl_ = w_control_limits; have_limits_ = true; }
  ilqr_hip::iLQR& impl() { return impl_; }

  // ilqr.hpp:27-31
  bool solve(const Eigen::VectorXd& x0, const std::vector<Eigen::VectorXd>& x_ref, const std::vector<Eigen::VectorXd>& u_ref,
             const std::vector<Eigen::Vector3d>& com_ref, double& cost_out) {
    pullProblem();
    const bool ok = impl_.solve(detail::to_vec(x0), detail::to_vecs(x_ref), detail::to_vecs(u_ref), detail::to_arr3(com_ref), cost_out);
    if (ok) fetch();
    return ok;
  }
  // ilqr.hpp:40-45
  void initializeWithReference(const Eigen::VectorXd& x0, const std::vector<Eigen::VectorXd>& x_ref, const std::vector<Eigen::VectorXd>& u_ref,
                               const std::vector<Eigen::Vector3d>& com_ref, const std::vector<Eigen::VectorXd>* prev_xbar = nullptr,
                               const std::vector<Eigen::VectorXd>* prev_ubar = nullptr) {
    pullProblem();
    if (prev_xbar && prev_ubar) {
      const std::vector<ilqr_hip::Vec> px = detail::to_vecs(*prev_xbar), pu = detail::to_vecs(*prev_ubar);
      impl_.initializeWithReference(detail::to_vec(x0), detail::to_vecs(x_ref), detail::to_vecs(u_ref), detail::to_arr3(com_ref), &px, &pu);
    } else {
      impl_.initializeWithReference(detail::to_vec(x0), detail::to_vecs(x_ref), detail::to_vecs(u_ref), detail::to_arr3(com_ref));
    }
    fetch();
  }
  // ilqr.hpp:34-37
  const std::vector<Eigen::VectorXd>& xbar() const { return xbar_; }
  const std::vector<Eigen::VectorXd>& ubar() const { return ubar_; }
  const std::vector<Eigen::MatrixXd>& gainsK() const { return K_; }
  const std::vector<Eigen::VectorXd>& gainsKff() const { return kff_; }

 private:
  // everything the reference's iLQR reads from RobotUtils while it solves, re-read before every call (the getters are cheap and the
  // application may change weights or load references after constructing the solver, as main/humanoid_mpc.cpp does)
  void pullProblem() {
    ilqr_hip::Vec q(ILQR_NX), r(ILQR_NU), qf(ILQR_NX);
    for (int i = 0; i < ILQR_NX; ++i) { q[(size_t)i] = robot_.Q()(i, i); qf[(size_t)i] = robot_.Qf()(i, i); }
    for (int i = 0; i < ILQR_NU; ++i) r[(size_t)i] = robot_.R()(i, i);
    impl_.setCostWeights(q, r, qf);
    impl_.setTaskWeights(robot_.getCoMWeight(), robot_.getCoMVelWeight(), robot_.getEEPosWeight(), robot_.getEEVelWeight(), robot_.getUprightWeight(), robot_.getBalanceWeight());
    if constexpr (detail::has_limit_getters<Robot>::value) { if (!have_limits_) { w_joint_ = robot_.getJointLimitWeight(); w_ctrl_ = robot_.getControlLimitWeight(); } }
    impl_.setConstraintWeights(w_joint_, w_ctrl_);
    if constexpr (detail::has_mj_model<Robot>::value) { const auto* m = robot_.model(); if (m) impl_.setGravity(m->opt.gravity[0], m->opt.gravity[1], m->opt.gravity[2]); }
    // end-effector / CoM-velocity references and the contact schedule at the window-local knots 0..N (ilqr.cpp:403-434, 662-800)
    ilqr_hip::Vec ee((size_t)(N_ + 1) * 6, 0.0), cv((size_t)(N_ + 1) * 3, 0.0);
    std::vector<int> stance((size_t)(N_ + 1) * 2, 1);
    bool have_refs = true;
    for (int t = 0; t <= N_ && have_refs; ++t) {
      try {
        for (int e = 0; e < 2; ++e) { const Eigen::Vector3d p = robot_.getEEReference(t, e); for (int k = 0; k < 3; ++k) ee[(size_t)t * 6 + 3 * e + k] = p(k); stance[(size_t)t * 2 + e] = robot_.isStance(e, t) ? 1 : 0; }
        const Eigen::Vector3d c = robot_.getCoMVelReference(t); for (int k = 0; k < 3; ++k) cv[(size_t)t * 3 + k] = c(k);
      } catch (const std::exception&) { have_refs = false; }      // (references not loaded yet: the reference's getters throw)
    }
    if (have_refs) { impl_.setEEReferences(ee, &cv); impl_.setContactSchedule(stance); }
  }
  void fetch() {
    const auto xb = impl_.xbar(), ub = impl_.ubar(), kk = impl_.gainsK(), kf = impl_.gainsKff();
    xbar_.clear(); ubar_.clear(); K_.clear(); kff_.clear();
    for (const auto& v : xb) xbar_.push_back(detail::to_eigen(v));
    for (const auto& v : ub) ubar_.push_back(detail::to_eigen(v));
    for (const auto& v : kf) kff_.push_back(detail::to_eigen(v));
    for (const auto& v : kk) { Eigen::MatrixXd m(ILQR_NU, ILQR_NX); for (int i = 0; i < ILQR_NU; ++i) for (int j = 0; j < ILQR_NX; ++j) m(i, j) = v[(size_t)i * ILQR_NX + j]; K_.push_back(m); }
  }
  Robot& robot_;
  ilqr_hip::iLQR impl_;
  int N_;
  double w_joint_ = 1500.0, w_ctrl_ = 1500.0;      // shipped config.yaml constraint weights (main/humanoid_mpc.cpp passes them to RobotUtils only)
  bool have_limits_ = false;
  std::vector<Eigen::VectorXd> xbar_, ubar_, kff_;
  std::vector<Eigen::MatrixXd> K_;
};

// reference include/ilqr/mpc.hpp:18-47 / src/ilqr/mpc.cpp:40-127: same control flow as ilqr_hip::MPC (which it reuses for the logs)
template <class Robot>
class MPC {
  struct Window {
    Robot* robot; int N;
    void operator()(int t_idx, std::vector<ilqr_hip::Vec>& x_ref, std::vector<ilqr_hip::Vec>& u_ref, std::vector<std::array<double, 3>>& com_ref) const {
      std::vector<Eigen::VectorXd> xr, ur; std::vector<Eigen::Vector3d> cr;
      robot->getReferenceWindow(t_idx, N, xr, ur, cr);                               // mpc.cpp extractReferenceWindow
      x_ref = detail::to_vecs(xr); u_ref = detail::to_vecs(ur); com_ref = detail::to_arr3(cr);
    }
  };

 public:
  MPC(Robot& robot, int N, double dt, const std::string& /*urdf_path*/, int device = 0) : robot_(robot), impl_(N, dt, Window{&robot, N}, device), N_(N) {}

  // mpc.hpp:23
  bool stepOnce(const Eigen::VectorXd& x_measured, Eigen::VectorXd& u_apply) {
    pullProblem();
    ilqr_hip::Vec u;
    const bool ok = impl_.stepOnce(detail::to_vec(x_measured), u);
    u_apply = detail::to_eigen(u);
    return ok;
  }
  void reset() { impl_.reset(); }
  void setTimeIndex(int t_idx) { impl_.setTimeIndex(t_idx); }
  int getTimeIndex() const { return impl_.getTimeIndex(); }
  void enableCSVLogging(const std::string& filename) { impl_.enableCSVLogging(filename); }
  void logCurrentStep(const Eigen::VectorXd& x_measured, const Eigen::VectorXd& u_applied) { impl_.logCurrentStep(detail::to_vec(x_measured), detail::to_vec(u_applied)); }
  void finalizeCSVLog() { impl_.finalizeCSVLog(); }
  void enableOptimalTrajectoryLogging(const std::string& base_path) { impl_.enableOptimalTrajectoryLogging(base_path); }
  void logAppliedOptimal(const Eigen::VectorXd& x_applied, const Eigen::VectorXd& u_applied) { impl_.logAppliedOptimal(detail::to_vec(x_applied), detail::to_vec(u_applied)); }
  void finalizeOptimalTrajectoryLog() { impl_.finalizeOptimalTrajectoryLog(); }
  double getLastSolveCost() const { return impl_.getLastSolveCost(); }
  std::vector<Eigen::MatrixXd> gainsK() const {
    std::vector<Eigen::MatrixXd> out;
    for (const auto& v : impl_.gainsK()) { Eigen::MatrixXd m(ILQR_NU, ILQR_NX); for (int i = 0; i < ILQR_NU; ++i) for (int j = 0; j < ILQR_NX; ++j) m(i, j) = v[(size_t)i * ILQR_NX + j]; out.push_back(m); }
    return out;
  }
  void getNominalTrajectory(std::vector<Eigen::VectorXd>& x_traj, std::vector<Eigen::VectorXd>& u_traj) const {
    std::vector<ilqr_hip::Vec> x, u; impl_.getNominalTrajectory(x, u);
    x_traj.clear(); u_traj.clear();
    for (const auto& v : x) x_traj.push_back(detail::to_eigen(v));
    for (const auto& v : u) u_traj.push_back(detail::to_eigen(v));
  }
  void setConstraintWeights(double w_joint_limits, double w_control_limits) { w_joint_ = w_joint_limits; w_ctrl_ = w_control_limits; have_limits_ = true; }
  ilqr_hip::MPC<Window>& impl() { return impl_; }

 private:
  void pullProblem() {
    ilqr_hip::iLQR& s = impl_.solver();
    ilqr_hip::Vec q(ILQR_NX), r(ILQR_NU), qf(ILQR_NX);
    for (int i = 0; i < ILQR_NX; ++i) { q[(size_t)i] = robot_.Q()(i, i); qf[(size_t)i] = robot_.Qf()(i, i); }
    for (int i = 0; i < ILQR_NU; ++i) r[(size_t)i] = robot_.R()(i, i);
    s.setCostWeights(q, r, qf);
    s.setTaskWeights(robot_.getCoMWeight(), robot_.getCoMVelWeight(), robot_.getEEPosWeight(), robot_.getEEVelWeight(), robot_.getUprightWeight(), robot_.getBalanceWeight());
    if constexpr (detail::has_limit_getters<Robot>::value) { if (!have_limits_) { w_joint_ = robot_.getJointLimitWeight(); w_ctrl_ = robot_.getControlLimitWeight(); } }
    s.setConstraintWeights(w_joint_, w_ctrl_);
    if constexpr (detail::has_mj_model<Robot>::value) { const auto* m = robot_.model(); if (m) s.setGravity(m->opt.gravity[0], m->opt.gravity[1], m->opt.gravity[2]); }
    ilqr_hip::Vec ee((size_t)(N_ + 1) * 6, 0.0), cv((size_t)(N_ + 1) * 3, 0.0);
    std::vector<int> stance((size_t)(N_ + 1) * 2, 1);
    // (a RobotUtils whose end-effector / CoM-velocity tables are not loaded throws from these getters; the reference catches that and
    // carries on without the term, ilqr.cpp:401-434, 682-693, 703-720 -- so does this, like iLQR::pullProblem above)
    bool have_refs = true;
    for (int t = 0; t <= N_ && have_refs; ++t) {
      try {
        for (int e = 0; e < 2; ++e) { const Eigen::Vector3d p = robot_.getEEReference(t, e); for (int k = 0; k < 3; ++k) ee[(size_t)t * 6 + 3 * e + k] = p(k); stance[(size_t)t * 2 + e] = robot_.isStance(e, t) ? 1 : 0; }
        const Eigen::Vector3d c = robot_.getCoMVelReference(t); for (int k = 0; k < 3; ++k) cv[(size_t)t * 3 + k] = c(k);
      } catch (const std::exception&) { have_refs = false; }
    }
    if (have_refs) { s.setEEReferences(ee, &cv); s.setContactSchedule(stance); }
  }
  Robot& robot_;
  ilqr_hip::MPC<Window> impl_;
  int N_;
  double w_joint_ = 1500.0, w_ctrl_ = 1500.0;
  bool have_limits_ = false;
};

}  // namespace ilqr_hip_eigen
#endif
