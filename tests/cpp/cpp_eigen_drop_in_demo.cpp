// Compile-and-run check of include/ilqr_hip_eigen.hpp: the reference's call sites, unchanged in shape --
//     MPC mpc(robot, N, dt, urdf_path);  mpc.stepOnce(x_measured, u_apply);  mpc.solver-like accessors           (main/humanoid_mpc.cpp:94-190)
//     iLQR solver(robot, N, dt, urdf_path);  solver.initializeWithReference(...);  solver.solve(x0, x_ref, u_ref, com_ref, cost)   (mpc.cpp:60-80)
// against a RobotUtils-shaped class (the getters of include/common/robot_utils.hpp the adapter reads; data from the same input file
// tests/cpp/cpp_api_demo.cpp takes) and the stand-in Eigen of tests/cpp/fake_eigen.  Same input file, same output layout as
// cpp_api_demo: the -m gpu test requires the two to agree bit for bit.  Test harness, not product code.
//   usage: cpp_eigen_drop_in_demo <input.bin> <output.bin>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "ilqr_hip_eigen.hpp"

// the slice of RobotUtils (reference include/common/robot_utils.hpp) the solver classes read
class RobotUtils {
 public:
  Eigen::MatrixXd Q_, R_, Qf_;
  double w[6] = {0, 0, 0, 0, 0, 0}, wj = 0.0, wc = 0.0;
  std::vector<Eigen::VectorXd> x_ref_full_, u_ref_full_;
  std::vector<Eigen::Vector3d> com_ref_full_;
  std::vector<std::vector<Eigen::Vector3d>> ee_pos_ref_full_;
  std::vector<std::vector<int>> contact_schedule_;
  struct FakeOpt { double gravity[3]; };
  struct FakeModel { FakeOpt opt; } model_;
  const Eigen::MatrixXd& Q() const { return Q_; }
  const Eigen::MatrixXd& R() const { return R_; }
  const Eigen::MatrixXd& Qf() const { return Qf_; }
  double getCoMWeight() const { return w[0]; }
  double getCoMVelWeight() const { return w[1]; }
  double getEEPosWeight() const { return w[2]; }
  double getEEVelWeight() const { return w[3]; }
  double getUprightWeight() const { return w[4]; }
  double getBalanceWeight() const { return w[5]; }
  double getJointLimitWeight() const { return wj; }            // (the two getters INTEGRATION.md asks a maintainer to add)
  double getControlLimitWeight() const { return wc; }
  const FakeModel* model() const { return &model_; }
  bool isStance(int ee_idx, int t) const { return t < 0 || t >= (int)contact_schedule_.size() ? true : contact_schedule_[(size_t)t][(size_t)ee_idx] == 1; }
  Eigen::Vector3d getEEReference(int t, int ee_idx) const { if (t >= (int)ee_pos_ref_full_.size()) throw std::runtime_error("Invalid reference index"); return ee_pos_ref_full_[(size_t)t][(size_t)ee_idx]; }
  Eigen::Vector3d getEEVelReference(int, int) const { return Eigen::Vector3d::Zero(); }
  Eigen::Vector3d getCoMVelReference(int) const { return Eigen::Vector3d::Zero(); }
  void getReferenceWindow(int, int, std::vector<Eigen::VectorXd>& x, std::vector<Eigen::VectorXd>& u, std::vector<Eigen::Vector3d>& c) const { x = x_ref_full_; u = u_ref_full_; c = com_ref_full_; }
};
using iLQR = ilqr_hip_eigen::iLQR<RobotUtils>;
using MPC = ilqr_hip_eigen::MPC<RobotUtils>;

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<double> in; double v;
  while (std::fread(&v, sizeof(double), 1, f) == 1) in.push_back(v);
  std::fclose(f);
  size_t p = 0;
  const int N = (int)in[p++]; const double dt = in[p++];
  RobotUtils robot;
  robot.Q_ = Eigen::MatrixXd::Zero(51, 51); robot.R_ = Eigen::MatrixXd::Zero(19, 19); robot.Qf_ = Eigen::MatrixXd::Zero(51, 51);
  for (int i = 0; i < 51; ++i) robot.Q_(i, i) = in[p++];
  for (int i = 0; i < 19; ++i) robot.R_(i, i) = in[p++];
  for (int i = 0; i < 51; ++i) robot.Qf_(i, i) = in[p++];
  for (int i = 0; i < 6; ++i) robot.w[i] = in[p++];
  robot.wj = in[p++]; robot.wc = in[p++];
  for (int i = 0; i < 3; ++i) robot.model_.opt.gravity[i] = in[p++];
  Eigen::VectorXd x0(51); for (int i = 0; i < 51; ++i) x0(i) = in[p++];
  for (int t = 0; t <= N; ++t) { Eigen::VectorXd x(51); for (int i = 0; i < 51; ++i) x(i) = in[p++]; robot.x_ref_full_.push_back(x); }
  for (int t = 0; t < N; ++t) robot.u_ref_full_.push_back(Eigen::VectorXd::Zero(19));
  for (int t = 0; t <= N; ++t) { robot.com_ref_full_.push_back(Eigen::Vector3d(in[p], in[p + 1], in[p + 2])); p += 3; }
  for (int t = 0; t <= N; ++t) { std::vector<Eigen::Vector3d> e; for (int k = 0; k < 2; ++k) { e.push_back(Eigen::Vector3d(in[p], in[p + 1], in[p + 2])); p += 3; } robot.ee_pos_ref_full_.push_back(e); }
  for (int t = 0; t <= N; ++t) { robot.contact_schedule_.push_back({(int)in[p], (int)in[p + 1]}); p += 2; }
  try {
    const std::string urdf_path = "robots/h1_description/urdf/h1.urdf";
    MPC mpc(robot, N, dt, urdf_path);                           // main/humanoid_mpc.cpp: MPC mpc(robot, N, dt, urdf)
    mpc.impl().solver().setMaxIterations(3);
    std::vector<double> out;
    Eigen::VectorXd x = x0, u;
    for (int step = 0; step < 2; ++step) {
      const bool ok = mpc.stepOnce(x, u);                        // mpc.hpp:23
      out.push_back(ok ? 1.0 : 0.0); out.push_back(mpc.getLastSolveCost());
      for (int i = 0; i < 19; ++i) out.push_back(u(i));
      std::vector<Eigen::VectorXd> xt, ut; mpc.getNominalTrajectory(xt, ut);
      for (int i = 0; i < 19; ++i) out.push_back(ut[0](i));
      const std::vector<Eigen::MatrixXd> K = mpc.gainsK();
      if (K.size() != (size_t)N || K[0].rows() != 19 || K[0].cols() != 51) return 5;
      for (int j = 0; j < 51; ++j) out.push_back(K[0](0, j));
      Eigen::VectorXd xn(51);
      if (ilqr_hip_step(mpc.impl().solver().handle(), 1, x.data(), u.data(), xn.data()) != ILQR_OK) return 3;
      x = xn;
    }
    {   // the solver class on its own, as mpc.cpp drives it: cold start, solve, accessors with the reference's types
      iLQR solver(robot, N, dt, urdf_path);
      solver.setMaxIterations(3); solver.setRegularization(1e-6); solver.setTolerance(1e-4);
      solver.initializeWithReference(x0, robot.x_ref_full_, robot.u_ref_full_, robot.com_ref_full_);
      double cost = 0.0;
      if (!solver.solve(x0, robot.x_ref_full_, robot.u_ref_full_, robot.com_ref_full_, cost)) return 6;
      if (solver.xbar().size() != (size_t)N + 1 || solver.ubar().size() != (size_t)N || solver.gainsK().size() != (size_t)N || solver.gainsKff().size() != (size_t)N) return 7;
      if (!(cost == out[1])) { std::fprintf(stderr, "iLQR::solve cost %.17g vs the MPC's first step %.17g\n", cost, out[1]); return 8; }
      for (int i = 0; i < 19; ++i) if (solver.ubar()[0](i) != out[2 + 19 + i]) return 9;
      // a wrongly sized reference window is refused as the reference refuses it (ilqr.cpp:526-532)
      std::vector<Eigen::VectorXd> shortx(robot.x_ref_full_.begin(), robot.x_ref_full_.end() - 1);
      if (solver.solve(x0, shortx, robot.u_ref_full_, robot.com_ref_full_, cost)) return 10;
    }
    std::FILE* o = std::fopen(argv[2], "wb");
    std::fwrite(out.data(), sizeof(double), out.size(), o);
    std::fclose(o);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "cpp_eigen_drop_in_demo: %s\n", e.what());
    return 1;
  }
  return 0;
}
