// Drives the C++ mirror of the reference's classes (include/ilqr_hip.hpp: ilqr_hip::iLQR / ilqr_hip::MPC) the way
// main/humanoid_mpc.cpp drives iLQR / MPC (setupSimulation :94-118, runSimulation :126-190): configure, two MPC steps
// (cold start, then warm start), dump costs / first controls / first gain rows.  Test harness, not product code.
//   usage: cpp_api_demo <input.bin> <output.bin> [log_dir]
//   with log_dir: mpc_log.csv, q_optimal.csv, u_optimal.csv in the reference's formats (mpc.cpp:181-355) and the profiling
//   table (main/humanoid_mpc.cpp:195-226) on stdout
//   input  (doubles): N, dt, Q[51], R[19], Qf[51], task w[6], constraint w[2], gravity[3], x0[51],
//                     x_ref[(N+1)*51], com_ref[(N+1)*3], ee_ref[(N+1)*6], stance[(N+1)*2]
//   output (doubles): per step: ok, cost, u_apply[19], ubar0[19], K0 row 0 [51]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ilqr_hip.hpp"

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
  std::vector<double> in;
  double v;
  while (std::fread(&v, sizeof(double), 1, f) == 1) in.push_back(v);
  std::fclose(f);
  size_t p = 0;
  auto take = [&](size_t n) { ilqr_hip::Vec o(in.begin() + p, in.begin() + p + n); p += n; return o; };
  const int N = (int)in[p++]; const double dt = in[p++];
  const ilqr_hip::Vec Q = take(51), R = take(19), Qf = take(51), tw = take(6), cw = take(2), g = take(3), x0 = take(51);
  std::vector<ilqr_hip::Vec> x_ref, u_ref; std::vector<std::array<double, 3>> com_ref;
  for (int t = 0; t <= N; ++t) x_ref.push_back(take(51));
  for (int t = 0; t < N; ++t) u_ref.push_back(ilqr_hip::Vec(19, 0.0));
  for (int t = 0; t <= N; ++t) { ilqr_hip::Vec c = take(3); com_ref.push_back({c[0], c[1], c[2]}); }
  const ilqr_hip::Vec ee = take((size_t)(N + 1) * 6);
  const ilqr_hip::Vec st = take((size_t)(N + 1) * 2);
  std::vector<int> stance(st.begin(), st.end());
  try {
    auto window = [&](int, std::vector<ilqr_hip::Vec>& xr, std::vector<ilqr_hip::Vec>& ur, std::vector<std::array<double, 3>>& cr) { xr = x_ref; ur = u_ref; cr = com_ref; };
    ilqr_hip::MPC<decltype(window)> mpc(N, dt, window);
    ilqr_hip::iLQR& s = mpc.solver();
    {   // Config::buildCostMatrices with the shipped weights must reproduce the diagonals the harness was given
      ilqr_hip::Vec q2, r2, qf2; ilqr_hip::buildCostMatrices(ilqr_hip::CostConfig(), q2, r2, qf2);
      if (q2 != Q || r2 != R || qf2 != Qf) { std::fprintf(stderr, "buildCostMatrices differs from the shipped configuration\n"); return 4; }
    }
    if (argc > 3) {
      mpc.enableCSVLogging(std::string(argv[3]) + "/mpc_log.csv");
      mpc.enableOptimalTrajectoryLogging(argv[3]);
      mpc.enableProfiling(true);
    }
    s.setCostWeights(Q, R, Qf);
    s.setTaskWeights(tw[0], tw[1], tw[2], tw[3], tw[4], tw[5]);
    s.setConstraintWeights(cw[0], cw[1]);
    s.setGravity(g[0], g[1], g[2]);
    s.setContactSchedule(stance);
    s.setEEReferences(ee);
    s.setMaxIterations(3);
    std::vector<double> out;
    ilqr_hip::Vec x = x0, u;
    for (int step = 0; step < 2; ++step) {
      const bool ok = mpc.stepOnce(x, u);
      out.push_back(ok ? 1.0 : 0.0); out.push_back(mpc.getLastSolveCost());
      out.insert(out.end(), u.begin(), u.end());
      const auto ub = s.ubar(); const auto K = s.gainsK();
      out.insert(out.end(), ub[0].begin(), ub[0].end());
      out.insert(out.end(), K[0].begin(), K[0].begin() + 51);
      // plant: the model's own step through the C ABI
      ilqr_hip::Vec xn(51);
      if (ilqr_hip_step(s.handle(), 1, x.data(), u.data(), xn.data()) != ILQR_OK) return 3;
      x = xn;
    }
    {   // accessors of include/ilqr/mpc.hpp:27-28, 41-47
      std::vector<ilqr_hip::Vec> xt, ut; mpc.getNominalTrajectory(xt, ut);
      if (mpc.getTimeIndex() != 2 || xt.size() != (size_t)N + 1 || ut.size() != (size_t)N || mpc.gainsK().size() != (size_t)N || mpc.gainsK()[0].size() != 19u * 51u) return 5;
      const ilqr_hip::Vec utv = mpc.computeTVLQRControl(xt[0]);
      for (int i = 0; i < 19; ++i) if (utv[i] != ut[0][i]) return 6;          // zero state error: u = ubar[0]
      mpc.setTimeIndex(7); if (mpc.getTimeIndex() != 7) return 7;
    }
    if (argc > 3) { mpc.finalizeCSVLog(); mpc.finalizeOptimalTrajectoryLog(); mpc.profiler().print(std::cout); }
    std::FILE* o = std::fopen(argv[2], "wb");
    std::fwrite(out.data(), sizeof(double), out.size(), o);
    std::fclose(o);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "cpp_api_demo: %s\n", e.what());
    return 1;
  }
  return 0;
}
