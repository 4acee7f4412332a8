// BASELINE.json configs[0] in its AS-SHIPPED form (config.yaml: walking references q_ref2_mj / v_ref2 / contact_walking,
// gravity [0, 0, -1], N = 25, 10 iterations, tolerance 1e-4) through the C++ mirror of MPC::stepOnce (include/ilqr_hip.hpp):
// the reference window of x_ref / com_ref moves with the MPC step (MPC::extractReferenceWindow, mpc.cpp:163-166) while the
// stance flags and foot references keep the horizon-local index.  Test harness, not product code.
//   usage: cpp_mpc_walk_demo <input.bin> <output.bin>
//   input  (doubles): N, dt, steps, Q[51], R[19], Qf[51], task w[6], constraint w[2], gravity[3], x0[51],
//                     ee_ref[(N+1)*6], com_vel_ref[(N+1)*3], stance[(N+1)*2], then per step: x_ref[(N+1)*51], com_ref[(N+1)*3]
//   output (doubles): per step: ok, cost, x_measured[51], u_apply[19], ubar0[19], K0 row 0 [51], xbar[(N+1)*51], ubar[N*19]
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
  const int N = (int)in[p++]; const double dt = in[p++]; const int steps = (int)in[p++];
  const ilqr_hip::Vec Q = take(51), R = take(19), Qf = take(51), tw = take(6), cw = take(2), g = take(3), x0 = take(51);
  const ilqr_hip::Vec ee = take((size_t)(N + 1) * 6), cv = take((size_t)(N + 1) * 3), st = take((size_t)(N + 1) * 2);
  std::vector<int> stance(st.begin(), st.end());
  std::vector<std::vector<ilqr_hip::Vec>> xw(steps);
  std::vector<std::vector<std::array<double, 3>>> cwin(steps);
  for (int s = 0; s < steps; ++s) {
    for (int t = 0; t <= N; ++t) xw[s].push_back(take(51));
    for (int t = 0; t <= N; ++t) { ilqr_hip::Vec c = take(3); cwin[s].push_back({c[0], c[1], c[2]}); }
  }
  if (p != in.size()) { std::fprintf(stderr, "input size mismatch\n"); return 2; }
  try {
    const std::vector<ilqr_hip::Vec> u_zero((size_t)N, ilqr_hip::Vec(19, 0.0));
    auto window = [&](int t_idx, std::vector<ilqr_hip::Vec>& xr, std::vector<ilqr_hip::Vec>& ur, std::vector<std::array<double, 3>>& cr) {
      const int s = t_idx < steps ? t_idx : steps - 1;
      xr = xw[s]; ur = u_zero; cr = cwin[s];
    };
    ilqr_hip::MPC<decltype(window)> mpc(N, dt, window);
    ilqr_hip::iLQR& s = mpc.solver();
    s.setCostWeights(Q, R, Qf);
    s.setTaskWeights(tw[0], tw[1], tw[2], tw[3], tw[4], tw[5]);
    s.setConstraintWeights(cw[0], cw[1]);
    s.setGravity(g[0], g[1], g[2]);
    s.setContactSchedule(stance);
    s.setEEReferences(ee, &cv);
    // iLQR defaults of the reference (ilqr.cpp:16): lambda 1e-6, 10 iterations, tolerance 1e-4 -- main never changes them
    std::vector<double> out;
    ilqr_hip::Vec x = x0, u;
    for (int step = 0; step < steps; ++step) {
      const bool ok = mpc.stepOnce(x, u);
      out.push_back(ok ? 1.0 : 0.0); out.push_back(mpc.getLastSolveCost());
      out.insert(out.end(), x.begin(), x.end());
      out.insert(out.end(), u.begin(), u.end());
      const auto ub = s.ubar(); const auto K = s.gainsK();
      out.insert(out.end(), ub[0].begin(), ub[0].end());
      out.insert(out.end(), K[0].begin(), K[0].begin() + 51);
      {   // the nominal trajectory the next step warm-starts from (MPC::getNominalTrajectory, mpc.cpp:150-161)
        std::vector<ilqr_hip::Vec> xt, ut; mpc.getNominalTrajectory(xt, ut);
        if (xt.size() != (size_t)N + 1 || ut.size() != (size_t)N) return 6;
        for (const auto& r : xt) out.insert(out.end(), r.begin(), r.end());
        for (const auto& r : ut) out.insert(out.end(), r.begin(), r.end());
      }
      ilqr_hip::Vec xn(51);      // plant: the model's own step through the C ABI (plant and solver state stay separate, Appendix D #15)
      if (ilqr_hip_step(s.handle(), 1, x.data(), u.data(), xn.data()) != ILQR_OK) return 3;
      x = xn;
    }
    if (mpc.getTimeIndex() != steps) return 5;
    std::FILE* o = std::fopen(argv[2], "wb");
    std::fwrite(out.data(), sizeof(double), out.size(), o);
    std::fclose(o);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "cpp_mpc_walk_demo: %s\n", e.what());
    return 1;
  }
  return 0;
}
