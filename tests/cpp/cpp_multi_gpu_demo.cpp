// Multi-GPU use of the C ABI without Python: one host thread and one handle per GPU, the global batch cut
// into contiguous shards, ONE RCCL gather of the first-knot payload [u0 | cost | K0] to rank 0 per MPC step
// (ilqr_hip_comm_* / ilqr_hip_gather_first_knot in include/ilqr_hip.h).  The consumer of the gathered rows in the reference
// is MPC::stepOnce (src/ilqr/mpc.cpp:97-113).  Test harness, not product code.
//   usage: cpp_multi_gpu_demo <world> <batch_per_gpu> <with_gains 0|1> <out.bin> [device_of_every_rank [root]]
//   Every rank draws its shard from one global, deterministic batch; the root (default 0) writes the gathered [world * B][width] rows.
//   With the optional fifth argument all ranks share that device: world must then be 1 (RCCL refuses duplicate GPUs) unless the
//   stand-in library tests/cpp/fake_rccl.cpp is selected with ILQR_RCCL_LIB (the -m gpu test of the world > 1 gather branch).
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "ilqr_hip.hpp"

namespace {
// deterministic global batch: rollout g of the global batch, independent of the number of GPUs
void make_rollout(int g, int N, double* x0, double* u) {
  unsigned long long s = 0x9E3779B97F4A7C15ull * (unsigned long long)(g + 1);
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0; };
  for (int i = 0; i < ILQR_NX; ++i) x0[i] = 0.0;
  x0[2] = 1.0432;
  for (int i = 0; i < 3; ++i) x0[i] += 0.02 * rnd();
  double w[3] = {0.05 * rnd(), 0.05 * rnd(), 0.05 * rnd()};
  const double ang = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]), sh = ang > 1e-12 ? std::sin(0.5 * ang) / ang : 0.5;
  x0[3] = std::cos(0.5 * ang); x0[4] = sh * w[0]; x0[5] = sh * w[1]; x0[6] = sh * w[2];
  for (int i = 7; i < ILQR_NQ; ++i) x0[i] = 0.05 * rnd();
  for (int i = ILQR_NQ; i < ILQR_NX; ++i) x0[i] = 0.1 * rnd();
  for (int t = 0; t < N; ++t) for (int i = 0; i < ILQR_NU; ++i) u[t * ILQR_NU + i] = rnd();
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: %s world batch_per_gpu with_gains out.bin [device]\n", argv[0]); return 2; }
  const int world = std::atoi(argv[1]), B = std::atoi(argv[2]), with_gains = std::atoi(argv[3]);
  const int forced_dev = argc > 5 ? std::atoi(argv[5]) : -1;
  const int root = argc > 6 ? std::atoi(argv[6]) : 0;
  if (root < 0 || root >= world) { std::fprintf(stderr, "root out of range\n"); return 2; }
  const int N = 25; const double dt = 0.02;
  const int W = ilqr_hip_payload_width(with_gains);
  char id[ILQR_COMM_ID_BYTES] = {0};
  if (world > 1 && ilqr_hip_comm_get_unique_id(id) != ILQR_OK) { std::fprintf(stderr, "no RCCL\n"); return 3; }
  std::vector<double> gathered((size_t)world * B * W, 0.0);
  std::atomic<int> failures{0};
  auto rank_main = [&](int rank) {
    try {
      ilqr_hip::iLQR s(N, dt, B, forced_dev >= 0 ? forced_dev : rank);
      ilqr_hip_ctx* c = s.handle();
      ilqr_hip::Vec Q, R, Qf; ilqr_hip::buildCostMatrices(ilqr_hip::CostConfig(), Q, R, Qf);
      s.setCostWeights(Q, R, Qf);
      s.setTaskWeights(100.0, 0.0, 400.0, 400.0, 20.0, 30.0);
      s.setConstraintWeights(1500.0, 1500.0);
      s.setGravity(0.0, 0.0, -1.0);
      s.setMaxIterations(3);
      // standing reference window shared by all rollouts
      ilqr_hip::Vec xs(ILQR_NX, 0.0); xs[2] = 1.0432; xs[3] = 1.0;
      double com[3], ee[6];
      if (ilqr_hip_reference_kinematics(xs.data(), com, ee) != ILQR_OK) throw std::runtime_error("kinematics");
      std::vector<double> xr((size_t)(N + 1) * ILQR_NX), ur((size_t)N * ILQR_NU, 0.0), cr((size_t)(N + 1) * 3), er((size_t)(N + 1) * 6);
      for (int t = 0; t <= N; ++t) { std::memcpy(&xr[(size_t)t * ILQR_NX], xs.data(), sizeof(double) * ILQR_NX); std::memcpy(&cr[(size_t)t * 3], com, sizeof(com)); std::memcpy(&er[(size_t)t * 6], ee, sizeof(ee)); }
      if (ilqr_hip_set_ee_references(c, er.data(), nullptr, 1) || ilqr_hip_set_references(c, xr.data(), ur.data(), cr.data(), 1)) throw std::runtime_error("references");
      // this rank's contiguous shard [rank B, (rank + 1) B) of the global batch
      std::vector<double> x0((size_t)B * ILQR_NX), ui((size_t)B * N * ILQR_NU);
      for (int b = 0; b < B; ++b) make_rollout(rank * B + b, N, &x0[(size_t)b * ILQR_NX], &ui[(size_t)b * N * ILQR_NU]);
      if (ilqr_hip_comm_init(c, world, rank, world > 1 ? id : nullptr) != ILQR_OK) throw std::runtime_error(std::string("comm_init: ") + ilqr_hip_last_error(c));
      if (ilqr_hip_initialize(c, x0.data(), ui.data(), nullptr, nullptr) != ILQR_OK || ilqr_hip_solve(c, x0.data(), nullptr) != ILQR_OK)
        throw std::runtime_error(std::string("solve: ") + ilqr_hip_last_error(c));
      double* recv = nullptr;
      if (rank == root && hipMalloc((void**)&recv, gathered.size() * sizeof(double)) != hipSuccess) throw std::runtime_error("hipMalloc");
      if (ilqr_hip_gather_first_knot(c, root, with_gains, recv) != ILQR_OK || ilqr_hip_synchronize(c) != ILQR_OK)
        throw std::runtime_error(std::string("gather: ") + ilqr_hip_last_error(c));
      if (rank == root) {
        if (hipMemcpy(gathered.data(), recv, gathered.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("hipMemcpy");
        (void)hipFree(recv);
      }
      ilqr_hip_comm_destroy(c);
    } catch (const std::exception& e) {
      std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
      failures++;
    }
  };
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r) th.emplace_back(rank_main, r);
  for (auto& t : th) t.join();
  if (failures) return 1;
  std::FILE* o = std::fopen(argv[4], "wb");
  if (!o) return 2;
  std::fwrite(gathered.data(), sizeof(double), gathered.size(), o);
  std::fclose(o);
  std::printf("gathered %d x %d rollouts, payload width %d\n", world, B, W);
  return 0;
}
