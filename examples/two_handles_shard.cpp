// two_handles_shard.cpp -- the multi-GPU split of a batch of planner instances from a C++ host, through the C ABI only (include/mpc_hip.h).
//
// BASELINE.json's north star shards "batches of independent planner instances" over the GPUs of one node with no data-path collective:
// the instances share nothing, so a host with several devices creates ONE HANDLE PER DEVICE (mpc_create(cfg, max_batch, device, &h)), gives every handle its
// contiguous shard of the batch and drives the handles from one thread each -- a handle is thread-compatible exactly like the reference's Controller, which is not
// re-entrant (include/mpc_local_planner/controller.h:118-142).  This program does that with two handles: on device 0 and device 1 when the box has two devices,
// otherwise both on device 0 (two streams of one GPU), and checks that the two shards' answers are, bit for bit, what ONE handle returns for the whole batch
// (the solve of an instance does not depend on its neighbours: candidates, iteration counts and trajectories are functions of the instance's inputs alone).
//
//   g++ -O2 -std=c++17 -pthread examples/two_handles_shard.cpp -Iinclude -Lmpc_local_planner_amd/csrc -lmpc_hip -Wl,-rpath,$PWD/mpc_local_planner_amd/csrc -o two_handles_shard
//   ./two_handles_shard [B]          prints SHARD_OK on success
//
// No scaling curve has been measured with this repository (no multi-GPU node was available to the build sessions); bench.py --gpus N is the torch.distributed
// launcher of the same split, one process per GPU.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/mpc_hip.h"

namespace {
struct Batch {
    int B, n;
    std::vector<double> x0, xf, up, dtp, x, u, dt;
    std::vector<int32_t> st, it;
    Batch(int B_, int n_) : B(B_), n(n_), x0(3 * B_), xf(3 * B_), up(2 * B_), dtp(B_), x((size_t)B_ * n_ * 3), u((size_t)B_ * n_ * 2), dt(B_), st(B_), it(B_) {}
};
// car-like minimum-time instances (BASELINE configs[1] shape): start at the origin with a random heading, goal 1..6 m away
void fill(Batch& b, unsigned seed) {
    auto rnd = [&seed]() { seed = seed * 1664525u + 1013904223u; return (seed >> 8) * (1.0 / 16777216.0); };
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < b.B; ++i) {
        const double th0 = -pi + 2 * pi * rnd(), r = 1.0 + 5.0 * rnd(), bearing = -pi + 2 * pi * rnd(), yaw = -pi + 2 * pi * rnd();
        b.x0[3 * i] = 0; b.x0[3 * i + 1] = 0; b.x0[3 * i + 2] = th0;
        b.xf[3 * i] = r * std::cos(bearing); b.xf[3 * i + 1] = r * std::sin(bearing); b.xf[3 * i + 2] = yaw;
        b.up[2 * i] = -0.2 + 0.6 * rnd(); b.up[2 * i + 1] = -0.5 + rnd();
        b.dtp[i] = 0.2;
    }
}
int solve_range(mpc_solver* h, Batch& b, int lo, int hi) {
    const int n = b.n;
    return mpc_solve_batch(h, hi - lo, b.x0.data() + 3 * lo, b.xf.data() + 3 * lo, b.up.data() + 2 * lo, b.dtp.data() + lo, nullptr, nullptr, nullptr, nullptr,
                           b.x.data() + (size_t)lo * n * 3, b.u.data() + (size_t)lo * n * 2, b.dt.data() + lo, b.st.data() + lo, b.it.data() + lo);
}
}  // namespace

int main(int argc, char** argv) {
    const int B = argc > 1 ? std::atoi(argv[1]) : 512, n = 50;
    mpc_config cfg;
    mpc_config_defaults(&cfg);
    cfg.model = MPC_MODEL_SIMPLE_CAR; cfg.model_params[0] = 0.4;
    cfg.n = n; cfg.dt_ref = 0.3; cfg.dt_free = 1; cfg.objective = MPC_OBJ_MIN_TIME;
    cfg.u_lb[0] = -0.2; cfg.u_ub[0] = 0.4; cfg.u_lb[1] = -1.4; cfg.u_ub[1] = 1.4;
    cfg.du_lb[0] = cfg.du_lb[1] = -0.5; cfg.du_ub[0] = cfg.du_ub[1] = 0.5;
    cfg.n_candidates = 3;                                     // hedged candidates: the split must not change which candidate answers
    cfg.candidate_kind[0] = MPC_CAND_REFERENCE; cfg.candidate_kind[1] = MPC_CAND_HERMITE_FF; cfg.candidate_kind[2] = MPC_CAND_HERMITE_FR;
    cfg.candidate_max_iter[0] = 100; cfg.candidate_max_iter[1] = 45; cfg.candidate_max_iter[2] = 40;
    cfg.candidate_param[1] = 2.0; cfg.candidate_param[2] = 1.5;

    Batch whole(B, n), split(B, n);
    fill(whole, 20260925u);
    split.x0 = whole.x0; split.xf = whole.xf; split.up = whole.up; split.dtp = whole.dtp;

    // one handle, the whole batch
    mpc_solver* h = nullptr;
    if (mpc_create(&cfg, B, 0, &h) != MPC_OK) { std::printf("mpc_create: %s\n", mpc_last_error()); return 1; }
    if (solve_range(h, whole, 0, B) != MPC_OK) { std::printf("solve: %s\n", mpc_last_error()); return 1; }
    mpc_destroy(h);

    // two handles, one shard each, one host thread each; device 1 when there is one
    const int cut = B / 2 + 7;                                // ragged on purpose
    mpc_solver* ha = nullptr; mpc_solver* hb = nullptr;
    if (mpc_create(&cfg, cut, 0, &ha) != MPC_OK) { std::printf("mpc_create: %s\n", mpc_last_error()); return 1; }
    int dev_b = 1;
    if (mpc_create(&cfg, B - cut, dev_b, &hb) != MPC_OK) {    // MPC_ENODEV on a one-GPU box: the second handle shares device 0
        dev_b = 0;
        if (mpc_create(&cfg, B - cut, dev_b, &hb) != MPC_OK) { std::printf("mpc_create: %s\n", mpc_last_error()); return 1; }
    }
    int ra = -1, rb = -1;
    std::thread ta([&] { ra = solve_range(ha, split, 0, cut); });
    std::thread tb([&] { rb = solve_range(hb, split, cut, B); });
    ta.join(); tb.join();
    if (ra != MPC_OK || rb != MPC_OK) { std::printf("sharded solve failed: %d %d\n", ra, rb); return 1; }
    mpc_destroy(ha); mpc_destroy(hb);

    int conv = 0, diff = 0;
    for (int i = 0; i < B; ++i) {
        conv += whole.st[i] == MPC_CONVERGED;
        const bool same = whole.st[i] == split.st[i] && whole.it[i] == split.it[i] && whole.dt[i] == split.dt[i] &&
                          std::memcmp(&whole.x[(size_t)i * n * 3], &split.x[(size_t)i * n * 3], sizeof(double) * n * 3) == 0 &&
                          std::memcmp(&whole.u[(size_t)i * n * 2], &split.u[(size_t)i * n * 2], sizeof(double) * n * 2) == 0;
        diff += !same;
    }
    std::printf("B = %d, shards %d + %d on devices 0 and %d: %d converged, %d instances differ between the one-handle and the two-handle run\n", B, cut, B - cut, dev_b, conv, diff);
    if (diff != 0 || conv < (int)(0.95 * B)) return 1;
    std::printf("SHARD_OK\n");
    return 0;
}
