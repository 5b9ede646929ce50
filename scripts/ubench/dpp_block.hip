// Micro-benchmark: cost of the DPP blocks of the Riccati sweep in isolation (one wave per SIMD, 4 workgroups per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double T;
#include "../../mpc_local_planner_amd/csrc/mpc_dpp_blocks.inc"

template <int MODE> __global__ __launch_bounds__(64) void k(double* out, long long* ticks, int iters) {
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    T V[6], G[3], t[6], h[8], wn[3] = {0, 0, 0}, om = 0, ec = lane & 1, E3 = lane == 6, E4 = lane == 7, nK0 = 0.5, nK1 = 0.25, R00, R01, R11;
    for (int i = 0; i < 6; ++i) V[i] = out[lane] + i;
    for (int i = 0; i < 3; ++i) G[i] = 0.001 * (lane + i);
    for (int i = 0; i < 8; ++i) h[i] = 0.01 * i;
    for (int i = lane; i < 4096; i += 64) sm[i] = 1e-3 * i;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { MPC_DPP_BLOCK_T1 for (int i = 0; i < 6; ++i) V[i] = t[i] * 1e-3; }
        if (MODE == 1) { MPC_DPP_BLOCK_T1 for (int i = 0; i < 8; ++i) h[i] = t[i % 6] * 1e-3; MPC_DPP_BLOCK_H MPC_DPP_BLOCK_R nK0 = R00 * 1e-3; nK1 = R01 * R11 * 1e-3;
                         for (int i = 0; i < 6; ++i) V[i] = h[i]; MPC_DPP_BLOCK_V for (int i = 0; i < 6; ++i) V[i] *= 1e-3; }
        if (MODE == 2) { for (int i = 0; i < 6; ++i) t[i] = V[i]; MPC_DPP_BLOCK_H MPC_DPP_BLOCK_R nK0 = R00 * 1e-3; nK1 = R01 * R11 * 1e-3; for (int i = 0; i < 6; ++i) V[i] = h[i % 8] * 1e-3; }
        if (MODE == 3) { MPC_DPP_BLOCK_V for (int i = 0; i < 6; ++i) V[i] *= 1e-3; }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = V[0] + V[1] + V[2] + V[3] + V[4] + V[5] + wn[0] + wn[1] + wn[2] + om + h[2] + h[5];
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, int blocks) {
    double* out; long long* ticks;
    hipMalloc(&out, blocks * 64 * sizeof(double)); hipMalloc(&ticks, blocks * sizeof(long long));
    hipMemset(out, 0, blocks * 64 * sizeof(double));
    const int iters = 2000; const size_t lds = 39 * 1024;
    k<MODE><<<blocks, 64, lds>>>(out, ticks, 10);
    k<MODE><<<blocks, 64, lds>>>(out, ticks, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks); hipMemcpy(h.data(), ticks, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto t : h) mean += t; mean /= blocks;
    printf("%-40s blocks=%5d  ticks/iteration %8.1f\n", name, blocks, mean / iters);
}
int main() {
    for (int blocks : {1, 1024}) {
        run<0>("T1 block (39 dpp + 6 mul) + 6 mul", blocks);
        run<1>("T1+H+R+V blocks (~75 dpp) + ~25", blocks);
        run<2>("H+R blocks (14 dpp) + ~14", blocks);
        run<3>("V block (18 dpp) + 6 mul", blocks);
    }
}
