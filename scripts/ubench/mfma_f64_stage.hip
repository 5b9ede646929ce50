// Micro-benchmark (VERDICT r02 item 8 / "next" item 3): would the backward-sweep stage be cheaper on the fp64 matrix cores?
// north_star: "MFMA only on the dense per-stage Hessian tiles ... evidenced by MFMA utilisation".  The stage is
//     T1 = P+ [Ghat | c] (+ [0 | p+ S+]),   Hhat = Ghat' T1 + cost,   V = Hhat_xx - Hhat_xu R^-1 Hhat_ux        (6 x 12 value block, 8 x 12 Hhat)
// As 4 x 4 blocks (rows 6 -> 8, no other padding): T1 = 2 x 3 output blocks x 2 inner, Hhat = 2 x 3 x 2, V = 2 x 3 x 1 (the rank-2 update padded to 4):
// 30 block products; v_mfma_f64_4x4x4f64 does FOUR independent 4x4x4 block products per instruction (one per 16-lane group) -> 10 instructions with the
// two inner blocks of an output chained through SrcC (4 + 4 + 2).  Between the three dependent products the operands must change lane groups (an output block
// computed in group g is needed as the B operand of other groups) -- the D layout (lane = j + 4 i) IS the B layout (lane = j + 4 k), so a move is a pure
// lane permutation: 2 ds_bpermute_b32 per 64-bit register.  The 2 x 2 pivot R is broadcast with v_readlane and inverted in closed form as today.
//
// What is measured (one wave per SIMD, 4 workgroups per CU, 39 KB LDS each = the solve kernel's residency; `s_memtime`-free: clock64 ticks):
//   mode 0  64 dependent v_mfma_f64_4x4x4f64 (chained through SrcC)          -> latency of one
//   mode 1  64 independent ones (4 accumulators)                             -> issue cost of one
//   mode 2  16 dependent v_mfma_f64_16x16x4f64                               -> latency (the shape DESIGN.md argued about in r02)
//   mode 3  16 independent ones
//   mode 4  the MFMA stage: 6 operand loads from LDS (Ghat blocks in A / B layout, cost blocks as SrcC -- kkt_pass could store them that way), 10 MFMAs with
//           the dependency structure above, 2 + 2 cross-group operand moves (ds_bpermute pairs), R broadcast (6 v_readlane), closed-form 2 x 2 inverse,
//           gain scaling, 2 gain stores
//   mode 5  today's stage: the four generated DPP blocks (75 DPP FMAs) + ~25 plain instructions (scripts/ubench/dpp_block.hip mode 1), same harness
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_f64_stage.hip -o mfma_f64_stage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double T;
#include "../../mpc_local_planner_amd/csrc/mpc_dpp_blocks.inc"

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double4_t mfma16(double a, double b, double4_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double permute(double v, int byte_addr) {      // lane permutation of a 64-bit register: 2 ds_bpermute_b32
    int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rdl(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

template <int MODE> __global__ __launch_bounds__(64) void k(double* out, long long* ticks, int iters) {
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) sm[i] = 1e-3 * (i % 97) + 0.5;
    __syncthreads();
    double acc = out[lane];
    long long t0 = 0, t1 = 0;
    if (MODE == 0) {
        double a = 1.0 + 1e-3 * lane, b = 1e-3, c = acc;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 64; ++r) c = mfma4(a, b, c);
        }
        t1 = __builtin_readcyclecounter();
        acc = c;
    } else if (MODE == 1) {
        double a = 1.0 + 1e-3 * lane, b = 1e-3, c0 = acc, c1 = acc + 1, c2 = acc + 2, c3 = acc + 3;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0 = mfma4(a, b, c0); c1 = mfma4(a, b, c1); c2 = mfma4(a, b, c2); c3 = mfma4(a, b, c3); }
        }
        t1 = __builtin_readcyclecounter();
        acc = c0 + c1 + c2 + c3;
    } else if (MODE == 2) {
        double a = 1.0 + 1e-3 * lane, b = 1e-3;
        double4_t c = {acc, acc, acc, acc};
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c = mfma16(a, b, c);
        }
        t1 = __builtin_readcyclecounter();
        acc = c[0] + c[1] + c[2] + c[3];
    } else if (MODE == 3) {
        double a = 1.0 + 1e-3 * lane, b = 1e-3;
        double4_t c0 = {acc, acc, acc, acc}, c1 = c0, c2 = c0, c3 = c0;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { c0 = mfma16(a, b, c0); c1 = mfma16(a, b, c1); c2 = mfma16(a, b, c2); c3 = mfma16(a, b, c3); }
        }
        t1 = __builtin_readcyclecounter();
        acc = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (MODE == 4) {
        // value block of the next stage: X = blocks (00, 01, 02, 10) in the four lane groups, Y = blocks (11, 12, -, -)
        double X = acc, Y = acc + 1.0;
        const int grp = lane >> 4, in = lane & 15;
        const int mv_a = (((grp + 1) & 3) * 16 + in) * 4, mv_b = (((grp + 2) & 3) * 16 + in) * 4;     // byte addresses of ds_bpermute: take the operand from another group
        const double* rec = sm + lane;        // per-lane operand words of the stage record (operand layout), stride 64 words per operand register
        double* gain = sm + 2048 + lane;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
            // operands of this stage (prefetched one stage ahead in a real sweep; here: the loads are simply issued first)
            const double* rk = rec + (it & 3) * 384;       // the stage record moves with the stage: the loads cannot be hoisted out of the loop
            const double gA0 = rk[0], gA1 = rk[64], gB0 = rk[128], gB1 = rk[192], cA = rk[256], cB = rk[320];
            // T1 = P+ Ghat_ext: outputs (00, 01, 02, 10) = inner 0 then inner 1 chained; outputs (11, 12) the same
            const double Xm = permute(X, mv_a), Ym = permute(Y, mv_b);       // P+ blocks brought to the groups that need them as A operands
            double t1x = mfma4(X, gB0, 0.0);
            t1x = mfma4(Xm, gB1, t1x);
            double t1y = mfma4(Y, gB0, 0.0);
            t1y = mfma4(Ym, gB1, t1y);
            // Hhat = Ghat' T1 + cost: T1 is already in B layout; blocks move between groups
            const double t1xm = permute(t1x, mv_a), t1ym = permute(t1y, mv_b);
            double hx = mfma4(gA0, t1x, cA);
            hx = mfma4(gA1, t1xm, hx);
            double hy = mfma4(gA0, t1y, cB);
            hy = mfma4(gA1, t1ym, hy);
            // pivot: R = Hhat[6:8][6:8] sits in one group of hy; broadcast, closed-form inverse (as backward_dpp does)
            const double R00 = rdl(hy, 10), R01 = rdl(hy, 11), R11 = rdl(hy, 15);
            const double det = R00 * R11 - R01 * R01;
            double r = __builtin_amdgcn_rcp(det);
            r = __builtin_fma(__builtin_fma(-det, r, 1.0), r, r);
            r = __builtin_fma(__builtin_fma(-det, r, 1.0), r, r);
            const double nid = -r, i00 = R11 * nid, i01 = -(R01 * nid), i11 = R00 * nid;
            // negated gains nK = -R^-1 Hhat_u: (2 x 12), one value per lane in the u-row groups; written to LDS for the forward pass
            const double hu0 = permute(hy, mv_a);
            const double nK = (in < 4 ? i00 : i01) * hy + (in < 4 ? i01 : i11) * hu0;
            gain[0] = nK; gain[64] = hu0 * nid;
            // V = Hhat_xx + Hhat_xu nK (rank 2, padded to one 4-deep block product per output block): 6 output blocks = 2 instructions
            X = mfma4(hx, nK, hx) * 1e-3;
            Y = mfma4(hy, nK, hy) * 1e-3;
        }
        t1 = __builtin_readcyclecounter();
        acc = X + Y;
    } else {
        T V[6], G[3], t[6], h[8], wn[3] = {0, 0, 0}, om = 0, ec = lane & 1, E3 = lane == 6, E4 = lane == 7, nK0 = 0.5, nK1 = 0.25, R00, R01, R11;
        for (int i = 0; i < 6; ++i) V[i] = acc + i;
        for (int i = 0; i < 3; ++i) G[i] = 0.001 * (lane + i);
        for (int i = 0; i < 8; ++i) h[i] = 0.01 * i;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
            MPC_DPP_BLOCK_T1 for (int i = 0; i < 8; ++i) h[i] = t[i % 6] * 1e-3; MPC_DPP_BLOCK_H MPC_DPP_BLOCK_R nK0 = R00 * 1e-3; nK1 = R01 * R11 * 1e-3;
            for (int i = 0; i < 6; ++i) V[i] = h[i]; MPC_DPP_BLOCK_V for (int i = 0; i < 6; ++i) V[i] *= 1e-3;
        }
        t1 = __builtin_readcyclecounter();
        acc = V[0] + V[1] + V[2] + V[3] + V[4] + V[5] + wn[0] + wn[1] + wn[2] + om + h[2] + h[5];
    }
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, int blocks, int per_iter, size_t lds = 39 * 1024, int waves_per_simd = 1) {
    double* out; long long* ticks;
    hipMalloc(&out, blocks * 64 * sizeof(double)); hipMalloc(&ticks, blocks * sizeof(long long));
    hipMemset(out, 0, blocks * 64 * sizeof(double));
    const int iters = 2000;
    k<MODE><<<blocks, 64, lds>>>(out, ticks, 10);
    k<MODE><<<blocks, 64, lds>>>(out, ticks, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks); hipMemcpy(h.data(), ticks, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto t : h) mean += t; mean /= blocks;
    if (waves_per_simd > 1) printf("%-78s blocks=%5d  %d waves per SIMD: ticks per stage of ONE wave %8.1f = %8.1f per stage and SIMD\n", name, blocks, waves_per_simd, mean / iters / per_iter, mean / iters / per_iter / waves_per_simd);
    else printf("%-78s blocks=%5d  ticks per %s %8.1f\n", name, blocks, per_iter == 1 ? "stage      " : "instruction", mean / iters / per_iter);
    hipFree(out); hipFree(ticks);
}
int main() {
    for (int blocks : {1, 1024}) {
        run<0>("v_mfma_f64_4x4x4f64, dependent chain (SrcC = previous result)", blocks, 64);
        run<1>("v_mfma_f64_4x4x4f64, 4 independent accumulators", blocks, 64);
        run<2>("v_mfma_f64_16x16x4f64, dependent chain", blocks, 16);
        run<3>("v_mfma_f64_16x16x4f64, 4 independent accumulators", blocks, 16);
        run<4>("backward-sweep stage on v_mfma_f64_4x4x4f64 (10 MFMAs + 5 lane permutations + pivot + gains)", blocks, 1);
        run<5>("backward-sweep stage as shipped (T1+H+R+V DPP blocks, 75 DPP FMAs + ~25)", blocks, 1);
    }
    // r06: the same two stages with TWO waves per SIMD -- 16 KB of LDS per one-wave workgroup, 2048 workgroups = eight per CU -- so that the second wave can
    // hide the ds_bpermute round trips and the MFMA -> VALU hand-offs of the first.  A wave's own clock then runs longer per stage; what counts is ticks per stage and SIMD.
    run<4>("backward-sweep stage on v_mfma_f64_4x4x4f64 (10 MFMAs + 5 lane permutations + pivot + gains)", 2048, 1, 16 * 1024, 2);
    run<5>("backward-sweep stage as shipped (T1+H+R+V DPP blocks, 75 DPP FMAs + ~25)", 2048, 1, 16 * 1024, 2);
    run<1>("v_mfma_f64_4x4x4f64, 4 independent accumulators", 2048, 64, 16 * 1024, 2);
}
