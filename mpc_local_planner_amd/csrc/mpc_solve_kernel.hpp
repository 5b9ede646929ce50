// mpc_solve_kernel.hpp -- the solve kernel (one wavefront = one planner instance x one candidate initial trajectory) and its launcher.
// Instantiated per (arithmetic type, model) x three levels (IpmWave's EXT parameter): in the split build every (type, model) pair is its own
// object file (mpc_solve_inst.hip, compiled in parallel by mpc_local_planner_amd/_lib.py); without -DMPC_SPLIT_BUILD everything is
// instantiated in the translation unit that includes this header (plain `hipcc mpc_capi.hip`, developer builds).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "../../include/mpc_hip.h"
#include "mpc_core.hpp"
#include "mpc_wave.hpp"

namespace mpc {

// Candidate bookkeeping of a launch with n_candidates > 1 (device pointers; all NULL / 0 for a single candidate).
//   win[b]      lowest candidate index of instance b that has converged so far (INT_MAX-like: none)
//   exited[b]   candidates of instance b that have finished; the LAST one to finish copies the winner's record to the caller's outputs
//   it_sum[b]   iterations spent on instance b by all its candidates
//   rec         [C][B][5 n + 3 (+ multipliers)] doubles: x (n x 3), u (n x 2), dt, status, iterations of hedge c >= 1 of instance b, written when it
//               converged (candidate 0 delivers straight into the caller's arrays)
// win / exited / it_sum are restored to their idle values by that last workgroup, so consecutive launches need no memset.
struct CandCtl {
    int n_cand;
    int* win;
    int* exited;
    int* it_sum;
    double* rec;
    int32_t* winner_out;
    int32_t* iters_total_out;
    int32_t* rows_dropped;     // [B] clearance rows of candidate 0 that did not fit into max_obstacle_rows (NULL without obstacles)
    double* dual;              // [B][dual_words] multipliers kept between control cycles (dual_warm_start) or NULL; word 0 = grid size, 0 = nothing kept
    int dual_words;            // doubles per instance in `dual` (and appended to every candidate record)
    int dual_read;             // this launch STARTS from the kept multipliers (dual_warm_start, or the fp64 phase of MPC_MIXED); 0: it only leaves them
};
constexpr int kWinIdle = 0x7f7f7f7f;

// One wavefront = one (planner instance, candidate initial trajectory); the whole working set lives in LDS (mpc_wave.hpp).
// Grid: n_cand * B workgroups, candidate-major, so that the hardware dispatches every instance's candidate 0 before any hedge.
// W2: the variant for TWO waves per SIMD (throughput regime: more instances than SIMDs): at most 256 registers, every phase of an iteration on a lane index of its own
// (IpmWave::local_lane: no per-lane address arithmetic survives from one phase into the next), line-search trials on the generic path.  Same arithmetic, same results bit for bit.
template <typename T, int MODEL, int EXT, bool OBST, int NSC = 0, bool GS = false, bool W2 = false>
__global__ __launch_bounds__(mpc::kWave)
__attribute__((amdgpu_waves_per_eu(W2 ? 2 : 1)))
void mpc_ipm_wave_kernel(
    mpc::Problem<T> P, mpc::WaveLayout L, int B,
    const double* __restrict__ x0, const double* __restrict__ xf, const double* __restrict__ u_prev,
    const double* __restrict__ dt_prev, const double* __restrict__ x_init, const double* __restrict__ u_init,
    const double* __restrict__ dt_init, mpc_obstacles obst, const int32_t* __restrict__ n_grid, const int32_t* __restrict__ n_via,
    const double* __restrict__ via, CandCtl cc, const int32_t* __restrict__ iters_add, double* __restrict__ x_out,
    double* __restrict__ u_out, double* __restrict__ dt_out, int32_t* __restrict__ status, int32_t* __restrict__ iters, void* gstage, int* __restrict__ gslots, int n_gslots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mpc_smem[];
    T* sm = reinterpret_cast<T*>(mpc_smem);
    // problem record at the end of the dynamic LDS block (16-byte aligned); the layout stays in scalar registers
    const size_t coff = (((size_t)L.total * sizeof(T)) + 15) & ~(size_t)15;
    mpc::Problem<T>* Ps = reinterpret_cast<mpc::Problem<T>*>(mpc_smem + coff);
    const int NC = cc.n_cand;                        // wave-uniform kernel argument
    const int cand = NC > 1 ? (int)blockIdx.x / B : 0;
    const int inst = (int)blockIdx.x - cand * B;
    const int lane = threadIdx.x;
    if (inst >= B || cand >= (NC > 1 ? NC : 1)) return;
    const int nmax = L.n;          // stride of the instance-major arrays
    int n = nmax;                  // grid points of THIS instance (grid adaptation: n_i <= n_max)
    if (n_grid) { n = n_grid[inst]; n = n < 3 ? 3 : (n > nmax ? nmax : n); }
    // a hedge whose instance already has a converged higher-priority candidate never starts
    bool run = true;
    if (NC > 1 && cand > 0) {
        const int w = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cc.win + inst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));     // a hint only: decides how much of a LOSING candidate runs, never the result
        run = !(w < cand);
    }
    int st_status = mpc::ST_SUPERSEDED, st_iters = 0;
    if (run) {
        mpc::WaveLayout Lv = L;            // from the kernel arguments: wave-uniform, lives in SGPRs
        Lv.n = __builtin_amdgcn_readfirstlane(n);
#ifdef MPC_POISON_LDS      // developer check: any read of an LDS word the solver did not write first turns into NaN
        for (int e = lane; e < L.total; e += mpc::kWave) sm[e] = T(NAN);
        __syncthreads();
#endif
        if (lane == 0) { *Ps = P; Ps->n = n; }
        __syncthreads();
        mpc::IpmWave<T, MODEL, EXT, OBST, NSC, GS, W2> S(*Ps, Lv, sm, lane);
        int gslot = -1;
        if (GS || (OBST && L.GSW > 0)) {
            // this workgroup's block of factorisation data (GlobalStage): one of the n_gslots blocks OF ITS XCD, claimed for the lifetime of the workgroup.  Per XCD because
            // the eight L2s are not coherent with each other inside a launch: a block that only ever moves through ONE L2 needs no cache maintenance, a block that changed
            // XCDs could be clobbered by the write-back of the previous owner's dirty lines.  There are more blocks per XCD than workgroups it can hold at once
            // (mpc_capi.hip), so the probe ends after a step or two.  What it buys over a block per workgroup of the GRID: the memory a launch touches is resident waves
            // x 63 n words (62 MB at n = 120 in fp64 on 256 CUs) whatever the batch -- inside the 256 MB Infinity Cache -- instead of batch x candidates x 63 n words
            // (2 GB at 8192 x 4).
            if (lane == 0) {
                const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;          // HW_REG_XCC_ID (id 20), bits 3:0
                const unsigned base = xcc * (unsigned)n_gslots;
                unsigned g = ((unsigned)blockIdx.x * 2654435761u) % (unsigned)n_gslots;
                while (atomicCAS(gslots + base + g, 0, 1) != 0) g = g + 1 == (unsigned)n_gslots ? 0u : g + 1;
                gslot = (int)(base + g);
            }
            gslot = __builtin_amdgcn_readfirstlane(gslot);
            S.gmb = reinterpret_cast<T*>(gstage) + (size_t)gslot * (size_t)L.GSW;
        }
        for (int i = 0; i < 3; ++i) { S.x0[i] = T(x0[3 * inst + i]); S.xf[i] = T(xf[3 * inst + i]); }
        S.x0[2] = mpc::normalize_theta(S.x0[2]);
        S.xf[2] = mpc::normalize_theta(S.xf[2]);
        S.uprev[0] = u_prev ? T(u_prev[2 * inst]) : T(0);
        S.uprev[1] = u_prev ? T(u_prev[2 * inst + 1]) : T(0);
        S.dtprev = dt_prev ? T(dt_prev[inst]) : T(0);
        // (indexed through the LDS copy: a run-time index into the by-value kernel argument would put the arrays into scratch memory)
        const int kind = NC > 1 ? Ps->cand_kind[cand] : 0;
        if (NC > 1) { S.my_cand = cand; S.iter_cap = Ps->cand_max_iter[cand]; S.win_ptr = cand > 0 ? cc.win + inst : nullptr; }
        if (cc.dual && cc.dual_read && cand == 0) S.dual_in = cc.dual + (long)inst * cc.dual_words;
        if (kind == 0 && x_init && u_init && dt_init) {
            // coalesced read of this instance's contiguous [n][3] / [n][2] blocks
            const double* xi = x_init + (long)inst * nmax * 3;
            const double* ui = u_init + (long)inst * nmax * 2;
            for (int e = lane; e < 3 * n; e += mpc::kWave) S.F(L.X, e % 3, e / 3) = T(xi[e]);
            for (int e = lane; e < 2 * (n - 1); e += mpc::kWave) S.F(L.U, e % 2, e / 2) = T(ui[e]);
            if (lane == 0) S.SCL(mpc::SC_D) = T(dt_init[inst]);
            S.warm_guess = true;
        } else if (kind == 0) {
            S.cold_start();
        } else {
            S.seed_start(kind, Ps->cand_param[cand]);
        }
        if (OBST && L.M > 0) S.load_obstacles(obst.n_obstacles, obst.n_vertices, obst.vertices, obst.radius, obst.velocity, inst);
        if (EXT && L.NV > 0) S.load_via_points(n_via, via, inst);
        __syncthreads();
        mpc::SolveStats<T> st = S.solve();
        __syncthreads();
        if (GS || (OBST && L.GSW > 0)) { if (lane == 0) __hip_atomic_store(gslots + gslot, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }      // (every access of the block happened before the barrier)
        st_status = st.status; st_iters = st.iters;
        if (cc.rows_dropped && cand == 0 && lane == 0) cc.rows_dropped[inst] = S.rows_dropped;
        if (cand == 0) {
            // candidate 0 (the only one when NC <= 1) delivers straight into the caller's arrays: whenever it converges it IS the result (lowest
            // index), and when no candidate converges its last iterate and status are what is returned.  Only when a hedge wins does the last
            // workgroup of the instance overwrite this with the hedge's record (it runs after every candidate, this one included, has left).
            double* xo = x_out + (long)inst * nmax * 3;
            double* uo = u_out + (long)inst * nmax * 2;
            for (int e = lane; e < 3 * nmax; e += mpc::kWave) { int k = e / 3; int ks = k < n ? k : n - 1; xo[e] = double(S.F(L.X, e % 3, ks)); }
            for (int e = lane; e < 2 * nmax; e += mpc::kWave) { int k = e / 2; int ks = k < n - 1 ? k : n - 2; uo[e] = double(S.F(L.U, e % 2, ks)); }
            if (lane == 0) {
                dt_out[inst] = double(S.SCL(mpc::SC_D));
                if (status) status[inst] = st.status;
                if (iters) iters[inst] = st.iters + (iters_add ? iters_add[inst] : 0);
            }
            if (cc.dual) {
                double* blk = cc.dual + (long)inst * cc.dual_words;
                __syncthreads();                                   // every lane has read its share of the old block (load_duals) long ago; keep the order explicit
                if (st.status == mpc::ST_CONVERGED) S.store_duals(blk);
                else if (lane == 0) blk[0] = 0.0;
            }
            if (NC <= 1) return;
        } else if (st.status == mpc::ST_CONVERGED) {
            // a hedge that converged leaves its record: x (n x 3), u (n x 2), dt, status, iterations [, multipliers]
            double* r = cc.rec + ((long)cand * B + inst) * (5 * nmax + 3 + cc.dual_words);
            for (int e = lane; e < 3 * nmax; e += mpc::kWave) { int k = e / 3; int ks = k < n ? k : n - 1; r[e] = double(S.F(L.X, e % 3, ks)); }
            for (int e = lane; e < 2 * nmax; e += mpc::kWave) { int k = e / 2; int ks = k < n - 1 ? k : n - 2; r[3 * nmax + e] = double(S.F(L.U, e % 2, ks)); }
            if (lane == 0) { r[5 * nmax] = double(S.SCL(mpc::SC_D)); r[5 * nmax + 1] = double(st.status); r[5 * nmax + 2] = double(st.iters); }
            if (cc.dual) S.store_duals(r + 5 * nmax + 3);
        }
    }
    // ---- exit protocol (n_cand > 1): publish, count, and let the last candidate of the instance deliver a hedge's result.
    // Hand-off of a hedge's record (and of candidate 0's outputs) from the workgroup that wrote it to the workgroup that leaves last, possibly on
    // another XCD (own L2): RELEASE by the writer -- every lane's agent-scope release fence after its own plain stores, the workgroup barrier,
    // then lane 0's release-ordered atomics on `win` / `exited` -- and ACQUIRE by the reader -- lane 0's acquire-ordered read-modify-write of
    // `exited` (it reads the value the last writer released), the broadcast of its result, every lane's agent-scope acquire fence, an
    // acquire-ordered load of `win`, then ordinary loads of the record.  No relaxed load decides what is read, no non-temporal load reads it.
    __threadfence();                                // agent-scope fence (release side), executed by EVERY lane after its own stores
    __syncthreads();
    int last = 0;
    if (lane == 0) {
        if (st_status == mpc::ST_CONVERGED) __hip_atomic_fetch_min(cc.win + inst, cand, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (st_iters > 0) __hip_atomic_fetch_add(cc.it_sum + inst, st_iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = __hip_atomic_fetch_add(cc.exited + inst, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == NC - 1;
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    __threadfence();                                // agent-scope fence (acquire side), every lane, before any lane reads the record
    const int w = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cc.win + inst, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT));
    if (w > 0 && w < NC) {
        const double* r = cc.rec + ((long)w * B + inst) * (5 * nmax + 3 + cc.dual_words);
        double* xo = x_out + (long)inst * nmax * 3;
        double* uo = u_out + (long)inst * nmax * 2;
        for (int e = lane; e < 3 * nmax; e += mpc::kWave) xo[e] = r[e];
        for (int e = lane; e < 2 * nmax; e += mpc::kWave) uo[e] = r[3 * nmax + e];
        if (cc.dual) {
            double* blk = cc.dual + (long)inst * cc.dual_words;
            for (int e = lane; e < cc.dual_words; e += mpc::kWave) blk[e] = r[5 * nmax + 3 + e];
        }
        if (lane == 0) {
            dt_out[inst] = r[5 * nmax];
            if (status) status[inst] = (int32_t)r[5 * nmax + 1];
            if (iters) iters[inst] = (int32_t)r[5 * nmax + 2] + (iters_add ? iters_add[inst] : 0);
        }
    }
    if (lane == 0) {
        if (cc.winner_out) cc.winner_out[inst] = w < NC ? w : -1;
        if (cc.iters_total_out) cc.iters_total_out[inst] = __hip_atomic_load(cc.it_sum + inst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // back to idle for the next launch
        __hip_atomic_store(cc.win + inst, kWinIdle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cc.it_sum + inst, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cc.exited + inst, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


// everything one launch needs besides the problem record (plain pointers: device memory of the handle / of the caller)
struct SolveLaunch {
    int level;                  // 0 headline kernel, 1 + rare rows / terms / coupling slots, 2 + cost variants
    size_t lds;                 // dynamic LDS of one workgroup
    hipStream_t stream;
    WaveLayout L;
    int B;
    const double *x0, *xf, *u_prev, *dt_prev, *x_init, *u_init, *dt_init;
    mpc_obstacles obst;
    const int32_t *n_grid, *n_via;
    const double* via;
    CandCtl cc;
    const int32_t* iters_add;
    double *x_out, *u_out, *dt_out;
    int32_t *status, *iters;
    void* gstage;               // L.GSW > 0: 8 x n_gslots blocks of factorisation data in global memory (L.GSW words of T each; n_gslots per XCD), claimed by the workgroups through `gslots`; else NULL
    int* gslots;                // [8][n_gslots] 0 = free, 1 = taken (all 0 between launches)
    int n_gslots;
    bool w2;                    // the two-waves-per-SIMD kernel (fp64, level 0, no clearance rows, L.GSF: mpc_capi.hip decides per launch)
};

constexpr int kFixedLayoutNS = 50;

// the kernel instantiation that serves a launch record (nullptr + an error for a combination that does not exist)
template <typename T, int MODEL>
auto select_kernel(const SolveLaunch& a, hipError_t& err) -> decltype(&mpc_ipm_wave_kernel<T, MODEL, 0, true>) {
    err = hipSuccess;
    // four instantiations per (arithmetic type, model): the headline level without / with clearance rows, and the two extended levels (always with)
#ifdef MPC_DEV_SWITCHES
    static const bool force_obst = getenv("MPC_FORCE_OBST_KERNEL") != nullptr;      // developer switch (A/B of the two headline instantiations); not in the shipped library
#else
    constexpr bool force_obst = false;
#endif
    auto kern = a.level == 0 ? ((a.L.M > 0 || force_obst) ? mpc_ipm_wave_kernel<T, MODEL, 0, true> : mpc_ipm_wave_kernel<T, MODEL, 0, false>)
                             : (a.level == 2 ? mpc_ipm_wave_kernel<T, MODEL, 2, true> : mpc_ipm_wave_kernel<T, MODEL, 1, true>);
    // factorisation data in global memory (WaveLayout::GSF; mpc_capi.hip decides per handle and precision)
    if (a.L.GSF) {      // (r06: the extended levels too -- turning footprints, moving obstacles, cost variants at grid sizes whose LDS record fits fewer than four times)
        kern = a.level == 0 ? ((a.L.M > 0 || force_obst) ? mpc_ipm_wave_kernel<T, MODEL, 0, true, 0, true> : mpc_ipm_wave_kernel<T, MODEL, 0, false, 0, true>)
                            : (a.level == 2 ? mpc_ipm_wave_kernel<T, MODEL, 2, true, 0, true> : mpc_ipm_wave_kernel<T, MODEL, 1, true, 0, true>);
    }
    // fp64 headline kernel on a grid of kFixedLayoutNS points per record (the grid size of BASELINE configs[1] / [3]): the instantiation whose LDS layout is a
    // compile-time constant (mpc_wave.hpp::FixedLayout) -- same code, same results bit for bit, ~3 % fewer instructions; every other size runs the generic one
    if constexpr (sizeof(T) == 8) {
#ifdef MPC_DEV_SWITCHES
        static const bool no_fixed = getenv("MPC_NO_FIXED_LAYOUT") != nullptr;      // developer switch (A/B); not in the shipped library
#else
        constexpr bool no_fixed = false;
#endif
        using IW = IpmWave<T, MODEL, 0, false, kFixedLayoutNS>;
        if (a.level == 0 && a.L.M == 0 && a.L.GSF == 0 && !force_obst && !no_fixed && IW::LayoutT::matches(a.L)) kern = mpc_ipm_wave_kernel<T, MODEL, 0, false, kFixedLayoutNS>;
    }
    if constexpr (sizeof(T) == 8) {
        if (a.w2) {
            if (a.level != 0 || a.L.M > 0) { err = hipErrorInvalidConfiguration; return nullptr; }
#ifdef MPC_DEV_SWITCHES
            kern = a.L.GSF ? mpc_ipm_wave_kernel<T, MODEL, 0, false, 0, true, true> : mpc_ipm_wave_kernel<T, MODEL, 0, false, 0, false, true>;
            {      // developer A/B: the 256-register variant of the fixed-layout kernel (one wave per SIMD all the same: its record is 40 KB)
                using IW = IpmWave<T, MODEL, 0, false, kFixedLayoutNS>;
                if (!a.L.GSF && IW::LayoutT::matches(a.L)) kern = mpc_ipm_wave_kernel<T, MODEL, 0, false, kFixedLayoutNS, false, true>;
            }
#else
            if (a.L.GSF) { err = hipErrorInvalidConfiguration; return nullptr; }
            kern = mpc_ipm_wave_kernel<T, MODEL, 0, false, 0, false, true>;
#endif
        }
    }
    return kern;
}

template <typename T, int MODEL>
hipError_t launch_solve(const SolveLaunch& a, const Problem<T>& P) {
    if (a.L.GSW > 0 && !a.gstage) return hipErrorInvalidConfiguration;
    hipError_t err;
    auto kern = select_kernel<T, MODEL>(a, err);
    if (!kern) return err;
    if (a.lds > 48u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)a.B * (unsigned)(P.n_cand > 1 ? P.n_cand : 1)), dim3(kWave), a.lds, a.stream, P, a.L, a.B, a.x0, a.xf, a.u_prev, a.dt_prev,
                       a.x_init, a.u_init, a.dt_init, a.obst, a.n_grid, a.n_via, a.via, a.cc, a.iters_add, a.x_out, a.u_out, a.dt_out, a.status, a.iters, a.gstage, a.gslots, a.n_gslots);
    return hipSuccess;
}

// resident one-wave workgroups per CU of the kernel that serves a launch record (registers, LDS): what sizes the per-XCD block pools (mpc_capi.hip)
template <typename T, int MODEL>
hipError_t solve_occupancy(const SolveLaunch& a, int* out) {
    hipError_t err;
    auto kern = select_kernel<T, MODEL>(a, err);
    if (!kern) return err;
    if (a.lds > 48u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, kern, kWave, a.lds);
}

#if defined(MPC_SPLIT_BUILD) && !defined(MPC_SOLVE_INST)
#define MPC_EXTERN_LAUNCH(T, M) extern template hipError_t launch_solve<T, M>(const SolveLaunch&, const Problem<T>&); extern template hipError_t solve_occupancy<T, M>(const SolveLaunch&, int*);
MPC_EXTERN_LAUNCH(double, 0) MPC_EXTERN_LAUNCH(double, 1) MPC_EXTERN_LAUNCH(double, 2) MPC_EXTERN_LAUNCH(double, 3)
MPC_EXTERN_LAUNCH(float, 0) MPC_EXTERN_LAUNCH(float, 1) MPC_EXTERN_LAUNCH(float, 2) MPC_EXTERN_LAUNCH(float, 3)
#undef MPC_EXTERN_LAUNCH
#endif

}  // namespace mpc
