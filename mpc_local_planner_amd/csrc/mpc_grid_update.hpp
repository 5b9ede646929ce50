// mpc_grid_update.hpp -- the grid update between two control cycles for a whole batch, on the device (no host round trip in a batched
// closed loop).  Restates, per instance and in place on the solver's own output layout (x [n][3], u [n][2] with the duplicated last
// control, dt, grid size):
//   fixed grid    warmStartShifting + findNearestState (src/optimal_control/full_discretization_grid_base_se2.cpp:241-339): shift by the
//                 index of the state nearest to the new start (search stops at the first non-improving sample, look-ahead <= 20), linear
//                 extrapolation of the tail (angles with interpolate_angle(.., 2.0)), last control held.  The multipliers the handle keeps
//                 for the slot (dual_warm_start) are shifted with the same index, the tail repeats the last stage.
//   variable grid adaptGridTimeBasedSingleStep (src/optimal_control/finite_differences_variable_grid_se2.cpp:99-121) + resampleTrajectory
//                 (...grid_base_se2.cpp:440-524): n + 1 when dt > dt_ref (1 + hyst) and n < n_max, n - 1 when dt < dt_ref (1 - hyst) and
//                 n > n_min; theta-aware linear re-interpolation, controls held, dt rescaled to keep the horizon length; no shifting
//                 (finite_differences_variable_grid_se2.h:85).  The slot's kept multipliers are dropped when n changes.
// One 64-lane workgroup per instance; the instance's trajectory (5 n doubles) goes through LDS.  All double arithmetic un-fused so that the
// result is bit-identical to the host restatements (include/mpc_controller.hpp, oracle/se2_nlp.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpc {

struct GridUpdateArgs {
    const double* x0;      // [B][3] new start states
    double* x;             // [B][n_stride][3]
    double* u;             // [B][n_stride][2]
    double* dt;            // [B]
    int32_t* n_grid;       // [B] grid sizes (in/out) or NULL (= n_stride everywhere, fixed)
    int32_t n_stride;
    int32_t mode;          // 0: fixed grid (shift), 1: variable grid (adapt + resample)
    int32_t n_min, n_max;
    double dt_ref, hyst;
    double* dual;          // [B][dual_words] kept multipliers or NULL
    int32_t dual_words, dual_ns;
};

__device__ __forceinline__ double gu_normalize_theta(double th) {
#pragma clang fp contract(off)
    const double pi = 3.14159265358979323846;
    if (th >= -pi && th < pi) return th;
    double m = ::floor(th / (2.0 * pi));
    th = th - m * 2.0 * pi;
    if (th >= pi) th -= 2.0 * pi;
    if (th < -pi) th += 2.0 * pi;
    return th;
}

__global__ __launch_bounds__(64) void grid_update_kernel(GridUpdateArgs a) {
#pragma clang fp contract(off)
    extern __shared__ double gsm[];          // xo [n][3] | uo [n][2]
    const int b = blockIdx.x, lane = threadIdx.x, ns = a.n_stride;
    int n = a.n_grid ? a.n_grid[b] : ns;
    n = n < 3 ? 3 : (n > ns ? ns : n);
    double* x = a.x + (size_t)b * ns * 3;
    double* u = a.u + (size_t)b * ns * 2;
    double* xo = gsm;
    double* uo = gsm + 3 * ns;
    for (int e = lane; e < 3 * n; e += 64) xo[e] = x[e];
    for (int e = lane; e < 2 * n; e += 64) uo[e] = u[e];
    __syncthreads();
    if (a.mode == 0) {
        // ---- findNearestState (serial, <= 20 steps), wave-uniform result
        __shared__ int s_ns;
        if (lane == 0) {
            const double* p = a.x0 + 3 * b;
            auto dist = [&](int i) { const double d0 = p[0] - xo[3 * i], d1 = p[1] - xo[3 * i + 1], d2 = p[2] - xo[3 * i + 2];
                                     const double q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2; const double s01 = q0 + q1; return __builtin_sqrt(s01 + q2); };
            const double first = dist(0);
            int best = 0;
            if (!(__builtin_fabs(first) < 1e-12)) {
                const int look = (n - 2) < 20 ? (n - 2) : 20;
                double cache = first;
                for (int i = 1; i <= look; ++i) { const double d = dist(i); if (d < cache) { cache = d; best = i; } else break; }
            }
            s_ns = best;
        }
        __syncthreads();
        const int sh = s_ns;
        if (sh <= 0 || sh > n - 2) return;
        // shifted part (parallel): rows 0 .. n-sh-1 take rows sh .. n-1
        for (int i = lane; i < n - sh; i += 64) {
            const int idx = i + sh;
            for (int c = 0; c < 3; ++c) x[3 * i + c] = xo[3 * idx + c];
            if (idx != n - 1) { u[2 * i] = uo[2 * idx]; u[2 * i + 1] = uo[2 * idx + 1]; }
        }
        __syncthreads();
        // extrapolated tail (serial recurrence over <= 20 rows)
        if (lane == 0) {
            int idx = n - sh;
            for (int i = 0; i < sh; ++i, ++idx) {
                for (int c = 0; c < 2; ++c) { const double d = x[3 * (idx - 1) + c] - x[3 * (idx - 2) + c]; const double t = 2.0 * d; x[3 * idx + c] = x[3 * (idx - 2) + c] + t; }
                const double a1 = x[3 * (idx - 2) + 2], a2 = x[3 * (idx - 1) + 2];
                const double w = gu_normalize_theta(a2 - a1); const double t = 2.0 * w;
                x[3 * idx + 2] = gu_normalize_theta(a1 + t);
                u[2 * (idx - 1)] = u[2 * (idx - 2)]; u[2 * (idx - 1) + 1] = u[2 * (idx - 2) + 1];
            }
            u[2 * (n - 1)] = u[2 * (n - 2)]; u[2 * (n - 1) + 1] = u[2 * (n - 2) + 1];
        }
        // kept multipliers: stage-indexed arrays move with the trajectory, the tail repeats the last stage
        if (a.dual) {
            double* blk = a.dual + (size_t)b * a.dual_words;
            if ((int)blk[0] == n) {
                const int NS = a.dual_ns;
                __syncthreads();
                // each array goes through the LDS block (the trajectory copy in it is no longer needed): any grid size the block holds, no
                // per-lane staging buffer (a fixed `keep[4]` overflowed for n > 256, which an fp32 solver's LDS budget allows)
                for (int comp = 0; comp < 11; ++comp) {
                    double* arr = blk + 4 + comp * NS;
                    for (int k = lane; k < n; k += 64) gsm[k] = arr[k];
                    __syncthreads();
                    const bool per_interval = comp < 3 || comp >= 7;        // lam, pl, pu live on the n - 1 intervals: the tail repeats interval n - 2
                    for (int k = lane; k < n; k += 64) { const int src = k + sh < n ? k + sh : n - 1; arr[k] = gsm[per_interval ? (src < n - 1 ? src : n - 2) : src]; }
                    __syncthreads();
                }
            }
        }
        return;
    }
    // ---- variable grid: adaptGridTimeBasedSingleStep + resampleTrajectory
    const double dt_old = a.dt[b];
    int n_new = n;
    if (dt_old > a.dt_ref * (1.0 + a.hyst) && n < a.n_max) n_new = n + 1;
    else if (dt_old < a.dt_ref * (1.0 - a.hyst) && n > a.n_min) n_new = n - 1;
    if (n_new > ns) n_new = ns;
    if (n_new == n) return;
    const double dt_new = dt_old * (double)(n - 1) / (double)(n_new - 1);
    for (int idx_new = 1 + lane; idx_new < n_new - 1; idx_new += 64) {
        const double t_new = dt_new * (double)idx_new;
        int idx_old = 1;
        while (t_new > (double)idx_old * dt_old && idx_old < n) ++idx_old;
        const double t_old_p1 = (double)idx_old * dt_old;
        const double* xp = &xo[3 * (idx_old - 1)];
        const double* xc = idx_old < n - 1 ? &xo[3 * idx_old] : &xo[3 * (n - 1)];
        const double fr = (t_new - (t_old_p1 - dt_old)) / dt_old;
        for (int i = 0; i < 2; ++i) { const double d = xc[i] - xp[i]; const double t = fr * d; x[3 * idx_new + i] = xp[i] + t; }
        { const double w = gu_normalize_theta(xc[2] - xp[2]); const double t = fr * w; x[3 * idx_new + 2] = gu_normalize_theta(xp[2] + t); }
        for (int j = 0; j < 2; ++j) u[2 * idx_new + j] = uo[2 * (idx_old - 1) + j];
    }
    __syncthreads();
    if (lane == 0) {
        for (int i = 0; i < 3; ++i) x[3 * (n_new - 1) + i] = xo[3 * (n - 1) + i];
        for (int j = 0; j < 2; ++j) u[2 * (n_new - 1) + j] = u[2 * (n_new - 2) + j];
        a.dt[b] = dt_new;
        if (a.n_grid) a.n_grid[b] = n_new;
        if (a.dual) a.dual[(size_t)b * a.dual_words] = 0.0;      // the kept multipliers belong to another grid
    }
}

}  // namespace mpc
