// mpc_feasibility.hpp -- post-solve costmap feasibility check of the planned pose trajectories on the device.
//
// Restates Controller::isPoseTrajectoryFeasible (src/controller.cpp:859-917): every grid point up to look_ahead_idx, plus interpolated poses
// where two neighbours are farther apart than the inscribed radius or turn by more than min_resolution_collision_check_angular (:889-912; the
// intermediate pose is accumulated step by step as the reference does), is tested with base_local_planner::CostmapModel::footprintCost and the
// trajectory is infeasible iff one call returns -1.
// PINNED third-party convention (ROS navigation 1.17 / noetic: costmap_model.cpp, line_iterator.h, Costmap2D::worldToMap -- absent from the
// reference tree, restated from their published source): footprintCost returns the FIRST negative code it meets in the order
//   centre cell outside the map -> -3 | (< 3 footprint points: centre cell NO_INFORMATION -> -2, LETHAL or INSCRIBED -> -1) |
//   per edge i -> i+1, then last -> first: an endpoint outside the map -> -3; cells of the Bresenham line in order: NO_INFORMATION (255) -> -2,
//   LETHAL (254) -> -1.
// Kernel: one workgroup per planner instance, one thread per grid interval i (pose i and the poses interpolated towards i + 1); the costmap
// of an instance is read where the footprint outlines lie (a few hundred byte reads per pose, served by L2).  Byte / index work: the double
// arithmetic that decides cell indices is compiled un-fused so that it is bit-identical to the reference's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpc {

constexpr int kFeasThreads = 128;
constexpr int kFeasMaxSpec = 32;

struct FeasArgs {
    const double* x;          // [B][n_stride][3] planned states
    const int32_t* n_grid;    // [B] grid points per instance or NULL (= n_stride)
    int32_t n_stride;
    const uint8_t* cost;      // [B][size_y][size_x]
    const double* origin;     // [B][2]
    int32_t size_x, size_y;
    double resolution;
    int32_t n_spec;           // footprint points (robot frame)
    double spec[2 * kFeasMaxSpec];
    double inscribed_radius, min_res_angular;
    int32_t look_ahead_idx;
    int32_t* feasible;        // [B] 1 = feasible, 0 = not
};

__device__ __forceinline__ bool feas_world_to_map(const FeasArgs& a, double ox, double oy, double wx, double wy, int& mx, int& my) {
#pragma clang fp contract(off)
    if (wx < ox || wy < oy) return false;
    const double fx = (wx - ox) / a.resolution, fy = (wy - oy) / a.resolution;
    if (!(fx < 2147483000.0) || !(fy < 2147483000.0)) return false;
    mx = (int)fx; my = (int)fy;
    return mx < a.size_x && my < a.size_y;
}

// base_local_planner::CostmapModel::footprintCost == -1 ?   (the first negative code decides; see the header comment)
__device__ __forceinline__ bool feas_pose_hits_lethal(const FeasArgs& a, const uint8_t* cost, double ox, double oy, double x, double y, double th) {
#pragma clang fp contract(off)
    int cx, cy;
    if (!feas_world_to_map(a, ox, oy, x, y, cx, cy)) return false;                    // -3
    if (a.n_spec < 3) {
        const uint8_t v = cost[(size_t)cy * a.size_x + cx];
        return v == 254 || v == 253;                                                     // 255 -> -2
    }
    const double ct = ::cos(th), st = ::sin(th);
    double pxw, pyw;
    {
        const double sx = a.spec[0], sy = a.spec[1];
        const double t0 = sx * ct, t1 = sy * st, t2 = sx * st, t3 = sy * ct;
        pxw = x + (t0 - t1); pyw = y + (t2 + t3);
    }
    const double fx0 = pxw, fy0 = pyw;
    for (int i = 0; i < a.n_spec; ++i) {
        double qx, qy;
        if (i + 1 < a.n_spec) {
            const double sx = a.spec[2 * (i + 1)], sy = a.spec[2 * (i + 1) + 1];
            const double t0 = sx * ct, t1 = sy * st, t2 = sx * st, t3 = sy * ct;
            qx = x + (t0 - t1); qy = y + (t2 + t3);
        } else { qx = fx0; qy = fy0; }
        int x0, y0, x1, y1;
        if (!feas_world_to_map(a, ox, oy, pxw, pyw, x0, y0)) return false;             // -3
        if (!feas_world_to_map(a, ox, oy, qx, qy, x1, y1)) return false;               // -3
        // LineIterator (Bresenham)
        const int dx = x1 >= x0 ? x1 - x0 : x0 - x1, dy = y1 >= y0 ? y1 - y0 : y0 - y1;
        int xinc1 = x1 >= x0 ? 1 : -1, xinc2 = xinc1, yinc1 = y1 >= y0 ? 1 : -1, yinc2 = yinc1;
        int den, num, numadd, numpixels;
        if (dx >= dy) { xinc1 = 0; yinc2 = 0; den = dx; num = dx / 2; numadd = dy; numpixels = dx; }
        else { xinc2 = 0; yinc1 = 0; den = dy; num = dy / 2; numadd = dx; numpixels = dy; }
        int cxp = x0, cyp = y0;
        for (int p = 0; p <= numpixels; ++p) {
            const uint8_t v = cost[(size_t)cyp * a.size_x + cxp];
            if (v == 255) return false;                                                  // -2
            if (v == 254) return true;                                                   // -1
            num += numadd;
            if (num >= den) { num -= den; cxp += xinc1; cyp += yinc1; }
            cxp += xinc2; cyp += yinc2;
        }
        pxw = qx; pyw = qy;
    }
    return false;
}

__device__ __forceinline__ double feas_normalize_theta(double th) {
#pragma clang fp contract(off)
    const double pi = 3.14159265358979323846;
    if (th >= -pi && th < pi) return th;
    double m = th - ::floor(th / (2 * pi)) * 2 * pi;
    if (m >= pi) m -= 2 * pi;
    if (m < -pi) m += 2 * pi;
    return m;
}

__global__ __launch_bounds__(kFeasThreads) void feasibility_kernel(FeasArgs a) {
#pragma clang fp contract(off)
    __shared__ int bad;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    int n = a.n_stride;
    if (a.n_grid) { n = a.n_grid[b]; n = n < 1 ? 1 : (n > a.n_stride ? a.n_stride : n); }
    const double* x = a.x + (size_t)b * a.n_stride * 3;
    const uint8_t* cost = a.cost + (size_t)b * a.size_x * a.size_y;
    const double ox = a.origin[2 * b], oy = a.origin[2 * b + 1];
    int la = a.look_ahead_idx;
    if (la < 0 || la >= n) la = n - 1;
    if (n < 2) { if (threadIdx.x == 0) a.feasible[b] = 0; return; }
    for (int i = threadIdx.x; i <= la; i += kFeasThreads) {
        const double xi = x[3 * i], yi = x[3 * i + 1], thi = x[3 * i + 2];
        bool hit = feas_pose_hits_lethal(a, cost, ox, oy, xi, yi, thi);
        if (!hit && i < la) {
            const double delta_rot = feas_normalize_theta(x[3 * (i + 1) + 2] - thi);
            const double ddx = x[3 * (i + 1)] - xi, ddy = x[3 * (i + 1) + 1] - yi;
            const double d2x = ddx * ddx, d2y = ddy * ddy;
            const double dist = __builtin_sqrt(d2x + d2y);
            const double arot = delta_rot < 0 ? -delta_rot : delta_rot;
            if (arot > a.min_res_angular || dist > a.inscribed_radius) {
                const double c1 = ::ceil(arot / a.min_res_angular), c2 = ::ceil(dist / a.inscribed_radius);
                const int n_add = (int)(c1 > c2 ? c1 : c2) - 1;
                double px = xi, py = yi, pth = thi;
                const double den = (double)n_add + 1.0;
                for (int s = 0; s < n_add && !hit; ++s) {
                    px = px + ddx / den; py = py + ddy / den;
                    pth = feas_normalize_theta(pth + delta_rot / den);
                    hit = feas_pose_hits_lethal(a, cost, ox, oy, px, py, pth);
                }
            }
        }
        if (hit) bad = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) a.feasible[b] = bad ? 0 : 1;
}

}  // namespace mpc
