// mpc_costmap.hpp -- costmap -> point obstacles on the device (the step in front of the solve).
//
// Restates MpcLocalPlannerROS::updateObstacleContainerWithCostmap (src/mpc_local_planner_ros.cpp:474-499): every LETHAL cell of the
// local costmap becomes a point obstacle at the cell centre, except cells behind the robot (negative projection on the heading) that are
// farther away than costmap_obstacles_behind_robot_dist.  Third-party pieces it calls, restated from costmap_2d (ROS navigation, any
// version): getCost(mx, my) = costmap[my * size_x + mx], LETHAL_OBSTACLE = 254, mapToWorld: w = origin + (m + 0.5) * resolution.
// The obstacle ORDER is the reference's loop order (x index outer, y index inner, the last row and column are not visited): the
// association in the solve breaks distance ties by container order, so the compaction is stable in that order.
//
// Kernel: one workgroup (256 threads) per planner instance.  The columns are cut into groups of 16 (one 16-byte load per row), the rows
// into R bands; thread (band, group) owns its 16 columns x band rows exclusively, neighbouring threads read neighbouring 16-byte pieces of
// a costmap row (coalesced).  Pass 1 counts the kept cells per (column, band) into LDS -- a 16-byte piece without a lethal byte is
// rejected with one SWAR test -- a workgroup scan over (column outer, band inner) turns the counts into output positions, pass 2
// (served from L2) walks the same pieces again and writes the obstacles: stable in the reference's order without any sorting.
// HBM-bound byte work: algorithmic traffic = size_x * size_y bytes read + 20 bytes written per obstacle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpc {

constexpr int kCostmapThreads = 256;
constexpr uint8_t kLethal = 254;      // costmap_2d::LETHAL_OBSTACLE

struct CostmapArgs {
    const uint8_t* cost;      // [B][size_y][size_x]
    const double* origin;     // [B][2]   world coordinates of the lower-left corner of cell (0, 0)
    const double* pose;       // [B][3]   robot pose x, y, theta
    int32_t size_x, size_y;
    double resolution, behind_dist;
    int32_t O, V;             // capacity (obstacles per instance) and vertex stride of the mpc_obstacles layout
    int32_t* n_obstacles;     // [B]
    int32_t* n_vertices;      // [B][O]
    double* vertices;         // [B][O][V][2]
    int32_t* dropped;         // [B] or NULL: kept cells that did not fit into O
};

// products and sums are kept un-fused (hipcc contracts a + b * c into an fma by default, and the __d*_rn intrinsics are plain operators
// that get contracted as well) so that the cell centres are bit-identical to the reference's double arithmetic
__device__ __forceinline__ double cm_world(double origin, int m, double res) {
#pragma clang fp contract(off)
    const double c = ((double)m + 0.5) * res;
    return origin + c;
}

__device__ __forceinline__ bool cm_keep(double wx, double wy, double px, double py, double ox, double oy, double behind) {
#pragma clang fp contract(off)
    const double dx = wx - px, dy = wy - py;
    const double a = dx * ox, b = dy * oy;
    const double dot = a + b;
    if (!(dot < 0.0)) return true;
    const double xx = dx * dx, yy = dy * dy;
    const double nrm = __builtin_sqrt(xx + yy);
    return !(nrm > behind);
}

constexpr int kCmPiece = 16;                      // bytes per load = columns per group
constexpr int kCmRows = 4;                        // rows loaded back to back per thread (memory-level parallelism)
constexpr int kCmSlabCols = kCmPiece * kCostmapThreads;   // widest slab handled in one sweep (4096 columns)

__device__ __forceinline__ bool cm_has_lethal(uint32_t v) {          // any byte == 254 ?
    const uint32_t x = v ^ 0xFEFEFEFEu;
    return ((x - 0x01010101u) & ~x & 0x80808080u) != 0u;
}

struct CmPiece { uint32_t w[4]; };
// 16 bytes (unaligned), bytes >= valid read as 0.  All indices are compile-time constants after unrolling: a dynamically indexed
// piece would live in scratch memory.
__device__ __forceinline__ CmPiece cm_load(const uint8_t* p, int valid) {
    CmPiece q;
    if (valid >= kCmPiece) __builtin_memcpy(&q, p, kCmPiece);
    else {
        q.w[0] = q.w[1] = q.w[2] = q.w[3] = 0u;
#pragma unroll
        for (int e = 0; e < kCmPiece; ++e) if (e < valid) q.w[e >> 2] |= (uint32_t)p[e] << (8 * (e & 3));
    }
    return q;
}
__device__ __forceinline__ bool cm_any_lethal(const CmPiece& q) {
    return cm_has_lethal(q.w[0]) | cm_has_lethal(q.w[1]) | cm_has_lethal(q.w[2]) | cm_has_lethal(q.w[3]);
}
// bit e set <=> byte e of the piece is LETHAL (branch-free SWAR per dword; exact, no cross-byte carries: x ^ 0xFE is 0 only there)
__device__ __forceinline__ uint32_t cm_lethal_mask(const CmPiece& q) {
    uint32_t m = 0u;
#pragma unroll
    for (int e = 0; e < kCmPiece; ++e) m |= (((q.w[e >> 2] >> (8 * (e & 3))) & 0xFFu) == kLethal ? 1u : 0u) << e;
    return m;
}

__global__ __launch_bounds__(kCostmapThreads) void costmap_to_obstacles_kernel(CostmapArgs a) {
    __shared__ int cnt[kCmSlabCols];                  // (column in slab) * R + band  ->  count, then output position
    __shared__ int part[kCostmapThreads];
    __shared__ int base_s;
    const int b = blockIdx.x, t = threadIdx.x;
    const int W = a.size_x, H = a.size_y;
    const uint8_t* cm = a.cost + (size_t)b * W * H;
    const double ox0 = a.origin[2 * b], oy0 = a.origin[2 * b + 1];
    const double px = a.pose[3 * b], py = a.pose[3 * b + 1], th = a.pose[3 * b + 2];
    const double hx = cos(th), hy = sin(th);          // PoseSE2::orientationUnitVec
    int32_t* nv = a.n_vertices + (size_t)b * a.O;
    double* vv = a.vertices + (size_t)b * a.O * a.V * 2;
    const int ncol = W - 1, nrow = H - 1;             // visited columns / rows (:481-483)
    if (t == 0) base_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < ncol; c0 += kCmSlabCols) {
        const int scol = ncol - c0 < kCmSlabCols ? ncol - c0 : kCmSlabCols;       // columns of this slab
        const int G = (scol + kCmPiece - 1) / kCmPiece;                           // column groups (<= 256)
        const int R = kCostmapThreads / G;                                        // row bands
        const int rows_per = (nrow + R - 1) / R;
        const int grp = t % G, band = t / G;
        const bool act = band < R;
        const int i0 = c0 + grp * kCmPiece;                                       // first column of the group
        const int valid = ncol - i0 < kCmPiece ? ncol - i0 : kCmPiece;            // visited columns in the group
        const int j0 = band * rows_per, j1 = (j0 + rows_per < nrow) ? j0 + rows_per : nrow;
        const int N = G * kCmPiece * R;
        for (int e = t; e < N; e += kCostmapThreads) cnt[e] = 0;
        __syncthreads();
        // ---- pass 1: counts per (column, band); kCmRows rows are loaded back to back (loads in flight) before they are examined
        if (act) {
            for (int jb = j0; jb < j1; jb += kCmRows) {
                CmPiece q[kCmRows];
#pragma unroll
                for (int r = 0; r < kCmRows; ++r) { const int j = jb + r < j1 ? jb + r : j1 - 1; q[r] = cm_load(cm + (size_t)j * W + i0, valid); }
#pragma unroll
                for (int r = 0; r < kCmRows; ++r) {
                    const int j = jb + r;
                    if (j >= j1 || !cm_any_lethal(q[r])) continue;
                    const double wy = cm_world(oy0, j, a.resolution);
                    for (uint32_t m = cm_lethal_mask(q[r]); m; m &= m - 1) {
                        const int e = __builtin_ctz(m);
                        if (cm_keep(cm_world(ox0, i0 + e, a.resolution), wy, px, py, hx, hy, a.behind_dist)) cnt[(grp * kCmPiece + e) * R + band] += 1;
                    }
                }
            }
        }
        __syncthreads();
        // ---- exclusive scan of cnt[0..N) (order: column outer, band inner): serial over a thread's segment + Hillis-Steele over the segments
        const int seg = (N + kCostmapThreads - 1) / kCostmapThreads;
        const int s0 = t * seg, s1 = s0 + seg < N ? s0 + seg : N;
        int sum = 0;
        for (int e = s0; e < s1; ++e) sum += cnt[e];
        part[t] = sum;
        __syncthreads();
        for (int off = 1; off < kCostmapThreads; off <<= 1) {
            const int v = t >= off ? part[t - off] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        const int base = base_s;
        const int total = part[kCostmapThreads - 1];
        int run = base + part[t] - sum;
        for (int e = s0; e < s1; ++e) { const int c = cnt[e]; cnt[e] = run; run += c; }
        __syncthreads();
        // ---- pass 2: write the obstacles at their positions
        if (act) {
            for (int jb = j0; jb < j1; jb += kCmRows) {
                CmPiece q[kCmRows];
#pragma unroll
                for (int r = 0; r < kCmRows; ++r) { const int j = jb + r < j1 ? jb + r : j1 - 1; q[r] = cm_load(cm + (size_t)j * W + i0, valid); }
#pragma unroll
                for (int r = 0; r < kCmRows; ++r) {
                    const int j = jb + r;
                    if (j >= j1 || !cm_any_lethal(q[r])) continue;
                    const double wy = cm_world(oy0, j, a.resolution);
                    for (uint32_t m = cm_lethal_mask(q[r]); m; m &= m - 1) {
                        const int e = __builtin_ctz(m);
                        const double wx = cm_world(ox0, i0 + e, a.resolution);
                        if (!cm_keep(wx, wy, px, py, hx, hy, a.behind_dist)) continue;
                        const int pos = cnt[(grp * kCmPiece + e) * R + band]++;
                        if (pos < a.O) { vv[(size_t)pos * a.V * 2] = wx; vv[(size_t)pos * a.V * 2 + 1] = wy; nv[pos] = 1; }
                    }
                }
            }
        }
        __syncthreads();
        if (t == 0) base_s = base + total;
        __syncthreads();
    }
    if (t == 0) {
        const int total = base_s;
        a.n_obstacles[b] = total < a.O ? total : a.O;
        if (a.dropped) a.dropped[b] = total > a.O ? total - a.O : 0;
    }
}

}  // namespace mpc
