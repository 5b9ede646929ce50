// ORACLE (test infrastructure only).  This one file SHADOWS a reference header: src/optimal_control/stage_inequality_se2.cpp and
// min_time_via_points_cost.cpp include the reference's full_discretization_grid_base_se2.h only to read the grid (getState(k), getN(), findClosestPose);
// the real header drags in corbo's vertex / edge machinery.  Stand-in: those members, plus math_utils.h (the real one), which the real header brings along
// for cross2d.  findClosestPose is a RESTATEMENT (the reference's lives in the class this file replaces, src/optimal_control/
// full_discretization_grid_base_se2.cpp:364-388): first minimum over the states in front of the final one, the final state only if strictly closer.
#pragma once
#include <corbo-optimal-control/functions/stage_functions.h>
#include <mpc_local_planner/utils/math_utils.h>

namespace mpc_local_planner {
class FullDiscretizationGridBaseSE2 : public corbo::DiscretizationGridInterface {
 public:
    virtual const Eigen::VectorXd& getState(int k) const = 0;
    virtual int getN() const = 0;
    int findClosestPose(double x_ref, double y_ref, int start_idx = 0, double* distance = nullptr) const {
        const int n = getN();
        double min_dist = 1.7976931348623157e308;
        int min_idx = -1;
        for (int i = start_idx; i < n - 1; ++i) {
            const Eigen::VectorXd& x = getState(i);
            const double dist = distance_points2d(x_ref, y_ref, x[0], x[1]);
            if (dist < min_dist) { min_dist = dist; min_idx = i; }
        }
        const Eigen::VectorXd& xf = getState(n - 1);
        const double dist = distance_points2d(x_ref, y_ref, xf[0], xf[1]);
        if (dist < min_dist) { min_dist = dist; min_idx = n - 1; }
        if (distance) *distance = min_dist;
        return min_idx;
    }
};
}  // namespace mpc_local_planner
