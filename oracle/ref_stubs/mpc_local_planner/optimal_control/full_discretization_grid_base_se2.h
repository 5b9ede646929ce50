// ORACLE (test infrastructure only).  This one file SHADOWS a reference header: src/optimal_control/stage_inequality_se2.cpp includes the
// reference's full_discretization_grid_base_se2.h only to read the grid's states (fd_grid->getState(k), :80-86); the real header drags in corbo's
// vertex / edge machinery.  Stand-in: the two members that .cpp uses, plus math_utils.h (the real one), which the real header brings along for cross2d.
#pragma once
#include <corbo-optimal-control/functions/stage_functions.h>
#include <mpc_local_planner/utils/math_utils.h>

namespace mpc_local_planner {
class FullDiscretizationGridBaseSE2 : public corbo::DiscretizationGridInterface {
 public:
    virtual const Eigen::VectorXd& getState(int k) const = 0;
};
}  // namespace mpc_local_planner
