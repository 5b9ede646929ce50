// ORACLE SUPPORT (test infrastructure only).  The ONE piece of the reference that compiles here from its own sources: utils/math_utils.h needs nothing but the C++ standard
// library.  This file includes it from where it lies under /root/reference (nothing is copied) and exports its functions with C linkage, so that the tests can hold the
// oracle's restatements (oracle/se2_nlp.py, oracle/mpc_oracle.c), the host build of the kernel core (tests/host_harness) and the device kernels against the reference's own
// arithmetic.  Every other translation unit of the reference includes Eigen, corbo (control_box_rst), ROS or teb_local_planner headers, none of which are in this image:
// unbuildable here, and no stand-ins are written for them.
//   make -C oracle ref   ->   oracle/_ref/libmpc_ref_math.so   (git-ignored; travels to the GPU box like the other built libraries)
#include <cmath>
#include <vector>      // math_utils.h uses std::vector without including it

#include <mpc_local_planner/utils/math_utils.h>

namespace {
struct V2 {      // the argument type of cross2d: anything with x() and y()
    double x_, y_;
    double x() const { return x_; }
    double y() const { return y_; }
};
struct P2 { double x, y; };      // the argument type of the templated distance_points2d: anything with members x and y
}  // namespace

extern "C" {
// inc/utils/math_utils.h:81-91
void ref_normalize_theta(int n, const double* theta, double* out) { for (int i = 0; i < n; ++i) out[i] = mpc_local_planner::normalize_theta(theta[i]); }
// :100-103
void ref_interpolate_angle(int n, const double* a1, const double* a2, const double* factor, double* out) {
    for (int i = 0; i < n; ++i) out[i] = mpc_local_planner::interpolate_angle(a1[i], a2[i], factor[i]);
}
// :36-48
double ref_average_angles(int n, const double* angles) { return mpc_local_planner::average_angles(std::vector<double>(angles, angles + n)); }
// :57-61 (templated) and :64 (scalar arguments)
void ref_distance_points2d(int n, const double* p1, const double* p2, double* out_templated, double* out_scalar) {
    for (int i = 0; i < n; ++i) {
        out_templated[i] = mpc_local_planner::distance_points2d(P2{p1[2 * i], p1[2 * i + 1]}, P2{p2[2 * i], p2[2 * i + 1]});
        out_scalar[i] = mpc_local_planner::distance_points2d(p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
    }
}
// :73-77
void ref_cross2d(int n, const double* v1, const double* v2, double* out) {
    for (int i = 0; i < n; ++i) out[i] = mpc_local_planner::cross2d(V2{v1[2 * i], v1[2 * i + 1]}, V2{v2[2 * i], v2[2 * i + 1]});
}
}
