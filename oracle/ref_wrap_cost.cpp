// ORACLE (test infrastructure only): C entry points around the reference's SE(2) cost and terminal-condition classes, compiled from where they lie under
// /root/reference:
//   src/optimal_control/quadratic_cost_se2.cpp            QuadraticFormCostSE2, QuadraticStateCostSE2: the state term / the integrand l(x_k, u_k)
//   src/optimal_control/final_state_conditions_se2.cpp    QuadraticFinalStateCostSE2, TerminalBallSE2
// (with their real headers).  Stand-ins: the corbo base classes reduced to the data members these sources read (oracle/ref_stubs/corbo-optimal-control/functions/);
// the weights are handed over as full matrices, the way Controller::configureOcp does (src/controller.cpp:605-612, :668, :703), or as diagonals (diagonal mode).
// Summing the terms over the grid (corbo's edges) is NOT reference code in this repository's sense and is not compiled.
#include <mpc_local_planner/optimal_control/final_state_conditions_se2.h>
#include <mpc_local_planner/optimal_control/quadratic_cost_se2.h>

namespace {
Eigen::VectorXd vec(const double* p, int n) { Eigen::VectorXd v(n); for (int i = 0; i < n; ++i) v[i] = p[i]; return v; }
Eigen::MatrixXd mat(const double* p, int n) { Eigen::MatrixXd m(n, n); for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) m(i, j) = p[n * i + j]; return m; }
Eigen::DiagonalMatrix<double, -1> diag(const double* p, int n, bool root) {
    Eigen::VectorXd d(n);
    for (int i = 0; i < n; ++i) d[i] = root ? std::sqrt(p[(n + 1) * i]) : p[(n + 1) * i];
    return Eigen::DiagonalMatrix<double, -1>(d);
}
corbo::ReferenceTrajectoryInterface table(const double* p, int rows, int dim) {
    corbo::ReferenceTrajectoryInterface r;
    r.dim = dim; r.is_static = false;
    for (int k = 0; k < rows; ++k) r.table.push_back(vec(p + dim * k, dim));
    return r;
}
}  // namespace

extern "C" {
// QuadraticFormCostSE2 (which != 0) or QuadraticStateCostSE2 (which == 0) at `count` samples: Q [3][3], R [2][2] row-major; diagonal != 0: diagonal mode (the
// diagonals of Q / R); integral != 0: computeIntegralStateControlTerm = l(x_k, u_k), one value per sample; else computeNonIntegralStateTerm: one value per sample,
// or with lsq != 0 (diagonal mode only) three.  u_ref == NULL: zero control reference (_zero_u_ref).  out [count][lsq ? 3 : 1]
void ref_quadratic_cost(int which, const double* Q, const double* R, int diagonal, int integral, int lsq, int count, const double* x, const double* x_ref,
                        const double* u, const double* u_ref, double* out) {
    using namespace mpc_local_planner;
    corbo::ReferenceTrajectoryInterface xr = table(x_ref, count, 3), ur = u_ref ? table(u_ref, count, 2) : corbo::ReferenceTrajectoryInterface();
    const int w = lsq ? 3 : 1;
    if (which) {
        QuadraticFormCostSE2 c(mat(Q, 3), mat(R, 2), integral != 0, lsq != 0);
        c._Q_diagonal_mode = c._R_diagonal_mode = diagonal != 0;
        c._Q_diag = diag(Q, 3, false); c._R_diag = diag(R, 2, false); c._Q_diag_sqrt = diag(Q, 3, true);
        c._x_ref = &xr; c._u_ref = &ur; c._zero_u_ref = u_ref == nullptr;
        for (int k = 0; k < count; ++k) {
            Eigen::VectorXd r(w);
            if (integral) c.computeIntegralStateControlTerm(k, vec(x + 3 * k, 3), vec(u + 2 * k, 2), r);
            else c.computeNonIntegralStateTerm(k, vec(x + 3 * k, 3), r);
            for (int i = 0; i < w; ++i) out[w * k + i] = r[i];
        }
    } else {
        QuadraticStateCostSE2 c(mat(Q, 3), integral != 0, lsq != 0);
        c._diagonal_mode = diagonal != 0;
        c._Q_diag = diag(Q, 3, false); c._Q_diag_sqrt = diag(Q, 3, true);
        c._x_ref = &xr; c._u_ref = &ur;
        for (int k = 0; k < count; ++k) {
            Eigen::VectorXd r(w);
            if (integral) c.computeIntegralStateControlTerm(k, vec(x + 3 * k, 3), vec(u + 2 * k, 2), r);
            else c.computeNonIntegralStateTerm(k, vec(x + 3 * k, 3), r);
            for (int i = 0; i < w; ++i) out[w * k + i] = r[i];
        }
    }
}
// QuadraticFinalStateCostSE2::computeNonIntegralStateTerm: Qf [3][3]; out [count][lsq ? 3 : 1]
void ref_final_state_cost(const double* Qf, int diagonal, int lsq, int count, const double* x, const double* x_ref, double* out) {
    mpc_local_planner::QuadraticFinalStateCostSE2 c(mat(Qf, 3), lsq != 0);
    corbo::ReferenceTrajectoryInterface xr = table(x_ref, count, 3);
    c._diagonal_mode = diagonal != 0; c._Qf_diag = diag(Qf, 3, false); c._Qf_diag_sqrt = diag(Qf, 3, true); c._x_ref = &xr;
    const int w = lsq ? 3 : 1;
    for (int k = 0; k < count; ++k) {
        Eigen::VectorXd r(w);
        c.computeNonIntegralStateTerm(k, vec(x + 3 * k, 3), r);
        for (int i = 0; i < w; ++i) out[w * k + i] = r[i];
    }
}
// TerminalBallSE2::computeNonIntegralStateTerm: S [3][3], gamma; out [count]
void ref_terminal_ball(const double* S, double gamma, int diagonal, int count, const double* x, const double* x_ref, double* out) {
    mpc_local_planner::TerminalBallSE2 c(mat(S, 3), gamma);
    corbo::ReferenceTrajectoryInterface xr = table(x_ref, count, 3);
    c._diagonal_mode = diagonal != 0; c._S_diag = diag(S, 3, false); c._x_ref = &xr;
    for (int k = 0; k < count; ++k) {
        Eigen::VectorXd r(1);
        c.computeNonIntegralStateTerm(k, vec(x + 3 * k, 3), r);
        out[k] = r[0];
    }
}
}  // extern "C"
