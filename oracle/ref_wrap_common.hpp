// ORACLE (test infrastructure only): helpers shared by the wrapper translation units of oracle/_ref (see ref_wrap_grid.cpp): a reference grid object that holds
// a GIVEN trajectory, built through the reference's own update() -> initializeSequences(x0, xf, xinit, uinit, ...).
#pragma once
#include <mpc_local_planner/optimal_control/finite_differences_variable_grid_se2.h>

namespace {
using namespace mpc_local_planner;
struct Model3 : corbo::SystemDynamicsInterface {       // only the dimensions are asked for (update(): asserts, Zero(getInputDimension()))
    Ptr getInstance() const override { return std::make_shared<Model3>(); }
    int getInputDimension() const override { return 2; }
    int getStateDimension() const override { return 3; }
    bool isContinuousTime() const override { return true; }
    bool isLinear() const override { return false; }
    void dynamics(const Eigen::Ref<const StateVector>&, const Eigen::Ref<const ControlVector>&, Eigen::Ref<StateVector>) const override {}
};
template <class Base>
struct Probe : Base {
    using Base::_x_seq; using Base::_u_seq; using Base::_xf; using Base::_dt; using Base::_n_adapt;
    using Base::warmStartShifting; using Base::findNearestState; using Base::resampleTrajectory; using Base::adaptGrid;
};
Eigen::VectorXd vec(const double* p, int n) { Eigen::VectorXd v(n); for (int i = 0; i < n; ++i) v[i] = p[i]; return v; }
corbo::ReferenceTrajectoryInterface table(const double* p, int rows, int dim, bool is_static) {
    corbo::ReferenceTrajectoryInterface r;
    r.dim = dim; r.is_static = is_static;
    for (int k = 0; k < rows; ++k) r.table.push_back(vec(p + dim * k, dim));
    return r;
}
// a grid holding the given trajectory: x [n][3], u [n-1][2], dt -- through the reference's own update() -> initializeSequences(x0, xf, xinit, uinit, ...)
template <class G>
void fill(G& g, corbo::NlpFunctions& nlp, int n, const double* x, const double* u, double dt, const bool xf_fixed[3]) {
    g.setNRef(n);
    g.setDtRef(dt);
    Eigen::Matrix<bool, -1, 1> fx(3);
    for (int i = 0; i < 3; ++i) fx[i] = xf_fixed[i];
    g.setXfFixed(fx);
    corbo::ReferenceTrajectoryInterface xtab = table(x, n, 3, false), utab = table(u, n - 1, 2, false);
    corbo::OptimizationEdgeSet edges;
    g.update(xtab.table[0], xtab, utab, nlp, edges, std::make_shared<Model3>(), true, corbo::Time(0.0), nullptr, nullptr, 0.0, &xtab, &utab);
}
template <class G>
int dump(const G& g, double* x, double* u, double* dt) {
    const int n = g.getN();
    for (int k = 0; k < n; ++k) { const Eigen::VectorXd& s = g.getState(k); for (int i = 0; i < 3; ++i) x[3 * k + i] = s[i]; }
    for (int k = 0; k < n - 1; ++k) for (int j = 0; j < 2; ++j) u[2 * k + j] = g._u_seq[(size_t)k].values()[j];
    *dt = g.getDt();
    return n;
}
}  // namespace

