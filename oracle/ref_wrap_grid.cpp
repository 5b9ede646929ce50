// ORACLE (test infrastructure only): C entry points around the reference's grid classes, compiled from where they lie under /root/reference:
//   src/optimal_control/full_discretization_grid_base_se2.cpp      update(), initializeSequences() (both), warmStartShifting(), findNearestState(),
//                                                                  resampleTrajectory(), findClosestPose(), getStateAndControlTimeSeries()
//   src/optimal_control/finite_differences_variable_grid_se2.cpp   adaptGrid() / adaptGridTimeBasedSingleStep()
//   src/utils/time_series_se2.cpp                                  TimeSeriesSE2::getValuesInterpolate (the initial state trajectory is one of these)
// (with their real headers full_discretization_grid_base_se2.h, finite_differences_grid_se2.h, finite_differences_variable_grid_se2.h).  Stand-ins, all under
// oracle/ref_stubs/: corbo's grid / vertex / reference-trajectory / time-series interfaces reduced to what these sources touch, the element-wise part of
// Eigen, and a simplified vector_vertex_se2.h (shadowed: the grid only stores values in its vertices).  createEdges() -- corbo's hyper-graph edges -- is NOT
// compiled; FiniteDifferencesGridSE2::createEdges is an empty function here.
#include "ref_wrap_common.hpp"
#include <mpc_local_planner/utils/time_series_se2.h>

namespace mpc_local_planner {
void FiniteDifferencesGridSE2::createEdges(NlpFunctions&, OptimizationEdgeSet&, SystemDynamicsInterface::Ptr) {}      // see above
}

extern "C" {
// the cold start of an EMPTY grid: update() with a static goal reference.  xinit == NULL: initializeSequences(x0, xf, uref, ...) (:136-190, heading = direction of
// travel); else xinit [n][3]: initializeSequences(x0, xf, xinit, uref, ...) (:192-239).  Controls = uref (zero).  x_out [n][3], u_out [n-1][2].
void ref_grid_cold_start(int n, double dt_ref, const double* x0, const double* xf, const double* xinit, double* x_out, double* u_out) {
    Probe<FiniteDifferencesGridSE2> g;
    corbo::NlpFunctions nlp;
    g.setNRef(n); g.setDtRef(dt_ref);
    Eigen::Matrix<bool, -1, 1> fx(3); for (int i = 0; i < 3; ++i) fx[i] = true;
    g.setXfFixed(fx);
    corbo::ReferenceTrajectoryInterface xref, uref, xi;
    xref.dim = 3; xref.is_static = true; xref.table.push_back(vec(xf, 3));            // StaticReference(xf) (src/controller.cpp:169)
    uref.dim = 2; uref.is_static = true; uref.table.push_back(Eigen::VectorXd(2));    // ZeroReference
    if (xinit) xi = table(xinit, n, 3, false);
    corbo::OptimizationEdgeSet edges;
    g.update(vec(x0, 3), xref, uref, nlp, edges, std::make_shared<Model3>(), true, corbo::Time(0.0), nullptr, nullptr, 0.0, xinit ? &xi : nullptr, nullptr);
    double dt;
    dump(g, x_out, u_out, &dt);
}
int ref_grid_find_nearest_state(int n, const double* x, const double* u, double dt, const double* x0_new) {
    Probe<FiniteDifferencesGridSE2> g; corbo::NlpFunctions nlp; const bool fx[3] = {true, true, true};
    fill(g, nlp, n, x, u, dt, fx);
    return g.findNearestState(vec(x0_new, 3));
}
// the next cycle of a NON-EMPTY fixed grid with the moving-horizon warm start: update(new_run = true) = warmStartShifting(x0) (:241-302), then the start state is
// overwritten by x0 and the fixed goal components by the reference (:107-116)
void ref_grid_warm_start_cycle(int n, const double* x, const double* u, double dt, const double* x0_new, const double* xf_new, const int* xf_fixed, double* x_out, double* u_out) {
    Probe<FiniteDifferencesGridSE2> g; corbo::NlpFunctions nlp; const bool fx[3] = {xf_fixed[0] != 0, xf_fixed[1] != 0, xf_fixed[2] != 0};
    fill(g, nlp, n, x, u, dt, fx);
    g.setWarmStart(true);
    corbo::ReferenceTrajectoryInterface xref, uref;
    xref.dim = 3; xref.is_static = true; xref.table.push_back(vec(xf_new, 3));
    uref.dim = 2; uref.is_static = true; uref.table.push_back(Eigen::VectorXd(2));
    corbo::OptimizationEdgeSet edges;
    g.update(vec(x0_new, 3), xref, uref, nlp, edges, std::make_shared<Model3>(), true, corbo::Time(0.1));
    double d;
    dump(g, x_out, u_out, &d);
}
// resampleTrajectory(n_new) (:440-524): capacity of the outputs max(n, n_new); returns the new grid size, *dt_out the new dt
int ref_grid_resample(int n, const double* x, const double* u, double dt, int n_new, double* x_out, double* u_out, double* dt_out) {
    Probe<FiniteDifferencesGridSE2> g; corbo::NlpFunctions nlp; const bool fx[3] = {true, true, true};
    fill(g, nlp, n, x, u, dt, fx);
    g.resampleTrajectory(n_new);
    return dump(g, x_out, u_out, dt_out);
}
// the variable grid's adaptation step at the start of a cycle: adaptGrid() -> adaptGridTimeBasedSingleStep (finite_differences_variable_grid_se2.cpp:99-121)
int ref_grid_adapt(int n, const double* x, const double* u, double dt, double dt_ref, int n_max, int n_min, double hyst, double* x_out, double* u_out, double* dt_out) {
    Probe<FiniteDifferencesVariableGridSE2> g; corbo::NlpFunctions nlp; const bool fx[3] = {true, true, true};
    fill(g, nlp, n, x, u, dt, fx);
    g.setDtRef(dt_ref);
    g.setGridAdaptTimeBasedSingleStep(n_max, hyst, true);
    g.setNmin(n_min);
    g.adaptGrid(true, nlp);
    return dump(g, x_out, u_out, dt_out);
}
int ref_grid_find_closest_pose(int n, const double* x, const double* u, double dt, double x_ref, double y_ref, int start_idx) {
    Probe<FiniteDifferencesGridSE2> g; corbo::NlpFunctions nlp; const bool fx[3] = {true, true, true};
    fill(g, nlp, n, x, u, dt, fx);
    return g.findClosestPose(x_ref, y_ref, start_idx);
}
// getStateAndControlTimeSeries (:579-615): returns the number of samples; t [n], xs [n][3], us [n][2] (the last control duplicated)
int ref_grid_time_series(int n, const double* x, const double* u, double dt, double* t, double* xs, double* us) {
    Probe<FiniteDifferencesGridSE2> g; corbo::NlpFunctions nlp; const bool fx[3] = {true, true, true};
    fill(g, nlp, n, x, u, dt, fx);
    auto X = std::make_shared<corbo::TimeSeries>(), U = std::make_shared<corbo::TimeSeries>();
    g.getStateAndControlTimeSeries(X, U);
    const int m = (int)X->times().size();
    for (int k = 0; k < m; ++k) { t[k] = X->times()[(size_t)k]; for (int i = 0; i < 3; ++i) xs[3 * k + i] = X->samples()[(size_t)k][i]; }
    for (int k = 0; k < (int)U->times().size(); ++k) for (int j = 0; j < 2; ++j) us[2 * k + j] = U->samples()[(size_t)k][j];
    return m == (int)U->times().size() ? m : -1;
}
// TimeSeriesSE2::getValuesInterpolate (src/utils/time_series_se2.cpp:34-111) of the series (times [m], values [m][3]) at t [count]; linear != 0: Linear, else
// ZeroOrderHold interpolation; hold != 0: ZeroOrderHold extrapolation, else none.  out [count][3] (left untouched where the call returns false), ok [count].
void ref_time_series_se2_interpolate(int m, const double* times, const double* values, int linear, int hold, int count, const double* t, double* out, int* ok) {
    TimeSeriesSE2 ts(3);
    for (int i = 0; i < m; ++i) ts.add(times[i], vec(values + 3 * i, 3));
    for (int k = 0; k < count; ++k) {
        Eigen::VectorXd v(3);
        for (int i = 0; i < 3; ++i) v[i] = out[3 * k + i];
        ok[k] = ts.getValuesInterpolate(t[k], v, linear ? corbo::TimeSeries::Interpolation::Linear : corbo::TimeSeries::Interpolation::ZeroOrderHold,
                                        hold ? corbo::TimeSeries::Extrapolation::ZeroOrderHold : corbo::TimeSeries::Extrapolation::NoExtrapolation) ? 1 : 0;
        for (int i = 0; i < 3; ++i) out[3 * k + i] = v[i];
    }
}
}  // extern "C"
