// ORACLE (test infrastructure only): the reference's own vertex classes, compiled from include/mpc_local_planner/optimal_control/vector_vertex_se2.h where it
// lies under /root/reference (no stand-in shadows it any more: oracle/ref_stubs/ only supplies the corbo interface it overrides) and EXECUTED:
//   VectorVertexSE2::plus(const double*) / plus(int, double) / plusUnfixed / setData / set          vector_vertex_se2.h:79-118
//   PartiallyFixedVectorVertexSE2::plusUnfixed, setFixed, getNumberFinite*Bounds                     vector_vertex_se2.h:186-316
// This is SURVEY.md section 8 row a15: the retraction the reference's solver applies to every state vertex (x, y as reals, the heading back into [-pi, pi)).
// tests/golden/make_ref_vectors.py records the outputs (tests/golden/ref_vertex.npz), tests/test_reference_pinned.py holds the numpy oracle's retraction,
// the C oracle's and the host build of the kernel core's accept step to them bit for bit.
#include <mpc_local_planner/optimal_control/vector_vertex_se2.h>

namespace {
Eigen::VectorXd vvec(const double* p, int n) { Eigen::VectorXd v(n); for (int i = 0; i < n; ++i) v[i] = p[i]; return v; }
}

extern "C" {
// out = values after plus(inc) (whole-vector form), count vertices of dimension dim each
void ref_vertex_plus(int count, int dim, const double* values, const double* inc, double* out) {
    for (int i = 0; i < count; ++i) {
        mpc_local_planner::VectorVertexSE2 v(vvec(values + dim * i, dim));
        v.plus(inc + dim * i);
        for (int a = 0; a < dim; ++a) out[dim * i + a] = v.values()[a];
    }
}
// the same through the per-component form plus(idx, inc), component by component
void ref_vertex_plus_idx(int count, int dim, const double* values, const double* inc, double* out) {
    for (int i = 0; i < count; ++i) {
        mpc_local_planner::VectorVertexSE2 v(vvec(values + dim * i, dim));
        for (int a = 0; a < dim; ++a) v.plus(a, inc[dim * i + a]);
        for (int a = 0; a < dim; ++a) out[dim * i + a] = v.values()[a];
    }
}
// partially fixed vertex: inc carries one entry per UNFIXED component, in order (plusUnfixed); also returns getDimensionUnfixed()
int ref_vertex_plus_unfixed(int dim, const double* values, const int* fixed, const double* inc_unfixed, double* out) {
    Eigen::Array<bool, -1, 1> fx(dim);
    for (int a = 0; a < dim; ++a) fx[a] = fixed[a] != 0;
    mpc_local_planner::PartiallyFixedVectorVertexSE2 v(vvec(values, dim), fx);
    v.plusUnfixed(inc_unfixed);
    for (int a = 0; a < dim; ++a) out[a] = v.values()[a];
    return v.getDimensionUnfixed();
}
// setData(idx, data) on every component, then set(values, lb, ub): both wrap the heading
void ref_vertex_set(int dim, const double* data, double* out_set_data, double* out_set) {
    mpc_local_planner::VectorVertexSE2 v(dim);
    for (int a = 0; a < dim; ++a) v.setData(a, data[a]);
    for (int a = 0; a < dim; ++a) out_set_data[a] = v.values()[a];
    Eigen::VectorXd lb(dim), ub(dim);
    for (int a = 0; a < dim; ++a) { lb[a] = -corbo::CORBO_INF_DBL; ub[a] = corbo::CORBO_INF_DBL; }
    mpc_local_planner::VectorVertexSE2 w;
    w.set(vvec(data, dim), lb, ub, false);
    for (int a = 0; a < dim; ++a) out_set[a] = w.values()[a];
}
// bounds bookkeeping of the partially fixed vertex: [finite lower, finite upper, finite any] x (all, unfixed only)
void ref_vertex_bound_counts(int dim, const double* lb, const double* ub, const int* fixed, int* out6) {
    Eigen::Array<bool, -1, 1> fx(dim);
    for (int a = 0; a < dim; ++a) fx[a] = fixed[a] != 0;
    Eigen::VectorXd z(dim);
    mpc_local_planner::PartiallyFixedVectorVertexSE2 v(z, fx);
    v.setLowerBounds(vvec(lb, dim)); v.setUpperBounds(vvec(ub, dim));
    out6[0] = v.getNumberFiniteLowerBounds(false); out6[1] = v.getNumberFiniteUpperBounds(false); out6[2] = v.getNumberFiniteBounds(false);
    out6[3] = v.getNumberFiniteLowerBounds(true); out6[4] = v.getNumberFiniteUpperBounds(true); out6[5] = v.getNumberFiniteBounds(true);
}
}
