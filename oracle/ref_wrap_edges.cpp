// ORACLE (test infrastructure only): the reference's FiniteDifferencesGridSE2::createEdges (src/optimal_control/finite_differences_grid_se2.cpp:36-154), compiled from where it
// lies and EXECUTED through the grid's update(); corbo's edge classes and NlpFunctions are records (oracle/ref_stubs/): what is dumped is WHICH edge kinds createEdges asks for
// at every grid point and WHICH vertices it connects them to -- the previous control and its dt at k = 0, the final state as x_{k+1} of the last interval, the integral edge kind
// per integration rule, the final-state edges only while the final state is not fixed, the final control-deviation edge on (u_ref, u_{n-2}, dt).  Own library
// (oracle/_ref/libmpc_ref_edges.so): libmpc_ref.so carries an empty createEdges (oracle/ref_wrap_grid.cpp).
#include <cstring>
#include <map>
#include <sstream>
#include "ref_wrap_common.hpp"

extern "C" {
// a grid holding the trajectory x [n][3], u [n-1][2], dt with the given fixed-goal flags; cost_integration 0 left sum, 1 trapezoidal rule; the *_integral flags say whether
// the stage cost / equalities / inequalities report integral terms; final_cost / final_constraint (0 none, 1 inequality, 2 equality).  Text out: one edge per line,
// "<set>|<kind>|<k>|<vertex>,<vertex>,..." with vertices named x0.. / xf / u0.. / dt / u_prev / u_prev_dt / u_ref; returns the text length
int ref_edges_dump(int n, const double* x, const double* u, double dt, const int* xf_fixed, int cost_integration, int cost_integral, int eq_integral, int ineq_integral, int final_cost,
                   int final_constraint, char* out, int cap) {
    struct EdgeProbe : Probe<FiniteDifferencesGridSE2> { using FiniteDifferencesGridSE2::_u_prev; using FiniteDifferencesGridSE2::_u_prev_dt; using FiniteDifferencesGridSE2::_u_ref; };
    EdgeProbe g;
    corbo::NlpFunctions nlp;
    auto handle = [](bool integral, bool equality = false) { auto h = std::make_shared<corbo::StageFunctionHandle>(); h->integral_terms = integral; h->equality = equality; return h; };
    nlp.stage_cost = handle(cost_integral != 0);
    nlp.stage_equalities = handle(eq_integral != 0);
    nlp.stage_inequalities = handle(ineq_integral != 0);
    if (final_constraint) nlp.final_stage_constraints = handle(false, final_constraint == 2);
    nlp.has_final_cost = final_cost != 0;
    g.setCostIntegrationRule(cost_integration ? FullDiscretizationGridBaseSE2::CostIntegrationRule::TrapezoidalRule : FullDiscretizationGridBaseSE2::CostIntegrationRule::LeftSum);
    const bool fx[3] = {xf_fixed[0] != 0, xf_fixed[1] != 0, xf_fixed[2] != 0};
    // fill() = update() on the empty grid, which ends in createEdges(); the edge set lives inside fill, so run update() once more on an own edge set
    fill(g, nlp, n, x, u, dt, fx);
    corbo::OptimizationEdgeSet edges;
    corbo::ReferenceTrajectoryInterface xref = table(x, n, 3, false), uref = table(u, n - 1, 2, false);
    g.setModified(true);
    g.update(xref.table[0], xref, uref, nlp, edges, std::make_shared<Model3>(), false, corbo::Time(0.0));
    std::map<const corbo::VertexInterface*, std::string> name;
    for (int k = 0; k < (int)g._x_seq.size(); ++k) name[&g._x_seq[(size_t)k]] = "x" + std::to_string(k);
    for (int k = 0; k < (int)g._u_seq.size(); ++k) name[&g._u_seq[(size_t)k]] = "u" + std::to_string(k);
    name[&g._xf] = "xf"; name[&g._dt] = "dt"; name[&g._u_prev] = "u_prev"; name[&g._u_prev_dt] = "u_prev_dt"; name[&g._u_ref] = "u_ref";
    std::ostringstream o;
    auto dump = [&](const char* set, const std::vector<corbo::BaseEdge::Ptr>& list) {
        for (const auto& e : list) {
            o << set << "|" << e->kind << "|" << e->k << "|";
            for (size_t i = 0; i < e->vertices.size(); ++i) { auto it = name.find(e->vertices[i]); o << (i ? "," : "") << (it == name.end() ? "?" : it->second); }
            o << "\n";
        }
    };
    dump("objective", edges.objective); dump("equality", edges.equalities); dump("inequality", edges.inequalities);
    const std::string s = o.str();
    std::strncpy(out, s.c_str(), (size_t)cap - 1); out[cap - 1] = 0;
    return (int)s.size();
}
}  // extern "C"
