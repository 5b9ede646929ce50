// ORACLE (test infrastructure only).  NOT control-box-rst: the edge classes createEdges (src/optimal_control/finite_differences_grid_se2.cpp) instantiates, as RECORDS of
// their constructor arguments (kind, grid point, vertices); compiled into oracle/_ref/libmpc_ref_edges.so only (oracle/ref_wrap_edges.cpp).
#pragma once
#include <corbo-numerics/finite_differences_collocation.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/discretization_grid_interface.h>
namespace corbo {
#define MPC_STUB_EDGE3(NAME)                                                                                                                      \
    class NAME : public BaseEdge {                                                                                                                \
     public:                                                                                                                                      \
        using Ptr = std::shared_ptr<NAME>;                                                                                                         \
        NAME(VectorVertex& x_k, VectorVertex& u_k, ScalarVertex& dt, StageFunctionHandle::Ptr, int k) : BaseEdge(#NAME, k, {&x_k, &u_k, &dt}) {}   \
    };
#define MPC_STUB_EDGE4(NAME)                                                                                                                                        \
    class NAME : public BaseEdge {                                                                                                                                  \
     public:                                                                                                                                                        \
        using Ptr = std::shared_ptr<NAME>;                                                                                                                           \
        NAME(VectorVertex& x_k, VectorVertex& u_k, VectorVertex& x_next, ScalarVertex& dt, StageFunctionHandle::Ptr, int k) : BaseEdge(#NAME, k, {&x_k, &u_k, &x_next, &dt}) {} \
    };
MPC_STUB_EDGE3(LeftSumCostEdge) MPC_STUB_EDGE3(LeftSumEqualityEdge) MPC_STUB_EDGE3(LeftSumInequalityEdge)
MPC_STUB_EDGE4(TrapezoidalIntegralCostEdge) MPC_STUB_EDGE4(TrapezoidalIntegralInequalityEdge)
#undef MPC_STUB_EDGE3
#undef MPC_STUB_EDGE4
class FDCollocationEdge : public BaseEdge {
 public:
    using Ptr = std::shared_ptr<FDCollocationEdge>;
    FDCollocationEdge(SystemDynamicsInterface::Ptr, VectorVertex& x_k, VectorVertex& u_k, VectorVertex& x_next, ScalarVertex& dt) : BaseEdge("FDCollocationEdge", -1, {&x_k, &u_k, &x_next, &dt}) {}
    void setFiniteDifferencesCollocationMethod(FiniteDifferencesCollocationInterface::Ptr m) { fd = m; }
    FiniteDifferencesCollocationInterface::Ptr fd;
};
class TrapezoidalIntegralEqualityDynamicsEdge : public BaseEdge {
 public:
    using Ptr = std::shared_ptr<TrapezoidalIntegralEqualityDynamicsEdge>;
    TrapezoidalIntegralEqualityDynamicsEdge(SystemDynamicsInterface::Ptr, VectorVertex& x_k, VectorVertex& u_k, VectorVertex& x_next, ScalarVertex& dt, StageFunctionHandle::Ptr, int k)
        : BaseEdge("TrapezoidalIntegralEqualityDynamicsEdge", k, {&x_k, &u_k, &x_next, &dt}) {}
    void setFiniteDifferencesCollocationMethod(FiniteDifferencesCollocationInterface::Ptr m) { fd = m; }
    FiniteDifferencesCollocationInterface::Ptr fd;
};
}  // namespace corbo
