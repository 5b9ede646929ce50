// ORACLE (test infrastructure only).  This file SHADOWS the reference's vector_vertex_se2.h: the real one is written against corbo's hyper-graph vertex
// API and a good part of Eigen (bool arrays, maps); the reference's GRID code -- the thing compiled and executed here,
// src/optimal_control/full_discretization_grid_base_se2.cpp -- only stores values in these vertices.  Kept from the real header: set() wraps the heading
// into [-pi, pi) (vector_vertex_se2.h:104-117, :186-213) and the partially fixed vertex carries per-component flags.
#pragma once
#include <corbo-optimization/hyper_graph/vector_vertex.h>
#include <mpc_local_planner/utils/math_utils.h>
namespace mpc_local_planner {
class VectorVertexSE2 : public corbo::VectorVertex {
 public:
    VectorVertexSE2() = default;
    explicit VectorVertexSE2(const Eigen::Ref<const Eigen::VectorXd>& values, bool fixed = false) : corbo::VectorVertex(values, fixed) {}
    VectorVertexSE2(const Eigen::Ref<const Eigen::VectorXd>& values, const Eigen::Ref<const Eigen::VectorXd>& lb, const Eigen::Ref<const Eigen::VectorXd>& ub, bool fixed = false)
        : corbo::VectorVertex(values, lb, ub, fixed) {}
};
class PartiallyFixedVectorVertexSE2 : public VectorVertexSE2 {
 public:
    PartiallyFixedVectorVertexSE2() = default;
    void set(const Eigen::Ref<const Eigen::VectorXd>& values, const Eigen::Ref<const Eigen::VectorXd>& lb, const Eigen::Ref<const Eigen::VectorXd>& ub,
             const Eigen::Matrix<bool, -1, 1>& fixed) {
        _values = Eigen::VectorXd(values);
        _values[2] = normalize_theta(_values[2]);
        setLowerBounds(lb); setUpperBounds(ub);
        _fixed = fixed;
    }
    void setFixed(bool fixed) override { _fixed.setConstant(_values.size(), fixed); }
    bool isFixed() const override { return _fixed.size() > 0 && _fixed.count() == _fixed.size(); }
    bool isFixedComponent(int i) const { return _fixed[i]; }
 private:
    Eigen::Matrix<bool, -1, 1> _fixed;
};
}  // namespace mpc_local_planner
