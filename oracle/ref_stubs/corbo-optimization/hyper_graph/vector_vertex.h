// ORACLE (test infrastructure only): corbo's vertex classes reduced to what the reference's grid (full_discretization_grid_base_se2.cpp) does with them:
// values, bounds, a fixed flag.  No hyper-graph.
#pragma once
#include <corbo-core/types.h>
namespace corbo {
class VertexInterface { public: virtual ~VertexInterface() = default; };
class VectorVertex : public VertexInterface {
 public:
    VectorVertex() = default;
    explicit VectorVertex(bool fixed) : _fixed_all(fixed) {}
    explicit VectorVertex(const Eigen::Ref<const Eigen::VectorXd>& values, bool fixed = false) : _values(values), _fixed_all(fixed) {}
    VectorVertex(const Eigen::Ref<const Eigen::VectorXd>& values, const Eigen::Ref<const Eigen::VectorXd>& lb, const Eigen::Ref<const Eigen::VectorXd>& ub, bool fixed = false)
        : _values(values), _lb(lb), _ub(ub), _fixed_all(fixed) {}
    int getDimension() const { return _values.size(); }
    Eigen::VectorXd& values() { return _values; }
    const Eigen::VectorXd& values() const { return _values; }
    void setLowerBounds(const Eigen::Ref<const Eigen::VectorXd>& lb) { _lb = Eigen::VectorXd(lb); }
    void setUpperBounds(const Eigen::Ref<const Eigen::VectorXd>& ub) { _ub = Eigen::VectorXd(ub); }
    virtual void setFixed(bool fixed) { _fixed_all = fixed; }
    virtual bool isFixed() const { return _fixed_all; }
    void clear() { _values = Eigen::VectorXd(); _lb = Eigen::VectorXd(); _ub = Eigen::VectorXd(); }
 protected:
    Eigen::VectorXd _values, _lb, _ub;
    bool _fixed_all = false;
};
using PartiallyFixedVectorVertex = VectorVertex;
}  // namespace corbo
