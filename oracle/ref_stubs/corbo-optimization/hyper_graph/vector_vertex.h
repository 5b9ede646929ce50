// ORACLE (test infrastructure only): corbo's vertex classes reduced to the INTERFACE the reference's own vertex header
// (include/mpc_local_planner/optimal_control/vector_vertex_se2.h, compiled and executed as it is) overrides, plus what the reference's grid
// (full_discretization_grid_base_se2.cpp) does with vertices: values, bounds, a fixed flag.  No hyper-graph.  Signatures follow control_box_rst's
// corbo-optimization/hyper_graph/vector_vertex.h as far as the reference's overrides pin them down (every `override` there has to find its virtual here).
#pragma once
#include <corbo-core/types.h>
namespace corbo {
class VertexInterface { public: virtual ~VertexInterface() = default; };
class VectorVertex : public VertexInterface {
 public:
    VectorVertex() = default;
    explicit VectorVertex(bool fixed) : _fixed_all(fixed) {}
    explicit VectorVertex(int dimension, bool fixed = false) : _values(dimension), _lb(dimension), _ub(dimension), _fixed_all(fixed) {
        for (int i = 0; i < dimension; ++i) { _lb[i] = -CORBO_INF_DBL; _ub[i] = CORBO_INF_DBL; }
    }
    explicit VectorVertex(const Eigen::Ref<const Eigen::VectorXd>& values, bool fixed = false) : _values(values), _lb(values.size()), _ub(values.size()), _fixed_all(fixed) {
        for (int i = 0; i < values.size(); ++i) { _lb[i] = -CORBO_INF_DBL; _ub[i] = CORBO_INF_DBL; }
    }
    VectorVertex(const Eigen::Ref<const Eigen::VectorXd>& values, const Eigen::Ref<const Eigen::VectorXd>& lb, const Eigen::Ref<const Eigen::VectorXd>& ub, bool fixed = false)
        : _values(values), _lb(lb), _ub(ub), _fixed_all(fixed) {}
    virtual int getDimension() const { return _values.size(); }
    virtual int getDimensionUnfixed() const { return _fixed_all ? 0 : getDimension(); }
    virtual void setDimension(int dim) { _values = Eigen::VectorXd(dim); _lb = Eigen::VectorXd(dim); _ub = Eigen::VectorXd(dim); for (int i = 0; i < dim; ++i) { _lb[i] = -CORBO_INF_DBL; _ub[i] = CORBO_INF_DBL; } }
    virtual void plus(int idx, double inc) { _values[idx] += inc; }
    virtual void plus(const double* inc) { for (int i = 0; i < getDimension(); ++i) _values[i] += inc[i]; }
    virtual void plusUnfixed(const double* inc) { plus(inc); }
    virtual void setData(int idx, double data) { _values[idx] = data; }
    virtual void set(const Eigen::Ref<const Eigen::VectorXd>& values, const Eigen::Ref<const Eigen::VectorXd>& lb, const Eigen::Ref<const Eigen::VectorXd>& ub, bool fixed = false) {
        _values = Eigen::VectorXd(values); setLowerBounds(lb); setUpperBounds(ub); setFixed(fixed);
    }
    virtual bool hasFixedComponents() const { return _fixed_all; }
    virtual bool isFixedComponent(int) const { return _fixed_all; }
    virtual int getNumberFiniteLowerBounds(bool unfixed_only) const { return (unfixed_only && _fixed_all) ? 0 : (_lb.array() > -CORBO_INF_DBL).count(); }
    virtual int getNumberFiniteUpperBounds(bool unfixed_only) const { return (unfixed_only && _fixed_all) ? 0 : (_ub.array() < CORBO_INF_DBL).count(); }
    virtual int getNumberFiniteBounds(bool unfixed_only) const { return (unfixed_only && _fixed_all) ? 0 : (_ub.array() < CORBO_INF_DBL || _lb.array() > -CORBO_INF_DBL).count(); }
    Eigen::VectorXd& values() { return _values; }
    const Eigen::VectorXd& values() const { return _values; }
    const Eigen::VectorXd& lowerBound() const { return _lb; }
    const Eigen::VectorXd& upperBound() const { return _ub; }
    void setLowerBounds(const Eigen::Ref<const Eigen::VectorXd>& lb) { _lb = Eigen::VectorXd(lb); }
    void setUpperBounds(const Eigen::Ref<const Eigen::VectorXd>& ub) { _ub = Eigen::VectorXd(ub); }
    virtual void setFixed(bool fixed) { _fixed_all = fixed; }
    // corbo: a vertex is fixed when it has no unfixed component (VertexInterface::isFixed); what makes the reference's partially fixed vertex report "fixed" once
    // all of its components are (it overrides getDimensionUnfixed, not isFixed).  An empty vertex keeps its flag.
    virtual bool isFixed() const { return getDimension() > 0 ? getDimensionUnfixed() == 0 : _fixed_all; }
    void clear() { _values = Eigen::VectorXd(); _lb = Eigen::VectorXd(); _ub = Eigen::VectorXd(); }
 protected:
    Eigen::VectorXd _values, _lb, _ub;
    bool _fixed_all = false;
};
using PartiallyFixedVectorVertex = VectorVertex;
}  // namespace corbo
