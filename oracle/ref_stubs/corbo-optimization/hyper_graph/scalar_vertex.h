// ORACLE (test infrastructure only): corbo::ScalarVertex as the reference's grid uses it for dt (value, bounds, fixed flag).
#pragma once
#include <corbo-optimization/hyper_graph/vector_vertex.h>
namespace corbo {
class ScalarVertex : public VertexInterface {
 public:
    double& value() { return _value; }
    const double& value() const { return _value; }
    void set(double v, double lb, double ub, bool fixed) { _value = v; _lb = lb; _ub = ub; _fixed = fixed; }
    void setLowerBound(double lb) { _lb = lb; }
    void setUpperBound(double ub) { _ub = ub; }
    bool isFixed() const { return _fixed; }
 private:
    double _value = 0, _lb = -CORBO_INF_DBL, _ub = CORBO_INF_DBL;
    bool _fixed = false;
};
}  // namespace corbo
