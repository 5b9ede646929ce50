// ORACLE (test infrastructure only): the part of corbo::ReferenceTrajectoryInterface that the reference's grid reads -- values per grid point
// ("cached" for the grid's dt and n), dimension, static or not.  One concrete kind: a table of values (oracle/ref_wrap_grid.cpp fills it).
#pragma once
#include <corbo-core/types.h>
namespace corbo {
class Time { public: explicit Time(double t = 0) : _t(t) {} double toSec() const { return _t; } private: double _t; };
class ReferenceTrajectoryInterface {
 public:
    virtual ~ReferenceTrajectoryInterface() = default;
    std::vector<Eigen::VectorXd> table;        // value at grid point k (the last one holds beyond the end)
    bool is_static = false;
    int dim = 0;
    int getDimension() const { return dim; }
    bool isStatic() const { return is_static; }
    bool isCached(double, int, const Time&) const { return true; }
    void precompute(double, int, const Time&) {}
    const Eigen::VectorXd& getReferenceCached(int k) const { return table[(size_t)(k < (int)table.size() ? k : (int)table.size() - 1)]; }
};
}  // namespace corbo
