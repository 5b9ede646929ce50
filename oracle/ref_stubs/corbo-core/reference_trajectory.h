// ORACLE (test infrastructure only): the part of corbo::ReferenceTrajectoryInterface that the reference reads -- values per grid point ("cached" for the grid's dt
// and n), dimension, static or not -- with the three kinds the reference creates: a constant (StaticReference), zero (ZeroReference) and a time series sampled at
// t_k = k dt after the time it was anchored at (DiscreteTimeReferenceTrajectory; the sampling itself is TimeSeries::getValuesInterpolate, i.e. for the initial
// state trajectory the reference's own TimeSeriesSE2 code, linear with zero-order hold beyond the end).  A plain table serves the wrappers that supply the values.
#pragma once
#include <corbo-core/time_series.h>
#include <corbo-core/types.h>
namespace corbo {
class Time { public: explicit Time(double t = 0) : _t(t) {} double toSec() const { return _t; } private: double _t; };
class Duration { public: explicit Duration(double t = 0) : _t(t) {} double toSec() const { return _t; } private: double _t; };
class ReferenceTrajectoryInterface {
 public:
    virtual ~ReferenceTrajectoryInterface() = default;
    std::vector<Eigen::VectorXd> table;        // value at grid point k (the last one holds beyond the end)
    bool is_static = false;
    int dim = 0;
    int getDimension() const { return dim; }
    bool isStatic() const { return is_static; }
    virtual bool isCached(double, int, const Time&) const { return true; }
    virtual void precompute(double, int, const Time&) {}
    const Eigen::VectorXd& getReferenceCached(int k) const { return table[(size_t)(k < (int)table.size() ? k : (int)table.size() - 1)]; }
};
class StaticReference : public ReferenceTrajectoryInterface {
 public:
    explicit StaticReference(const Eigen::Ref<const Eigen::VectorXd>& ref) { table.push_back(Eigen::VectorXd(ref)); is_static = true; dim = ref.size(); }
};
class ZeroReference : public ReferenceTrajectoryInterface {
 public:
    explicit ZeroReference(int dimension) { table.push_back(Eigen::VectorXd(dimension)); is_static = true; dim = dimension; }
};
class DiscreteTimeReferenceTrajectory : public ReferenceTrajectoryInterface {
 public:
    void setTrajectory(TimeSeries::Ptr trajectory, TimeSeries::Interpolation interpolation) { _ts = trajectory; _interp = interpolation; dim = trajectory ? trajectory->getValueDimension() : 0; _cached_n = -1; }
    void setTimeFromStart(const Time& t) { _t0 = t.toSec(); }
    bool isCached(double dt, int n, const Time& t) const override { return dt == _cached_dt && n == _cached_n && t.toSec() == _cached_t; }
    void precompute(double dt, int n, const Time& t) override {
        table.clear();
        const double d0 = t.toSec() - _t0;
        for (int k = 0; k < n; ++k) {
            Eigen::VectorXd v(dim);
            if (_ts) _ts->getValuesInterpolate(d0 + (double)k * dt, v, _interp, TimeSeries::Extrapolation::ZeroOrderHold);
            table.push_back(v);
        }
        _cached_dt = dt; _cached_n = n; _cached_t = t.toSec();
        sample_dts.push_back(dt);
    }
    std::vector<double> sample_dts;            // every dt a precompute() was asked for (inspected by the tests)
 private:
    TimeSeries::Ptr _ts;
    TimeSeries::Interpolation _interp = TimeSeries::Interpolation::Linear;
    double _t0 = 0, _cached_dt = -1, _cached_t = 0;
    int _cached_n = -1;
};
}  // namespace corbo
