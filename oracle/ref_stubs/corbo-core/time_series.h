// ORACLE (test infrastructure only).  NOT control-box-rst: the storage part of corbo::TimeSeries that the reference's TimeSeriesSE2
// (src/utils/time_series_se2.cpp) and its grid (getStateAndControlTimeSeries) touch -- time stamps, one value vector per stamp, the two enums.
#pragma once
#include <Eigen/Core>
#include <algorithm>
#include <memory>
#include <vector>

namespace corbo {
class TimeSeries {
 public:
    using Ptr = std::shared_ptr<TimeSeries>;
    using ConstPtr = std::shared_ptr<const TimeSeries>;
    enum class Interpolation { ZeroOrderHold, Linear };
    enum class Extrapolation { NoExtrapolation, ZeroOrderHold };
    // one sample per column
    struct ValuesMatMap {
        const std::vector<Eigen::VectorXd>* cols;
        Eigen::VectorXd col(int i) const { return (*cols)[(size_t)i]; }
        double mean() const { double s = 0; int m = 0; for (const auto& c : *cols) for (int i = 0; i < c.size(); ++i, ++m) s += c[i]; return m ? s / m : 0.0; }
        struct Rowwise {
            const std::vector<Eigen::VectorXd>* cols;
            Eigen::VectorXd mean() const {
                Eigen::VectorXd r(cols->empty() ? 0 : cols->front().size());
                for (const auto& c : *cols) for (int i = 0; i < r.size(); ++i) r[i] += c[i];
                for (int i = 0; i < r.size(); ++i) r[i] /= (double)cols->size();
                return r;
            }
        };
        Rowwise rowwise() const { return Rowwise{cols}; }
    };
    TimeSeries() = default;
    explicit TimeSeries(int value_dim) : _value_dim(value_dim) {}
    virtual ~TimeSeries() = default;
    void clear() { _time.clear(); _x.clear(); }
    void add(double t, const Eigen::Ref<const Eigen::VectorXd>& x) { _time.push_back(t); _x.push_back(Eigen::VectorXd(x)); _value_dim = x.size(); }
    int getTimeDimension() const { return (int)_time.size(); }
    bool isEmpty() const { return _time.empty(); }
    const std::vector<double>& getTime() const { return _time; }
    std::vector<double> getValues() const { std::vector<double> v; for (const auto& c : _x) for (int i = 0; i < c.size(); ++i) v.push_back(c[i]); return v; }      // dim x N, column-major
    int getValueDimension() const { return _value_dim; }
    const Eigen::VectorXd& getValuesMap(int idx) const { return _x[(size_t)idx]; }
    ValuesMatMap getValuesMatrixView() const { return ValuesMatMap{&_x}; }
    virtual bool getValuesInterpolate(double, Eigen::Ref<Eigen::VectorXd>, Interpolation = Interpolation::Linear, Extrapolation = Extrapolation::NoExtrapolation,
                                      double = 1e-6) const { return false; }
    virtual double computeMeanOverall() { return getValuesMatrixView().mean(); }
    virtual void computeMeanCwise(Eigen::Ref<Eigen::VectorXd> mean_values) { mean_values = getValuesMatrixView().rowwise().mean(); }
    const std::vector<double>& times() const { return _time; }
    const std::vector<Eigen::VectorXd>& samples() const { return _x; }
 protected:
    std::vector<double> _time;
    std::vector<Eigen::VectorXd> _x;
    int _value_dim = 0;
};
}  // namespace corbo
