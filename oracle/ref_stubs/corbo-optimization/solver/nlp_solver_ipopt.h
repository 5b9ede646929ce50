// ORACLE (test infrastructure only).  NOT control-box-rst: the solver objects the reference's Controller::configureSolver creates, reduced to records of what was set
#pragma once
#include <map>
#include <memory>
#include <string>
namespace corbo {
class NlpSolverInterface {
 public:
    using Ptr = std::shared_ptr<NlpSolverInterface>;
    virtual ~NlpSolverInterface() = default;
    virtual bool isLsqSolver() const = 0;
    virtual bool initialize() { return true; }
};
class SolverIpopt : public NlpSolverInterface {
 public:
    using Ptr = std::shared_ptr<SolverIpopt>;
    bool isLsqSolver() const override { return false; }
    void setIterations(int n) { iterations = n; }
    void setMaxCpuTime(double t) { max_cpu_time = t; }
    bool setIpoptOptionNumeric(const std::string& k, double v) { numeric[k] = v; return true; }
    bool setIpoptOptionString(const std::string& k, const std::string& v) { strings[k] = v; return true; }
    bool setIpoptOptionInt(const std::string& k, int v) { integers[k] = v; return true; }
    int iterations = -1;
    double max_cpu_time = 0;
    std::map<std::string, double> numeric; std::map<std::string, std::string> strings; std::map<std::string, int> integers;
};
}  // namespace corbo
