// ORACLE (test infrastructure only).  NOT ROS: what the reference's src/controller.cpp touches of roscpp -- a parameter lookup (ros::NodeHandle::param), time stamps,
// a publisher that keeps the last message, console macros that append to a log -- so that Controller::configure / step / isPoseTrajectoryFeasible can be compiled
// and executed here (oracle/ref_wrap_controller.cpp).  The parameter store is a flat map "a/b/c" -> typed value; conversions as roscpp's param.cpp does them:
// a double parameter accepts an int, an int parameter accepts a double (rounded), numeric lists convert element-wise, everything else must match or the
// default is kept.
#pragma once
#include <boost/shared_ptr.hpp>
#include <cmath>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace ros {
struct StubLog { std::vector<std::pair<int, std::string>> lines; };       // level 1 info, 2 warn, 3 error
inline StubLog& stub_log() { static StubLog l; return l; }
inline void stub_log_add(int level, const std::string& s) { stub_log().lines.emplace_back(level, s); }

struct ParamValue {
    enum Kind { Int, Double, Bool, String, NumList, BoolList, NumMap, StrMap, ListOfLists } kind = Int;
    long i = 0; double d = 0; bool b = false; std::string s;
    std::vector<double> nums; std::vector<bool> num_is_int; std::vector<bool> bools;
    std::vector<std::vector<double>> lists; std::vector<std::vector<int>> lists_kind;      // ListOfLists: element kind 0 int, 1 double, 2 text (footprint_model/vertices)
    std::map<std::string, double> num_map; std::map<std::string, bool> num_map_is_int; std::map<std::string, std::string> str_map;
};
using ParamStore = std::map<std::string, ParamValue>;
inline ParamStore*& stub_default_store() { static ParamStore* s = nullptr; return s; }       // what NodeHandle("~/name") reads (the plugin's own namespace)
}  // namespace ros

// the part of XmlRpc::XmlRpcValue the reference's footprint parsing uses (src/mpc_local_planner_ros.cpp:1003-1095)
namespace XmlRpc {
class XmlRpcValue {
 public:
    enum Type { TypeInvalid, TypeBoolean, TypeInt, TypeDouble, TypeString, TypeDateTime, TypeBase64, TypeArray, TypeStruct };
    XmlRpcValue() = default;
    Type getType() const { return _type; }
    int size() const { return (int)_array.size(); }
    XmlRpcValue& operator[](int i) { return _array[(size_t)i]; }
    operator int&() { return _i; }
    operator double&() { return _d; }
    operator std::string&() { return _s; }
    static XmlRpcValue of_int(int v) { XmlRpcValue x; x._type = TypeInt; x._i = v; x._s = std::to_string(v); return x; }
    static XmlRpcValue of_double(double v) { XmlRpcValue x; x._type = TypeDouble; x._d = v; x._s = std::to_string(v); return x; }
    static XmlRpcValue of_string(const std::string& v) { XmlRpcValue x; x._type = TypeString; x._s = v; return x; }
    static XmlRpcValue of_array(const std::vector<XmlRpcValue>& v) { XmlRpcValue x; x._type = TypeArray; x._array = v; x._s = "<array>"; return x; }
 private:
    Type _type = TypeInvalid;
    int _i = 0; double _d = 0; std::string _s;
    std::vector<XmlRpcValue> _array;
};
}  // namespace XmlRpc

namespace ros {

class Duration { public: explicit Duration(double s = 0) : _s(s) {} double toSec() const { return _s; } private: double _s; };
class Time {
 public:
    Time() = default;
    explicit Time(double s) : _s(s) {}
    double toSec() const { return _s; }
    static Time now() { return Time(0.0); }
    Duration operator-(const Time& o) const { return Duration(_s - o._s); }
 private:
    double _s = 0;
};

class Subscriber {};
class Publisher {
 public:
    std::shared_ptr<std::function<void(const void*)>> sink;
    template <class M> void publish(const M& msg) const { if (sink && *sink) (*sink)(&msg); }
};

class Rate { public: explicit Rate(double) {} };
class NodeHandle {
 public:
    ParamStore* store = nullptr;
    std::string prefix;                                    // keys are looked up as prefix + key
    NodeHandle() = default;
    explicit NodeHandle(const std::string& ns) : store(stub_default_store()), prefix(ns == "~" ? "~/" : "") {}       // "~/<plugin name>": the store IS that namespace; "~": move_base's
    NodeHandle(const NodeHandle& parent, const std::string& ns) : store(parent.store), prefix(parent.prefix + ns + "/"), publish_sink(parent.publish_sink) {}
    std::string getNamespace() const { return "/move_base/MpcLocalPlannerROS"; }
    bool hasParam(const std::string& key) const {
        if (!store) return false;
        const std::string k = prefix + key;
        if (store->count(k)) return true;
        auto it = store->lower_bound(k + "/");
        return it != store->end() && it->first.compare(0, k.size() + 1, k + "/") == 0;
    }
    template <class T> bool getParam(const std::string& key, T& v) const { return get(key, v); }
    bool get(const std::string& k, XmlRpc::XmlRpcValue& v) const {
        const ParamValue* p = find(k); if (!p) return false;
        using X = XmlRpc::XmlRpcValue;
        auto num = [](double d, bool is_int) { return is_int ? X::of_int((int)d) : X::of_double(d); };
        switch (p->kind) {
            case ParamValue::Int: v = X::of_int((int)p->i); return true;
            case ParamValue::Double: v = X::of_double(p->d); return true;
            case ParamValue::String: v = X::of_string(p->s); return true;
            case ParamValue::NumList: { std::vector<X> a; for (size_t i = 0; i < p->nums.size(); ++i) a.push_back(num(p->nums[i], p->num_is_int[i])); v = X::of_array(a); return true; }
            case ParamValue::ListOfLists: {
                std::vector<X> a;
                for (size_t i = 0; i < p->lists.size(); ++i) {
                    std::vector<X> b;
                    for (size_t j = 0; j < p->lists[i].size(); ++j) b.push_back(p->lists_kind[i][j] == 2 ? X::of_string("text") : num(p->lists[i][j], p->lists_kind[i][j] == 0));
                    a.push_back(X::of_array(b));
                }
                v = X::of_array(a); return true;
            }
            default: return false;
        }
    }
    std::shared_ptr<std::function<void(const void*)>> publish_sink = std::make_shared<std::function<void(const void*)>>();
    const ParamValue* find(const std::string& key) const { if (!store) return nullptr; auto it = store->find(prefix + key); return it == store->end() ? nullptr : &it->second; }
    bool get(const std::string& k, int& v) const {
        const ParamValue* p = find(k); if (!p) return false;
        if (p->kind == ParamValue::Int) { v = (int)p->i; return true; }
        if (p->kind == ParamValue::Double) { double d = p->d; d = std::fmod(d, 1.0) < 0.5 ? std::floor(d) : std::ceil(d); v = (int)d; return true; }
        return false;
    }
    bool get(const std::string& k, double& v) const {
        const ParamValue* p = find(k); if (!p) return false;
        if (p->kind == ParamValue::Double) { v = p->d; return true; }
        if (p->kind == ParamValue::Int) { v = (double)p->i; return true; }
        return false;
    }
    bool get(const std::string& k, bool& v) const { const ParamValue* p = find(k); if (!p || p->kind != ParamValue::Bool) return false; v = p->b; return true; }
    bool get(const std::string& k, std::string& v) const { const ParamValue* p = find(k); if (!p || p->kind != ParamValue::String) return false; v = p->s; return true; }
    bool get(const std::string& k, std::vector<double>& v) const {
        const ParamValue* p = find(k); if (!p) return false;
        if (p->kind == ParamValue::NumList) { v = p->nums; return true; }
        if (p->kind == ParamValue::BoolList) { v.clear(); for (bool b : p->bools) v.push_back(b ? 1.0 : 0.0); return true; }
        return false;
    }
    bool get(const std::string& k, std::vector<bool>& v) const {
        const ParamValue* p = find(k); if (!p) return false;
        if (p->kind == ParamValue::BoolList) { v = p->bools; return true; }
        if (p->kind == ParamValue::NumList) { v.clear(); for (double d : p->nums) v.push_back(d != 0.0); return true; }
        return false;
    }
    bool get(const std::string& k, std::map<std::string, double>& v) const { const ParamValue* p = find(k); if (!p || p->kind != ParamValue::NumMap) return false; v = p->num_map; return true; }
    bool get(const std::string& k, std::map<std::string, int>& v) const {
        const ParamValue* p = find(k); if (!p || p->kind != ParamValue::NumMap) return false;
        for (const auto& e : p->num_map) if (!p->num_map_is_int.at(e.first)) return false;        // a map<string, int> does not take doubles
        v.clear(); for (const auto& e : p->num_map) v[e.first] = (int)e.second; return true;
    }
    bool get(const std::string& k, std::map<std::string, std::string>& v) const { const ParamValue* p = find(k); if (!p || p->kind != ParamValue::StrMap) return false; v = p->str_map; return true; }
    template <class T> bool param(const std::string& key, T& out, const T& fallback) const { if (get(key, out)) return true; out = fallback; return false; }
    template <class M, class C> Subscriber subscribe(const std::string&, int, void (C::*)(const typename M::ConstPtr&), C*) const { return Subscriber(); }
    template <class C, class A> Subscriber subscribe(const std::string&, int, void (C::*)(A), C*) const { return Subscriber(); }
    template <class M> Publisher advertise(const std::string&, int) const { Publisher p; p.sink = publish_sink; return p; }
};
}  // namespace ros

#define MPC_STUB_ROS_STREAM(level, args) do { std::ostringstream mpc_stub_ss; mpc_stub_ss << args; ::ros::stub_log_add(level, mpc_stub_ss.str()); } while (0)
#define ROS_INFO(...) ::ros::stub_log_add(1, #__VA_ARGS__)
#define ROS_INFO_ONCE(...) ::ros::stub_log_add(1, #__VA_ARGS__)
#define ROS_DEBUG(...) do { } while (0)
#define ROS_FATAL(...) ::ros::stub_log_add(4, #__VA_ARGS__)
#define ROS_WARN_COND(cond, ...) do { if (cond) ::ros::stub_log_add(2, #__VA_ARGS__); } while (0)
#define ROS_WARN(...) ::ros::stub_log_add(2, #__VA_ARGS__)
#define ROS_ERROR(...) ::ros::stub_log_add(3, #__VA_ARGS__)
#define ROS_INFO_STREAM(args) MPC_STUB_ROS_STREAM(1, args)
#define ROS_WARN_STREAM(args) MPC_STUB_ROS_STREAM(2, args)
#define ROS_ERROR_STREAM(args) MPC_STUB_ROS_STREAM(3, args)
#define ROS_INFO_STREAM_COND(cond, args) do { if (cond) MPC_STUB_ROS_STREAM(1, args); } while (0)
