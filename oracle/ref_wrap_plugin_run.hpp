// ORACLE (test infrastructure only): the reference's whole plugin, cycle by cycle -- initialize() (parameters, footprint model, Controller::configure, ...), setPlan(),
// computeVelocityCommands() -- on an array-backed costmap, an identity tf buffer and the odometry the test provides.  Included by oracle/ref_wrap_plugin.cpp (the plugin with
// the reference's own Controller, a stand-in solver plugged into it; entry points ref_plugin_*) and by oracle/ref_wrap_plugin_on_binding.cpp (the SAME plugin source built on
// include/mpc_reference_binding.hpp; entry points amd_plugin_*).  The including file defines PLUGIN_ENTRY(name) and a struct SolverPort { void attach(MpcLocalPlannerROS&);
// void set(solve_cb); int last_guess(int cap, double* x, double* u, double* dt); void begin_cycle(); int guess_n(); }.
#pragma once
#include <sstream>
namespace plugin_run {
using mpc_local_planner::MpcLocalPlannerROS;
typedef int (*solve_cb)(int n, double* x, double* u, double* dt, const double* u_prev, double u_prev_dt);
inline std::vector<std::string> split(const std::string& s, char c) { std::vector<std::string> out; std::stringstream ss(s); std::string item; while (std::getline(ss, item, c)) out.push_back(item); return out; }
// as oracle/ref_wrap_controller.cpp::parse_params, plus "ll": a list of lists "i:1|d:2.5;d:0|s:x" (footprint_model/vertices)
inline void parse_params_plugin(const char* text, ros::ParamStore& store) {
    for (const std::string& line : split(text, '\n')) {
        const auto f = split(line, '\t');
        if (f.size() < 2) continue;
        const std::string val = f.size() > 2 ? f[2] : "";
        ros::ParamValue p;
        if (f[1] == "i") { p.kind = ros::ParamValue::Int; p.i = std::stol(val); }
        else if (f[1] == "d") { p.kind = ros::ParamValue::Double; p.d = std::stod(val); }
        else if (f[1] == "b") { p.kind = ros::ParamValue::Bool; p.b = val == "1"; }
        else if (f[1] == "s") { p.kind = ros::ParamValue::String; p.s = val; }
        else if (f[1] == "nl") { p.kind = ros::ParamValue::NumList; for (const auto& e : split(val, ',')) { p.num_is_int.push_back(e[0] == 'i'); p.nums.push_back(std::stod(e.substr(2))); } }
        else if (f[1] == "bl") { p.kind = ros::ParamValue::BoolList; for (const auto& e : split(val, ',')) p.bools.push_back(e == "1"); }
        else if (f[1] == "nm") { p.kind = ros::ParamValue::NumMap; for (const auto& e : split(val, ',')) { const auto kv = split(e, ':'); p.num_map[kv[0]] = std::stod(kv[2]); p.num_map_is_int[kv[0]] = kv[1] == "i"; } }
        else if (f[1] == "sm") { p.kind = ros::ParamValue::StrMap; for (const auto& e : split(val, ',')) { const auto kv = split(e, ':'); p.str_map[kv[0]] = kv.size() > 1 ? kv[1] : ""; } }
        else if (f[1] == "ll") {
            p.kind = ros::ParamValue::ListOfLists;
            for (const auto& row : split(val, ';')) {
                std::vector<double> r; std::vector<int> k;
                for (const auto& e : split(row, '|')) { k.push_back(e[0] == 'i' ? 0 : e[0] == 'd' ? 1 : 2); r.push_back(e[0] == 's' ? 0.0 : std::stod(e.substr(2))); }
                p.lists.push_back(r); p.lists_kind.push_back(k);
            }
        } else continue;
        store[f[0]] = p;
    }
}
}  // namespace plugin_run
