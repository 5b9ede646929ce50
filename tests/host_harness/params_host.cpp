// tests only: C entry point around include/mpc_params.hpp (calls mpc_config_defaults of the product library; no GPU calls)
// include/mpc_params.hpp: the parameter set as lines "key\tkind\tvalue" (kind b i d s dv bv pv; lists comma separated, points "x,y;x,y") ->
// mpc_config, the facade options (as doubles, in declaration order) and the report text ("error\nnote\nnote...")
#include "../../include/mpc_params.hpp"
#include <cstring>
#include <sstream>
extern "C" int ctl_config_from_params(const char* text, const char* costmap_fp, mpc_config* cfg, double* opt, char* report, int report_cap) {
    using namespace mpc_local_planner_amd;
    MapParamSource src;
    auto split = [](const std::string& s, char sep) { std::vector<std::string> out; std::string cur; std::istringstream is(s); while (std::getline(is, cur, sep)) out.push_back(cur); return out; };
    auto points = [&](const std::string& s) { std::vector<std::vector<double>> pv; for (auto& q : split(s, ';')) { std::vector<double> pt; for (auto& x : split(q, ',')) pt.push_back(std::stod(x)); pv.push_back(pt); } return pv; };
    for (auto& line : split(text, '\n')) {
        auto f = split(line, '\t');
        if (f.size() < 2) continue;
        const std::string val = f.size() > 2 ? f[2] : "";
        if (f[1] == "b") src.set(f[0], val == "1");
        else if (f[1] == "i") src.set(f[0], std::stoi(val));
        else if (f[1] == "d") src.set(f[0], std::stod(val));
        else if (f[1] == "s") src.set(f[0], val);
        else if (f[1] == "dv") { std::vector<double> v; for (auto& x : split(val, ',')) v.push_back(std::stod(x)); src.set(f[0], v); }
        else if (f[1] == "bv") { std::vector<bool> v; for (auto& x : split(val, ',')) v.push_back(x == "1"); src.set(f[0], v); }
        else if (f[1] == "pv") src.set(f[0], points(val));
    }
    std::vector<std::vector<double>> cfp;
    if (costmap_fp && costmap_fp[0]) cfp = points(costmap_fp);
    ControllerOptions o; ParamReport rep;
    const ParamStatus st = config_from_params(src, *cfg, o, rep, cfp.empty() ? nullptr : &cfp);
    const double ov[] = {(double)o.grid_adaptation, (double)o.max_grid_size, o.dt_hyst_ratio, (double)o.min_grid_size, (double)o.n_max, (double)o.warm_start,
                         (double)o.outer_ocp_iterations, o.force_reinit_new_goal_dist, o.force_reinit_new_goal_angular, (double)o.allow_init_with_backward_motion,
                         (double)o.force_reinit_num_steps, (double)o.prefer_x_feedback, (double)o.publish_ocp_results, (double)o.print_cpu_time};
    for (size_t i = 0; i < sizeof(ov) / sizeof(ov[0]); ++i) opt[i] = ov[i];
    std::string r = rep.error;
    for (auto& s : rep.notes) r += "\n" + s;
    std::strncpy(report, r.c_str(), (size_t)report_cap - 1);
    report[report_cap - 1] = 0;
    return (int)st;
}

// plugin-level parameters (include/mpc_params.hpp::plugin_options_from_params): text / move_base_text as above; out [14] in the order of mpc_local_planner_amd/params.py::plugin_options_from_params (its key order),
// strings "odom_topic\ncostmap_converter_plugin"
extern "C" void ctl_plugin_options(const char* text, const char* move_base_text, double* out, char* strings, int cap) {
    using namespace mpc_local_planner_amd;
    auto split = [](const std::string& s, char sep) { std::vector<std::string> o; std::string cur; std::istringstream is(s); while (std::getline(is, cur, sep)) o.push_back(cur); return o; };
    auto fill = [&](const char* t, MapParamSource& src) {
        for (auto& line : split(t, '\n')) {
            auto f = split(line, '\t');
            if (f.size() < 2) continue;
            const std::string val = f.size() > 2 ? f[2] : "";
            if (f[1] == "b") src.set(f[0], val == "1");
            else if (f[1] == "i") src.set(f[0], std::stoi(val));
            else if (f[1] == "d") src.set(f[0], std::stod(val));
            else if (f[1] == "s") src.set(f[0], val);
        }
    };
    MapParamSource a, b;
    fill(text, a); fill(move_base_text, b);
    const PluginOptions o = plugin_options_from_params(a, &b);
    const double v[14] = {o.xy_goal_tolerance, o.yaw_goal_tolerance, (double)o.global_plan_overwrite_orientation, o.global_plan_prune_distance, o.max_global_plan_lookahead_dist,
                          (double)o.is_footprint_dynamic, (double)o.include_costmap_obstacles, o.costmap_obstacles_behind_robot_dist, o.global_plan_viapoint_sep,
                          o.collision_check_min_resolution_angular, (double)o.collision_check_no_poses, o.controller_frequency, o.costmap_converter_rate, (double)o.costmap_converter_spin_thread};
    for (int i = 0; i < 14; ++i) out[i] = v[i];
    const std::string s = o.odom_topic + "\n" + o.costmap_converter_plugin;
    std::strncpy(strings, s.c_str(), (size_t)cap - 1); strings[cap - 1] = 0;
}
