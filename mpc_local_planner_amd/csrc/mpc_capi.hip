// mpc_capi.hip -- HIP kernels + the C ABI declared in include/mpc_hip.h.
// gfx950 only.  There is no CPU path in this library: mpc_create fails with MPC_ENODEV
// when no HIP device is usable.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/mpc_hip.h"
#include "mpc_core.hpp"
#include "mpc_problem.hpp"
#include "mpc_wave.hpp"
#include "mpc_solve_kernel.hpp"
#include "mpc_costmap.hpp"
#include "mpc_feasibility.hpp"
#include "mpc_grid_update.hpp"

namespace {

thread_local char g_err[512] = "";

void set_err(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}
void set_err(const char* what) { snprintf(g_err, sizeof(g_err), "%s", what); }

#define HIP_TRY(call)                                     \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) { set_err(#call, e_); return MPC_EHIP; } \
    } while (0)

}  // namespace

struct mpc_solver {
    mpc_config cfg;
    mpc::Problem<double> P64;
    mpc::Problem<float> P32;
    mpc::WaveLayout WL;         // everything in LDS
    mpc::WaveLayout WLg;        // factorisation data (GAIN, STG) in global memory (mpc_wave.hpp::GlobalStage): the LDS record is a third
    bool w2_gs;                 // (developer builds) the two-wave kernel in the global form
    bool w2_ok;                 // fp64 launches with more instances than the device has SIMDs may take the two-waves-per-SIMD kernel (global form, <= 256 registers): decided per launch (launch_model)
    size_t wave_lds_w2;         // its dynamic LDS
    int w2_min_batch;           // from this many instances per launch on
    bool gs64, gs32;            // which of the two the fp64 / fp32 launches of this handle use (mpc_config.stage_data; MPC_STAGE_AUTO: the one that puts more workgroups on a CU)
    void* d_gstage;             // n_gslots blocks of factorisation data (WLg.GSW words each), NULL when neither precision uses them
    int* d_gslots;              // [n_gslots] claim words of the blocks (0 = free)
    int n_gslots;
    size_t wave_lds;            // dynamic LDS of the kernel instantiation of cfg.precision (MPC_MIXED: the fp64 one, the larger)
    size_t wave_lds32;          // MPC_MIXED: dynamic LDS of the fp32 phase
    int32_t* d_iters1;          // MPC_MIXED: iterations of the fp32 phase
    int device;
    int max_batch;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    // staging for the host-pointer entry point
    // staging of the host-pointer entry point: ONE pinned host block and ONE device block each way (one H2D and one D2H per call)
    unsigned char *h_in, *h_out, *d_in, *d_out;
    size_t in_cap, out_cap;
    int32_t* d_ngrid;
    int32_t *d_nvia;            // own copy of the via-point counts / poses (mpc_set_via_points) ...
    double *d_via;
    const int32_t* p_nvia;      // ... and what the kernel reads: the own copy or borrowed device pointers
    const double* p_via;
    int use_ngrid;
    int ngrid_B, nvia_B;        // batch sizes the per-instance grid sizes / own via-point copies were set for (solves must not exceed them)
    hipEvent_t cev0, cev1;      // costmap kernel timing (kept apart from the solve kernel's events)
    // candidate initial trajectories (n_candidates > 1): bookkeeping words, candidate records, per-instance winner / total iterations
    int *d_cwin, *d_cexited, *d_citsum;
    double* d_crec;
    int32_t *d_winner, *d_iters_total;
    int32_t* d_rows_dropped;    // per instance: clearance rows that did not fit (solvers with obstacles)
    double* d_dual;             // per instance: multipliers of the last converged solve (dual_warm_start)
    void* d_stage = nullptr;    // device staging of the host-pointer helpers (mpc_costmap_to_obstacles, mpc_check_feasibility): kept across calls, grows on demand
    size_t stage_bytes = 0;
    int dual_words;
    int32_t* last_status;       // device pointers of the most recent solve (mpc_last_candidates without candidates)
    int32_t* last_iters;
    bool timed;
};

// which kernel instantiation serves this solver: the one with the rarely used rows / terms / coupling slots, or the headline one
static bool solver_ext(const mpc_solver* s) {
    const mpc::Problem<double>& P = s->P64;
    return P.ball || P.via || P.integral_form || P.dyn_obst || P.hess_mode || P.costx ||
           (P.n_obst > 0 && (P.footprint_kind == MPC_FOOTPRINT_LINE || P.footprint_kind == MPC_FOOTPRINT_TWO_CIRCLES || P.footprint_kind == MPC_FOOTPRINT_POLYGON));
}

// which XCC ids the device's workgroups run on (bit i of *mask: some workgroup saw HW_REG_XCC_ID & 7 == i): 8 bits on an MI355X, 1 in a partitioned mode
namespace mpc {
__global__ void xcc_probe_kernel(unsigned* mask) {
    if (threadIdx.x == 0) atomicOr(mask, 1u << ((unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u));
}
}  // namespace mpc
// resident workgroups per CU of the solve kernel a launch record selects (per precision and model: the instantiations live in mpc_solve_inst.hip)
static hipError_t pool_kernel_occupancy(bool f32, int model, const mpc::SolveLaunch& a, int* out) {
#ifdef MPC_DEV_ONE_MODEL
    (void)model; return f32 ? hipErrorInvalidConfiguration : mpc::solve_occupancy<double, MPC_DEV_ONE_MODEL>(a, out);
#else
#define MPC_OCC(M) (f32 ? mpc::solve_occupancy<float, M>(a, out) : mpc::solve_occupancy<double, M>(a, out))
    switch (model) {
        case MPC_MODEL_UNICYCLE: return MPC_OCC(mpc::MODEL_UNICYCLE);
        case MPC_MODEL_SIMPLE_CAR: return MPC_OCC(mpc::MODEL_SIMPLE_CAR);
        case MPC_MODEL_SIMPLE_CAR_FRONT: return MPC_OCC(mpc::MODEL_SIMPLE_CAR_FRONT);
        default: return MPC_OCC(mpc::MODEL_KINEMATIC_BICYCLE);
    }
#undef MPC_OCC
#endif
}

// Device staging of the host-pointer helpers: ONE allocation kept in the handle and carved into 256-byte aligned pieces (these calls sit in a B = 1 control
// loop next to a sub-millisecond solve; a hipMalloc / hipFree pair per temporary per call cost more than the kernels they feed).  Grows on demand, freed by mpc_destroy.
static hipError_t stage_carve(mpc_solver* s, const size_t* sz, int count, void** out) {
    size_t total = 0;
    for (int i = 0; i < count; ++i) total += (sz[i] + 255) & ~(size_t)255;
    if (total > s->stage_bytes) {
        if (s->d_stage) { (void)hipStreamSynchronize(s->stream); (void)hipFree(s->d_stage); s->d_stage = nullptr; s->stage_bytes = 0; }
        const hipError_t er = hipMalloc(&s->d_stage, total);
        if (er != hipSuccess) return er;
        s->stage_bytes = total;
    }
    size_t off = 0;
    for (int i = 0; i < count; ++i) { out[i] = (char*)s->d_stage + off; off += (sz[i] + 255) & ~(size_t)255; }
    return hipSuccess;
}

extern "C" {

void mpc_config_defaults(mpc_config* c) {
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->model = MPC_MODEL_UNICYCLE;       // src/controller.cpp:346
    c->model_params[0] = 0.5;            // :355
    c->model_params[1] = 1.0;
    c->n = 20;                           // :274
    c->dt_ref = 0.3;                     // :278
    c->dt_free = 1;                      // :236
    c->dt_lb = 0.0;                      // :242
    c->dt_ub = 10.0;                     // :244
    c->xf_fixed[0] = c->xf_fixed[1] = c->xf_fixed[2] = 1;   // :282
    c->collocation = MPC_COLLOC_FORWARD; // :298
    c->objective = MPC_OBJ_MIN_TIME;     // :551
    c->u_lb[0] = -0.2; c->u_ub[0] = 0.4; // :497-511
    c->u_lb[1] = -0.3; c->u_ub[1] = 0.3;
    for (int j = 0; j < 2; ++j) { c->du_lb[j] = -1e30; c->du_ub[j] = 1e30; }   // :756-770 (0 => inf)
    c->max_iter = 100;                   // :391
    c->tol = 1e-8;
    c->mu_init = 0.1;
    c->precision = MPC_FP64;
    c->min_obstacle_dist = 0.5;          // :717
    c->force_inclusion_dist = 0.5;       // :725
    c->cutoff_dist = 2.0;                // :727
    c->footprint_kind = MPC_FOOTPRINT_POINT;   // src/mpc_local_planner_ros.cpp:894-898
    c->max_obstacles = 0;
    c->max_vertices = 1;
    c->max_obstacle_rows = 4;
}

const char* mpc_last_error(void) { return g_err; }
int32_t mpc_version(void) { return 600; }      // 0.6.0: mpc_config.two_wave_min_batch and mpc_config.line_search took the last reserved words (same size).  0.5.0: mpc_config.stage_data took a reserved word (same size); a solve restores clearance rows that jam.  0.4.0: mpc_config.mu_strategy / max_time_us took reserved words (same size), MPC_TIME_LIMIT; a solve accepts factorisations on their inertia
// history: 0.2.0: mpc_config grew (candidates, kept multipliers, hessian_mode), new entry points; 0.2.1: cost variants (off-diagonal weights, trapezoidal rule, hybrid cost)

#ifdef MPC_PROFILE
// developer build only (-DMPC_PROFILE): per-wave phase cycle counters of the last wave-kernel launch
int mpc_debug_profile(long long* out, int rows) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mpc_prof), sizeof(long long) * 16 * (size_t)rows, 0, hipMemcpyDeviceToHost);
}
#endif

int mpc_create(const mpc_config* cfg, int32_t max_batch, int32_t device, mpc_solver** out) {
    g_err[0] = 0;
    if (!cfg || !out || max_batch <= 0) { set_err("mpc_create: bad argument"); return MPC_EINVAL; }
    *out = nullptr;
    if (cfg->n < 3 || cfg->n > 4096) { set_err("mpc_create: n out of range [3,4096]"); return MPC_EINVAL; }
    if (cfg->model < 0 || cfg->model > 3) { set_err("mpc_create: unknown model"); return MPC_EINVAL; }
    if (cfg->collocation != MPC_COLLOC_FORWARD && cfg->collocation != MPC_COLLOC_MIDPOINT && cfg->collocation != MPC_COLLOC_CRANK_NICOLSON) {
        set_err("mpc_create: unknown collocation method"); return MPC_EINVAL; }
    if (cfg->objective != MPC_OBJ_MIN_TIME && cfg->objective != MPC_OBJ_QUADRATIC && cfg->objective != MPC_OBJ_MIN_TIME_VIA_POINTS) { set_err("mpc_create: unknown objective"); return MPC_EINVAL; }
    if (cfg->objective == MPC_OBJ_MIN_TIME_VIA_POINTS && (cfg->max_via_points < 1 || cfg->max_via_points > 64)) {
        set_err("mpc_create: minimum_time_via_points needs 1 <= max_via_points <= 64"); return MPC_EINVAL; }
    if (cfg->objective != MPC_OBJ_QUADRATIC && !cfg->dt_free) { set_err("mpc_create: minimum_time needs a variable grid (dt_free)"); return MPC_EINVAL; }
    if (!(cfg->dt_ref > 0)) { set_err("mpc_create: dt_ref must be > 0"); return MPC_EINVAL; }
    if (cfg->max_obstacles < 0 || cfg->max_obstacles > 4096 || (cfg->max_obstacles > 0 && (cfg->max_vertices < 1 || cfg->max_vertices > 64)) || cfg->max_obstacle_rows > 16) {
        set_err("mpc_create: obstacle capacities out of range (max_obstacles <= 4096, max_vertices <= 64, max_obstacle_rows <= 16)"); return MPC_EINVAL; }
    if (cfg->max_obstacles > 0 && cfg->footprint_kind != MPC_FOOTPRINT_POINT && cfg->footprint_kind != MPC_FOOTPRINT_CIRCLE && cfg->footprint_kind != MPC_FOOTPRINT_LINE &&
        cfg->footprint_kind != MPC_FOOTPRINT_TWO_CIRCLES && cfg->footprint_kind != MPC_FOOTPRINT_POLYGON) {
        set_err("mpc_create: unknown footprint model"); return MPC_EINVAL; }
    if (cfg->max_obstacles > 0 && cfg->footprint_kind == MPC_FOOTPRINT_POLYGON && (cfg->footprint_n_vertices < 1 || cfg->footprint_n_vertices > 16)) {
        set_err("mpc_create: the polygon footprint needs 1..16 vertices"); return MPC_EINVAL; }
    for (int j = 0; j < 2; ++j)
        if (!(cfg->u_lb[j] < cfg->u_ub[j])) { set_err("mpc_create: control box must be finite and non-empty"); return MPC_EINVAL; }
    if (cfg->precision != MPC_FP64 && cfg->precision != MPC_FP32 && cfg->precision != MPC_MIXED) { set_err("mpc_create: unknown precision"); return MPC_EINVAL; }
    if (cfg->cost_integration != MPC_COST_LEFT_SUM && cfg->cost_integration != MPC_COST_TRAPEZOIDAL) { set_err("mpc_create: unknown cost_integration"); return MPC_EINVAL; }
    if (cfg->hybrid_cost_minimum_time && cfg->objective != MPC_OBJ_QUADRATIC) { set_err("mpc_create: hybrid_cost_minimum_time belongs to the quadratic_form objective"); return MPC_EINVAL; }
    if (cfg->n_candidates == 1 && cfg->candidate_kind[0] != MPC_CAND_REFERENCE) {
        set_err("mpc_create: a single candidate must be MPC_CAND_REFERENCE (other kinds only run as hedges next to it: n_candidates >= 2)"); return MPC_EINVAL; }
    if (cfg->precision == MPC_MIXED && (cfg->max_obstacles > 0 || cfg->objective == MPC_OBJ_MIN_TIME_VIA_POINTS)) {
        set_err("mpc_create: MPC_MIXED is implemented for problems without clearance rows and via-points (their association would be redone by the refinement phase)"); return MPC_EINVAL; }
    if (cfg->hessian_mode != MPC_HESSIAN_EXACT && cfg->hessian_mode != MPC_HESSIAN_CONVEXIFIED) { set_err("mpc_create: unknown hessian_mode"); return MPC_EINVAL; }
    if (cfg->mu_strategy != MPC_MU_ADAPTIVE && cfg->mu_strategy != MPC_MU_MONOTONE) { set_err("mpc_create: unknown mu_strategy"); return MPC_EINVAL; }
    if (cfg->line_search != MPC_LS_DEFAULT && cfg->line_search != MPC_LS_MERIT && cfg->line_search != MPC_LS_FILTER) { set_err("mpc_create: unknown line_search"); return MPC_EINVAL; }
    if (cfg->max_time_us < 0) { set_err("mpc_create: max_time_us must be >= 0 (0 = no budget)"); return MPC_EINVAL; }
    if (cfg->n_candidates < 0 || cfg->n_candidates > MPC_MAX_CANDIDATES) { set_err("mpc_create: n_candidates must be in [0, MPC_MAX_CANDIDATES]"); return MPC_EINVAL; }
    for (int k = 0; k < cfg->n_candidates; ++k)
        if (cfg->candidate_kind[k] < MPC_CAND_REFERENCE || cfg->candidate_kind[k] > MPC_CAND_HERMITE_RF || cfg->candidate_max_iter[k] < 0) {
            set_err("mpc_create: unknown candidate kind or negative candidate_max_iter"); return MPC_EINVAL; }
    if (cfg->dt_free && !(cfg->dt_lb < cfg->dt_ub)) { set_err("mpc_create: dt_lb must be below dt_ub on the variable grid"); return MPC_EINVAL; }
    if (cfg->dt_free && !(cfg->dt_ref >= cfg->dt_lb && cfg->dt_ref <= cfg->dt_ub)) { set_err("mpc_create: dt_ref must lie in [dt_lb, dt_ub] on the variable grid"); return MPC_EINVAL; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_err("mpc_create: no HIP device available (this library has no CPU fallback)");
        return MPC_ENODEV;
    }
    if (device < 0 || device >= ndev) { set_err("mpc_create: device index out of range"); return MPC_ENODEV; }
    HIP_TRY(hipSetDevice(device));
    mpc_solver* s = new (std::nothrow) mpc_solver;
    if (!s) return MPC_ENOMEM;
    memset(s, 0, sizeof(*s));
    s->cfg = *cfg;
    mpc::fill_problem<double>(*cfg, s->P64);
    mpc::fill_problem<float>(*cfg, s->P32);
#ifdef MPC_DEV_SWITCHES      // developer A/B switches read from the environment: compiled out of the shipped library
    if (const char* e = getenv("MPC_NO_PIT")) { if (e[0] == '1') { s->P64.pit = 0; s->P32.pit = 0; } }      // serial sweeps only
    if (const char* e = getenv("MPC_PIT_MU")) { char* end = nullptr; const double v = strtod(e, &end); if (end != e && v >= 0) { s->P64.pit_mu_min = v; s->P32.pit_mu_min = (float)v; } }      // threshold of the partitioned sweeps
#endif
    if (cfg->precision == MPC_MIXED) {
        s->P32.tol = 1e-4f; s->P32.pit_mu_min = 1e-4f;        // phase 1 stops where fp32 residuals stop making sense
        s->P64.n_cand = 1;                                   // phase 2 refines the winner
        if (!(cfg->mu_init_dual > 0)) s->P64.mu_init_dual = 1e-5;      // phase 1 ended at a barrier of ~1e-5
        s->P64.mu_init_warm = 1e-3;                          // instances phase 1 did not converge start phase 2 from its last iterate
        s->P64.mu_strategy = 1;                              // the refinement follows the central path from mu = 1e-5 down: the monotone rule (measured: 4.2 refinement
                                                             // iterations against 9.4 with the adaptive one, which re-derives mu from the fp32 iterate's complementarity)
        if (s->P64.max_iter > 40) s->P64.max_iter = 40;      // a refinement that needs more than that is a solve of its own (phase-1 failures would hold the launch for 100)
    }
    {
        const int O = cfg->max_obstacles > 0 ? cfg->max_obstacles : 0;
        const int M = O > 0 ? (cfg->max_obstacle_rows > 0 ? cfg->max_obstacle_rows : 4) : 0;
        const int ntrig = ((cfg->model == MPC_MODEL_KINEMATIC_BICYCLE || cfg->model == MPC_MODEL_SIMPLE_CAR_FRONT) ? 4 : 3) +
                          (cfg->collocation == MPC_COLLOC_CRANK_NICOLSON ? 2 : 0);
        auto layout = [&](bool gs) {
            return mpc::WaveLayout::make(cfg->n, M, O, cfg->max_vertices > 0 ? cfg->max_vertices : 1, ntrig, s->P64.n_via,
                                         (O > 0 && (cfg->footprint_kind == MPC_FOOTPRINT_LINE || cfg->footprint_kind == MPC_FOOTPRINT_TWO_CIRCLES ||
                                                    cfg->footprint_kind == MPC_FOOTPRINT_POLYGON || cfg->enable_dynamic_obstacles)) ? M : 0,
                                         (O > 0 && cfg->enable_dynamic_obstacles) ? O : 0, solver_ext(s) ? mpc::NSTG_EXT : mpc::NSTG_BASE,
                                         (O > 0 && cfg->enable_dynamic_obstacles && (cfg->footprint_kind == MPC_FOOTPRINT_LINE || cfg->footprint_kind == MPC_FOOTPRINT_TWO_CIRCLES ||
                                                                                      cfg->footprint_kind == MPC_FOOTPRINT_POLYGON)) ? M : 0,
                                         cfg->precision == MPC_FP32 ? 4 : 8, gs);      // (MPC_MIXED has no clearance rows: both of its phases see the same layout)
        };
        s->WL = layout(false);
        s->WLg = layout(true);
    }
    auto lds_of = [](const mpc::WaveLayout& L, size_t tsize, size_t psize) { return ((((size_t)L.total * tsize) + 15) & ~(size_t)15) + 16 + ((psize + 15) & ~(size_t)15); };
    // Where the factorisation data lives, per precision (MPC_STAGE_AUTO).  The register file holds these kernels at one wave per SIMD, so a CU has room for four workgroups; an
    // LDS record that fits fewer than four times leaves SIMDs without a wave.  Measured on the MI355X (profiles/r05_stage_data_probe.log): at EQUAL residency the global form
    // costs 24 % per iteration at n = 50 and ~5 % at n = 80 / 120; it pays x2.2 - x2.8 at n = 120 in fp64 (1 -> 4 workgroups per CU) and x1.3 - x1.5 from residency alone at n = 80
    // with 16 polygons.  The rule, stated once (also in include/mpc_hip.h): fp64 takes the global form when the LDS form leaves at least HALF of a CU's SIMDs empty and the global
    // form fills more of them; plain fp32 already when the LDS form leaves ONE of four empty (n = 120: 7.06 -> 6.43 ms since the block is laid out in tiles) -- in fp32 the two
    // forms agree to rounding, not bit for bit (the compiler contracts / packs the fp32 lane-parallel passes differently around global loads), so an fp32 handle under
    // MPC_STAGE_AUTO reproduces itself run to run but not the results of builds before 0.5.0; both phases of MPC_MIXED keep the LDS form.  Every kernel level exists in both forms (r06: the extended levels too).
    {
        const bool can_gs = true;      // (r06: every kernel level exists in the global form)
        auto per_cu = [](size_t lds) { const size_t k = (160u * 1024u) / lds; return k > 4 ? (size_t)4 : k; };
        auto choose = [&](size_t tsize, size_t psize) {
            if (!can_gs || cfg->stage_data == MPC_STAGE_LDS) return false;
            if (cfg->stage_data == MPC_STAGE_GLOBAL) return true;
            const size_t a = lds_of(s->WL, tsize, psize), g = lds_of(s->WLg, tsize, psize);
            if (a > 160u * 1024u) return g <= 160u * 1024u;           // only the global form fits at all
            // (plain fp32: 3 -> 4 workgroups per CU at n = 120 measured 10 % faster since the block is laid out in tiles -- 7.06 -> 6.43 ms; the two forms agree to rounding there, not
            //  bit for bit.  Not the fp32 phase of MPC_MIXED: 9.06 -> 9.88 ms for both phases, the hand-off between the forms costs more than the phase gains)
            return per_cu(a) <= ((tsize == 4 && cfg->precision == MPC_FP32) ? 3u : 2u) && per_cu(g) > per_cu(a);
        };
        if (cfg->stage_data != MPC_STAGE_AUTO && cfg->stage_data != MPC_STAGE_LDS && cfg->stage_data != MPC_STAGE_GLOBAL) { set_err("mpc_create: unknown stage_data"); delete s; return MPC_EINVAL; }
        s->gs32 = cfg->precision != MPC_FP64 && choose(4, sizeof(mpc::Problem<float>));
        // (the refinement phase of MPC_MIXED -- one candidate, a handful of iterations -- measured faster in the LDS form: 9.0 against 9.4 ms for both phases at n = 120, B = 1024)
        s->gs64 = cfg->precision != MPC_FP32 && (cfg->precision != MPC_MIXED || cfg->stage_data == MPC_STAGE_GLOBAL) && choose(8, sizeof(mpc::Problem<double>));
    }
    // Two waves per SIMD (mpc_solve_kernel.hpp, W2): the fp64 headline level without clearance rows, when the LDS record fits eight times into a CU (everything in LDS: about
    // n <= 24 grid points -- the grid sizes of the reference's shipped parameter files).  A launch takes it when it has at least w2_min_batch instances (the device then has waves
    // waiting for a SIMD, and a second resident wave fills the issue slots the first leaves idle); below that a launch lasts as long as its slowest wave, which runs fastest
    // alone.  Same arithmetic on the same numbers: results bit for bit those of the one-wave kernels.  (With the factorisation data in global memory the second wave buys
    // nothing -- measured at n = 50, profiles/r06_w2_probe.log: the form's cost is the CU's vector-memory path, which eight waves share --, so grids whose LDS form does not fit
    // eight times keep the one-wave kernels.)
    s->w2_gs = false;
    s->wave_lds_w2 = lds_of(s->WL, 8, sizeof(mpc::Problem<double>));
    s->w2_ok = cfg->precision == MPC_FP64 && cfg->stage_data == MPC_STAGE_AUTO && !solver_ext(s) && cfg->max_obstacles <= 0 && s->wave_lds_w2 <= (160u * 1024u) / 8u;
    s->w2_min_batch = cfg->two_wave_min_batch > 0 ? cfg->two_wave_min_batch : 4096;      // (measured under the filter line search, profiles/r06_w2_probe.log, n = 12 / 20 / 24: x1.03 / x1.00-1.02 / x1.30-1.34 there, x1.06 / x1.4 / x1.37 at 8192, x1.46 / x1.64 / x1.6 at 32768; below: x0.83 / x1.13 / x1.29 at 2048, x0.81 / x1.02 / x1.02 at 1024 -- a small launch lasts as long as its slowest wave, which runs fastest alone)
    if (cfg->two_wave_min_batch < 0) s->w2_ok = false;
#ifdef MPC_DEV_SWITCHES
    if (const char* e = getenv("MPC_W2_FORCE")) {      // developer A/B: the 256-register kernel variant whatever the record's size (LDS form)
        if (e[0] == '1' && cfg->two_wave_min_batch >= 0 && cfg->precision == MPC_FP64 && !solver_ext(s) && cfg->max_obstacles <= 0) { s->w2_ok = true; s->w2_gs = false; s->wave_lds_w2 = lds_of(s->WL, 8, sizeof(mpc::Problem<double>)); }
    }
    if (const char* e = getenv("MPC_W2_GS")) {      // developer A/B: the two-wave kernel in the global form where the LDS form does not fit eight times
        if (e[0] == '1' && !s->w2_ok && cfg->two_wave_min_batch >= 0 && cfg->precision == MPC_FP64 && cfg->stage_data == MPC_STAGE_AUTO && !solver_ext(s) && cfg->max_obstacles <= 0 && lds_of(s->WLg, 8, sizeof(mpc::Problem<double>)) <= (160u * 1024u) / 8u) {
            s->w2_gs = true; s->w2_ok = true; s->wave_lds_w2 = lds_of(s->WLg, 8, sizeof(mpc::Problem<double>));
        }
    }
#endif
    s->wave_lds32 = lds_of(s->gs32 ? s->WLg : s->WL, 4, sizeof(mpc::Problem<float>));
    s->wave_lds = cfg->precision == MPC_FP32 ? s->wave_lds32 : lds_of(s->gs64 ? s->WLg : s->WL, 8, sizeof(mpc::Problem<double>));
    if (s->wave_lds > 160u * 1024u) {
        set_err("mpc_create: the working set of one instance (n, max_obstacles, max_vertices, precision) does not fit in the 160 KB of LDS "
                "of a compute unit (about n <= 215 grid points in fp64 without obstacles; n <= 590 with the factorisation data in global memory)");
        delete s;
        return MPC_EINVAL;
    }
    s->device = device;
    s->max_batch = max_batch;
    const size_t n = cfg->n;
    hipError_t er = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (er == hipSuccess) er = hipEventCreate(&s->ev0);
    if (er == hipSuccess) er = hipEventCreate(&s->ev1);
    if (er == hipSuccess) er = hipEventCreate(&s->cev0);
    if (er == hipSuccess) er = hipEventCreate(&s->cev1);
    const size_t Bm = max_batch;
    {
        const size_t O = cfg->max_obstacles > 0 ? cfg->max_obstacles : 0, V = cfg->max_vertices > 0 ? cfg->max_vertices : 1;
        // inputs: x0 xf u_prev dt_prev | x_init u_init dt_init | n_obstacles n_vertices vertices radius velocity (each piece 256-byte aligned)
        s->in_cap = Bm * (3 + 3 + 2 + 1) * 8 + Bm * (5 * n + 1) * 8 + (O ? Bm * 4 + Bm * O * 4 + Bm * O * V * 2 * 8 + Bm * O * 8 + Bm * O * 2 * 8 : 0) + 16 * 256;
        // outputs: x_out u_out dt_out status iters
        s->out_cap = Bm * (5 * n + 1) * 8 + Bm * 12 + 10 * 256;
        if (er == hipSuccess) er = hipHostMalloc((void**)&s->h_in, s->in_cap, hipHostMallocDefault);
        if (er == hipSuccess) er = hipHostMalloc((void**)&s->h_out, s->out_cap, hipHostMallocDefault);
        if (er == hipSuccess) er = hipMalloc((void**)&s->d_in, s->in_cap);
        if (er == hipSuccess) er = hipMalloc((void**)&s->d_out, s->out_cap);
    }
    if (er == hipSuccess) er = hipMalloc((void**)&s->d_ngrid, Bm * 4);
    if (er == hipSuccess) { std::vector<int32_t> full(Bm, cfg->n); er = hipMemcpy(s->d_ngrid, full.data(), Bm * 4, hipMemcpyHostToDevice); }      // never uninitialised
    if (s->P64.n_via > 0) {
        if (er == hipSuccess) er = hipMalloc((void**)&s->d_nvia, Bm * 4);
        if (er == hipSuccess) er = hipMalloc((void**)&s->d_via, Bm * (size_t)s->P64.n_via * 3 * 8);
        if (er == hipSuccess) er = hipMemset(s->d_nvia, 0, Bm * 4);
        s->p_nvia = s->d_nvia; s->p_via = s->d_via;
    }
    if (cfg->max_obstacles > 0) {
        if (er == hipSuccess) er = hipMalloc((void**)&s->d_rows_dropped, Bm * 4);
        if (er == hipSuccess) er = hipMemset(s->d_rows_dropped, 0, Bm * 4);
    }
    if (cfg->precision == MPC_MIXED && er == hipSuccess) er = hipMalloc((void**)&s->d_iters1, Bm * 4);
    if (s->gs32 || s->gs64 || s->w2_gs || s->WL.GSW > 0) {
        // blocks of factorisation data / of the clearance rows' elastic arrays (every layout with clearance rows has one, WaveLayout::GSW), a pool per XCD (mpc_solve_kernel.hpp): per XCD as many as the whole device has CUs -- an XCD has an eighth of them and a CU holds at
        // most 8 of these one-wave workgroups (2 per SIMD under the register budget of any build), so a pool can never run dry --, never more than the largest launch has
        // workgroups.  Stale contents are never read: every word is written before it is read within a solve.
        hipDeviceProp_t prop;
        if (er == hipSuccess) er = hipGetDeviceProperties(&prop, device);
        const size_t c32 = (cfg->precision != MPC_FP64 && s->P32.n_cand > 1) ? (size_t)s->P32.n_cand : 1, c64 = (cfg->precision != MPC_FP32 && s->P64.n_cand > 1) ? (size_t)s->P64.n_cand : 1;
        const size_t grid = Bm * (c32 > c64 ? c32 : c64);
        // Pool size per XCD = twice what ONE XCD can hold at once, from the runtime: workgroups per CU of the kernel that uses the pool (occupancy API: registers,
        // LDS, one-wave workgroups) x the CUs of an XCD (the device's CUs / the number of distinct XCC ids a probe launch sees: 8 on an MI355X, 1 in a partitioned mode).  A
        // pool can then never run dry, and the factor two keeps the claim probe of mpc_solve_kernel.hpp at a step or two when every CU is full.  Never more than the largest
        // launch has workgroups.  Memory: 8 pools x per_xcd x block bytes (MI355X, n = 120 in fp64: 8 x 256 x 73 KB = 150 MB per handle; include/mpc_hip.h, mpc_create).
        size_t per_cu = 8;      // (two one-wave workgroups per SIMD: the upper limit of any build)
        int n_xcc = 1;
        if (er == hipSuccess) {
            int occ = 0;
            mpc::SolveLaunch a{};
            a.level = !solver_ext(s) ? 0 : (s->P64.costx ? 2 : 1);
            const bool f32 = cfg->precision == MPC_FP32;
            a.w2 = !f32 && s->w2_gs;
            const bool gsf = f32 ? s->gs32 : (s->gs64 || s->w2_gs);
            a.L = gsf ? s->WLg : s->WL;
            a.lds = f32 ? s->wave_lds32 : (s->w2_gs ? s->wave_lds_w2 : s->wave_lds);
            if (pool_kernel_occupancy(f32, cfg->model, a, &occ) == hipSuccess && occ > 0) per_cu = (size_t)occ;
            unsigned* d_mask = nullptr; unsigned h_mask = 0;
            if (hipMalloc((void**)&d_mask, 4) == hipSuccess) {
                if (hipMemset(d_mask, 0, 4) == hipSuccess) {
                    hipLaunchKernelGGL(mpc::xcc_probe_kernel, dim3(16u * (unsigned)prop.multiProcessorCount), dim3(64), 0, 0, d_mask);
                    if (hipMemcpy(&h_mask, d_mask, 4, hipMemcpyDeviceToHost) == hipSuccess && h_mask) n_xcc = __builtin_popcount(h_mask & 0xffu);
                }
                (void)hipFree(d_mask);
            }
        }
        const size_t cus_per_xcd = ((size_t)prop.multiProcessorCount + (size_t)n_xcc - 1) / (size_t)n_xcc;
        size_t per_xcd = er == hipSuccess ? 2 * per_cu * cus_per_xcd : 512;
        if (per_xcd < 64) per_xcd = 64;
        s->n_gslots = (int)per_xcd;
        const size_t blk64 = cfg->precision != MPC_FP32 ? (size_t)((s->gs64 || s->w2_gs) ? s->WLg.GSW : s->WL.GSW) * 8 : 0, blk32 = cfg->precision != MPC_FP64 ? (size_t)(s->gs32 ? s->WLg.GSW : s->WL.GSW) * 4 : 0;
        if (er == hipSuccess) er = hipMalloc(&s->d_gstage, 8 * per_xcd * (blk64 > blk32 ? blk64 : blk32) + (size_t)mpc::GlobalStage::kPrefetchPad * 8);
        // developer check (scripts/dev/gs_sweep.py under MPC_POISON_GSTAGE=1): every word of the pool starts as a NaN pattern, so a word that some path consumes before writing it shows up in the results
        if (er == hipSuccess) { if (const char* e = getenv("MPC_POISON_GSTAGE")) { if (e[0] == '1') er = hipMemset(s->d_gstage, 0xFF, 8 * per_xcd * (blk64 > blk32 ? blk64 : blk32) + (size_t)mpc::GlobalStage::kPrefetchPad * 8); } }
        if (er == hipSuccess) er = hipMalloc((void**)&s->d_gslots, 8 * per_xcd * 4);
        if (er == hipSuccess) er = hipMemset(s->d_gslots, 0, 8 * per_xcd * 4);
    }
    if (cfg->dual_warm_start || cfg->precision == MPC_MIXED) {
        s->dual_words = mpc::IpmWave<double, 0, 0>::dual_words(s->WL.NS);
        if (er == hipSuccess) er = hipMalloc((void**)&s->d_dual, Bm * (size_t)s->dual_words * 8);
        if (er == hipSuccess) er = hipMemset(s->d_dual, 0, Bm * (size_t)s->dual_words * 8);
    }
    if (s->P32.n_cand > 1) {
        const size_t C_ = s->P32.n_cand;
        if (er == hipSuccess) er = hipMalloc((void**)&s->d_cwin, Bm * 4);
        if (er == hipSuccess) er = hipMalloc((void**)&s->d_cexited, Bm * 4);
        if (er == hipSuccess) er = hipMalloc((void**)&s->d_citsum, Bm * 4);
        if (er == hipSuccess) er = hipMalloc((void**)&s->d_crec, C_ * Bm * (5 * n + 3 + (size_t)s->dual_words) * 8);
        if (er == hipSuccess) er = hipMalloc((void**)&s->d_winner, Bm * 4);
        if (er == hipSuccess) er = hipMalloc((void**)&s->d_iters_total, Bm * 4);
        if (er == hipSuccess) er = hipMemset(s->d_cwin, 0x7f, Bm * 4);
        if (er == hipSuccess) er = hipMemset(s->d_cexited, 0, Bm * 4);
        if (er == hipSuccess) er = hipMemset(s->d_citsum, 0, Bm * 4);
    }
    // the fills above run on the null stream, the solves on the handle's own non-blocking stream: nothing orders the two, so wait here
    if (er == hipSuccess) er = hipDeviceSynchronize();
    if (er != hipSuccess) {
        set_err("mpc_create: allocation", er);
        mpc_destroy(s);
        return er == hipErrorOutOfMemory ? MPC_ENOMEM : MPC_EHIP;
    }
    *out = s;
    return MPC_OK;
}

int mpc_reset(mpc_solver* s) {
    if (!s) return MPC_EINVAL;
    g_err[0] = 0;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->d_dual) HIP_TRY(hipMemset(s->d_dual, 0, (size_t)s->max_batch * s->dual_words * 8));      // forget the multipliers
    if (s->d_gslots) HIP_TRY(hipMemset(s->d_gslots, 0, 8 * (size_t)s->n_gslots * 4));                      // claim words of the factorisation-data blocks (self-restoring unless a launch was aborted)
    if (s->d_cwin) {      // candidate bookkeeping back to idle (it is self-restoring unless a launch was aborted)
        HIP_TRY(hipMemset(s->d_cwin, 0x7f, (size_t)s->max_batch * 4));
        HIP_TRY(hipMemset(s->d_cexited, 0, (size_t)s->max_batch * 4));
        HIP_TRY(hipMemset(s->d_citsum, 0, (size_t)s->max_batch * 4));
    }
    HIP_TRY(hipDeviceSynchronize());      // null-stream fills vs the handle's non-blocking stream (see mpc_create)
    return MPC_OK;
}

void mpc_destroy(mpc_solver* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    void* bufs[] = {s->d_gslots, s->d_gstage, s->d_stage, s->d_iters1, s->d_dual, s->d_rows_dropped, s->d_cwin, s->d_cexited, s->d_citsum, s->d_crec, s->d_winner, s->d_iters_total, s->d_nvia, s->d_via, s->d_ngrid, s->d_in, s->d_out};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (s->h_in) (void)hipHostFree(s->h_in);
    if (s->h_out) (void)hipHostFree(s->h_out);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->cev0) (void)hipEventDestroy(s->cev0);
    if (s->cev1) (void)hipEventDestroy(s->cev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

}  // extern "C"

// fills the launch record of mpc_solve_kernel.hpp from the handle; the kernels themselves are instantiated per (precision, model) in
// mpc_solve_inst.hip (split build: one object each, compiled in parallel) or right here (single translation unit)
template <typename T, int MODEL>
static hipError_t launch_model(mpc_solver* s, const mpc::Problem<T>& P, int B, const double* x0, const double* xf, const double* up,
                               const double* dtp, const double* xi, const double* ui, const double* dti, const mpc_obstacles& ob, double* xo, double* uo,
                               double* dto, int32_t* st, int32_t* it) {
    mpc::SolveLaunch a;
    a.level = !solver_ext(s) ? 0 : (s->P64.costx ? 2 : 1);
    a.lds = sizeof(T) == 4 ? s->wave_lds32 : s->wave_lds;
    a.stream = s->stream;
    a.w2 = sizeof(T) == 8 && s->w2_ok && B >= s->w2_min_batch;
    if (a.w2) a.lds = s->wave_lds_w2;
    const bool gs = a.w2 ? s->w2_gs : (sizeof(T) == 4 ? s->gs32 : s->gs64);
    a.L = gs ? s->WLg : s->WL; a.B = B;
    a.gstage = s->d_gstage; a.gslots = s->d_gslots; a.n_gslots = s->n_gslots;
    a.x0 = x0; a.xf = xf; a.u_prev = up; a.dt_prev = dtp; a.x_init = xi; a.u_init = ui; a.dt_init = dti; a.obst = ob;
    a.n_grid = s->use_ngrid ? s->d_ngrid : nullptr; a.n_via = s->p_nvia; a.via = s->p_via;
    // kept multipliers: a launch starts from them under dual_warm_start; in MPC_MIXED without it the block is only the hand-off from the fp32 phase
    // (which leaves its multipliers) to the fp64 phase (which starts from them) -- nothing is carried from the slot's previous control cycle
    const int dual_read = (s->cfg.dual_warm_start || (s->cfg.precision == MPC_MIXED && sizeof(T) == 8)) ? 1 : 0;
    a.cc = {P.n_cand, s->d_cwin, s->d_cexited, s->d_citsum, s->d_crec, s->d_winner, s->d_iters_total, s->d_rows_dropped, s->d_dual, s->dual_words, dual_read};
    a.iters_add = (s->cfg.precision == MPC_MIXED && sizeof(T) == 8) ? s->d_iters1 : nullptr;
    a.x_out = xo; a.u_out = uo; a.dt_out = dto; a.status = st; a.iters = it;
    return mpc::launch_solve<T, MODEL>(a, P);
}

template <typename T>
static hipError_t launch_prec(mpc_solver* s, const mpc::Problem<T>& P, int B, const double* x0, const double* xf, const double* up,
                        const double* dtp, const double* xi, const double* ui, const double* dti, const mpc_obstacles& ob, double* xo, double* uo,
                        double* dto, int32_t* st, int32_t* it) {
#ifdef MPC_DEV_ONE_MODEL       // developer builds (fast compile, asm inspection): only the fp64 instantiations of ONE model exist (-DMPC_DEV_ONE_MODEL=<model id>)
    if (sizeof(T) == 8) return launch_model<double, MPC_DEV_ONE_MODEL>(s, s->P64, B, x0, xf, up, dtp, xi, ui, dti, ob, xo, uo, dto, st, it);
    return hipErrorInvalidConfiguration;
#else
    switch (s->cfg.model) {
        case MPC_MODEL_UNICYCLE: return launch_model<T, mpc::MODEL_UNICYCLE>(s, P, B, x0, xf, up, dtp, xi, ui, dti, ob, xo, uo, dto, st, it);
        case MPC_MODEL_SIMPLE_CAR: return launch_model<T, mpc::MODEL_SIMPLE_CAR>(s, P, B, x0, xf, up, dtp, xi, ui, dti, ob, xo, uo, dto, st, it);
        case MPC_MODEL_SIMPLE_CAR_FRONT: return launch_model<T, mpc::MODEL_SIMPLE_CAR_FRONT>(s, P, B, x0, xf, up, dtp, xi, ui, dti, ob, xo, uo, dto, st, it);
        default: return launch_model<T, mpc::MODEL_KINEMATIC_BICYCLE>(s, P, B, x0, xf, up, dtp, xi, ui, dti, ob, xo, uo, dto, st, it);
    }
#endif
}

extern "C" {

int mpc_solve_batch_device(mpc_solver* s, int32_t B, const double* d_x0, const double* d_xf, const double* d_u_prev,
                           const double* d_dt_prev, const double* d_x_init, const double* d_u_init, const double* d_dt_init,
                           const mpc_obstacles* d_obstacles, double* d_x_out, double* d_u_out, double* d_dt_out, int32_t* d_status,
                           int32_t* d_iters) {
    g_err[0] = 0;
    if (!s || !d_x0 || !d_xf || !d_x_out || !d_u_out || !d_dt_out) { set_err("mpc_solve_batch_device: null argument"); return MPC_EINVAL; }
    if (B <= 0) return MPC_OK;
    if (B > s->max_batch) { set_err("mpc_solve_batch_device: B exceeds max_batch"); return MPC_EBATCH; }
    if ((d_x_init != nullptr) != (d_u_init != nullptr) || (d_x_init != nullptr) != (d_dt_init != nullptr)) {
        set_err("mpc_solve_batch: x_init, u_init and dt_init must be given together (all three or none)"); return MPC_EINVAL; }
    if (s->use_ngrid && B > s->ngrid_B) { set_err("mpc_solve_batch: B exceeds the batch the per-instance grid sizes were set for (mpc_set_grid_sizes)"); return MPC_EBATCH; }
    if (s->P64.n_via > 0 && s->p_nvia == s->d_nvia && s->nvia_B > 0 && B > s->nvia_B) {
        set_err("mpc_solve_batch: B exceeds the batch the via-points were set for (mpc_set_via_points)"); return MPC_EBATCH; }
    mpc_obstacles ob = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (s->cfg.max_obstacles > 0) {
        if (!d_obstacles || !d_obstacles->n_obstacles || !d_obstacles->n_vertices || !d_obstacles->vertices) {
            set_err("mpc_solve_batch_device: the solver was created with max_obstacles > 0 but no obstacles were passed");
            return MPC_EINVAL;
        }
        ob = *d_obstacles;
    }
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventRecord(s->ev0, s->stream));
    hipError_t le;
    if (s->cfg.precision == MPC_MIXED) {
        // phase 1 (fp32, candidates, tol 1e-4) leaves iterate + multipliers; phase 2 (fp64, one candidate) refines them in place
        le = launch_prec<float>(s, s->P32, B, d_x0, d_xf, d_u_prev, d_dt_prev, d_x_init, d_u_init, d_dt_init, ob, d_x_out, d_u_out, d_dt_out, d_status, s->d_iters1);
        if (le == hipSuccess)
            le = launch_prec<double>(s, s->P64, B, d_x0, d_xf, d_u_prev, d_dt_prev, d_x_out, d_u_out, d_dt_out, ob, d_x_out, d_u_out, d_dt_out, d_status, d_iters);
    } else if (s->cfg.precision == MPC_FP32)
        le = launch_prec<float>(s, s->P32, B, d_x0, d_xf, d_u_prev, d_dt_prev, d_x_init, d_u_init, d_dt_init, ob, d_x_out, d_u_out, d_dt_out, d_status, d_iters);
    else
        le = launch_prec<double>(s, s->P64, B, d_x0, d_xf, d_u_prev, d_dt_prev, d_x_init, d_u_init, d_dt_init, ob, d_x_out, d_u_out, d_dt_out, d_status, d_iters);
    HIP_TRY(le);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev1, s->stream));
    s->timed = true;
    s->last_status = d_status; s->last_iters = d_iters;
    return MPC_OK;
}

int mpc_step_batch_device(mpc_solver* s, int32_t B, const double* d_x0, const double* d_xf, const double* d_u_prev, const double* d_dt_prev,
                          const double* d_x_init, const double* d_u_init, const double* d_dt_init, const mpc_obstacles* d_obstacles,
                          int32_t outer_iterations, int32_t adapt, int32_t n_min, int32_t n_max, double dt_hyst_ratio,
                          double* d_x_out, double* d_u_out, double* d_dt_out, int32_t* d_status, int32_t* d_iters) {
    // PredictiveController::step repeats (grid update -> solve) outer_ocp_iterations times per control cycle (src/controller.cpp:70-72,172): every
    // repetition after the first starts from the solution just computed, in place on the output arrays; everything is enqueued on the solver's stream
    int rc = mpc_solve_batch_device(s, B, d_x0, d_xf, d_u_prev, d_dt_prev, d_x_init, d_u_init, d_dt_init, d_obstacles, d_x_out, d_u_out, d_dt_out, d_status, d_iters);
    for (int it = 1; it < outer_iterations && rc == MPC_OK; ++it) {
        // variable grid: single-step adaptation + resampling; fixed grid: nothing -- its warm-start shift belongs to the first outer iteration of a cycle (`new_run`,
        // full_discretization_grid_base_se2.cpp:96-100), which is the caller's (the repetition starts from the solution just computed as it is)
        if (s->cfg.dt_free) rc = mpc_grid_update_device(s, B, d_x0, d_x_out, d_u_out, d_dt_out, adapt, n_min, n_max, dt_hyst_ratio);
        if (rc == MPC_OK)
            rc = mpc_solve_batch_device(s, B, d_x0, d_xf, d_u_prev, d_dt_prev, d_x_out, d_u_out, d_dt_out, d_obstacles, d_x_out, d_u_out, d_dt_out, d_status, d_iters);
    }
    return rc;
}

int mpc_last_candidates(mpc_solver* s, int32_t B, int32_t* winner, int32_t* iters_total) {
    g_err[0] = 0;
    if (!s) return MPC_EINVAL;
    if (B <= 0) return MPC_OK;
    if (B > s->max_batch) { set_err("mpc_last_candidates: B exceeds max_batch"); return MPC_EBATCH; }
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->P32.n_cand > 1) {
        if (winner) HIP_TRY(hipMemcpy(winner, s->d_winner, (size_t)B * 4, hipMemcpyDeviceToHost));
        if (iters_total) HIP_TRY(hipMemcpy(iters_total, s->d_iters_total, (size_t)B * 4, hipMemcpyDeviceToHost));
        return MPC_OK;
    }
    if (winner) {
        if (!s->last_status) { set_err("mpc_last_candidates: the last solve kept no status array"); return MPC_EINVAL; }
        HIP_TRY(hipMemcpy(winner, s->last_status, (size_t)B * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) winner[b] = winner[b] == MPC_CONVERGED ? 0 : -1;
    }
    if (iters_total) {
        if (!s->last_iters) { set_err("mpc_last_candidates: the last solve kept no iteration array"); return MPC_EINVAL; }
        HIP_TRY(hipMemcpy(iters_total, s->last_iters, (size_t)B * 4, hipMemcpyDeviceToHost));
    }
    return MPC_OK;
}

int mpc_last_rows_dropped(mpc_solver* s, int32_t B, int32_t* rows_dropped) {
    g_err[0] = 0;
    if (!s || !rows_dropped) return MPC_EINVAL;
    if (B <= 0) return MPC_OK;
    if (B > s->max_batch) { set_err("mpc_last_rows_dropped: B exceeds max_batch"); return MPC_EBATCH; }
    if (!s->d_rows_dropped) { memset(rows_dropped, 0, (size_t)B * 4); return MPC_OK; }
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(rows_dropped, s->d_rows_dropped, (size_t)B * 4, hipMemcpyDeviceToHost));
    return MPC_OK;
}

int mpc_set_via_points(mpc_solver* s, int32_t B, const int32_t* n_via, const double* via) {
    g_err[0] = 0;
    if (!s) return MPC_EINVAL;
    if (s->P64.n_via <= 0) { set_err("mpc_set_via_points: the solver was not created with objective MPC_OBJ_MIN_TIME_VIA_POINTS"); return MPC_EINVAL; }
    HIP_TRY(hipSetDevice(s->device));
    s->p_nvia = s->d_nvia; s->p_via = s->d_via;
    if (!n_via || !via) {
        HIP_TRY(hipMemsetAsync(s->d_nvia, 0, (size_t)s->max_batch * 4, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        s->nvia_B = 0;             // every instance: no via-points
        return MPC_OK;
    }
    if (B <= 0 || B > s->max_batch) { set_err("mpc_set_via_points: B out of range"); return MPC_EBATCH; }
    for (int b = 0; b < B; ++b)
        if (n_via[b] < 0 || n_via[b] > s->P64.n_via) { set_err("mpc_set_via_points: n_via[b] must be in [0, cfg.max_via_points]"); return MPC_EINVAL; }
    HIP_TRY(hipMemcpyAsync(s->d_nvia, n_via, (size_t)B * 4, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(s->d_via, via, (size_t)B * s->P64.n_via * 3 * 8, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->nvia_B = B;
    return MPC_OK;
}

int mpc_set_via_points_device(mpc_solver* s, const int32_t* d_n_via, const double* d_via) {
    g_err[0] = 0;
    if (!s) return MPC_EINVAL;
    if (s->P64.n_via <= 0) { set_err("mpc_set_via_points_device: the solver was not created with objective MPC_OBJ_MIN_TIME_VIA_POINTS"); return MPC_EINVAL; }
    if (!d_n_via || !d_via) return mpc_set_via_points(s, 0, nullptr, nullptr);
    s->p_nvia = d_n_via; s->p_via = d_via;
    return MPC_OK;
}

int mpc_costmap_to_obstacles_device(mpc_solver* s, int32_t B, const uint8_t* d_cost, int32_t size_x, int32_t size_y, double resolution,
                                    const double* d_origin, const double* d_robot_pose, double behind_robot_dist,
                                    int32_t* d_n_obstacles, int32_t* d_n_vertices, double* d_vertices, int32_t* d_dropped) {
    g_err[0] = 0;
    if (!s || !d_cost || !d_origin || !d_robot_pose || !d_n_obstacles || !d_n_vertices || !d_vertices) { set_err("mpc_costmap_to_obstacles_device: null argument"); return MPC_EINVAL; }
    if (s->cfg.max_obstacles <= 0) { set_err("mpc_costmap_to_obstacles_device: the solver was created with max_obstacles = 0"); return MPC_EINVAL; }
    if (size_x < 1 || size_y < 1 || !(resolution > 0)) { set_err("mpc_costmap_to_obstacles_device: bad costmap geometry"); return MPC_EINVAL; }
    if (B <= 0) return MPC_OK;
    if (B > s->max_batch) { set_err("mpc_costmap_to_obstacles_device: B exceeds max_batch"); return MPC_EBATCH; }
    HIP_TRY(hipSetDevice(s->device));
    mpc::CostmapArgs a;
    a.cost = d_cost; a.origin = d_origin; a.pose = d_robot_pose;
    a.size_x = size_x; a.size_y = size_y; a.resolution = resolution; a.behind_dist = behind_robot_dist;
    a.O = s->cfg.max_obstacles; a.V = s->cfg.max_vertices > 0 ? s->cfg.max_vertices : 1;
    a.n_obstacles = d_n_obstacles; a.n_vertices = d_n_vertices; a.vertices = d_vertices; a.dropped = d_dropped;
    HIP_TRY(hipEventRecord(s->cev0, s->stream));
    hipLaunchKernelGGL(mpc::costmap_to_obstacles_kernel, dim3(B), dim3(mpc::kCostmapThreads), 0, s->stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->cev1, s->stream));
    return MPC_OK;
}

int mpc_costmap_to_obstacles(mpc_solver* s, int32_t B, const uint8_t* cost, int32_t size_x, int32_t size_y, double resolution,
                             const double* origin, const double* robot_pose, double behind_robot_dist,
                             int32_t* n_obstacles, int32_t* n_vertices, double* vertices, int32_t* dropped) {
    g_err[0] = 0;
    if (!s || !cost || !origin || !robot_pose || !n_obstacles || !n_vertices || !vertices) { set_err("mpc_costmap_to_obstacles: null argument"); return MPC_EINVAL; }
    if (B <= 0) return MPC_OK;
    if (s->cfg.max_obstacles <= 0 || size_x < 1 || size_y < 1) { set_err("mpc_costmap_to_obstacles: bad argument"); return MPC_EINVAL; }
    HIP_TRY(hipSetDevice(s->device));
    const size_t O = s->cfg.max_obstacles, V = s->cfg.max_vertices > 0 ? s->cfg.max_vertices : 1, nb = (size_t)B;
    const size_t sz[7] = {nb * size_x * size_y, nb * 2 * 8, nb * 3 * 8, nb * 4, nb * O * 4, nb * O * V * 2 * 8, nb * 4};
    void* d[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipError_t er = stage_carve(s, sz, 7, d);
    int rc = MPC_OK;
    if (er == hipSuccess) er = hipMemcpyAsync(d[0], cost, sz[0], hipMemcpyHostToDevice, s->stream);
    if (er == hipSuccess) er = hipMemcpyAsync(d[1], origin, sz[1], hipMemcpyHostToDevice, s->stream);
    if (er == hipSuccess) er = hipMemcpyAsync(d[2], robot_pose, sz[2], hipMemcpyHostToDevice, s->stream);
    if (er == hipSuccess) er = hipMemsetAsync(d[4], 0, sz[4], s->stream);
    if (er == hipSuccess) er = hipMemsetAsync(d[5], 0, sz[5], s->stream);
    if (er == hipSuccess)
        rc = mpc_costmap_to_obstacles_device(s, B, (const uint8_t*)d[0], size_x, size_y, resolution, (const double*)d[1], (const double*)d[2], behind_robot_dist,
                                             (int32_t*)d[3], (int32_t*)d[4], (double*)d[5], (int32_t*)d[6]);
    if (er == hipSuccess && rc == MPC_OK) er = hipMemcpyAsync(n_obstacles, d[3], sz[3], hipMemcpyDeviceToHost, s->stream);
    if (er == hipSuccess && rc == MPC_OK) er = hipMemcpyAsync(n_vertices, d[4], sz[4], hipMemcpyDeviceToHost, s->stream);
    if (er == hipSuccess && rc == MPC_OK) er = hipMemcpyAsync(vertices, d[5], sz[5], hipMemcpyDeviceToHost, s->stream);
    if (er == hipSuccess && rc == MPC_OK && dropped) er = hipMemcpyAsync(dropped, d[6], sz[6], hipMemcpyDeviceToHost, s->stream);
    if (er == hipSuccess) er = hipStreamSynchronize(s->stream);
    if (er != hipSuccess) { set_err("mpc_costmap_to_obstacles", er); return er == hipErrorOutOfMemory ? MPC_ENOMEM : MPC_EHIP; }
    return rc;
}

int mpc_grid_update_device(mpc_solver* s, int32_t B, const double* d_x0_new, double* d_x, double* d_u, double* d_dt,
                           int32_t adapt, int32_t n_min, int32_t n_max, double dt_hyst_ratio) {
    g_err[0] = 0;
    if (!s || !d_x || !d_u || !d_dt) { set_err("mpc_grid_update_device: null argument"); return MPC_EINVAL; }
    if (B <= 0) return MPC_OK;
    if (B > s->max_batch) { set_err("mpc_grid_update_device: B exceeds max_batch"); return MPC_EBATCH; }
    if (s->use_ngrid && B > s->ngrid_B) { set_err("mpc_grid_update_device: B exceeds the batch the per-instance grid sizes were set for (mpc_set_grid_sizes)"); return MPC_EBATCH; }
    HIP_TRY(hipSetDevice(s->device));
    mpc::GridUpdateArgs a;
    memset(&a, 0, sizeof(a));
    a.x0 = d_x0_new; a.x = d_x; a.u = d_u; a.dt = d_dt; a.n_stride = s->cfg.n;
    a.dual = s->d_dual; a.dual_words = s->dual_words; a.dual_ns = s->WL.NS;
    if (!s->cfg.dt_free) {
        if (!d_x0_new) { set_err("mpc_grid_update_device: the fixed grid shifts towards the new start state (d_x0_new)"); return MPC_EINVAL; }
        a.mode = 0;
        a.n_grid = s->use_ngrid ? s->d_ngrid : nullptr;
    } else {
        if (!adapt) return MPC_OK;                      // variable grid without adaptation: nothing moves (x0 is overwritten by the solve)
        if (n_min < 3) n_min = 3;                       // the solver needs 3 grid points (the reference allows 2)
        if (n_max > s->cfg.n) n_max = s->cfg.n;
        if (!s->use_ngrid) {                            // first adaptation: every slot starts at the uniform size
            HIP_TRY(hipMemsetAsync(s->d_ngrid, 0, (size_t)s->max_batch * 4, s->stream));
            std::vector<int32_t> full((size_t)s->max_batch, s->cfg.n);
            HIP_TRY(hipMemcpyAsync(s->d_ngrid, full.data(), full.size() * 4, hipMemcpyHostToDevice, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
            s->use_ngrid = 1; s->ngrid_B = s->max_batch;
        }
        a.mode = 1; a.n_grid = s->d_ngrid; a.n_min = n_min; a.n_max = n_max; a.dt_ref = s->cfg.dt_ref; a.hyst = dt_hyst_ratio;
    }
    hipLaunchKernelGGL(mpc::grid_update_kernel, dim3(B), dim3(64), (size_t)s->cfg.n * 5 * 8, s->stream, a);
    HIP_TRY(hipGetLastError());
    return MPC_OK;
}

int mpc_get_grid_sizes(mpc_solver* s, int32_t B, int32_t* n_grid) {
    g_err[0] = 0;
    if (!s || !n_grid) return MPC_EINVAL;
    if (B <= 0) return MPC_OK;
    if (B > s->max_batch) { set_err("mpc_get_grid_sizes: B exceeds max_batch"); return MPC_EBATCH; }
    if (!s->use_ngrid) { for (int b = 0; b < B; ++b) n_grid[b] = s->cfg.n; return MPC_OK; }
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(n_grid, s->d_ngrid, (size_t)B * 4, hipMemcpyDeviceToHost));
    return MPC_OK;
}

int mpc_check_feasibility_device(mpc_solver* s, int32_t B, const double* d_x, const uint8_t* d_cost, int32_t size_x, int32_t size_y, double resolution,
                                 const double* d_origin, const double* footprint_spec, int32_t n_spec, double inscribed_radius,
                                 double min_resolution_collision_check_angular, int32_t look_ahead_idx, int32_t* d_feasible) {
    g_err[0] = 0;
    if (!s || !d_x || !d_cost || !d_origin || !d_feasible) { set_err("mpc_check_feasibility_device: null argument"); return MPC_EINVAL; }
    if (size_x < 1 || size_y < 1 || !(resolution > 0) || n_spec < 0 || n_spec > mpc::kFeasMaxSpec || (n_spec > 0 && !footprint_spec) ||
        !(inscribed_radius > 0) || !(min_resolution_collision_check_angular > 0)) {
        set_err("mpc_check_feasibility_device: bad costmap geometry, footprint (<= 32 points) or resolution parameters"); return MPC_EINVAL; }
    if (B <= 0) return MPC_OK;
    if (B > s->max_batch) { set_err("mpc_check_feasibility_device: B exceeds max_batch"); return MPC_EBATCH; }
    if (s->use_ngrid && B > s->ngrid_B) { set_err("mpc_check_feasibility_device: B exceeds the batch the per-instance grid sizes were set for"); return MPC_EBATCH; }
    HIP_TRY(hipSetDevice(s->device));
    mpc::FeasArgs a;
    memset(&a, 0, sizeof(a));
    a.x = d_x; a.n_grid = s->use_ngrid ? s->d_ngrid : nullptr; a.n_stride = s->cfg.n;
    a.cost = d_cost; a.origin = d_origin; a.size_x = size_x; a.size_y = size_y; a.resolution = resolution;
    a.n_spec = n_spec;
    for (int i = 0; i < 2 * n_spec; ++i) a.spec[i] = footprint_spec[i];
    a.inscribed_radius = inscribed_radius; a.min_res_angular = min_resolution_collision_check_angular; a.look_ahead_idx = look_ahead_idx;
    a.feasible = d_feasible;
    hipLaunchKernelGGL(mpc::feasibility_kernel, dim3(B), dim3(mpc::kFeasThreads), 0, s->stream, a);
    HIP_TRY(hipGetLastError());
    return MPC_OK;
}

int mpc_check_feasibility(mpc_solver* s, int32_t B, const double* x, const uint8_t* cost, int32_t size_x, int32_t size_y, double resolution,
                          const double* origin, const double* footprint_spec, int32_t n_spec, double inscribed_radius,
                          double min_resolution_collision_check_angular, int32_t look_ahead_idx, int32_t* feasible) {
    g_err[0] = 0;
    if (!s || !x || !cost || !origin || !feasible) { set_err("mpc_check_feasibility: null argument"); return MPC_EINVAL; }
    if (B <= 0) return MPC_OK;
    if (B > s->max_batch || size_x < 1 || size_y < 1) { set_err("mpc_check_feasibility: bad argument"); return MPC_EINVAL; }
    HIP_TRY(hipSetDevice(s->device));
    const size_t nb = B, n = s->cfg.n;
    const size_t sz[4] = {nb * n * 3 * 8, nb * size_x * size_y, nb * 2 * 8, nb * 4};
    void* d[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t er = stage_carve(s, sz, 4, d);
    int rc = MPC_OK;
    if (er == hipSuccess) er = hipMemcpyAsync(d[0], x, sz[0], hipMemcpyHostToDevice, s->stream);
    if (er == hipSuccess) er = hipMemcpyAsync(d[1], cost, sz[1], hipMemcpyHostToDevice, s->stream);
    if (er == hipSuccess) er = hipMemcpyAsync(d[2], origin, sz[2], hipMemcpyHostToDevice, s->stream);
    if (er == hipSuccess)
        rc = mpc_check_feasibility_device(s, B, (const double*)d[0], (const uint8_t*)d[1], size_x, size_y, resolution, (const double*)d[2], footprint_spec, n_spec,
                                          inscribed_radius, min_resolution_collision_check_angular, look_ahead_idx, (int32_t*)d[3]);
    if (er == hipSuccess && rc == MPC_OK) er = hipMemcpyAsync(feasible, d[3], sz[3], hipMemcpyDeviceToHost, s->stream);
    if (er == hipSuccess) er = hipStreamSynchronize(s->stream);
    if (er != hipSuccess) { set_err("mpc_check_feasibility", er); return er == hipErrorOutOfMemory ? MPC_ENOMEM : MPC_EHIP; }
    return rc;
}

int mpc_set_grid_sizes(mpc_solver* s, const int32_t* n_grid, int32_t B) {
    g_err[0] = 0;
    if (!s) return MPC_EINVAL;
    if (!n_grid) { s->use_ngrid = 0; return MPC_OK; }
    if (B <= 0 || B > s->max_batch) { set_err("mpc_set_grid_sizes: B out of range"); return MPC_EBATCH; }
    for (int b = 0; b < B; ++b)
        if (n_grid[b] < 3 || n_grid[b] > s->cfg.n) { set_err("mpc_set_grid_sizes: n_grid[b] must be in [3, cfg.n]"); return MPC_EINVAL; }
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpyAsync(s->d_ngrid, n_grid, (size_t)B * 4, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->use_ngrid = 1;
    s->ngrid_B = B;
    return MPC_OK;
}

int mpc_synchronize(mpc_solver* s) {
    if (!s) return MPC_EINVAL;
    HIP_TRY(hipStreamSynchronize(s->stream));
    return MPC_OK;
}

int mpc_lds_bytes(const mpc_solver* s, int64_t* bytes) {
    if (!s || !bytes) return MPC_EINVAL;
    *bytes = (int64_t)s->wave_lds;
    return MPC_OK;
}

int mpc_occupancy(mpc_solver* s, int32_t B, int32_t* workgroups_per_cu, int64_t* lds_bytes) {
    if (!s || !workgroups_per_cu || B <= 0) return MPC_EINVAL;
    g_err[0] = 0;
    HIP_TRY(hipSetDevice(s->device));
    const bool f32 = s->cfg.precision == MPC_FP32;
    mpc::SolveLaunch a{};
    a.level = !solver_ext(s) ? 0 : (s->P64.costx ? 2 : 1);
    a.w2 = !f32 && s->w2_ok && B >= s->w2_min_batch;
    const bool gs = a.w2 ? s->w2_gs : (f32 ? s->gs32 : s->gs64);
    a.L = gs ? s->WLg : s->WL;
    a.lds = f32 ? s->wave_lds32 : (a.w2 ? s->wave_lds_w2 : s->wave_lds);
    int occ = 0;
    HIP_TRY(pool_kernel_occupancy(f32, s->cfg.model, a, &occ));
    *workgroups_per_cu = occ;
    if (lds_bytes) *lds_bytes = (int64_t)a.lds;
    return MPC_OK;
}

int mpc_last_kernel_ms(mpc_solver* s, float* ms) {
    if (!s || !ms) return MPC_EINVAL;
    if (!s->timed) { *ms = 0.f; return MPC_OK; }
    HIP_TRY(hipEventSynchronize(s->ev1));
    HIP_TRY(hipEventElapsedTime(ms, s->ev0, s->ev1));
    return MPC_OK;
}

// host-pointer entry of one control cycle: `outer` x (grid update -> solve) without a host round trip in between
static int step_host(mpc_solver* s, int32_t B, const double* x0, const double* xf, const double* u_prev, const double* dt_prev,
                     const double* x_init, const double* u_init, const double* dt_init, const mpc_obstacles* obstacles,
                     int32_t outer, int32_t adapt, int32_t n_min, int32_t n_max, double dt_hyst_ratio,
                     double* x_out, double* u_out, double* dt_out, int32_t* status, int32_t* iters, int32_t* n_grid_out) {
    g_err[0] = 0;
    if (!s || !x0 || !xf || !x_out || !u_out || !dt_out) { set_err("mpc_solve_batch / mpc_step_batch: null argument"); return MPC_EINVAL; }
    if (B <= 0) return MPC_OK;
    if (B > s->max_batch) { set_err("mpc_solve_batch: B exceeds max_batch"); return MPC_EBATCH; }
    if ((x_init != nullptr) != (u_init != nullptr) || (x_init != nullptr) != (dt_init != nullptr)) {
        set_err("mpc_solve_batch: x_init, u_init and dt_init must be given together (all three or none)"); return MPC_EINVAL; }
    HIP_TRY(hipSetDevice(s->device));
    const size_t n = s->cfg.n, b = B;
    hipStream_t q = s->stream;
    const bool warm = x_init != nullptr;
    if (s->cfg.max_obstacles > 0 && (!obstacles || !obstacles->n_obstacles || !obstacles->n_vertices || !obstacles->vertices)) {
        set_err("mpc_solve_batch: the solver was created with max_obstacles > 0 but no obstacles were passed");
        return MPC_EINVAL;
    }
    // ---- pack every input into the pinned block (256-byte aligned pieces), ONE host-to-device copy
    size_t off = 0;
    auto put = [&](const void* src, size_t bytes) -> const unsigned char* {
        const unsigned char* dptr = s->d_in + off;
        memcpy(s->h_in + off, src, bytes);
        off = (off + bytes + 255) & ~(size_t)255;
        return dptr;
    };
    const double* dx0 = (const double*)put(x0, b * 3 * 8);
    const double* dxf = (const double*)put(xf, b * 3 * 8);
    const double* dup = u_prev ? (const double*)put(u_prev, b * 2 * 8) : nullptr;
    const double* ddtp = dt_prev ? (const double*)put(dt_prev, b * 8) : nullptr;
    const double *dxi = nullptr, *dui = nullptr, *ddti = nullptr;
    if (warm) {
        dxi = (const double*)put(x_init, b * n * 3 * 8);
        dui = (const double*)put(u_init, b * n * 2 * 8);
        ddti = (const double*)put(dt_init, b * 8);
    }
    mpc_obstacles dob = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (s->cfg.max_obstacles > 0) {
        const size_t O = s->cfg.max_obstacles, V = s->cfg.max_vertices > 0 ? s->cfg.max_vertices : 1;
        dob.n_obstacles = (const int32_t*)put(obstacles->n_obstacles, b * 4);
        dob.n_vertices = (const int32_t*)put(obstacles->n_vertices, b * O * 4);
        dob.vertices = (const double*)put(obstacles->vertices, b * O * V * 2 * 8);
        if (obstacles->radius) dob.radius = (const double*)put(obstacles->radius, b * O * 8);
        if (obstacles->velocity && s->cfg.enable_dynamic_obstacles) dob.velocity = (const double*)put(obstacles->velocity, b * O * 2 * 8);
    }
    if (off > s->in_cap) { set_err("mpc_solve_batch: internal staging overflow"); return MPC_EINVAL; }
    HIP_TRY(hipMemcpyAsync(s->d_in, s->h_in, off, hipMemcpyHostToDevice, q));
    // ---- outputs: one device block, ONE device-to-host copy
    size_t oo = 0;
    auto take = [&](size_t bytes) { size_t at = oo; oo = (oo + bytes + 255) & ~(size_t)255; return at; };
    const size_t o_x = take(b * n * 3 * 8), o_u = take(b * n * 2 * 8), o_dt = take(b * 8), o_st = take(b * 4), o_it = take(b * 4), o_ng = take(b * 4);
    int rc = mpc_step_batch_device(s, B, dx0, dxf, dup, ddtp, dxi, dui, ddti, &dob, outer, adapt, n_min, n_max, dt_hyst_ratio, (double*)(s->d_out + o_x),
                                   (double*)(s->d_out + o_u), (double*)(s->d_out + o_dt), (int32_t*)(s->d_out + o_st), (int32_t*)(s->d_out + o_it));
    if (rc != MPC_OK) return rc;
    if (n_grid_out && s->use_ngrid) HIP_TRY(hipMemcpyAsync(s->d_out + o_ng, s->d_ngrid, b * 4, hipMemcpyDeviceToDevice, q));
    HIP_TRY(hipMemcpyAsync(s->h_out, s->d_out, oo, hipMemcpyDeviceToHost, q));
    HIP_TRY(hipStreamSynchronize(q));
    memcpy(x_out, s->h_out + o_x, b * n * 3 * 8);
    memcpy(u_out, s->h_out + o_u, b * n * 2 * 8);
    memcpy(dt_out, s->h_out + o_dt, b * 8);
    if (status) memcpy(status, s->h_out + o_st, b * 4);
    if (iters) memcpy(iters, s->h_out + o_it, b * 4);
    if (n_grid_out) { if (s->use_ngrid) memcpy(n_grid_out, s->h_out + o_ng, b * 4); else for (size_t i = 0; i < b; ++i) n_grid_out[i] = s->cfg.n; }
    return MPC_OK;
}

int mpc_solve_batch(mpc_solver* s, int32_t B, const double* x0, const double* xf, const double* u_prev, const double* dt_prev,
                    const double* x_init, const double* u_init, const double* dt_init, const mpc_obstacles* obstacles,
                    double* x_out, double* u_out, double* dt_out, int32_t* status, int32_t* iters) {
    return step_host(s, B, x0, xf, u_prev, dt_prev, x_init, u_init, dt_init, obstacles, 1, 0, 0, 0, 0.0, x_out, u_out, dt_out, status, iters, nullptr);
}

int mpc_step_batch(mpc_solver* s, int32_t B, const double* x0, const double* xf, const double* u_prev, const double* dt_prev,
                   const double* x_init, const double* u_init, const double* dt_init, const mpc_obstacles* obstacles,
                   int32_t outer_iterations, int32_t adapt, int32_t n_min, int32_t n_max, double dt_hyst_ratio,
                   double* x_out, double* u_out, double* dt_out, int32_t* status, int32_t* iters, int32_t* n_grid_out) {
    return step_host(s, B, x0, xf, u_prev, dt_prev, x_init, u_init, dt_init, obstacles, outer_iterations, adapt, n_min, n_max, dt_hyst_ratio, x_out, u_out, dt_out, status,
                     iters, n_grid_out);
}

}  // extern "C"
