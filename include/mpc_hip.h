/*
 * mpc_hip.h -- C ABI of the MI355X-native batched receding-horizon NLP solve.
 *
 * This is the drop-in boundary for ONE hot path of rst-tu-dortmund/mpc_local_planner:
 * the NLP solve that Controller::step() triggers every control cycle
 * (reference: mpc_local_planner/src/controller.cpp:111-179 -> :172
 *  corbo::PredictiveController::step -> StructuredOptimalControlProblem::compute
 *  -> SolverIpopt::solve; grid/edges: src/optimal_control/finite_differences_grid_se2.cpp:36-154).
 *
 * Plain C, plain pointers and sizes, no C++/torch types.  All functions return 0 on
 * success or a negative MPC_E* code; nothing throws across the ABI.  A solver handle
 * is thread-compatible (one handle per thread / per GPU), exactly like the reference's
 * Controller, which is not re-entrant (include/mpc_local_planner/controller.h:118-142).
 *
 * Array layouts at the ABI are instance-major ("AoS", what a caller holding B
 * independent planners naturally has):
 *   x0[B][3], xf[B][3], u_prev[B][2], dt_prev[B]
 *   x_init / x_out [B][n][3]   states  x_0 .. x_{n-2}, xf   (getStateAndControlTimeSeries,
 *   u_init / u_out [B][n][2]   controls u_0 .. u_{n-2} + duplicate of the last
 *                              (src/optimal_control/full_discretization_grid_base_se2.cpp:579-615)
 *   dt_init / dt_out [B]
 *   status[B], iters[B]
 * Inside the solver everything is re-laid out instance-minor (SoA) for coalesced HBM access.
 */
#ifndef MPC_HIP_H_
#define MPC_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums (values shared with oracle/se2_nlp.py) ------------------------------------ */
enum mpc_model {                      /* include/mpc_local_planner/systems/ */
    MPC_MODEL_UNICYCLE = 0,           /* unicycle_robot.h:59-68 */
    MPC_MODEL_SIMPLE_CAR = 1,         /* simple_car.h:68-77 (rear wheel) */
    MPC_MODEL_SIMPLE_CAR_FRONT = 2,   /* simple_car.h:131-141 */
    MPC_MODEL_KINEMATIC_BICYCLE = 3   /* kinematic_bicycle_model.h:65-77 */
};
enum mpc_collocation {                /* include/.../optimal_control/fd_collocation_se2.h */
    MPC_COLLOC_FORWARD = 0,           /* :54-69 (the default, src/controller.cpp:298) */
    MPC_COLLOC_MIDPOINT = 1,          /* :91-108 midpoint_differences */
    MPC_COLLOC_CRANK_NICOLSON = 2     /* :130-147 crank_nicolson_differences, restated literally: 1.5 f(x_{k+1}) + 0.5 f(x_k) */
};
enum mpc_objective {                  /* src/controller.cpp:551-640 */
    MPC_OBJ_MIN_TIME = 0,
    MPC_OBJ_QUADRATIC = 1,
    MPC_OBJ_MIN_TIME_VIA_POINTS = 2     /* planning/objective/type minimum_time_via_points (src/controller.cpp:597-612) */
};
enum mpc_precision { MPC_FP64 = 0, MPC_FP32 = 1,
                     MPC_MIXED = 2      /* fp32 main phase (to tol 1e-4; the candidates run here) + fp64 refinement started from its iterate AND its
                                         * multipliers (barrier 1e-5, at most 40 iterations) down to cfg.tol: fp64-accurate results at roughly the fp32 residency */ };
enum mpc_footprint { MPC_FOOTPRINT_POINT = 0, MPC_FOOTPRINT_CIRCLE = 1,     /* every footprint works with every obstacle kind (point, circle, line, polygon), static or dynamic */
                     MPC_FOOTPRINT_LINE = 2,        /* teb LineRobotFootprint (the car-like example's footprint) */
                     MPC_FOOTPRINT_TWO_CIRCLES = 3, /* teb TwoCirclesRobotFootprint */
                     MPC_FOOTPRINT_POLYGON = 4      /* teb PolygonRobotFootprint */ };

/* Candidate initial trajectories of one planner instance (BASELINE north star: "batches of independent planner instances (and
 * candidate initial trajectories)").  Every candidate is a complete solve of the same NLP from its own initial vertex values. */
enum mpc_candidate_kind {
    MPC_CAND_REFERENCE = 0,           /* what Controller::step builds: x_init/u_init/dt_init when given, else the 2-pose-plan cold start
                                       * (src/controller.cpp:807-857 + full_discretization_grid_base_se2.cpp:192-239) */
    MPC_CAND_TRAVEL = 1,              /* initializeSequences without xinit (full_discretization_grid_base_se2.cpp:136-190): straight line, heading =
                                       * direction of travel, turned by pi when the goal lies behind the start pose */
    MPC_CAND_TRAVEL_REVERSE = 2,      /* the same with the other driving direction (ours) */
    MPC_CAND_BLEND = 3,               /* ours: MPC_CAND_TRAVEL with the heading blended from the start heading over the first candidate_blend
                                       * grid points and into the goal heading over the last candidate_blend grid points */
    MPC_CAND_BLEND_REVERSE = 4,       /* ours: the same around MPC_CAND_TRAVEL_REVERSE */
    /* ours: positions on the cubic Hermite curve from the start pose to the goal pose whose end tangents point along the two headings, scaled by
     * candidate_param * |goal - start| (0 -> 2.0) and signed by the driving direction at that end (F forward, R reverse: first letter = at the
     * start, second = at the goal); heading = tangent direction, + pi where the robot drives backwards (first half of the horizon: the start's
     * direction, second half: the goal's).  A kinematically plausible guess: on config 2 it converges in <= 60 iterations from 98 % of the
     * cold starts (reference guess: 84 %). */
    MPC_CAND_HERMITE_FF = 5, MPC_CAND_HERMITE_RR = 6, MPC_CAND_HERMITE_FR = 7, MPC_CAND_HERMITE_RF = 8
};
#define MPC_MAX_CANDIDATES 4

enum mpc_cost_integration { MPC_COST_LEFT_SUM = 0, MPC_COST_TRAPEZOIDAL = 1 };

enum mpc_hessian_mode {
    MPC_HESSIAN_EXACT = 0,            /* exact Lagrangian Hessian (analytic), delta_w raised until the factorisation has the inertia (n, m, 0) (Ipopt's test): the default */
    MPC_HESSIAN_CONVEXIFIED = 1       /* every stage block of the constraint curvature lam' D over (theta, v, w, dt) replaced by its positive semidefinite
                                       * part: hardly any regularisation retries (1.03 instead of 1.2 factorisations per iteration) but linear instead of
                                       * quadratic local convergence.  With tol = 1e-4 this is the "reference-like" setting: the car-like example runs
                                       * Ipopt with tol 1e-4 and a limited-memory (positive definite) Hessian because the exact one "does currently not
                                       * work well with the carlike model" (cfg/carlike/mpc_local_planner_params.yaml:91-95). */
};

enum mpc_stage_data {                 /* mpc_config.stage_data */
    MPC_STAGE_AUTO = 0,               /* decided per handle and precision at mpc_create.  fp64: global memory when the LDS form leaves at least half of a CU's SIMDs without a wave and
                                       * the global form fills more of them (bit-identical results either way).  Plain fp32: already when the LDS form leaves one of four empty --
                                       * there the two forms agree to rounding only, so an fp32 handle reproduces itself run to run but not the results of library versions
                                       * before 0.5.0 (which kept fp32 in LDS).  Both phases of MPC_MIXED keep the LDS form */
    MPC_STAGE_LDS = 1,                /* the whole working set of an instance in LDS (97 words per grid point) */
    MPC_STAGE_GLOBAL = 2              /* stage records and gains in a per-workgroup block of global memory (L2 / Infinity-Cache resident), 34 words per grid point in LDS;
                                       * every kernel level (r06: the extended terms too -- terminal ball, via-points, turning footprints, dynamic obstacles, cost variants) */
};

enum mpc_mu_strategy {
    MPC_MU_ADAPTIVE = 0,              /* the default (corbo's SolverIpopt is believed to set mu_strategy adaptive): every iteration
                                       * mu = sigma x (average complementarity), sigma = clamp((1 - min(alpha, alpha_dual))^3, 0.05, 1) from the step lengths the
                                       * previous iteration achieved (Mehrotra's sigma = (mu_aff / mu)^3 read off the step that was actually taken, no second solve),
                                       * never below min(mu, 0.03 x the optimality error), inside [tol / 10, 1000 mu_init].  Against the monotone rule on the BASELINE
                                       * workloads: 3 .. 8 % fewer iterations and 0 .. 2.5 points more converged instances from the reference start */
    MPC_MU_MONOTONE = 1               /* Fiacco-McCormick: mu falls (x 0.2 / ^1.5) when the barrier subproblem is solved to 10 mu -- Ipopt's own default mu_strategy */
};

enum mpc_line_search {                /* the globalisation of the interior-point iteration (Ipopt option line_search_method; solver/ipopt/ipopt_string_options, src/controller.cpp:407-418) */
    MPC_LS_DEFAULT = 0,               /* the library's default: MPC_LS_FILTER since 0.6.0 (mpc_problem.hpp::kDefaultLineSearch; DESIGN.md section 3.1a has the measurements) */
    MPC_LS_MERIT = 1,                 /* backtracking on the l1 merit function f - mu sum log + rho theta with Ipopt's penalty-parameter rule (the globalisation of rounds 1-5) */
    MPC_LS_FILTER = 2                 /* Ipopt's own default: the filter line search of Waechter & Biegler (2006), Algorithm A, with its published constants (gamma_theta 1e-5,
                                       * gamma_phi 1e-8, s_phi 2.3, s_theta 1.1, eta_phi 1e-8, delta 1, gamma_alpha 0.05, theta_max / theta_min = 1e4 / 1e-4 x max(1, theta_0));
                                       * the filter holds the last 16 (theta, phi) pairs of the current barrier problem and is emptied when mu changes.  Not restated: the
                                       * second-order correction, and the restoration phase -- when every trial step is refused the filter is emptied and the shortest trial
                                       * step taken (clearance rows have their own restoration) */
};

enum mpc_status {                     /* per-instance result; 0 == what corbo reports as Converged */
    MPC_CONVERGED = 0,
    MPC_MAX_ITER = 1,
    MPC_LINESEARCH_FAILED = 2,
    MPC_LINSOLVE_FAILED = 3,
    MPC_NUMERICAL_ERROR = 4,
    MPC_TIME_LIMIT = 5                /* mpc_config.max_time_us ran out (Ipopt: Maximum_CpuTime_Exceeded; a failure for the reference's wrapper) */
};

enum mpc_error {
    MPC_OK = 0,
    MPC_EINVAL = -1,                  /* bad argument / unsupported configuration */
    MPC_ENODEV = -2,                  /* no usable HIP device */
    MPC_ENOMEM = -3,
    MPC_EHIP = -4,                    /* a HIP runtime call failed (see mpc_last_error) */
    MPC_EBATCH = -5                   /* B exceeds the capacity given to mpc_create */
};

/* ---- configuration = the parameter set read in src/controller.cpp:225-805 -------------- */
typedef struct mpc_config {
    int32_t model;                    /* robot/type                       (:346) */
    double  model_params[4];          /* wheelbase | (lr, lf)             (:355,:366-369) */
    int32_t n;                        /* grid/grid_size_ref               (:274) */
    double  dt_ref;                   /* grid/dt_ref                      (:278) */
    int32_t dt_free;                  /* grid/variable_grid/enable        (:236) */
    double  dt_lb, dt_ub;             /* .../min_dt, max_dt               (:242,:244) */
    int32_t xf_fixed[3];              /* grid/xf_fixed                    (:282) */
    int32_t collocation;              /* grid/collocation_method          (:298) */
    int32_t objective;                /* planning/objective/type          (:551) */
    double  Q[3], R[2];               /* quadratic_form weights: diagonals (:561-592); off-diagonal terms: Q_offdiag, R_offdiag below */
    int32_t integral_form;            /* .../integral_form                (:594) */
    int32_t has_Qf;                   /* planning/terminal_cost/type == quadratic (:645); applies with EVERY objective type, on the goal components
                                       * that are free (the edge exists while the final state is not completely fixed, finite_differences_grid_se2.cpp:128-133) */
    double  Qf[3];                    /* final_state_weights: diagonal    (:652-668); off-diagonal terms: Qf_offdiag */
    double  u_lb[2], u_ub[2];         /* control box                      (:511,:527,:543) */
    double  du_lb[2], du_ub[2];       /* control-rate box; +-1e30 = inf   (:756-797) */
    /* solver (replaces solver/ipopt/..., :388-421) */
    int32_t max_iter;                 /* iterations                       (:391) */
    double  tol;                      /* ipopt_numeric_options/tol */
    double  mu_init;                  /* barrier start (0 -> default 0.1) */
    int32_t precision;                /* MPC_FP64 | MPC_FP32 | MPC_MIXED */
    /* collision avoidance (src/controller.cpp:717-729; footprint: src/mpc_local_planner_ros.cpp:890-1001) */
    double  min_obstacle_dist;        /* collision_avoidance/min_obstacle_dist */
    double  force_inclusion_dist;     /* .../force_inclusion_dist */
    double  cutoff_dist;              /* .../cutoff_dist */
    int32_t footprint_kind;           /* MPC_FOOTPRINT_POINT | _CIRCLE | _LINE | _TWO_CIRCLES | _POLYGON */
    double  footprint_radius;         /* circular footprint radius */
    int32_t max_obstacles;            /* O: obstacles per instance the solver is sized for (0 = none) */
    int32_t max_vertices;             /* V: vertices per obstacle (1 point, 2 line, >=3 polygon) */
    int32_t max_obstacle_rows;        /* M: clearance rows kept per grid point (default 4, <= 16).  The reference keeps EVERY obstacle closer than
                                       * force_inclusion_dist plus the nearest one on the left and on the right (stage_inequality_se2.cpp:99-160);
                                       * here the kept rows are: dynamic obstacles, forced ones in container order -- the M CLOSEST of them when they do
                                       * not all fit --, then left, right.  mpc_last_rows_dropped tells how many did not fit. */
    double  mu_init_warm;             /* barrier start of a solve that is given an initial guess (x_init != NULL); 0 -> mu_init.
                                       * Closed-loop cycles start next to a solution: 1e-2 saves ~25 % of the iterations. */
    int32_t terminal_ball;            /* planning/terminal_constraint/type == l2_ball (src/controller.cpp:683); the row exists only when at
                                       * least one goal component is free (finite_differences_grid_se2.cpp:128-143) */
    double  terminal_ball_S[3];       /* .../l2_ball/weight_matrix: diagonal  (:686-692); off-diagonal terms: terminal_ball_S_offdiag */
    double  terminal_ball_gamma;      /* .../l2_ball/radius: row  xd' S xd - gamma <= 0  (final_state_conditions_se2.cpp:54-64) */
    double  vp_position_weight;       /* objective/minimum_time_via_points/position_weight    (src/controller.cpp:601) */
    double  vp_orientation_weight;    /* .../orientation_weight (:603); as coded the term is LINEAR in the heading error
                                       * (src/optimal_control/min_time_via_points_cost.cpp:139-142) */
    int32_t via_points_ordered;       /* .../via_points_ordered  (:605) */
    int32_t max_via_points;           /* via-points per instance the solver is sized for (objective MIN_TIME_VIA_POINTS; <= 64) */
    int32_t footprint_n_vertices;     /* MPC_FOOTPRINT_POLYGON: footprint_model/vertices (<= 16), robot frame */
    double  footprint_vertices[32];   /* x0, y0, x1, y1, ... */
    int32_t enable_dynamic_obstacles; /* collision_avoidance/enable_dynamic_obstacles (src/controller.cpp:721) */
    double  footprint_params[4];      /* MPC_FOOTPRINT_LINE: footprint_model/line_start (x, y), line_end (x, y) in the robot frame;
                                       * MPC_FOOTPRINT_TWO_CIRCLES: front_offset, front_radius, rear_offset, rear_radius (src/mpc_local_planner_ros.cpp:900-960) */
    /* candidate initial trajectories (0 or 1 = a single solve from MPC_CAND_REFERENCE, the reference's behaviour).  With n_candidates > 1 every
     * instance is solved from each candidate as its own workgroup of the same launch; RULE (deterministic, timing-independent): the candidate
     * with the LOWEST index that converges within its own iteration cap supplies the result; without any, candidate 0's last iterate and
     * status are returned.  Candidate 0 should be MPC_CAND_REFERENCE so that every instance the reference path solves keeps that answer.
     * Lower-priority candidates stop at their next iteration once a higher-priority one has converged (hedging: they only cost time
     * on SIMDs that would otherwise idle while the slow instances of a batch finish). */
    int32_t n_candidates;
    int32_t candidate_kind[MPC_MAX_CANDIDATES];
    int32_t candidate_max_iter[MPC_MAX_CANDIDATES];   /* iteration cap of candidate c (0 -> max_iter) */
    int32_t candidate_blend;          /* grid points of the heading blend of MPC_CAND_BLEND* (0 -> 8) */
    /* warm start of the multipliers (the reference's Ipopt keeps its multipliers between control cycles: warm_start_init_point in corbo's
     * SolverIpopt).  With dual_warm_start != 0 the handle keeps, per instance slot b, the collocation multipliers and the multipliers of the
     * control / dt boxes and the control-rate rows of the last CONVERGED solve; a later solve of slot b that is given an initial guess starts
     * from them -- every inequality multiplier max(previous, mu0 / slack), slacks re-derived from the new point, mu0 = mu_init_dual -- as
     * long as the slot's grid size is unchanged.  mpc_reset forgets them. */
    int32_t dual_warm_start;
    double  mu_init_dual;             /* barrier start of such a solve (0 -> 1e-3) */
    double  candidate_param[MPC_MAX_CANDIDATES];      /* per candidate: tangent scale of the MPC_CAND_HERMITE_* kinds (0 -> 2.0) */
    int32_t hessian_mode;             /* MPC_HESSIAN_EXACT | MPC_HESSIAN_CONVEXIFIED (solver/ipopt/ipopt_string_options/hessian_approximation, :407-418) */
    int32_t hybrid_cost_minimum_time; /* planning/objective/quadratic_form/hybrid_cost_minimum_time (src/controller.cpp:596,616-618): with objective
                                       * MPC_OBJ_QUADRATIC the cost is minimum time PLUS the quadratic form, (n-1) dt + sum ...  (corbo::MinTimeQuadraticControls;
                                       * the reference only takes this branch when the state weights are zero -- the reader of mpc_params.hpp mirrors that) */
    int32_t cost_integration;         /* grid/cost_integration_method (:318-333): MPC_COST_LEFT_SUM | MPC_COST_TRAPEZOIDAL.  Only integral-form terms are
                                       * integrated (integral_form = 1): trapezoidal = 0.5 dt (l(x_k, u_k) + l(x_{k+1}, u_k)) per interval
                                       * (corbo::TrapezoidalIntegralCostEdge, finite_differences_grid_se2.cpp:63-68) */
    int32_t mu_strategy;              /* solver/ipopt/ipopt_string_options/mu_strategy: MPC_MU_ADAPTIVE (0, the default) | MPC_MU_MONOTONE */
    int32_t stage_data;               /* where a solve keeps its factorisation data (stage records + Riccati gains, 63 of the 97 words per grid point): MPC_STAGE_AUTO (0: global
                                       * memory exactly when that puts more workgroups on a compute unit -- long horizons, clearance rows), MPC_STAGE_LDS, MPC_STAGE_GLOBAL.
                                       * Results are bit-identical either way in fp64 (in fp32 the two forms agree to rounding); no counterpart in the reference (its solver's
                                       * working memory is Ipopt's) */
    int32_t two_wave_min_batch;       /* launches with at least this many instances run the kernel variant for TWO resident waves per SIMD (236 registers, no scratch; fp64, headline
                                       * kernel level without clearance rows, and only where the LDS record fits eight times into a compute unit: about n <= 24 grid points -- the
                                       * grid sizes of the reference's shipped parameter files).  0 -> the default 4096 (measured on the MI355X for n = 12 / 20 / 24: x1.03 / x1.0 / x1.3 there, x1.06 / x1.4 / x1.37
                                       * at 8192, x1.46 / x1.64 / x1.6 at 32768 instances; below the threshold n = 12 loses -- x0.83 at 2048 -- and n = 20 / 24 gain x1.13 / x1.29: a small
                                       * launch lasts as long as its slowest wave, which runs fastest alone); negative -> never.  Results are bit-identical either way.  No counterpart in the reference */
    int32_t line_search;              /* enum mpc_line_search (solver/ipopt/ipopt_string_options/line_search_method) */
    /* full weight matrices (state_weights / control_weights / final_state_weights / weight_matrix given as n x n lists, column major,
     * src/controller.cpp:565-573,580-588,656-664,690-698): Q, R, Qf, terminal_ball_S above hold the DIAGONALS, these the off-diagonal terms
     * (0,1), (0,2), (1,2) of the symmetric parts (x' W x only sees (W + W') / 2); all zero = diagonal weights */
    double  Q_offdiag[3];
    double  R_offdiag;
    double  Qf_offdiag[3];
    double  terminal_ball_S_offdiag[3];
    /* Ipopt's "solved to acceptable level", which the reference's wrapper counts as success (src/controller.cpp:388-421 configures corbo's
     * SolverIpopt; its result is success for Converged and for EarlyTerminated alike).  Two halves, both at the level acceptable_tol:
     *   - acceptable_iter iterations in a row with a scaled optimality error of at most acceptable_tol end the solve with MPC_CONVERGED;
     *   - when the line search refuses every trial step, or accepts only one below 1e-6 of the fraction-to-boundary step, at a point whose error
     *     is at most acceptable_tol, the solve ends THERE (neither the point nor the multipliers move) with MPC_CONVERGED.
     * Without it the tiny grids the plugin reaches close to its goal (n = 4..10 after grid adaptation) stall a factor ~1.2 above tol = 1e-8 at
     * the rounding level of the merit function and end in MPC_MAX_ITER (the recorded to-the-goal run of the reference's plugin then answers
     * NO_VALID_CMD).  The headline workloads are untouched by it (bit-identical trajectories / iteration counts). */
    double  acceptable_tol;           /* solver/ipopt/ipopt_numeric_options/acceptable_tol: 0 -> Ipopt's default 1e-6, < 0 -> rule off */
    int32_t acceptable_iter;          /* .../ipopt_integer_options/acceptable_iter: 0 -> Ipopt's default 15, < 0 -> counting half off */
    /* solver/ipopt/max_cpu_time (src/controller.cpp:395-397 -> SolverIpopt::setMaxCpuTime): wall-clock budget of ONE solve in microseconds, measured on the device from
     * the moment the solve's wavefront starts (every candidate initial trajectory has its own); tested once per interior-point iteration, so a solve overruns by at most one
     * iteration (~45 us at n = 50).  0 = no limit (the reference's default -1).  A solve that runs out ends with MPC_TIME_LIMIT and returns its last iterate.  By nature the one
     * setting whose results depend on the machine's load; everything stays bit-reproducible with 0. */
    int32_t max_time_us;
} mpc_config;

/* Obstacles of a batch (teb_local_planner ObstContainer of every instance, flattened; borrowed for the call).
 * Point: 1 vertex; line: 2 vertices; polygon: >= 3 vertices (closed loop, any orientation); an optional radius
 * turns a 1-vertex obstacle into a circle.  With mpc_config.enable_dynamic_obstacles an obstacle with a non-zero centroid velocity is
 * a DYNAMIC obstacle: it is kept at every grid point and its row at grid point k is evaluated against the obstacle moved by k*dt*velocity
 * (src/optimal_control/stage_inequality_se2.cpp:99-106,177-189). */
typedef struct mpc_obstacles {
    const int32_t* n_obstacles;       /* [B]                 number of valid obstacles of instance b (<= O) */
    const int32_t* n_vertices;        /* [B][O]              number of valid vertices of obstacle (b,o) (<= V) */
    const double*  vertices;          /* [B][O][V][2] */
    const double*  radius;            /* [B][O] or NULL */
    const double*  velocity;          /* [B][O][2] centroid velocities or NULL (all static) */
} mpc_obstacles;

typedef struct mpc_solver mpc_solver;     /* opaque */

/* Fill *cfg with the in-code defaults of src/controller.cpp (unicycle, n=20, dt_ref=.3,
 * variable grid, min-time, xf fixed, no rate limits, 100 iterations). */
void mpc_config_defaults(mpc_config* cfg);

/* Create a solver for one (model, objective, n, flags) tuple on HIP device `device`
 * with room for `max_batch` instances.  Fails with MPC_ENODEV when no GPU is present:
 * there is NO CPU fallback.  Replaces Controller::configure (include/.../controller.h:61-62).
 * Device memory of a handle besides the per-instance inputs / outputs: with the factorisation data in global memory (mpc_config.stage_data; also every handle with
 * clearance rows, for their elastic arrays) a pool of blocks per XCD -- 8 pools x 2 x (resident workgroups per CU, from the occupancy API) x (CUs of an XCD) blocks of
 * ~63 n words: 150 MB at n = 120 in fp64 on an MI355X, independent of max_batch; fleets with several such handles add it up. */
int mpc_create(const mpc_config* cfg, int32_t max_batch, int32_t device, mpc_solver** out);

/* Replaces Controller::reset (controller.h:104): waits for the stream and returns the handle's per-instance state (candidate
 * bookkeeping, the duals kept for warm starts, the claim words of the factorisation-data blocks of mpc_config.stage_data) to its initial values -- all of it restores
 * itself at the end of every launch, so this only matters after a launch that was aborted. */
int mpc_reset(mpc_solver* s);

void mpc_destroy(mpc_solver* s);

/* One control cycle for B independent planner instances -- the batched equivalent of
 * Controller::step (include/.../controller.h:63-67; src/controller.cpp:111-179).
 * HOST pointers; the call copies in, solves, copies out and returns when done.
 *   x_init/u_init/dt_init : nullable, all three or none (a partial triple is MPC_EINVAL).  NULL -> cold start built on the device exactly as the
 *       reference does for a 2-pose plan (src/controller.cpp:807-857 +
 *       full_discretization_grid_base_se2.cpp:192-239: linear x0->xf, shortest-arc theta, u=0,
 *       dt=dt_ref).  Non-NULL -> used as the vertex values (warm start), with x_0 := x0 and the
 *       fixed goal components := xf (full_discretization_grid_base_se2.cpp:101-110).
 *   obstacles : nullable.  When given, the relevant obstacles of every grid point are associated on the device
 *       exactly as StageInequalitySE2::update does (src/optimal_control/stage_inequality_se2.cpp:50-162) on the
 *       initial vertex values, then frozen for the solve; rows d_min - dist(footprint(x_k), obstacle) <= 0
 *       (stage_inequality_se2.cpp:164-175) for k = 1..n-2.
 *   status/iters : nullable. */
int mpc_solve_batch(mpc_solver* s, int32_t B,
                    const double* x0, const double* xf, const double* u_prev, const double* dt_prev,
                    const double* x_init, const double* u_init, const double* dt_init,
                    const mpc_obstacles* obstacles /* nullable; host pointers inside */,
                    double* x_out, double* u_out, double* dt_out,
                    int32_t* status, int32_t* iters);

/* Same contract with DEVICE pointers (HBM-resident inputs/outputs), asynchronous on the
 * solver's stream; pair with mpc_synchronize.  This is what bench.py times.
 * The solver's stream is a NON-BLOCKING one (hipStreamNonBlocking): it is not ordered against the null stream or any
 * other stream of the caller.  Work of the caller that produces the inputs -- or still writes the output buffers, e.g. a
 * fill enqueued on another stream -- has to be complete (stream / event synchronised) before this call. */
int mpc_solve_batch_device(mpc_solver* s, int32_t B,
                           const double* d_x0, const double* d_xf, const double* d_u_prev, const double* d_dt_prev,
                           const double* d_x_init, const double* d_u_init, const double* d_dt_init,
                           const mpc_obstacles* d_obstacles /* nullable; DEVICE pointers inside */,
                           double* d_x_out, double* d_u_out, double* d_dt_out,
                           int32_t* d_status, int32_t* d_iters);

/* One control cycle of B planners = what Controller::step does through corbo's PredictiveController (src/controller.cpp:70-72,172): the OCP -- grid update,
 * then solve -- is repeated controller/outer_ocp_iterations times, every repetition after the first starting from the solution just computed.  The first
 * solve is mpc_solve_batch with the given initial guess (NULL = cold start); before each further one the grid update of mpc_grid_update_device runs on the
 * outputs in place on the VARIABLE grid (adapt != 0: single-step adaptation + resampling with n_min / n_max / dt_hyst_ratio); on the fixed grid nothing runs
 * between the solves: its warm-start shift belongs to the first outer iteration of a cycle only (`new_run`, full_discretization_grid_base_se2.cpp:96-100).  Everything is enqueued on the solver's stream without a host round trip in between; the result is bit for
 * bit what the separate calls give.  status / iters: those of the LAST solve (what the reference's step() returns); n_grid_out (nullable, HOST variant):
 * the grid sizes in force after the call. */
int mpc_step_batch(mpc_solver* s, int32_t B,
                   const double* x0, const double* xf, const double* u_prev, const double* dt_prev,
                   const double* x_init, const double* u_init, const double* dt_init, const mpc_obstacles* obstacles,
                   int32_t outer_iterations, int32_t adapt, int32_t n_min, int32_t n_max, double dt_hyst_ratio,
                   double* x_out, double* u_out, double* dt_out, int32_t* status, int32_t* iters, int32_t* n_grid_out);
int mpc_step_batch_device(mpc_solver* s, int32_t B,
                          const double* d_x0, const double* d_xf, const double* d_u_prev, const double* d_dt_prev,
                          const double* d_x_init, const double* d_u_init, const double* d_dt_init, const mpc_obstacles* d_obstacles,
                          int32_t outer_iterations, int32_t adapt, int32_t n_min, int32_t n_max, double dt_hyst_ratio,
                          double* d_x_out, double* d_u_out, double* d_dt_out, int32_t* d_status, int32_t* d_iters);

/* Per-instance grid sizes for the following mpc_solve_batch* calls (grid adaptation of the variable grid,
 * src/optimal_control/finite_differences_variable_grid_se2.cpp:99-121): instance b uses n_grid[b] grid points
 * (3 <= n_grid[b] <= cfg.n); the array layouts keep the stride cfg.n and only the first n_grid[b] rows of
 * x_init/u_init/x_out/u_out are meaningful.  HOST pointer, copied; NULL restores the uniform size cfg.n. */
int mpc_set_grid_sizes(mpc_solver* s, const int32_t* n_grid, int32_t B);

/* The grid update between two control cycles for a whole batch, on the device and in place on the previous solve's outputs
 * (d_x / d_u / d_dt = that solve's x_out / u_out / dt_out, handed back as x_init / u_init / dt_init of the next solve), so that a batched
 * closed loop needs no host round trip:
 *   fixed grid (cfg.dt_free == 0): warmStartShifting + findNearestState (src/optimal_control/full_discretization_grid_base_se2.cpp:241-339)
 *       towards the new start states d_x0_new [B][3]; multipliers kept by dual_warm_start move with the trajectory;
 *   variable grid with adapt != 0: adaptGridTimeBasedSingleStep (src/optimal_control/finite_differences_variable_grid_se2.cpp:99-121) with
 *       n_min (clamped to 3), n_max (clamped to cfg.n), dt_hyst_ratio, + resampleTrajectory (...grid_base_se2.cpp:440-524); the
 *       per-instance grid sizes live in the handle (as after mpc_set_grid_sizes) and are read back with mpc_get_grid_sizes;
 *   variable grid with adapt == 0: nothing to do (no shifting on the variable grid, finite_differences_variable_grid_se2.h:85).
 * DEVICE pointers; runs on the solver's stream. */
int mpc_grid_update_device(mpc_solver* s, int32_t B, const double* d_x0_new, double* d_x, double* d_u, double* d_dt,
                           int32_t adapt, int32_t n_min, int32_t n_max, double dt_hyst_ratio);
/* Grid sizes currently in force (HOST pointer): cfg.n everywhere unless mpc_set_grid_sizes / mpc_grid_update_device changed them. */
int mpc_get_grid_sizes(mpc_solver* s, int32_t B, int32_t* n_grid);

/* Via-points of the minimum_time_via_points objective (the reference's borrowed ViaPointContainer,
 * include/mpc_local_planner/optimal_control/min_time_via_points_cost.h:102, refilled by the planner between steps,
 * src/mpc_local_planner_ros.cpp:619-635): instance b has n_via[b] (<= cfg.max_via_points) poses via[b][p] = (x, y, theta).
 * Every solve attaches each via-point to its closest grid point of the trajectory the solve STARTS from
 * (MinTimeViaPointsCost::update, src/optimal_control/min_time_via_points_cost.cpp:39-117; runs on the device).
 * HOST pointers, copied; they stay in force until the next call; NULL clears them (plain minimum time). */
int mpc_set_via_points(mpc_solver* s, int32_t B, const int32_t* n_via, const double* via /* [B][max_via_points][3] */);
/* Same with DEVICE pointers, borrowed until the next mpc_set_via_points* call. */
int mpc_set_via_points_device(mpc_solver* s, const int32_t* d_n_via, const double* d_via);

/* Costmap -> point obstacles, the step in front of the solve (MpcLocalPlannerROS::updateObstacleContainerWithCostmap,
 * src/mpc_local_planner_ros.cpp:474-499): every LETHAL (254) cell of instance b's local costmap cost[b][my][mx]
 * (costmap_2d layout, cell (mx,my) at index my*size_x + mx; the last row and column are not visited, as in the reference)
 * becomes a point obstacle at origin[b] + (m + 0.5) * resolution, unless it lies behind the robot pose[b] = (x, y, theta)
 * and farther away than behind_robot_dist (collision_avoidance/costmap_obstacles_behind_robot_dist, default 1.5).
 * Output in the mpc_obstacles layout of THIS solver (capacity cfg.max_obstacles, vertex stride cfg.max_vertices), in the
 * reference's container order (mx outer, my inner); dropped[b] (nullable) = cells that did not fit.  The result can be
 * handed to mpc_solve_batch_device as is (radius = NULL); converter / custom obstacles are appended by the caller
 * behind n_obstacles[b].  DEVICE pointers; runs on the solver's stream. */
int mpc_costmap_to_obstacles_device(mpc_solver* s, int32_t B, const uint8_t* d_cost, int32_t size_x, int32_t size_y, double resolution,
                                    const double* d_origin /* [B][2] */, const double* d_robot_pose /* [B][3] */, double behind_robot_dist,
                                    int32_t* d_n_obstacles, int32_t* d_n_vertices, double* d_vertices, int32_t* d_dropped);
/* Same with HOST pointers (staged through temporary device buffers; blocking). */
int mpc_costmap_to_obstacles(mpc_solver* s, int32_t B, const uint8_t* cost, int32_t size_x, int32_t size_y, double resolution,
                             const double* origin, const double* robot_pose, double behind_robot_dist,
                             int32_t* n_obstacles, int32_t* n_vertices, double* vertices, int32_t* dropped);

/* Clearance rows that did NOT fit into max_obstacle_rows in the most recent solve, per instance (summed over the grid points k = 1..n-2;
 * association of candidate 0).  0 everywhere = the solve saw exactly the rows the reference would have built.  HOST pointer. */
int mpc_last_rows_dropped(mpc_solver* s, int32_t B, int32_t* rows_dropped);

/* Candidate bookkeeping of the most recent mpc_solve_batch* call (after mpc_synchronize): winner[b] = index of the candidate that supplied
 * instance b's result (-1: none converged, candidate 0's last iterate was returned); iters_total[b] = interior-point iterations spent on
 * instance b over all its candidates (iters[] of the solve call is the winner's count).  HOST pointers, either may be NULL.  With
 * n_candidates <= 1: winner = 0 / -1 from the status, iters_total = iters. */
int mpc_last_candidates(mpc_solver* s, int32_t B, int32_t* winner, int32_t* iters_total);

/* Post-solve feasibility check of the planned pose trajectories against the local costmaps -- Controller::isPoseTrajectoryFeasible
 * (src/controller.cpp:859-917; caller: src/mpc_local_planner_ros.cpp:410-421): feasible[b] = 0 iff a footprint placed at a grid point
 * x[b][0..look_ahead_idx] -- or at a pose interpolated between two neighbours that lie farther apart than inscribed_radius or turn by more
 * than min_resolution_collision_check_angular -- covers a LETHAL cell in the sense footprintCost(...) == -1 of
 * base_local_planner::CostmapModel (PINNED to ROS navigation 1.17 / noetic: -1 lethal, -2 no information, -3 outside the map, the first
 * negative code met along centre, edges in order, Bresenham cells in order decides; fewer than 3 footprint points = centre cell only, where
 * INSCRIBED counts as lethal).  cost[b][my][mx] as in mpc_costmap_to_obstacles; x = the solver's x_out layout [B][cfg.n][3] (per-instance
 * grid sizes of mpc_set_grid_sizes are honoured); footprint_spec = n_spec (<= 32) points (x, y) in the robot frame, HOST pointer in both
 * variants; look_ahead_idx < 0 = the whole horizon.  circumscribed_radius of the reference's signature is unused by footprintCost. */
int mpc_check_feasibility_device(mpc_solver* s, int32_t B, const double* d_x, const uint8_t* d_cost, int32_t size_x, int32_t size_y, double resolution,
                                 const double* d_origin /* [B][2] */, const double* footprint_spec, int32_t n_spec, double inscribed_radius,
                                 double min_resolution_collision_check_angular, int32_t look_ahead_idx, int32_t* d_feasible);
int mpc_check_feasibility(mpc_solver* s, int32_t B, const double* x, const uint8_t* cost, int32_t size_x, int32_t size_y, double resolution,
                          const double* origin, const double* footprint_spec, int32_t n_spec, double inscribed_radius,
                          double min_resolution_collision_check_angular, int32_t look_ahead_idx, int32_t* feasible);

int mpc_synchronize(mpc_solver* s);

/* Duration (ms) of the solve kernel of the most recent mpc_solve_batch* call, measured with
 * HIP events on the solver's own stream (call after mpc_synchronize). */
int mpc_last_kernel_ms(mpc_solver* s, float* ms);

/* Dynamic LDS bytes of one workgroup of the solve kernel for this handle = the working set of ONE planner instance that lives in LDS (mpc_wave_layout.hpp::WaveLayout + the
 * problem record; with mpc_config.stage_data in the global form the factorisation data is not part of it).  A compute unit of the MI355X has 160 KB and its register file
 * holds four of these one-wave workgroups: min(4, 163840 / bytes) are resident per CU -- 4 at BASELINE configs[1] (n = 50, fp64: 40 128 B), 4 at configs[2] (n = 80, 16 polygons,
 * four clearance rows per grid point: 33 200 B in the global form MPC_STAGE_AUTO picks; 83 760 B = 1 per CU in the LDS form), 4 at configs[4]'s shape in fp64 (n = 120: 34 928 B;
 * 95 408 B = 1 in the LDS form) and in fp32 (17 552 B in the global form; 47 792 B = 3 in the LDS form).  MPC_MIXED reports its fp64 phase. */
int mpc_lds_bytes(const mpc_solver* s, int64_t* bytes);

/* What a launch of B instances on this handle occupies, from the runtime's occupancy calculation on the kernel instantiation such a launch selects (registers, LDS, one-wave
 * workgroups): resident workgroups per compute unit (4 = one wave per SIMD; 8 with the two-waves-per-SIMD kernel of mpc_config.two_wave_min_batch) and the dynamic LDS of one of
 * them.  MPC_MIXED reports its fp64 phase.  Measurement aid (bench.py reports it next to the throughput); no counterpart in the reference. */
int mpc_occupancy(mpc_solver* s, int32_t B, int32_t* workgroups_per_cu, int64_t* lds_bytes);

/* Human-readable text of the last HIP/runtime error on this thread ("" if none). */
const char* mpc_last_error(void);

/* Library/ABI version: major*10000 + minor*100 + patch. */
int32_t mpc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MPC_HIP_H_ */
