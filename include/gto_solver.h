/*
 * gto_solver.h — C ABI of the MI355X-native GTO inner solver (libgto_hip.so).
 *
 * Drop-in boundary for the ONE hot path of IRVLUTD/GraspTrajOpt: the optimisation solve behind
 *   GTOPlanner.plan_goalset()/plan()            reference gto/gto_planner.py:145-245
 *   -> optas.CasADiSolver("ipopt").solve()      reference optas/solver.py:126-159, 388-400
 * The reference has no native FFI (it is 100 % Python on top of CasADi/IPOPT); the entry points
 * below are what a ctypes binding inside the reference's GTOPlanner would call instead of
 * `self.solver.solve()` (see INTEGRATION.md for the stub).  Each function cites the reference
 * code whose role it takes over.
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature; `void* stream` is a hipStream_t passed as an
 *     opaque pointer (NULL = the handle's own stream).
 *   - host arrays are row-major (NumPy default) double / float / int32 unless stated.
 *   - every function returns GTO_OK (0) or a negative error code; gto_last_error() gives text.
 *   - a handle is bound to one HIP device and one stream; it is not thread-safe; host-pointer
 *     calls are synchronous on return (the reference is synchronous single-threaded Python).
 *   - the caller owns all host buffers; the library owns all device memory behind the handle.
 */
#ifndef GTO_SOLVER_H
#define GTO_SOLVER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GTO_OK 0
#define GTO_ERR_INVALID_ARG (-1)
#define GTO_ERR_HIP (-2)
#define GTO_ERR_NO_DEVICE (-3)
#define GTO_ERR_UNSUPPORTED (-4)
#define GTO_ERR_NO_SCENE (-5)
#define GTO_ERR_ALLOC (-6)

/* compile-time capacity of the kernels */
#define GTO_MAX_FRAMES 32 /* kinematic frames after pruning            */
#define GTO_MAX_LINKS 32  /* collision links carrying surface points   */
#define GTO_MAX_OPT 16    /* optimised joints (Panda/Fetch arm: 7, mobile Fetch: 10); IK / base placement: up to 8 */
#define GTO_MAX_DOF 32    /* actuated joints (Panda 9, Fetch 15)        */

/* joint types (optas/models.py:850-866) */
#define GTO_JOINT_FIXED 0
#define GTO_JOINT_REVOLUTE 1 /* revolute and continuous */
#define GTO_JOINT_PRISMATIC 2

/* obstacle-gradient source */
#define GTO_GRAD_CENTRAL_DIFF 0 /* gto/sdf_callback.py:90-114 JacFun numerics (shipped default) */
#define GTO_GRAD_ZERO 1         /* what CasADi AD sees through floor()+gather in the reference   */

/* how gto_solve_batch[_device] runs the Levenberg-Marquardt iterations (same algorithm, same results to round-off) */
#define GTO_MODE_ROUNDS 0        /* default: rounds of two launches (evaluate / step) over at most 384 instances in flight,
                                    every evaluation spread over the whole GPU: highest throughput, lowest latency */
#define GTO_MODE_SINGLE_LAUNCH 1 /* reserved: the single-launch kernel of rounds 1-3 (one workgroup runs an instance's whole
                                    solve; 2.4-2.9x slower than the rounds) was removed; gto_set_mode answers
                                    GTO_ERR_UNSUPPORTED */

/* per-instance solver status (mirrors "return the iterate anyway", optas/solver.py:135) */
#define GTO_STATUS_CONVERGED 0
#define GTO_STATUS_MAX_ITER 1
#define GTO_STATUS_NUMERICAL 2

/*
 * Robot description: the facts the reference pulls from the URDF through optas.RobotModel
 * (optas/models.py:236-321, 826-868) and GTORobotModel (gto/gto_models.py:62-101).
 * Frames are listed parents-before-children; frame i is reached from parent[i] by the fixed
 * origin transform rt2tr(rpy2r(rpy), xyz) followed by the joint motion about/along `axis`.
 */
typedef struct gto_robot_desc {
  int32_t n_frames;
  const int32_t* parent;     /* [n_frames]    -1 for the root                                  */
  const int32_t* joint_type; /* [n_frames]    GTO_JOINT_*                                      */
  const int32_t* q_index;    /* [n_frames]    index into q (0..ndof-1) or -1 for fixed joints  */
  const double* origin_xyz;  /* [n_frames*3]  joint origin (optas/models.py:642-651)           */
  const double* origin_rpy;  /* [n_frames*3]                                                   */
  const double* axis;        /* [n_frames*3]  NOT yet normalised (optas/models.py:653-659)     */
  int32_t ndof;              /* actuated joints, URDF order (optas/models.py:349-354)          */
  int32_t n_opt;             /* optimised joints (optas/models.py:366-386)                     */
  const int32_t* opt_index;  /* [n_opt]       indexes into q                                   */
  const double* lower;       /* [n_opt]       joint limits of the optimised joints             */
  const double* upper;       /* [n_opt]                                                        */
  int32_t n_links;           /* links with surface points (gto/gto_models.py:62-80)            */
  const int32_t* link_frame; /* [n_links]     frame index of each collision link               */
  const double* visual_xyz;  /* [n_links*3]   visual origin (gto/gto_models.py:95-100)         */
  const double* visual_rpy;  /* [n_links*3]                                                    */
  int32_t n_points;
  const double* points;      /* [n_points*3]  surface points in the visual-mesh frame          */
  const int32_t* point_link; /* [n_points]    in [0, n_links)                                  */
  int32_t frame_ee;          /* link_ee      (gto/gto_planner.py:35)                           */
  int32_t frame_gripper;     /* link_gripper (gto/gto_planner.py:36)                           */
  int32_t n_gripper_points;
  const double* gripper_points; /* [n_gripper_points*3] in the gripper LINK frame (:37)        */
} gto_robot_desc;

/* Planner constants the reference hard-codes (gto/gto_planner.py:25-30,131,135,141). */
typedef struct gto_solver_opts {
  int32_t T;               /* waypoints, reference 50                                        */
  double Tmax;             /* reference 10.0 -> dt = Tmax/(T-1)                              */
  int32_t standoff_offset; /* reference -10 -> standoff waypoint T+offset                    */
  double w_obstacle;       /* reference 10                                                   */
  double w_vel;            /* reference 0.01                                                 */
  int32_t max_iter;        /* Gauss-Newton/LM iterations (reference IPOPT cap: 100)          */
  double tol_step;         /* stop when max|dq| of an accepted step is below this [rad]      */
  double tol_rel_f;        /* stop when an accepted step lowers f by less than this * (1+f)  */
  double lambda0;          /* initial LM damping                                             */
  int32_t grad_mode;       /* GTO_GRAD_*                                                     */
} gto_solver_opts;

typedef struct gto_handle gto_handle;

/* Fill opts with the reference's constants and this solver's defaults. */
void gto_default_opts(gto_solver_opts* opts);

/* Library/ABI version (major*1000 + minor): GTO_ABI_VERSION of the header the library was built from.  A binding checks it
 * when it loads the library and refuses another number (grasptrajopt_amd/_capi.py load_library): every change of a
 * signature or of a struct in this header bumps the minor. */
#define GTO_ABI_VERSION 1007
int32_t gto_version(void);

/*
 * Create a solver bound to HIP device `device` (a negative value keeps the current device).
 * Copies everything it needs out of `desc`.  Fails loudly (GTO_ERR_NO_DEVICE) without a GPU:
 * there is no CPU fallback behind this ABI.
 * Replaces: GTOPlanner.__init__ + setup_optimization graph construction
 *           (gto/gto_planner.py:22-142) and CasADiSolver.setup (optas/solver.py:335-386).
 */
int gto_create(const gto_robot_desc* desc, const gto_solver_opts* opts, int device, gto_handle** out);
void gto_destroy(gto_handle* h);
const char* gto_last_error(const gto_handle* h); /* h may be NULL: error of the last failed create */

/* Change solver options that do not alter problem dimensions (max_iter, tolerances, weights,
 * lambda0, grad_mode). T must stay the same. */
int gto_set_opts(gto_handle* h, const gto_solver_opts* opts);

/*
 * Upload (or replace) the voxelised cost fields of scene `scene_id` (0 <= scene_id < 65536).
 * c_all / c_obs: float32 [shape0*shape1*shape2], C order, x slowest (gto/gto_models.py:155-171,
 * 184-186); c_obs may be NULL (then c_obs = c_all).  origin/res: gto/gto_models.py:159,46.
 * Replaces: reset_parameters({"sdf_cost_all":..,"sdf_cost_obstacle":..}) (gto/gto_planner.py:227-236).
 * One-time per scene; not part of the timed solve.
 */
int gto_set_scene(gto_handle* h, int32_t scene_id, const float* c_all, const float* c_obs,
                  const int32_t shape[3], const double origin[3], double res);
/* The same upload without the solver's acceleration structures (voxel records, distance fields: 3.6x the bytes of the
 * two fields and 48 relaxation sweeps): such a scene serves gto_plan_cost and gto_eval_points only (seed scoring of a
 * field that is never solved on: GTORobotModel.compute_plan_cost, gto/gto_models.py:204-215); every solve or
 * objective entry point rejects it with GTO_ERR_NO_SCENE. */
int gto_set_scene_values(gto_handle* h, int32_t scene_id, const float* c_all, const float* c_obs,
                         const int32_t shape[3], const double origin[3], double res);
int gto_drop_scene(gto_handle* h, int32_t scene_id);

/*
 * Solve B independent (scene, goal-set) instances.  Host pointers; synchronous.
 *   scene_id  [B]
 *   qc        [B, ndof]        current configuration (gto/gto_planner.py:47-49,59-62)
 *   goals     [B, n_max, 16]   goal poses of link_ee in the robot-base frame, each 4x4 row-major
 *                              (tf_goal column = RT.flatten(), gto/gto_planner.py:188-191)
 *   n_goals   [B]              1 <= n_goals[b] <= n_max (goal-set min, gto/gto_planner.py:105)
 *   standoff  [B, 16] or NULL  standoff pose S (optas/spatialmath.py:160-183); NULL = use_standoff False
 *   base_pos  [B, 3]           base_position parameter (gto/gto_planner.py:56,116)
 *   Q0        [B, ndof, T]     seed trajectory incl. parameter-joint rows (gto/gto_planner.py:193-224,234)
 * Outputs (any may be NULL):
 *   Q_out [B, ndof, T], dQ_out [B, ndof, T-1], cost_out [B] (objective f, Appendix A of SURVEY.md),
 *   iters_out [B], status_out [B] (GTO_STATUS_*).
 * B may be as large as the caller has work: the solver keeps at most 384 instances in flight (environment
 * GTO_SLOTS when the handle is created) and hands the slot of an instance that finishes to the next one that
 * has not started, so large calls keep the GPU full to the end; every instance gets bit for bit the result it
 * gets in a call of its own.  Device workspace: ~75 KB per instance of the call.
 * Replaces: reset_initial_seed + reset_parameters + solve + solution unpacking
 *           (gto/gto_planner.py:222-245, optas/solver.py:103-159).
 */
int gto_solve_batch(gto_handle* h, int32_t B, int32_t n_max, const int32_t* scene_id,
                    const double* qc, const double* goals, const int32_t* n_goals,
                    const double* standoff, const double* base_pos, const double* Q0,
                    double* Q_out, double* dQ_out, double* cost_out, int32_t* iters_out,
                    int32_t* status_out);

/*
 * Same contract with every array already resident in device memory (HBM) and the work enqueued
 * on `stream` (asynchronous; outputs are valid after the stream is synchronised).
 */
int gto_solve_batch_device(gto_handle* h, int32_t B, int32_t n_max, const int32_t* scene_id,
                           const double* qc, const double* goals, const int32_t* n_goals,
                           const double* standoff, const double* base_pos, const double* Q0,
                           double* Q_out, double* dQ_out, double* cost_out, int32_t* iters_out,
                           int32_t* status_out, void* stream);

/*
 * Let `dst` use scene `src_id` of `src` under the id `dst_id` without a second copy in HBM (fields, voxel
 * records, distance fields: 148 MB for a 128^3 scene).  For handles that work on the same scene side by side
 * (grasptrajopt_amd.parallel.BatchPipeline).  `src` keeps ownership: it must outlive `dst`'s use of the
 * scene, and replacing or dropping the scene in `src` invalidates the borrowed entry.
 */
int gto_share_scene(gto_handle* dst, int32_t dst_id, gto_handle* src, int32_t src_id);
/* The same with a choice of halves: `dst`'s sdf_cost_all is `src`'s field number all_from, its sdf_cost_obstacle `src`'s
 * field number obs_from (0: the source's sdf_cost_all, 1: its sdf_cost_obstacle; gto_share_scene = (0, 1)).  For a
 * caller that holds ONE field of a resident scene and hands it to an entry point that reads the obstacle half
 * (IKSolver.solve_ik and GTORobotModel.compute_plan_cost take `sdf_cost_obstacle`, gto/ik_solver.py:78,
 * gto/gto_models.py:204): whichever half the field is, it is the half the entry point reads. */
int gto_share_scene_halves(gto_handle* dst, int32_t dst_id, gto_handle* src, int32_t src_id, int32_t all_from,
                           int32_t obs_from);

/*
 * Inverse kinematics for B goal poses of link_ee: the pre-step that produces the
 * q_solutions of plan_goalset.  T = 1 problem of gto/ik_solver.py:30-110:
 *   min_q sum_k ||T_g(q) p_k - RT G p_k||^2 + w_obstacle * sum_pts c_obs[off(x(q))],  lo <= q <= hi,
 * seeded at q0 (parameter joints of q0 are kept), solved by the same projected Levenberg-Marquardt as the
 * trajectory problem; the whole iteration runs on the GPU, one workgroup per goal.
 *   scene_id  [B] or NULL.  NULL = no collision term (IKSolver(collision_avoidance=False), :63)
 *   q0        [B][ndof]     seed configurations (:79)
 *   goals     [B][16]       RT of link_ee, row-major 4x4 (:83)
 *   base_pos  [B][3] or NULL (zeros)
 *   max_iter  iteration cap (reference IPOPT cap: 50, :76)
 * Outputs (host, may be NULL except q_out): q_out [B][ndof], cost_out [B] objective value, iters_out,
 * status_out as in gto_solve_batch.  The reference's err_pos / err_rot / plan cost (:88-97) follow from
 * gto_eval_fk and gto_plan_cost.
 * Replaces: IKSolver.setup_optimization + solve_ik (gto/ik_solver.py:30-110).
 */
int gto_solve_ik_batch(gto_handle* h, int32_t B, const int32_t* scene_id, const double* q0, const double* goals,
                       const double* base_pos, int32_t max_iter, double* q_out, double* cost_out, int32_t* iters_out,
                       int32_t* status_out);

/*
 * Base placement of a mobile manipulator for B goal sets: where to park the base so that
 * every goal of the set is reachable.  Problem of gto/base_planner.py:35-94 with T = goal_size:
 *   min  effort_weight * |(x,y,theta)|^2 + sum_i sum_k | T_g(q_i) p_k - B(x,y,theta) RT_i G p_k |^2 ,
 *   lo <= q_i <= hi (:92),  -pi <= theta <= pi (:55),  B = rt2tr(rotz(theta), [x,y,0]) (:49-51),
 * unknowns: the base pose and ONE arm configuration per goal (:58-60); seeded at the zero pose and at qc
 * for every goal (:104-105).  Same projected Levenberg-Marquardt as the trajectory problem, the whole
 * iteration on the GPU, one workgroup per goal set; the arrow-shaped normal equations are eliminated
 * goal block by goal block onto the 3x3 base block.
 *   n_max     row stride of goals / q_out, 1 <= n_goals[b] <= n_max <= 32
 *   n_goals   [B]
 *   qc        [B][ndof]          current configuration (:96)
 *   goals     [B][n_max][16]     RT_i of link_ee in the CURRENT base frame, row-major 4x4 (:98-101)
 *   max_iter  iteration cap (reference IPOPT cap: 100, :95)
 * Outputs (host): y_out [B][3] = (x, y, theta), the old base in the new base frame (:52);
 * q_out [B][n_max][ndof] arm configuration per goal (rows >= n_goals[b]: qc); cost_out, iters_out,
 * status_out as in gto_solve_batch (may be NULL).  err_pos / err_rot (:127-143) follow from gto_eval_fk,
 * the occupancy statistic (:146-158) from gto_eval_points' transformed points.
 * Replaces: BasePlanner.setup_optimization + the solve inside plan_goalset (gto/base_planner.py:35-123).
 */
int gto_solve_base_batch(gto_handle* h, int32_t B, int32_t n_max, const int32_t* n_goals, const double* qc,
                         const double* goals, double effort_weight, int32_t max_iter, double* y_out, double* q_out,
                         double* cost_out, int32_t* iters_out, int32_t* status_out);

/* Select the solver mode for the following solves: GTO_MODE_ROUNDS is the only one left. */
int gto_set_mode(gto_handle* h, int32_t mode);

/*
 * Lanes of a solve call.  gto_solve_batch / gto_solve_batch_device deal the instances of a call to up to `max_lanes`
 * lanes (contiguous ranges of at least `min_per_lane` instances), each with a HIP stream and lists of its own over the
 * one workspace of the call and a host thread of the call feeding it (lane 0: the calling thread): one lane's evaluation
 * launch overlaps another's step launch while the GPU is full.  A lane with at most `adopt_below` instances left
 * (0: never) hands them to lane 0, which runs the stragglers of the whole call as one chain of launches.
 * DEFAULTS of a new handle: max_lanes 1, min_per_lane 256, adopt_below 0 (environment: GTO_LANES, GTO_LANE_MIN,
 * GTO_ADOPT) -- one lane, no thread but the caller's, everything on the handle's stream.
 * With ONE lane the call runs on the handle's stream (the `stream` argument of gto_solve_batch_device).  With SEVERAL,
 * every lane -- lane 0 too -- runs on a stream of its own (the handle's, created with the greatest priority, or the
 * caller's: gto_set_lane_streams); the caller's stream only BRACKETS the call: it carries the seeds and the static-link
 * pass in front, the lanes wait for that, and the results are ordered behind all lanes on it.  Results do not depend on
 * any of the three numbers (tests/test_gpu_parity.py).  The reference has no counterpart: gto/gto_planner.py:185-245
 * solves one instance per call.
 */
int gto_set_lanes(gto_handle* h, int32_t max_lanes, int32_t min_per_lane, int32_t adopt_below);
/*
 * The lanes' streams.  By default the handle creates a non-blocking stream per lane.  The runtime deals streams to a few
 * hardware queues (GPU_MAX_HW_QUEUES, default 4) and the queues to the dispatcher's pipes; which stream lands where
 * depends on every stream the process has created, and two lanes behind one pipe run 1.5x slower.  A caller that manages
 * its streams (torch.cuda.Stream, one per lane) hands them over here: lane l of every following call runs on streams[l]
 * (n = 0: the handle's own again).  The streams stay the caller's; they must outlive the calls.
 */
int gto_set_lane_streams(gto_handle* h, int32_t n, void* const* streams);

/*
 * Bind the handle to the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream): every
 * launch and copy of every entry point then goes to that stream and the handle creates none of its
 * own.  NULL gives the handle a private non-blocking stream again (the state after gto_create).
 * The runtime maps streams onto a small number of hardware queues (GPU_MAX_HW_QUEUES, default 4);
 * handles that are meant to overlap on one GPU (grasptrajopt_amd.parallel.BatchPipeline) should each
 * own exactly one stream so that no two of them share a queue.
 */
int gto_set_stream(gto_handle* h, void* stream);

/* Time spent inside the dominant kernel (gto_obstacle_gram) during the most recent solve,
 * measured with HIP events on the launch stream: total milliseconds and launch count. */
int gto_last_kernel_time(gto_handle* h, double* total_ms, int32_t* launches);
/* Work the dominant kernel did during the most recent solve (profiling enabled): surface points it looked up in
 * a field (one 32-B voxel record or one float each; the broad phase skips the rest, which would read exact
 * zeros) and chunk bounding spheres it tested.  bench.py prices the kernel's roofline on the points gathered. */
int gto_last_kernel_work(gto_handle* h, uint64_t* points_gathered, uint64_t* chunk_tests);
/* The same per kernel VARIANT of the solve loop (profiling enabled), so that time, launches and work of one variant are
 * never mixed with another's: launches, their summed HIP-event time, the workgroups they were laid out for and (obstacle
 * variants) the surface points they gathered during the most recent gto_solve_batch[_device] call. */
#define GTO_PROF_OBSTACLE 0      /* k_obstacle_gram<NP,1>: the launches that fill the GPU */
#define GTO_PROF_OBSTACLE_FEW 1  /* k_obstacle_gram<8,8>: launches with few instances in flight */
#define GTO_PROF_STEP 2          /* k_lm_step<4,1> (k_lm_step_wide for nine to sixteen optimised joints) */
#define GTO_PROF_STEP_FEW 3      /* k_lm_step<8,4>: few instances in flight, candidate trial points */
#define GTO_PROF_VARIANTS 4
int gto_last_kernel_profile(gto_handle* h, int32_t variant, double* total_ms, int32_t* launches, uint64_t* workgroups,
                            uint64_t* points_gathered);
/* Enable/disable per-launch event timing of the solve loop's kernels (off by default). */
int gto_set_profiling(gto_handle* h, int32_t enabled);

/* ---- evaluation entry points (host pointers, synchronous): the pieces of the objective the
 * reference evaluates through CasADi Functions; used by the parity tests and by the seed /
 * collision-filter steps around the solve. ---------------------------------------------------- */

/* Global transform of every frame, [nq, n_frames, 16] row-major 4x4
 * (optas/models.py:826-868 get_global_link_transform). */
int gto_eval_fk(gto_handle* h, int32_t nq, const double* q /*[nq,ndof]*/, double* frames_out);

/* World surface points x = visual_tf(q) p + base (gto/gto_planner.py:114-116), their flat voxel
 * offsets (gto/gto_models.py:174-187), nearest-voxel cost values of both fields and the
 * central-difference gradient of the field selected by `use_obs` (gto/sdf_callback.py:43-49,
 * 90-114).  Any output may be NULL.  Points are reported in the caller's original order. */
int gto_eval_points(gto_handle* h, int32_t scene_id, int32_t nq, const double* q /*[nq,ndof]*/,
                    const double* base_pos /*[nq,3]*/, int32_t use_obs,
                    double* xyz_out /*[nq,P,3]*/, int32_t* offset_out /*[nq,P]*/,
                    double* value_out /*[nq,P]*/, double* grad_out /*[nq,P,3]*/);

/* The Hessian of the field selected by `use_obs` at the same surface points, [nq, P, 9] row-major 3x3: mixed central
 * differences (f(i+e_a+e_b) - f(i+e_a-e_b) - f(i-e_a+e_b) + f(i-e_a-e_b)) / (4 res^2) of the nearest-voxel values with every
 * sample's indices clipped on their own (gto/sdf_callback.py:159-183, HesFun.eval / get_value).  The solve never reads it
 * (Gauss-Newton); it completes the SDFCallback / JacFun / HesFun triple of the reference behind this ABI. */
int gto_eval_points_hessian(gto_handle* h, int32_t scene_id, int32_t nq, const double* q /*[nq,ndof]*/,
                            const double* base_pos /*[nq,3]*/, int32_t use_obs, double* hess_out /*[nq,P,9]*/);

/* Objective terms of SURVEY.md Appendix A at given trajectories Q [B, ndof, T]:
 * f_goal (min over the goal set), f_obs (incl. w_obstacle), f_vel (incl. w_vel), arg-min goal.
 * (gto/gto_planner.py:84-105, 108-131, 133-135.) */
int gto_eval_objective(gto_handle* h, int32_t B, int32_t n_max, const int32_t* scene_id,
                       const double* goals, const int32_t* n_goals, const double* standoff,
                       const double* base_pos, const double* Q, double* f_goal, double* f_obs,
                       double* f_vel, int32_t* goal_argmin);

/* Gauss-Newton normal equations of the obstacle term at Q [B, ndof, T], per waypoint:
 * JtJ [B, T, n_opt, n_opt], Jtr [B, T, n_opt], sumsq [B, T] (unweighted: residual = cost value). */
int gto_eval_obstacle_normal_eq(gto_handle* h, int32_t B, const int32_t* scene_id,
                                const double* base_pos, const double* Q, double* JtJ, double* Jtr,
                                double* sumsq);

/* Base-placement objective of gto/base_planner.py:57-87 at a given point: y [B][3] = (x, y, theta),
 * q [B][n_max][ndof] one arm configuration per goal (parameter joints of row 0 are used for every goal, as the
 * reference's q/p parameter), goals [B][n_max][16]:  effort_weight |y|^2 + sum_i sum_k |A(q_i) p_k - B(y) RT_i G p_k|^2.
 * Evaluated by the solve kernel itself (a run of gto_solve_base_batch's kernel capped at 0 iterations, started
 * at (y, q)): theta is clipped to [-pi, pi] and q to the joint limits first. */
int gto_eval_base_objective(gto_handle* h, int32_t B, int32_t n_max, const int32_t* n_goals, const double* y,
                            const double* q, const double* goals, double effort_weight, double* cost_out);

/* Seed scoring: compute_plan_cost (gto/gto_models.py:204-215): plain sum of c_obs over all
 * waypoints and surface points, and ||q_0 - q_{T-1}||.  plans [n, ndof, T]. */
int gto_plan_cost(gto_handle* h, int32_t scene_id, int32_t n, const double* plans,
                  const double* base_pos /*[3]*/, double* cost_out /*[n]*/, double* dist_out /*[n]*/);

/*
 * Cost field from a depth image: what the reference's DepthPointCloud does with a KD-tree on
 * the CPU (mesh_to_sdf/depth_point_cloud.py:9-141) to produce sdf_cost_all / sdf_cost_obstacle.  Stand-alone
 * (no handle): uses HIP device `device`.
 *   depth [height*width] float32, row-major; K, Kinv 3x3 row-major (the reference inverts with np.linalg.inv;
 *   pass the same inverse for bit parity); cam_pose, cam_inv 4x4 row-major (camera in the world / its inverse);
 *   target_mask [height*width] or NULL (pixels != 0 are dropped, :41-44); threshold: depth cut-off (:13)
 *   query [nq][3] world points (the voxel centres `workspace_points`, gto/gto_models.py:155-171)
 * Outputs (host; each may be NULL): sdf_out [nq] signed distance to the nearest back-projected point (:56-61),
 * inside_out [nq] = !is_outside (:126-141), cost_out [nq] = get_sdf_cost(query, epsilon, w_inside) (:64-91),
 * points_out [height*width][3] + valid_out [height*width]: the back-projected world points in pixel order.
 * Bit-identical to the reference on tests/golden/depth_cost.npz.
 */
int gto_depth_sdf_cost(int device, const float* depth, int32_t height, int32_t width, const double* K, const double* Kinv,
                       const double* cam_pose, const double* cam_inv, const uint8_t* target_mask, double threshold,
                       const double* query, int64_t nq, float epsilon, float w_inside, float* sdf_out,
                       uint8_t* inside_out, float* cost_out, double* points_out, uint8_t* valid_out);

/*
 * The per-object perception steps of examples/pybullet_gto_planning.py:176-190 in ONE call, leaving a scene resident:
 * depth image -> cloud of all pixels and cloud without the target's pixels (mesh_to_sdf/depth_point_cloud.py:9-53) ->
 * grid = bounding box of the first cloud + margin at grid_res (gto/gto_models.py:155-171; numpy.arange's values) ->
 * sdf_cost_all and sdf_cost_obstacle at the voxel centres (depth_point_cloud.py:64-91; bit-identical to two
 * gto_depth_sdf_cost calls) -> scene `scene_id` of the handle with its voxel records and distance fields.  The image is
 * uploaded once, back-projection and query ordering are shared by the two fields, and nothing but the geometry returns
 * to the host: shape_out [3], origin_out [3], bounds_out [6] = (min x, y, z, max x, y, z) of the first cloud.
 * depth_obstacle [height*width] or NULL: the image of the SECOND cloud (the driver's `depth_obstacle`, :187-189: a copy of
 * the depth image with the target's pixels pushed to the threshold); its points are back-projected from it, without the
 * pixels of target_mask, and its visibility test (depth_point_cloud.py:126-141) reads it.  NULL: the second cloud is
 * `depth` without the masked pixels.  target_mask == NULL and depth_obstacle == NULL: one cloud, sdf_cost_obstacle =
 * sdf_cost_all.
 */
int gto_scene_from_depth(gto_handle* h, int32_t scene_id, const float* depth, int32_t H, int32_t W, const double* K,
                         const double* Kinv, const double* cam_pose, const double* cam_inv, const uint8_t* target_mask,
                         const float* depth_obstacle, double threshold, double grid_res, double margin, float epsilon,
                         float w_inside, int32_t* shape_out, double* origin_out, double* bounds_out);
/* The two float32 cost fields of a resident scene, device to host (either pointer may be NULL). */
int gto_get_scene_fields(gto_handle* h, int32_t scene_id, float* c_all_out, float* c_obs_out);

#ifdef __cplusplus
}
#endif
#endif /* GTO_SOLVER_H */
