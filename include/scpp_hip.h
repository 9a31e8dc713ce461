/* scpp_hip.h -- C ABI of the MI355X (gfx950) batched Successive-Convexification engine.
 *
 * Drop-in boundary for the ONE hot path of EmbersArc/SCpp that this library replaces:
 *
 *   reference interface (file:line in the reference tree)                      entry point(s) here
 *   -------------------------------------------------------------------------  ----------------------------
 *   scpp::discretization::multipleShooting(model, td, dd)                      scpp_hip_upload_traj
 *       scpp_core/include/discretization.hpp:16-19                             scpp_hip_set_flow_params
 *       scpp_core/src/discretization.cpp:42-55                                 scpp_hip_discretize
 *       (td: trajectoryData.hpp:8-32, dd: discretizationData.hpp:8-53)         scpp_hip_download_dd
 *   scpp::simulate(model, dt, u0, u1, x)  scpp_core/src/simulation.cpp:25-42   scpp_hip_simulate
 *   SystemModel::updateModelParameters    scpp_core/include/systemModel.hpp:90 scpp_hip_set_flow_params
 *   cvx::ecos::ECOSSolver::solve(false)   scpp_core/src/SCAlgorithm.cpp:78     scpp_hip_socp_solve
 *       on buildSCProblem (SCProblem.cpp:6-138) + RocketQuat::addApplicationConstraints
 *       (scpp_models/src/rocketQuat.cpp:70-144)
 *   SCAlgorithm::initialize/solve/iterate scpp_core/src/SCAlgorithm.cpp:48-189 scpp_hip_sc_setup, scpp_hip_sc_iterate,
 *   (incl. RocketQuat (non|re)dimensionalize*, getInitializedTrajectory,       scpp_hip_sc_solve
 *    getNewModelParameters: rocketQuat.cpp:39-68,156-201,291-332)
 *   SCAlgorithm::getSolution              scpp_core/include/SCAlgorithm.hpp:37 scpp_hip_download
 *   SCvxAlgorithm::initialize/solve/       scpp_core/src/SCvxAlgorithm.cpp:46-227, scpp_hip_scvx_setup, scpp_hip_scvx_solve,
 *   iterate/getNonlinearCost on            SCvxProblem.cpp:6-71                   scpp_hip_scvx_download_state
 *   buildSCvxProblem
 *   SC_sim closed loop (warm start, plant  scpp/src/SC_sim.cpp:28-66            scpp_hip_sc_setup(warm_start=1),
 *   step, stop rule)                                                            scpp_hip_simulate, scpp_hip_sc_set_active
 *   MPCAlgorithm::initialize               scpp_core/src/MPCAlgorithm.cpp:34-69  scpp_hip_mpc_setup, scpp_hip_mpc_get_model
 *     exactLinearDiscretization           scpp_core/src/discretization.cpp:9-40
 *   MPCAlgorithm::setInitialState/        scpp_core/src/MPCAlgorithm.cpp:71-139 scpp_hip_mpc_solve, scpp_hip_mpc_download
 *     setFinalState/solve/getSolution on  MPCProblem.cpp:6-87,
 *     buildMPCProblem + Rocket2d          scpp_models/src/rocket2d.cpp:40-84
 *     constraints
 *   MPC_sim closed loop                    scpp/src/MPC_sim.cpp:16-86            scpp_hip_mpc_sim, scpp_hip_mpc_sim_download
 *
 * Conventions: every function returns 0 on success or a negative SCPP_E_* code; nothing throws or
 * exits.  One context per GPU, one host thread per context.  All host buffers are caller-owned,
 * float64, C-contiguous.  Trajectories are [B][K][nx] / [B][K][nu]; discretization blocks are
 * ROW-major [B][K-1][nx][nx] etc. (the reference stores Eigen column-major blocks per segment).
 * Batch entries are independent problem instances ("B" below).
 */
#ifndef SCPP_HIP_H
#define SCPP_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

#define SCPP_MODEL_ROCKETQUAT 0
#define SCPP_MODEL_ROCKET2D 1
#define SCPP_MODEL_LANDER3DOF 2 /* not a model of the reference: this repository's third model (csrc/model_lander3dof.h), added through the plugin registry */

#define SCPP_MODE_FOH 1 /* interpolate_input   (discretizationData.hpp:56-59) */
#define SCPP_MODE_VT 2  /* free_final_time     (discretizationData.hpp:62-65) */

#define SCPP_OK 0
#define SCPP_E_ARG -1
#define SCPP_E_HIP -2
#define SCPP_E_UNSUPPORTED -3
#define SCPP_E_STATE -4
#define SCPP_STATUS_REJECTION_CAP -5 /* a per-instance STATUS (scpp_hip_download), not a return code; distinct from every SCPP_E_* value */

    typedef struct scpp_hip_ctx scpp_hip_ctx;

    /* MPC.info (MPCAlgorithm.cpp:17-32) + the Rocket2d data the controller reads (rocket2d.cpp:40-84), SI units, radians */
    typedef struct
    {
        int32_t K;                 /* 3..8: the eliminated problem has 2(K-1)+2 <= 16 variables (one FP64 MFMA tile) */
        int32_t nondimensionalize; /* must be 0 (shipped; the reference discretises before it would rescale) */
        int32_t constant_dynamics; /* 1 (shipped) or 0: the reference never changes A, B, z after initialize(), same problem either way */
        int32_t intermediate_cost_active; /* must be 0 (shipped; MPCProblem.cpp:67 is out of bounds otherwise) */
        double time_horizon;
        double state_weights_intermediate[6], state_weights_terminal[6], input_weights[2];
        double x_eq[6], u_eq[2];   /* getOperatingPoint (rocket2d.cpp:40-44) */
        double tan_gamma_gs, theta_max, w_B_max, gimbal_max, T_min, T_max; /* addApplicationConstraints (rocket2d.cpp:62-83);
                                      constrain_initial_final must be off for MPC (model.info:55-58) */
        double x_scale_ref;        /* position magnitude used to scale the error-cost variable, e.g. |r_init| */
        double feastol, abstol, reltol; /* <= 0: 1e-8 (ECOS defaults) */
        int32_t maxit;                  /* <= 0: 50 */
    } scpp_mpc_opts;

    /* RocketQuat::Parameters after loadFromFile (rocketQuat.cpp:234-289): SI units, angles in radians */
    typedef struct
    {
        double g_I[3], J_B[3], r_T_B[3];
        double alpha_m, T_min, T_max, t_max;
        double gimbal_max, theta_max, gamma_gs, w_B_max;
        double x_final[14];
        double final_time;
        int exact_minimum_thrust;
        /* must be 0.  Outside this engine's tile design (round 4; round 6 wrote down the bordered-stage design that would carry it, DESIGN.md section 8,
           and did not build it): with roll
           control (rocketQuat.cpp:135-138) state 13 (omega_z) and input 3 (tau_z) are free, a node then has 14 + 4 = 18 free variables,
           and every stage operation of the solver -- the Hessian block, its inverse Cholesky factor, the coupling products, the
           right-hand sides of the three substitution sweeps, the packed factor record -- is ONE 16 x 16 FP64 tile in the register layout of
           v_mfma_f64_16x16x4_f64 (csrc/tile_engine.h).  Neither extra variable can be presolved (tau_z drives omega_z, omega_z enters the
           quaternion rows of every segment), and eliminating tau_z by its own Schur complement couples the multipliers of segments k-1 and
           k directly, which breaks the block-tridiagonal order the sweeps rely on.  Two tiles per block would be a second solver (32-wide
           eliminations, four products where there is one, a different record layout and lane algebra); a rank-2 border per stage keeps the
           16-wide tile and adds two skinny columns to every stage operation -- a second sweep set all the same.
           The shipped RocketQuat configuration has roll control off (model.info:203), as has every BASELINE configuration.
           scpp_hip_sc_setup / scpp_hip_scvx_setup / scpp_hip_scvx_solve_stream return SCPP_E_UNSUPPORTED for 1. */
        int enable_roll_control;
    } scpp_rocketquat_params;

    /* Rocket2d::Parameters after loadFromFile (rocket2d.cpp:150-198): SI units, angles in radians; constrain_initial_final
       must be on (the SC configuration, model.info:55-56) */
    typedef struct
    {
        double g_I[2], r_T_B[2];
        double m, J_B;
        double T_min, T_max;
        double gimbal_max, theta_max, gamma_gs, w_B_max;
        double x_final[6];
        double final_time;
    } scpp_rocket2d_params;

    /* Lander3dof (ABI revision 7, round 6): point mass with a thrust vector, states [m, r(3), v(3)], inputs T(3) -- a model the reference does
       NOT have; what a user of the reference's SystemModel interface (systemModel.hpp:64-82) adds here for a model of their own: this struct,
       the three *_lander3dof entry points below, a flow map, a constraint table and a plugin struct.  Angles in radians. */
    typedef struct
    {
        int exact_minimum_thrust;
        double g_I[3];
        double alpha_m, T_min, T_max, pointing_max, gamma_gs;
        double x_final[7];
        double final_time;
    } scpp_lander3dof_params;

    /* SC.info (SCAlgorithm.cpp:22-46) */
    typedef struct
    {
        int K;
        int free_final_time, interpolate_input, nondimensionalize, max_iterations;
        double weight_time, weight_trust_region_time, weight_trust_region_trajectory, weight_virtual_control;
        double nu_tol, delta_tol;
    } scpp_sc_opts;

    /* SCvx.info (SCvxAlgorithm.cpp:22-44).  interpolate_input: 1 = first-order hold (shipped), 0 = zero-order hold (round 4:
       buildSCvxProblem without C and with K-1 input trust regions, SCvxProblem.cpp:32-35,58-68; u1 = u0 in getNonlinearCost,
       SCvxAlgorithm.cpp:269; results keep the [K][nu] pitch with row K-1 zero) */
    typedef struct
    {
        int K;
        int interpolate_input, nondimensionalize, max_iterations;
        double alpha, beta, rho_0, rho_1, rho_2;
        double change_threshold, weight_virtual_control, trust_region;
    } scpp_scvx_opts;

    /* interior-point settings (ECOS-style tolerances: the replacement of ECOSSolver::solve, SCAlgorithm.cpp:78 / SCvxAlgorithm.cpp:81, is ECOS's
       algorithm -- NT scaling, Mehrotra predictor-corrector, sigma = (1 - alpha_aff)^3, step-to-boundary 0.99, its exits -- on the structured
       system, with ONE departure since round 6: the primal and the dual variables take step lengths of their own (csrc/ipm_solve.h:
       IPM_SPLIT_STEPS; -11 % interior-point iterations per trajectory) */
    typedef struct
    {
        double feastol, abstol, reltol;
        int maxit;
        int use_mfma; /* kept for ABI stability; ignored: every tile product runs on v_mfma_f64_16x16x4_f64 */
    } scpp_socp_opts;

    /* accumulated device time per kernel family, measured with hipEvents on the context stream */
    typedef struct
    {
        double ms_discretize, ms_socp, ms_other;
        long long n_discretize, n_socp;          /* kernel launches */
        long long inst_discretize, inst_socp;    /* sum over launches of ACTIVE instances processed */
        /* launches of different slot pools (streams) overlap in time, so ms_socp / ms_discretize (sums of spans) can exceed
           the wall clock; *_union is the length of the UNION of the spans on a common time axis since the last reset: the
           time with at least one launch of the family in flight (<= wall clock by construction) */
        double ms_discretize_union, ms_socp_union;
    } scpp_timing;

    int scpp_hip_create(scpp_hip_ctx **ctx, int device_id, int model_id, int K, int batch_max, unsigned flags);
    int scpp_hip_destroy(scpp_hip_ctx *ctx);
    const char *scpp_hip_version(void);
    /* Build-defined constants of THIS library, so that a binding never hard-codes them (ABI revision 5, round 5: the per-instance status
       SCPP_STATUS_REJECTION_CAP moved from -4 to -5 in revision 4 without a way to ask).  Returns SCPP_E_ARG for an unknown `what` or a NULL
       `value`.  Bindings compare SCPP_Q_ABI_REVISION with the header they were written against and refuse a different library. */
#define SCPP_ABI_REVISION 7
#define SCPP_Q_ABI_REVISION 0          /* SCPP_ABI_REVISION of the build */
#define SCPP_Q_STATUS_REJECTION_CAP 1  /* the per-instance status of an SCvx run retired in the reject loop */
#define SCPP_Q_SCVX_SOLVE_CAP 2        /* sub-problem solves per configured SCvx iteration before that happens (csrc/scvx_kernels.h) */
#define SCPP_Q_MAX_K 3                 /* largest K of scpp_hip_create (lane = stage: one wavefront) */
#define SCPP_Q_MPC_MAX_K 4             /* largest horizon of scpp_hip_mpc_setup */
    int scpp_hip_query(int what, long long *value);

    /* ---- multipleShooting / simulate boundary (any model) ---- */
    int scpp_hip_set_flow_params(scpp_hip_ctx *ctx, const double *par /* [B][np] */, int B);
    int scpp_hip_upload_traj(scpp_hip_ctx *ctx, const double *X, const double *U, const double *sigma, int B);
    /* zero-order-hold input: K-1 inputs per trajectory, U [B][K-1][nu] (trajectoryData.hpp:27-32, td.interpolatedInput() == false);
       follow with scpp_hip_discretize(mode without SCPP_MODE_FOH): multipleShootingImplementation<false, VT>
       (discretization.cpp:42-55, discretizationImplementation.hpp:96-101); dd.C reads back as zeros */
    int scpp_hip_upload_traj_zoh(scpp_hip_ctx *ctx, const double *X, const double *U, const double *sigma, int B);
    int scpp_hip_discretize(scpp_hip_ctx *ctx, int mode);
    /* RKF78 steps per shooting segment for every later discretisation of this context (the open-loop call above, the SC / SCvx /
       MPC loops).  5 (DEFAULT since round 4) = the reference's fixed count (scpp_core/include/discretizationImplementation.hpp:141,154),
       literally.  0 = opt-in step-length rule n = clamp(ceil(segment seconds / 0.1714 s), 1, 5): never more steps than the reference,
       never a longer step than the reference's own at the K = 15 it ships; at K = 50 / 12 s that is 2 steps, 1e-13 from the 5-step
       A .. z -- but SCvx accept / reject decisions hinge on the sign of a dJ of ~1e-10, so a handful of decision
       sequences differ from the reference-faithful scheme; that is why it is not the default.  1 .. 4 pin the count. */
    int scpp_hip_set_discretization_steps(scpp_hip_ctx *ctx, int steps);
    int scpp_hip_download_dd(scpp_hip_ctx *ctx, double *A, double *B, double *C, double *S, double *Z);
    int scpp_hip_simulate(scpp_hip_ctx *ctx, const double *dt /* [B] */, const double *u0, const double *u1,
                          double *x /* [B][nx] in/out */, int B);

    /* ---- SCAlgorithm boundary (RocketQuat) ---- */
    int scpp_hip_set_socp_opts(scpp_hip_ctx *ctx, const scpp_socp_opts *opts);
    /* How the interior-point solve (the replacement of ECOSSolver::solve, SCAlgorithm.cpp:78 / SCvxAlgorithm.cpp:81) is scheduled on the device.
       The arithmetic is the same in every schedule -- the results are bitwise identical (tests) -- only the launch structure differs:
         SCPP_IPM_RESIDENT     one kernel per solve, one resident wavefront per instance for all its interior-point iterations (2 waves/SIMD);
         SCPP_IPM_SPLIT        two kernels per interior-point iteration, split by register budget (csrc/ipm_split.h): the factor sweep at two
                               wavefronts per SIMD, everything else at three.  RocketQuat with first-order hold; other configurations run
                               SCPP_IPM_RESIDENT whatever is set.  split_pairs: launch pairs per solve, 0 = the worst case 2 maxit + 1;
         SCPP_IPM_RESIDENT_WS  SCPP_IPM_RESIDENT without the LDS-resident segment fields (diagnostic of that layout choice; CPU emulation build only,
                               SCPP_E_UNSUPPORTED on the device). */
#define SCPP_IPM_RESIDENT 0
#define SCPP_IPM_SPLIT 1
#define SCPP_IPM_RESIDENT_WS 2
#ifndef SCPP_IPM_SCHEDULE_DEFAULT
#define SCPP_IPM_SCHEDULE_DEFAULT SCPP_IPM_RESIDENT
#endif
    int scpp_hip_set_ipm_schedule(scpp_hip_ctx *ctx, int schedule, int split_pairs);
    int scpp_hip_sc_setup(scpp_hip_ctx *ctx, const scpp_rocketquat_params *model, const scpp_sc_opts *opts,
                          const double *x_init /* [B][14] dimensional */, int B, int warm_start);
    /* the same for a Rocket2d context (the reference's default active model, scpp_core/include/activeModel.hpp:10): the
       sub-problem is the same structured solver instantiated for Rocket2d's constraint table (csrc/constraint_table.h) */
    int scpp_hip_sc_setup_rocket2d(scpp_hip_ctx *ctx, const scpp_rocket2d_params *model, const scpp_sc_opts *opts,
                                   const double *x_init /* [B][6] dimensional */, int B, int warm_start);
    /* the same for a Lander3dof context (SCPP_MODEL_LANDER3DOF; not a model of the reference, see scpp_lander3dof_params) */
    int scpp_hip_sc_setup_lander3dof(scpp_hip_ctx *ctx, const scpp_lander3dof_params *model, const scpp_sc_opts *opts,
                                     const double *x_init /* [B][7] dimensional */, int B, int warm_start);
    /* restrict the next sc_iterate / sc_solve to a subset (mask[i] != 0); call after sc_setup.  This is the batched
       form of SC_sim's per-closed-loop stop rule (scpp/src/SC_sim.cpp:57-62): finished loops are not solved again */
    int scpp_hip_sc_set_active(scpp_hip_ctx *ctx, const int32_t *mask /* [B] */, int B);
    int scpp_hip_sc_iterate(scpp_hip_ctx *ctx, int *n_active);  /* one SCAlgorithm::iterate on every active instance */
    int scpp_hip_sc_solve(scpp_hip_ctx *ctx, int *n_converged); /* whole SCAlgorithm::solve loop on the device */
    /* tail of SCAlgorithm::solve (redimensionalizeTrajectory, SCAlgorithm.cpp:182-187) for callers that drive
       sc_iterate themselves, e.g. to record every iterate like getAllSolutions (SCAlgorithm.cpp:217-232) */
    int scpp_hip_sc_finish(scpp_hip_ctx *ctx, int *n_converged);
    int scpp_hip_socp_solve(scpp_hip_ctx *ctx);                 /* sub-problem only, on the current td/dd */
    /* ---- SCvxAlgorithm boundary: fixed final time, hard input trust region, rho-ratio radius update
       (scpp_core/src/SCvxProblem.cpp:6-71, SCvxAlgorithm.cpp:61-227).  Results through scpp_hip_download
       (sc_iters = SCvx iterations); per-instance radius, last nonlinear cost J, number of sub-problem solves and
       [rho, dJ, dL, code] of the last decision (code 0 rejected, 1 accepted, 2 first pass, 3 converged) through
       scpp_hip_scvx_download_state (any pointer may be NULL) ---- */
    int scpp_hip_scvx_setup(scpp_hip_ctx *ctx, const scpp_rocketquat_params *model, const scpp_scvx_opts *opts,
                            const double *x_init /* [B][14] dimensional */, int B, int warm_start);
    /* the same for a Rocket2d context (the reference ships scpp_models/config/Rocket2D/SCvx.info, and SCvxAlgorithm is
       model-generic: SCvxAlgorithm.cpp:46-59, constraints rocket2d.cpp:46-84): the SCvx mode of the solver instantiated for
       Rocket2d's constraint table.  scvx_solve / download / scvx_download_state serve both models. */
    int scpp_hip_scvx_setup_rocket2d(scpp_hip_ctx *ctx, const scpp_rocket2d_params *model, const scpp_scvx_opts *opts,
                                     const double *x_init /* [B][6] dimensional */, int B, int warm_start);
    int scpp_hip_scvx_setup_lander3dof(scpp_hip_ctx *ctx, const scpp_lander3dof_params *model, const scpp_scvx_opts *opts,
                                       const double *x_init /* [B][7] dimensional */, int B, int warm_start);
    int scpp_hip_scvx_solve(scpp_hip_ctx *ctx, int *n_converged);
    int scpp_hip_scvx_download_state(scpp_hip_ctx *ctx, double *trust_region, double *nonlinear_cost, int32_t *solves,
                                     double *last_decision /* [B][4] */);
    /* SCvxAlgorithm::getAllSolutions (scpp_core/include/SCvxAlgorithm.hpp:48, src/SCvxAlgorithm.cpp:245-260; all_td is filled by solve():
       :192 before the first iteration, :201 after every iteration, i.e. after every ACCEPTED candidate -- rejected candidates never appear).
       Opt-in (ABI revision 6, round 6): scpp_hip_scvx_record_iterates(ctx, 1) before scpp_hip_scvx_setup makes that set-up allocate a record of
       max_iterations + 1 trajectories per instance (K x (nx + nu) doubles each: 223 KB per RocketQuat instance at K = 50 / 30 iterations) and
       store the initial trajectory; scpp_hip_scvx_solve then appends the trajectory at the end of every iteration on the device (both engines,
       csrc/scvx_kernels.h: scvxRecordIterate).  The record restarts with every set-up.  Streaming jobs do not record (a slot is re-used).
       scpp_hip_scvx_download_iterates copies the record of instances [first, first + count) redimensionalised like getAllSolutions does:
       X [count][capacity][K][nx], U [count][capacity][K][nu], scalars [count][capacity][4] = {trust radius after the iteration's update,
       sub-problem solves so far (accepted + rejected candidates), nonlinear cost J, decision code: 2 first pass / 1 accepted / 3 converged, 0 for the
       initial trajectory} (entries beyond an instance's count are left untouched), n_iterates [count] = trajectories recorded (1 + iterations
       done); any pointer may be NULL.  The last recorded trajectory is bitwise scpp_hip_download's. */
    int scpp_hip_scvx_record_iterates(scpp_hip_ctx *ctx, int enable);
    int scpp_hip_scvx_download_iterates(scpp_hip_ctx *ctx, int first, int count, int capacity, double *X, double *U, double *scalars,
                                        int32_t *n_iterates);
    /* ---- SCvx streaming engine (continuous batching): SCvxAlgorithm::solve (cold start, SCvxAlgorithm.cpp:166-227) of N
       independent instances pushed through `slots` (<= batch_max; 0: batch_max) resident problem slots.  Instances need very
       different numbers of sub-problem solves (accepted + rejected candidates, SCvxAlgorithm.cpp:132-138); a slot whose
       loop has terminated is refilled from the queue at the next round, so the device stays full until the queue is empty.
       `pools` slot ranges run on their own HIP streams: 0 = heuristic (pools of about 1365 slots = 2/3 of the wavefronts the chip
       holds: 6 pools at 8192 slots, 3 at 4096, 1 below 2731; measured); an explicit value (or the SCPP_STREAM_POOLS environment variable, which overrides it) is honoured
       and only clamped to the slot count and to 8; scpp_hip_stream_info reports the number used.  A failed job (any return code != 0) has joined all pool streams and invalidated its rows:
       scpp_hip_stream_rows / _download then return SCPP_E_STATE.  Every instance computes exactly what
       scpp_hip_scvx_setup + scpp_hip_scvx_solve compute for it (bitwise), whatever slot it lands in.
       Results: one row of K*18 + 10 float64 per instance, in instance order: X [K][14], U [K][4] (dimensional), then
       sigma, ||nu||_1, last nonlinear cost, trust radius, SCvx iterations, sub-problem solves, converged, status,
       interior-point iterations, instance id.  scpp_hip_stream_rows exposes the DEVICE buffer (for the RCCL all-gather of
       the converged trajectories), scpp_hip_stream_download copies rows [first, first+count) to the host. ---- */
    int scpp_hip_scvx_solve_stream(scpp_hip_ctx *ctx, const scpp_rocketquat_params *model, const scpp_scvx_opts *opts,
                                   const double *x_init /* [N][14] dimensional */, int N, int slots, int pools,
                                   int *n_converged);
    /* Rocket2d: rows of K*8 + 10 float64 (X [K][6], U [K][2], then the same ten scalars) */
    int scpp_hip_scvx_solve_stream_rocket2d(scpp_hip_ctx *ctx, const scpp_rocket2d_params *model, const scpp_scvx_opts *opts,
                                            const double *x_init /* [N][6] dimensional */, int N, int slots, int pools,
                                            int *n_converged);
    /* Lander3dof: rows of K*10 + 10 float64 (X [K][7], U [K][3], then the same ten scalars) */
    int scpp_hip_scvx_solve_stream_lander3dof(scpp_hip_ctx *ctx, const scpp_lander3dof_params *model, const scpp_scvx_opts *opts,
                                              const double *x_init /* [N][7] dimensional */, int N, int slots, int pools,
                                              int *n_converged);
    /* Engine of scpp_hip_scvx_solve_stream.  The result rows are bitwise the same with either (every instance's arithmetic is identical):
         SCPP_STREAM_POOLS       rounds of four launches per slot pool (refill, multipleShooting, sub-problem solve, cost + accept / reject);
         SCPP_STREAM_PERSISTENT  (default) ONE launch: a wavefront per slot takes instance after instance through the whole of
                                 SCvxAlgorithm::solve (csrc/scvx_persistent.h).  RocketQuat with first-order hold and `pools` = 0; an
                                 explicit pool count, and every other configuration, runs the pool engine whatever is set. */
#define SCPP_STREAM_POOLS 0
#define SCPP_STREAM_PERSISTENT 1
#ifndef SCPP_STREAM_ENGINE_DEFAULT
#define SCPP_STREAM_ENGINE_DEFAULT SCPP_STREAM_PERSISTENT
#endif
    int scpp_hip_set_stream_engine(scpp_hip_ctx *ctx, int engine);
    /* wavefront time of the last persistent job per step, summed over wavefronts (s_memtime ticks): refill, multipleShooting, sub-problem
       solve, cost + accept / reject; zeros after a pool-engine job */
    int scpp_hip_stream_profile(scpp_hip_ctx *ctx, double *ticks /* [4] */);
    int scpp_hip_stream_rows(scpp_hip_ctx *ctx, void **rows, int *row_doubles, int *n);
    int scpp_hip_stream_download(scpp_hip_ctx *ctx, double *rows /* [count][K*18+10] */, int first, int count);
    int scpp_hip_stream_info(scpp_hip_ctx *ctx, long long *rounds_enqueued, int *pools_used); /* diagnostics of the last job */
    /* results; any pointer may be NULL. status: 0 ok, -1 IPM iteration limit, -2 numerical failure, SCPP_STATUS_REJECTION_CAP (-5):
       SCvx only -- the instance used 64 x max_iterations sub-problem solves without leaving the reject / re-solve loop of
       SCvxAlgorithm::iterate (SCvxAlgorithm.cpp:75-153 has no other exit; seen with the shipped Rocket2D SCvx.info, whose
       trust radius collapses) and was retired; its trajectory is the last accepted iterate */
    int scpp_hip_download(scpp_hip_ctx *ctx, double *X, double *U, double *sigma, int32_t *sc_iters, double *nu_norm,
                          int32_t *converged, int32_t *status, int32_t *ipm_iters, double *sum_delta);
    int scpp_hip_download_socp_info(scpp_hip_ctx *ctx, double *info /* [B][32]: pcost,gap,pres,dres,iters,status,norm1_nu,sum_delta, then 24 profiling slots */);

    /* ---- MPCAlgorithm boundary (Rocket2D, linear MPC with constant dynamics).  mpc_setup = MPCAlgorithm::initialize:
       exact discretisation at the operating point on the host (one 8x8 matrix exponential) and elimination of the states;
       mpc_solve = setInitialState + setFinalState + solve for B independent controllers, one wavefront each.
       status: 0 optimal, 1 reduced accuracy (ECOS "close to optimal"), -1 iteration limit, -2 numerics, -3 the given state
       violates its own glide-slope / tilt / rate constraint (the reference problem is then infeasible).  A failed solve
       leaves that instance's X / U untouched. ---- */
    int scpp_hip_mpc_setup(scpp_hip_ctx *ctx, const scpp_mpc_opts *opts, const double *flow_par /* [6] rocket2d.cpp:143-148 */);
    int scpp_hip_mpc_get_model(scpp_hip_ctx *ctx, double *A /* [6][6] */, double *B /* [6][2] */, double *z /* [6] */);
    int scpp_hip_mpc_solve(scpp_hip_ctx *ctx, const double *x_init /* [B][6] */, const double *x_final /* [B][6] */, int B,
                           int *n_solved /* status >= 0 */);
    int scpp_hip_mpc_download(scpp_hip_ctx *ctx, double *X /* [B][K][6] */, double *U /* [B][K-1][2] */,
                              double *cost /* [B][2]: input_cost, error_cost */, int32_t *status, int32_t *iters);
    /* MPC_sim.cpp:49-86 for B closed loops, entirely on the device: solve at the current state, advance the plant by
       time_step under the PREVIOUS input (scpp::simulate), then apply u = U[0] (a failed solve holds the previous input);
       a loop retires when |x - x_final| < stop_tol or its clock reaches sim_time.  The reference advances by the measured
       solve time floored at 10 ms (MPC_sim.cpp:62-67); here the step is the given constant.  max_steps <= 0: no cap. */
    int scpp_hip_mpc_sim(scpp_hip_ctx *ctx, const double *x_start /* [B][6] */, const double *x_final /* [B][6] */, int B,
                         double time_step, double sim_time, double stop_tol, int max_steps, int *n_reached);
    int scpp_hip_mpc_sim_download(scpp_hip_ctx *ctx, double *x /* [B][6] */, double *u /* [B][2] */, double *t /* [B] */,
                                  int32_t *steps, int32_t *failed_solves, int32_t *ipm_iters, int32_t *reached);

    /* ---- plumbing ---- */
    int scpp_hip_get_timing(scpp_hip_ctx *ctx, scpp_timing *out, int reset);
    /* device pointers of the result buffers (for zero-copy wrapping, e.g. the RCCL all-gather of
       converged trajectories): X [B][K][14], U [B][K][4], sigma [B] */
    int scpp_hip_device_ptrs(scpp_hip_ctx *ctx, void **X, void **U, void **sigma);
    int scpp_hip_synchronize(scpp_hip_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
