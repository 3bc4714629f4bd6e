/* ============================================================================
 * rootba_hip.h — C ABI of the MI355X-native square-root BA inner solver.
 *
 * Drop-in boundary for the reference's QR solver path (NikolausDemmel/rootba).
 * The reference has no C ABI: its seams are the C++ interfaces
 *   Linearizor<Scalar>               src/rootba/solver/linearizor.hpp:48-83
 *   LinearizationQR<Scalar, 9>       src/rootba/qr/linearization_qr.hpp:54-841
 *   LandmarkBlock<Scalar>            src/rootba/qr/landmark_block.hpp:49-150
 * Each entry point below cites the reference call it replaces (file:line under
 * /root/reference). A `LinearizorHIP<Scalar> : Linearizor<Scalar>` that binds
 * these calls is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C: pointers, sizes, POD structs; no C++/torch types.
 *  - one opaque handle per solver instance; `dtype` chosen at creation
 *    (RBA_F32 / RBA_F64  <->  SolverOptions::use_double,
 *    src/rootba/bal/solver_options.hpp:257-259). All `void*` vectors are arrays
 *    of that scalar type in HOST memory, borrowed for the duration of the call.
 *  - every function returns an int status: RBA_OK (0), RBA_NUMERICAL_FAILURE (1:
 *    the reference returns an empty vector / NaN / LOG(FATAL)s here),
 *    negative = API or HIP/RCCL error (see rba_last_error). Nothing aborts, no
 *    exception crosses the boundary.
 *  - a handle is not thread-safe (single caller, like the reference's LM loop);
 *    the library owns all device memory.
 *  - landmarks are given in CSR form: observations of landmark l are
 *    [lm_obs_offsets[l], lm_obs_offsets[l+1]), camera indices ascending inside a
 *    landmark (std::map order, src/rootba/bal/bal_problem.hpp:131), every
 *    landmark has >= 2 observations (landmark_block_base.ipp:70-73).
 *  - environment: the library reads a handful of DEBUG / TEST variables once per rba_create (none is needed in
 *    production, none changes results beyond rounding): RBA_VERBOSE, RBA_EXPLICIT_AFTER (overrides
 *    rba_options.explicit_after), RBA_EX_PAIR_BUDGET_GB, RBA_FORCE_EXPLICIT_FALLBACK, RBA_HX_LDS, RBA_HX_WIN,
 *    RBA_HX_TIMING_STRIDE, RBA_SORT_BY_CAMERA, RBA_VERIFY_ASSEMBLED (diagnostic, default off since round 4),
 *    RBA_VERIFY_TOLERANCE, RBA_PCG_SPLIT, RBA_HALF_LOWER_MAX, RBA_S1_FUSED, RBA_HX_WIDE_INSIDE, RBA_PCG_PERSISTENT (0: the PCG on
 *    the assembled matrix in two launches per iteration instead of the persistent kernel), RBA_SPMV_STREAM (0: the product
 *    with an assembled matrix that does not fit the register files always one wavefront per work item; 1, default:
 *    persistent streaming wavefronts for matrices of >= 4 items per resident wavefront; 2: always), RBA_SPMV_STREAM_WAVES,
 *    RBA_SPMV_STREAM_BUFFERS (chunks in flight per streaming wavefront: 1, default - seven wavefronts per compute unit; 2 - four),
 *    RBA_SERIES_F32 (0: the terms of the power-series preconditioner through the double matrix instead of its float copy),
 *    RBA_PCGP_TRACE (file for that
 *    kernel's phase stamps), RBA_STAGE_TIMERS (0: rba_iter_timings stays zero; 1, default: device clock stamps at the stage
 *    boundaries inside rba_lm_step, HIP events around single calls; 2: HIP events everywhere), RBA_DETERMINISTIC (1: the matrix-free products are summed camera-major in a fixed order
 *    instead of with floating-point atomics and the measured break-even of the operator switch is frozen - float32
 *    runs repeat bit by bit, a product costs about twice as much) (rootba_amd/csrc/solver.hip: Solver::DebugEnv).
 *    The ~25 kernel-selection
 *    switches of rounds 1-2 are gone with the kernels they selected.
 *  - camera state: 10 scalars (qx,qy,qz,qw,tx,ty,tz,f,k1,k2) = Camera::params()
 *    (bal_problem.hpp:84-95); pose/intrinsics increments: 9 per camera.
 * ==========================================================================*/
#ifndef ROOTBA_HIP_H_
#define ROOTBA_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBA_OK 0
#define RBA_NUMERICAL_FAILURE 1
#define RBA_ERR_INVALID_ARGUMENT (-1)
#define RBA_ERR_HIP (-2)
#define RBA_ERR_UNSUPPORTED (-3)
#define RBA_ERR_COMM (-4)

#define RBA_F32 0
#define RBA_F64 1
/* Mixed precision (BASELINE config 5; no reference counterpart - the reference is templated on one
 * Scalar): the optimisation state (cameras, landmarks), the observations and every cost evaluation
 * are double; linearisation, landmark QR, the reduced camera system, PCG and back-substitution run
 * in float on the rounded state, with the accumulations the float path already keeps in double
 * (cost sums, PCG scalars, Jp_diag2, b, H*x partial sums, l_diff). The float increments are applied
 * to the double state, so rounding does not accumulate from one LM iteration to the next and the
 * step-quality ratio is computed from double costs.
 * At the boundary: `obs_xy` of rba_create and the arrays of rba_set_state / rba_get_state are
 * DOUBLE; every camera-sized vector (increments, b, x, y, blocks, Jp_diag2, scalings) is FLOAT. */
#define RBA_MIXED 2

/* SolverOptions fields consumed by the hot path
 * (src/rootba/bal/solver_options.hpp:111-281; read at
 *  src/rootba/solver/linearizor_qr.cpp:58-68, linearizor_base.cpp:87-93,
 *  bal_bundle_adjustment.cpp:264-272). Field names follow SolverOptions. */
typedef struct rba_options {
  int use_householder;            /* use_householder_marginalization; 0 (Givens, ipp:700-715) is accepted and
                                     executed with Householder reflectors: every consumer of the block is
                                     invariant to the choice of the orthogonal factor */
  int use_valid_projections_only; /* = use_projection_validity_check()            */
  int robust_norm;                /* 0 NONE, 1 HUBER (bal_residual_options.hpp)   */
  double huber_parameter;
  double jacobi_scaling_eps;      /* 0 -> Sophus epsilonSqrt<Scalar>              */
  int preconditioner_type;        /* 0 JACOBI, 1 SCHUR_JACOBI, 2 POWER_SCHUR_COMPLEMENT */
  int reduction_alg;              /* accepted, ignored (always device scatter-add)*/
  int power_order;                /* order m of the power-series preconditioner   */
  int min_cg_it;                  /* min_linear_solver_iterations                 */
  int max_cg_it;                  /* max_linear_solver_iterations                 */
  double eta;
  int num_threads;                /* accepted, ignored                            */
  int max_num_iterations;
  double min_relative_decrease;
  double initial_trust_region_radius;
  double min_trust_region_radius;
  double max_trust_region_radius;
  double function_tolerance;
  double initial_vee;
  double vee_factor;
  int optimized_cost;             /* 0 ERROR, 1 ERROR_VALID, 2 ERROR_VALID_AVG    */
  int staged_execution;           /* 1 (default): stage timers only. 0: the reference's unstaged sub-stage timers
                                     are measured as well (rba_get_substage_timings); same kernels either way */
  int implicit_q;                 /* accepted, ignored (kept for ABI stability): matrix-free products H*x are always
                                     evaluated from the factors (Jp rows, Householder vectors, damping rotations);
                                     no dense landmark block is ever written, so there is no limit on the track
                                     length. The dense-block configuration of rounds 1-2 (= 0) was removed in round 3 */
  int solver_type;                /* SolverOptions::SolverType: 0 SQUARE_ROOT (default, LinearizorQR),
                                     1 SCHUR_COMPLEMENT (LinearizorSC, linearizor_sc.cpp:70-211:
                                     explicit block-sparse reduced camera matrix + SpMV; preconditioners
                                     SCHUR_JACOBI and POWER_SCHUR_COMPLEMENT like the reference) */
  int explicit_after;             /* square-root solver: after this many matrix-free products a PCG solve
                                     assembles S = sum_l A_l^T A_l explicitly (block-CSR, DOUBLE values in
                                     half storage since round 4 - for a float solver derived in double from
                                     the float factors, rootba_amd/csrc/kernels_a64.hpp: a float matrix costs
                                     the PCG the accuracy the square-root form exists for) and continues with
                                     S x; 0 = never; -1 (default) = 6 for the first long solve, then the
                                     measured break-even (assembly time / (matrix-free product time - time of an iteration on S), >= 2),
                                     left early - at iteration 5 - by solves whose stopping quantity rises
                                     from iteration 3 to 4                         */
} rba_options;

/* ResidualInfo (src/rootba/bal/residual_info.hpp:57-96), sums in double */
typedef struct rba_residual_info {
  int all_num_obs;
  double all_error;
  double all_residual_sum;
  int valid_num_obs;
  double valid_error;
  double valid_residual_sum;
  int is_numerically_valid;
} rba_residual_info;

/* ConjugateGradientsSolver::Summary (src/rootba/cg/conjugate_gradient.hpp:97-108) */
typedef struct rba_cg_summary {
  int termination_type; /* 0 NO_CONVERGENCE, 1 SUCCESS, 2 FAILURE */
  int num_iterations;
} rba_cg_summary;

/* IterationSummary stage timings (src/rootba/solver/solver_summary.hpp:183-204),
 * seconds, measured with HIP events on the solver stream. */
typedef struct rba_iter_timings {
  double residual_evaluation_time;
  double stage1_time;
  double stage2_time;
  double compute_preconditioner_time;
  double solve_reduced_system_time;
  double back_substitution_time;
  double update_cameras_time;
  double hx_time;      /* sum over the TIMED right_multiply launches of the last solve:
                          every 8th product (calls 1, 9, 17, ...; env RBA_HX_TIMING_STRIDE,
                          1 = all, 0 = none) is bracketed by HIP events               */
  int hx_calls;        /* number of launches hx_time sums over                        */
} rba_iter_timings;

/* Sub-stage timers of the reference's UNSTAGED execution (`staged_execution = false`,
 * src/rootba/solver/linearizor_qr.cpp:94-112 and 166-187; IterationSummary fields of the same names,
 * solver_summary.hpp:183-204). With rba_options.staged_execution = 0 the kernel groups of stage 1 and
 * stage 2 are separated by HIP events and their times are accumulated here per LM iteration (reset with
 * the other timings); with staged_execution = 1 (default) all fields stay 0, as in the reference.
 * The kernels are the same in both modes; where this implementation fuses what the reference times
 * separately the mapping is:
 *   jacobian_evaluation_time       geometry pass (linearize_problem)
 *   scale_landmark_jacobian_time   camera-major Gram pass -> Jp_diag2, scaling vector (get_Jp_diag2; the Jl
 *                                  column scaling itself is part of the QR kernels)
 *   stage1_preconditioner_time     D G D scaling of the Gram blocks (get_Jp_T_Jp_blockdiag)
 *   perform_qr_time                Jl column scaling + Householder QR kernels (scale_Jl_cols + perform_qr)
 *   landmark_damping_time          always 0: the six damping rotations per landmark (set_landmark_damping) are
 *                                  evaluated inside the per-observation pass of stage 2 (measured faster than a
 *                                  pass of their own) and are timed with it
 *   scale_pose_jacobian_time       per-observation pass of stage 2: landmark damping, Jp column scaling, the
 *                                  observation's stage-2 record (set_landmark_damping + scale_Jp_cols + the
 *                                  per-column part of the damping)
 *   stage2_preconditioner_and_gradient_time   camera-major pass: SCHUR_JACOBI blocks and b
 *                                  (get_Q2TJp_T_Q2TJp_blockdiag + get_Q2TJp_T_Q2Tr, one fused pass) */
typedef struct rba_substage_timings {
  double jacobian_evaluation_time;
  double scale_landmark_jacobian_time;
  double stage1_preconditioner_time;
  double perform_qr_time;
  double landmark_damping_time;
  double scale_pose_jacobian_time;
  double stage2_preconditioner_and_gradient_time;
} rba_substage_timings;

/* One row of the LM log: subset of IterationSummary
 * (src/rootba/solver/solver_summary.hpp:99-204). */
typedef struct rba_lm_iteration {
  int iteration;
  int step_is_valid;
  int step_is_successful;
  int cg_iterations;
  int cg_termination;
  double cost;       /* cost.all.error after the step attempt   */
  double cost_valid; /* cost.valid.error                        */
  double lambda;     /* damping used for this iteration's solve */
  double relative_decrease;
  double l_diff;
  double inc_norm;
  double iteration_time;
  double stage1_time, stage2_time, precond_time, pcg_time, backsub_time,
      residual_time;
  /* ResidualInfo of the evaluated state (what BaLog::BaIteration derives
   * num_obs*, residual_block_*mean from, ba_log_utils.cpp:100-118) */
  int num_obs, num_obs_valid;
  double residual_sum, residual_sum_valid;
} rba_lm_iteration;

typedef struct rba_solver* rba_handle;

/* Defaults = examples/config/rootba_config_default.toml of the reference. */
void rba_default_options(rba_options* out);
const char* rba_last_error(void);
/* Number of visible HIP devices (>= 1 required for rba_create). */
int rba_device_count(int* out);

/* LinearizationQR ctor (linearization_qr.hpp:80-111) + LinearizorQR ctor option
 * mapping (linearizor_qr.cpp:58-68). Allocates all device storage. */
int rba_create(int dtype, int device, int32_t n_cams, int32_t n_lms,
               const int64_t* lm_obs_offsets, const int32_t* obs_cam_idx,
               const void* obs_xy, const rba_options* options, rba_handle* out);
int rba_destroy(rba_handle h);

/* The same for `n_gpus` devices of THIS process behind ONE handle ("..., int n_gpus, handle*"). The
 * reference builds one Linearizor for the whole problem in one process (src/rootba/solver/linearizor.cpp:133-150,
 * bal_bundle_adjustment.cpp:249-254), so this is the entry a drop-in binding can reach several GPUs through: the
 * library shards the landmarks itself - contiguous ranges in the caller's order, balanced by the bytes a landmark moves
 * per LM iteration in this layout (~120 per observation + 100) - creates one solver per device on a host thread of its
 * own, joins them (RCCL over the devices; devices that REPEAT in `device_ids` - a single-GPU test box - exchange
 * through host memory instead) and fans every call of this header out to all of them. State and per-landmark outputs
 * keep the caller's landmark order; replicated results (costs, b, increments, LM rows) are bit-identical on all
 * devices and reported from the first. `device_ids` NULL = devices 0 .. n_gpus - 1. rba_comm_init* do not apply to
 * such a handle (RBA_ERR_UNSUPPORTED). */
int rba_create_sharded(int dtype, int n_gpus, const int* device_ids, int32_t n_cams, int32_t n_lms,
                       const int64_t* lm_obs_offsets, const int32_t* obs_cam_idx, const void* obs_xy,
                       const rba_options* options, rba_handle* out);
/* The landmark ranges of a sharded handle: device r holds landmarks cuts[r] .. cuts[r + 1] - 1 (n_ranks + 1 entries;
 * a handle of rba_create reports one rank and writes nothing). */
int rba_get_shard_ranges(rba_handle h, int* n_ranks_out, int32_t* cuts_out, int max_cuts);

/* Landmark sharding across GPUs (no reference equivalent; the reference's
 * thread-level reductions become all-reduces). Each rank creates
 * a handle for ITS landmarks and ALL cameras; `unique_id` is the 128-byte
 * ncclUniqueId produced by rank 0 with rba_comm_unique_id and distributed by
 * the caller's launcher. Both solver types: SQUARE_ROOT all-reduces camera-sized
 * vectors and, per assembly, the reduced matrix; SCHUR_COMPLEMENT (since round 4)
 * all-reduces S, b (and Hpp for the power series) once per stage 2 in the united
 * block structure and runs its PCG replicated. */
int rba_comm_unique_id(void* out128);
int rba_comm_init(rba_handle h, int rank, int nranks, const void* unique_id128);
/* Same sharding with a caller-provided collective instead of RCCL (MPI, gloo,
 * a test harness ...): `fn` must all-reduce `count` elements of HOST memory in
 * place across the ranks and return 0. dtype: 0 f32, 1 f64, 2 i32; op: 0 sum,
 * 1 max. The library stages device <-> host around the call. */
typedef int (*rba_allreduce_fn)(void* ctx, void* host_buf, int64_t count, int dtype, int op);
int rba_comm_init_callback(rba_handle h, int rank, int nranks, rba_allreduce_fn fn, void* ctx);

/* What the handle's communicator looks like: *transport_out = 0 none (single GPU), 1 RCCL, 2 caller
 * callback; *nranks_out = number of ranks (for RCCL: as reported by ncclCommCount on the live
 * communicator, not the value passed in). */
int rba_comm_info(rba_handle h, int* rank_out, int* nranks_out, int* transport_out);
/* All-reduce traffic since rba_create: number of collectives, payload bytes, and device seconds
 * (HIP events around every collective on the solver stream; the callback transport reports host
 * wall time). Five call sites, SURVEY.md 8e: Jp_diag2 (+ failure flag), [b | block diagonal],
 * every matrix-free product, the assembled matrix (once per assembly), residual sums / l_diff. */
int rba_get_comm_stats(rba_handle h, int64_t* calls_out, int64_t* bytes_out, double* seconds_out);

/* BalProblem state upload/download (Camera::params()/from_params(),
 * bal_problem.hpp:84-95; copy_to/from_camera_state, bal_problem.cpp:570-588). */
int rba_set_state(rba_handle h, const void* cams10, const void* lms3);
/* (rba_get_state: either output may be NULL - the reference-side binding fetches the cameras after every apply and
 * the landmarks only when BalProblem is read, integration/rootba/solver/linearizor_hip.hpp) */
int rba_get_state(rba_handle h, void* cams10, void* lms3);
/* BalProblem::backup / restore (bal_problem.cpp:590-608), device-side copies. */
int rba_backup(rba_handle h);
int rba_restore(rba_handle h);

/* LinearizorBase::compute_error -> BalBundleAdjustmentHelper::compute_error
 * (linearizor_base.cpp:60-68, bal_bundle_adjustment_helper.cpp:68-109). */
int rba_compute_error(rba_handle h, rba_residual_info* out);

/* LinearizationQR::get_stage1 (linearization_qr.hpp:634-712) + the scaling
 * vector of LinearizorQR::linearize (linearizor_qr.cpp:78-138).
 * jp_diag2_out (9*n_cams, nullable) receives the squared column norms of Jp.
 * Returns RBA_NUMERICAL_FAILURE where the reference returns an empty vector. */
int rba_linearize(rba_handle h, void* jp_diag2_out);

/* LinearizorQR::solve (linearizor_qr.cpp:141-265): set_pose_damping,
 * get_stage2 (linearization_qr.hpp:716-815), BlockDiagonalPreconditioner
 * (preconditioner.hpp:79-120), PCG (conjugate_gradient.hpp:113-298). inc_out
 * (9*n_cams, not NULL: RBA_ERR_INVALID_ARGUMENT) is already negated
 * (linearizor_base.cpp:100). (rba_lm_step keeps the increment on the device
 * between its solve and its back-substitution; this entry always returns it.) */
int rba_solve(rba_handle h, double lambda, void* inc_out, rba_cg_summary* cg);

/* Pieces of rba_solve, exposed because the reference's tests call them
 * directly (linearization_qr.test.cpp:150-187): get_stage2 -> (b, SCHUR_JACOBI
 * blocks 81*n_cams, nullable) and right_multiply (linearization_qr.hpp:821-825). */
int rba_stage2(rba_handle h, double lambda, void* b_out, void* blocks_out);
int rba_right_multiply(rba_handle h, const void* x, void* y);
/* The same product through the explicitly assembled reduced matrix (see
 * rba_options.explicit_after); for tests. Needs rba_stage2 / rba_solve first. */
int rba_right_multiply_explicit(rba_handle h, const void* x, void* y);

/* LinearizorQR::apply (linearizor_qr.cpp:268-291): back_substitute
 * (linearization_qr.hpp:165-179, landmark_block_base.ipp:212-284), un-scale,
 * camera retraction. l_diff_out is NaN (status 1) on a non-finite update.
 * inc (9*n_cams) must not be NULL. */
int rba_apply(rba_handle h, const void* inc, double* l_diff_out);
/* LinearizationQR::back_substitute alone (landmarks only; reference tests call
 * it with a random increment, linearization_qr.test.cpp:194-200). */
int rba_back_substitute(rba_handle h, const void* inc, double* l_diff_out);

/* optimize_lm_ours (src/rootba/solver/bal_bundle_adjustment.cpp:249-544): the
 * serial LM loop, same lambda schedule and termination. Writes at most max_rows
 * log rows; *n_rows_out = rows produced; *termination_out = 0 NO_CONVERGENCE,
 * 1 CONVERGENCE, -1 numerical failure. */
int rba_optimize_lm(rba_handle h, rba_lm_iteration* log, int max_rows,
                    int* n_rows_out, int* termination_out);

/* The same loop, resumable: rba_lm_begin resets lambda / counters, every
 * rba_lm_step performs exactly ONE LM iteration (= one row of the reference's
 * iteration log; iteration 0 is the evaluation-only row) and sets *more_out to
 * 0 once the loop has terminated. rba_optimize_lm == begin + step until 0. */
int rba_lm_begin(rba_handle h);
int rba_lm_step(rba_handle h, rba_lm_iteration* row, int* more_out);
int rba_lm_termination(rba_handle h, int* termination_out);
/* Block until all work queued on the handle's stream has finished. */
int rba_synchronize(rba_handle h);

int rba_get_timings(rba_handle h, rba_iter_timings* out);
int rba_get_substage_timings(rba_handle h, rba_substage_timings* out);
/* Profiling aid: streams the Jacobian-row storage (72 bytes per observation in float) once with 4-byte
 * (vec_width = 1) or 16-byte (vec_width = 4) loads and reports the bytes read -
 * a known byte count for calibrating rocprofv3's FETCH_SIZE on gfx950. */
int rba_debug_read_blocks(rba_handle h, int vec_width, int64_t* bytes_out);

/* Introspection used by the parity tests (invariants of SURVEY.md §8c). */
int rba_get_jl_col_scale(rba_handle h, void* out3_per_lm);
int rba_get_pose_scaling(rba_handle h, void* out9_per_cam);
/* R (3x3 upper, 6 scalars r00 r01 r02 r11 r12 r22) and Q1^T r (3) per landmark,
 * undamped (damped = 0) or with the current landmark damping (damped = 1). */
int rba_get_landmark_R(rba_handle h, int damped, void* R6_per_lm,
                       void* q1tr3_per_lm);
/* |Q2^T r| per landmark of the current linearisation point (undamped; the part of the residual that stays after
 * the landmark is marginalised, landmark_block_base.ipp:717-743 applied to the last column) - with R and Q1^T r the
 * whole of what the landmark QR hands on; the accuracy ensemble of tests/test_gpu_qr_accuracy.py compares it. */
int rba_get_landmark_q2tr_norm(rba_handle h, void* out1_per_lm);
/* Algorithmic byte/flop counts of the resident topology. */
int rba_get_problem_stats(rba_handle h, int64_t* block_storage_bytes,
                          int64_t* hx_algorithmic_bytes, int64_t* hx_flops);

/* Bytes that one launch group must move through HBM in THIS library's data layout (DESIGN.md 2-3;
 * compulsory traffic: every record read or written once, camera-sized vectors once per kernel).
 * bench.py divides them by the measured stage times for the per-stage and whole-iteration rooflines. */
typedef struct rba_byte_model {
  int64_t compute_error;       /* one cost evaluation                                        */
  int64_t stage1;              /* one linearisation: geometry, landmark QR, camera-major sums */
  int64_t stage2;              /* landmark damping + per-observation rotation + camera-major sums */
  int64_t product_matrix_free; /* one H x from the QR factors                                 */
  int64_t product_assembled;   /* one S x on the assembled block-CSR matrix                   */
  int64_t assembly;            /* one assembly of the reduced camera matrix                   */
  int64_t pcg_vectors;         /* vector / preconditioner work of one PCG iteration           */
  int64_t back_substitution;   /* back-substitution + landmark update                         */
  int64_t persistent_solve;    /* persistent PCG kernel (kernels_pcgp.hpp), once per solve: the matrix into the
                                  register files (full storage), row state, M^-1; 0 when the matrix does not fit */
  int64_t persistent_iteration;/* ... and per iteration: the exchanged records of z and of the partial sums */
} rba_byte_model;
int rba_get_byte_model(rba_handle h, rba_byte_model* out);
/* What the PCG solves since rba_create actually executed (launches queued past a solve's termination are no-ops and
 * not counted): matrix-free products, products on the assembled matrix, assemblies of that matrix, PCG iterations.
 * With rba_byte_model this prices the PCG phase of a run (bench.py: roofline.whole_iteration). */
typedef struct rba_pcg_counters {
  int64_t products_matrix_free;
  int64_t products_assembled;
  int64_t assemblies;
  int64_t iterations;
  int64_t solves_repeated_matrix_free; /* solves whose assembled operator broke down (S + E lost definiteness) */
  int64_t early_switches;              /* solves switched to the assembled matrix at iteration 5 (rising zeta) */
  int64_t solves_persistent;           /* solves whose iterations on the assembled matrix ran as ONE persistent kernel
                                          with the matrix in the register files (kernels_pcgp.hpp) */
  int64_t products_assembled_resident; /* ... the part of products_assembled those solves executed (no matrix bytes moved) */
  int64_t iterations_resident;         /* ... and of iterations (their vector work stays on chip) */
  int64_t cost_evaluations;            /* cost evaluations launched (the LM loop reuses the trial evaluation of an
                                          accepted step instead of repeating it) */
} rba_pcg_counters;
int rba_get_pcg_counters(rba_handle h, rba_pcg_counters* out);
/* The block structure of the reduced camera matrix S a long PCG solve assembles (rba_options.explicit_after; for
 * solver_type SCHUR_COMPLEMENT the matrix of LinearizorSC, linearizor_sc.cpp:101-140) - i.e. which regime a workload is
 * in: a banded S of a few per cent density fits the register files and its PCG runs as one persistent kernel; a nearly
 * dense S (heavy-tailed co-visibility) streams from HBM in two launches per iteration. All zero when no matrix is ever
 * assembled (explicit_after = 0, or the pair lists exceed their budget). */
typedef struct rba_reduced_matrix_info {
  int64_t blocks_stored;      /* 9x9 blocks held in memory (square-root solver: half storage, diagonal included) */
  int64_t blocks_full;        /* structural non-zero blocks of the full symmetric matrix */
  double density;             /* blocks_full / n_cams^2 */
  int64_t bytes_stored;       /* blocks_stored x 81 x sizeof(value): double for the square-root solver */
  int resident_in_registers;  /* 1: PCG solves on S run as ONE persistent kernel, S in the register files */
  int persistent_workgroups;  /* workgroups of that kernel (one per compute unit), 0 if the matrix does not fit */
} rba_reduced_matrix_info;
int rba_get_reduced_matrix_info(rba_handle h, rba_reduced_matrix_info* out);

#ifdef __cplusplus
}
#endif
#endif /* ROOTBA_HIP_H_ */
