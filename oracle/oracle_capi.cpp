// oracle/oracle_capi.cpp — C ABI over the CPU restatement (rootba_oracle.hpp)
// so tests/ and bench.py's cpu_baseline leg can drive it through ctypes.
// TEST INFRASTRUCTURE ONLY — see the header of rootba_oracle.hpp.
#include "rootba_oracle.hpp"

using orc::Options;

extern "C" {

// mirrors orc::Options field by field (and `rba_options` in
// include/rootba_hip.h)
struct orc_options {
  int use_householder;
  int use_valid_projections_only;
  int robust_norm;
  double huber_parameter;
  double jacobi_scaling_eps;
  int preconditioner_type;
  int reduction_alg;
  int power_order;
  int min_cg_it;
  int max_cg_it;
  double eta;
  int num_threads;
  int max_num_iterations;
  double min_relative_decrease;
  double initial_trust_region_radius;
  double min_trust_region_radius;
  double max_trust_region_radius;
  double function_tolerance;
  double initial_vee;
  double vee_factor;
  int optimized_cost;
  int staged_execution;
  int implicit_q;  // product-only switch (ignored here)
  int solver_type;
  int explicit_after;  // product-only switch (ignored here)
};

struct orc_residual_info {
  int all_num_obs;
  double all_error;
  double all_residual_sum;
  int valid_num_obs;
  double valid_error;
  double valid_residual_sum;
  int is_numerically_valid;
};

struct orc_cg_summary {
  int termination_type;
  int num_iterations;
};

void orc_default_options(orc_options* o) {
  Options d;
  o->use_householder = d.use_householder;
  o->use_valid_projections_only = d.use_valid_projections_only;
  o->robust_norm = d.robust_norm;
  o->huber_parameter = d.huber_parameter;
  o->jacobi_scaling_eps = d.jacobi_scaling_eps;
  o->preconditioner_type = d.preconditioner_type;
  o->reduction_alg = d.reduction_alg;
  o->power_order = d.power_order;
  o->min_cg_it = d.min_cg_it;
  o->max_cg_it = d.max_cg_it;
  o->eta = d.eta;
  o->num_threads = d.num_threads;
  o->max_num_iterations = d.max_num_iterations;
  o->min_relative_decrease = d.min_relative_decrease;
  o->initial_trust_region_radius = d.initial_trust_region_radius;
  o->min_trust_region_radius = d.min_trust_region_radius;
  o->max_trust_region_radius = d.max_trust_region_radius;
  o->function_tolerance = d.function_tolerance;
  o->initial_vee = d.initial_vee;
  o->vee_factor = d.vee_factor;
  o->optimized_cost = d.optimized_cost;
  o->staged_execution = d.staged_execution;
  o->implicit_q = 0;
  o->solver_type = d.solver_type;
  o->explicit_after = 0;
}

int orc_sizeof_lm_iteration() { return int(sizeof(orc::LmIteration)); }
int orc_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
}  // extern "C"

static Options to_options(const orc_options* o) {
  Options d;
  d.use_householder = o->use_householder;
  d.use_valid_projections_only = o->use_valid_projections_only;
  d.robust_norm = o->robust_norm;
  d.huber_parameter = o->huber_parameter;
  d.jacobi_scaling_eps = o->jacobi_scaling_eps;
  d.preconditioner_type = o->preconditioner_type;
  d.reduction_alg = o->reduction_alg;
  d.power_order = o->power_order;
  d.min_cg_it = o->min_cg_it;
  d.max_cg_it = o->max_cg_it;
  d.eta = o->eta;
  d.num_threads = o->num_threads;
  d.max_num_iterations = o->max_num_iterations;
  d.min_relative_decrease = o->min_relative_decrease;
  d.initial_trust_region_radius = o->initial_trust_region_radius;
  d.min_trust_region_radius = o->min_trust_region_radius;
  d.max_trust_region_radius = o->max_trust_region_radius;
  d.function_tolerance = o->function_tolerance;
  d.initial_vee = o->initial_vee;
  d.vee_factor = o->vee_factor;
  d.optimized_cost = o->optimized_cost;
  d.staged_execution = o->staged_execution;
  d.solver_type = o->solver_type;
  return d;
}

template <class S>
static void fill_ri(const orc::ResidualInfo& ri, orc_residual_info* out) {
  out->all_num_obs = ri.all.num_obs;
  out->all_error = ri.all.error;
  out->all_residual_sum = ri.all.residual_sum;
  out->valid_num_obs = ri.valid.num_obs;
  out->valid_error = ri.valid.error;
  out->valid_residual_sum = ri.valid.residual_sum;
  out->is_numerically_valid = ri.is_numerically_valid;
}

#define ORC_DEFINE_API(SUF, S)                                                 \
  extern "C" {                                                                 \
  void* orc_create_##SUF(int n_cams, int n_lms, const int64_t* lm_obs_offsets, \
                         const int32_t* obs_cam_idx, const S* obs_xy,          \
                         const orc_options* opt) {                             \
    return new orc::Oracle<S>(n_cams, n_lms, lm_obs_offsets, obs_cam_idx,      \
                              obs_xy, to_options(opt));                        \
  }                                                                            \
  void orc_destroy_##SUF(void* h) { delete static_cast<orc::Oracle<S>*>(h); }  \
  int orc_num_threads_##SUF(void* h) {                                         \
    return static_cast<orc::Oracle<S>*>(h)->n_threads();                       \
  }                                                                            \
  void orc_set_state_##SUF(void* h, const S* cams, const S* lms) {             \
    auto* o = static_cast<orc::Oracle<S>*>(h);                                 \
    std::copy(cams, cams + o->cams().size(), o->cams().begin());               \
    std::copy(lms, lms + o->lms().size(), o->lms().begin());                   \
  }                                                                            \
  void orc_get_state_##SUF(void* h, S* cams, S* lms) {                         \
    auto* o = static_cast<orc::Oracle<S>*>(h);                                 \
    std::copy(o->cams().begin(), o->cams().end(), cams);                       \
    std::copy(o->lms().begin(), o->lms().end(), lms);                          \
  }                                                                            \
  void orc_backup_##SUF(void* h) { static_cast<orc::Oracle<S>*>(h)->backup(); } \
  void orc_restore_##SUF(void* h) {                                            \
    static_cast<orc::Oracle<S>*>(h)->restore();                                \
  }                                                                            \
  void orc_compute_error_##SUF(void* h, orc_residual_info* out) {              \
    orc::ResidualInfo ri;                                                      \
    static_cast<orc::Oracle<S>*>(h)->compute_error(ri);                        \
    fill_ri<S>(ri, out);                                                       \
  }                                                                            \
  /* get_stage1: returns 0 ok, 1 numerical failure */                          \
  int orc_stage1_##SUF(void* h, S* jp_diag2_out, S* jacobi_blocks_out) {       \
    auto* o = static_cast<orc::Oracle<S>*>(h);                                 \
    std::vector<S> d, blocks;                                                  \
    const bool ok = o->get_stage1(d, jacobi_blocks_out ? &blocks : nullptr);   \
    if (jp_diag2_out) std::copy(d.begin(), d.end(), jp_diag2_out);             \
    if (jacobi_blocks_out)                                                     \
      std::copy(blocks.begin(), blocks.end(), jacobi_blocks_out);              \
    return ok ? 0 : 1;                                                         \
  }                                                                            \
  void orc_set_pose_damping_##SUF(void* h, S lambda) {                         \
    static_cast<orc::Oracle<S>*>(h)->set_pose_damping(lambda);                 \
  }                                                                            \
  /* get_stage2: jacobian_scaling nullable, blocks_out nullable */             \
  void orc_stage2_##SUF(void* h, S lambda, const S* jacobian_scaling,          \
                        S* b_out, S* blocks_out) {                             \
    auto* o = static_cast<orc::Oracle<S>*>(h);                                 \
    std::vector<S> b, blocks;                                                  \
    o->get_stage2(lambda, jacobian_scaling, blocks_out ? &blocks : nullptr,    \
                  b);                                                          \
    std::copy(b.begin(), b.end(), b_out);                                      \
    if (blocks_out) std::copy(blocks.begin(), blocks.end(), blocks_out);       \
  }                                                                            \
  void orc_right_multiply_##SUF(void* h, const S* x, S* y) {                   \
    static_cast<orc::Oracle<S>*>(h)->right_multiply(x, y);                     \
  }                                                                            \
  S orc_back_substitute_##SUF(void* h, const S* pose_inc) {                    \
    return static_cast<orc::Oracle<S>*>(h)->back_substitute_all(pose_inc);     \
  }                                                                            \
  /* LinearizorQR::linearize / solve / apply */                                \
  int orc_linearize_##SUF(void* h) {                                           \
    return static_cast<orc::Oracle<S>*>(h)->linearize() ? 0 : 1;               \
  }                                                                            \
  void orc_solve_##SUF(void* h, S lambda, S* inc_out, orc_cg_summary* cg) {    \
    auto* o = static_cast<orc::Oracle<S>*>(h);                                 \
    orc::CgSummary s;                                                          \
    std::vector<S> inc = o->solve(lambda, &s);                                 \
    std::copy(inc.begin(), inc.end(), inc_out);                                \
    if (cg) {                                                                  \
      cg->termination_type = s.termination_type;                               \
      cg->num_iterations = s.num_iterations;                                   \
    }                                                                          \
  }                                                                            \
  S orc_apply_##SUF(void* h, const S* inc) {                                   \
    auto* o = static_cast<orc::Oracle<S>*>(h);                                 \
    std::vector<S> v(inc, inc + size_t(9) * o->n_cams());                      \
    return o->apply(std::move(v));                                             \
  }                                                                            \
  int orc_optimize_lm_##SUF(void* h, void* log, int max_rows,                  \
                            int* termination) {                                \
    return static_cast<orc::Oracle<S>*>(h)->optimize_lm(                       \
        static_cast<orc::LmIteration*>(log), max_rows, termination);           \
  }                                                                            \
  /* introspection for invariant tests */                                      \
  void orc_block_shape_##SUF(void* h, int l, int* rows, int* cols,             \
                             int* lm_idx) {                                    \
    auto* o = static_cast<orc::Oracle<S>*>(h);                                 \
    *rows = o->rows(l);                                                        \
    *cols = o->cols(l);                                                        \
    *lm_idx = o->lm_idx(l);                                                    \
  }                                                                            \
  void orc_get_block_##SUF(void* h, int l, S* out) {                           \
    auto* o = static_cast<orc::Oracle<S>*>(h);                                 \
    const S* b = o->block(l);                                                  \
    std::copy(b, b + size_t(o->rows(l)) * o->cols(l), out);                    \
  }                                                                            \
  void orc_get_jl_col_scale_##SUF(void* h, S* out) {                           \
    auto* o = static_cast<orc::Oracle<S>*>(h);                                 \
    std::copy(o->jl_col_scale().begin(), o->jl_col_scale().end(), out);        \
  }                                                                            \
  void orc_get_pose_scaling_##SUF(void* h, S* out) {                           \
    auto* o = static_cast<orc::Oracle<S>*>(h);                                 \
    std::copy(o->pose_jacobian_scaling().begin(),                              \
              o->pose_jacobian_scaling().end(), out);                          \
  }                                                                            \
  void orc_get_last_b_##SUF(void* h, S* out) {                                 \
    auto* o = static_cast<orc::Oracle<S>*>(h);                                 \
    std::copy(o->last_b().begin(), o->last_b().end(), out);                    \
  }                                                                            \
  void orc_get_precond_blocks_##SUF(void* h, S* out) {                         \
    auto* o = static_cast<orc::Oracle<S>*>(h);                                 \
    std::copy(o->precond_blocks().begin(), o->precond_blocks().end(), out);    \
  }                                                                            \
  /* per-observation geometry: res[2], Jp[12], Ji[6], Jl[6]; returns valid */  \
  int orc_linearize_point_##SUF(const S* obs, const S* p_w, const S* cam,      \
                                int ignore_validity_check, S* res, S* Jp,      \
                                S* Ji, S* Jl) {                                \
    return orc::linearize_point<S>(obs, p_w, cam, ignore_validity_check != 0,  \
                                   res, Jp, Ji, Jl)                            \
               ? 1                                                             \
               : 0;                                                            \
  }                                                                            \
  void orc_apply_inc_camera_##SUF(S* cam, const S* inc9) {                     \
    orc::apply_inc_camera<S>(cam, inc9);                                       \
  }                                                                            \
  /* power-series preconditioner (after orc_linearize): prepare + apply */      \
  void orc_power_precond_##SUF(void* h, S lambda, const S* b, S* x) {          \
    auto* o = static_cast<orc::Oracle<S>*>(h);                                 \
    o->power_precond_prepare(lambda);                                          \
    o->power_precond_solve(b, x);                                              \
  }                                                                            \
  /* explicit Schur complement cross-check (dense H, small problems) */        \
  void orc_sc_build_##SUF(void* h, S lambda, S pose_lambda,                    \
                          const S* pose_scaling, S* H_out, S* b_out,           \
                          S* jp_diag2_out) {                                   \
    auto* o = static_cast<orc::Oracle<S>*>(h);                                 \
    std::vector<S> H, b, d;                                                    \
    o->sc_build(lambda, pose_lambda, pose_scaling, H_out ? &H : nullptr, b,    \
                jp_diag2_out ? &d : nullptr);                                  \
    if (H_out) std::copy(H.begin(), H.end(), H_out);                           \
    std::copy(b.begin(), b.end(), b_out);                                      \
    if (jp_diag2_out) std::copy(d.begin(), d.end(), jp_diag2_out);             \
  }                                                                            \
  S orc_sc_back_substitute_##SUF(void* h, S lambda, const S* pose_scaling,     \
                                 const S* pose_inc) {                          \
    return static_cast<orc::Oracle<S>*>(h)->sc_back_substitute(                \
        lambda, pose_scaling, pose_inc);                                       \
  }                                                                            \
  }

ORC_DEFINE_API(f32, float)
ORC_DEFINE_API(f64, double)
