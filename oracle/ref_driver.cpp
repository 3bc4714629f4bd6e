// oracle/ref_driver.cpp - C ABI over the REFERENCE'S OWN solver sources, for the parity tests.
//
// TEST INFRASTRUCTURE ONLY (tests/, never the product). Built by oracle/build_ref.sh into
// oracle/_ref/librootba_ref.so when /root/reference is present:
//
//   compiled UNMODIFIED from /root/reference/src/rootba (nothing is copied into this repository):
//     bal/bal_bundle_adjustment_helper.cpp  bal/residual_info.cpp  bal/solver_options.cpp
//     bal/bal_problem.cpp                   qr/landmark_block.cpp  qr/impl/landmark_block_dynamic.cpp
//     solver/linearizor.cpp  solver/linearizor_base.cpp  solver/linearizor_qr.cpp
//     solver/linearizor_sc.cpp  solver/linearizor_power_sc.cpp  solver/bal_bundle_adjustment.cpp
//     (+ every header they include: qr/linearization_qr.hpp, qr/impl/landmark_block_base.ipp,
//      cg/conjugate_gradient.hpp, cg/preconditioner.hpp, cg/block_sparse_matrix.hpp, sc/*.hpp, ...)
//   against oracle/ref_shims/: stand-ins, written for this repository, for the third-party libraries
//     that are absent here (Eigen, Sophus, basalt-headers, oneTBB, glog, fmt, abseil, magic_enum) and
//     for the reference's option-reflection / log-file / cereal plumbing.
//
// So the calls below run the reference's block layout, linearisation, QR, damping, stage 1 / stage 2,
// preconditioners, PCG, back-substitution, LM loop, BAL loader, normalisation and filtering as the
// reference wrote them. What they do NOT pin is the arithmetic inside the third-party libraries (see
// the headers of oracle/ref_shims/Eigen/Dense, sophus/so3.hpp, basalt/camera/bal_camera.hpp).
//
// This file: (a) the few functions of the reference that live in translation units which cannot be
// compiled here (util/tbb_utils.cpp, util/system_utils.cpp: thread-count and memory bookkeeping);
// (b) the extern "C" entry points, which mirror oracle/oracle_capi.cpp so that tests can drive the
// restated oracle and the reference build with the same code.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "rootba/bal/bal_bundle_adjustment_helper.hpp"
#include "rootba/bal/bal_dataset_options.hpp"
#include "rootba/bal/bal_pipeline_summary.hpp"
#include "rootba/bal/bal_problem.hpp"
#include "rootba/cg/preconditioner.hpp"
#include "rootba/qr/landmark_block_dynamic.hpp"
#include "rootba/qr/linearization_qr.hpp"
#include "rootba/solver/bal_bundle_adjustment.hpp"
#include "rootba/solver/host_state_sync.hpp"  // (integration/: the binding's host <-> device state protocol)
#include "rootba/solver/linearizor.hpp"
#include "rootba/solver/solver_summary.hpp"
#include "rootba/util/system_utils.hpp"
#include "rootba/util/tbb_utils.hpp"

// ---- (a) bookkeeping functions of util/tbb_utils.cpp and util/system_utils.cpp -------------------
namespace rootba {
int hardware_concurrency() { return 1; }
int tbb_task_arena_max_concurrency() { return 1; }
int tbb_global_max_allowed_parallelism() { return 1; }
int tbb_effective_max_concurrency() { return 1; }
struct ScopedTbbThreadLimit::Impl {};
ScopedTbbThreadLimit::ScopedTbbThreadLimit(int /*num_threads*/) : impl_(nullptr) {}
ScopedTbbThreadLimit::~ScopedTbbThreadLimit() = default;
struct TbbConcurrencyObserver::Impl {};
TbbConcurrencyObserver::TbbConcurrencyObserver() : impl_(nullptr) {}
TbbConcurrencyObserver::~TbbConcurrencyObserver() = default;
int TbbConcurrencyObserver::get_current_concurrency() const { return 1; }
int TbbConcurrencyObserver::get_peak_concurrency() const { return 1; }
bool get_memory_info(MemoryInfo& /*info*/) { return false; }
}  // namespace rootba

namespace {

using rootba::BalProblem;
using rootba::SolverOptions;

// same layout as orc_options (oracle/oracle_capi.cpp) and `Options` in oracle/oracle.py
struct ref_options {
  int use_householder;
  int use_valid_projections_only;
  int robust_norm;
  double huber_parameter;
  double jacobi_scaling_eps;
  int preconditioner_type;  // this repository's numbering: 0 JACOBI, 1 SCHUR_JACOBI, 2 POWER_SCHUR_COMPLEMENT
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
  int implicit_q;
  int solver_type;  // 0 SQUARE_ROOT, 1 SCHUR_COMPLEMENT, 2 POWER_SCHUR_COMPLEMENT
  int explicit_after;
};
struct ref_residual_info {
  int all_num_obs;
  double all_error;
  double all_residual_sum;
  int valid_num_obs;
  double valid_error;
  double valid_residual_sum;
  int is_numerically_valid;
};
struct ref_cg_summary {
  int termination_type;
  int num_iterations;
};
struct ref_lm_iteration {
  int iteration;
  int step_is_valid;
  int step_is_successful;
  int cg_iterations;
  int cg_termination;
  double cost;
  double cost_valid;
  double lambda_;
  double relative_decrease;
  double l_diff;
  double inc_norm;
  double iteration_time;
  double stage1_time;
  double stage2_time;
  double precond_time;
  double pcg_time;
  double backsub_time;
  double residual_time;
  int num_obs;
  int num_obs_valid;
  double residual_sum;
  double residual_sum_valid;
};

// the reference's defaults are the ones of ITS SolverOptions declaration; the fields of ref_options
// overwrite them one by one
SolverOptions to_solver_options(const ref_options& o) {
  SolverOptions s;
  s.verbosity_level = 0;
  s.use_householder_marginalization = o.use_householder != 0;
  // optimized_cost decides the validity check (SolverOptions::use_projection_validity_check)
  s.optimized_cost = static_cast<SolverOptions::OptimizedCost>(o.optimized_cost);
  s.residual.robust_norm = static_cast<rootba::BalResidualOptions::RobustNorm>(o.robust_norm);
  s.residual.huber_parameter = o.huber_parameter;
  s.jacobi_scaling_epsilon = o.jacobi_scaling_eps;
  switch (o.preconditioner_type) {
    case 0:
      s.preconditioner_type = SolverOptions::PreconditionerType::JACOBI;
      break;
    case 1:
      s.preconditioner_type = SolverOptions::PreconditionerType::SCHUR_JACOBI;
      break;
    default:
      s.preconditioner_type = SolverOptions::PreconditionerType::POWER_SCHUR_COMPLEMENT;
  }
  s.reduction_alg = o.reduction_alg;
  s.power_order = o.power_order;
  s.min_linear_solver_iterations = o.min_cg_it;
  s.max_linear_solver_iterations = o.max_cg_it;
  s.eta = o.eta;
  s.num_threads = o.num_threads;
  s.max_num_iterations = o.max_num_iterations;
  s.min_relative_decrease = o.min_relative_decrease;
  s.initial_trust_region_radius = o.initial_trust_region_radius;
  s.min_trust_region_radius = o.min_trust_region_radius;
  s.max_trust_region_radius = o.max_trust_region_radius;
  s.function_tolerance = o.function_tolerance;
  s.initial_vee = o.initial_vee;
  s.vee_factor = o.vee_factor;
  s.staged_execution = o.staged_execution != 0;
  s.solver_type = o.solver_type == 0   ? SolverOptions::SolverType::SQUARE_ROOT
                  : o.solver_type == 1 ? SolverOptions::SolverType::SCHUR_COMPLEMENT
                                       : SolverOptions::SolverType::POWER_SCHUR_COMPLEMENT;
  return s;
}

void fill_ri(const rootba::ResidualInfo& ri, ref_residual_info* out) {
  out->all_num_obs = ri.all.num_obs;
  out->all_error = ri.all.error;
  out->all_residual_sum = ri.all.residual_sum;
  out->valid_num_obs = ri.valid.num_obs;
  out->valid_error = ri.valid.error;
  out->valid_residual_sum = ri.valid.residual_sum;
  out->is_numerically_valid = ri.is_numerically_valid ? 1 : 0;
}

// termination type of the last PCG run, from the message ConjugateGradientsSolver::solve wrote
// (src/rootba/cg/conjugate_gradient.hpp:113-298); the numbers are Summary::TerminationType
int cg_termination_from_message(const std::string& m) {
  if (m.find("Convergence") != std::string::npos) return 1;        // LINEAR_SOLVER_SUCCESS
  if (m.find("Numerical failure") != std::string::npos) return 2;  // LINEAR_SOLVER_FAILURE
  return 0;                                                        // NO_CONVERGENCE (max its / indefinite)
}

template <class S>
struct LqrOpen : public rootba::LinearizationQR<S, 9> {
  using Base = rootba::LinearizationQR<S, 9>;
  using Base::Base;
  const std::vector<typename Base::LandmarkBlockPtr>& blocks() const { return this->landmark_blocks_; }
};
template <class S>
struct BlockOpen : public rootba::LandmarkBlockDynamic<S, 9> {
  using B = rootba::LandmarkBlockDynamic<S, 9>;
  static Eigen::Matrix<S, 3, 1> jl_col_scale(const B& b) { return b.*(&BlockOpen::Jl_col_scale_); }
};

template <class S>
struct Handle {
  using VecX = Eigen::Matrix<S, Eigen::Dynamic, 1>;
  BalProblem<S> problem;
  ref_options ropt;
  SolverOptions sopt;
  rootba::SolverSummary summary;
  rootba::IterationSummary it_summary;
  std::unique_ptr<rootba::Linearizor<S>> lin;  // through the reference's factory (solver_type)
  std::unique_ptr<LqrOpen<S>> lqr;             // stage-level access, square-root solver only
  rootba::IndexedBlocks<S> blocks;

  rootba::Linearizor<S>& linearizor() {
    if (!lin) {
      lin = rootba::Linearizor<S>::create(problem, sopt, &summary);
      lin->start_iteration(&it_summary);
    }
    return *lin;
  }
  LqrOpen<S>& linearization() {
    if (!lqr) {
      typename rootba::LinearizationQR<S, 9>::Options o;  // as LinearizorQR's constructor fills them
      o.lb_options.use_householder = sopt.use_householder_marginalization;
      o.lb_options.use_valid_projections_only = sopt.use_projection_validity_check();
      o.lb_options.jacobi_scaling_eps =
          sopt.jacobi_scaling_epsilon > 0 ? S(sopt.jacobi_scaling_epsilon) : Sophus::Constants<S>::epsilonSqrt();
      o.reduction_alg = sopt.reduction_alg;
      o.lb_options.residual_options = sopt.residual;
      lqr = std::make_unique<LqrOpen<S>>(problem, o);
    }
    return *lqr;
  }
};

template <class S>
void blocks_out(const rootba::IndexedBlocks<S>& blocks, int n_cams, S* out) {
  std::memset(out, 0, sizeof(S) * 81 * size_t(n_cams));
  for (const auto& [key, m] : blocks) {
    if (key.first != key.second) continue;
    for (int a = 0; a < 9; ++a)
      for (int b = 0; b < 9; ++b) out[81 * key.first + 9 * a + b] = m(a, b);
  }
}

template <class S>
Handle<S>* h_create(int n_cams, int n_lms, const int64_t* off, const int32_t* cam, const S* xy, const ref_options* o) {
  // the reference has ONE knob: LinearizorQR derives the blocks' use_valid_projections_only from
  // SolverOptions::use_projection_validity_check(), i.e. from optimized_cost (linearizor_qr.cpp:58-68)
  if ((o->use_valid_projections_only != 0) != (o->optimized_cost != 0)) {
    std::cerr << "oracle/_ref: use_valid_projections_only = " << o->use_valid_projections_only
              << " with optimized_cost = " << o->optimized_cost << " is not a configuration of the reference\n";
    return nullptr;
  }
  auto* h = new Handle<S>();
  h->ropt = *o;
  h->sopt = to_solver_options(*o);
  h->problem.cameras().resize(n_cams);
  h->problem.landmarks().resize(n_lms);
  for (int l = 0; l < n_lms; ++l) {
    auto& lm = h->problem.landmarks()[l];
    lm.p_w.setZero();
    for (int64_t i = off[l]; i < off[l + 1]; ++i) {
      typename BalProblem<S>::Observation ob;
      ob.pos = Eigen::Matrix<S, 2, 1>(xy[2 * i], xy[2 * i + 1]);
      lm.obs[cam[i]] = ob;
    }
  }
  h->problem.set_quiet(true);
  return h;
}
template <class S>
void h_set_state(Handle<S>* h, const S* cams, const S* lms) {
  using VecX = typename Handle<S>::VecX;
  struct Announce {  // (a LinearizorHIP keeps the state on the device: tell it that BalProblem is authoritative again)
    Handle<S>* h;
    ~Announce() {
      if (auto* s = dynamic_cast<rootba::HostStateSync*>(h->lin.get())) s->host_state_changed();
    }
  } announce{h};
  for (size_t c = 0; c < h->problem.cameras().size(); ++c) {
    VecX p(10);
    for (int k = 0; k < 10; ++k) p(k) = cams[10 * c + k];
    h->problem.cameras()[c].from_params(p);
  }
  for (size_t l = 0; l < h->problem.landmarks().size(); ++l)
    h->problem.landmarks()[l].p_w = Eigen::Matrix<S, 3, 1>(lms[3 * l], lms[3 * l + 1], lms[3 * l + 2]);
}
template <class S>
void h_get_state(Handle<S>* h, S* cams, S* lms) {
  if (auto* s = dynamic_cast<rootba::HostStateSync*>(h->lin.get())) s->sync_host();  // (landmarks come back on demand)
  for (size_t c = 0; c < h->problem.cameras().size(); ++c) {
    const auto p = h->problem.cameras()[c].params();
    for (int k = 0; k < 10; ++k) cams[10 * c + k] = p(k);
  }
  for (size_t l = 0; l < h->problem.landmarks().size(); ++l)
    for (int k = 0; k < 3; ++k) lms[3 * l + k] = h->problem.landmarks()[l].p_w(k);
}
template <class S>
void h_compute_error(Handle<S>* h, ref_residual_info* out) {
  rootba::ResidualInfo ri;
  rootba::BalBundleAdjustmentHelper<S>::compute_error(h->problem, h->sopt, ri);
  fill_ri(ri, out);
}
// Linearizor::compute_error of the linearizor the factory returned (LinearizorBase::compute_error, or the binding's)
template <class S>
void h_linearizor_compute_error(Handle<S>* h, ref_residual_info* out) {
  rootba::ResidualInfo ri;
  h->linearizor().compute_error(ri);
  fill_ri(ri, out);
}
// LinearizationQR::get_stage1 (linearization_qr.hpp:634-712): 1 = numerical failure (empty vector)
template <class S>
int h_stage1(Handle<S>* h, S* jp_diag2, S* jacobi_blocks) {
  auto& lqr = h->linearization();
  rootba::IndexedBlocks<S> blocks;
  const auto d = lqr.get_stage1(jacobi_blocks ? &blocks : nullptr);
  if (d.size() == 0) return 1;
  for (Eigen::Index i = 0; i < d.size(); ++i) jp_diag2[i] = d(i);
  if (jacobi_blocks) blocks_out(blocks, int(h->problem.cameras().size()), jacobi_blocks);
  return 0;
}
template <class S>
void h_set_pose_damping(Handle<S>* h, S lambda) {
  h->linearization().set_pose_damping(lambda);
}
// LinearizationQR::get_stage2 (linearization_qr.hpp:716-815)
template <class S>
void h_stage2(Handle<S>* h, S lambda, const S* scaling, S* b_out, S* blocks_o) {
  using VecX = typename Handle<S>::VecX;
  auto& lqr = h->linearization();
  const int n = 9 * int(h->problem.cameras().size());
  VecX sc;
  if (scaling) {
    sc.resize(n);
    for (int i = 0; i < n; ++i) sc(i) = scaling[i];
  }
  VecX b;
  rootba::IndexedBlocks<S> blocks;
  lqr.get_stage2(lambda, scaling ? &sc : nullptr, blocks_o ? &blocks : nullptr, b);
  for (int i = 0; i < n; ++i) b_out[i] = b(i);
  if (blocks_o) blocks_out(blocks, n / 9, blocks_o);
}
template <class S>
void h_right_multiply(Handle<S>* h, const S* x, S* y) {
  using VecX = typename Handle<S>::VecX;
  const int n = 9 * int(h->problem.cameras().size());
  VecX xv(n);
  for (int i = 0; i < n; ++i) xv(i) = x[i];
  const VecX yv = h->linearization().right_multiply(xv);
  for (int i = 0; i < n; ++i) y[i] = yv(i);
}
template <class S>
S h_back_substitute(Handle<S>* h, const S* inc) {
  using VecX = typename Handle<S>::VecX;
  const int n = 9 * int(h->problem.cameras().size());
  VecX xv(n);
  for (int i = 0; i < n; ++i) xv(i) = inc[i];
  return h->linearization().back_substitute(xv);
}
template <class S>
void h_block_shape(Handle<S>* h, int l, int* rows, int* cols, int* lm_idx) {
  const auto* b = dynamic_cast<const rootba::LandmarkBlockDynamic<S, 9>*>(h->linearization().blocks()[l].get());
  *rows = int(b->get_num_rows());
  *cols = int(b->get_num_cols());
  *lm_idx = int(b->get_lm_idx());
}
template <class S>
void h_get_block(Handle<S>* h, int l, S* out) {
  const auto* b = dynamic_cast<const rootba::LandmarkBlockDynamic<S, 9>*>(h->linearization().blocks()[l].get());
  const auto& st = b->get_storage();
  for (Eigen::Index i = 0; i < st.rows(); ++i)
    for (Eigen::Index j = 0; j < st.cols(); ++j) out[i * st.cols() + j] = st(i, j);
}
template <class S>
void h_get_jl_col_scale(Handle<S>* h, S* out) {
  const auto& bl = h->linearization().blocks();
  for (size_t l = 0; l < bl.size(); ++l) {
    const auto* b = dynamic_cast<const rootba::LandmarkBlockDynamic<S, 9>*>(bl[l].get());
    const auto s = BlockOpen<S>::jl_col_scale(*b);
    for (int k = 0; k < 3; ++k) out[3 * l + k] = s(k);
  }
}
// Linearizor::{linearize, solve, apply} of the solver type chosen in the options
template <class S>
int h_linearize(Handle<S>* h) {
  h->it_summary = rootba::IterationSummary();
  h->linearizor().linearize();  // (numerical failure: the reference CHECK-fails, i.e. aborts)
  return 0;
}
template <class S>
void h_solve(Handle<S>* h, S lambda, S* inc_out, ref_cg_summary* cg) {
  const auto inc = h->linearizor().solve(lambda);
  for (Eigen::Index i = 0; i < inc.size(); ++i) inc_out[i] = inc(i);
  cg->num_iterations = h->it_summary.linear_solver_iterations;
  cg->termination_type = cg_termination_from_message(h->it_summary.linear_solver_message);
}
template <class S>
S h_apply(Handle<S>* h, const S* inc) {
  using VecX = typename Handle<S>::VecX;
  const int n = 9 * int(h->problem.cameras().size());
  VecX xv(n);
  for (int i = 0; i < n; ++i) xv(i) = inc[i];
  return h->linearizor().apply(std::move(xv));
}
// bundle_adjust_manual -> optimize_lm_ours (src/rootba/solver/bal_bundle_adjustment.cpp:249-578)
template <class S>
int h_optimize_lm(Handle<S>* h, ref_lm_iteration* log, int max_rows, int* termination) {
  rootba::SolverSummary summary;
  std::ostringstream sink;  // the LM loop reports every iteration on std::cout
  std::streambuf* old = std::cout.rdbuf(sink.rdbuf());
  rootba::bundle_adjust_manual(h->problem, h->sopt, &summary, nullptr);
  std::cout.rdbuf(old);
  const int n = int(summary.iterations.size());
  for (int i = 0; i < n && i < max_rows; ++i) {
    const auto& it = summary.iterations[i];
    ref_lm_iteration& r = log[i];
    std::memset(&r, 0, sizeof r);
    r.iteration = it.iteration;
    r.step_is_valid = it.step_is_valid;
    r.step_is_successful = it.step_is_successful;
    r.cg_iterations = it.linear_solver_iterations;
    r.cg_termination = cg_termination_from_message(it.linear_solver_message);
    r.cost = it.cost.all.error;
    r.cost_valid = it.cost.valid.error;
    r.lambda_ = 1.0 / it.trust_region_radius;  // the damping the NEXT solve will use
    r.relative_decrease = it.relative_decrease;
    r.iteration_time = it.iteration_time_in_seconds;
    r.stage1_time = it.stage1_time_in_seconds;
    r.stage2_time = it.stage2_time_in_seconds;
    r.precond_time = it.compute_preconditioner_time_in_seconds;
    r.pcg_time = it.solve_reduced_system_time_in_seconds;
    r.backsub_time = it.back_substitution_time_in_seconds;
    r.residual_time = it.residual_evaluation_time_in_seconds;
    r.num_obs = it.cost.all.num_obs;
    r.num_obs_valid = it.cost.valid.num_obs;
    r.residual_sum = it.cost.all.residual_sum;
    r.residual_sum_valid = it.cost.valid.residual_sum;
  }
  // this repository's numbering (include/rootba_hip.h): 0 NO_CONVERGENCE, 1 CONVERGED, -1 FAILURE
  *termination = summary.termination_type == rootba::CONVERGENCE      ? 1
                 : summary.termination_type == rootba::NO_CONVERGENCE ? 0
                                                                      : -1;
  return n;
}

// ---- loader ----------------------------------------------------------------------------------------
struct Loaded {
  BalProblem<double> problem;
  std::vector<int64_t> off;
  std::vector<int32_t> cam;
  std::vector<double> xy;
};

}  // namespace

#define REF_API(S, SUF)                                                                                            \
  void* ref_create_##SUF(int n_cams, int n_lms, const int64_t* off, const int32_t* cam, const S* xy,               \
                         const ref_options* o) {                                                                   \
    return h_create<S>(n_cams, n_lms, off, cam, xy, o);                                                            \
  }                                                                                                                \
  void ref_destroy_##SUF(void* h) { delete static_cast<Handle<S>*>(h); }                                           \
  void ref_set_state_##SUF(void* h, const S* c, const S* l) { h_set_state(static_cast<Handle<S>*>(h), c, l); }     \
  void ref_get_state_##SUF(void* h, S* c, S* l) { h_get_state(static_cast<Handle<S>*>(h), c, l); }                 \
  void ref_backup_##SUF(void* h) { static_cast<Handle<S>*>(h)->problem.backup(); }                                 \
  void ref_restore_##SUF(void* h) { static_cast<Handle<S>*>(h)->problem.restore(); }                               \
  void ref_compute_error_##SUF(void* h, ref_residual_info* out) {                                                  \
    h_compute_error(static_cast<Handle<S>*>(h), out);                                                              \
  }                                                                                                                \
  void ref_linearizor_compute_error_##SUF(void* h, ref_residual_info* out) {                                       \
    h_linearizor_compute_error(static_cast<Handle<S>*>(h), out);                                                   \
  }                                                                                                                \
  int ref_stage1_##SUF(void* h, S* d, S* blocks) { return h_stage1(static_cast<Handle<S>*>(h), d, blocks); }       \
  void ref_set_pose_damping_##SUF(void* h, S lam) { h_set_pose_damping(static_cast<Handle<S>*>(h), lam); }         \
  void ref_stage2_##SUF(void* h, S lam, const S* sc, S* b, S* blocks) {                                            \
    h_stage2(static_cast<Handle<S>*>(h), lam, sc, b, blocks);                                                      \
  }                                                                                                                \
  void ref_right_multiply_##SUF(void* h, const S* x, S* y) { h_right_multiply(static_cast<Handle<S>*>(h), x, y); } \
  S ref_back_substitute_##SUF(void* h, const S* inc) { return h_back_substitute(static_cast<Handle<S>*>(h), inc); } \
  void ref_block_shape_##SUF(void* h, int l, int* r, int* c, int* li) {                                            \
    h_block_shape(static_cast<Handle<S>*>(h), l, r, c, li);                                                        \
  }                                                                                                                \
  void ref_get_block_##SUF(void* h, int l, S* out) { h_get_block(static_cast<Handle<S>*>(h), l, out); }            \
  void ref_get_jl_col_scale_##SUF(void* h, S* out) { h_get_jl_col_scale(static_cast<Handle<S>*>(h), out); }        \
  int ref_linearize_##SUF(void* h) { return h_linearize(static_cast<Handle<S>*>(h)); }                             \
  void ref_solve_##SUF(void* h, S lam, S* inc, ref_cg_summary* cg) {                                               \
    h_solve(static_cast<Handle<S>*>(h), lam, inc, cg);                                                             \
  }                                                                                                                \
  S ref_apply_##SUF(void* h, const S* inc) { return h_apply(static_cast<Handle<S>*>(h), inc); }                    \
  int ref_optimize_lm_##SUF(void* h, ref_lm_iteration* log, int max_rows, int* term) {                             \
    return h_optimize_lm(static_cast<Handle<S>*>(h), log, max_rows, term);                                         \
  }                                                                                                                \
  int ref_linearize_point_##SUF(const S* obs, const S* p_w, const S* cam, int ignore_validity_check, S* res,       \
                                S* Jp, S* Ji, S* Jl) {                                                             \
    using H = rootba::BalBundleAdjustmentHelper<S>;                                                                \
    typename BalProblem<S>::Camera c;                                                                              \
    Eigen::Matrix<S, Eigen::Dynamic, 1> p(10);                                                                     \
    for (int k = 0; k < 10; ++k) p(k) = cam[k];                                                                    \
    c.from_params(p);                                                                                              \
    typename H::VecR r;                                                                                            \
    typename H::MatRP jp;                                                                                          \
    typename H::MatRI ji;                                                                                          \
    typename H::MatRL jl;                                                                                          \
    const bool valid = H::linearize_point(typename H::Vec2(obs[0], obs[1]), typename H::Vec3(p_w[0], p_w[1], p_w[2]), \
                                          c.T_c_w, c.intrinsics, ignore_validity_check != 0, r, &jp, &ji, &jl);    \
    for (int i = 0; i < 2; ++i) {                                                                                  \
      res[i] = r(i);                                                                                               \
      for (int j = 0; j < 6; ++j) Jp[6 * i + j] = jp(i, j);                                                        \
      for (int j = 0; j < 3; ++j) Ji[3 * i + j] = ji(i, j);                                                        \
      for (int j = 0; j < 3; ++j) Jl[3 * i + j] = jl(i, j);                                                        \
    }                                                                                                              \
    return valid ? 1 : 0;                                                                                          \
  }                                                                                                                \
  void ref_apply_inc_camera_##SUF(S* cam, const S* inc9) {                                                         \
    typename BalProblem<S>::Camera c;                                                                              \
    Eigen::Matrix<S, Eigen::Dynamic, 1> p(10);                                                                     \
    for (int k = 0; k < 10; ++k) p(k) = cam[k];                                                                    \
    c.from_params(p);                                                                                              \
    Eigen::Matrix<S, 6, 1> ip;                                                                                     \
    for (int k = 0; k < 6; ++k) ip(k) = inc9[k];                                                                   \
    c.apply_inc_pose(ip);                                                                                          \
    c.apply_inc_intrinsics(Eigen::Matrix<S, 3, 1>(inc9[6], inc9[7], inc9[8]));                                     \
    const auto q = c.params();                                                                                     \
    for (int k = 0; k < 10; ++k) cam[k] = q(k);                                                                    \
  }

extern "C" {
int ref_sizeof_lm_iteration() { return int(sizeof(ref_lm_iteration)); }
int ref_sizeof_options() { return int(sizeof(ref_options)); }

// the defaults of the reference's own SolverOptions declaration, in this repository's option struct
void ref_default_options(ref_options* o) {
  const SolverOptions s;
  std::memset(o, 0, sizeof *o);
  o->use_householder = s.use_householder_marginalization;
  o->use_valid_projections_only = s.use_projection_validity_check();
  o->robust_norm = int(s.residual.robust_norm);
  o->huber_parameter = s.residual.huber_parameter;
  o->jacobi_scaling_eps = s.jacobi_scaling_epsilon;
  o->preconditioner_type = s.preconditioner_type == SolverOptions::PreconditionerType::JACOBI         ? 0
                           : s.preconditioner_type == SolverOptions::PreconditionerType::SCHUR_JACOBI ? 1
                                                                                                      : 2;
  o->reduction_alg = s.reduction_alg;
  o->power_order = s.power_order;
  o->min_cg_it = s.min_linear_solver_iterations;
  o->max_cg_it = s.max_linear_solver_iterations;
  o->eta = s.eta;
  o->num_threads = s.num_threads;
  o->max_num_iterations = s.max_num_iterations;
  o->min_relative_decrease = s.min_relative_decrease;
  o->initial_trust_region_radius = s.initial_trust_region_radius;
  o->min_trust_region_radius = s.min_trust_region_radius;
  o->max_trust_region_radius = s.max_trust_region_radius;
  o->function_tolerance = s.function_tolerance;
  o->initial_vee = s.initial_vee;
  o->vee_factor = s.vee_factor;
  o->optimized_cost = int(s.optimized_cost);
  o->staged_execution = s.staged_execution;
  o->implicit_q = 0;
  o->solver_type = int(s.solver_type);
  o->explicit_after = 0;
}

REF_API(float, f32)
REF_API(double, f64)

// BalProblem::load_bal + normalize + perturb + filter_obs through load_normalized_bal_problem<double>
// (src/rootba/bal/bal_problem.cpp:773-852). seed < 0: no perturbation is requested by the tests then.
// input_type: 0 AUTO (by file name, autodetect_input_type), 2 BAL, 3 BUNDLER (BalDatasetOptions::DatasetType)
void* ref_load_bal(const char* path, int normalize, double normalization_scale, double rotation_sigma,
                   double translation_sigma, double point_sigma, int seed, double init_depth_threshold, int input_type) {
  rootba::BalDatasetOptions o;
  o.input = path;
  o.input_type = static_cast<rootba::BalDatasetOptions::DatasetType>(input_type);
  o.normalize = normalize != 0;
  o.normalization_scale = normalization_scale;
  o.rotation_sigma = rotation_sigma;
  o.translation_sigma = translation_sigma;
  o.point_sigma = point_sigma;
  o.random_seed = seed;
  o.init_depth_threshold = init_depth_threshold;
  o.quiet = true;
  auto* L = new Loaded();
  try {
    L->problem = rootba::load_normalized_bal_problem<double>(o);
  } catch (const std::exception& e) {
    delete L;
    return nullptr;
  }
  L->off.push_back(0);
  for (const auto& lm : L->problem.landmarks()) {
    for (const auto& [c, ob] : lm.obs) {
      L->cam.push_back(c);
      L->xy.push_back(ob.pos(0));
      L->xy.push_back(ob.pos(1));
    }
    L->off.push_back(int64_t(L->cam.size()));
  }
  return L;
}
void ref_loaded_sizes(void* p, int* n_cams, int* n_lms, int64_t* n_obs) {
  auto* L = static_cast<Loaded*>(p);
  *n_cams = L->problem.num_cameras();
  *n_lms = L->problem.num_landmarks();
  *n_obs = int64_t(L->cam.size());
}
void ref_loaded_get(void* p, double* cams, double* lms, int64_t* off, int32_t* cam, double* xy) {
  auto* L = static_cast<Loaded*>(p);
  for (int c = 0; c < L->problem.num_cameras(); ++c) {
    const auto q = L->problem.cameras()[c].params();
    for (int k = 0; k < 10; ++k) cams[10 * c + k] = q(k);
  }
  for (int l = 0; l < L->problem.num_landmarks(); ++l)
    for (int k = 0; k < 3; ++k) lms[3 * l + k] = L->problem.landmarks()[l].p_w(k);
  std::memcpy(off, L->off.data(), sizeof(int64_t) * L->off.size());
  std::memcpy(cam, L->cam.data(), sizeof(int32_t) * L->cam.size());
  std::memcpy(xy, L->xy.data(), sizeof(double) * L->xy.size());
}
// BalProblem::compute_rcs_sparsity / max_num_observations_per_lm (bal_problem.cpp:619-712)
void ref_loaded_stats(void* p, double* rcs_sparsity, int* max_obs_per_lm) {
  auto* L = static_cast<Loaded*>(p);
  *rcs_sparsity = L->problem.compute_rcs_sparsity();
  *max_obs_per_lm = L->problem.max_num_observations_per_lm();
}
void ref_loaded_destroy(void* p) { delete static_cast<Loaded*>(p); }
}  // extern "C"
