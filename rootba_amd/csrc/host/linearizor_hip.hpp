// linearizor_hip.hpp — C++17 host mirror of the reference's solver seam, bound
// to the C ABI of include/rootba_hip.h.
//
//   SolverOptions        reference src/rootba/bal/solver_options.hpp:46-284
//   ResidualInfo         reference src/rootba/bal/residual_info.hpp:57-96
//   LinearizorHIP<S>     same five calls as Linearizor<Scalar>
//                        (reference src/rootba/solver/linearizor.hpp:56-82), i.e.
//                        the class a maintainer registers in Linearizor::create
//                        (reference src/rootba/solver/linearizor.cpp:133-150)
//   bundle_adjust_manual reference src/rootba/solver/bal_bundle_adjustment.cpp:548-564
// Error behaviour follows the reference: conditions it CHECK-aborts on throw
// std::runtime_error here (the apps turn that into a non-zero exit).
#pragma once

#include <cmath>
#include <sys/resource.h>

#include <chrono>
#include <cstdio>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../../../include/rootba_hip.h"
#include "bal_problem.hpp"

namespace rootba_hip {

struct BalResidualOptions {
  enum class RobustNorm { NONE, HUBER };
  RobustNorm robust_norm = RobustNorm::NONE;
  double huber_parameter = 1.0;
};

struct SolverOptions {
  // reference src/rootba/bal/solver_options.hpp:58-62, 89-90: SQUARE_ROOT (default), SCHUR_COMPLEMENT; POWER_SCHUR_COMPLEMENT
  // (LinearizorPowerSC) is not built here
  enum class SolverType { SQUARE_ROOT, SCHUR_COMPLEMENT };
  SolverType solver_type = SolverType::SQUARE_ROOT;
  enum class PreconditionerType { JACOBI, SCHUR_JACOBI, POWER_SCHUR_COMPLEMENT };
  enum class OptimizedCost { ERROR, ERROR_VALID, ERROR_VALID_AVG };
  int verbosity_level = 2;
  BalResidualOptions residual;
  OptimizedCost optimized_cost = OptimizedCost::ERROR;
  int max_num_iterations = 20;
  double min_relative_decrease = 0.0;
  double initial_trust_region_radius = 1e4;
  double min_trust_region_radius = 1e-32;
  double max_trust_region_radius = 1e16;
  int min_linear_solver_iterations = 0;
  int max_linear_solver_iterations = 500;
  double eta = 1e-1;
  bool jacobi_scaling = true;
  double jacobi_scaling_epsilon = 0.0;
  PreconditionerType preconditioner_type = PreconditionerType::SCHUR_JACOBI;
  double function_tolerance = 1e-6;
  bool use_double = true;
  bool use_householder_marginalization = true;
  bool staged_execution = true;
  bool mixed_precision = false;  // RBA_MIXED (needs use_double: the host problem is double, the device algebra float)
  int reduction_alg = 1;
  int power_order = 10;
  double initial_vee = 2.0;
  double vee_factor = 2.0;
  bool implicit_q = true;   // not in the reference; accepted, ignored: products are always evaluated from the QR factors
  int explicit_after = -1;  // not in the reference: rba_options.explicit_after (-1 = measured break-even)
  bool use_projection_validity_check() const { return optimized_cost != OptimizedCost::ERROR; }

  rba_options to_rba() const {
    rba_options o;
    rba_default_options(&o);
    o.use_householder = use_householder_marginalization;
    o.use_valid_projections_only = use_projection_validity_check();
    o.robust_norm = residual.robust_norm == BalResidualOptions::RobustNorm::HUBER;
    o.huber_parameter = residual.huber_parameter;
    o.jacobi_scaling_eps = jacobi_scaling_epsilon;
    o.preconditioner_type = int(preconditioner_type);
    o.reduction_alg = reduction_alg;
    o.power_order = power_order;
    o.min_cg_it = min_linear_solver_iterations;
    o.max_cg_it = max_linear_solver_iterations;
    o.eta = eta;
    o.max_num_iterations = max_num_iterations;
    o.min_relative_decrease = min_relative_decrease;
    o.initial_trust_region_radius = initial_trust_region_radius;
    o.min_trust_region_radius = min_trust_region_radius;
    o.max_trust_region_radius = max_trust_region_radius;
    o.function_tolerance = function_tolerance;
    o.initial_vee = initial_vee;
    o.vee_factor = vee_factor;
    o.optimized_cost = int(optimized_cost);
    o.staged_execution = staged_execution;
    o.implicit_q = implicit_q;
    o.explicit_after = explicit_after;
    o.solver_type = solver_type == SolverType::SCHUR_COMPLEMENT ? 1 : 0;
    return o;
  }
};

struct ResidualItem {
  int num_obs = 0;
  double error = 0, residual_sum = 0;
  double residual_mean() const { return num_obs > 0 ? residual_sum / num_obs : 0.0; }
};
struct ResidualInfo {
  ResidualItem all, valid;
  bool is_numerically_valid = true;
};

struct IterationSummary {  // subset of reference solver_summary.hpp:99-204
  int iteration = 0;
  bool step_is_valid = false, step_is_successful = false;
  ResidualInfo cost;
  ResidualInfo prev_cost;  // cost of the previous summary (the reference stores cost_change = cost - that)
  double relative_decrease = 0, trust_region_radius = 0, step_norm = 0;
  int linear_solver_iterations = 0;
  double iteration_time_in_seconds = 0, cumulative_time_in_seconds = 0, stage1_time_in_seconds = 0,
         stage2_time_in_seconds = 0, compute_preconditioner_time_in_seconds = 0,
         solve_reduced_system_time_in_seconds = 0, back_substitution_time_in_seconds = 0,
         residual_evaluation_time_in_seconds = 0;
  // unstaged execution only (staged_execution = false; rba_substage_timings, include/rootba_hip.h)
  double jacobian_evaluation_time_in_seconds = 0, scale_landmark_jacobian_time_in_seconds = 0,
         perform_qr_time_in_seconds = 0, scale_pose_jacobian_time_in_seconds = 0,
         landmark_damping_time_in_seconds = 0, compute_gradient_time_in_seconds = 0;
  uint64_t resident_memory_peak = 0;
};
struct SolverSummary {
  std::vector<IterationSummary> iterations;
  std::string solver_type = "bal_qr_hip";  // bal_sc_hip for the SC backend (reference: bal_qr / bal_sc / bal_power_sc)
  std::string message;
  int termination_type = 0;  // 0 NO_CONVERGENCE, 1 CONVERGENCE
  double initial_cost = 0, final_cost = 0;
  double preprocessor_time_in_seconds = 0, minimizer_time_in_seconds = 0, postprocessor_time_in_seconds = 0,
         total_time_in_seconds = 0;
};

inline void check_rba(int status, const char* what) {
  if (status < 0) throw std::runtime_error(std::string(what) + ": " + rba_last_error());
}

template <class Scalar>
class LinearizorHIP {
 public:
  using VecX = std::vector<Scalar>;

  static std::unique_ptr<LinearizorHIP> create(BalProblem<Scalar>& bal_problem, const SolverOptions& options,
                                               SolverSummary* summary = nullptr, int device = 0, int n_gpus = 1) {
    return std::unique_ptr<LinearizorHIP>(new LinearizorHIP(bal_problem, options, summary, device, n_gpus));
  }
  ~LinearizorHIP() { rba_destroy(h_); }

  void start_iteration(IterationSummary* it_summary = nullptr) { it_summary_ = it_summary; }
  void finish_iteration() { it_summary_ = nullptr; }

  void compute_error(ResidualInfo& ri) {
    rba_residual_info r;
    check_rba(rba_compute_error(h_, &r), "rba_compute_error");
    ri.all = {r.all_num_obs, r.all_error, r.all_residual_sum};
    ri.valid = {r.valid_num_obs, r.valid_error, r.valid_residual_sum};
    ri.is_numerically_valid = r.is_numerically_valid != 0;
  }
  void linearize() {
    const int st = rba_linearize(h_, nullptr);
    check_rba(st, "rba_linearize");
    if (st == RBA_NUMERICAL_FAILURE)
      throw std::runtime_error("did not expect numerical failure during linearization");
  }
  VecX solve(Scalar lambda) {
    VecX inc(size_t(9) * bal_problem_.num_cameras());
    rba_cg_summary cg;
    if (mixed_) {  // camera-sized vectors cross the RBA_MIXED boundary as float
      std::vector<float> f(inc.size());
      check_rba(rba_solve(h_, double(lambda), f.data(), &cg), "rba_solve");
      for (size_t i = 0; i < f.size(); ++i) inc[i] = Scalar(f[i]);
    } else
    check_rba(rba_solve(h_, double(lambda), inc.data(), &cg), "rba_solve");
    if (it_summary_) it_summary_->linear_solver_iterations = cg.num_iterations;
    return inc;
  }
  Scalar apply(VecX&& inc) {
    double l_diff = 0;
    std::vector<float> f;
    if (mixed_) {
      f.resize(inc.size());
      for (size_t i = 0; i < f.size(); ++i) f[i] = float(inc[i]);
    }
    const int st = mixed_ ? rba_apply(h_, f.data(), &l_diff) : rba_apply(h_, inc.data(), &l_diff);
    check_rba(st, "rba_apply");
    if (st == RBA_NUMERICAL_FAILURE) return std::numeric_limits<Scalar>::quiet_NaN();
    return Scalar(l_diff);
  }
  void backup() { check_rba(rba_backup(h_), "rba_backup"); }
  void restore() { check_rba(rba_restore(h_), "rba_restore"); }
  // device state -> BalProblem (the reference mutates BalProblem in place)
  void download() {
    VecX cams(size_t(10) * bal_problem_.num_cameras()), lms(size_t(3) * bal_problem_.num_landmarks());
    check_rba(rba_get_state(h_, cams.data(), lms.data()), "rba_get_state");
    bal_problem_.copy_from_state(cams, lms);
  }
  rba_handle handle() { return h_; }

 private:
  LinearizorHIP(BalProblem<Scalar>& bal_problem, const SolverOptions& options, SolverSummary* summary, int device, int n_gpus)
      : bal_problem_(bal_problem), options_(options), summary_(summary),
        mixed_(options.mixed_precision && std::is_same<Scalar, double>::value) {
    if (options.mixed_precision && !mixed_)
      throw std::runtime_error("mixed_precision needs use_double (the host problem is double)");
    const rba_options o = options.to_rba();
    // BalProblem already stores the CSR topology the C ABI takes
    const int dt = mixed_ ? RBA_MIXED : std::is_same<Scalar, float>::value ? RBA_F32 : RBA_F64;
    if (n_gpus > 1) {
      // ONE handle over devices device .. device + n_gpus - 1: the library shards the landmarks itself (rba_create_sharded)
      std::vector<int> ids(static_cast<size_t>(n_gpus));
      for (int i = 0; i < n_gpus; ++i) ids[static_cast<size_t>(i)] = device + i;
      check_rba(rba_create_sharded(dt, n_gpus, ids.data(), bal_problem.num_cameras(), bal_problem.num_landmarks(),
                                   bal_problem.lm_off.data(), bal_problem.obs_cam.data(), bal_problem.obs_xy.data(), &o, &h_),
                "rba_create_sharded");
    } else {
      check_rba(rba_create(dt, device, bal_problem.num_cameras(), bal_problem.num_landmarks(), bal_problem.lm_off.data(),
                           bal_problem.obs_cam.data(), bal_problem.obs_xy.data(), &o, &h_),
                "rba_create");
    }
    VecX cams, lms;
    bal_problem.copy_to_state(cams, lms);
    check_rba(rba_set_state(h_, cams.data(), lms.data()), "rba_set_state");
  }
  BalProblem<Scalar>& bal_problem_;
  SolverOptions options_;
  SolverSummary* summary_;
  IterationSummary* it_summary_ = nullptr;
  bool mixed_ = false;
  rba_handle h_ = nullptr;
};

// bundle_adjust_manual / optimize_lm_ours: the loop runs inside the library
// (rba_lm_step, one call per iteration) so nothing but the log crosses the bus;
// console lines follow the reference's format (bal_bundle_adjustment.cpp:303-460).
template <class Scalar>
void bundle_adjust_manual(BalProblem<Scalar>& bal_problem, const SolverOptions& options, SolverSummary* summary_out,
                          int device = 0, int n_gpus = 1) {
  SolverSummary local;
  SolverSummary& summary = summary_out ? *summary_out : local;
  summary = SolverSummary();
  if (options.solver_type == SolverOptions::SolverType::SCHUR_COMPLEMENT) summary.solver_type = "bal_sc_hip";
  const auto t_total = std::chrono::steady_clock::now();
  auto seconds_since = [](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  // preprocessor = building the linearizor: device allocation, topology and state upload
  auto lin = LinearizorHIP<Scalar>::create(bal_problem, options, &summary, device, n_gpus);
  summary.preprocessor_time_in_seconds = seconds_since(t_total);
  const auto t_minimizer = std::chrono::steady_clock::now();
  check_rba(rba_lm_begin(lin->handle()), "rba_lm_begin");
  for (;;) {
    rba_lm_iteration row;
    int more = 0;
    check_rba(rba_lm_step(lin->handle(), &row, &more), "rba_lm_step");
    int term = 0;
    rba_lm_termination(lin->handle(), &term);
    if (term == -1 && !more) throw std::runtime_error("did not expect numerical failure during linearization");
    if (row.iteration > options.max_num_iterations) break;
    IterationSummary it;
    it.iteration = row.iteration;
    it.step_is_valid = row.step_is_valid;
    it.step_is_successful = row.step_is_successful;
    it.cost.all = {row.num_obs, row.cost, row.residual_sum};
    it.cost.valid = {row.num_obs_valid, row.cost_valid, row.residual_sum_valid};
    if (!summary.iterations.empty()) it.prev_cost = summary.iterations.back().cost;
    it.step_norm = row.inc_norm;
    it.cumulative_time_in_seconds = seconds_since(t_minimizer);
    {
      struct rusage ru;
      if (getrusage(RUSAGE_SELF, &ru) == 0) it.resident_memory_peak = uint64_t(ru.ru_maxrss) * 1024u;
    }
    it.relative_decrease = row.relative_decrease;
    it.trust_region_radius = row.lambda > 0 ? 1.0 / row.lambda : 0.0;
    it.linear_solver_iterations = row.cg_iterations;
    it.iteration_time_in_seconds = row.iteration_time;
    it.stage1_time_in_seconds = row.stage1_time;
    it.stage2_time_in_seconds = row.stage2_time;
    it.compute_preconditioner_time_in_seconds = row.precond_time;
    it.solve_reduced_system_time_in_seconds = row.pcg_time;
    it.back_substitution_time_in_seconds = row.backsub_time;
    it.residual_evaluation_time_in_seconds = row.residual_time;
    if (!options.staged_execution) {
      // the reference's unstaged timers (linearizor_qr.cpp:94-112, 166-187); the fused camera-major pass of
      // stage 2 (preconditioner blocks + gradient) is filed under compute_preconditioner_time like the
      // reference's SCHUR_JACOBI branch, compute_gradient_time stays 0
      rba_substage_timings sub{};
      check_rba(rba_get_substage_timings(lin->handle(), &sub), "rba_get_substage_timings");
      it.jacobian_evaluation_time_in_seconds = sub.jacobian_evaluation_time;
      it.scale_landmark_jacobian_time_in_seconds = sub.scale_landmark_jacobian_time;
      it.perform_qr_time_in_seconds = sub.perform_qr_time;
      it.scale_pose_jacobian_time_in_seconds = sub.scale_pose_jacobian_time;
      it.landmark_damping_time_in_seconds = sub.landmark_damping_time;
      it.compute_preconditioner_time_in_seconds +=
          sub.stage1_preconditioner_time + sub.stage2_preconditioner_and_gradient_time;
    }
    summary.iterations.push_back(it);
    if (options.verbosity_level >= 1) {
      if (row.iteration == 0) {
        std::printf("Iteration 0, error: %.4e\n", row.cost);
      } else {
        std::printf("Iteration %d\n\t[INFO] Stage 1 time %.3fs.\n\t[INFO] Stage 2 time %.3fs.\n", row.iteration,
                    row.stage1_time, row.stage2_time);
        std::printf("\t[CG] Summary: %d iterations. Time %.3fs. Time per iteration %.4fs\n", row.cg_iterations,
                    row.pcg_time, row.cg_iterations ? row.pcg_time / row.cg_iterations : 0.0);
        std::printf("\t[%s] error: %.4e, lambda: %.1e, cg_iter: %d, it_time: %.3fs\n",
                    row.step_is_successful ? "Success" : (row.step_is_valid ? "Reject" : "Invalid"), row.cost,
                    row.lambda, row.cg_iterations, row.iteration_time);
      }
    }
    if (!more) break;
  }
  int term = 0;
  rba_lm_termination(lin->handle(), &term);
  summary.termination_type = term == 1 ? 1 : 0;
  summary.message = term == 1 ? "Function tolerance reached."
                              : "Solver did not converge after maximum number of iterations";
  if (!summary.iterations.empty()) {
    summary.initial_cost = summary.iterations.front().cost.all.error;
    for (auto it = summary.iterations.rbegin(); it != summary.iterations.rend(); ++it)
      if (it->step_is_successful) {
        summary.final_cost = it->cost.all.error;
        break;
      }
  }
  summary.minimizer_time_in_seconds = seconds_since(t_minimizer);
  const auto t_post = std::chrono::steady_clock::now();
  lin->download();
  summary.postprocessor_time_in_seconds = seconds_since(t_post);
  summary.total_time_in_seconds = seconds_since(t_total);
  if (options.verbosity_level >= 1)
    std::printf("Final Cost: %.4e\n%s: %s\n", summary.final_cost,
                summary.termination_type ? "CONVERGENCE" : "NO_CONVERGENCE", summary.message.c_str());
}

}  // namespace rootba_hip
