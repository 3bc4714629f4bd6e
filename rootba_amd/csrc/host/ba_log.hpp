// ba_log.hpp — `ba_log.json` in the reference's layout, so that its evaluation
// tooling (python/rootba/log.py, plot_logs.py, generate_tables.py) reads the logs of
// this solver unchanged.
//
// Layout (reference src/rootba/bal/ba_log.cpp:62-149): ONE flat JSON object with an
// array per `BaIteration` member (ba_log.hpp:139-237, same names, one entry per LM
// iteration), plus "_type": "rootba" and "_static": {problem_info, timing, solver}
// (ba_log.hpp:45-137). How the per-iteration values derive from the iteration
// summaries follows ba_log_utils.cpp:97-160: a rejected step repeats the previous
// row's cost columns so that plots stay monotonic. Own writer (no nlohmann/json).
#pragma once

#include <sys/resource.h>

#include <cmath>
#include <cstdint>
#include <fstream>
#include <functional>
#include <iomanip>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "bal_problem.hpp"
#include "linearizor_hip.hpp"

namespace rootba_hip {

struct PipelineTimingSummary {  // reference bal_pipeline_summary.hpp
  double load_time = 0, preprocess_time = 0, optimize_time = 0, postprocess_time = 0;
};

struct DatasetStats {
  double mean = 0, min = 0, max = 0, stddev = 0;
};
struct DatasetSummary {  // reference bal_dataset_summary / BaLog::ProblemInfo
  std::string type = "bal", input_path;
  int num_cameras = 0, num_landmarks = 0;
  int64_t num_observations = 0;
  double rcs_sparsity = 0;
  DatasetStats per_lm_obs, per_host_lms;
};

template <class Scalar>
DatasetSummary summarize_dataset(const BalProblem<Scalar>& p, const std::string& input_path) {
  DatasetSummary s;
  s.input_path = input_path;
  s.num_cameras = p.num_cameras();
  s.num_landmarks = p.num_landmarks();
  s.num_observations = p.num_observations();
  auto stats = [](const std::vector<int64_t>& counts) {
    DatasetStats st;
    if (counts.empty()) return st;
    double sum = 0, sq = 0;
    st.min = st.max = double(counts[0]);
    for (int64_t c : counts) {
      sum += double(c);
      st.min = std::min(st.min, double(c));
      st.max = std::max(st.max, double(c));
    }
    st.mean = sum / double(counts.size());
    for (int64_t c : counts) sq += (double(c) - st.mean) * (double(c) - st.mean);
    st.stddev = std::sqrt(sq / double(counts.size()));
    return st;
  };
  std::vector<int64_t> per_lm(p.num_landmarks()), per_cam(p.num_cameras(), 0);
  for (int l = 0; l < p.num_landmarks(); ++l) per_lm[l] = p.lm_off[l + 1] - p.lm_off[l];
  for (int32_t c : p.obs_cam) ++per_cam[c];
  s.rcs_sparsity = p.num_cameras() <= 20000 ? p.compute_rcs_sparsity() : 0.0;  // n_c^2 byte mask
  s.per_lm_obs = stats(per_lm);
  s.per_host_lms = stats(per_cam);  // landmarks observed per camera
  return s;
}

inline uint64_t resident_memory_peak_bytes() {
  struct rusage ru;
  return getrusage(RUSAGE_SELF, &ru) == 0 ? uint64_t(ru.ru_maxrss) * 1024u : 0u;
}

namespace detail {
inline std::string json_escape(const std::string& s) {
  std::string o;
  for (char c : s) {
    if (c == '"' || c == '\\') {
      o += '\\';
      o += c;
    } else if (c == '\n') {
      o += "\\n";
    } else if (static_cast<unsigned char>(c) < 0x20) {
      o += ' ';
    } else {
      o += c;
    }
  }
  return o;
}
inline std::string json_num(double v) {
  if (!std::isfinite(v)) return "null";  // what nlohmann::json emits for NaN / inf
  std::ostringstream ss;
  ss << std::setprecision(17) << v;
  return ss.str();
}
}  // namespace detail

// one BaIteration row (ba_log.hpp:139-237)
struct BaIteration {
  int iteration = -1;
  std::string linear_solver_type;
  bool step_is_valid = false, step_is_nonmonotonic = false, step_is_successful = false;
  int num_obs = 0, num_obs_valid = 0, num_obs_valid_change = 0;
  double cost = 0, cost_change = 0, cost_valid = 0, cost_valid_change = 0, cost_avg_valid = 0,
         cost_avg_valid_change = 0;
  double grad_projected_norm = 0, grad_projected_max_norm = 0, grad_norm = 0, grad_max_norm = 0;
  double residual_block_mean = 0, residual_block_valid_mean = 0, step_norm = 0, relative_decrease = 0,
         trust_region_radius = 0;
  int linear_solver_iterations = 0;
  double iteration_time = 0, cumulative_time = 0, logging_time = 0, step_solver_time = 0;
  double residual_evaluation_time = 0, jacobian_evaluation_time = 0, scale_landmark_jacobian_time = 0,
         perform_qr_time = 0, stage1_time = 0, scale_pose_jacobian_time = 0, landmark_damping_time = 0,
         compute_preconditioner_time = 0, compute_gradient_time = 0, stage2_time = 0, prepare_time = 0,
         solve_reduced_system_time = 0, back_substitution_time = 0, update_cameras_time = 0;
  uint64_t resident_memory = 0, resident_memory_peak = 0;
};

inline std::vector<BaIteration> to_ba_iterations(const SolverSummary& summary) {
  std::vector<BaIteration> rows;
  rows.reserve(summary.iterations.size());
  for (const IterationSummary& s : summary.iterations) {
    BaIteration r;
    const BaIteration* prev = rows.empty() ? nullptr : &rows.back();
    r.iteration = s.iteration;
    r.linear_solver_type = summary.solver_type;
    r.step_is_valid = s.step_is_valid;
    r.step_is_successful = s.step_is_successful;
    if (s.step_is_successful || !prev) {
      r.num_obs = s.cost.all.num_obs;
      r.num_obs_valid = s.cost.valid.num_obs;
      r.cost = s.cost.all.error;
      r.cost_valid = s.cost.valid.error;
      r.cost_avg_valid = s.cost.valid.num_obs > 0 ? s.cost.valid.error / s.cost.valid.num_obs : 0.0;
      r.residual_block_mean = s.cost.all.residual_mean();
      r.residual_block_valid_mean = s.cost.valid.residual_mean();
      if (s.iteration > 0) {  // change w.r.t. the previous summary (bal_bundle_adjustment.cpp:69-73)
        r.num_obs_valid_change = s.cost.valid.num_obs - s.prev_cost.valid.num_obs;
        r.cost_change = s.cost.all.error - s.prev_cost.all.error;
        r.cost_valid_change = s.cost.valid.error - s.prev_cost.valid.error;
        const double prev_avg = s.prev_cost.valid.num_obs > 0 ? s.prev_cost.valid.error / s.prev_cost.valid.num_obs : 0.0;
        r.cost_avg_valid_change = r.cost_avg_valid - prev_avg;
      }
      r.step_norm = s.step_norm;
      r.relative_decrease = s.relative_decrease;
    } else {
      r.num_obs = prev->num_obs;
      r.num_obs_valid = prev->num_obs_valid;
      r.cost = prev->cost;
      r.cost_valid = prev->cost_valid;
      r.cost_avg_valid = prev->cost_avg_valid;
      r.residual_block_mean = prev->residual_block_mean;
      r.residual_block_valid_mean = prev->residual_block_valid_mean;
    }
    r.trust_region_radius = s.trust_region_radius;
    r.linear_solver_iterations = s.linear_solver_iterations;
    r.iteration_time = s.iteration_time_in_seconds;
    r.cumulative_time = s.cumulative_time_in_seconds;
    // step_solver_time like the reference's finish_iteration (bal_bundle_adjustment.cpp:57-67);
    // Jacobian scaling and QR are fused into stage 1 here, i.e. its "staged" accounting
    r.step_solver_time = s.stage2_time_in_seconds + s.solve_reduced_system_time_in_seconds + s.back_substitution_time_in_seconds;
    r.residual_evaluation_time = s.residual_evaluation_time_in_seconds;
    r.stage1_time = s.stage1_time_in_seconds;
    r.jacobian_evaluation_time = s.jacobian_evaluation_time_in_seconds;
    r.scale_landmark_jacobian_time = s.scale_landmark_jacobian_time_in_seconds;
    r.perform_qr_time = s.perform_qr_time_in_seconds;
    r.scale_pose_jacobian_time = s.scale_pose_jacobian_time_in_seconds;
    r.landmark_damping_time = s.landmark_damping_time_in_seconds;
    r.compute_gradient_time = s.compute_gradient_time_in_seconds;
    r.compute_preconditioner_time = s.compute_preconditioner_time_in_seconds;
    r.stage2_time = s.stage2_time_in_seconds;
    r.solve_reduced_system_time = s.solve_reduced_system_time_in_seconds;
    r.back_substitution_time = s.back_substitution_time_in_seconds;
    r.resident_memory_peak = s.resident_memory_peak;
    r.resident_memory = s.resident_memory_peak;
    rows.push_back(r);
  }
  return rows;
}

inline bool save_ba_log_json(const std::string& path, const SolverSummary& summary, const DatasetSummary& dataset,
                             const PipelineTimingSummary& timing) {
  const std::vector<BaIteration> rows = to_ba_iterations(summary);
  std::ofstream f(path);
  if (!f.is_open()) return false;
  using detail::json_num;
  bool first = true;
  auto column = [&](const char* name, const std::function<std::string(const BaIteration&)>& get) {
    f << (first ? "" : ",\n") << "    \"" << name << "\": [";
    first = false;
    for (size_t i = 0; i < rows.size(); ++i) f << (i ? ", " : "") << get(rows[i]);
    f << "]";
  };
#define RBA_LOG_NUM(member) column(#member, [](const BaIteration& r) { return json_num(double(r.member)); })
#define RBA_LOG_INT(member) column(#member, [](const BaIteration& r) { return std::to_string(r.member); })
#define RBA_LOG_BOOL(member) column(#member, [](const BaIteration& r) { return std::string(r.member ? "true" : "false"); })
  f << "{\n";
  // nlohmann::json orders keys alphabetically; readers do not depend on the order
  RBA_LOG_INT(iteration);
  column("linear_solver_type", [](const BaIteration& r) { return "\"" + detail::json_escape(r.linear_solver_type) + "\""; });
  RBA_LOG_BOOL(step_is_valid);
  RBA_LOG_BOOL(step_is_nonmonotonic);
  RBA_LOG_BOOL(step_is_successful);
  RBA_LOG_INT(num_obs);
  RBA_LOG_INT(num_obs_valid);
  RBA_LOG_INT(num_obs_valid_change);
  RBA_LOG_NUM(cost);
  RBA_LOG_NUM(cost_change);
  RBA_LOG_NUM(cost_valid);
  RBA_LOG_NUM(cost_valid_change);
  RBA_LOG_NUM(cost_avg_valid);
  RBA_LOG_NUM(cost_avg_valid_change);
  RBA_LOG_NUM(grad_projected_norm);
  RBA_LOG_NUM(grad_projected_max_norm);
  RBA_LOG_NUM(grad_norm);
  RBA_LOG_NUM(grad_max_norm);
  RBA_LOG_NUM(residual_block_mean);
  RBA_LOG_NUM(residual_block_valid_mean);
  RBA_LOG_NUM(step_norm);
  RBA_LOG_NUM(relative_decrease);
  RBA_LOG_NUM(trust_region_radius);
  RBA_LOG_INT(linear_solver_iterations);
  RBA_LOG_NUM(iteration_time);
  RBA_LOG_NUM(cumulative_time);
  RBA_LOG_NUM(logging_time);
  RBA_LOG_NUM(step_solver_time);
  RBA_LOG_NUM(residual_evaluation_time);
  RBA_LOG_NUM(jacobian_evaluation_time);
  RBA_LOG_NUM(scale_landmark_jacobian_time);
  RBA_LOG_NUM(perform_qr_time);
  RBA_LOG_NUM(stage1_time);
  RBA_LOG_NUM(scale_pose_jacobian_time);
  RBA_LOG_NUM(landmark_damping_time);
  RBA_LOG_NUM(compute_preconditioner_time);
  RBA_LOG_NUM(compute_gradient_time);
  RBA_LOG_NUM(stage2_time);
  RBA_LOG_NUM(prepare_time);
  RBA_LOG_NUM(solve_reduced_system_time);
  RBA_LOG_NUM(back_substitution_time);
  RBA_LOG_NUM(update_cameras_time);
  RBA_LOG_INT(resident_memory);
  RBA_LOG_INT(resident_memory_peak);
#undef RBA_LOG_NUM
#undef RBA_LOG_INT
#undef RBA_LOG_BOOL
  f << (first ? "" : ",\n") << "    \"_type\": \"rootba\",\n";

  auto stats = [&](const DatasetStats& s) {
    return "{\"mean\": " + json_num(s.mean) + ", \"min\": " + json_num(s.min) + ", \"max\": " + json_num(s.max) +
           ", \"stddev\": " + json_num(s.stddev) + "}";
  };
  double linear_solver_time = 0, residual_time = 0;
  int successful = -1, unsuccessful = 0;  // iteration 0 counts as successful in the rows, not in the total
  for (const BaIteration& r : rows) {
    linear_solver_time += r.step_solver_time;
    residual_time += r.residual_evaluation_time;
    (r.step_is_successful ? successful : unsuccessful) += 1;
  }
  const double total = timing.load_time + timing.preprocess_time + timing.optimize_time;
  const int n_solves = rows.empty() ? 0 : int(rows.size()) - 1;
  f << "    \"_static\": {\n"
    << "        \"problem_info\": {\"type\": \"" << detail::json_escape(dataset.type) << "\", \"input_path\": \""
    << detail::json_escape(dataset.input_path) << "\", \"num_cameras\": " << dataset.num_cameras
    << ", \"num_landmarks\": " << dataset.num_landmarks << ", \"num_observations\": " << dataset.num_observations
    << ", \"rcs_sparsity\": " << json_num(dataset.rcs_sparsity) << ", \"per_lm_obs\": " << stats(dataset.per_lm_obs)
    << ", \"per_host_lms\": " << stats(dataset.per_host_lms) << "},\n"
    << "        \"timing\": {\"total\": " << json_num(total) << ", \"load\": " << json_num(timing.load_time)
    << ", \"preprocess\": " << json_num(timing.preprocess_time) << ", \"optimize\": " << json_num(timing.optimize_time)
    << ", \"postprocess\": " << json_num(timing.postprocess_time) << "},\n"
    << "        \"solver\": {\"solver_type\": \"" << detail::json_escape(summary.solver_type)
    << "\", \"termination_type\": \"" << (summary.termination_type == 1 ? "CONVERGENCE" : "NO_CONVERGENCE")
    << "\", \"message\": \"" << detail::json_escape(summary.message) << "\", \"num_successful_steps\": "
    << std::max(successful, 0) << ", \"num_unsuccessful_steps\": " << unsuccessful
    << ", \"logging_time_in_seconds\": 0.0, \"preprocessor_time_in_seconds\": "
    << json_num(summary.preprocessor_time_in_seconds) << ", \"minimizer_time_in_seconds\": "
    << json_num(summary.minimizer_time_in_seconds) << ", \"postprocessor_time_in_seconds\": "
    << json_num(summary.postprocessor_time_in_seconds) << ", \"total_time_in_seconds\": "
    << json_num(summary.total_time_in_seconds) << ", \"linear_solver_time_in_seconds\": " << json_num(linear_solver_time)
    << ", \"num_linear_solves\": " << n_solves << ", \"residual_evaluation_time_in_seconds\": " << json_num(residual_time)
    << ", \"num_residual_evaluations\": " << int(rows.size()) << ", \"jacobian_evaluation_time_in_seconds\": 0.0"
    << ", \"num_jacobian_evaluations\": " << std::max(successful, 0) + (rows.empty() ? 0 : 1)
    << ", \"num_threads_given\": 0, \"num_threads_used\": 1, \"num_threads_available\": "
    << std::thread::hardware_concurrency() << ", \"resident_memory_peak\": " << resident_memory_peak_bytes()
    << ", \"initial_cost\": " << json_num(summary.initial_cost) << ", \"final_cost\": " << json_num(summary.final_cost)
    << "}\n    }\n}\n";
  return bool(f);
}

}  // namespace rootba_hip
