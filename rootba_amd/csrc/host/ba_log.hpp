// ba_log.hpp — `ba_log.json` in the reference's layout, so that its evaluation
// tooling (python/rootba/log.py, plot_logs.py, generate_tables.py) reads the logs of
// this solver unchanged.
//
// Layout (reference src/rootba/bal/ba_log.cpp:62-149): ONE flat JSON object with an
// array per `BaIteration` member (ba_log.hpp:139-237, same names, one entry per LM
// iteration), plus "_type": "rootba" and "_static": {problem_info, timing, solver}
// (ba_log.hpp:45-137). How the per-iteration values derive from the iteration
// summaries follows ba_log_utils.cpp:97-160: a rejected step repeats the previous
// row's cost columns so that plots stay monotonic. Own writers (no nlohmann/json): pretty-printed JSON and
// the UBJSON twin `<log>.ubjson` (SaveLogFlag UBJSON), byte layout of nlohmann::json::to_ubjson.
#pragma once

#include <sys/resource.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
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

namespace detail {
// Minimal JSON value (what nlohmann::json holds for the log): objects keep their keys sorted like
// nlohmann's std::map-based object, so both writers emit the reference's key order.
struct JValue {
  enum Kind { Null, Bool, Int, UInt, Float, String, Array, Object } kind = Null;
  bool b = false;
  int64_t i = 0;
  uint64_t u = 0;
  double d = 0;
  std::string s;
  std::vector<JValue> arr;
  std::vector<std::pair<std::string, JValue>> obj;  // kept sorted by key
  static JValue boolean(bool v) { JValue j; j.kind = Bool; j.b = v; return j; }
  static JValue integer(int64_t v) { JValue j; j.kind = Int; j.i = v; return j; }
  static JValue unsigned_integer(uint64_t v) { JValue j; j.kind = UInt; j.u = v; return j; }
  static JValue number(double v) { JValue j; j.kind = Float; j.d = v; return j; }
  static JValue string(const std::string& v) { JValue j; j.kind = String; j.s = v; return j; }
  static JValue array() { JValue j; j.kind = Array; return j; }
  static JValue object() { JValue j; j.kind = Object; return j; }
  JValue& operator[](const std::string& key) {
    kind = Object;
    auto it = std::lower_bound(obj.begin(), obj.end(), key,
                               [](const std::pair<std::string, JValue>& e, const std::string& k) { return e.first < k; });
    if (it == obj.end() || it->first != key) it = obj.insert(it, {key, JValue()});
    return it->second;
  }
  void push_back(JValue v) {
    kind = Array;
    arr.push_back(std::move(v));
  }
};

// text, `std::setw(4) << json` style: 4-space indentation, one element per line
inline void write_json(const JValue& v, std::ostream& os, int indent = 0) {
  const std::string pad(size_t(indent) + 4, ' '), pad_close(size_t(indent), ' ');
  switch (v.kind) {
    case JValue::Null: os << "null"; break;
    case JValue::Bool: os << (v.b ? "true" : "false"); break;
    case JValue::Int: os << v.i; break;
    case JValue::UInt: os << v.u; break;
    case JValue::Float: os << json_num(v.d); break;
    case JValue::String: os << '"' << json_escape(v.s) << '"'; break;
    case JValue::Array:
      if (v.arr.empty()) {
        os << "[]";
        break;
      }
      os << "[\n";
      for (size_t k = 0; k < v.arr.size(); ++k) {
        os << pad;
        write_json(v.arr[k], os, indent + 4);
        os << (k + 1 < v.arr.size() ? ",\n" : "\n");
      }
      os << pad_close << "]";
      break;
    case JValue::Object:
      if (v.obj.empty()) {
        os << "{}";
        break;
      }
      os << "{\n";
      for (size_t k = 0; k < v.obj.size(); ++k) {
        os << pad << '"' << json_escape(v.obj[k].first) << "\": ";
        write_json(v.obj[k].second, os, indent + 4);
        os << (k + 1 < v.obj.size() ? ",\n" : "\n");
      }
      os << pad_close << "}";
      break;
  }
}

// UBJSON as nlohmann::json::to_ubjson(j, os) writes it (no size / type optimisation): the twin file
// `<log>.ubjson` of the reference (ba_log.cpp:127-145). Integers take the smallest of i/U/I/l/L that
// holds them, doubles are 'D' + 8 bytes big-endian, object keys are size-prefixed without the 'S' marker.
inline void ubjson_bytes(std::ostream& os, uint64_t v, int n) {
  for (int k = n - 1; k >= 0; --k) os.put(char((v >> (8 * k)) & 0xff));
}
inline void ubjson_signed(std::ostream& os, int64_t n) {
  if (n >= -128 && n <= 127) { os.put('i'); ubjson_bytes(os, uint64_t(n), 1); }
  else if (n >= 0 && n <= 255) { os.put('U'); ubjson_bytes(os, uint64_t(n), 1); }
  else if (n >= -32768 && n <= 32767) { os.put('I'); ubjson_bytes(os, uint64_t(n), 2); }
  else if (n >= -2147483648LL && n <= 2147483647LL) { os.put('l'); ubjson_bytes(os, uint64_t(n), 4); }
  else { os.put('L'); ubjson_bytes(os, uint64_t(n), 8); }
}
inline void ubjson_unsigned(std::ostream& os, uint64_t n) {
  if (n <= 127) { os.put('i'); ubjson_bytes(os, n, 1); }
  else if (n <= 255) { os.put('U'); ubjson_bytes(os, n, 1); }
  else if (n <= 32767) { os.put('I'); ubjson_bytes(os, n, 2); }
  else if (n <= 2147483647ULL) { os.put('l'); ubjson_bytes(os, n, 4); }
  else { os.put('L'); ubjson_bytes(os, n, 8); }  // (values above INT64_MAX do not occur in the log)
}
inline void write_ubjson(const JValue& v, std::ostream& os) {
  switch (v.kind) {
    case JValue::Null: os.put('Z'); break;
    case JValue::Bool: os.put(v.b ? 'T' : 'F'); break;
    case JValue::Int: ubjson_signed(os, v.i); break;
    case JValue::UInt: ubjson_unsigned(os, v.u); break;
    case JValue::Float: {
      uint64_t bits;
      static_assert(sizeof(bits) == sizeof(v.d), "IEEE double");
      std::memcpy(&bits, &v.d, sizeof(bits));
      os.put('D');
      ubjson_bytes(os, bits, 8);
      break;
    }
    case JValue::String:
      os.put('S');
      ubjson_unsigned(os, v.s.size());
      os.write(v.s.data(), std::streamsize(v.s.size()));
      break;
    case JValue::Array:
      os.put('[');
      for (const JValue& e : v.arr) write_ubjson(e, os);
      os.put(']');
      break;
    case JValue::Object:
      os.put('{');
      for (const auto& e : v.obj) {
        ubjson_unsigned(os, e.first.size());
        os.write(e.first.data(), std::streamsize(e.first.size()));
        write_ubjson(e.second, os);
      }
      os.put('}');
      break;
  }
}
}  // namespace detail

// the log as one JSON value: an array per BaIteration member, "_type", "_static" (ba_log.cpp:62-115)
inline detail::JValue ba_log_to_json(const SolverSummary& summary, const DatasetSummary& dataset,
                                     const PipelineTimingSummary& timing) {
  using detail::JValue;
  const std::vector<BaIteration> rows = to_ba_iterations(summary);
  JValue j = JValue::object();
#define RBA_LOG_NUM(member)  { JValue& a = j[#member]; a = JValue::array(); for (const BaIteration& r : rows) a.push_back(JValue::number(double(r.member))); }
#define RBA_LOG_INT(member)  { JValue& a = j[#member]; a = JValue::array(); for (const BaIteration& r : rows) a.push_back(JValue::integer(int64_t(r.member))); }
#define RBA_LOG_UINT(member) { JValue& a = j[#member]; a = JValue::array(); for (const BaIteration& r : rows) a.push_back(JValue::unsigned_integer(uint64_t(r.member))); }
#define RBA_LOG_BOOL(member) { JValue& a = j[#member]; a = JValue::array(); for (const BaIteration& r : rows) a.push_back(JValue::boolean(r.member)); }
  RBA_LOG_INT(iteration);
  {
    JValue& a = j["linear_solver_type"];
    a = JValue::array();
    for (const BaIteration& r : rows) a.push_back(JValue::string(r.linear_solver_type));
  }
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
  RBA_LOG_UINT(resident_memory);
  RBA_LOG_UINT(resident_memory_peak);
#undef RBA_LOG_NUM
#undef RBA_LOG_INT
#undef RBA_LOG_UINT
#undef RBA_LOG_BOOL
  j["_type"] = JValue::string("rootba");

  auto stats = [&](const DatasetStats& s) {
    JValue o = JValue::object();
    o["mean"] = JValue::number(s.mean);
    o["min"] = JValue::number(s.min);
    o["max"] = JValue::number(s.max);
    o["stddev"] = JValue::number(s.stddev);
    return o;
  };
  double linear_solver_time = 0, residual_time = 0, jacobian_time = 0;
  int successful = -1, unsuccessful = 0;  // iteration 0 counts as successful in the rows, not in the total
  for (const BaIteration& r : rows) {
    linear_solver_time += r.step_solver_time;
    residual_time += r.residual_evaluation_time;
    jacobian_time += r.jacobian_evaluation_time;
    (r.step_is_successful ? successful : unsuccessful) += 1;
  }
  const double total = timing.load_time + timing.preprocess_time + timing.optimize_time;
  const int n_solves = rows.empty() ? 0 : int(rows.size()) - 1;
  JValue& st = j["_static"];
  st = JValue::object();
  JValue& pi = st["problem_info"];
  pi["type"] = JValue::string(dataset.type);
  pi["input_path"] = JValue::string(dataset.input_path);
  pi["num_cameras"] = JValue::integer(dataset.num_cameras);
  pi["num_landmarks"] = JValue::integer(dataset.num_landmarks);
  pi["num_observations"] = JValue::integer(dataset.num_observations);
  pi["rcs_sparsity"] = JValue::number(dataset.rcs_sparsity);
  pi["per_lm_obs"] = stats(dataset.per_lm_obs);
  pi["per_host_lms"] = stats(dataset.per_host_lms);
  JValue& tm = st["timing"];
  tm["total"] = JValue::number(total);
  tm["load"] = JValue::number(timing.load_time);
  tm["preprocess"] = JValue::number(timing.preprocess_time);
  tm["optimize"] = JValue::number(timing.optimize_time);
  tm["postprocess"] = JValue::number(timing.postprocess_time);
  JValue& so = st["solver"];
  so["solver_type"] = JValue::string(summary.solver_type);
  so["termination_type"] = JValue::string(summary.termination_type == 1 ? "CONVERGENCE" : "NO_CONVERGENCE");
  so["message"] = JValue::string(summary.message);
  so["num_successful_steps"] = JValue::integer(std::max(successful, 0));
  so["num_unsuccessful_steps"] = JValue::integer(unsuccessful);
  so["logging_time_in_seconds"] = JValue::number(0.0);
  so["preprocessor_time_in_seconds"] = JValue::number(summary.preprocessor_time_in_seconds);
  so["minimizer_time_in_seconds"] = JValue::number(summary.minimizer_time_in_seconds);
  so["postprocessor_time_in_seconds"] = JValue::number(summary.postprocessor_time_in_seconds);
  so["total_time_in_seconds"] = JValue::number(summary.total_time_in_seconds);
  so["linear_solver_time_in_seconds"] = JValue::number(linear_solver_time);
  so["num_linear_solves"] = JValue::integer(n_solves);
  so["residual_evaluation_time_in_seconds"] = JValue::number(residual_time);
  so["num_residual_evaluations"] = JValue::integer(int(rows.size()));
  so["jacobian_evaluation_time_in_seconds"] = JValue::number(jacobian_time);
  so["num_jacobian_evaluations"] = JValue::integer(std::max(successful, 0) + (rows.empty() ? 0 : 1));
  so["num_threads_given"] = JValue::integer(0);
  so["num_threads_used"] = JValue::integer(1);
  so["num_threads_available"] = JValue::integer(int(std::thread::hardware_concurrency()));
  so["resident_memory_peak"] = JValue::unsigned_integer(resident_memory_peak_bytes());
  so["initial_cost"] = JValue::number(summary.initial_cost);
  so["final_cost"] = JValue::number(summary.final_cost);
  return j;
}

// BaLog::save_json (ba_log.cpp:62-149): SaveLogFlag JSON and / or UBJSON (`<path without extension>.ubjson`)
enum SaveLogFlag { SAVE_LOG_JSON = 1, SAVE_LOG_UBJSON = 2 };

inline bool save_ba_log(const std::string& path, int flags, const SolverSummary& summary,
                        const DatasetSummary& dataset, const PipelineTimingSummary& timing) {
  if (flags == 0) return true;
  const detail::JValue j = ba_log_to_json(summary, dataset, timing);
  if (flags & SAVE_LOG_JSON) {
    std::ofstream f(path);
    if (!f.is_open()) return false;
    detail::write_json(j, f);
    f << "\n";
    if (!f) return false;
  }
  if (flags & SAVE_LOG_UBJSON) {
    const std::string ub = path.substr(0, path.find_last_of('.')) + ".ubjson";
    std::ofstream f(ub, std::ios_base::binary);
    if (!f.is_open()) return false;
    detail::write_ubjson(j, f);
    if (!f) return false;
  }
  return true;
}

inline bool save_ba_log_json(const std::string& path, const SolverSummary& summary, const DatasetSummary& dataset,
                             const PipelineTimingSummary& timing) {
  return save_ba_log(path, SAVE_LOG_JSON, summary, dataset, timing);
}

}  // namespace rootba_hip
