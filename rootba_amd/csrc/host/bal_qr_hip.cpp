// bal_qr_hip — command-line entry mirroring the reference's `bal_qr`
// (reference src/app/bal_qr.cpp:44-115): load a BAL problem, run the square-root
// solver (here: the MI355X-native library behind include/rootba_hip.h), write
// ba_log.json. Flags keep the reference's names (docs/Configuration.md:45-259),
// restricted to the ones that reach the hot path; the TOML/clipp machinery of
// the reference is out of scope.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <fstream>
#include <random>
#include <string>

#include "ba_log.hpp"
#include "linearizor_hip.hpp"

using namespace rootba_hip;

namespace {
void usage() {
  std::puts(
      "Solve BAL problem with the MI355X-native Square Root solver.\n"
      "usage: bal_qr_hip --input <bal file> [options]\n"
      "  --input <path>                         BAL text file\n"
      "  --[no-]normalize, --normalization-scale <s>\n"
      "  --rotation-sigma <s> --translation-sigma <s> --point-sigma <s> --random-seed <n>\n"
      "  --init-depth-threshold <z>\n"
      "  --max-num-iterations <n>               (default 20)\n"
      "  --[no-]use-double                      (default double)\n"
      "  --mixed-precision                      (with --use-double: double state and costs, float linear algebra)\n"
      "  --[no-]staged-execution                (default staged; unstaged also measures the sub-stage timers)\n"
      "  --preconditioner-type JACOBI|SCHUR_JACOBI|POWER_SCHUR_COMPLEMENT  --power-order <m>\n"
      "  --robust-norm NONE|HUBER --huber-parameter <t>\n"
      "  --optimized-cost ERROR|ERROR_VALID|ERROR_VALID_AVG\n"
      "  --eta <e> --max-linear-solver-iterations <n> --function-tolerance <t>\n"
      "  --jacobi-scaling-epsilon <e> --log-path <ba_log.json> --device <n>\n"
      "  --gpus <N>   shard the landmarks over devices n .. n + N - 1 of this process (rba_create_sharded)\n"
      "  --solver-type SQUARE_ROOT|SCHUR_COMPLEMENT   (default SQUARE_ROOT)\n"
      "  --save-log-flags <JSON,UBJSON>         (default JSON; UBJSON writes <log>.ubjson next to it)\n"
      "  --input-type <AUTO|ROOTBA|BAL|BUNDLER> AUTO: '*.cereal' = rootba problem cache, '*bundle*' = Bundler, else BAL text\n"
      "  --[no-]save-output, --output-optimized-path <p>   write the optimised problem as a .cereal cache (default optimized.cereal)\n"
      "                                         (the written cache round-trips through THIS loader; it has not been read back by\n"
      "                                          the reference's cereal / basalt-headers serialisers, which are not available here)\n"
      "  --dry-run                              load + preprocess only, print problem statistics\n"
      "  --dump-options                         print the solver options this command line selects (JSON) and exit");
}

static int g_save_log_flags = rootba_hip::SAVE_LOG_JSON;  // BaLogOptions::save_log_flags (ba_log_options.hpp:48-50)

template <class Scalar>
int run(const BalDatasetOptions& ds, const SolverOptions& so, const std::string& log_path, bool dry_run, int device, int n_gpus) {
  const auto t_load = std::chrono::steady_clock::now();
  double load_only_seconds = 0, preprocess_seconds = 0;
  auto prob = load_normalized_bal_problem<Scalar>(ds, &load_only_seconds, &preprocess_seconds);
  const double load_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_load).count();
  double sx = 0, sy = 0, sz = 0;
  for (const auto& p : prob.points) {
    sx += p[0];
    sy += p[1];
    sz += p[2];
  }
  // order-sensitive inside a landmark (position weight), so it also pins the camera order
  double obs_checksum = 0;
  for (int l = 0; l < prob.num_landmarks(); ++l)
    for (int64_t o = prob.lm_off[l]; o < prob.lm_off[l + 1]; ++o)
      obs_checksum += double(o - prob.lm_off[l] + 1) * (double(prob.obs_cam[o] + 1) * double(prob.obs_xy[2 * o]) + double(prob.obs_xy[2 * o + 1]));
  std::printf("Loaded BAL problem (%d cams, %d lms, %lld obs) from '%s' in %.3fs\n", prob.num_cameras(),
              prob.num_landmarks(), static_cast<long long>(prob.num_observations()), ds.input.c_str(), load_seconds);
  if (dry_run) {
    // (--dry-run --save-output writes the PREPROCESSED problem: the cache the reference's tools produce)
    if (ds.save_output && !prob.save_rootba(ds.output_optimized_path)) {
      std::fprintf(stderr, "Failed to save %s.\n", ds.output_optimized_path.c_str());
      return 2;
    }
    std::printf("{\"num_cameras\": %d, \"num_landmarks\": %d, \"num_observations\": %lld, \"load_seconds\": %.6f, "
                "\"rcs_sparsity\": %.12e, \"obs_checksum\": %.12e, "
                "\"landmark_sum\": [%.12e, %.12e, %.12e], \"cam0\": [%.12e, %.12e, %.12e, %.12e, %.12e, %.12e, %.12e]}\n",
                prob.num_cameras(), prob.num_landmarks(), static_cast<long long>(prob.num_observations()), load_seconds,
                prob.compute_rcs_sparsity(), obs_checksum, sx, sy, sz,
                double(prob.cameras[0][0]), double(prob.cameras[0][1]), double(prob.cameras[0][2]),
                double(prob.cameras[0][3]), double(prob.cameras[0][4]), double(prob.cameras[0][5]),
                double(prob.cameras[0][6]));
    return 0;
  }
  SolverSummary summary;
  const auto t_opt = std::chrono::steady_clock::now();
  bundle_adjust_manual(prob, so, &summary, device, n_gpus);
  PipelineTimingSummary timing;
  timing.load_time = load_only_seconds;
  timing.preprocess_time = preprocess_seconds;
  timing.optimize_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_opt).count();
  // BalProblem::postprocress (bal_problem.cpp:556-568): the optimised problem as a `.cereal` cache
  if (ds.save_output && !prob.save_rootba(ds.output_optimized_path)) {
    std::fprintf(stderr, "Failed to save %s.\n", ds.output_optimized_path.c_str());
    return 2;
  }
  // ba_log.json in the reference's layout (src/rootba/bal/ba_log.cpp:62-149)
  if (!save_ba_log(log_path, g_save_log_flags, summary, summarize_dataset(prob, ds.input), timing)) {
    std::fprintf(stderr, "Could not save BA log to %s.\n", log_path.c_str());
    return 2;
  }
  return 0;
}
}  // namespace

// parse_double against strtod on random doubles printed in the formats BAL writers use
static int self_test_parser(long n) {
  std::mt19937_64 rng(38401);
  long bad = 0, tokens = 0;
  char buf[512];
  const char* fmts[] = {"%.16e", "%.17g", "%.9e", "%.6f", "%.18e", "%.15e", "%g", "%+.16e"};
  for (long it = 0; it < n; ++it) {
    const uint64_t bits = rng();
    double x;
    if (it % 3 == 0) {
      std::memcpy(&x, &bits, 8);
      if (!std::isfinite(x)) continue;
    } else {
      x = std::ldexp(double(int64_t(bits)), -int(rng() % 90));
    }
    for (const char* f : fmts) {
      const int len = std::snprintf(buf, sizeof buf, f, x);
      if (len <= 0 || len >= int(sizeof buf)) continue;
      double got = 0;
      const double ref = std::strtod(buf, nullptr);
      const char* e = rootba_hip::detail::parse_double(buf, buf + len, got);
      ++tokens;
      if (e != buf + len || std::memcmp(&ref, &got, 8) != 0) {
        if (bad++ < 5) std::fprintf(stderr, "MISMATCH '%s': strtod %.17g parse_double %.17g\n", buf, ref, got);
      }
    }
  }
  for (const char* b : {"", "-", ".", "1e", "abc", "1.5x", "e5"}) {
    double g;
    const char* e = rootba_hip::detail::parse_double(b, b + std::strlen(b), g);
    if (e == b + std::strlen(b)) {
      std::fprintf(stderr, "accepted malformed token '%s'\n", b);
      ++bad;
    }
  }
  std::printf("{\"tokens\": %ld, \"mismatches\": %ld}\n", tokens, bad);
  return bad == 0 ? 0 : 3;
}

// ba_log.json writer on a hand-made summary (success, reject, success): no GPU needed
static int self_test_log(const std::string& path) {
  SolverSummary summary;
  summary.message = "Solver did not converge after maximum number of iterations";
  const double costs[4] = {100.0, 60.0, 75.0, 50.0};
  const bool ok[4] = {true, true, false, true};
  for (int i = 0; i < 4; ++i) {
    IterationSummary it;
    it.iteration = i;
    it.step_is_valid = true;
    it.step_is_successful = ok[i];
    it.cost.all = {1000, costs[i], 2000.0 + i};
    it.cost.valid = {990 + i, costs[i] - 1.0, 1900.0 + i};
    if (i > 0) it.prev_cost = summary.iterations.back().cost;
    it.step_norm = 0.5 * i;
    it.relative_decrease = ok[i] ? 0.9 : -0.3;
    it.trust_region_radius = 1e4 * (i + 1);
    it.linear_solver_iterations = 10 * i;
    it.iteration_time_in_seconds = 0.01;
    it.cumulative_time_in_seconds = 0.01 * (i + 1);
    it.stage1_time_in_seconds = 0.001;
    it.stage2_time_in_seconds = 0.002;
    it.solve_reduced_system_time_in_seconds = 0.003;
    it.back_substitution_time_in_seconds = 0.004;
    summary.iterations.push_back(it);
  }
  summary.initial_cost = costs[0];
  summary.final_cost = costs[3];
  DatasetSummary dataset;
  dataset.input_path = "self \"test\"";
  dataset.num_cameras = 3;
  dataset.num_landmarks = 5;
  dataset.num_observations = 12;
  PipelineTimingSummary timing;
  timing.load_time = 1;
  timing.preprocess_time = 2;
  timing.optimize_time = 3;
  return save_ba_log(path, SAVE_LOG_JSON | SAVE_LOG_UBJSON, summary, dataset, timing) ? 0 : 2;
}

int main(int argc, char** argv) {
  BalDatasetOptions ds;
  SolverOptions so;
  std::string log_path = "ba_log.json";
  bool dry_run = false, dump_options = false;
  int device = 0, n_gpus = 1;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&]() -> std::string {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "missing value for %s\n", a.c_str());
        std::exit(1);
      }
      return argv[++i];
    };
    if (a == "--help" || a == "-h") { usage(); return 0; }
    else if (a == "--input") ds.input = val();
    else if (a == "--input-type") {
      const std::string v = val();
      if (v == "AUTO") ds.input_type = BalDatasetOptions::DatasetType::AUTO;
      else if (v == "ROOTBA") ds.input_type = BalDatasetOptions::DatasetType::ROOTBA;
      else if (v == "BAL") ds.input_type = BalDatasetOptions::DatasetType::BAL;
      else if (v == "BUNDLER") ds.input_type = BalDatasetOptions::DatasetType::BUNDLER;
      else { std::fprintf(stderr, "input type %s not implemented (AUTO, ROOTBA, BAL, BUNDLER)\n", v.c_str()); return 1; }
    }
    else if (a == "--save-output") ds.save_output = true;
    else if (a == "--no-save-output") ds.save_output = false;
    else if (a == "--output-optimized-path") ds.output_optimized_path = val();
    else if (a == "--normalize") ds.normalize = true;
    else if (a == "--no-normalize") ds.normalize = false;
    else if (a == "--normalization-scale") ds.normalization_scale = std::stod(val());
    else if (a == "--rotation-sigma") ds.rotation_sigma = std::stod(val());
    else if (a == "--translation-sigma") ds.translation_sigma = std::stod(val());
    else if (a == "--point-sigma") ds.point_sigma = std::stod(val());
    else if (a == "--random-seed") ds.random_seed = std::stoi(val());
    else if (a == "--init-depth-threshold") ds.init_depth_threshold = std::stod(val());
    else if (a == "--max-num-iterations") so.max_num_iterations = std::stoi(val());
    else if (a == "--mixed-precision") so.mixed_precision = true;
    else if (a == "--staged-execution") so.staged_execution = true;
    else if (a == "--no-staged-execution") so.staged_execution = false;
    else if (a == "--use-double") so.use_double = true;
    else if (a == "--no-use-double") so.use_double = false;
    else if (a == "--preconditioner-type") {
      const std::string v = val();
      if (v == "JACOBI") so.preconditioner_type = SolverOptions::PreconditionerType::JACOBI;
      else if (v == "SCHUR_JACOBI") so.preconditioner_type = SolverOptions::PreconditionerType::SCHUR_JACOBI;
      else if (v == "POWER_SCHUR_COMPLEMENT") so.preconditioner_type = SolverOptions::PreconditionerType::POWER_SCHUR_COMPLEMENT;
      else { std::fprintf(stderr, "preconditioner %s not implemented\n", v.c_str()); return 1; }
    } else if (a == "--solver-type") {
      const std::string v = val();
      if (v == "SQUARE_ROOT") so.solver_type = SolverOptions::SolverType::SQUARE_ROOT;
      else if (v == "SCHUR_COMPLEMENT") so.solver_type = SolverOptions::SolverType::SCHUR_COMPLEMENT;
      else { std::fprintf(stderr, "solver type %s not implemented\n", v.c_str()); return 1; }
    } else if (a == "--robust-norm") {
      const std::string v = val();
      so.residual.robust_norm = v == "HUBER" ? BalResidualOptions::RobustNorm::HUBER : BalResidualOptions::RobustNorm::NONE;
    } else if (a == "--huber-parameter") so.residual.huber_parameter = std::stod(val());
    else if (a == "--optimized-cost") {
      const std::string v = val();
      so.optimized_cost = v == "ERROR_VALID" ? SolverOptions::OptimizedCost::ERROR_VALID
                          : v == "ERROR_VALID_AVG" ? SolverOptions::OptimizedCost::ERROR_VALID_AVG
                                                   : SolverOptions::OptimizedCost::ERROR;
    } else if (a == "--eta") so.eta = std::stod(val());
    else if (a == "--max-linear-solver-iterations") so.max_linear_solver_iterations = std::stoi(val());
    else if (a == "--function-tolerance") so.function_tolerance = std::stod(val());
    else if (a == "--jacobi-scaling-epsilon") so.jacobi_scaling_epsilon = std::stod(val());
    else if (a == "--log-path") log_path = val();
    else if (a == "--save-log-flags") {
      // comma-separated subset of JSON, UBJSON (empty string: save nothing)
      const std::string v = val();
      g_save_log_flags = 0;
      size_t p0 = 0;
      while (p0 <= v.size() && !v.empty()) {
        const size_t p1 = v.find(',', p0);
        const std::string t = v.substr(p0, p1 == std::string::npos ? std::string::npos : p1 - p0);
        if (t == "JSON") g_save_log_flags |= rootba_hip::SAVE_LOG_JSON;
        else if (t == "UBJSON") g_save_log_flags |= rootba_hip::SAVE_LOG_UBJSON;
        else {
          std::fprintf(stderr, "unknown --save-log-flags entry '%s' (JSON, UBJSON)\n", t.c_str());
          return 1;
        }
        if (p1 == std::string::npos) break;
        p0 = p1 + 1;
      }
    }
    else if (a == "--device") device = std::stoi(val());
    else if (a == "--gpus") n_gpus = std::stoi(val());
    else if (a == "--dry-run") dry_run = true;
    else if (a == "--dump-options") dump_options = true;
    else if (a == "--self-test-parser") return self_test_parser(std::stol(val()));
    else if (a == "--self-test-log") return self_test_log(val());
    else if (a == "--implicit-q") so.implicit_q = true;
    else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); usage(); return 1; }
  }
  if (dump_options) {
    // the rba_options this command line hands to rba_create (SolverOptions::to_rba), one JSON object
    const rba_options o = so.to_rba();
    std::printf("{\"use_householder\": %d, \"use_valid_projections_only\": %d, \"robust_norm\": %d, \"huber_parameter\": %.17g, "
                "\"jacobi_scaling_eps\": %.17g, \"preconditioner_type\": %d, \"reduction_alg\": %d, \"power_order\": %d, "
                "\"min_cg_it\": %d, \"max_cg_it\": %d, \"eta\": %.17g, \"max_num_iterations\": %d, "
                "\"min_relative_decrease\": %.17g, \"initial_trust_region_radius\": %.17g, \"min_trust_region_radius\": %.17g, "
                "\"max_trust_region_radius\": %.17g, \"function_tolerance\": %.17g, \"initial_vee\": %.17g, \"vee_factor\": %.17g, "
                "\"optimized_cost\": %d, \"staged_execution\": %d, \"solver_type\": %d, \"use_double\": %d}\n",
                o.use_householder, o.use_valid_projections_only, o.robust_norm, o.huber_parameter, o.jacobi_scaling_eps,
                o.preconditioner_type, o.reduction_alg, o.power_order, o.min_cg_it, o.max_cg_it, o.eta, o.max_num_iterations,
                o.min_relative_decrease, o.initial_trust_region_radius, o.min_trust_region_radius, o.max_trust_region_radius,
                o.function_tolerance, o.initial_vee, o.vee_factor, o.optimized_cost, o.staged_execution, o.solver_type,
                int(so.use_double));
    return 0;
  }
  if (ds.input.empty()) { usage(); return 1; }
  try {
    return so.use_double ? run<double>(ds, so, log_path, dry_run, device, n_gpus) : run<float>(ds, so, log_path, dry_run, device, n_gpus);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "FATAL: %s\n", e.what());
    return 2;
  }
}
