// oracle/mock_rootba_hip.cpp - a TEST DOUBLE of the C ABI of include/rootba_hip.h, backed by the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY. It exists so that the reference-side binding (integration/rootba/solver/linearizor_hip.hpp)
// can be exercised on a machine WITHOUT a GPU: tests/test_reference_loop_on_hip.py loads this library in place of
// rootba_amd/librootba_hip.so and lets the reference's own LM loop run through the binding. It is not a fallback of
// the product: nothing in rootba_amd/ or include/ knows about it, it is built into oracle/_ref/ only, and the real
// library keeps failing loudly when there is no GPU (tests/test_cabi_cpu.py). Only the entry points the binding calls
// are defined.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

// only the rba_* entry points leave this library (it is loaded with RTLD_GLOBAL): the oracle's own inline /
// template code must not interpose on oracle/liboracle.so in the same process
#pragma GCC visibility push(default)
#include "../include/rootba_hip.h"
#pragma GCC visibility pop
#ifdef _OPENMP
#include <omp.h>  // (declared before the push: the OpenMP runtime's functions are not ours to hide)
#endif
#pragma GCC visibility push(hidden)
#include "rootba_oracle.hpp"
#pragma GCC visibility pop

namespace {
thread_local std::string g_err;
struct Mock {
  int dtype;
  orc::Oracle<float>* f = nullptr;
  orc::Oracle<double>* d = nullptr;
  // Test hook (tests/test_reference_loop_on_hip.py): RBA_MOCK_ZERO_INC_SOLVE=n makes the n-th rba_solve of a handle
  // return a ZERO camera increment - what the real library returns when its PCG ends with x = 0 (|b| = 0, numerical
  // failure at the first iteration) - and the rba_apply that follows report a negative model cost change, so that the
  // reference's LM loop rejects the landmark-only step and calls bal_problem.restore() behind the binding's back.
  int zero_inc_solve = 0, solves = 0;
  bool negate_next_l_diff = false;
};
orc::Options to_orc(const rba_options& o) {
  orc::Options d;
  d.use_householder = o.use_householder;
  d.use_valid_projections_only = o.use_valid_projections_only;
  d.robust_norm = o.robust_norm;
  d.huber_parameter = o.huber_parameter;
  d.jacobi_scaling_eps = o.jacobi_scaling_eps;
  d.preconditioner_type = o.preconditioner_type;
  d.reduction_alg = o.reduction_alg;
  d.power_order = o.power_order;
  d.min_cg_it = o.min_cg_it;
  d.max_cg_it = o.max_cg_it;
  d.eta = o.eta;
  d.optimized_cost = o.optimized_cost;
  d.staged_execution = o.staged_execution;
  d.solver_type = o.solver_type;
  return d;
}
template <class F>
int guarded(F f) {
  try {
    return f();
  } catch (const std::exception& e) {
    g_err = e.what();
    return RBA_ERR_INVALID_ARGUMENT;
  }
}
}  // namespace

extern "C" {
const char* rba_last_error(void) { return g_err.c_str(); }

void rba_default_options(rba_options* o) {
  const orc::Options d;
  std::memset(o, 0, sizeof *o);
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
  o->implicit_q = 1;
  o->solver_type = d.solver_type;
  o->explicit_after = -1;
}

int rba_create(int dtype, int /*device*/, int32_t n_cams, int32_t n_lms, const int64_t* off, const int32_t* cam,
               const void* xy, const rba_options* options, rba_handle* out) {
  return guarded([&] {
    auto* m = new Mock{dtype};
    if (const char* ev = std::getenv("RBA_MOCK_ZERO_INC_SOLVE")) m->zero_inc_solve = std::atoi(ev);
    if (dtype == RBA_F32)
      m->f = new orc::Oracle<float>(n_cams, n_lms, off, cam, static_cast<const float*>(xy), to_orc(*options));
    else if (dtype == RBA_F64)
      m->d = new orc::Oracle<double>(n_cams, n_lms, off, cam, static_cast<const double*>(xy), to_orc(*options));
    else {
      delete m;
      g_err = "mock: dtype";
      return int(RBA_ERR_INVALID_ARGUMENT);
    }
    *out = reinterpret_cast<rba_handle>(m);
    return int(RBA_OK);
  });
}
// (the test double has no devices: a "sharded" handle is the plain one)
int rba_create_sharded(int dtype, int /*n_gpus*/, const int* /*device_ids*/, int32_t n_cams, int32_t n_lms, const int64_t* off,
                       const int32_t* cam, const void* xy, const rba_options* options, rba_handle* out) {
  return rba_create(dtype, 0, n_cams, n_lms, off, cam, xy, options, out);
}
int rba_get_shard_ranges(rba_handle, int* n_ranks_out, int32_t*, int) {
  if (n_ranks_out) *n_ranks_out = 1;
  return RBA_OK;
}
int rba_destroy(rba_handle h) {
  auto* m = reinterpret_cast<Mock*>(h);
  delete m->f;
  delete m->d;
  delete m;
  return RBA_OK;
}
#define MOCK_DISPATCH(EXPR_F, EXPR_D)     \
  auto* m = reinterpret_cast<Mock*>(h);   \
  if (m->f) {                             \
    auto* o = m->f;                       \
    using S = float;                      \
    (void)sizeof(S);                      \
    EXPR_F;                               \
  } else {                                \
    auto* o = m->d;                       \
    using S = double;                     \
    (void)sizeof(S);                      \
    EXPR_D;                               \
  }
int rba_set_state(rba_handle h, const void* cams, const void* lms) {
  MOCK_DISPATCH(({
                  std::copy(static_cast<const S*>(cams), static_cast<const S*>(cams) + o->cams().size(), o->cams().begin());
                  std::copy(static_cast<const S*>(lms), static_cast<const S*>(lms) + o->lms().size(), o->lms().begin());
                }),
                ({
                  std::copy(static_cast<const S*>(cams), static_cast<const S*>(cams) + o->cams().size(), o->cams().begin());
                  std::copy(static_cast<const S*>(lms), static_cast<const S*>(lms) + o->lms().size(), o->lms().begin());
                }))
  return RBA_OK;
}
int rba_backup(rba_handle h) {
  MOCK_DISPATCH(o->backup(), o->backup())
  return RBA_OK;
}
int rba_restore(rba_handle h) {
  MOCK_DISPATCH(o->restore(), o->restore())
  return RBA_OK;
}
int rba_get_state(rba_handle h, void* cams, void* lms) {
  MOCK_DISPATCH(({
                  if (cams) std::copy(o->cams().begin(), o->cams().end(), static_cast<S*>(cams));
                  if (lms) std::copy(o->lms().begin(), o->lms().end(), static_cast<S*>(lms));
                }),
                ({
                  if (cams) std::copy(o->cams().begin(), o->cams().end(), static_cast<S*>(cams));
                  if (lms) std::copy(o->lms().begin(), o->lms().end(), static_cast<S*>(lms));
                }))
  return RBA_OK;
}
int rba_compute_error(rba_handle h, rba_residual_info* out) {
  orc::ResidualInfo ri;
  MOCK_DISPATCH(o->compute_error(ri), o->compute_error(ri))
  out->all_num_obs = ri.all.num_obs;
  out->all_error = ri.all.error;
  out->all_residual_sum = ri.all.residual_sum;
  out->valid_num_obs = ri.valid.num_obs;
  out->valid_error = ri.valid.error;
  out->valid_residual_sum = ri.valid.residual_sum;
  out->is_numerically_valid = ri.is_numerically_valid;
  return RBA_OK;
}
int rba_linearize(rba_handle h, void* /*jp_diag2_out*/) {
  bool ok = false;
  MOCK_DISPATCH(ok = o->linearize(), ok = o->linearize())
  return ok ? RBA_OK : RBA_NUMERICAL_FAILURE;
}
int rba_solve(rba_handle h, double lambda, void* inc_out, rba_cg_summary* cg) {
  orc::CgSummary s;
  MOCK_DISPATCH(({
                  const auto inc = o->solve(S(lambda), &s);
                  std::copy(inc.begin(), inc.end(), static_cast<S*>(inc_out));
                }),
                ({
                  const auto inc = o->solve(S(lambda), &s);
                  std::copy(inc.begin(), inc.end(), static_cast<S*>(inc_out));
                }))
  if (cg) {
    cg->termination_type = s.termination_type;
    cg->num_iterations = s.num_iterations;
  }
  {
    auto* mk = reinterpret_cast<Mock*>(h);
    if (++mk->solves == mk->zero_inc_solve) {
      const size_t n = size_t(9) * (mk->f ? mk->f->n_cams() : mk->d->n_cams());
      std::memset(inc_out, 0, n * (mk->f ? sizeof(float) : sizeof(double)));
      mk->negate_next_l_diff = true;
    }
  }
  return RBA_OK;
}
int rba_apply(rba_handle h, const void* inc, double* l_diff_out) {
  double l = 0;
  // (hooked step: the cameras stay BIT-identical - a retraction by zero may re-normalise the quaternion's last bit, and
  //  the case under test is the one where it does not)
  const bool keep_cams = reinterpret_cast<Mock*>(h)->negate_next_l_diff;
  MOCK_DISPATCH(({
                  std::vector<S> v(static_cast<const S*>(inc), static_cast<const S*>(inc) + size_t(9) * o->n_cams());
                  const auto cams = o->cams();
                  l = double(o->apply(std::move(v)));
                  if (keep_cams) std::copy(cams.begin(), cams.end(), o->cams().begin());
                }),
                ({
                  std::vector<S> v(static_cast<const S*>(inc), static_cast<const S*>(inc) + size_t(9) * o->n_cams());
                  const auto cams = o->cams();
                  l = double(o->apply(std::move(v)));
                  if (keep_cams) std::copy(cams.begin(), cams.end(), o->cams().begin());
                }))
  {
    auto* mk = reinterpret_cast<Mock*>(h);
    if (mk->negate_next_l_diff) l = -std::abs(l) - 1e-30;
    mk->negate_next_l_diff = false;
  }
  *l_diff_out = l;
  return std::isfinite(l) ? RBA_OK : RBA_NUMERICAL_FAILURE;
}
}  // extern "C"
