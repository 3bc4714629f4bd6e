// integration/rootba/solver/linearizor_hip.hpp - THE REFERENCE-SIDE BINDING of librootba_hip.so
// (as real code): a sixth `Linearizor<Scalar>` for NikolausDemmel/rootba that forwards the five
// virtual calls of src/rootba/solver/linearizor.hpp:56-82 to the C ABI of include/rootba_hip.h.
//
// This file is written against the REFERENCE'S headers (BalProblem, SolverOptions, LinearizorBase, Eigen types); a
// maintainer drops it into src/rootba/solver/, adds one enum entry + one `case` to the factory
// (integration/linearizor_factory_hip.cpp shows the case), and links -lrootba_hip. Nothing else in the reference
// changes: its CLI, loader, options, LM loop (optimize_lm_ours) and logging run as they are.
//
// It is compiled and exercised in this repository by oracle/build_ref.sh (against the reference tree + the
// third-party stand-ins of oracle/ref_shims), see tests/test_reference_loop_on_hip.py:
//   on the GPU: the reference's own LM loop drives the HIP library through this class;
//   on CPU:     the same object code runs against a test double of the C ABI (oracle/mock_rootba_hip.cpp).
#pragma once

#include <cstdlib>

#include <limits>
#include <string>
#include <type_traits>
#include <vector>

#include "rootba/solver/host_state_sync.hpp"
#include "rootba/solver/linearizor_base.hpp"
#include "rootba/util/time_utils.hpp"
#include "rootba_hip.h"

namespace rootba {

template <class Scalar_>
class LinearizorHIP : public LinearizorBase<Scalar_>, public HostStateSync {
 public:
  using Scalar = Scalar_;
  using Base = LinearizorBase<Scalar>;
  using VecX = typename Base::VecX;

  // `n_gpus` > 1: devices device .. device + n_gpus - 1 of THIS process behind the one handle (rba_create_sharded: the
  // library shards the landmarks itself). The reference is one process with one Linearizor for the whole problem
  // (linearizor.cpp:133-150), so this is how its unchanged driver reaches several GPUs. A maintainer would feed it from
  // a SolverOptions field (e.g. `num_gpus` beside `num_threads`); here: the constructor argument or ROOTBA_HIP_GPUS.
  LinearizorHIP(BalProblem<Scalar>& bal_problem, const SolverOptions& options, SolverSummary* summary = nullptr,
                int device = 0, int n_gpus = 0)
      : Base(bal_problem, options, summary) {
    if (n_gpus <= 0) {
      const char* ev = std::getenv("ROOTBA_HIP_GPUS");
      n_gpus = ev ? std::max(1, std::atoi(ev)) : 1;
    }
    // CSR topology from the std::map of every landmark (ascending camera index = map order,
    // landmark_block_dynamic.hpp:52-54)
    std::vector<int64_t> off(1, 0);
    std::vector<int32_t> cam;
    std::vector<Scalar> xy;
    for (const auto& lm : bal_problem.landmarks()) {
      for (const auto& [cam_idx, obs] : lm.obs) {
        cam.push_back(cam_idx);
        xy.push_back(obs.pos(0));
        xy.push_back(obs.pos(1));
      }
      off.push_back(int64_t(cam.size()));
    }
    // SolverOptions -> rba_options: the mapping of LinearizorQR's constructor (linearizor_qr.cpp:58-68) and of
    // LinearizorBase::pcg (linearizor_base.cpp:87-93)
    rba_options o;
    rba_default_options(&o);
    o.use_householder = options.use_householder_marginalization;
    o.use_valid_projections_only = options.use_projection_validity_check();
    o.robust_norm = options.residual.robust_norm == BalResidualOptions::RobustNorm::HUBER ? 1 : 0;
    o.huber_parameter = options.residual.huber_parameter;
    o.jacobi_scaling_eps = options.jacobi_scaling_epsilon;
    o.preconditioner_type =
        options.preconditioner_type == SolverOptions::PreconditionerType::JACOBI         ? 0
        : options.preconditioner_type == SolverOptions::PreconditionerType::SCHUR_JACOBI ? 1
                                                                                         : 2;  // power series
    o.reduction_alg = options.reduction_alg;
    o.power_order = options.power_order;
    o.min_cg_it = options.min_linear_solver_iterations;
    o.max_cg_it = options.max_linear_solver_iterations;
    o.eta = options.eta;
    o.optimized_cost = int(options.optimized_cost);
    o.staged_execution = options.staged_execution;
    // (the LM-loop fields of rba_options - trust region, vee, tolerances - are used by rba_optimize_lm only; here
    //  the reference's own loop runs)
    const int dt = std::is_same<Scalar, float>::value ? RBA_F32 : RBA_F64;
    int st;
    if (n_gpus > 1) {
      std::vector<int> ids(static_cast<size_t>(n_gpus));
      for (int i = 0; i < n_gpus; ++i) ids[static_cast<size_t>(i)] = device + i;
      st = rba_create_sharded(dt, n_gpus, ids.data(), bal_problem.num_cameras(), bal_problem.num_landmarks(), off.data(),
                              cam.data(), xy.data(), &o, &h_);
    } else {
      st = rba_create(dt, device, bal_problem.num_cameras(), bal_problem.num_landmarks(), off.data(), cam.data(), xy.data(), &o,
                      &h_);
    }
    CHECK(st == RBA_OK) << "rba_create: " << rba_last_error();
    cams_.resize(size_t(10) * bal_problem.num_cameras());
    lms_.resize(size_t(3) * bal_problem.num_landmarks());
    upload();
  }
  ~LinearizorHIP() override {
    if (h_) {
      // optimize_lm_ours destroys its linearizor before it returns: BalProblem receives the final state here
      ensure_device_state();
      sync_host();
      rba_destroy(h_);
    }
  }

  // ---- state protocol ---------------------------------------------------------------------------------------
  // Round 2 treated BalProblem as the source of truth at every call: three host loops over all landmarks plus the
  // PCIe copies per LM iteration (at venice size an order of magnitude more than the GPU work). Now the DEVICE holds
  // the state between calls:
  //  * cameras (n_c x 10 scalars) go back to BalProblem after every apply(), landmarks only on demand (sync_host():
  //    the destructor, i.e. the end of optimize_lm_ours, or an explicit call);
  //  * the reference's LM loop brackets a step with bal_problem.backup() ... bal_problem.restore()
  //    (bal_bundle_adjustment.cpp:401, 509): apply() takes the matching device-side backup (rba_backup), and a
  //    restore() is RECOGNISED at the next call by the cameras of BalProblem being those of the backup again, upon
  //    which the device restores too (rba_restore) - no upload;
  //  * that recognition needs the step to have CHANGED the cameras. A step that leaves every camera parameter
  //    bit-identical (a zero camera increment: the PCG ended with x = 0 - |b| = 0 or a numerical failure at its first
  //    iteration - and only landmarks moved) makes "restored" and "not restored" look alike from here: BalProblem's
  //    cameras equal the backup either way and its landmarks are stale either way. apply() then does not try to infer
  //    anything: it brings the moved landmarks to BalProblem at once, and until the next step every call compares
  //    BalProblem's LANDMARKS with that copy as well: if they differ the driver has restored (whatever its backup held
  //    - landmarks that were stale already when it was taken, possibly - differs from the moved ones), and the device
  //    restores its own twin of the backup. BalProblem's restored landmarks are never uploaded: they are only as
  //    complete as they were when the driver backed them up;
  //  * any other change of the cameras (the caller edited the problem) falls back to a full upload, after the
  //    pending landmark download so that BalProblem really is complete; callers that edit LANDMARKS announce it with
  //    host_state_changed().
  void sync_host() override {
    if (!host_lms_stale_) return;
    CHECK(rba_get_state(h_, cams_.data(), lms_.data()) == RBA_OK) << rba_last_error();
    for (int l = 0; l < bal_problem_.num_landmarks(); ++l)
      for (int k = 0; k < 3; ++k) bal_problem_.landmarks()[l].p_w(k) = lms_[size_t(3) * l + k];
    host_lms_stale_ = false;
  }
  void host_state_changed() override {
    host_lms_stale_ = false;
    upload();
  }

  // LinearizorBase::compute_error (linearizor_base.cpp:60-68)
  void compute_error(ResidualInfo& ri) override {
    Timer<> timer;
    ensure_device_state();  // (the driver may have called bal_problem.restore())
    rba_residual_info r;
    CHECK(rba_compute_error(h_, &r) == RBA_OK) << rba_last_error();
    ri.all.num_obs = r.all_num_obs;
    ri.all.error = r.all_error;
    ri.all.residual_sum = r.all_residual_sum;
    ri.valid.num_obs = r.valid_num_obs;
    ri.valid.error = r.valid_error;
    ri.valid.residual_sum = r.valid_residual_sum;
    ri.is_numerically_valid = r.is_numerically_valid != 0;
    if (it_summary_) it_summary_->residual_evaluation_time_in_seconds += timer.elapsed();
    if (summary_) summary_->num_residual_evaluations += 1;
  }

  // LinearizorQR::linearize (linearizor_qr.cpp:78-138)
  void linearize() override {
    Timer<> timer;
    ensure_device_state();
    const int st = rba_linearize(h_, nullptr);
    CHECK(st == RBA_OK) << "did not expect numerical failure during linearization (" << rba_last_error() << ")";
    if (it_summary_) it_summary_->stage1_time_in_seconds = timer.elapsed();
    if (summary_) summary_->num_jacobian_evaluations += 1;
  }

  // LinearizorQR::solve (linearizor_qr.cpp:141-265)
  VecX solve(Scalar lambda) override {
    Timer<> timer;
    VecX inc(9 * bal_problem_.num_cameras());
    rba_cg_summary cg;
    const int st = rba_solve(h_, double(lambda), inc.data(), &cg);
    CHECK(st >= 0) << rba_last_error();
    if (it_summary_) {
      it_summary_->solve_reduced_system_time_in_seconds = timer.elapsed();
      it_summary_->linear_solver_iterations = cg.num_iterations;
      it_summary_->linear_solver_message =
          cg.termination_type == 1 ? "Convergence." : cg.termination_type == 2 ? "Numerical failure." : "No convergence.";
      it_summary_->linear_solver_type = "bal_qr_hip";
    }
    if (summary_) summary_->num_linear_solves += 1;
    return inc;
  }

  // LinearizorQR::apply (linearizor_qr.cpp:268-291). The driver has called bal_problem.backup() before and calls
  // bal_problem.restore() afterwards when it rejects the step: the host problem is the source of truth, so the
  // state goes up before the update and comes back after it.
  Scalar apply(VecX&& inc) override {
    Timer<> timer;
    ensure_device_state();
    // the device-side twin of the bal_problem.backup() the driver has just made
    CHECK(rba_backup(h_) == RBA_OK) << rba_last_error();
    host_cams_backup_ = host_cams_;
    stale_at_backup_ = host_lms_stale_;
    have_backup_ = true;
    double l_diff = 0;
    const int st = rba_apply(h_, inc.data(), &l_diff);
    CHECK(st >= 0) << rba_last_error();
    if (it_summary_) it_summary_->back_substitution_time_in_seconds = timer.elapsed();
    if (st != RBA_OK) {
      // numerical failure: the library left cameras untouched and may have moved landmarks; back to the backup
      CHECK(rba_restore(h_) == RBA_OK) << rba_last_error();
      return std::numeric_limits<Scalar>::quiet_NaN();
    }
    download_cameras();
    host_lms_stale_ = true;
    if (host_cams_ == host_cams_backup_) {
      // the step moved no camera: a bal_problem.restore() would be invisible in BalProblem's cameras (see "state
      // protocol"). BalProblem gets the moved landmarks now; ensure_device_state() compares them from here on.
      sync_host();
      lms_tracked_ = true;
    } else {
      lms_tracked_ = false;
    }
    return Scalar(l_diff);
  }

 private:
  using Base::bal_problem_;
  using Base::it_summary_;
  using Base::summary_;

  // what BalProblem's cameras say right now (Camera::params(), bal_problem.hpp:84-89: qx qy qz qw tx ty tz f k1 k2)
  void read_host_cameras(std::vector<Scalar>& out) const {
    // (the pieces of Camera::params() read in place: params() itself returns a heap-allocated VecX per camera, and this
    //  probe runs four times per LM iteration)
    out.resize(size_t(10) * bal_problem_.num_cameras());
    for (int c = 0; c < bal_problem_.num_cameras(); ++c) {
      const auto& cam = bal_problem_.cameras()[c];
      const auto pose = cam.T_c_w.params();          // qx qy qz qw tx ty tz
      const auto intr = cam.intrinsics.getParam();   // f k1 k2
      Scalar* o = out.data() + size_t(10) * c;
      for (int k = 0; k < 7; ++k) o[k] = pose(k);
      for (int k = 0; k < 3; ++k) o[7 + k] = intr(k);
    }
  }
  // BalProblem -> device, everything
  void upload() {
    read_host_cameras(host_cams_);
    cams_ = host_cams_;
    for (int l = 0; l < bal_problem_.num_landmarks(); ++l)
      for (int k = 0; k < 3; ++k) lms_[size_t(3) * l + k] = bal_problem_.landmarks()[l].p_w(k);
    CHECK(rba_set_state(h_, cams_.data(), lms_.data()) == RBA_OK) << rba_last_error();
    have_backup_ = false;
    lms_tracked_ = false;
  }
  // device cameras -> BalProblem (Camera::from_params, bal_problem.hpp:91-95); `host_cams_` then records what
  // BalProblem REPORTS (from_params normalises the quaternion), the reference value of the change detection
  void download_cameras() {
    CHECK(rba_get_state(h_, cams_.data(), nullptr) == RBA_OK) << rba_last_error();
    VecX p(10);
    for (int c = 0; c < bal_problem_.num_cameras(); ++c) {
      for (int k = 0; k < 10; ++k) p(k) = cams_[size_t(10) * c + k];
      bal_problem_.cameras()[c].from_params(p);
    }
    read_host_cameras(host_cams_);
  }
  // make the device hold the state BalProblem describes (see "state protocol" above)
  void ensure_device_state() {
    read_host_cameras(probe_);
    const bool cams_same = probe_ == host_cams_;
    if (cams_same && !lms_tracked_) return;               // nothing happened on the host side (the common case)
    if (cams_same && host_landmarks_are(lms_)) return;    // ... after a step that moved no camera the landmarks decide
    if (have_backup_ && probe_ == host_cams_backup_) {
      // bal_problem.restore() after a rejected step (cameras back at the backup; after a landmark-only step: cameras
      // unchanged, landmarks no longer the moved ones): the device restores its twin of that backup
      CHECK(rba_restore(h_) == RBA_OK) << rba_last_error();
      host_cams_ = host_cams_backup_;
      host_lms_stale_ = stale_at_backup_;
      lms_tracked_ = false;
      return;
    }
    // the caller changed the cameras: complete BalProblem first (landmarks it has not seen yet), then upload
    sync_host();
    upload();
  }
  bool host_landmarks_are(const std::vector<Scalar>& v) const {
    for (int l = 0; l < bal_problem_.num_landmarks(); ++l)
      for (int k = 0; k < 3; ++k)
        if (!(bal_problem_.landmarks()[l].p_w(k) == v[size_t(3) * l + k])) return false;
    return true;
  }

  rba_handle h_ = nullptr;
  std::vector<Scalar> cams_, lms_;              // transfer buffers
  std::vector<Scalar> host_cams_, host_cams_backup_, probe_;  // BalProblem's cameras as last seen / at the backup
  bool host_lms_stale_ = false;  // the device holds newer landmarks than BalProblem
  bool stale_at_backup_ = false, have_backup_ = false;
  // the last step moved no camera: BalProblem's landmarks were completed and `lms_` is the copy of what the device
  // holds - compared at every call until a step moves a camera again
  bool lms_tracked_ = false;
};

}  // namespace rootba
